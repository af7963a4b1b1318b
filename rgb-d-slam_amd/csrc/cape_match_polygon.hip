// "Next" row N2 with the reference's own area measure: MapPlane::find_matches (reference
// src/map_management/map_features/map_primitive.cpp:91-161, driven by feature_map.hpp:647-670) between CONSECUTIVE frames of a
// batch, on the boundary polygons built by cape_build_polygons -- `detectedPolygon.inter_area(projectedPolygon)` in mm^2
// (map_primitive.cpp:137 -> src/utils/polygon.cpp:525-545) instead of the shared cells of the label grids that
// cape_match_consecutive counts.  The planes of frame f-1 play the map planes, seen through the identity pose.
//
//   cape_polygon_gate_kernel   : one wavefront per frame: the kept planes of the frame and of its predecessor, the gates
//        is_distance_similar / is_normal_similar (shape_primitives.cpp:66-86) of every (previous plane j, plane i) pair, and
//        the work list of the pairs to intersect.
//   cape_polygon_inter_kernel  : persistent wavefronts, one pair at a time: the polygon of j is projected into the frame of i
//        (Polygon::project, polygon.cpp:338-382) and the area of the intersection of the two rings is computed.  Two
//        instances by capacity (LDS carve); a pair beyond the small one's moves to the large one's list.
//   cape_polygon_select_kernel : one wavefront per frame: the selection loop (greatest intersection above the overlap
//        threshold, is-matched flags updated between previous planes, the `selectedIndex <= 0` quirk).
//
// The intersection is this repo's host algorithm (host/boundary_polygon.cpp: rings_inter_area), statement for statement: the
// plane is cut into vertical slabs at every vertex and every edge crossing; inside a slab each ring is a stack of edges sorted
// by height and the overlap of the two stacks is a sum of trapezoids, added in slab order.  Lanes take the edge pairs (for the
// crossings), the compare-exchanges of a bitonic sort (slab boundaries), the (edge, slab) incidences of a window of 64 slabs
// (heights into per-slab buckets, then their ranks) and one slab each for the trapezoids, which are then added to the running
// area in order: the sum's rounding is observable.  + - x / and comparisons only: the areas are compared BIT FOR BIT with the
// host class (tests/test_gpu_match_polygon.py).
#include <hip/hip_runtime.h>

#include "cape_internal.h"
#include "cape_wave.h"

namespace cape {

namespace {

constexpr int kWaves = 2; // frames per workgroup of the gate / select kernels

// Three instances of the intersection kernel, by capacity: vertices of one ring (simplified polygons: 13 on average, 90 at
// most on the test streams), slab boundaries (vertices of both rings + edge crossings), edges of one ring over one slab,
// independent waves per workgroup.  The capacities only size the LDS carve, i.e. how many waves a CU holds: the work is one
// wave's dependent LDS round trips and only other waves fill the gaps.  The small instance takes ~95 % of the pairs; a pair
// that exceeds one of an instance's capacities moves to the next one's work list.
template <int TIER> struct Tier;
#ifndef CAPE_MP_T0_STACK
#define CAPE_MP_T0_STACK 6
#endif
#ifndef CAPE_MP_T0_XS
#define CAPE_MP_T0_XS 256
#endif
#ifndef CAPE_MP_T0_GROUPS
#define CAPE_MP_T0_GROUPS 4
#endif
// kCoop (round 6): the four waves of a workgroup work on ONE pair -- the tiers behind the first hold the long pairs (an outline of
// 70 vertices over 145 slabs: 0.1 ms on a lone wave), whose edge-pair crossings, boundary sort and (edge, slab) incidences are
// data-parallel; only the final sum of the slab terms is ordered.  Same LDS per pair as before (the carve is shared), a quarter of
// the time for the parallel phases.  CAPE_MP_COOP=0: one wave per pair in every tier (the round-5 kernel, A/B builds).
#ifndef CAPE_MP_COOP
#define CAPE_MP_COOP 1
#endif
#ifndef CAPE_MP_COOP0
#define CAPE_MP_COOP0 0 // the first tier cooperative as well (A/B builds: -DCAPE_MP_COOP0=1 -DCAPE_MP_T0_GROUPS=8)
#endif
template <> struct Tier<0>
{
    static constexpr bool kCoop = CAPE_MP_COOP0 != 0;
    static constexpr int kRing = 32, kXs = CAPE_MP_T0_XS, kStack = CAPE_MP_T0_STACK, kWavesPerGroup = kCoop ? 4 : 2, kGroupsPerCu = CAPE_MP_T0_GROUPS;
};
template <> struct Tier<1>
{
    static constexpr bool kCoop = CAPE_MP_COOP != 0;
    static constexpr int kRing = 128, kXs = 1024, kStack = 16, kWavesPerGroup = kCoop ? 4 : 1, kGroupsPerCu = 2;
};
template <> struct Tier<2> // the comb-shaped outline that comes along once in a few thousand frames
{
    static constexpr bool kCoop = CAPE_MP_COOP != 0;
    static constexpr int kRing = 128, kXs = 1024, kStack = 32, kWavesPerGroup = kCoop ? 4 : 1, kGroupsPerCu = 1;
};
template <> struct Tier<3> // outlines of more than 128 vertices: the 64 x 48 cell grid of 1280 x 960 frames shows them (207 on the
{                          // TUM-like stream, profiles/r04_capacity_probe.txt); 145 KB of LDS, one workgroup per CU
    static constexpr bool kCoop = CAPE_MP_COOP != 0;
    static constexpr int kRing = 512, kXs = 2048, kStack = 32, kWavesPerGroup = kCoop ? 4 : 1, kGroupsPerCu = 1;
};
constexpr int kTiers = 4;
// the largest capacity any LATER tier offers (a pair beyond a tier's capacity moves on while one of them can hold it)
template <int TIER> constexpr int later_ring() { return TIER + 1 < kTiers ? (Tier<(TIER + 1 < kTiers ? TIER + 1 : TIER)>::kRing > later_ring<TIER + 1>() ? Tier<(TIER + 1 < kTiers ? TIER + 1 : TIER)>::kRing : later_ring<TIER + 1>()) : 0; }
template <> constexpr int later_ring<kTiers>() { return 0; }
template <int TIER> constexpr int later_xs() { return TIER + 1 < kTiers ? (Tier<(TIER + 1 < kTiers ? TIER + 1 : TIER)>::kXs > later_xs<TIER + 1>() ? Tier<(TIER + 1 < kTiers ? TIER + 1 : TIER)>::kXs : later_xs<TIER + 1>()) : 0; }
template <> constexpr int later_xs<kTiers>() { return 0; }
template <int TIER> constexpr int later_stack() { return TIER + 1 < kTiers ? (Tier<(TIER + 1 < kTiers ? TIER + 1 : TIER)>::kStack > later_stack<TIER + 1>() ? Tier<(TIER + 1 < kTiers ? TIER + 1 : TIER)>::kStack : later_stack<TIER + 1>()) : 0; }
template <> constexpr int later_stack<kTiers>() { return 0; }
constexpr int MP = CAPE_MATCH_MAX_PLANES;

#define CAPE_MP_SYNC()                                                                                        \
    do                                                                                                       \
    {                                                                                                        \
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");                                               \
        __builtin_amdgcn_s_waitcnt(0);                                                                       \
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");                                               \
    } while (0)

struct Edge // a.x < b.x
{
    double2 a, b;
};

struct MpLds
{
    int ringCap;
    double2* ringA; // ringCap
    double2* ringB; // ringCap
    Edge* ea;       // ringCap
    Edge* eb;       // ringCap
    double* xs;     // the tier's kXs
    double* terms;  // 64 x CAP (the intervals of a ring over a slab are disjoint: two sorted families of CAP/2 overlap in < CAP pairs)
    // the edges of both rings as ONE list: ring A's at [0, nea), ring B's at [ringCap, ringCap + neb)
    int* elo;             // 2 x ringCap: first slab an edge spans (index of a.x among the boundaries)
    int* ehi;             // 2 x ringCap: one past the last (index of b.x)
    int* pre;             // 2 x ringCap + 1: exclusive prefix of the edges' slab counts inside the current window
    int* cnt;             // 2 x 64: edges over each slab of the window, per ring
    double* by;           // 2 x 64 x CAP: their heights at the middle of the slab, in arrival order
    unsigned short* bk;   // 2 x 64 x CAP: their edge indices
    unsigned short* inc;  // 2 x 64 x CAP: where each (edge, slab) incidence of the window went: bucket << 8 | position
    unsigned char* sidx;  // 2 x 64 x CAP: positions in sorted order
    int* sh;              // 8 words the waves of a cooperative workgroup hand uniform values over in (kCoop tiers)
};

__device__ __forceinline__ double y_at(const Edge& e, double x) { return e.a.y + (e.b.y - e.a.y) * ((x - e.a.x) / (e.b.x - e.a.x)); }

// ring_area_signed of a ring in LDS (ordered sum; every lane walks it)
__device__ __forceinline__ double ring_area_signed(const double2* r, int n)
{
    double s = 0;
    for (int i = 0, j = n - 1; i < n; j = i++)
        s += (r[j].x * r[i].y - r[i].x * r[j].y);
    return 0.5 * s;
}

// edges_of: the non-vertical edges (r[i-1], r[i]) in ring order, left end first.  Sequential compaction in ring order
// (every lane runs it, lane 0 writes): the order of the edges decides ties of the stable sort below.
__device__ __forceinline__ int edges_of(const double2* r, int n, Edge* out, int lane)
{
    int cnt = 0;
    for (int base = 0; base < n; base += 64)
    {
        const int i = base + lane;
        bool keep = false;
        Edge e;
        e.a = e.b = make_double2(0, 0);
        if (i < n)
        {
            double2 p = r[i == 0 ? n - 1 : i - 1], q = r[i];
            keep = !(p.x == q.x); // vertical edges bound no area in x
            if (p.x > q.x)
            {
                const double2 t = p;
                p = q;
                q = t;
            }
            e.a = p;
            e.b = q;
        }
        const unsigned long long kb = __ballot(keep);
        if (keep)
            out[cnt + __popcll(kb & ((1ull << lane) - 1ull))] = e;
        cnt += __popcll(kb);
    }
    CAPE_MP_SYNC();
    return cnt;
}

// bitonic sort of xs[0, n) ascending, padded with +inf to a power of two
__device__ inline void sort_xs(double* xs, int n, int lane)
{
    int np = 64;
    while (np < n)
        np <<= 1;
    for (int i = n + lane; i < np; i += 64)
        xs[i] = __builtin_inf();
    CAPE_MP_SYNC();
    for (int size = 2; size <= np; size <<= 1)
        for (int stride = size >> 1; stride > 0; stride >>= 1)
        {
            for (int t = lane; t < np / 2; t += 64)
            {
                const int lo = (t / stride) * (2 * stride) + (t % stride), hi = lo + stride;
                const bool up = ((lo / size) & 1) == 0;
                const double a = xs[lo], b = xs[hi];
                if ((b < a) == up && a != b)
                {
                    xs[lo] = b;
                    xs[hi] = a;
                }
            }
            CAPE_MP_SYNC();
        }
}

// index of the value v among the sorted, distinct boundaries xs[0, n) (v is one of them)
__device__ __forceinline__ int boundary_index(const double* xs, int n, double v)
{
    int lo = 0, hi = n - 1;
    while (lo < hi)
    {
        const int mid = (lo + hi) >> 1;
        if (xs[mid] < v)
            lo = mid + 1;
        else
            hi = mid;
    }
    return lo;
}

// why a pair has no area (quiet NaNs told apart by their payload)
__device__ __forceinline__ double nan_code(int code) { return __longlong_as_double(0x7ff8000000000000ll | code); }
constexpr int kNanRing = 1, kNanSlabs = 2, kNanStack = 3, kNanPending = 4;
__device__ __forceinline__ bool is_nan_code(double v, int code) { return __double_as_longlong(v) == (0x7ff8000000000000ll | code); }

// rings_inter_area(A, B): A, B in LDS.  Returns the area, or NaN if a capacity of this kernel was exceeded (the caller flags
// the pair; the host class has no such limit).
#ifdef CAPE_MP_PROFILE
#define CAPE_MP_TICK(k) (prof[k] = __builtin_amdgcn_s_memtime())
#else
#define CAPE_MP_TICK(k)
#endif
template <int kStackCap, int kXsCap>
__device__ inline double rings_inter_area(const MpLds& L, int na, int nb, int lane, unsigned long long* prof = nullptr)
{
    constexpr int kTermsPerSlab = kStackCap;
    CAPE_MP_TICK(0);
    if (na < 3 || nb < 3)
        return 0.0;
    const int nea = edges_of(L.ringA, na, L.ea, lane), neb = edges_of(L.ringB, nb, L.eb, lane);
    // slab boundaries: every vertex, every isolated edge crossing
    int nx = 0;
    for (int i = lane; i < na; i += 64)
        L.xs[i] = L.ringA[i].x;
    for (int i = lane; i < nb; i += 64)
        L.xs[na + i] = L.ringB[i].x;
    nx = na + nb;
    bool overflow = false;
    const int pairs = nea * neb;
    for (int base = 0; base < pairs; base += 64)
    {
        const int q = base + lane;
        bool has = false;
        double x = 0.0;
        if (q < pairs)
        {
            const Edge e = L.ea[q / neb], f = L.eb[q % neb];
            const double d1x = e.b.x - e.a.x, d1y = e.b.y - e.a.y, d2x = f.b.x - f.a.x, d2y = f.b.y - f.a.y;
            const double den = d1x * d2y - d1y * d2x;
            if (den != 0) // parallel / collinear: no isolated crossing
            {
                const double t = ((f.a.x - e.a.x) * d2y - (f.a.y - e.a.y) * d2x) / den;
                const double u = ((f.a.x - e.a.x) * d1y - (f.a.y - e.a.y) * d1x) / den;
                if (t > 0 && t < 1 && u > 0 && u < 1)
                {
                    has = true;
                    x = e.a.x + t * d1x;
                }
            }
        }
        const unsigned long long hb = __ballot(has);
        const int add = __popcll(hb);
        if (nx + add > kXsCap)
        {
            overflow = true;
            break;
        }
        if (has)
            L.xs[nx + __popcll(hb & ((1ull << lane) - 1ull))] = x;
        nx += add;
    }
    if (overflow)
        return nan_code(kNanSlabs);
    CAPE_MP_SYNC();
    CAPE_MP_TICK(1);
#ifdef CAPE_MP_PROFILE
    prof[5] = (unsigned long long)nx;
#endif
    sort_xs(L.xs, nx, lane);
    // std::unique
    {
        int n = 0;
        for (int base = 0; base < nx; base += 64)
        {
            const int i = base + lane;
            double v = 0.0;
            bool keep = false;
            if (i < nx)
            {
                v = L.xs[i];
                keep = i == 0 || !(L.xs[i - 1] == v);
            }
            const unsigned long long kb = __ballot(keep);
            CAPE_MP_SYNC();
            if (keep)
                L.xs[n + __popcll(kb & ((1ull << lane) - 1ull))] = v;
            n += __popcll(kb);
            CAPE_MP_SYNC();
        }
        nx = n;
    }
    CAPE_MP_TICK(2);
#ifdef CAPE_MP_PROFILE
    prof[6] = (unsigned long long)nx;
#endif
    // Every boundary is a vertex abscissa or a crossing, so an edge spans exactly the slabs between the boundary at its left end
    // and the one at its right end (`a.x <= x0 && b.x >= x1` of the host loop)
    const int R = L.ringCap;
    for (int k = lane; k < 2 * R; k += 64)
    {
        const bool isB = k >= R;
        const int kk = isB ? k - R : k;
        int lo = 0, hi = 0;
        if (kk < (isB ? neb : nea))
        {
            const Edge e = isB ? L.eb[kk] : L.ea[kk];
            lo = boundary_index(L.xs, nx, e.a.x);
            hi = boundary_index(L.xs, nx, e.b.x);
        }
        L.elo[k] = lo;
        L.ehi[k] = hi;
    }
    CAPE_MP_SYNC();
    // slabs, 64 at a time.  The (edge, slab) incidences of the window are spread over the lanes: each computes the height of its
    // edge at the middle of its slab and drops it into the slab's bucket; a second pass ranks every entry inside its bucket by
    // (height, edge index) -- the order the host's stable sort by height leaves; then lane l computes the trapezoids of slab
    // base + l from the sorted stacks and the terms are added in slab order.
    double area = 0.0;
    for (int base = 0; base + 1 < nx; base += 64)
    {
        const int top = (base + 64 < nx - 1) ? base + 64 : nx - 1; // slabs [base, top)
        // incidences per edge, exclusive prefix
        int carry = 0;
        for (int kb = 0; kb < 2 * R; kb += 64)
        {
            const int k = kb + lane;
            int c = 0;
            if (k < 2 * R)
            {
                const int lo = L.elo[k] > base ? L.elo[k] : base, hi = L.ehi[k] < top ? L.ehi[k] : top;
                c = hi > lo ? hi - lo : 0;
            }
            const int incl = wave_scan_i32(c);
            if (k < 2 * R)
                L.pre[k] = carry + incl - c;
            carry += __builtin_amdgcn_readlane(incl, 63);
        }
        const int total = carry;
        if (lane == 0)
            L.pre[2 * R] = total;
        for (int q = lane; q < 128; q += 64)
            L.cnt[q] = 0;
        CAPE_MP_SYNC();
        if (total > 2 * 64 * kStackCap)
            return nan_code(kNanStack); // (some bucket must overflow)
        bool over = false;
        for (int t = lane; t < total; t += 64)
        {
            // the edge of incidence t: the last k with pre[k] <= t
            int lo = 0, hi = 2 * R;
            while (hi - lo > 1)
            {
                const int mid = (lo + hi) >> 1;
                if (L.pre[mid] <= t)
                    lo = mid;
                else
                    hi = mid;
            }
            const int k = lo;
            const bool isB = k >= R;
            const int kk = isB ? k - R : k;
            const int first = L.elo[k] > base ? L.elo[k] : base;
            const int sl = first + (t - L.pre[k]) - base;
            const Edge e = isB ? L.eb[kk] : L.ea[kk];
            const double x0 = L.xs[base + sl], x1 = L.xs[base + sl + 1], xm = 0.5 * (x0 + x1);
            const double y = y_at(e, xm);
            const int bucket = (isB ? 64 : 0) + sl;
            const int pos = atomicAdd(&L.cnt[bucket], 1);
            if (pos < kStackCap)
            {
                L.by[bucket * kStackCap + pos] = y;
                L.bk[bucket * kStackCap + pos] = (unsigned short)kk;
                L.inc[t] = (unsigned short)((bucket << 8) | pos);
            }
            else
                over = true;
        }
        CAPE_MP_SYNC();
        if (__any(over))
            return nan_code(kNanStack);
        for (int t = lane; t < total; t += 64)
        {
            const int bucket = L.inc[t] >> 8, pos = L.inc[t] & 255;
            const double y = L.by[bucket * kStackCap + pos];
            const int kk = L.bk[bucket * kStackCap + pos];
            const int c = L.cnt[bucket];
            int rank = 0;
            for (int m = 0; m < c; ++m)
            {
                const double y2 = L.by[bucket * kStackCap + m];
                const int k2 = L.bk[bucket * kStackCap + m];
                rank += (y2 < y || (y2 == y && k2 < kk)) ? 1 : 0;
            }
            L.sidx[bucket * kStackCap + rank] = (unsigned char)pos;
        }
        CAPE_MP_SYNC();
        const int s = base + lane;
        int myTerms = 0;
        if (s < top)
        {
            const double x0 = L.xs[s], x1 = L.xs[s + 1];
            const int ca = L.cnt[lane], cb = L.cnt[64 + lane];
            const int oa = lane * kStackCap, ob = (64 + lane) * kStackCap;
            for (int i = 0; i + 1 < ca; i += 2)
            {
                const int pa0 = L.sidx[oa + i], pa1 = L.sidx[oa + i + 1];
                const double ya0 = L.by[oa + pa0], ya1 = L.by[oa + pa1];
                for (int j = 0; j + 1 < cb; j += 2)
                {
                    const int pb0 = L.sidx[ob + j], pb1 = L.sidx[ob + j + 1];
                    const double yb0 = L.by[ob + pb0], yb1 = L.by[ob + pb1];
                    const bool loA = ya0 > yb0;
                    const double loY = loA ? ya0 : yb0;
                    const bool hiA = ya1 < yb1;
                    const double hiY = hiA ? ya1 : yb1;
                    if (!(hiY <= loY))
                    {
                        const Edge lo = loA ? L.ea[L.bk[oa + pa0]] : L.eb[L.bk[ob + pb0]];
                        const Edge hi = hiA ? L.ea[L.bk[oa + pa1]] : L.eb[L.bk[ob + pb1]];
                        const double h0 = y_at(hi, x0) - y_at(lo, x0);
                        const double h1 = y_at(hi, x1) - y_at(lo, x1);
                        if (myTerms < kTermsPerSlab)
                            L.terms[lane * kTermsPerSlab + myTerms] = 0.5 * (h0 + h1) * (x1 - x0);
                        ++myTerms;
                    }
                }
            }
        }
        CAPE_MP_SYNC();
        if (__any(myTerms > kTermsPerSlab))
            return nan_code(kNanStack); // (cannot happen with simple rings)
        // the ordered sum: slab by slab, term by term.  The first two terms of a slab travel through registers (a slab of two
        // convex-ish outlines has one), the rest through LDS
        const double t0 = myTerms > 0 ? L.terms[lane * kTermsPerSlab] : 0.0, t1 = myTerms > 1 ? L.terms[lane * kTermsPerSlab + 1] : 0.0;
        const int slabs = top - base;
        for (int l = 0; l < slabs; ++l)
        {
            const int c = __builtin_amdgcn_readlane(myTerms, l);
            if (c > 0)
                area += readlane_f64(t0, l);
            if (c > 1)
                area += readlane_f64(t1, l);
            for (int t = 2; t < c; ++t)
                area += L.terms[l * kTermsPerSlab + t];
        }
        CAPE_MP_SYNC();
    }
    CAPE_MP_TICK(3);
    return area;
}

// rings_inter_area for the FOUR waves of a workgroup on one pair (Tier::kCoop): the same statements, the data-parallel loops
// spread over 256 threads, the wave-collective ones (ordered compaction of the edges, std::unique, the prefix over the edges'
// slab counts, the slabs' trapezoids and the ordered sum) on wave 0, workgroup barriers between them.  Uniform values travel
// through L.sh.  The area is wave 0's (thread 0 stores it); every thread returns the same NaN code when a capacity is exceeded.
template <int kStackCap, int kXsCap>
__device__ inline double rings_inter_area_coop(const MpLds& L, int na, int nb, int tid)
{
    constexpr int kTermsPerSlab = kStackCap;
    constexpr int NT = 256;
    const int lane = tid & 63, wave = tid >> 6;
    int* sh = L.sh;
    if (na < 3 || nb < 3)
        return 0.0;
    if (wave == 0)
    {
        const int nea0 = edges_of(L.ringA, na, L.ea, lane), neb0 = edges_of(L.ringB, nb, L.eb, lane);
        if (lane == 0)
        {
            sh[0] = nea0;
            sh[1] = neb0;
            sh[2] = na + nb; // boundaries so far: every vertex
            sh[3] = 0;       // overflow
        }
    }
    for (int i = tid; i < na; i += NT)
        L.xs[i] = L.ringA[i].x;
    for (int i = tid; i < nb; i += NT)
        L.xs[na + i] = L.ringB[i].x;
    __syncthreads();
    const int nea = sh[0], neb = sh[1];
    // isolated edge crossings: every wave takes its share of the edge pairs and reserves room for what it finds with one atomic
    // (the boundaries are sorted afterwards: their arrival order is free)
    const int pairs = nea * neb;
    for (int base = 64 * wave; base < pairs; base += NT)
    {
        const int q = base + lane;
        bool has = false;
        double x = 0.0;
        if (q < pairs)
        {
            const Edge e = L.ea[q / neb], f = L.eb[q % neb];
            const double d1x = e.b.x - e.a.x, d1y = e.b.y - e.a.y, d2x = f.b.x - f.a.x, d2y = f.b.y - f.a.y;
            const double den = d1x * d2y - d1y * d2x;
            if (den != 0) // parallel / collinear: no isolated crossing
            {
                const double t = ((f.a.x - e.a.x) * d2y - (f.a.y - e.a.y) * d2x) / den;
                const double u = ((f.a.x - e.a.x) * d1y - (f.a.y - e.a.y) * d1x) / den;
                if (t > 0 && t < 1 && u > 0 && u < 1)
                {
                    has = true;
                    x = e.a.x + t * d1x;
                }
            }
        }
        const unsigned long long hb = __ballot(has);
        const int add = __popcll(hb);
        if (add)
        {
            int at = 0;
            if (lane == 0)
                at = atomicAdd(&sh[2], add);
            at = __builtin_amdgcn_readfirstlane(at);
            if (at + add > kXsCap)
            {
                if (lane == 0)
                    sh[3] = 1;
            }
            else if (has)
                L.xs[at + __popcll(hb & ((1ull << lane) - 1ull))] = x;
        }
    }
    __syncthreads();
    if (sh[3])
        return nan_code(kNanSlabs);
    int nx = sh[2];
    // bitonic sort of the boundaries over the workgroup
    {
        int np = 64;
        while (np < nx)
            np <<= 1;
        for (int i = nx + tid; i < np; i += NT)
            L.xs[i] = __builtin_inf();
        __syncthreads();
        for (int size = 2; size <= np; size <<= 1)
            for (int stride = size >> 1; stride > 0; stride >>= 1)
            {
                for (int t = tid; t < np / 2; t += NT)
                {
                    const int lo = (t / stride) * (2 * stride) + (t % stride), hi = lo + stride;
                    const bool up = ((lo / size) & 1) == 0;
                    const double a = L.xs[lo], b = L.xs[hi];
                    if ((b < a) == up && a != b)
                    {
                        L.xs[lo] = b;
                        L.xs[hi] = a;
                    }
                }
                __syncthreads();
            }
    }
    // std::unique (wave 0, in order)
    if (wave == 0)
    {
        int n = 0;
        for (int base = 0; base < nx; base += 64)
        {
            const int i = base + lane;
            double v = 0.0;
            bool keep = false;
            if (i < nx)
            {
                v = L.xs[i];
                keep = i == 0 || !(L.xs[i - 1] == v);
            }
            const unsigned long long kb = __ballot(keep);
            CAPE_MP_SYNC();
            if (keep)
                L.xs[n + __popcll(kb & ((1ull << lane) - 1ull))] = v;
            n += __popcll(kb);
            CAPE_MP_SYNC();
        }
        if (lane == 0)
            sh[2] = n;
    }
    __syncthreads();
    nx = sh[2];
    const int R = L.ringCap;
    for (int k = tid; k < 2 * R; k += NT)
    {
        const bool isB = k >= R;
        const int kk = isB ? k - R : k;
        int lo = 0, hi = 0;
        if (kk < (isB ? neb : nea))
        {
            const Edge e = isB ? L.eb[kk] : L.ea[kk];
            lo = boundary_index(L.xs, nx, e.a.x);
            hi = boundary_index(L.xs, nx, e.b.x);
        }
        L.elo[k] = lo;
        L.ehi[k] = hi;
    }
    __syncthreads();
    double area = 0.0; // (wave 0's)
    for (int base = 0; base + 1 < nx; base += 64)
    {
        const int top = (base + 64 < nx - 1) ? base + 64 : nx - 1; // slabs [base, top)
        if (wave == 0)
        {
            int carry = 0;
            for (int kb = 0; kb < 2 * R; kb += 64)
            {
                const int k = kb + lane;
                int c = 0;
                if (k < 2 * R)
                {
                    const int lo = L.elo[k] > base ? L.elo[k] : base, hi = L.ehi[k] < top ? L.ehi[k] : top;
                    c = hi > lo ? hi - lo : 0;
                }
                const int incl = wave_scan_i32(c);
                if (k < 2 * R)
                    L.pre[k] = carry + incl - c;
                carry += __builtin_amdgcn_readlane(incl, 63);
            }
            if (lane == 0)
            {
                L.pre[2 * R] = carry;
                sh[4] = carry;
                sh[5] = 0; // a bucket overflowed
                sh[6] = 0; // more terms than a slab holds
            }
        }
        for (int q = tid; q < 128; q += NT)
            L.cnt[q] = 0;
        __syncthreads();
        const int total = sh[4];
        if (total > 2 * 64 * kStackCap)
            return nan_code(kNanStack); // (some bucket must overflow)
        bool over = false;
        for (int t = tid; t < total; t += NT)
        {
            int lo = 0, hi = 2 * R;
            while (hi - lo > 1)
            {
                const int mid = (lo + hi) >> 1;
                if (L.pre[mid] <= t)
                    lo = mid;
                else
                    hi = mid;
            }
            const int k = lo;
            const bool isB = k >= R;
            const int kk = isB ? k - R : k;
            const int first = L.elo[k] > base ? L.elo[k] : base;
            const int sl = first + (t - L.pre[k]) - base;
            const Edge e = isB ? L.eb[kk] : L.ea[kk];
            const double x0 = L.xs[base + sl], x1 = L.xs[base + sl + 1], xm = 0.5 * (x0 + x1);
            const double y = y_at(e, xm);
            const int bucket = (isB ? 64 : 0) + sl;
            const int pos = atomicAdd(&L.cnt[bucket], 1);
            if (pos < kStackCap)
            {
                L.by[bucket * kStackCap + pos] = y;
                L.bk[bucket * kStackCap + pos] = (unsigned short)kk;
                L.inc[t] = (unsigned short)((bucket << 8) | pos);
            }
            else
                over = true;
        }
        if (over)
            sh[5] = 1;
        __syncthreads();
        if (sh[5])
            return nan_code(kNanStack);
        for (int t = tid; t < total; t += NT)
        {
            const int bucket = L.inc[t] >> 8, pos = L.inc[t] & 255;
            const double y = L.by[bucket * kStackCap + pos];
            const int kk = L.bk[bucket * kStackCap + pos];
            const int c = L.cnt[bucket];
            int rank = 0;
            for (int m = 0; m < c; ++m)
            {
                const double y2 = L.by[bucket * kStackCap + m];
                const int k2 = L.bk[bucket * kStackCap + m];
                rank += (y2 < y || (y2 == y && k2 < kk)) ? 1 : 0;
            }
            L.sidx[bucket * kStackCap + rank] = (unsigned char)pos;
        }
        __syncthreads();
        if (wave == 0)
        {
            const int s = base + lane;
            int myTerms = 0;
            if (s < top)
            {
                const double x0 = L.xs[s], x1 = L.xs[s + 1];
                const int ca = L.cnt[lane], cb = L.cnt[64 + lane];
                const int oa = lane * kStackCap, ob = (64 + lane) * kStackCap;
                for (int i = 0; i + 1 < ca; i += 2)
                {
                    const int pa0 = L.sidx[oa + i], pa1 = L.sidx[oa + i + 1];
                    const double ya0 = L.by[oa + pa0], ya1 = L.by[oa + pa1];
                    for (int j = 0; j + 1 < cb; j += 2)
                    {
                        const int pb0 = L.sidx[ob + j], pb1 = L.sidx[ob + j + 1];
                        const double yb0 = L.by[ob + pb0], yb1 = L.by[ob + pb1];
                        const bool loA = ya0 > yb0;
                        const double loY = loA ? ya0 : yb0;
                        const bool hiA = ya1 < yb1;
                        const double hiY = hiA ? ya1 : yb1;
                        if (!(hiY <= loY))
                        {
                            const Edge lo = loA ? L.ea[L.bk[oa + pa0]] : L.eb[L.bk[ob + pb0]];
                            const Edge hi = hiA ? L.ea[L.bk[oa + pa1]] : L.eb[L.bk[ob + pb1]];
                            const double h0 = y_at(hi, x0) - y_at(lo, x0);
                            const double h1 = y_at(hi, x1) - y_at(lo, x1);
                            if (myTerms < kTermsPerSlab)
                                L.terms[lane * kTermsPerSlab + myTerms] = 0.5 * (h0 + h1) * (x1 - x0);
                            ++myTerms;
                        }
                    }
                }
            }
            CAPE_MP_SYNC();
            if (__any(myTerms > kTermsPerSlab))
            {
                if (lane == 0)
                    sh[6] = 1;
            }
            else
            {
                const double t0 = myTerms > 0 ? L.terms[lane * kTermsPerSlab] : 0.0, t1 = myTerms > 1 ? L.terms[lane * kTermsPerSlab + 1] : 0.0;
                const int slabs = top - base;
                for (int l = 0; l < slabs; ++l)
                {
                    const int c = __builtin_amdgcn_readlane(myTerms, l);
                    if (c > 0)
                        area += readlane_f64(t0, l);
                    if (c > 1)
                        area += readlane_f64(t1, l);
                    for (int t = 2; t < c; ++t)
                        area += L.terms[l * kTermsPerSlab + t];
                }
            }
        }
        __syncthreads();
        if (sh[6])
            return nan_code(kNanStack); // (cannot happen with simple rings)
    }
    return area;
}

// the kept planes of a frame (output plane whose polygon Primitive_Detection keeps), in segment order: lane k < count holds
// the segment index of plane k
// hostOnly: some output plane of the frame has no device polygon (CAPE_POLY_OVERFLOW: its outline is left to the host class,
// which may well keep it) -- the kept-plane indices of such a frame cannot be told here, so the caller flags it
__device__ __forceinline__ int valid_planes(const cape_frame_record& rec, const cape_polygon* pol, int lane, int& mySeg, bool& hostOnly)
{
    int nSeg = rec.header.n_plane_segments;
    nSeg = nSeg < 0 ? 0 : (nSeg > CAPE_MAX_PLANES ? CAPE_MAX_PLANES : nSeg);
    const bool isOut = lane < nSeg && rec.segments[lane].is_output != 0;
    const unsigned flags = isOut ? pol[lane].flags : 0u;
    const bool ok = isOut && (flags & CAPE_POLY_VALID) != 0 && pol[lane].vertex_count >= 3;
    // (a frame that continues in spill records -- more than 64 plane segments -- is the host class's as well: its kept planes are
    // not all in this record)
    hostOnly = __ballot(isOut && (flags & CAPE_POLY_OVERFLOW) != 0) != 0ull || rec.header.next_record >= 0;
    const unsigned long long m = __ballot(ok);
    // lane k takes the k-th set bit
    int seg = -1;
    unsigned long long mm = m;
    for (int k = 0; k <= lane && mm; ++k, mm &= mm - 1)
        if (k == lane)
            seg = __ffsll((long long)mm) - 1;
    mySeg = seg;
    return __popcll(m);
}

// a pair of the work lists
__device__ __forceinline__ unsigned pack_pair(int frame, int j, int i) { return ((unsigned)frame << 8) | ((unsigned)j << 4) | (unsigned)i; }

} // namespace

// One wavefront per frame, sixteen frames per workgroup: the kept planes of the frame and of its predecessor, the gates of every (previous plane j, plane i)
// pair (Plane::is_distance_similar / is_normal_similar on the planes' parametrisations, shape_primitives.cpp:66-86) and the
// work list of the pairs whose polygons are to be intersected.  A pair the gates reject holds -1.
constexpr int kGateFrames = 16; // frames (waves) of a gate workgroup
__global__ __launch_bounds__(64 * kGateFrames) void cape_polygon_gate_kernel(MatchPolygonParams p, int nFrames)
{
    __shared__ unsigned s_count[kGateFrames];
    __shared__ unsigned s_base;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int frameRaw = blockIdx.x * kGateFrames + wave;
    const bool live = frameRaw < nFrames;
    const int frame = live ? frameRaw : nFrames - 1; // (idle waves of the last workgroup shadow a real frame and store nothing)
    cape_frame_match_exact& out = p.matches[frame];
    const cape_frame_record& recC = p.records[frame];
    const cape_polygon* polC = p.polygons + (size_t)frame * CAPE_MAX_PLANES;
    int segC = -1, segP = -1;
    bool hostOnlyC = false, hostOnlyP = false;
    const int nCur = valid_planes(recC, polC, lane, segC, hostOnlyC);
    const int nPrev = frame > 0 ? valid_planes(p.records[frame - 1], polC - CAPE_MAX_PLANES, lane, segP, hostOnlyP) : 0;
    const bool fits = nCur <= MP && nPrev <= MP && !hostOnlyC && !hostOnlyP;
    if (live && lane == 0)
    {
        out.n_prev = nPrev;
        out.n_cur = nCur;
        out.flags = fits ? 0u : (uint32_t)CAPE_MATCH_EXACT_OVERFLOW;
        out.pad = 0;
    }
    if (live && lane < MP)
    {
        out.match[lane] = -1;
        out.seg_prev[lane] = (lane < nPrev) ? segP : -1;
        out.seg_cur[lane] = (lane < nCur) ? segC : -1;
    }
    // my plane's parametrisation, read once (lane k: plane k of either frame)
    double cn[3] = {0, 0, 0}, cd = 0, pn[3] = {0, 0, 0}, pd = 0;
    if (segC >= 0)
    {
        const cape_plane_segment& S = recC.segments[segC];
        cn[0] = S.out_normal[0], cn[1] = S.out_normal[1], cn[2] = S.out_normal[2], cd = S.d;
    }
    if (segP >= 0)
    {
        const cape_plane_segment& Q = p.records[frame - 1].segments[segP];
        pn[0] = Q.out_normal[0], pn[1] = Q.out_normal[1], pn[2] = Q.out_normal[2], pd = Q.d;
        if (p.poses)
        {
            // the map plane seen from this frame's camera: PlaneWorldCoordinates::to_camera_coordinates (plane_coordinates.cpp:20-24)
            // with the plane matrix of camera_transformation.cpp:53-71, [R 0; -t^T R 1]; the PlaneCameraCoordinates constructor
            // normalises the rotated normal (host: utils::plane_to_camera)
            const double* T = p.poses + (size_t)frame * 16;
            const double r0 = (T[0] * pn[0] + T[1] * pn[1]) + T[2] * pn[2], r1 = (T[4] * pn[0] + T[5] * pn[1]) + T[6] * pn[2],
                         r2 = (T[8] * pn[0] + T[9] * pn[1]) + T[10] * pn[2];
            const double t0 = T[3], t1 = T[7], t2 = T[11];
            const double m0 = -((t0 * T[0] + t1 * T[4]) + t2 * T[8]), m1 = -((t0 * T[1] + t1 * T[5]) + t2 * T[9]),
                         m2 = -((t0 * T[2] + t1 * T[6]) + t2 * T[10]);
            pd = ((m0 * pn[0] + m1 * pn[1]) + m2 * pn[2]) + pd;
            const double nn = sqrt((r0 * r0 + r1 * r1) + r2 * r2);
            pn[0] = r0, pn[1] = r1, pn[2] = r2;
            if (nn > 0)
                pn[0] = r0 / nn, pn[1] = r1 / nn, pn[2] = r2 / nn;
        }
    }
    unsigned long long gatedMask[MP * MP / 64];
    bool mine[MP * MP / 64];
#pragma unroll
    for (int k = 0; k < MP * MP / 64; ++k)
    {
        const int pair = k * 64 + lane, j = pair / MP, i = pair % MP;
        const double qn0 = __shfl(pn[0], j), qn1 = __shfl(pn[1], j), qn2 = __shfl(pn[2], j), qd = __shfl(pd, j);
        const double sn0 = __shfl(cn[0], i), sn1 = __shfl(cn[1], i), sn2 = __shfl(cn[2], i), sd = __shfl(cd, i);
        bool gated = false;
        if (fits && j < nPrev && i < nCur)
        {
            const double cosAngle = (sn0 * qn0 + sn1 * qn1) + sn2 * qn2;
            gated = fabs(sd - qd) < p.maxDistance && fabs(cosAngle) > p.minCosAngle;
        }
        if (live)
            out.inter_area[j][i] = gated ? nan_code(kNanPending) : -1.0;
        gatedMask[k] = __ballot(gated);
        mine[k] = gated;
    }
    // ONE atomic per workgroup on the list's counter (a counter every frame's wave bumps on its own serialises thousands of
    // atomics on one address: 50 us of the call)
    const unsigned myCount = (unsigned)(__popcll(gatedMask[0]) + __popcll(gatedMask[1]) + __popcll(gatedMask[2]) + __popcll(gatedMask[3]));
    if (lane == 0)
        s_count[wave] = live ? myCount : 0u;
    __syncthreads();
    if (threadIdx.x == 0)
    {
        unsigned total = 0;
        for (int w = 0; w < kGateFrames; ++w)
        {
            const unsigned c = s_count[w];
            s_count[w] = total;
            total += c;
        }
        s_base = total ? atomicAdd(&p.listCounts[0], total) : 0u;
    }
    __syncthreads();
    if (!live)
        return;
    unsigned at = s_base + s_count[wave];
#pragma unroll
    for (int k = 0; k < MP * MP / 64; ++k)
    {
        const int pair = k * 64 + lane;
        if (mine[k])
            p.pairLists[at + __popcll(gatedMask[k] & ((1ull << lane) - 1ull))] = pack_pair(frame, pair / MP, pair % MP);
        at += (unsigned)__popcll(gatedMask[k]);
    }
}

// Persistent waves over the work list of tier TIER.  A pair beyond this tier's capacities moves to the next tier's list when
// that one is larger in the resource that ran out; otherwise its area stays a NaN that names the resource.
template <int TIER>
__global__ __launch_bounds__(64 * Tier<TIER>::kWavesPerGroup) void cape_polygon_inter_kernel(MatchPolygonParams p, int ldsPerWave)
{
    using T = Tier<TIER>;
    constexpr bool kHasNext = TIER + 1 < kTiers;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_all[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    constexpr bool kCoop = T::kCoop;
    // cooperative tiers: ONE carve for the workgroup's four waves, `tid` strides of 256; else a carve per (independent) wave
    const int tid = kCoop ? (int)threadIdx.x : lane;
    constexpr int kStride = kCoop ? 256 : 64;
    unsigned char* smem = smem_all + (kCoop ? (size_t)0 : (size_t)wave * ldsPerWave);
    MpLds L;
    L.ringCap = T::kRing;
    L.ringA = reinterpret_cast<double2*>(smem);
    L.ringB = L.ringA + T::kRing;
    L.ea = reinterpret_cast<Edge*>(L.ringB + T::kRing);
    L.eb = L.ea + T::kRing;
    L.xs = reinterpret_cast<double*>(L.eb + T::kRing);
    L.terms = L.xs + T::kXs;
    L.by = L.terms + 64 * T::kStack;
    L.elo = reinterpret_cast<int*>(L.by + 128 * T::kStack);
    L.ehi = L.elo + 2 * T::kRing;
    L.pre = L.ehi + 2 * T::kRing;
    L.cnt = L.pre + 2 * T::kRing + 2;
    L.bk = reinterpret_cast<unsigned short*>(L.cnt + 128);
    L.inc = L.bk + 128 * T::kStack;
    L.sidx = reinterpret_cast<unsigned char*>(L.inc + 128 * T::kStack);
    L.sh = reinterpret_cast<int*>(smem + ldsPerWave - 32); // the carve's last 32 bytes (tier_lds_bytes)
    const unsigned* list = p.pairLists + (size_t)TIER * p.pairCapacity;
    const unsigned count = p.listCounts[TIER];
    // The tiers behind the first hold few pairs of very unequal cost (an outline of 70 vertices over 145 slabs keeps a lone wave busy
    // for 0.1 ms, its neighbours on the list for a tenth of that): their waves take pairs off a TICKET counter (listCounts[4 + TIER],
    // cleared with the counts) instead of a fixed stride, so a wave that drew a long pair does not also sit on the pairs queued
    // behind it.  CAPE_MP_TICKETS: bit TIER set = that tier draws tickets (A/B builds; default: tiers 1..3).
#ifndef CAPE_MP_TICKETS
#define CAPE_MP_TICKETS 0xE
#endif
#ifndef CAPE_MP_REVERSE
#define CAPE_MP_REVERSE 0
#endif
    constexpr bool kTickets = ((CAPE_MP_TICKETS >> TIER) & 1) != 0;
    constexpr bool kReverse = ((CAPE_MP_REVERSE >> TIER) & 1) != 0;
    auto next_index = [&](unsigned prev, bool first) -> unsigned {
        if (kCoop)
        {
            // one pair per WORKGROUP: thread 0 draws (or strides), everybody reads the number
            if (threadIdx.x == 0)
                L.sh[7] = (int)(kTickets ? atomicAdd(&p.listCounts[4 + TIER], 1u) : (first ? blockIdx.x : prev + gridDim.x));
            __syncthreads();
            const unsigned t = (unsigned)L.sh[7];
            __syncthreads(); // (everybody has read it before thread 0 may write the next one)
            return t;
        }
        if (!kTickets)
            return first ? blockIdx.x * T::kWavesPerGroup + wave : prev + gridDim.x * T::kWavesPerGroup;
        unsigned t = 0;
        if (lane == 0)
            t = atomicAdd(&p.listCounts[4 + TIER], 1u);
        return (unsigned)__builtin_amdgcn_readfirstlane((int)t);
    };
    // ordering point between the lanes that share a carve: the wave (fence + wait) or the workgroup (barrier)
    auto sync = [&]() {
        if (kCoop)
            __syncthreads();
        else
            CAPE_MP_SYNC();
    };
    for (unsigned t = next_index(0u, true); t < count; t = next_index(t, false))
    {
        // (tier 0's list is the gate kernel's, front only; the later tiers': front entries, then the back entries)
        const unsigned front = TIER == 0 ? count : p.listCounts[24 + TIER];
        const unsigned u = kReverse ? count - 1u - t : t;
        const unsigned pair = u < front ? list[u] : list[p.pairCapacity - 1u - (u - front)];
        const int frame = (int)(pair >> 8), j = (int)((pair >> 4) & 15u), i = (int)(pair & 15u);
        cape_frame_match_exact& out = p.matches[frame];
        const int si = out.seg_cur[i], sj = out.seg_prev[j];
        const cape_polygon& PS = p.polygons[(size_t)frame * CAPE_MAX_PLANES + si];       // detected polygon
        const cape_polygon& PQ = p.polygons[(size_t)(frame - 1) * CAPE_MAX_PLANES + sj]; // projected polygon (identity pose)
        const int na = (int)PS.vertex_count, nb = (int)PQ.vertex_count;
        double result;
        if (na > T::kRing || nb > T::kRing)
            result = nan_code(kNanRing);
        else
        {
            const double2* vertsC = p.vertices + (size_t)frame * p.boundaryCapacity + PS.vertex_offset;
            const double2* vertsP = p.vertices + (size_t)(frame - 1) * p.boundaryCapacity + PQ.vertex_offset;
            for (int v = tid; v < na; v += kStride)
                L.ringA[v] = vertsC[v];
            // Polygon::project (polygon.cpp:338-382): every vertex of the previous plane's ring lifted to 3-D and expressed in
            // the frame of plane i; the projected ring is re-oriented clockwise like every polygon (OpenRing constructor)
            // the frame the previous plane's polygon lives in: its own, or -- with a pose -- the one to_camera_space gives it
            // (polygon_coordinates.cpp:135-165: centre through the transform, axes through its rotation and re-normalised)
            double qc[3] = {PQ.center[0], PQ.center[1], PQ.center[2]}, qx[3] = {PQ.x_axis[0], PQ.x_axis[1], PQ.x_axis[2]},
                   qy[3] = {PQ.y_axis[0], PQ.y_axis[1], PQ.y_axis[2]};
            if (p.poses)
            {
                const double* T = p.poses + (size_t)frame * 16;
                double nc[3], nx[3], ny[3];
#pragma unroll
                for (int r = 0; r < 3; ++r)
                {
                    nc[r] = ((T[4 * r] * qc[0] + T[4 * r + 1] * qc[1]) + T[4 * r + 2] * qc[2]) + T[4 * r + 3];
                    nx[r] = (T[4 * r] * qx[0] + T[4 * r + 1] * qx[1]) + T[4 * r + 2] * qx[2];
                    ny[r] = (T[4 * r] * qy[0] + T[4 * r + 1] * qy[1]) + T[4 * r + 2] * qy[2];
                }
                const double lx = sqrt((nx[0] * nx[0] + nx[1] * nx[1]) + nx[2] * nx[2]), ly = sqrt((ny[0] * ny[0] + ny[1] * ny[1]) + ny[2] * ny[2]);
                if (lx > 0)
                    nx[0] /= lx, nx[1] /= lx, nx[2] /= lx;
                if (ly > 0)
                    ny[0] /= ly, ny[1] /= ly, ny[2] /= ly;
                // transform_boundary (polygon.cpp:430-451): every vertex lifted to 3-D, moved, re-expressed in the new frame; then the
                // explicit-ring constructor's orientation fix (polygon.cpp:236-266)
                for (int v = tid; v < nb; v += kStride)
                {
                    const double2 q = vertsP[v];
                    const double X = qc[0] + q.x * qx[0] + q.y * qy[0], Y = qc[1] + q.x * qx[1] + q.y * qy[1], Z = qc[2] + q.x * qx[2] + q.y * qy[2];
                    const double mx = ((T[0] * X + T[1] * Y) + T[2] * Z) + T[3], my = ((T[4] * X + T[5] * Y) + T[6] * Z) + T[7],
                                 mz = ((T[8] * X + T[9] * Y) + T[10] * Z) + T[11];
                    const double dx = mx - nc[0], dy = my - nc[1], dz = mz - nc[2];
                    L.ringB[v] = make_double2((nx[0] * dx + nx[1] * dy) + nx[2] * dz, (ny[0] * dx + ny[1] * dy) + ny[2] * dz);
                }
                sync();
                if (ring_area_signed(L.ringB, nb) > 0)
                {
                    for (int v = tid; v < nb / 2; v += kStride)
                    {
                        const double2 a = L.ringB[v], b = L.ringB[nb - 1 - v];
                        L.ringB[v] = b;
                        L.ringB[nb - 1 - v] = a;
                    }
                    sync();
                }
#pragma unroll
                for (int r = 0; r < 3; ++r)
                    qc[r] = nc[r], qx[r] = nx[r], qy[r] = ny[r];
            }
            for (int v = tid; v < nb; v += kStride)
            {
                const double2 q = p.poses ? L.ringB[v] : vertsP[v];
                const double X = qc[0] + q.x * qx[0] + q.y * qy[0];
                const double Y = qc[1] + q.x * qx[1] + q.y * qy[1];
                const double Z = qc[2] + q.x * qx[2] + q.y * qy[2];
                const double dx = X - PS.center[0], dy = Y - PS.center[1], dz = Z - PS.center[2];
                L.ringB[v] = make_double2((PS.x_axis[0] * dx + PS.x_axis[1] * dy) + PS.x_axis[2] * dz,
                                          (PS.y_axis[0] * dx + PS.y_axis[1] * dy) + PS.y_axis[2] * dz);
            }
            sync();
            if (ring_area_signed(L.ringB, nb) > 0)
            {
                // reverse in place: lane v swaps v and nb - 1 - v
                for (int v = tid; v < nb / 2; v += kStride)
                {
                    const double2 a = L.ringB[v], b = L.ringB[nb - 1 - v];
                    L.ringB[v] = b;
                    L.ringB[nb - 1 - v] = a;
                }
                sync();
            }
#ifdef CAPE_MP_PROFILE
            unsigned long long prof[8] = {0, 0, 0, 0, 0, 0, 0, 0};
            const unsigned long long tStart = __builtin_amdgcn_s_memtime();
            if constexpr (kCoop)
                result = rings_inter_area_coop<T::kStack, T::kXs>(L, na, nb, tid);
            else
                result = rings_inter_area<T::kStack, T::kXs>(L, na, nb, lane, prof);
            if (!kCoop && lane == 0 && i < 8 && j < 8)
            {
                // unused slots of the area matrix carry the ticks of this pair (profiles/match_polygons_bench.py decodes them)
                out.inter_area[j + 8][i + 8] = (double)((prof[3] - tStart) >> 4) + 1e6 * na + 1e9 * nb + 1e12 * (double)prof[5];
                out.inter_area[j + 8][i] = (double)((prof[1] - prof[0]) >> 4) + 1e6 * (double)((prof[2] - prof[1]) >> 4);
                out.inter_area[j][i + 8] = (double)((prof[3] - prof[2]) >> 4) + 1e6 * (double)prof[6] + 1e12 * TIER;
            }
#else
            if constexpr (kCoop)
                result = rings_inter_area_coop<T::kStack, T::kXs>(L, na, nb, tid);
            else
                result = rings_inter_area<T::kStack, T::kXs>(L, na, nb, lane);
#endif
            sync();
        }
        if (tid == 0)
        {
            bool again = false;
            if (kHasNext)
                again = (is_nan_code(result, kNanStack) && later_stack<TIER>() > T::kStack) || (is_nan_code(result, kNanSlabs) && later_xs<TIER>() > T::kXs) ||
                        (is_nan_code(result, kNanRing) && later_ring<TIER>() > T::kRing);
            if (again)
            {
                // the next tier's list fills from both ends: pairs that left for their ring's size -- the big outlines, the long pairs --
                // at the front (their number in listCounts[24 + tier]), the others from the back; the tier works front first, so its
                // longest pairs start first.  A/B (profiles/r06_match_tickets.txt): no gain over arrival order once the waves draw tickets -- the tier ends with its longest PAIR, wherever that starts; kept as a build knob, off (0: one list in arrival order)
#ifndef CAPE_MP_HEAVY_FIRST
#define CAPE_MP_HEAVY_FIRST 0
#endif
                atomicAdd(&p.listCounts[TIER + 1], 1u);
                unsigned* nextList = p.pairLists + (size_t)(TIER + 1) * p.pairCapacity;
                if (CAPE_MP_HEAVY_FIRST == 0 || is_nan_code(result, kNanRing))
                    nextList[atomicAdd(&p.listCounts[24 + TIER + 1], 1u)] = pair;
                else
                    nextList[p.pairCapacity - 1u - atomicAdd(&p.listCounts[28 + TIER + 1], 1u)] = pair;
                // why the pair moves on (cape_debug_match_lists: words 8 + 4 * tier + reason; a handful of atomics per batch)
                atomicAdd(&p.listCounts[8 + 4 * TIER + (is_nan_code(result, kNanRing) ? 1 : (is_nan_code(result, kNanSlabs) ? 2 : 3))], 1u);
            }
            else
                out.inter_area[j][i] = result;
        }
    }
}

// One wavefront per frame: the selection loop of find_matches over the areas.
__global__ __launch_bounds__(64 * kWaves) void cape_polygon_select_kernel(MatchPolygonParams p, int nFrames)
{
    const int lane = threadIdx.x & 63;
    const int frame = blockIdx.x * kWaves + (threadIdx.x >> 6);
    if (frame >= nFrames)
        return;
    cape_frame_match_exact& out = p.matches[frame];
    uint32_t flags = out.flags;
    if (flags & CAPE_MATCH_EXACT_OVERFLOW)
        return; // more than MP kept planes: nothing was intersected
    const int nc = out.n_cur, npv = out.n_prev;
    const int segC = lane < nc ? out.seg_cur[lane] : -1;
    const int segP = lane < npv ? out.seg_prev[lane] : -1;
    const cape_polygon* polC = p.polygons + (size_t)frame * CAPE_MAX_PLANES;
    const double myArea = (lane < nc) ? polC[segC].area : 0.0;                            // detectedPolygon.get_area()
    const double myPrevArea = (lane < npv) ? (polC - CAPE_MAX_PLANES)[segP].area : 0.0; // projectedPolygon.get_area()
    bool matched = false;
    int myMatch = -1;
    for (int j = 0; j < npv; ++j)
    {
        const double projectedArea = __shfl(myPrevArea, j);
        const double ia = (lane < nc) ? out.inter_area[j][lane] : -1.0;
        if (lane < nc && ia != ia)
            flags |= CAPE_MATCH_EXACT_OVERFLOW; // a polygon pair beyond the kernel's capacities
        // interArea > greatestSimilarity (starting at 0) and interArea / newPlaneArea >= threshold; ascending scan with a strict
        // comparison = the lowest index among the largest areas
        unsigned long long key = 0;
        if (lane < nc && !matched && projectedArea > 0.0 && ia > 0.0 && ia / myArea >= p.minOverlap)
            key = (unsigned long long)__double_as_longlong(ia);
        const unsigned long long best = ~wave_min_u64(~key); // maximum of the bit patterns (positive doubles order like them)
        const unsigned mine = (key != 0 && key == best) ? (unsigned)(63 - lane) : 0u;
        const unsigned win = wave_max_u32(mine);
        int selected = best ? 63 - (int)win : -1;
        if (!(p.flags & CAPE_MATCH_ALLOW_INDEX0) && selected <= 0) // map_primitive.cpp:146
            selected = -1;
        if (selected >= 0 && lane == selected)
            matched = true;
        if (lane == j)
            myMatch = selected;
    }
    flags = wave_or_u32(flags);
    if (lane == 0)
        out.flags = flags;
    if (lane < MP)
        out.match[lane] = (lane < npv && !(flags & CAPE_MATCH_EXACT_OVERFLOW)) ? myMatch : -1;
}

template <int TIER> static size_t tier_lds_bytes()
{
    using T = Tier<TIER>;
    const size_t b = (size_t)2 * T::kRing * sizeof(double2) + (size_t)2 * T::kRing * sizeof(Edge) + (size_t)T::kXs * 8 +
                     (size_t)64 * T::kStack * 8 + (size_t)128 * T::kStack * (8 + 2 + 2 + 1) + (size_t)(6 * T::kRing + 2 + 128) * 4;
    return ((b + 15) & ~(size_t)15) + 32; // + MpLds::sh
}

template <int TIER> static hipError_t launch_tier(const MatchPolygonParams& p, int blocks, hipStream_t stream)
{
    const int lds = (int)tier_lds_bytes<TIER>();
    // (a cooperative tier's four waves share ONE carve)
    hipLaunchKernelGGL(cape_polygon_inter_kernel<TIER>, dim3(blocks), dim3(64 * Tier<TIER>::kWavesPerGroup),
                       (size_t)lds * (Tier<TIER>::kCoop ? 1 : Tier<TIER>::kWavesPerGroup), stream, p, lds);
    return hipGetLastError();
}

hipError_t launch_match_polygons(const MatchPolygonParams& p, int nFrames, hipStream_t stream)
{
    if (const hipError_t e = hipMemsetAsync(p.listCounts, 0, 32 * sizeof(unsigned), stream); e != hipSuccess)
        return e;
    hipLaunchKernelGGL(cape_polygon_gate_kernel, dim3((nFrames + kGateFrames - 1) / kGateFrames), dim3(64 * kGateFrames), 0, stream, p, nFrames);
    if (const hipError_t e = hipGetLastError(); e != hipSuccess)
        return e;
    // persistent grids: as many workgroups as the chip holds at once
    const int cus = p.computeUnits > 0 ? p.computeUnits : 256;
    const int maxPairs = nFrames * MP * MP;
    auto blocks_for = [&](int perCu, int pairsPerGroup) { // pairs a workgroup works on at a time: its waves, or one (cooperative tiers)
        const int need = (maxPairs + pairsPerGroup - 1) / pairsPerGroup;
        return need < cus * perCu ? need : cus * perCu;
    };
    if (const hipError_t e = launch_tier<0>(p, blocks_for(Tier<0>::kGroupsPerCu, Tier<0>::kCoop ? 1 : Tier<0>::kWavesPerGroup), stream); e != hipSuccess)
        return e;
    if (const hipError_t e = launch_tier<1>(p, blocks_for(Tier<1>::kGroupsPerCu, Tier<1>::kCoop ? 1 : Tier<1>::kWavesPerGroup), stream); e != hipSuccess)
        return e;
    if (const hipError_t e = launch_tier<2>(p, blocks_for(Tier<2>::kGroupsPerCu, Tier<2>::kCoop ? 1 : Tier<2>::kWavesPerGroup), stream); e != hipSuccess)
        return e;
    if (p.boundaryCapacity > Tier<2>::kRing && tier_lds_bytes<3>() <= (size_t)p.ldsLimitBytes) // (a ring is a subset of its plane's candidates)
        if (const hipError_t e = launch_tier<3>(p, blocks_for(Tier<3>::kGroupsPerCu, Tier<3>::kCoop ? 1 : Tier<3>::kWavesPerGroup), stream); e != hipSuccess)
            return e;
    hipLaunchKernelGGL(cape_polygon_select_kernel, dim3((nFrames + kWaves - 1) / kWaves), dim3(64 * kWaves), 0, stream, p, nFrames);
    return hipGetLastError();
}

} // namespace cape
