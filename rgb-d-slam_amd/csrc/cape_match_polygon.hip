// "Next" row N2 with the reference's own area measure: MapPlane::find_matches (reference
// src/map_management/map_features/map_primitive.cpp:91-161, driven by feature_map.hpp:647-670) between CONSECUTIVE frames of a
// batch, on the boundary polygons built by cape_build_polygons -- `detectedPolygon.inter_area(projectedPolygon)` in mm^2
// (map_primitive.cpp:137 -> src/utils/polygon.cpp:525-545) instead of the shared cells of the label grids that
// cape_match_consecutive counts.  The planes of frame f-1 play the map planes, seen through the identity pose.
//
//   cape_polygon_inter_kernel  : one wavefront per (frame, previous plane j): for every plane i of the frame that passes
//        is_distance_similar / is_normal_similar (shape_primitives.cpp:66-86), the polygon of j is projected into the frame of
//        i (Polygon::project, polygon.cpp:338-382) and the area of the intersection of the two rings is computed.
//   cape_polygon_select_kernel : one wavefront per frame: the selection loop (greatest intersection above the overlap
//        threshold, is-matched flags updated between previous planes, the `selectedIndex <= 0` quirk).
//
// The intersection is this repo's host algorithm (host/boundary_polygon.cpp: rings_inter_area), statement for statement: the
// plane is cut into vertical slabs at every vertex and every edge crossing; inside a slab each ring is a stack of edges sorted
// by height and the overlap of the two stacks is a sum of trapezoids, added in slab order.  Lanes take the edge pairs (for the
// crossings), the compare-exchanges of a bitonic sort (slab boundaries) and one slab each (the trapezoids of 64 slabs are
// computed side by side, then added to the running area in order: the sum's rounding is observable).  + - x / and
// comparisons only: the areas are compared BIT FOR BIT with the host class (tests/test_gpu_match_polygon.py).
#include <hip/hip_runtime.h>
#include <cstdlib>

#include "cape_internal.h"
#include "cape_wave.h"

namespace cape {

namespace {

constexpr int kWaves = 2;          // independent waves per workgroup
constexpr int kRingCap = 128;      // vertices of one ring (simplified polygons: 13 on average, 90 at most on the test streams)
constexpr int kXsCap = 1024;       // slab boundaries: vertices of both rings + edge crossings
constexpr int kStackCap = 8;       // edges of one ring over one slab: first attempt (register arrays)
constexpr int kStackCapRetry = 16; // second attempt of the pairs that exceeded it (a lone wave per workgroup, 1 wave / SIMD)
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
    double2* ringA; // kRingCap
    double2* ringB; // kRingCap
    Edge* ea;       // kRingCap
    Edge* eb;       // kRingCap
    double* xs;     // kXsCap
    double* terms;  // 64 x (CAP/2)^2
    int* cand;      // 2 x kRingCap: the edges of either ring that reach into the current window of 64 slabs
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

// indices of the edges with a.x <= whi and b.x >= wlo, ascending
__device__ __forceinline__ int window_edges(const Edge* es, int ne, double wlo, double whi, int* out, int lane)
{
    int cnt = 0;
    for (int base = 0; base < ne; base += 64)
    {
        const int k = base + lane;
        const bool keep = k < ne && es[k].a.x <= whi && es[k].b.x >= wlo;
        const unsigned long long kb = __ballot(keep);
        if (keep)
            out[cnt + __popcll(kb & ((1ull << lane) - 1ull))] = k;
        cnt += __popcll(kb);
    }
    return cnt;
}

// the stack of one ring over the slab (x0, x1): its edges that span the slab, sorted by their height at the middle -- a
// STABLE insertion in edge order, like the insertion sort std::sort runs on so few elements.  Returns the count, or -1 if
// more than kStackCap edges span the slab.
template <int kStackCap>
__device__ __forceinline__ int build_stack(const Edge* es, const int* cand, int nc, double x0, double x1, double xm, double (&sy)[kStackCap],
                                           int (&se)[kStackCap])
{
    int cnt = 0;
    for (int c = 0; c < nc; ++c)
    {
        const int k = cand[c];
        const Edge e = es[k];
        if (!(e.a.x <= x0 && e.b.x >= x1))
            continue;
        if (cnt == kStackCap)
            return -1;
        double y = y_at(e, xm);
        int id = k;
        // placed in front of the first strictly greater height (equal ones stay in edge order: stable), the rest shifts up
        bool carrying = false;
#pragma unroll
        for (int q = 0; q < kStackCap; ++q)
        {
            if (q < cnt)
            {
                if (carrying || y < sy[q])
                {
                    const double ty = sy[q];
                    const int te = se[q];
                    sy[q] = y;
                    se[q] = id;
                    y = ty;
                    id = te;
                    carrying = true;
                }
            }
            else if (q == cnt)
            {
                sy[q] = y;
                se[q] = id;
            }
        }
        ++cnt;
    }
    return cnt;
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
template <int kStackCap>
__device__ inline double rings_inter_area(const MpLds& L, int na, int nb, int lane, unsigned long long* prof = nullptr)
{
    constexpr int kTermsPerSlab = (kStackCap / 2) * (kStackCap / 2);
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
    // slabs, 64 at a time: lane l computes the trapezoids of slab base + l, then they are added in slab order
    double area = 0.0;
    for (int base = 0; base + 1 < nx; base += 64)
    {
        const int s = base + lane;
        int myTerms = 0;
        // the edges that can span a slab of this window, in edge order (a superset: the stacks test every slab themselves)
        const int last = (base + 64 < nx - 1) ? base + 64 : nx - 1;
        const double wlo = L.xs[base], whi = L.xs[last];
        const int nca = window_edges(L.ea, nea, wlo, whi, L.cand, lane);
        const int ncb = window_edges(L.eb, neb, wlo, whi, L.cand + kRingCap, lane);
        CAPE_MP_SYNC();
        if (s + 1 < nx)
        {
            const double x0 = L.xs[s], x1 = L.xs[s + 1], xm = 0.5 * (x0 + x1);
            if (x1 > x0)
            {
                double ya[kStackCap], yb[kStackCap];
                int ia[kStackCap], ib[kStackCap];
                const int ca = build_stack<kStackCap>(L.ea, L.cand, nca, x0, x1, xm, ya, ia);
                const int cb = build_stack<kStackCap>(L.eb, L.cand + kRingCap, ncb, x0, x1, xm, yb, ib);
                if (ca < 0 || cb < 0)
                    myTerms = -1;
                else
                {
#pragma unroll
                    for (int i = 0; i + 1 < kStackCap; i += 2)
#pragma unroll
                        for (int j = 0; j + 1 < kStackCap; j += 2)
                        {
                            if (i + 1 < ca && j + 1 < cb)
                            {
                                const bool loA = ya[i] > yb[j];
                                const double loY = loA ? ya[i] : yb[j];
                                const Edge lo = loA ? L.ea[ia[i]] : L.eb[ib[j]];
                                const bool hiA = ya[i + 1] < yb[j + 1];
                                const double hiY = hiA ? ya[i + 1] : yb[j + 1];
                                const Edge hi = hiA ? L.ea[ia[i + 1]] : L.eb[ib[j + 1]];
                                if (!(hiY <= loY))
                                {
                                    const double h0 = y_at(hi, x0) - y_at(lo, x0);
                                    const double h1 = y_at(hi, x1) - y_at(lo, x1);
                                    L.terms[lane * kTermsPerSlab + myTerms] = 0.5 * (h0 + h1) * (x1 - x0);
                                    ++myTerms;
                                }
                            }
                        }
                }
            }
        }
        CAPE_MP_SYNC();
        if (__any(myTerms < 0))
            return nan_code(kNanStack);
        // the ordered sum: slab by slab, term by term.  The first two terms of a slab travel through registers (a slab of two
        // convex-ish outlines has one), the rest through LDS
        const double t0 = myTerms > 0 ? L.terms[lane * kTermsPerSlab] : 0.0, t1 = myTerms > 1 ? L.terms[lane * kTermsPerSlab + 1] : 0.0;
        const int slabs = (nx - 1 - base) < 64 ? (nx - 1 - base) : 64;
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

// the kept planes of a frame (output plane whose polygon Primitive_Detection keeps), in segment order: lane k < count holds
// the segment index of plane k
__device__ __forceinline__ int valid_planes(const cape_frame_record& rec, const cape_polygon* pol, int lane, int& mySeg)
{
    int nSeg = rec.header.n_plane_segments;
    nSeg = nSeg < 0 ? 0 : (nSeg > CAPE_MAX_PLANES ? CAPE_MAX_PLANES : nSeg);
    const bool ok = lane < nSeg && rec.segments[lane].is_output != 0 && (pol[lane].flags & CAPE_POLY_VALID) != 0 && pol[lane].vertex_count >= 3;
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

// One wavefront per frame: the kept planes of the frame and of its predecessor, the gates of every (previous plane j, plane i)
// pair (Plane::is_distance_similar / is_normal_similar on the planes' parametrisations, shape_primitives.cpp:66-86) and the
// work list of the pairs whose polygons are to be intersected.  A pair the gates reject holds -1.
__global__ __launch_bounds__(64 * kWaves) void cape_polygon_gate_kernel(MatchPolygonParams p, int nFrames)
{
    const int lane = threadIdx.x & 63;
    const int frame = blockIdx.x * kWaves + (threadIdx.x >> 6);
    if (frame >= nFrames)
        return;
    cape_frame_match_exact& out = p.matches[frame];
    const cape_frame_record& recC = p.records[frame];
    const cape_polygon* polC = p.polygons + (size_t)frame * CAPE_MAX_PLANES;
    int segC = -1, segP = -1;
    const int nCur = valid_planes(recC, polC, lane, segC);
    const int nPrev = frame > 0 ? valid_planes(p.records[frame - 1], polC - CAPE_MAX_PLANES, lane, segP) : 0;
    const bool fits = nCur <= MP && nPrev <= MP;
    if (lane == 0)
    {
        out.n_prev = nPrev;
        out.n_cur = nCur;
        out.flags = fits ? 0u : (uint32_t)CAPE_MATCH_EXACT_OVERFLOW;
        out.pad = 0;
    }
    if (lane < MP)
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
    }
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
        out.inter_area[j][i] = gated ? nan_code(kNanPending) : -1.0;
        const unsigned long long gb = __ballot(gated);
        if (gb)
        {
            unsigned base = 0;
            if (lane == 0)
                base = atomicAdd(&p.listCounts[0], (unsigned)__popcll(gb));
            base = __builtin_amdgcn_readfirstlane(base);
            if (gated)
                p.pairList[base + __popcll(gb & ((1ull << lane) - 1ull))] = pack_pair(frame, j, i);
        }
    }
}

// Persistent waves over a work list of pairs.  CAP = kStackCap: the gated pairs; CAP = kStackCapRetry: the pairs the first attempt
// left with kNanStack (it lists them).
template <int CAP, int WAVES>
__global__ __launch_bounds__(64 * WAVES) void cape_polygon_inter_kernel(MatchPolygonParams p, int ldsPerWave)
{
    constexpr bool kRetry = CAP != kStackCap;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_all[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    unsigned char* smem = smem_all + (size_t)wave * ldsPerWave;
    MpLds L;
    L.ringA = reinterpret_cast<double2*>(smem);
    L.ringB = L.ringA + kRingCap;
    L.ea = reinterpret_cast<Edge*>(L.ringB + kRingCap);
    L.eb = L.ea + kRingCap;
    L.xs = reinterpret_cast<double*>(L.eb + kRingCap);
    L.terms = L.xs + kXsCap;
    L.cand = reinterpret_cast<int*>(L.terms + 64 * (CAP / 2) * (CAP / 2));
    const unsigned* list = kRetry ? p.retryList : p.pairList;
    const unsigned count = p.listCounts[kRetry ? 1 : 0];
    for (unsigned t = blockIdx.x * WAVES + wave; t < count; t += gridDim.x * WAVES)
    {
        const unsigned pair = list[t];
        const int frame = (int)(pair >> 8), j = (int)((pair >> 4) & 15u), i = (int)(pair & 15u);
        cape_frame_match_exact& out = p.matches[frame];
        const int si = out.seg_cur[i], sj = out.seg_prev[j];
        const cape_polygon& PS = p.polygons[(size_t)frame * CAPE_MAX_PLANES + si];       // detected polygon
        const cape_polygon& PQ = p.polygons[(size_t)(frame - 1) * CAPE_MAX_PLANES + sj]; // projected polygon (identity pose)
        const int na = (int)PS.vertex_count, nb = (int)PQ.vertex_count;
        double result;
        if (na > kRingCap || nb > kRingCap)
            result = nan_code(kNanRing);
        else
        {
            const double2* vertsC = p.vertices + (size_t)frame * p.boundaryCapacity + PS.vertex_offset;
            const double2* vertsP = p.vertices + (size_t)(frame - 1) * p.boundaryCapacity + PQ.vertex_offset;
            for (int v = lane; v < na; v += 64)
                L.ringA[v] = vertsC[v];
            // Polygon::project (polygon.cpp:338-382): every vertex of the previous plane's ring lifted to 3-D and expressed in
            // the frame of plane i; the projected ring is re-oriented clockwise like every polygon (OpenRing constructor)
            for (int v = lane; v < nb; v += 64)
            {
                const double2 q = vertsP[v];
                const double X = PQ.center[0] + q.x * PQ.x_axis[0] + q.y * PQ.y_axis[0];
                const double Y = PQ.center[1] + q.x * PQ.x_axis[1] + q.y * PQ.y_axis[1];
                const double Z = PQ.center[2] + q.x * PQ.x_axis[2] + q.y * PQ.y_axis[2];
                const double dx = X - PS.center[0], dy = Y - PS.center[1], dz = Z - PS.center[2];
                L.ringB[v] = make_double2((PS.x_axis[0] * dx + PS.x_axis[1] * dy) + PS.x_axis[2] * dz,
                                          (PS.y_axis[0] * dx + PS.y_axis[1] * dy) + PS.y_axis[2] * dz);
            }
            CAPE_MP_SYNC();
            if (ring_area_signed(L.ringB, nb) > 0)
            {
                // reverse in place: lane v swaps v and nb - 1 - v
                for (int v = lane; v < nb / 2; v += 64)
                {
                    const double2 a = L.ringB[v], b = L.ringB[nb - 1 - v];
                    L.ringB[v] = b;
                    L.ringB[nb - 1 - v] = a;
                }
                CAPE_MP_SYNC();
            }
#ifdef CAPE_MP_PROFILE
            unsigned long long prof[8] = {0, 0, 0, 0, 0, 0, 0, 0};
            const unsigned long long tStart = __builtin_amdgcn_s_memtime();
            result = rings_inter_area<CAP>(L, na, nb, lane, prof);
            if (lane == 0 && !kRetry && i < 8 && j < 8)
            {
                // unused slots of the area matrix carry the ticks of this pair (profiles/match_polygons_bench.py decodes them)
                out.inter_area[j + 8][i + 8] = (double)(prof[3] - tStart) + 1e6 * na + 1e9 * nb + 1e12 * (double)prof[5];
                out.inter_area[j + 8][i] = (double)(prof[1] - prof[0]) + 1e6 * (double)(prof[2] - prof[1]);
                out.inter_area[j][i + 8] = (double)(prof[3] - prof[2]) + 1e6 * (double)prof[6];
            }
#else
            result = rings_inter_area<CAP>(L, na, nb, lane);
#endif
            CAPE_MP_SYNC();
        }
        if (lane == 0)
        {
            out.inter_area[j][i] = result;
            if (!kRetry && is_nan_code(result, kNanStack))
                p.retryList[atomicAdd(&p.listCounts[1], 1u)] = pair;
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

size_t match_polygon_lds_bytes(int cap)
{
    size_t b = (size_t)2 * kRingCap * sizeof(double2) + (size_t)2 * kRingCap * sizeof(Edge) + (size_t)kXsCap * 8 + (size_t)64 * (cap / 2) * (cap / 2) * 8 + (size_t)2 * kRingCap * 4;
    return (b + 15) & ~(size_t)15;
}

hipError_t launch_match_polygons(const MatchPolygonParams& p, int nFrames, hipStream_t stream)
{
    const int lds = (int)match_polygon_lds_bytes(kStackCap), ldsRetry = (int)match_polygon_lds_bytes(kStackCapRetry);
    if (const hipError_t e = hipMemsetAsync(p.listCounts, 0, 2 * sizeof(unsigned), stream); e != hipSuccess)
        return e;
    hipLaunchKernelGGL(cape_polygon_gate_kernel, dim3((nFrames + kWaves - 1) / kWaves), dim3(64 * kWaves), 0, stream, p, nFrames);
    if (const hipError_t e = hipGetLastError(); e != hipSuccess)
        return e;
    // persistent grids: as many workgroups as fit the chip at once (LDS: 2 of the first kind, 3 of the retry kind per CU)
    const int cus = p.computeUnits > 0 ? p.computeUnits : 256;
    int blocks = cus * 2;
    const int maxBlocks = (nFrames * MP * MP + kWaves - 1) / kWaves;
    blocks = blocks < maxBlocks ? blocks : maxBlocks;
    hipLaunchKernelGGL((cape_polygon_inter_kernel<kStackCap, kWaves>), dim3(blocks), dim3(64 * kWaves), (size_t)lds * kWaves, stream, p, lds);
    if (const hipError_t e = hipGetLastError(); e != hipSuccess)
        return e;
    {
        const char* eb = getenv("CAPE_MP_RETRY_BLOCKS");
        const char* el = getenv("CAPE_MP_RETRY_LDS");
        if (!(eb && atoi(eb) == 0))
            hipLaunchKernelGGL((cape_polygon_inter_kernel<kStackCapRetry, 1>), dim3(eb ? atoi(eb) : cus), dim3(64), (size_t)(el ? atoi(el) : ldsRetry), stream, p, ldsRetry);
    }
    if (const hipError_t e = hipGetLastError(); e != hipSuccess)
        return e;
    hipLaunchKernelGGL(cape_polygon_select_kernel, dim3((nFrames + kWaves - 1) / kWaves), dim3(64 * kWaves), 0, stream, p, nFrames);
    return hipGetLastError();
}

} // namespace cape
