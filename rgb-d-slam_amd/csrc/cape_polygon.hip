// "Next" row N1 on the device: the boundary polygon of every output plane, ONE WAVEFRONT PER PLANE.
//
// Replaces, for a batch, what the reference does on the host right after compute_plane_segment_boundary
// (primitive_detection.cpp:622): utils::Polygon(points, normal, center) -- reference src/utils/polygon.cpp:168-229:
// plane frame (:74-115), projection (:125-144, in reverse order :187-192), concave hull (:283-318 over
// third_party/concave_fitting.cpp:69-183: the Moreira-Santos k-nearest-neighbours walk on the k ladder 3,3,5,7,11,13,17,21
// with its DBL_EPSILON comparisons, its Intersects, its PointInPolygon -- zero-crossings quirk included -- and no duplicate
// removal on this call path), the repair of the hull into a closed clockwise ring (third_party/correct_boost_polygon.hpp),
// convex-hull fallback (:268-281), area (:453-461) and simplify (:578-601).  The statements are those of this repo's
// dependency-free host class (host/boundary_polygon.cpp), in the same operation order: + - x / and comparisons only (no libm
// call whose rounding could differ between glibc and ocml), so the vertices are compared BIT FOR BIT against the host class
// (tests/test_gpu_polygon.py) -- and both against the test suite's independent restatement of the reference's files
// (tests/test_gpu_polygon_oracle.py).  Two things cannot be the reference's: FLANN's approximate search over randomized
// kd-trees (here: the exact k nearest points) and the `-atan2` ordering of the candidates (here: exact turn predicates, which
// order like the angles and cannot disagree between host and device on a near-tie).
//
// Layout: the points of a plane are its boundary candidates (<= kPolyMaxPoints), projected and sorted in LDS; a hull is a
// list of point indices.  Lanes are parallel over points (distances, point-in-ring tests), over hull edges (intersection
// tests) and over the compare-exchanges of a bitonic sort; what the algorithm orders sequentially (the walk along the
// hull, the candidate tried first) is uniform control flow.  No workgroup barrier: a workgroup carries four independent
// waves, like the grow kernel.
#include <hip/hip_runtime.h>

#include "cape_device.h"
#include "cape_internal.h"
#include "cape_wave.h"

#include <algorithm>

namespace cape {

constexpr int kPolyWavesPerGroup = 4;
// Two instances: planes of up to kPolySmallPoints boundary candidates (every plane of the 640x480 test streams: at most 175)
// are built by waves that hold 7 KB of LDS each, twenty to a CU; a plane with more goes to the kPolyMaxPoints
// instance (27 KB per wave) launched right behind.  With one instance sized for the worst case a CU held four waves.
constexpr int kPolySmallPoints = 256;
constexpr int kPolyCutPoints = 8;   // crossing points a self-crossing hull may add to its plane's points (finish_polygon)
constexpr int kPolySortSelect = 5;  // neighbours from which the k-nearest selection sorts the lanes' keys instead of taking k minima
enum PolyList
{
    kPolyFirstRung = 0, // list 0: the planes of up to 256 candidates (static list of the task kernel)
    kPolyFull = 1       // list 1: planes of 257 .. 1 024 candidates (the large instance: the whole ladder in one wave)
};                      // (the task kernel's queue of spawned (plane, rung) tasks has a region of its own: PolygonParams::queue)

#ifdef CAPE_POLY_PROFILE
#define CAPE_PTICK(k)                                                                     \
    do                                                                                    \
    {                                                                                     \
        const unsigned long long _n = __builtin_amdgcn_s_memtime();                       \
        if (lane == 0 && p.prof)                                                          \
            atomicAdd(&p.prof[(size_t)frame * kProfileSlots + (k)], _n - _pt);            \
        _pt = _n;                                                                         \
    } while (0)
#define CAPE_PCOUNT(k, v)                                                                 \
    do                                                                                    \
    {                                                                                     \
        if (lane == 0 && p.prof)                                                          \
            atomicAdd(&p.prof[(size_t)frame * kProfileSlots + (k)], (unsigned long long)(v)); \
    } while (0)
#else
#define CAPE_PTICK(k)
#define CAPE_PCOUNT(k, v)
#endif

#define CAPE_POLY_SYNC()                                                                                      \
    do                                                                                                       \
    {                                                                                                        \
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");                                               \
        __builtin_amdgcn_s_waitcnt(0);                                                                       \
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");                                               \
    } while (0)

struct PolyLds
{
    double2* pts;          // kPolyMaxPoints (+ padding of the sort) projected points, sorted, deduplicated
    unsigned short* hull;  // kPolyMaxPoints + 1 point indices
    unsigned short* ring;  // kPolyMaxPoints + 1 point indices: the ring being built / simplified (closed copy for Douglas-Peucker)
    unsigned char* used;   // kPolyMaxPoints
    unsigned char* keep;   // kPolyMaxPoints + 1
    unsigned int* stack;   // kPolyMaxPoints range stack of the simplification
};

__device__ __forceinline__ double pcross2(const double2& o, const double2& a, const double2& b)
{
    return (a.x - o.x) * (b.y - o.y) - (a.y - o.y) * (b.x - o.x);
}
__device__ __forceinline__ double pmin(double a, double b) { return (b < a) ? b : a; } // std::min
__device__ __forceinline__ double pmax(double a, double b) { return (a < b) ? b : a; } // std::max

// host/boundary_polygon.cpp: segments_intersect -- proper or touching intersection of the open segments; shared endpoints
// do not count
__device__ __forceinline__ bool segments_intersect(const double2& a1, const double2& a2, const double2& b1, const double2& b2)
{
    if (pmax(a1.x, a2.x) < pmin(b1.x, b2.x) || pmax(b1.x, b2.x) < pmin(a1.x, a2.x) || pmax(a1.y, a2.y) < pmin(b1.y, b2.y) ||
        pmax(b1.y, b2.y) < pmin(a1.y, a2.y))
        return false;
    auto same = [](const double2& p, const double2& q) { return p.x == q.x && p.y == q.y; };
    if (same(a1, b1) || same(a1, b2) || same(a2, b1) || same(a2, b2))
        return false;
    const double d1 = pcross2(b1, b2, a1), d2 = pcross2(b1, b2, a2), d3 = pcross2(a1, a2, b1), d4 = pcross2(a1, a2, b2);
    if (((d1 > 0 && d2 < 0) || (d1 < 0 && d2 > 0)) && ((d3 > 0 && d4 < 0) || (d3 < 0 && d4 > 0)))
        return true;
    auto on = [](const double2& p, const double2& q, const double2& r) {
        return pmin(p.x, q.x) <= r.x && r.x <= pmax(p.x, q.x) && pmin(p.y, q.y) <= r.y && r.y <= pmax(p.y, q.y);
    };
    if (d1 == 0 && on(b1, b2, a1))
        return true;
    if (d2 == 0 && on(b1, b2, a2))
        return true;
    if (d3 == 0 && on(a1, a2, b1))
        return true;
    if (d4 == 0 && on(a1, a2, b2))
        return true;
    return false;
}

// ---- third_party/concave_fitting.cpp:186-201: the hull's comparisons carry a DBL_EPSILON slack (host: eq_eps ... points_equal)
constexpr double kHullEps = 2.220446049250313e-16;
__device__ __forceinline__ bool eq_eps(double a, double b) { return fabs(a - b) <= kHullEps; }
__device__ __forceinline__ bool zero_eps(double a) { return fabs(a) <= kHullEps; }
__device__ __forceinline__ bool lt_eps(double a, double b) { return a < (b - kHullEps); }
__device__ __forceinline__ bool le_eps(double a, double b) { return a <= (b + kHullEps); }
__device__ __forceinline__ bool gt_eps(double a, double b) { return a > (b + kHullEps); }
__device__ __forceinline__ bool points_equal(const double2& a, const double2& b) { return eq_eps(a.x, b.x) && eq_eps(a.y, b.y); }

// host/boundary_polygon.cpp: hull_edges_intersect = Intersects of concave_fitting.cpp:426-463 (crossing point of the carrier
// lines, eight bounding tests with the slack; parallel segments never intersect).  Boxes more than 1e-9 apart cannot both
// hold the crossing point: no division for nearly every pair.
__device__ __forceinline__ bool hull_edges_intersect(const double2& a1p, const double2& a2p, const double2& b1p, const double2& b2p)
{
    const double ax1 = a1p.x, ay1 = a1p.y, ax2 = a2p.x, ay2 = a2p.y;
    const double bx1 = b1p.x, by1 = b1p.y, bx2 = b2p.x, by2 = b2p.y;
    const double aminx = pmin(ax1, ax2), amaxx = pmax(ax1, ax2), aminy = pmin(ay1, ay2), amaxy = pmax(ay1, ay2);
    const double bminx = pmin(bx1, bx2), bmaxx = pmax(bx1, bx2), bminy = pmin(by1, by2), bmaxy = pmax(by1, by2);
    if (bminx - amaxx > 1e-9 || aminx - bmaxx > 1e-9 || bminy - amaxy > 1e-9 || aminy - bmaxy > 1e-9)
        return false;
    const double a1 = ay2 - ay1;
    const double b1 = ax1 - ax2;
    const double c1 = a1 * ax1 + b1 * ay1;
    const double a2 = by2 - by1;
    const double b2 = bx1 - bx2;
    const double c2 = a2 * bx1 + b2 * by1;
    const double det = a1 * b2 - a2 * b1;
    if (zero_eps(det))
        return false;
    const double x = (b2 * c1 - b1 * c2) / det;
    const double y = (a1 * c2 - a2 * c1) / det;
    return le_eps(aminx, x) && le_eps(x, amaxx) && le_eps(aminy, y) && le_eps(y, amaxy) && le_eps(bminx, x) && le_eps(x, bmaxx) &&
           le_eps(bminy, y) && le_eps(y, bmaxy);
}

// host: point_in_hull = PointInPolygon of concave_fitting.cpp:393-423 over the hull list as the walk left it (consecutive
// pairs, no wrap), zero-crossings quirk included.  One lane, whole hull.
__device__ __forceinline__ bool point_in_hull(const double2& p, const double2* pts, const unsigned short* hull, int hs)
{
    if (hs <= 2)
        return false;
    const double x = p.x, y = p.y;
    int inout = 0;
    double2 q0 = pts[hull[0]];
    for (int v = 0; v + 1 < hs; ++v)
    {
        const double2 q1 = pts[hull[v + 1]];
        if (((le_eps(q0.y, y) && lt_eps(y, q1.y)) || (le_eps(q1.y, y) && lt_eps(y, q0.y))) && !zero_eps(q1.y - q0.y) &&
            lt_eps(x, q0.x + ((q1.x - q0.x) * (y - q0.y) / (q1.y - q0.y))))
            inout++;
        q0 = q1;
    }
    if (inout == 0)
        return true;
    return (inout & 1) != 0;
}

// host: segment_distance2 (Boost's projected_point strategy, comparable form)
__device__ __forceinline__ double segment_distance2(const double2& p, const double2& a, const double2& b)
{
    const double vx = b.x - a.x, vy = b.y - a.y, wx = p.x - a.x, wy = p.y - a.y;
    const double c1 = wx * vx + wy * vy;
    if (c1 <= 0)
        return wx * wx + wy * wy;
    const double c2 = vx * vx + vy * vy;
    if (c2 <= c1)
    {
        const double ux = p.x - b.x, uy = p.y - b.y;
        return ux * ux + uy * uy;
    }
    const double t = c1 / c2;
    const double qx = a.x + t * vx, qy = a.y + t * vy;
    return (p.x - qx) * (p.x - qx) + (p.y - qy) * (p.y - qy);
}

// ring_area_signed: ordered sum, one rounding per add (uniform: every lane walks the ring)
__device__ __forceinline__ double ring_area_signed(const double2* pts, const unsigned short* ring, int n)
{
    double s = 0;
    for (int i = 0, j = n - 1; i < n; j = i++)
    {
        const double2 ri = pts[ring[i]], rj = pts[ring[j]];
        s += (rj.x * ri.y - ri.x * rj.y);
    }
    return 0.5 * s;
}

// ring_is_simple: no two non-adjacent edges touch, and the area is not zero.  Lanes over the first edge of a pair.
__device__ __forceinline__ bool ring_is_simple(const double2* pts, const unsigned short* ring, int n, int lane)
{
    if (n < 3)
        return false;
    bool bad = false;
    // edge i meets the edges j > i: the first edges have the longest lists, so a lane takes edge t from the front on even
    // rounds and from the back on odd ones
    for (int base = 0, round = 0; base < n; base += 64, ++round)
    {
        const int i = base + ((round & 1) ? 63 - lane : lane);
        if (i >= n)
            continue;
        const double2 a1 = pts[ring[i]], a2 = pts[ring[(i + 1) % n]];
        for (int j = i + 1; j < n && !bad; ++j)
        {
            if (j == i + 1 || (i == 0 && j == n - 1))
                continue; // adjacent edges share a vertex
            bad = segments_intersect(a1, a2, pts[ring[j]], pts[ring[(j + 1) % n]]);
        }
    }
    if (__any(bad))
        return false;
    return fabs(ring_area_signed(pts, ring, n)) > 0;
}

// host: first_contact -- the first pair (i, j), i < j, of non-adjacent edges of the open ring that share a point, in (i, j) order
// (lanes over i, a wave minimum over the packed pair); `proper`: the two edges cross at a point interior to both
__device__ inline bool first_contact(const double2* pts, const unsigned short* ring, int n, int lane, int& ci, int& cj, bool& proper)
{
    unsigned best = 0xFFFFFFFFu;
    for (int base = 0; base < n; base += 64)
    {
        const int i = base + lane;
        if (i < n)
        {
            const double2 a1 = pts[ring[i]], a2 = pts[ring[(i + 1) % n]];
            for (int j = i + 1; j < n; ++j)
            {
                if (j == i + 1 || (i == 0 && j == n - 1))
                    continue;
                if (segments_intersect(a1, a2, pts[ring[j]], pts[ring[(j + 1) % n]]))
                {
                    const unsigned key = ((unsigned)i << 16) | (unsigned)j;
                    best = key < best ? key : best;
                    break;
                }
            }
        }
    }
    const unsigned m = 0xFFFFFFFFu - wave_max_u32(0xFFFFFFFFu - best);
    if (m == 0xFFFFFFFFu)
        return false;
    ci = (int)(m >> 16);
    cj = (int)(m & 0xFFFFu);
    const double2 a1 = pts[ring[ci]], a2 = pts[ring[(ci + 1) % n]], b1 = pts[ring[cj]], b2 = pts[ring[(cj + 1) % n]];
    const double d1 = pcross2(b1, b2, a1), d2 = pcross2(b1, b2, a2), d3 = pcross2(a1, a2, b1), d4 = pcross2(a1, a2, b2);
    proper = ((d1 > 0 && d2 < 0) || (d1 < 0 && d2 > 0)) && ((d3 > 0 && d4 < 0) || (d3 < 0 && d4 > 0));
    return true;
}

// wave-wide arg-min of (a, b, idx) in lexicographic order (a, b: doubles without NaN)
__device__ __forceinline__ int wave_argmin2(double a, double b, int idx)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1)
    {
        const double oa = __shfl_xor(a, o), ob = __shfl_xor(b, o);
        const int oi = __shfl_xor(idx, o);
        const bool take = (oa < a) || (oa == a && (ob < b || (ob == b && oi < idx)));
        if (take)
        {
            a = oa;
            b = ob;
            idx = oi;
        }
    }
    return __builtin_amdgcn_readfirstlane(idx);
}

// Clockwise rotation from the direction `P` to the vector `V`, as a class and an exact in-class comparison (the host class
// orders candidates with the same predicates: no atan2, so host and device cannot disagree on a near-tie):
//   0: same direction   1: strictly clockwise, less than half a turn   2: opposite   3: more than half a turn
// Directions closer than 2 DBL_EPSILON radians are one direction (host class: same_direction_line -- what the reference's slack on
// its atan2 angles makes of them): sin^2 <= (2 eps)^2 in products, behind a cheap test that cannot hide a case while |a| |b| < 2.2e12 mm^2.
// (the product test sits behind WAVE-UNIFORM branches: left to itself the compiler turns the cheap first test into a select and
//  pays the seven products on every comparison -- the walk measured 20 % slower)
#ifndef CAPE_POLY_EPS
#define CAPE_POLY_EPS 1 // A/B knob: 0 = exact cross products only (round 5)
#endif
__device__ __forceinline__ bool same_direction_products(double cr, double ax, double ay, double bx, double by)
{
    constexpr double kEps2 = 1.9721522630525295135293214132069655741830160877724e-31; // 2^-102 = (2 DBL_EPSILON)^2
    return cr * cr <= kEps2 * ((ax * ax + ay * ay) * (bx * bx + by * by));
}
// lane-parallel: every lane classifies its own candidate
__device__ __forceinline__ int turn_class(double px, double py, double vx, double vy)
{
    const double cr = px * vy - py * vx, dt = px * vx + py * vy;
    int cls = cr < 0 ? 1 : (cr > 0 ? 3 : (dt > 0 ? 0 : 2));
    const bool nearLine = CAPE_POLY_EPS != 0 && fabs(cr) < 1e-3;
    if (__any(nearLine))
    {
        if (nearLine && same_direction_products(cr, px, py, vx, vy))
            cls = dt > 0 ? 0 : 2;
    }
    return cls;
}
// does candidate a turn further clockwise from P than candidate b (strictly)?  (uniform arguments: read out of the candidates' lanes)
// The scan below runs this plain form -- the sign of one cross product, a total order inside an open half turn -- and then checks, every
// untried candidate on its own lane, whether anything points within 1e-3 of a cross product of the winner's direction; only then (a few
// steps in a thousand) is the scan repeated with the careful form.  (If nothing does, the winner is the plain maximum and compares the
// same way under both forms with every candidate, so the careful scan would elect it too.)  Bookkeeping inside the comparisons cost
// the walk 6 %, a branch per comparison 15 % (profiles/r06_polygon_eps_ab.txt).
__device__ __forceinline__ bool turns_further(int ca, double ax, double ay, int cb, double bx, double by)
{
    if (ca != cb)
        return ca > cb;
    if (ca == 0 || ca == 2)
        return false;
    return (bx * ay - by * ax) < 0; // cross(Vb, Va) < 0: Va lies clockwise of Vb inside the same open half turn
}
__device__ __forceinline__ bool turns_further_careful(int ca, double ax, double ay, int cb, double bx, double by)
{
    if (ca != cb)
        return ca > cb;
    if (ca == 0 || ca == 2)
        return false;
    const double cr = bx * ay - by * ax;
    return cr < 0 && !(cr > -1e-3 && same_direction_products(cr, bx, by, ax, ay));
}

// One run of the k-nearest-neighbours walk (host: concave_hull_k = ConcaveHull of concave_fitting.cpp:93-183).  On success the
// hull's point indices are in L.hull[0, hs) exactly as the reference's vector holds them: ending with the start point again
// when the walk came back to it, every point once when it ran out of points first.
// `planeState` (task kernel only, else null): the plane's state word in global memory; a rung gives up as soon as a lower rung
// has its hull (or the plane is finalised) -- it can no longer win.  Looked at every sixteenth step (a read-modify-write: ~2 us).
#ifdef CAPE_POLY_WALK_NOINLINE
#define CAPE_WALK_ATTR __attribute__((noinline))
#else
#define CAPE_WALK_ATTR inline
#endif
// SLOTS: points a lane owns (i = lane + 64 j, j < SLOTS).  EXACT: the caller guarantees 64 (SLOTS - 1) < n <= 64 SLOTS, so no
// pass asks whether a slot exists -- the task kernel picks the instance by the plane's size (a median plane has 74 candidates:
// two slots, not four with two of them switched off by uniform masks in every loop of every step).
template <int CAP, int SLOTS, bool EXACT>
__device__ CAPE_WALK_ATTR bool concave_hull_k(const PolyLds& L, int n, int first, int k, int lane, int& hsOut, const uint32_t* planeState = nullptr,
                                              int myRung = 0)
{
    constexpr int kPolyPerLane = SLOTS; // points a lane owns in the lane-parallel passes
    const double2* pts = L.pts;
    if (n < 3)
    {
        hsOut = 0;
        return true;
    }
    if (n == 3)
    {
        if (lane < 3)
            L.hull[lane] = (unsigned short)lane;
        CAPE_POLY_SYNC();
        hsOut = 3;
        return true;
    }
    const double2 firstPt = pts[first];
    // Every lane keeps ITS points (i = lane + 64 j) in registers for the whole walk, and which of them are hidden from the
    // neighbour search as a bit mask: the walk's inner loop reads LDS only for the hull edges.  (A lone wave -- the end of a
    // batch, a one-frame call -- pays ~120 cycles per dependent LDS round trip; a step had eight of them.)
    // (the 1 024-point instance has sixteen points per lane: those stay in LDS, only the mask lives in a register)
    constexpr bool kInRegs = kPolyPerLane <= 4;
    double2 myPts[kInRegs ? kPolyPerLane : 1];
    unsigned myUsed = 0; // bit j: point lane + 64 j is hidden (on the hull)
    if (kInRegs)
    {
#pragma unroll
        for (int j = 0; j < (kInRegs ? kPolyPerLane : 1); ++j)
        {
            const int i = lane + 64 * j;
            myPts[j] = ((EXACT || 64 * j < n) && i < n) ? pts[i] : make_double2(0.0, 0.0);
        }
    }
    auto my_point = [&](int j) { return kInRegs ? myPts[kInRegs ? j : 0] : pts[lane + 64 * j]; };
    // the point with index idx, from its owner's registers (idx is uniform)
    auto point_of = [&](int idx) {
        if (!kInRegs)
            return pts[idx];
        const int owner = idx & 63, slot = idx >> 6;
        double2 q = make_double2(0.0, 0.0);
#pragma unroll
        for (int j = 0; j < (kInRegs ? kPolyPerLane : 1); ++j)
            if (slot == j)
                q = make_double2(readlane_f64(myPts[j].x, owner), readlane_f64(myPts[j].y, owner));
        return q;
    };
    auto hide = [&](int idx, bool hidden) {
        if (lane == (idx & 63))
            myUsed = hidden ? (myUsed | (1u << (idx >> 6))) : (myUsed & ~(1u << (idx >> 6)));
    };
    if (lane == 0)
        L.hull[0] = (unsigned short)first;
    hide(first, true);
    int hs = 1;
    int usedCount = 1; // points hidden from the neighbour search
    int current = first;
    double2 cur = firstPt;
    double Px = 1.0, Py = 0.0; // prevAngle = 0: the +x direction
    int step = 1;
    while ((!points_equal(cur, firstPt) || step == 1) && hs != n)
    {
        if (planeState && (step & 15) == 0)
        {
            uint32_t st = 0;
            if (lane == 0)
                st = atomicAdd(const_cast<uint32_t*>(planeState), 0u); // (as the atomics see it: see fresh_u32)
            st = (uint32_t)__builtin_amdgcn_readfirstlane((int)st);
            if ((st & (1u << 16)) || (((st >> 8) & 0xFFu) & ((1u << myRung) - 1u)))
                return false;
        }
        if (step == 4)
        {
            hide(first, false); // the start point is put back into the index once the hull has three edges
            --usedCount;
        }
        // ---- k nearest visible neighbours of the current point, ascending (squared distance, index) -- EXACTLY that order.
        // Fast key = the squared distance's bit pattern (>= +0: the bits order like the value) with its ten lowest mantissa bits
        // replaced by the point index: ONE 64-bit wave minimum per neighbour.  Two keys order like (distance, index) unless they
        // share their upper 54 bits (a "bucket": distances within 2^-42 of each other); every selection below notices when a
        // bucket it touched holds a second visible key (`sameBucket`) and the step is then selected again on the full bit patterns.
        unsigned long long key[kPolyPerLane];
#pragma unroll
        for (int j = 0; j < kPolyPerLane; ++j)
        {
            const int i = lane + 64 * j;
            key[j] = ~0ull;
            if (!EXACT && 64 * j >= n)
                continue; // (uniform: a plane of 150 points uses three of the sixteen slots)
            {
                // (no branch around the arithmetic: a select at the end -- the exec-mask bookkeeping of four predicated blocks cost
                //  more than the five f64 instructions they guarded)
                const double2 q = my_point(j);
                const double dx = cur.x - q.x, dy = cur.y - q.y;
                const unsigned long long kq = ((unsigned long long)__double_as_longlong(dx * dx + dy * dy) & ~1023ull) | (unsigned long long)i;
                key[j] = (i < n && !((myUsed >> j) & 1u)) ? kq : ~0ull;
            }
        }
        const int cnt = n - usedCount;
        if (cnt <= 0)
            return false; // no neighbour left: the reference leaves its candidate loop with `its` still set
        const int kk = k < cnt ? k : cnt;
        // candidate c lives in lane c (kk <= 21 < 64): index, edge vector, turn class
        int myCand = 0, myClass = 0;
        double myVx = 0.0, myVy = 0.0, myPx = 0.0, myPy = 0.0; // candidate c's point lives in lane c too
        bool selected = false;
        bool sameBucket = false; // (uniform)
        if (kk >= kPolySortSelect)
        {
            // Many neighbours: ONE sort instead of kk minima.  Every lane offers its smallest key; sorted across the wave, lane c
            // holds the c-th smallest of them -- and these are the kk nearest points if no lane hides a second key below the
            // kk-th.  Otherwise the minima below decide.
            unsigned long long head = ~0ull, second = ~0ull;
#pragma unroll
            for (int j = 0; j < kPolyPerLane; ++j)
                if (EXACT || 64 * j < n)
                {
                    const unsigned long long kj = key[j];
                    if (kj < head)
                    {
                        second = head;
                        head = kj;
                    }
                    else if (kj < second)
                        second = kj;
                }
            const unsigned long long sorted = wave_sort_u64(head, lane);
            const unsigned long long hidden = wave_min_u64(second);
            // one of the kk smallest heads shares its bucket with the next head, or with the smallest key behind the heads
            const unsigned long long after = wave_from_lane_above(sorted); // (lanes below kk <= 21 are the ones that look)
            sameBucket = __any(lane < kk && (((sorted ^ after) >> 10) == 0 || ((sorted ^ hidden) >> 10) == 0));
            if (__popcll(__ballot(sorted < hidden)) >= kk && !sameBucket)
            {
                selected = true;
                if (lane < kk)
                {
                    const int idx = (int)(sorted & 1023ull);
                    const double2 q = pts[idx];
                    myCand = idx;
                    myPx = q.x;
                    myPy = q.y;
                    myVx = q.x - cur.x;
                    myVy = q.y - cur.y;
                }
            }
        }
        if (!selected && !sameBucket)
        {
            unsigned long long prevMin = ~0ull;
            for (int c = 0; c < kk; ++c)
            {
                unsigned long long best = ~0ull;
#pragma unroll
                for (int j = 0; j < kPolyPerLane; ++j)
                    if ((EXACT || 64 * j < n) && key[j] < best)
                        best = key[j];
                const unsigned long long m = wave_min_u64_lead(best);
                sameBucket |= ((m ^ prevMin) >> 10) == 0; // two minima in a row out of one bucket
                prevMin = m;
                const int idx = (int)(m & 1023ull);
#pragma unroll
                for (int j = 0; j < kPolyPerLane; ++j)
                    if ((EXACT || 64 * j < n) && key[j] == m)
                        key[j] = ~0ull; // taken (keys are unique: they carry the index)
                const double2 q = point_of(idx);
                if (lane == c)
                {
                    myCand = idx;
                    myPx = q.x;
                    myPy = q.y;
                    myVx = q.x - cur.x;
                    myVy = q.y - cur.y;
                }
            }
            // ... or the last one shares its bucket with a key that stayed behind
            bool behind = false;
#pragma unroll
            for (int j = 0; j < kPolyPerLane; ++j)
                if ((EXACT || 64 * j < n) && ((key[j] ^ prevMin) >> 10) == 0)
                    behind = true;
            sameBucket |= __any(behind) != 0;
        }
        if (sameBucket)
        {
            // The rare step (distances within 2^-42 of each other among the nearest): kk minima of (bit pattern of the squared
            // distance, index), the order the host class and the oracle sort by.  Nothing is kept from the attempt above.
            // (the start point, back in the index from step 4 on, carries the id n like the reference's copy, concave_fitting.cpp:131:
            //  in an exact tie it comes last)
            unsigned taken = 0; // bit j: my point lane + 64 j is among the neighbours already
            for (int c = 0; c < kk; ++c)
            {
                unsigned long long bestD = ~0ull;
                unsigned bestI = ~0u;
#pragma unroll
                for (int j = 0; j < kPolyPerLane; ++j)
                {
                    const int i = lane + 64 * j;
                    if ((EXACT || 64 * j < n) && i < n && !((myUsed >> j) & 1u) && !((taken >> j) & 1u))
                    {
                        const double2 q = my_point(j);
                        const double dx = cur.x - q.x, dy = cur.y - q.y;
                        const unsigned long long dq = (unsigned long long)__double_as_longlong(dx * dx + dy * dy);
                        const unsigned id = (i == first && step >= 4) ? (unsigned)n : (unsigned)i;
                        if (dq < bestD || (dq == bestD && id < bestI))
                        {
                            bestD = dq;
                            bestI = id;
                        }
                    }
                }
                const unsigned long long md = wave_min_u64(bestD);
                const int id = (int)wave_min_u32(bestD == md ? bestI : ~0u);
                const int idx = id == n ? first : id;
                if (lane == (idx & 63))
                    taken |= 1u << (idx >> 6);
                const double2 q = point_of(idx);
                if (lane == c)
                {
                    myCand = idx;
                    myPx = q.x;
                    myPy = q.y;
                    myVx = q.x - cur.x;
                    myVy = q.y - cur.y;
                }
            }
        }
        if (myVx == 0 && myVy == 0)
            myVx = 1.0; // a duplicate of the current point: atan2(+0, +0) = 0, the +x direction
        myClass = turn_class(Px, Py, myVx, myVy);
        // ---- candidates by decreasing clockwise turn from the previous edge (a scan in nearest-first order, like the host's);
        //      the first whose edge crosses no hull edge wins
        unsigned tried = 0;
        bool found = false;
        int next = 0;
        double2 nextPt = make_double2(0.0, 0.0);
        for (int t = 0; t < kk && !found; ++t)
        {
            int b = -1, bClass = 0;
            double bVx = 0.0, bVy = 0.0;
            for (int c = 0; c < kk; ++c)
            {
                if ((tried >> c) & 1u)
                    continue;
                const int cClass = __builtin_amdgcn_readlane(myClass, c);
                const double cVx = readlane_f64(myVx, c), cVy = readlane_f64(myVy, c);
                if (b < 0 || turns_further(cClass, cVx, cVy, bClass, bVx, bVy))
                {
                    b = c;
                    bClass = cClass;
                    bVx = cVx;
                    bVy = cVy;
                }
            }
            const bool nearWinner = CAPE_POLY_EPS != 0 && lane < kk && lane != b && !((tried >> lane) & 1u) && myClass == bClass && (bClass & 1) &&
                                    fabs(bVx * myVy - bVy * myVx) < 1e-3;
            if (__any(nearWinner))
            {
                b = -1;
                for (int c = 0; c < kk; ++c)
                {
                    if ((tried >> c) & 1u)
                        continue;
                    const int cClass = __builtin_amdgcn_readlane(myClass, c);
                    const double cVx = readlane_f64(myVx, c), cVy = readlane_f64(myVy, c);
                    if (b < 0 || turns_further_careful(cClass, cVx, cVy, bClass, bVx, bVy))
                    {
                        b = c;
                        bClass = cClass;
                        bVx = cVx;
                        bVy = cVy;
                    }
                }
            }
            tried |= 1u << b;
            const int cnd = __builtin_amdgcn_readlane(myCand, b);
            const double2 cp = make_double2(readlane_f64(myPx, b), readlane_f64(myPy, b));
            const int jFirst = points_equal(cp, firstPt) ? 1 : 0;
            bool its = false;
            // hull edges (h[j], h[j+1]), j in [jFirst, hs - 3]: not the edge that ends at the current point, and not the first edge
            // when the candidate is the start point (concave_fitting.cpp:146-163)
            for (int j = jFirst + lane; j + 2 < hs && !its; j += 64)
                its = hull_edges_intersect(cur, cp, pts[L.hull[j]], pts[L.hull[j + 1]]);
            if (!__any(its))
            {
                found = true;
                next = cnd;
                nextPt = cp;
            }
        }
        if (!found)
            return false;
        const double2 nx = nextPt;
        Px = cur.x - nx.x; // looking back along the new edge
        Py = cur.y - nx.y;
        if (Px == 0 && Py == 0)
            Px = 1.0;
        current = next;
        cur = nx;
        if (lane == 0)
            L.hull[hs] = (unsigned short)current;
        hide(current, true);
        ++hs;
        ++usedCount;
        ++step;
        CAPE_POLY_SYNC();
    }
    // every point that is not a hull vertex must pass PointInPolygon (the start point is a hull vertex whether or not it is
    // hidden from the search at this moment)
    bool outside = false;
#pragma unroll
    for (int j = 0; j < kPolyPerLane; ++j)
    {
        const int i = lane + 64 * j;
        if ((EXACT || 64 * j < n) && i < n && !((myUsed >> j) & 1u) && i != first && !point_in_hull(my_point(j), pts, L.hull, hs))
            outside = true;
    }
    if (__any(outside))
        return false;
    hsOut = hs;
    return true;
}

// the instance of the walk for a plane of n candidates (n uniform)
template <int CAP>
__device__ __forceinline__ bool concave_hull(const PolyLds& L, int n, int first, int k, int lane, int& hsOut, const uint32_t* planeState = nullptr, int myRung = 0)
{
    if constexpr (CAP == kPolySmallPoints && kPolySmallPoints == 256)
    {
        const int slots = __builtin_amdgcn_readfirstlane((n + 63) >> 6);
        if (slots <= 1)
            return concave_hull_k<CAP, 1, true>(L, n, first, k, lane, hsOut, planeState, myRung);
        if (slots == 2)
            return concave_hull_k<CAP, 2, true>(L, n, first, k, lane, hsOut, planeState, myRung);
        if (slots == 3)
            return concave_hull_k<CAP, 3, true>(L, n, first, k, lane, hsOut, planeState, myRung);
        return concave_hull_k<CAP, 4, true>(L, n, first, k, lane, hsOut, planeState, myRung);
    }
    else
        return concave_hull_k<CAP, CAP / 64, false>(L, n, first, k, lane, hsOut, planeState, myRung);
}

// host: find_min_y_point = FindMinYPoint of concave_fitting.cpp:231-243: std::min_element under (y ascending, x DESCENDING) with
// the slack.  A comparison with a slack is not transitive, so the scan is the host's sequential one: every lane runs it.
__device__ inline int find_min_y_point(const double2* pts, int n)
{
    int smallest = 0;
    double2 b = pts[0];
    for (int i = 1; i < n; ++i)
    {
        const double2 a = pts[i];
        const bool less = eq_eps(a.y, b.y) ? gt_eps(a.x, b.x) : lt_eps(a.y, b.y);
        if (less)
        {
            smallest = i;
            b = a;
        }
    }
    return smallest;
}

// bitonic sort of L.pts[0, n) by (x, y), ascending; n is padded to a power of two with +inf points
__device__ inline void sort_points(const PolyLds& L, int n, int lane)
{
    int np = 64;
    while (np < n)
        np <<= 1;
    for (int i = n + lane; i < np; i += 64)
        L.pts[i] = make_double2(__builtin_inf(), __builtin_inf());
    CAPE_POLY_SYNC();
    for (int size = 2; size <= np; size <<= 1)
        for (int stride = size >> 1; stride > 0; stride >>= 1)
        {
            for (int t = lane; t < np / 2; t += 64)
            {
                const int lo = (t / stride) * (2 * stride) + (t % stride), hi = lo + stride;
                const bool up = ((lo / size) & 1) == 0;
                const double2 a = L.pts[lo], b = L.pts[hi];
                const bool aGreater = (b.x < a.x) || (b.x == a.x && b.y < a.y);
                if (aGreater == up)
                {
                    L.pts[lo] = b;
                    L.pts[hi] = a;
                }
            }
            CAPE_POLY_SYNC();
        }
}

// ---------------------------------------------------------------------------------------------------------------------------
// One plane's context: where it comes from, its plane frame (get_plane_coordinate_system, polygon.cpp:74-115 with
// select_correct_transform :50-68), and what the constructor would do with it.
struct PlaneCtx
{
    int frame, seg, nPts;
    double cx, cy, cz, xax, xay, xaz, yax, yay, yaz;
    uint32_t flags; // CAPE_POLY_REJECTED / CAPE_POLY_OVERFLOW decided up front
};

__device__ inline PlaneCtx plane_context(const PolygonParams& p, int frame, int seg)
{
    PlaneCtx c;
    const cape_plane_segment& S = p.records[frame].segments[seg];
    c.frame = frame;
    c.seg = seg;
    c.nPts = (int)S.boundary_count;
    c.flags = 0;
    const double nx = S.normal[0], ny = S.normal[1], nz = S.normal[2];
    // the polygon's origin is Plane_Segment::get_center() = PlaneCoordinates::get_center() = normal * (-d), the point of the plane
    // closest to the camera (primitive_detection.cpp:622, plane_segment.hpp:90, plane_coordinates.hpp:52) -- not the centroid
    c.cx = p.originInCentroid ? S.centroid[0] : nx * (-S.d);
    c.cy = p.originInCentroid ? S.centroid[1] : ny * (-S.d);
    c.cz = p.originInCentroid ? S.centroid[2] : nz * (-S.d);
    const double distX = fabs(nx), distY = fabs(ny), distZ = fabs(nz);
    const double res = pmin(distX, pmin(distY, distZ));
    double rx, ry, rz;
    if (fabs(res - distX) <= 0.1)
        rx = 1, ry = 0, rz = 0;
    else if (fabs(res - distY) <= 0.1)
        rx = 0, ry = 1, rz = 0;
    else if (fabs(res - distZ) <= 0.1)
        rx = 0, ry = 0, rz = 1;
    else
    {
        rx = nz, ry = nx, rz = ny;
        const double nn = sqrt((rx * rx + ry * ry) + rz * rz);
        if (nn > 0)
            rx /= nn, ry /= nn, rz /= nn;
    }
    // xAxis = normalized(cross(normal, r)) ; yAxis = normalized(cross(normal, xAxis))
    double ax = ny * rz - nz * ry, ay = nz * rx - nx * rz, az = nx * ry - ny * rx;
    const double an = sqrt((ax * ax + ay * ay) + az * az);
    if (an > 0)
        ax /= an, ay /= an, az /= an;
    double bx = ny * az - nz * ay, by = nz * ax - nx * az, bz = nx * ay - ny * ax;
    const double bn = sqrt((bx * bx + by * by) + bz * bz);
    if (bn > 0)
        bx /= bn, by /= bn, bz /= bn;
    c.xax = ax, c.xay = ay, c.xaz = az, c.yax = bx, c.yay = by, c.yaz = bz;
    const double normalNorm = sqrt((nx * nx + ny * ny) + nz * nz);
    if (!(fabs(normalNorm - 1.0) <= 1e-9) || c.nPts < 3)
        c.flags |= CAPE_POLY_REJECTED; // the host constructor throws (normal not unit / fewer than 3 points)
    else if (c.nPts > kPolyMaxPoints)
        c.flags |= CAPE_POLY_OVERFLOW; // left to the host class
    return c;
}

// projection (polygon.cpp:125-144), in REVERSE order like the reference (:187-192); no sort, no duplicate removal
// (concave_fitting.cpp:69: the overload this path binds does not call RemoveDuplicates)
__device__ inline void project_points(const PolygonParams& p, const PlaneCtx& c, const PolyLds& L, int lane)
{
    const cape_plane_segment& S = p.records[c.frame].segments[c.seg];
    const double* bnd = p.boundary + ((size_t)c.frame * p.boundaryCapacity + S.boundary_offset) * 3;
    for (int i = lane; i < c.nPts; i += 64)
    {
        const double* q = bnd + 3 * (c.nPts - 1 - i);
        const double dx = q[0] - c.cx, dy = q[1] - c.cy, dz = q[2] - c.cz;
        L.pts[i] = make_double2((c.xax * dx + c.xay * dy) + c.xaz * dz, (c.yax * dx + c.yay * dy) + c.yaz * dz);
    }
    CAPE_POLY_SYNC();
}

__device__ inline void write_record(const PolygonParams& p, const PlaneCtx& c, double area, int count, uint32_t flags, int lane)
{
    if (lane == 0)
    {
        cape_polygon* out = &p.polygons[(size_t)c.frame * CAPE_MAX_PLANES + c.seg];
        out->x_axis[0] = c.xax; out->x_axis[1] = c.xay; out->x_axis[2] = c.xaz;
        out->y_axis[0] = c.yax; out->y_axis[1] = c.yay; out->y_axis[2] = c.yaz;
        out->center[0] = c.cx; out->center[1] = c.cy; out->center[2] = c.cz;
        out->area = area;
        out->vertex_offset = p.records[c.frame].segments[c.seg].boundary_offset;
        out->vertex_count = (uint32_t)count;
        out->flags = flags;
        out->segment = (uint32_t)c.seg;
    }
}

// Everything the constructor does once the ladder has spoken (polygon.cpp:195-231): the hull in L.hull[0, hs) -- or none --
// becomes the oriented ring, the convex hull takes over if that ring is no polygon, then area, simplify, validity and the
// stores.  `frame` / `_pt`: the profile build's counters.
template <int CAP> __device__ inline void finish_polygon(const PolygonParams& p, const PlaneCtx& c, const PolyLds& L, bool haveRing, int hs, int lane)
{
    constexpr int kPolyPerLane = CAP / 64;
    const int nPts = c.nPts;
    uint32_t flags = 0;
    double2* vout = p.vertices + (size_t)c.frame * p.boundaryCapacity + p.records[c.frame].segments[c.seg].boundary_offset;
    unsigned short* ring = L.ring;
    int rn = 0;
#ifdef CAPE_POLY_PROFILE
    const int frame = c.frame;
    unsigned long long _pt = __builtin_amdgcn_s_memtime();
#endif
    if (haveRing)
    {
        // The repair of the constructor (polygon.cpp:195-226 -> correct_boost_polygon.hpp:188-195, :172-186): the walk's ring,
        // closed, is reversed when it runs counter-clockwise -- it does, and it still starts at the walk's start point.  Kept
        // open here: the closing vertex goes.
        int m = hs;
        if (m > 1 && points_equal(L.pts[L.hull[m - 1]], L.pts[L.hull[0]]))
            --m;
        const bool rev = m >= 3 && ring_area_signed(L.pts, L.hull, m) > 0;
        for (int i = lane; i < m; i += 64)
            ring[i] = L.hull[(rev && i > 0) ? m - i : i];
        rn = m;
        CAPE_POLY_SYNC();
        // A hull that CROSSES itself (the reference's Intersects misses a crossing with an axis-parallel edge by a rounding error:
        // about one plane in a hundred) is cut apart at the crossing like the reference's repair does
        // (correct_boost_polygon.hpp:127-160, :199-330; host: dissolve_crossings): the crossing point X becomes a vertex, the ring
        // falls into r[0..i], X, r[j+1..] and X, r[i+1..j], the piece of greater |area| stays, clockwise -- at most eight cuts.  A
        // ring that merely touches itself goes the way of a failed hull: the convex hull.
        if (!ring_is_simple(L.pts, ring, rn, lane))
        {
            haveRing = false;
            bool cutting = rn >= 4;
            int extra = 0; // crossing points appended behind the plane's points
            for (int cut = 0; cut < 8 && cutting; ++cut)
            {
                int ci = 0, cj = 0;
                bool proper = false;
                if (!first_contact(L.pts, ring, rn, lane, ci, cj, proper))
                {
                    haveRing = rn >= 3 && fabs(ring_area_signed(L.pts, ring, rn)) > 0;
                    break;
                }
                if (!proper)
                    break;
                const double2 a = L.pts[ring[ci]], b = L.pts[ring[(ci + 1) % rn]], c2 = L.pts[ring[cj]], d = L.pts[ring[(cj + 1) % rn]];
                const double rx = b.x - a.x, ry = b.y - a.y, sx = d.x - c2.x, sy = d.y - c2.y;
                const double t = ((c2.x - a.x) * sy - (c2.y - a.y) * sx) / (rx * sy - ry * sx);
                const int xi = nPts + extra;
                ++extra;
                if (lane == 0)
                    L.pts[xi] = make_double2(a.x + t * rx, a.y + t * ry);
                // the two pieces, side by side in the hull area (outer: ci + 2 + rn - 1 - cj entries, loop: cj - ci + 1)
                unsigned short* outer = L.hull;
                const int no = ci + 1 + 1 + (rn - 1 - cj), nl = 1 + (cj - ci);
                unsigned short* loop = L.hull + no;
                for (int k = lane; k < no; k += 64)
                    outer[k] = k <= ci ? ring[k] : (k == ci + 1 ? (unsigned short)xi : ring[cj + 1 + (k - ci - 2)]);
                for (int k = lane; k < nl; k += 64)
                    loop[k] = k == 0 ? (unsigned short)xi : ring[ci + k];
                CAPE_POLY_SYNC();
                const bool keepLoop = fabs(ring_area_signed(L.pts, loop, nl)) > fabs(ring_area_signed(L.pts, outer, no));
                const unsigned short* kept = keepLoop ? loop : outer;
                const int nk = keepLoop ? nl : no;
                const bool revk = nk >= 3 && ring_area_signed(L.pts, kept, nk) > 0; // clockwise (correct_boost_polygon.hpp:358-369)
                for (int k = lane; k < nk; k += 64)
                    ring[k] = kept[(revk && k > 0) ? nk - k : k];
                rn = nk;
                CAPE_POLY_SYNC();
                if (cut == 7)
                    cutting = false; // still crossing after eight cuts
            }
            if (haveRing)
            {
                haveRing = ring_is_simple(L.pts, ring, rn, lane);
                if (haveRing)
                    flags |= CAPE_POLY_DISSOLVED;
            }
        }
        CAPE_PTICK(2); // simple-ring test of the oriented hull
    }
    if (!haveRing)
    {
        // ---- compute_convex_hull (monotone chain over the sorted, distinct points), clockwise from the leftmost point.
        //      Sequential by nature; every lane runs it on the same values, lane 0 writes.
        flags |= CAPE_POLY_CONVEX_FALLBACK;
        // the walk needed the points in the reference's order; the chain needs them sorted by (x, y) and distinct
        int n = nPts;
        sort_points(L, n, lane);
        {
            int nd = 0;
            for (int base = 0; base < n; base += 64)
            {
                const int i = base + lane;
                double2 q = make_double2(0, 0);
                bool keepIt = false;
                if (i < n)
                {
                    q = L.pts[i];
                    keepIt = true;
                    if (i > 0)
                    {
                        const double2 prev = L.pts[i - 1];
                        keepIt = !(prev.x == q.x && prev.y == q.y);
                    }
                }
                const unsigned long long kb = __ballot(keepIt);
                CAPE_POLY_SYNC(); // every lane has read its point (and its left neighbour) before the chunk is compacted
                if (keepIt)
                    L.pts[nd + __popcll(kb & ((1ull << lane) - 1ull))] = q;
                nd += __popcll(kb);
                CAPE_POLY_SYNC();
            }
            n = nd;
        }
        if (n < 3)
        {
            if (lane < n)
                ring[lane] = (unsigned short)lane;
            rn = n;
        }
        else
        {
            unsigned short* hstk = L.hull; // 2 n entries at most: hull + ring areas are contiguous
            int kx = 0;
            for (int i = 0; i < n; ++i)
            {
                while (kx >= 2 && pcross2(L.pts[hstk[kx - 2]], L.pts[hstk[kx - 1]], L.pts[i]) <= 0)
                    kx--;
                if (lane == 0)
                    hstk[kx] = (unsigned short)i;
                kx++;
                CAPE_POLY_SYNC();
            }
            for (int i = n - 1, t = kx + 1; i > 0; --i)
            {
                while (kx >= t && pcross2(L.pts[hstk[kx - 2]], L.pts[hstk[kx - 1]], L.pts[i - 1]) <= 0)
                    kx--;
                if (lane == 0)
                    hstk[kx] = (unsigned short)(i - 1);
                kx++;
                CAPE_POLY_SYNC();
            }
            const int hn = kx - 1;
            // clockwise, starting at the leftmost point like the chain (host: reverse(h.begin() + 1, h.end())); through
            // registers: hull and ring share LDS when the chain ran past kPolyMaxPoints entries
            unsigned short tmp[kPolyPerLane];
#pragma unroll
            for (int j = 0; j < kPolyPerLane; ++j)
            {
                const int i = lane + 64 * j;
                tmp[j] = i < hn ? hstk[i == 0 ? 0 : hn - i] : (unsigned short)0;
            }
            CAPE_POLY_SYNC();
#pragma unroll
            for (int j = 0; j < kPolyPerLane; ++j)
            {
                const int i = lane + 64 * j;
                if (i < hn)
                    ring[i] = tmp[j];
            }
            rn = hn;
        }
        CAPE_POLY_SYNC();
    }
    bool knownSimple = haveRing; // the oriented (or cut) hull passed ring_is_simple above; a convex fallback ring was not tested
    CAPE_PTICK(3); // orientation, validity of the oriented ring, convex fallback
    CAPE_PCOUNT(9, rn); // vertices before simplification
    CAPE_PCOUNT(10, nPts); // points
    double area = rn >= 3 ? fabs(ring_area_signed(L.pts, ring, rn)) : 0.0;
    // ---- simplify (polygon.cpp:578-601): Douglas-Peucker on the closed ring, threshold max(area / 1e5, 10); kept if the
    //      result is a simple ring whose area stays above 75 %
    if (rn >= 4)
    {
        const double eps = pmax(area / 1e5, 10.0);
        unsigned short* closed = L.hull; // rn + 1 entries
        for (int i = lane; i <= rn; i += 64)
        {
            closed[i] = ring[i < rn ? i : 0];
            L.keep[i] = (i == 0 || i == rn) ? 1 : 0;
        }
        CAPE_POLY_SYNC();
        int sp = 0;
        if (lane == 0)
            L.stack[0] = (unsigned)rn; // (a << 16) | b with a = 0
        sp = 1;
        CAPE_POLY_SYNC();
        while (sp > 0)
        {
            const unsigned ab = L.stack[--sp];
            const int a = (int)(ab >> 16), b = (int)(ab & 0xFFFFu);
            if (b <= a + 1)
                continue;
            const double2 pa = L.pts[closed[a]], pb = L.pts[closed[b]];
            unsigned long long best = 0ull; // squared distance bits (>= +0); the first index wins a tie, like the host's strict >
            int bestI = 0x7FFFFFFF;
            for (int i = a + 1 + lane; i < b; i += 64)
            {
                const double d = segment_distance2(L.pts[closed[i]], pa, pb);
                const unsigned long long db = (unsigned long long)__double_as_longlong(d);
                if (bestI == 0x7FFFFFFF || db > best)
                {
                    best = db;
                    bestI = i;
                }
            }
            // wave maximum of the distance (on DPP: a __shfl_xor butterfly is six trips through the LDS crossbar), smallest index
            // among the equal ones
            const unsigned long long mx = ~wave_min_u64(~best);
            const unsigned mineKey = (bestI != 0x7FFFFFFF && best == mx) ? (unsigned)(0x7FFFFFFF - bestI) : 0u;
            const int idx = 0x7FFFFFFF - (int)wave_max_u32(mineKey);
            const double dmax = __longlong_as_double((long long)mx);
            if (dmax > eps * eps)
            {
                if (lane == 0)
                {
                    L.keep[idx] = 1;
                    L.stack[sp] = ((unsigned)a << 16) | (unsigned)idx;
                    L.stack[sp + 1] = ((unsigned)idx << 16) | (unsigned)b;
                }
                sp += 2;
                CAPE_POLY_SYNC();
            }
        }
        // candidate ring = the kept vertices of closed[0, rn) ; built in the stack area (as u16), then tested
        unsigned short* cand = reinterpret_cast<unsigned short*>(L.stack);
        int cn = 0;
        for (int base = 0; base < rn; base += 64)
        {
            const int i = base + lane;
            const bool k = i < rn && L.keep[i];
            const unsigned long long kb = __ballot(k);
            if (k)
                cand[cn + __popcll(kb & ((1ull << lane) - 1ull))] = closed[i];
            cn += __popcll(kb);
        }
        CAPE_POLY_SYNC();
        if (cn >= 3 && ring_is_simple(L.pts, cand, cn, lane))
        {
            const double newArea = fabs(ring_area_signed(L.pts, cand, cn));
            if (newArea > area * 0.75)
            {
                for (int i = lane; i < cn; i += 64)
                    ring[i] = cand[i];
                rn = cn;
                area = newArea;
                flags |= CAPE_POLY_SIMPLIFIED;
                knownSimple = true;
                CAPE_POLY_SYNC();
            }
        }
    }
    CAPE_PTICK(4); // area + simplify
    // ---- what Primitive_Detection keeps: a valid polygon of at least three vertices (primitive_detection.cpp:623-631)
    // (a ring that came through the hull's own test, or through the simplification's, is not tested a third time: the verdict
    //  is a pure function of the ring)
    if (rn >= 3 && (knownSimple || ring_is_simple(L.pts, ring, rn, lane)))
        flags |= CAPE_POLY_VALID;
    for (int i = lane; i < rn; i += 64)
        vout[i] = L.pts[ring[i]];
    CAPE_PTICK(5); // final validity, vertex stores
    CAPE_PCOUNT(11, 1); // planes
    write_record(p, c, area, rn, flags, lane);
    CAPE_POLY_SYNC();
}

template <int CAP> __device__ inline PolyLds carve_lds(unsigned char* smem)
{
    PolyLds L;
    L.pts = reinterpret_cast<double2*>(smem);
    L.hull = reinterpret_cast<unsigned short*>(L.pts + CAP + kPolyCutPoints);
    L.ring = L.hull + CAP + 2 + kPolyCutPoints;
    L.stack = reinterpret_cast<unsigned int*>(L.ring + CAP + 2 + kPolyCutPoints);
    L.used = reinterpret_cast<unsigned char*>(L.stack + CAP);
    L.keep = L.used + CAP;
    return L;
}

__device__ __constant__ const int kLadderK[8] = {3, 3, 5, 7, 11, 13, 17, 21};

// ---- the 1 024-point instance (planes of 257 .. 1 024 candidates: only the 64 x 48 cell grid shows them): one wave per plane walks
//      the whole ladder, rung after rung.  A fixed grid strides over list 2.
__global__ __launch_bounds__(64 * kPolyWavesPerGroup, 4) void cape_polygon_large_kernel(PolygonParams p, int nFrames, int ldsPerWave)
{
    constexpr int CAP = kPolyMaxPoints;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_all[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t* list = p.lists + (size_t)kPolyFull * p.listStride;
    const unsigned nEntries = list[0];
    const PolyLds L = carve_lds<CAP>(smem_all + (size_t)wave * ldsPerWave);
    for (;;)
    {
        unsigned entryNo = 0;
        if (lane == 0)
            entryNo = atomicAdd(&list[2], 1u);
        entryNo = (unsigned)__builtin_amdgcn_readfirstlane((int)entryNo);
        if (entryNo >= nEntries)
            break;
        const unsigned entry = list[kPolyListHeader + entryNo];
        const PlaneCtx c = plane_context(p, (int)(entry >> 8), (int)(entry & 255u));
        if (c.flags)
        {
            write_record(p, c, 0.0, 0, c.flags, lane);
            continue;
        }
        project_points(p, c, L, lane);
        const int n = c.nPts, first = find_min_y_point(L.pts, n);
        int hs = 0;
        bool haveRing = false;
        for (int a = 0; a < 8 && !haveRing; ++a)
        {
            const int k = kLadderK[a];
            if (a > 0 && k > n)
                break; // the next k exceeds the point count (concave_fitting.cpp:86-87)
            if (a == 1)
                continue; // (the second rung repeats the first: a run is a pure function of the points and k)
            haveRing = concave_hull<CAP>(L, n, first, k, lane, hs);
        }
        finish_polygon<CAP>(p, c, L, haveRing, hs, lane);
    }
}

// ---- planes of up to 256 candidates (every plane of the 640 x 480 streams): TASKS, one wavefront each ------------------------
// A task is one run of the walk for one (plane, rung).  A plane enters through the static list (the output planes of the batch,
// the big ones first) as its rung 0; a plane whose rungs all failed so far SPAWNS its next rungs into the dynamic queue, which
// every idle wave serves first.  The rungs of a plane run side by side on whatever waves are free, exactly as if they had run
// one after the other (a run is a pure function of the points and k; the lowest rung with a hull wins, concave_fitting.cpp:78-88):
//   * per plane a STATE word (done mask | hull mask << 8 | finalised << 16), updated with one atomicOr per finished task;
//   * a rung with a hull that cannot know yet whether a lower rung will have one PARKS its hull (point indices) in global
//     memory; the wave whose atomicOr makes the verdict definitive -- the lowest hull's lower rungs are all done -- builds the
//     polygon from it (every wave of a plane holds the same projected points);
//   * the wave whose atomicOr completes a stage without any hull spawns the next stage, or takes the convex fallback when the
//     ladder is exhausted.
// How many rungs start together depends on the plane's size, because a kernel lasts as long as its longest CHAIN of walks:
// below 100 candidates rung 0, then k = 5, 7, 11, then 13, 17, 21 (three short walks in a row at worst); from 100 on rung 0,
// then all six; from 130 on all seven at once (one long walk).  Rounds 1-3 ran a first-rung kernel and then a ladder kernel
// of three-wave workgroups: the second lasted as long as its slowest plane (0.93 of 1.0 ms) while the device idled.
constexpr uint32_t kStDoneShift = 0, kStHullShift = 8, kStFinal = 1u << 16, kStSpawnShift = 17;
constexpr uint32_t kTaskEmpty = 0xFFFFFFFFu, kTaskQuit = 0xFFFFFFFEu;
constexpr int kParkRungs = 6;       // rungs 2 .. 7 park; rung 0 never waits for a lower one
#ifndef CAPE_POLY_MID
#define CAPE_POLY_MID 100
#endif
#ifndef CAPE_POLY_ALL
#define CAPE_POLY_ALL 256 // (= never: with the chains prioritised the seven-at-once start of the biggest planes only costs throughput)
#endif
constexpr int kPolyMidPoints = CAPE_POLY_MID; // from here on the six remaining rungs start together
constexpr int kPolyAllPoints = CAPE_POLY_ALL; // from here on all seven rungs start together

__device__ __forceinline__ uint32_t rungs_that_exist(int n)
{
    uint32_t m = 1u; // rung 0; rung 1 repeats it
#pragma unroll
    for (int a = 2; a < 8; ++a)
        if (kLadderK[a] <= n)
            m |= 1u << a;
    return m;
}
// the rungs to start next for a plane of n candidates of which `spawned` have been started and none has a hull: all that remain
// when the plane is big or the device is running out of planes to start (its waves would idle while a chain of short walks
// dribbles on), else the next three
__device__ __forceinline__ uint32_t next_rungs(int n, uint32_t exist, uint32_t spawned, bool deviceIdle)
{
    uint32_t remaining = exist & ~spawned;
    if (n >= kPolyMidPoints || deviceIdle)
        return remaining;
    uint32_t out = 0;
    for (int k = 0; k < 3 && remaining; ++k)
    {
        const uint32_t low = remaining & (0u - remaining);
        out |= low;
        remaining &= ~low;
    }
    return out;
}

// a counter as the atomics see it
__device__ __forceinline__ uint32_t fresh_u32(uint32_t* a) { return atomicAdd(a, 0u); }
// polling loads: device scope (served by the L2 the atomics go to; the default system scope goes out to memory every time)
__device__ __forceinline__ uint32_t load_u32(const uint32_t* a) { return __hip_atomic_load(a, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// a plane has its polygon; the wave that finishes the LAST one sends every waiting wave home: one kTaskQuit per wave of the grid
// behind the last task (a wave holds at most one ticket at a time, so that many tickets can be out or still be taken)
__device__ inline void plane_finished(uint32_t* statList, uint32_t* dynList, unsigned listCapacity, unsigned totalWaves, int lane)
{
    unsigned left = 1;
    if (lane == 0)
        left = atomicSub(&statList[3], 1u) - 1u;
    left = (unsigned)__builtin_amdgcn_readfirstlane((int)left);
    if (left != 0)
        return;
    unsigned at = 0;
    if (lane == 0)
        at = atomicAdd(&dynList[0], totalWaves);
    at = (unsigned)__builtin_amdgcn_readfirstlane((int)at);
    for (unsigned i = lane; i < totalWaves; i += 64)
        if (at + i < listCapacity)
            __hip_atomic_store(&dynList[kPolyListHeader + at + i], kTaskQuit, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

#ifndef CAPE_POLY_OCC
#define CAPE_POLY_OCC 4
#endif
__global__ __launch_bounds__(64 * kPolyWavesPerGroup, CAPE_POLY_OCC) void cape_polygon_task_kernel(PolygonParams p, int nFrames, int ldsPerWave, unsigned queueLen)
{
    constexpr int CAP = kPolySmallPoints;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_all[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t* statList = p.lists + (size_t)kPolyFirstRung * p.listStride; // [0] front count [1] back count [2] next [3] planes not finished
    uint32_t* dynList = p.queue;                                          // [0] tail [1] head
    const unsigned nStatic = statList[0];
    // the slots launch_polygons marked "not written yet" for this call: the same length bounds the tickets, the spawned tasks
    // and the quit marks (a bound taken from the scratch's capacity let a ticket read a slot of an earlier call)
    const unsigned listCapacity = queueLen;
    const PolyLds L = carve_lds<CAP>(smem_all + (size_t)wave * ldsPerWave);
    // One wave in four serves the spawned rungs from the start (they sit on some plane's critical chain of walks and must not
    // queue behind the planes of the batch); the others take planes of the batch and become servers when those are handed out.
#ifndef CAPE_POLY_SERVER_EVERY
#define CAPE_POLY_SERVER_EVERY 1
#endif
    bool server = CAPE_POLY_SERVER_EVERY > 0 && wave == kPolyWavesPerGroup - 1 && blockIdx.x % (CAPE_POLY_SERVER_EVERY > 0 ? CAPE_POLY_SERVER_EVERY : 1) == 0; // (lane 0's copy counts)
    const unsigned totalWaves = gridDim.x * kPolyWavesPerGroup;
    if (statList[3] == 0u)
        return; // no plane in the batch: nobody would ever send the waiting waves home
#ifdef CAPE_POLY_PROFILE
    // (wall_clock64: the 100 MHz counter the whole device shares -- s_memtime is per XCD, with bases 1e12 ticks apart)
    // frame 0's spare slots: 6 kernel start (min), 7 static list handed out (min), 23 last polygon (max), 31 last wave leaves (max), 28 busy ticks (sum)
    unsigned long long* tl = p.prof;
    if (lane == 0)
        atomicMin(&tl[6], (unsigned long long)wall_clock64());
#endif
    for (;;)
    {
        // ---- next task: a spawned rung first (it sits on some plane's critical chain), else the next plane of the batch
        unsigned task = kTaskEmpty;
        bool isStatic = false;
#ifdef CAPE_POLY_PROFILE
        const unsigned long long acq0 = __builtin_amdgcn_s_memtime();
#endif
        if (lane == 0)
        {
            if (!server)
            {
                const unsigned e = atomicAdd(&statList[2], 1u);
                if (e < nStatic)
                {
                    const unsigned entry = statList[kPolyListHeader + e];
                    task = entry << 3; // rung 0
                    isStatic = true;
                }
                else
                {
                    server = true; // the planes of the batch are handed out: from now on this wave serves spawned rungs
#ifdef CAPE_POLY_PROFILE
                    atomicMin(&tl[7], (unsigned long long)wall_clock64());
#endif
                }
            }
            if (task == kTaskEmpty)
            {
                // a TICKET: the next slot of the queue is this wave's, whoever fills it and whenever.  No compare-and-swap, no shared
                // word to poll -- every waiting wave watches its own slot (read with a fetch-add of 0: a plain load may be served by
                // this XCD's L2, which is not coherent with the one the writer's store went through).  The wave that finishes the
                // batch's last plane fills every ticket still out with kTaskQuit.
                const unsigned ticket = atomicAdd(&dynList[1], 1u);
                if (ticket < listCapacity)
                {
                    uint32_t t;
                    while ((t = fresh_u32(&dynList[kPolyListHeader + ticket])) == kTaskEmpty)
                        __builtin_amdgcn_s_sleep(64);
                    task = t;
                }
                else
                    task = kTaskQuit; // (only a -DCAPE_POLY_QUEUE_LEN test build gets here: the queue holds a slot per possible
                                      // task and per wave.  Leaving is safe: a wave walks the rungs it could not enqueue itself)
            }
        }
        task = (unsigned)__builtin_amdgcn_readfirstlane((int)task);
        isStatic = __builtin_amdgcn_readfirstlane((int)isStatic) != 0;
#ifdef CAPE_POLY_PROFILE
        if (lane == 0 && p.prof)
        {
            // frame 0's slots 24 .. 27: ticks spent getting a task (static / spawned), tasks of either kind
            atomicAdd(&p.prof[isStatic ? 24 : 25], __builtin_amdgcn_s_memtime() - acq0);
            atomicAdd(&p.prof[isStatic ? 26 : 27], 1ull);
        }
#endif
        if (task == kTaskQuit)
        {
#ifdef CAPE_POLY_PROFILE
            if (lane == 0)
                atomicMax(&tl[31], (unsigned long long)wall_clock64());
#endif
            break;
        }
#ifdef CAPE_POLY_PROFILE
        const unsigned long long busy0 = __builtin_amdgcn_s_memtime();
        struct BusyScope
        {
            unsigned long long* tl;
            unsigned long long t0;
            int lane;
            __device__ ~BusyScope()
            {
                if (lane == 0)
                {
                    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
                    atomicAdd(&tl[28], t1 - t0);
                    atomicMax(&tl[23], (unsigned long long)wall_clock64());
                }
            }
        } busyScope {tl, busy0, lane};
#endif
        const int rung = (int)(task & 7u);
#ifndef CAPE_POLY_PRIO
#define CAPE_POLY_PRIO 1
#endif
        // A spawned rung sits on its plane's critical CHAIN of walks (the pass used to end with the last of a few three-walk chains:
        // rung 0 all around a 100-point outline at ~3.6 us per step on a full SIMD, then k = 5 .. 11, then 13 .. 21), a plane of the
        // batch does not: the chains get the SIMD's issue slots first (s_setprio).  k = 5 wins 85 % of the climbs, so it goes first,
        // k = 7 next; the higher rungs run behind them and give up as soon as a lower one has its hull.  Room stream 1.43 -> 1.17 ms
        // per 4 096 frames, TUM-like 1.00 -> 0.89 (with the all-seven-at-once start dropped: profiles/r04_polygon_tasks.txt).
        if (CAPE_POLY_PRIO)
        {
            if (isStatic)
                __builtin_amdgcn_s_setprio(0);
            else if (rung <= 2)
                __builtin_amdgcn_s_setprio(3);
            else if (rung == 3)
                __builtin_amdgcn_s_setprio(2);
            else
                __builtin_amdgcn_s_setprio(1);
        }
        const PlaneCtx c = plane_context(p, (int)(task >> 11), (int)((task >> 3) & 255u));
#ifdef CAPE_POLY_PROFILE
        const int frame = c.frame;
        unsigned long long _pt = __builtin_amdgcn_s_memtime();
#endif
        if (c.flags)
        {
            // nothing to walk (the list kernel sends these through as rung 0 only)
            write_record(p, c, 0.0, 0, c.flags, lane);
            plane_finished(statList, dynList, listCapacity, totalWaves, lane);
            continue;
        }
        const int n = c.nPts;
        uint32_t* state = p.state + (size_t)c.frame * CAPE_MAX_PLANES + c.seg;
        if (!isStatic)
        {
            // a spawned rung that can no longer win (the plane has its polygon, or a lower rung its hull) is not walked at all
            uint32_t st = 0;
            if (lane == 0)
                st = fresh_u32(state);
            st = (uint32_t)__builtin_amdgcn_readfirstlane((int)st);
            if ((st & kStFinal) || (((st >> kStHullShift) & 0xFFu) & ((1u << rung) - 1u)))
                continue;
        }
        const uint32_t exist = rungs_that_exist(n);
        auto spawn = [&](uint32_t rungs) {
            // lane 0: append one task per rung.  The queue holds a slot for every rung every plane of the call can spawn (six per
            // plane: each rung is spawned at most once, the state word's spawned mask) and one quit mark per wave, so it cannot
            // fill up; a test build with a tiny queue (-DCAPE_POLY_QUEUE_LEN=16) shows what happens if it did: EVERY reserved
            // slot below the capacity is written (a ticket may be watching it), the rungs beyond come back to the caller, which
            // walks them itself, and tickets beyond the capacity leave.  (Through round 4 a rung refused below the capacity left
            // its slot unwritten behind the moved tail: the ticket that landed on it would have spun for ever)
            uint32_t left = rungs;
            if (lane == 0 && rungs)
            {
                const unsigned cnt = (unsigned)__popc(rungs);
                const unsigned at = atomicAdd(&dynList[0], cnt);
                unsigned k = 0;
                for (int a = 0; a < 8; ++a)
                    if ((rungs >> a) & 1u)
                    {
                        if (at + k < listCapacity)
                        {
                            __hip_atomic_store(&dynList[kPolyListHeader + at + k], ((unsigned)c.frame << 11) | ((unsigned)c.seg << 3) | (unsigned)a, __ATOMIC_RELAXED,
                                               __HIP_MEMORY_SCOPE_AGENT);
                            left &= ~(1u << a);
                        }
                        ++k;
                    }
            }
            return (uint32_t)__builtin_amdgcn_readfirstlane((int)left);
        };
        project_points(p, c, L, lane);
        const int first = find_min_y_point(L.pts, n);
        CAPE_PTICK(0); // projection, start point
        uint32_t mine = 1u << rung; // the rungs this wave walks: its task, plus whatever a full queue hands back
        if (isStatic)
        {
            // the plane's first rungs: rung 0, and with it all the others when the plane is big (one long walk instead of a chain)
            const uint32_t firstRungs = n >= kPolyAllPoints ? exist : 1u;
            if (lane == 0)
                atomicOr(state, firstRungs << kStSpawnShift);
            mine |= spawn(firstRungs & ~1u);
        }
        bool finished = false;
        while (mine && !finished)
        {
            const int r = __ffs((int)mine) - 1;
            mine &= ~(1u << r);
            int hs = 0;
            CAPE_PCOUNT(8, 1); // hull attempts
            // (rung 0 cannot be overruled, so it never looks at the state word while it walks)
            const bool ok = concave_hull<CAP>(L, n, first, kLadderK[r], lane, hs, r > 0 ? state : nullptr, r);
            CAPE_PTICK(1); // hull walks (incl. the all-points-inside check)
            uint32_t old = 0;
            if (lane == 0)
            {
                if (ok && r > 0)
                {
                    // park the hull: [0] its length, then the point indices
                    unsigned short* park = p.park + ((size_t)c.frame * kParkRungs + (r - 2)) * p.parkStride + p.records[c.frame].segments[c.seg].boundary_offset + 2 * c.seg; // (offsets grow with the segment index: n + 2 entries fit)
                    park[0] = (unsigned short)hs;
                    for (int i = 0; i < hs; ++i)
                        park[1 + i] = L.hull[i];
                    __threadfence();
                }
                old = atomicOr(state, (1u << (kStDoneShift + r)) | (ok ? (1u << (kStHullShift + r)) : 0u));
            }
            old = (uint32_t)__builtin_amdgcn_readfirstlane((int)old);
            const uint32_t now = old | (1u << r) | (ok ? (1u << (kStHullShift + r)) : 0u);
            const uint32_t done = now & 0xFFu, hulls = (now >> kStHullShift) & 0xFFu;
            int winner = -2; // -2: nothing to decide yet, -1: convex fallback, >= 0: this rung's hull
            if (hulls)
            {
                const int w = __ffs((int)hulls) - 1;
                if ((done & exist & ((1u << w) - 1u)) == (exist & ((1u << w) - 1u)))
                    winner = w;
            }
            else
            {
                // every rung started so far is done and none has a hull: the wave whose bit completed the set moves the plane on
                const uint32_t spawned = (now >> kStSpawnShift) & 0xFFu;
                if ((done & spawned) == spawned && ((old & 0xFFu) & spawned) != spawned)
                {
                    const bool deviceIdle = load_u32(&statList[2]) >= nStatic; // (a stale value only delays the switch)
                    const uint32_t next = next_rungs(n, exist, spawned, deviceIdle);
                    if (!next)
                        winner = -1; // the ladder is exhausted
                    else
                    {
                        if (lane == 0)
                            atomicOr(state, next << kStSpawnShift);
                        mine |= spawn(next);
                    }
                }
            }
            if (winner == 0 && r != 0)
                winner = -2; // rung 0 never parks (no lower rung can overrule it): the wave that walked it builds the polygon itself
            if (winner != -2)
            {
                uint32_t before = kStFinal;
                if (lane == 0)
                    before = atomicOr(state, kStFinal);
                before = (uint32_t)__builtin_amdgcn_readfirstlane((int)before);
                if (!(before & kStFinal))
                {
                    if (winner >= 0 && !(ok && winner == r))
                    {
                        // somebody else's hull: out of the parking area (its writer fenced before it set the bit this wave saw)
                        __threadfence();
                        const unsigned short* park = p.park + ((size_t)c.frame * kParkRungs + (winner - 2)) * p.parkStride + p.records[c.frame].segments[c.seg].boundary_offset + 2 * c.seg; // (offsets grow with the segment index: n + 2 entries fit)
                        hs = (int)__hip_atomic_load(&park[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        for (int i = lane; i < hs; i += 64)
                            L.hull[i] = __hip_atomic_load(&park[1 + i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        CAPE_POLY_SYNC();
                    }
                    CAPE_PCOUNT(12 + (winner < 0 ? 0 : winner), 1);
                    finish_polygon<CAP>(p, c, L, winner >= 0, hs, lane);
#ifdef CAPE_POLY_DEBUG_RUNG
                    if (lane == 0)
                        p.polygons[(size_t)c.frame * CAPE_MAX_PLANES + c.seg].flags |= ((uint32_t)(winner + 1) << 8) | ((uint32_t)r << 12) | (now << 16);
#endif
#ifdef CAPE_POLY_PROFILE
                    if (lane == 0 && p.prof)
                    {
                        // per frame: when its last polygon was finished (slot 22), that plane's candidates (21) and winning rung + 1 (20)
                        const unsigned long long tEnd = (unsigned long long)wall_clock64();
                        if (atomicMax(&p.prof[(size_t)c.frame * kProfileSlots + 22], tEnd) < tEnd)
                        {
                            p.prof[(size_t)c.frame * kProfileSlots + 21] = (unsigned long long)n;
                            p.prof[(size_t)c.frame * kProfileSlots + 20] = (unsigned long long)(winner + 1);
                        }
                    }
#endif
                    plane_finished(statList, dynList, listCapacity, totalWaves, lane);
                }
                finished = true;
            }
        }
    }
}

// one wavefront per frame, lane j <- segment j: the static list of the task kernel (planes of up to 256 candidates), the list of
// the 1 024-point instance, and the count of planes to finish.  The static list is ordered by size, the biggest planes first
// (sixteen buckets of sixteen candidates): a kernel of independent walks ends with its last-started chains, and those should be
// the short ones -- a plane of 30 candidates taken last costs three walks of ten steps, one of 150 three of fifty.  Two passes
// of the same kernel: count per bucket, then place (the bucket's offset is the sum of the counts before it).
constexpr int kListFrames = 16; // frames (waves) of a list workgroup: ONE atomic per counter and workgroup -- a counter that every
                                // frame's wave bumps on its own serialises 4 096 atomics on one address (74 us of the pass)
constexpr int kSizeBuckets = 16;
// the headers of the two work lists and of the queue, zeroed by ONE small launch (three memset nodes through round 5)
__global__ void cape_polygon_reset_kernel(PolygonParams p)
{
    const int t = threadIdx.x;
    if (t < kPolyListHeader)
    {
        p.lists[t] = 0u;
        p.lists[(size_t)p.listStride + t] = 0u;
        p.queue[t] = 0u;
    }
}

// `queueLen`: slots of the task queue this call may use; pass 0 marks them "not written yet" and clears the polygon rows -- every
// virtual frame's wave its share, instead of two memsets over 7 MB + 25 MB per 4 096 frames in front of the pass
__global__ __launch_bounds__(64 * kListFrames) void cape_polygon_list_kernel(PolygonParams p, int nFrames, int pass, unsigned queueLen)
{
    __shared__ unsigned s_cnt[kSizeBuckets + 1], s_base[kSizeBuckets + 1];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // virtual frames: the batch's records [0, nFrames), then the spill records in use (record index poolBase + k)
    const int vframe = blockIdx.x * kListFrames + wave;
    int frame = vframe;
    bool live = vframe < nFrames;
    if (!live && p.poolUsed && vframe - nFrames < p.poolCapacity && (unsigned)(vframe - nFrames) < *p.poolUsed)
    {
        frame = p.poolBase + (vframe - nFrames);
        live = true;
    }
    if (pass == 0)
    {
        // this wave's share of the queue marks (every virtual frame takes part, in use or not) ...
        const unsigned vTotal = (unsigned)(nFrames + p.poolCapacity);
        const unsigned share = (queueLen + vTotal - 1u) / vTotal;
        const unsigned q0 = (unsigned)vframe * share;
        if ((unsigned)vframe < vTotal)
            for (unsigned q = q0 + lane; q < q0 + share && q < queueLen; q += 64)
                p.queue[kPolyListHeader + q] = 0xFFFFFFFFu;
        // ... and the polygon rows of its record (segments that are no output plane keep the empty record)
        if (live)
        {
            uint32_t* w = reinterpret_cast<uint32_t*>(p.polygons + (size_t)frame * CAPE_MAX_PLANES);
            for (int k = lane; k < (int)(CAPE_MAX_PLANES * sizeof(cape_polygon) / 4); k += 64)
                w[k] = 0u;
        }
    }
    if (threadIdx.x <= kSizeBuckets)
        s_cnt[threadIdx.x] = 0u;
    __syncthreads();
    bool isOut = false;
    int nPts = 0;
    if (live)
    {
        const cape_frame_record& rec = p.records[frame];
        isOut = lane < rec.header.n_plane_segments && rec.segments[lane].is_output != 0;
        nPts = isOut ? (int)rec.segments[lane].boundary_count : 0;
        if (pass == 0)
            p.state[(size_t)frame * CAPE_MAX_PLANES + lane] = 0u;
    }
    // (segments that are no output plane keep the empty record launch_polygons' memset left)
    // 257 .. 1 024 candidates: the large instance; everything else (incl. what it will only flag: too few / too many points)
    // goes to the task kernel
    const bool large = isOut && nPts > kPolySmallPoints && nPts <= kPolyMaxPoints;
    const bool small = isOut && !large;
    // bucket 0 holds the biggest planes (what the task kernel will only flag -- more than 1 024 candidates -- costs nothing: last)
    const int sized = nPts > kPolySmallPoints ? 0 : nPts;
    const int bucket = large ? kSizeBuckets : (kSizeBuckets - 1 - (sized * kSizeBuckets) / (kPolySmallPoints + 1));
    unsigned myPos = 0;
    if (small || large)
        myPos = atomicAdd(&s_cnt[bucket], 1u);
    __syncthreads();
    uint32_t* listS = p.lists + (size_t)kPolyFirstRung * p.listStride;
    uint32_t* listL = p.lists + (size_t)kPolyFull * p.listStride;
    uint32_t* counts = listS + 4;                 // [16] planes per bucket (pass 0)
    uint32_t* cursors = listS + 4 + kSizeBuckets; // [16] placed so far (pass 1)
    if (threadIdx.x <= kSizeBuckets && s_cnt[threadIdx.x])
    {
        const unsigned c = s_cnt[threadIdx.x];
        if (pass == 0)
        {
            if (threadIdx.x < kSizeBuckets)
            {
                atomicAdd(&counts[threadIdx.x], c);
                atomicAdd(&listS[0], c); // planes on the static list
                atomicAdd(&listS[3], c); // planes the task kernel has to finish
            }
        }
        else
            s_base[threadIdx.x] = threadIdx.x < kSizeBuckets ? atomicAdd(&cursors[threadIdx.x], c) : atomicAdd(&listL[0], c);
    }
    if (pass == 0)
        return;
    __syncthreads();
    const unsigned entry = ((unsigned)frame << 8) | (unsigned)lane;
    if (small)
    {
        unsigned before = 0; // planes in bigger buckets
        for (int k = 0; k < bucket; ++k)
            before += counts[k];
        listS[kPolyListHeader + before + s_base[bucket] + myPos] = entry;
    }
    if (large)
        listL[kPolyListHeader + s_base[kSizeBuckets] + myPos] = entry;
}

#ifndef CAPE_POLY_QUIT_SLOTS
#define CAPE_POLY_QUIT_SLOTS 8192
#endif
#ifndef CAPE_POLY_QUEUE_PER_FRAME
#define CAPE_POLY_QUEUE_PER_FRAME (6 * CAPE_MAX_PLANES)
#endif
constexpr size_t kPolyQuitSlots = CAPE_POLY_QUIT_SLOTS;          // one kTaskQuit per wave of the task kernel's grid behind the last task
[[maybe_unused]] constexpr size_t kPolyQueuePerFrame = CAPE_POLY_QUEUE_PER_FRAME; // rungs 2 .. 7 of every plane of a frame: each is spawned at most once
#ifdef CAPE_POLY_QUEUE_LEN
// test build: a queue of CAPE_POLY_QUEUE_LEN slots whatever the batch, so that spawned rungs overflow it (the grid keeps its size)
size_t polygon_queue_slots(size_t) { return CAPE_POLY_QUEUE_LEN; }
#else
size_t polygon_queue_slots(size_t frames) { return frames * kPolyQueuePerFrame + kPolyQuitSlots; }
#endif

size_t polygon_lds_bytes(int cap)
{
    size_t b = ((size_t)cap + kPolyCutPoints) * 16;            // pts (+ the crossing points of a dissolved hull)
    b += 2 * ((size_t)cap + 2 + kPolyCutPoints) * 2;         // hull, ring
    b += (size_t)cap * 4;                   // stack
    b += (size_t)cap + cap + 2;             // used, keep
    return (b + 15) & ~(size_t)15;
}

// scratch of a polygon pass over `frames` frames: three work lists, the state words, the parking area
size_t polygon_scratch_bytes(size_t frames, int boundaryCapacity)
{
    const size_t lists = (2 * (frames * CAPE_MAX_PLANES + kPolyListHeader) + kPolyListHeader + polygon_queue_slots(frames)) * sizeof(uint32_t);
    const size_t state = frames * CAPE_MAX_PLANES * sizeof(uint32_t);
    const size_t park = frames * kParkRungs * ((size_t)boundaryCapacity + 2 * CAPE_MAX_PLANES) * sizeof(unsigned short);
    return lists + state + park + 64;
}
void polygon_bind_scratch(PolygonParams& p, void* base, size_t frames, int boundaryCapacity)
{
    p.lists = static_cast<uint32_t*>(base);
    p.listStride = (uint32_t)(frames * CAPE_MAX_PLANES + kPolyListHeader);
    p.queue = p.lists + 2 * (size_t)p.listStride;
    p.queueCapacity = (uint32_t)polygon_queue_slots(frames);
    p.state = p.queue + kPolyListHeader + p.queueCapacity;
    p.park = reinterpret_cast<unsigned short*>(p.state + frames * CAPE_MAX_PLANES);
    p.parkStride = (uint32_t)(boundaryCapacity + 2 * CAPE_MAX_PLANES);
}

hipError_t launch_polygons(const PolygonParams& p, int nFrames, hipStream_t stream)
{
    // the headers of the two work lists (list m at p.lists + m * listStride) and of the queue: one small launch; the queue's "not
    // written yet" marks and the polygon rows: pass 0 of the list kernel.  ONE length -- what the frames can spawn plus a quit mark
    // per wave -- for the marks and for the task kernel's bounds
    hipLaunchKernelGGL(cape_polygon_reset_kernel, dim3(1), dim3(64), 0, stream, p);
    if (const hipError_t e = hipGetLastError(); e != hipSuccess)
        return e;
    const size_t wanted = polygon_queue_slots((size_t)nFrames + (size_t)p.poolCapacity);
    const size_t queue = wanted < (size_t)p.queueCapacity ? wanted : (size_t)p.queueCapacity;
    for (int pass = 0; pass < 2; ++pass)
    {
        hipLaunchKernelGGL(cape_polygon_list_kernel, dim3((nFrames + p.poolCapacity + kListFrames - 1) / kListFrames), dim3(64 * kListFrames), 0, stream, p, nFrames, pass, (unsigned)queue);
        if (const hipError_t e = hipGetLastError(); e != hipSuccess)
            return e;
    }
    const int ldsSmall = (int)polygon_lds_bytes(kPolySmallPoints), ldsLarge = (int)polygon_lds_bytes(kPolyMaxPoints);
    // persistent grids: as many workgroups as the device holds at once (four waves per SIMD), never more than there can be planes
    const int maxPlanes = (nFrames + p.poolCapacity) * CAPE_MAX_PLANES;
    const int gridSmall = std::min(std::min((maxPlanes + kPolyWavesPerGroup - 1) / kPolyWavesPerGroup, p.computeUnits * CAPE_POLY_OCC), (int)(kPolyQuitSlots / kPolyWavesPerGroup));
    const int gridLarge = std::min((maxPlanes + kPolyWavesPerGroup - 1) / kPolyWavesPerGroup, p.computeUnits);
    hipLaunchKernelGGL(cape_polygon_task_kernel, dim3(gridSmall), dim3(64 * kPolyWavesPerGroup), (size_t)ldsSmall * kPolyWavesPerGroup + 64, stream, p,
                       nFrames, ldsSmall, (unsigned)queue);
    if (const hipError_t e = hipGetLastError(); e != hipSuccess)
        return e;
    if (p.boundaryCapacity > kPolySmallPoints) // a plane cannot hold more boundary points than the frame
        hipLaunchKernelGGL(cape_polygon_large_kernel, dim3(gridLarge), dim3(64 * kPolyWavesPerGroup), (size_t)ldsLarge * kPolyWavesPerGroup + 64, stream, p,
                           nFrames, ldsLarge);
    return hipGetLastError();
}

} // namespace cape
