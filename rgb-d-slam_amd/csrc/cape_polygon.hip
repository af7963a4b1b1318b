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
// The k ladder.  Nearly every plane gets its hull on the first rung (1.13 attempts on average), but a plane that climbs to
// k = 21 spends ~3 ms in one wave, and a kernel lasts as long as its slowest wave (4.2 ms per 4 096 room frames, one plane in
// a thousand).  So the small instance runs the FIRST rung only and defers a plane that fails it (work list 1) to the
// ladder kernel: a workgroup of three waves per such plane, wave w on rung w + 2 (k = 5, 7, 11) all at once, then -- if none has
// a hull -- on rung w + 5 (k = 13, 17, 21); the
// lowest rung that yields a simple hull wins, exactly as if they had run one after the other (a run is a pure function of
// the points and k).
constexpr int kLadderWaves = 3;
constexpr int kPolySortSelect = 5;  // neighbours from which the k-nearest selection sorts the lanes' keys instead of taking k minima
constexpr int kPolyBigPoints = 100; // planes with at least as many boundary candidates are handed out first (longest first)
enum PolyMode
{
    kPolyFirstRung = 0, // small instance: rung 0, defer on failure
    kPolyLadder = 1,    // small instance, six waves per plane: rungs 2 .. 7 in parallel
    kPolyFull = 2       // large instance: the whole ladder in one wave (rare twice over)
};

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
__device__ __forceinline__ int turn_class(double px, double py, double vx, double vy)
{
    const double cr = px * vy - py * vx, dt = px * vx + py * vy;
    if (cr < 0)
        return 1;
    if (cr > 0)
        return 3;
    return dt > 0 ? 0 : 2;
}
// does candidate a turn further clockwise from P than candidate b (strictly)?
__device__ __forceinline__ bool turns_further(int ca, double ax, double ay, int cb, double bx, double by)
{
    if (ca != cb)
        return ca > cb;
    if (ca == 0 || ca == 2)
        return false;
    return (bx * ay - by * ax) < 0; // cross(Vb, Va) < 0: Va lies clockwise of Vb inside the same open half turn
}

// One run of the k-nearest-neighbours walk (host: concave_hull_k = ConcaveHull of concave_fitting.cpp:93-183).  On success the
// hull's point indices are in L.hull[0, hs) exactly as the reference's vector holds them: ending with the start point again
// when the walk came back to it, every point once when it ran out of points first.
// `lowestDone` (ladder kernel only, else null): LDS word holding the lowest rung that already has its hull; a higher rung gives
// up as soon as it sees one below it succeed -- it can no longer win.
template <int CAP> __device__ inline bool concave_hull_k(const PolyLds& L, int n, int first, int k, int lane, int& hsOut, const volatile int* lowestDone = nullptr,
                                                        int myRung = 0)
{
    constexpr int kPolyPerLane = CAP / 64; // points a lane owns in the lane-parallel passes
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
    for (int i = lane; i < n; i += 64)
        L.used[i] = 0;
    CAPE_POLY_SYNC();
    if (lane == 0)
    {
        L.hull[0] = (unsigned short)first;
        L.used[first] = 1;
    }
    CAPE_POLY_SYNC();
    int hs = 1;
    int usedCount = 1; // points hidden from the neighbour search
    int current = first;
    double2 cur = firstPt;
    double Px = 1.0, Py = 0.0; // prevAngle = 0: the +x direction
    int step = 1;
    while ((!points_equal(cur, firstPt) || step == 1) && hs != n)
    {
        if (lowestDone && __builtin_amdgcn_readfirstlane(*lowestDone) < myRung)
            return false;
        if (step == 4)
        {
            if (lane == 0)
                L.used[first] = 0; // the start point is put back into the index once the hull has three edges
            --usedCount;
            CAPE_POLY_SYNC();
        }
        // ---- k nearest visible neighbours of the current point, ascending (squared distance, index)
        // key = the squared distance's bit pattern (>= +0: the bits order like the value) with its ten lowest mantissa bits
        // replaced by the point index: ONE 64-bit wave minimum per neighbour yields the nearest point and breaks ties (and
        // distances within 2^-42 of each other) by index.  The host class sorts by the very same key.
        unsigned long long key[kPolyPerLane];
#pragma unroll
        for (int j = 0; j < kPolyPerLane; ++j)
        {
            const int i = lane + 64 * j;
            key[j] = ~0ull;
            if (64 * j >= n)
                continue; // (uniform: a plane of 150 points uses three of the sixteen slots)
            if (i < n && !L.used[i])
            {
                const double2 q = pts[i];
                const double dx = cur.x - q.x, dy = cur.y - q.y;
                key[j] = ((unsigned long long)__double_as_longlong(dx * dx + dy * dy) & ~1023ull) | (unsigned long long)i;
            }
        }
        const int cnt = n - usedCount;
        if (cnt <= 0)
            return false; // no neighbour left: the reference leaves its candidate loop with `its` still set
        const int kk = k < cnt ? k : cnt;
        // candidate c lives in lane c (kk <= 21 < 64): index, edge vector, turn class
        int myCand = 0, myClass = 0;
        double myVx = 0.0, myVy = 0.0;
        bool selected = false;
        if (kk >= kPolySortSelect)
        {
            // Many neighbours: ONE sort instead of kk minima.  Every lane offers its smallest key; sorted across the wave, lane c
            // holds the c-th smallest of them -- and these are the kk nearest points if no lane hides a second key below the
            // kk-th.  Otherwise the minima below decide.
            unsigned long long head = ~0ull, second = ~0ull;
#pragma unroll
            for (int j = 0; j < kPolyPerLane; ++j)
                if (64 * j < n)
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
            if (__popcll(__ballot(sorted < hidden)) >= kk)
            {
                selected = true;
                if (lane < kk)
                {
                    const int idx = (int)(sorted & 1023ull);
                    const double2 q = pts[idx];
                    myCand = idx;
                    myVx = q.x - cur.x;
                    myVy = q.y - cur.y;
                }
            }
        }
        for (int c = 0; c < kk && !selected; ++c)
        {
            unsigned long long best = ~0ull;
#pragma unroll
            for (int j = 0; j < kPolyPerLane; ++j)
                if (64 * j < n && key[j] < best)
                    best = key[j];
            const unsigned long long m = wave_min_u64(best);
            const int idx = (int)(m & 1023ull);
#pragma unroll
            for (int j = 0; j < kPolyPerLane; ++j)
                if (64 * j < n && key[j] == m)
                    key[j] = ~0ull; // taken (keys are unique: they carry the index)
            if (lane == c)
            {
                const double2 q = pts[idx];
                myCand = idx;
                myVx = q.x - cur.x;
                myVy = q.y - cur.y;
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
            tried |= 1u << b;
            const int cnd = __builtin_amdgcn_readlane(myCand, b);
            const double2 cp = pts[cnd];
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
            }
        }
        if (!found)
            return false;
        const double2 nx = pts[next];
        Px = cur.x - nx.x; // looking back along the new edge
        Py = cur.y - nx.y;
        if (Px == 0 && Py == 0)
            Px = 1.0;
        current = next;
        cur = nx;
        if (lane == 0)
        {
            L.hull[hs] = (unsigned short)current;
            L.used[current] = 1;
        }
        ++hs;
        ++usedCount;
        ++step;
        CAPE_POLY_SYNC();
    }
    // every point that is not a hull vertex must pass PointInPolygon (the start point is a hull vertex whether or not it is
    // hidden from the search at this moment)
    bool outside = false;
    for (int i = lane; i < n; i += 64)
        if (!L.used[i] && i != first && !point_in_hull(pts[i], pts, L.hull, hs))
            outside = true;
    if (__any(outside))
        return false;
    hsOut = hs;
    return true;
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

template <int CAP, int MODE>
__global__ __launch_bounds__(64 * (MODE == kPolyLadder ? kLadderWaves : kPolyWavesPerGroup), MODE == kPolyLadder ? 3 : 4) void cape_polygon_kernel(
        PolygonParams p, int nFrames, int ldsPerWave)
{
    constexpr int kPolyPerLane = CAP / 64;
    constexpr int kWaves = MODE == kPolyLadder ? kLadderWaves : kPolyWavesPerGroup;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_all[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // Work comes as lists of (frame << 8 | segment) written on the device: list 0 = the output planes of up to 256 boundary
    // candidates (cape_polygon_list_kernel), list 1 = planes whose first rung failed (written by the first-rung instance),
    // list 2 = planes of 257 .. 1 024 candidates.  A fixed grid strides over its list -- one WAVE per entry, or one WORKGROUP
    // per entry in the ladder kernel -- so every resident wave has a plane (the first version launched eight waves per frame
    // for ~2.5 planes: two thirds of the resident waves had nothing to do).
    int* s_ok = reinterpret_cast<int*>(smem_all + (size_t)kWaves * ldsPerWave); // ladder kernel: verdict of every rung
    // a list: [0] entries filled from the front (the big planes), [1] entries filled from the back, [2] the next entry to hand
    // out, then the entries.  Waves (workgroups in the ladder kernel) take the next entry when they are done with theirs --
    // big planes first, so that the longest walks start at once instead of at the end of somebody's queue.
    uint32_t* list = p.lists + (size_t)MODE * p.listStride;
    const unsigned nFront = list[0], nEntries = nFront + list[1];
    const unsigned listCapacity = p.listStride - kPolyListHeader;
    unsigned char* smem = smem_all + (size_t)wave * ldsPerWave;
    PolyLds L;
    L.pts = reinterpret_cast<double2*>(smem);
    L.hull = reinterpret_cast<unsigned short*>(L.pts + CAP);
    L.ring = L.hull + CAP + 2;
    L.stack = reinterpret_cast<unsigned int*>(L.ring + CAP + 2);
    L.used = reinterpret_cast<unsigned char*>(L.stack + CAP);
    L.keep = L.used + CAP;
    for (;;)
    {
        unsigned entryNo = 0;
        if (MODE == kPolyLadder)
        {
            if (threadIdx.x == 0)
                s_ok[9] = (int)atomicAdd(&list[2], 1u);
            __syncthreads();
            entryNo = (unsigned)s_ok[9];
        }
        else
        {
            if (lane == 0)
                entryNo = atomicAdd(&list[2], 1u);
            entryNo = (unsigned)__builtin_amdgcn_readfirstlane((int)entryNo);
        }
        if (entryNo >= nEntries)
            break;
        const unsigned entry = list[kPolyListHeader + (entryNo < nFront ? entryNo : listCapacity - 1 - (entryNo - nFront))];
        const int frame = (int)(entry >> 8), seg = (int)(entry & 255u);
        const cape_plane_segment& S = p.records[frame].segments[seg];
        cape_polygon* out = &p.polygons[(size_t)frame * CAPE_MAX_PLANES + seg];
        double2* vout = p.vertices + (size_t)frame * p.boundaryCapacity + S.boundary_offset;
        const int nPts = (int)S.boundary_count;
        uint32_t flags = 0;
#ifdef CAPE_POLY_PROFILE
        unsigned long long _pt = __builtin_amdgcn_s_memtime();
#endif
        // ---- plane frame: get_plane_coordinate_system (polygon.cpp:74-115 with select_correct_transform :50-68)
        const double nx = S.normal[0], ny = S.normal[1], nz = S.normal[2];
        // the polygon's origin is Plane_Segment::get_center() = PlaneCoordinates::get_center() = normal * (-d), the point of the plane
        // closest to the camera (primitive_detection.cpp:622, plane_segment.hpp:90, plane_coordinates.hpp:52) -- not the centroid
        const double cx = p.originInCentroid ? S.centroid[0] : nx * (-S.d), cy = p.originInCentroid ? S.centroid[1] : ny * (-S.d),
                     cz = p.originInCentroid ? S.centroid[2] : nz * (-S.d);
        double xax, xay, xaz, yax, yay, yaz;
        {
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
            xax = ax, xay = ay, xaz = az, yax = bx, yay = by, yaz = bz;
        }
        const double normalNorm = sqrt((nx * nx + ny * ny) + nz * nz);
        int count = 0;
        double area = 0.0;
        if (!(fabs(normalNorm - 1.0) <= 1e-9) || nPts < 3)
            flags |= CAPE_POLY_REJECTED; // the host constructor throws (normal not unit / fewer than 3 points)
        else if (nPts > kPolyMaxPoints)
            flags |= CAPE_POLY_OVERFLOW; // left to the host class
        else
        {
            // ---- projection (polygon.cpp:125-144), in REVERSE order like the reference (:187-192); no sort, no duplicate removal
            //      (concave_fitting.cpp:69: the overload this path binds does not call RemoveDuplicates)
            const double* bnd = p.boundary + ((size_t)frame * p.boundaryCapacity + S.boundary_offset) * 3;
            for (int i = lane; i < nPts; i += 64)
            {
                const double* q = bnd + 3 * (nPts - 1 - i);
                const double dx = q[0] - cx, dy = q[1] - cy, dz = q[2] - cz;
                L.pts[i] = make_double2((xax * dx + xay * dy) + xaz * dz, (yax * dx + yay * dy) + yaz * dz);
            }
            CAPE_POLY_SYNC();
            const int n = nPts;
            const int first = find_min_y_point(L.pts, n);
            CAPE_PTICK(0); // projection, start point
            // ---- concave hull on the k ladder (third_party/concave_fitting.cpp: k = 3, then the primes, at most 8 attempts)
            int hs = 0;
            bool haveRing = false;
            if (n >= 3 && MODE == kPolyLadder)
            {
                // rungs 2, 3, 4 (k = 5, 7, 11) at once, one per wave; if none has a hull, rungs 5, 6, 7 (k = 13, 17, 21).  The
                // lowest rung with a simple hull wins and finishes the polygon -- a higher rung gives up as soon as it sees a
                // lower one succeed -- and if none has one, wave 0 takes the convex fallback (every wave holds the same points).
                const int ladder[8] = {3, 3, 5, 7, 11, 13, 17, 21};
                constexpr int kNoRung = 99;
                if (threadIdx.x == 0)
                    s_ok[8] = kNoRung; // the lowest rung that has a hull so far
                __syncthreads();
#ifdef CAPE_POLY_PROFILE
                const unsigned long long ladderStart = __builtin_amdgcn_s_memtime();
#endif
                // wave w walks rung w + 2 and, if that fails, rung 7 - w (the wave with the longest first walk, k = 11, takes the
                // shortest second one, k = 13: 2-4 % on the whole pass) -- without waiting for the others; it stops as soon as a
                // rung below its own has a hull (it can no longer win)
                int myRung = kNoRung;
                for (int stage = 0; stage < 2; ++stage)
                {
                    const int rung = stage == 0 ? 2 + wave : 7 - wave; // (k = 5, 21), (7, 17), (11, 13): the long first walk is followed by the short second one
                    if (__builtin_amdgcn_readfirstlane(*(volatile int*)(s_ok + 8)) < rung)
                        break;
                    if (ladder[rung] > n)
                        continue; // the reference stops climbing when the next k exceeds the point count (concave_fitting.cpp:86-87)
                    CAPE_PCOUNT(8, 1); // hull attempts
                    const bool ok = concave_hull_k<CAP>(L, n, first, ladder[rung], lane, hs, s_ok + 8, rung);
                    CAPE_PTICK(1); // hull walks (incl. the all-points-inside check)
                    if (ok)
                    {
                        myRung = rung;
                        if (lane == 0)
                            atomicMin(s_ok + 8, rung);
                        break;
                    }
                }
                __syncthreads();
                const int best = s_ok[8];
                __syncthreads(); // the word is reset for the next plane
#ifdef CAPE_POLY_PROFILE
                if (wave == 0)
                {
                    // slots 12 (no rung: convex fallback), 14 .. 19 (winning rung 2 .. 7), 20 points, 21 ticks, 22 slowest plane of the frame
                    const unsigned long long dt = __builtin_amdgcn_s_memtime() - ladderStart;
                    CAPE_PCOUNT(12 + (best == kNoRung ? 0 : best), 1);
                    CAPE_PCOUNT(20, n);
                    CAPE_PCOUNT(21, dt);
                    if (lane == 0 && p.prof)
                        atomicMax(&p.prof[(size_t)frame * kProfileSlots + 22], dt);
                }
#endif
                const int winner = best == kNoRung ? -1 : (best <= 4 ? best - 2 : 7 - best);
                haveRing = winner >= 0 && myRung == best;
                if (wave != (winner < 0 ? 0 : winner))
                    continue;
                if (winner < 0)
                    haveRing = false;
            }
            else if (n >= 3)
            {
                const int ladder[8] = {3, 3, 5, 7, 11, 13, 17, 21};
                const int aLast = MODE == kPolyFull ? 7 : 0;
                for (int a = 0; a <= aLast && !haveRing; ++a)
                {
                    const int k = ladder[a];
                    if (a > 0 && k > n)
                        break; // the next k exceeds the point count (concave_fitting.cpp:86-87)
                    // (the ladder's second rung repeats the first: the run is a pure function of the points and k, so a
                    //  failed k = 3 fails again -- the host class runs it twice, the result is the same)
                    if (a == 1)
                        continue;
                    CAPE_PCOUNT(8, 1); // hull attempts
                    haveRing = concave_hull_k<CAP>(L, n, first, k, lane, hs);
                    CAPE_PTICK(1); // hull walks (incl. the all-points-inside check)
                }
                if (MODE == kPolyFirstRung && !haveRing)
                {
                    // the first rung failed: the ladder kernel takes the plane
                    if (lane == 0)
                    {
                        uint32_t* ladderList = p.lists + (size_t)kPolyLadder * p.listStride;
                        if (n >= kPolyBigPoints)
                            ladderList[kPolyListHeader + atomicAdd(&ladderList[0], 1u)] = entry;
                        else
                            ladderList[kPolyListHeader + listCapacity - 1 - atomicAdd(&ladderList[1], 1u)] = entry;
                    }
                    continue;
                }
            }
            unsigned short* ring = L.ring;
            int rn = 0;
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
                // A hull that touches or crosses itself is dissolved with Boost set operations in the reference
                // (correct_boost_polygon.hpp:229-330); here it goes the way of a failed hull: the convex hull
                if (!ring_is_simple(L.pts, ring, rn, lane))
                    haveRing = false;
                CAPE_PTICK(2); // simple-ring test of the oriented hull
            }
            if (!haveRing)
            {
                // ---- compute_convex_hull (monotone chain over the sorted, deduplicated points), reversed to clockwise.
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
            CAPE_PTICK(3); // orientation, validity of the oriented ring, convex fallback
            CAPE_PCOUNT(9, rn); // vertices before simplification
            CAPE_PCOUNT(10, nPts); // points
            area = rn >= 3 ? fabs(ring_area_signed(L.pts, ring, rn)) : 0.0;
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
                    // wave maximum of the distance, smallest index among the equal ones
                    unsigned long long mx = best;
#pragma unroll
                    for (int o = 32; o > 0; o >>= 1)
                    {
                        const unsigned long long ob = (unsigned long long)__shfl_xor((long long)mx, o);
                        mx = ob > mx ? ob : mx;
                    }
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
                        CAPE_POLY_SYNC();
                    }
                }
            }
            CAPE_PTICK(4); // area + simplify
            // ---- what Primitive_Detection keeps: a valid polygon of at least three vertices (primitive_detection.cpp:623-631)
            if (rn >= 3 && ring_is_simple(L.pts, ring, rn, lane))
                flags |= CAPE_POLY_VALID;
            for (int i = lane; i < rn; i += 64)
                vout[i] = L.pts[ring[i]];
            count = rn;
            CAPE_PTICK(5); // final validity, vertex stores
            CAPE_PCOUNT(11, 1); // planes
        }
        if (lane == 0)
        {
            out->x_axis[0] = xax; out->x_axis[1] = xay; out->x_axis[2] = xaz;
            out->y_axis[0] = yax; out->y_axis[1] = yay; out->y_axis[2] = yaz;
            out->center[0] = cx; out->center[1] = cy; out->center[2] = cz;
            out->area = area;
            out->vertex_offset = S.boundary_offset;
            out->vertex_count = (uint32_t)count;
            out->flags = flags;
            out->segment = (uint32_t)seg;
        }
        CAPE_POLY_SYNC();
    }
}

// one wavefront per frame, lane j <- segment j: the work lists of the three polygon kernels, and the records of the segments
// that need no kernel (not an output plane: empty record)
constexpr int kListFrames = 16; // frames (waves) of a list workgroup: ONE atomic per list and workgroup -- a counter that every
                                // frame's wave bumps on its own serialises 4 096 atomics on one address (74 us of the pass)
__global__ __launch_bounds__(64 * kListFrames) void cape_polygon_list_kernel(PolygonParams p, int nFrames)
{
    __shared__ unsigned s_count[3][kListFrames], s_base[3][kListFrames];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int frame = blockIdx.x * kListFrames + wave;
    bool isOut = false;
    int nPts = 0;
    if (frame < nFrames)
    {
        const cape_frame_record& rec = p.records[frame];
        isOut = lane < rec.header.n_plane_segments && rec.segments[lane].is_output != 0;
        nPts = isOut ? (int)rec.segments[lane].boundary_count : 0;
    }
    // (segments that are no output plane keep the empty record launch_polygons' memset left)
    // 257 .. 1 024 candidates: the large instance; everything else (incl. what it will only flag: too few / too many points)
    // goes to the small one
    const bool large = isOut && nPts > kPolySmallPoints && nPts <= kPolyMaxPoints;
    const bool small = isOut && !large;
    // small-instance planes: the big ones from the front of the list, the others from its back (see cape_polygon_kernel)
    const bool big = small && nPts >= kPolyBigPoints && nPts <= kPolySmallPoints;
    const bool rest = small && !big;
    const unsigned long long mb = __ballot(big), mr = __ballot(rest), ml = __ballot(large);
    uint32_t* listS = p.lists + (size_t)kPolyFirstRung * p.listStride;
    uint32_t* listL = p.lists + (size_t)kPolyFull * p.listStride;
    const unsigned listCapacity = p.listStride - kPolyListHeader;
    if (lane == 0)
    {
        s_count[0][wave] = (unsigned)__popcll(mb);
        s_count[1][wave] = (unsigned)__popcll(mr);
        s_count[2][wave] = (unsigned)__popcll(ml);
    }
    __syncthreads();
    if (threadIdx.x < 3)
    {
        unsigned total = 0;
        for (int w = 0; w < kListFrames; ++w)
        {
            s_base[threadIdx.x][w] = total;
            total += s_count[threadIdx.x][w];
        }
        unsigned* counter = threadIdx.x == 0 ? &listS[0] : threadIdx.x == 1 ? &listS[1] : &listL[0];
        const unsigned base = total ? atomicAdd(counter, total) : 0u;
        for (int w = 0; w < kListFrames; ++w)
            s_base[threadIdx.x][w] += base;
    }
    __syncthreads();
    const unsigned baseB = s_base[0][wave], baseR = s_base[1][wave], baseL = s_base[2][wave];
    const unsigned long long below = (1ull << lane) - 1ull;
    const unsigned entry = ((unsigned)frame << 8) | (unsigned)lane;
    if (big)
        listS[kPolyListHeader + baseB + __popcll(mb & below)] = entry;
    if (rest)
        listS[kPolyListHeader + listCapacity - 1 - (baseR + __popcll(mr & below))] = entry;
    if (large)
        listL[kPolyListHeader + baseL + __popcll(ml & below)] = entry;
}

size_t polygon_lds_bytes(int cap)
{
    size_t b = (size_t)cap * 16;            // pts
    b += 2 * ((size_t)cap + 2) * 2;         // hull, ring
    b += (size_t)cap * 4;                   // stack
    b += (size_t)cap + cap + 2;             // used, keep
    return (b + 15) & ~(size_t)15;
}

hipError_t launch_polygons(const PolygonParams& p, int nFrames, hipStream_t stream)
{
    // the headers of the three work lists (list m at p.lists + m * listStride)
    for (int m = 0; m < 3; ++m)
        if (const hipError_t e = hipMemsetAsync(p.lists + (size_t)m * p.listStride, 0, kPolyListHeader * sizeof(uint32_t), stream); e != hipSuccess)
            return e;
    if (const hipError_t e = hipMemsetAsync(p.polygons, 0, (size_t)nFrames * CAPE_MAX_PLANES * sizeof(cape_polygon), stream); e != hipSuccess)
        return e;
    hipLaunchKernelGGL(cape_polygon_list_kernel, dim3((nFrames + kListFrames - 1) / kListFrames), dim3(64 * kListFrames), 0, stream, p, nFrames);
    if (const hipError_t e = hipGetLastError(); e != hipSuccess)
        return e;
    const int ldsSmall = (int)polygon_lds_bytes(kPolySmallPoints), ldsLarge = (int)polygon_lds_bytes(kPolyMaxPoints);
    // fixed grids striding over their lists: enough workgroups to fill the device, never more than the list can hold
    const int maxPlanes = nFrames * CAPE_MAX_PLANES;
    const int gridSmall = std::min((maxPlanes + kPolyWavesPerGroup - 1) / kPolyWavesPerGroup, p.computeUnits * 5);
    const int gridLadder = std::min(maxPlanes, p.computeUnits * 6);
    const int gridLarge = std::min((maxPlanes + kPolyWavesPerGroup - 1) / kPolyWavesPerGroup, p.computeUnits);
    hipLaunchKernelGGL((cape_polygon_kernel<kPolySmallPoints, kPolyFirstRung>), dim3(gridSmall), dim3(64 * kPolyWavesPerGroup),
                       (size_t)ldsSmall * kPolyWavesPerGroup + 64, stream, p, nFrames, ldsSmall);
    if (const hipError_t e = hipGetLastError(); e != hipSuccess)
        return e;
    // planes that failed the first rung: one workgroup of three waves each
    hipLaunchKernelGGL((cape_polygon_kernel<kPolySmallPoints, kPolyLadder>), dim3(gridLadder), dim3(64 * kLadderWaves),
                       (size_t)ldsSmall * kLadderWaves + 64, stream, p, nFrames, ldsSmall);
    if (const hipError_t e = hipGetLastError(); e != hipSuccess)
        return e;
    if (p.boundaryCapacity > kPolySmallPoints) // a plane cannot hold more boundary points than the frame
        hipLaunchKernelGGL((cape_polygon_kernel<kPolyMaxPoints, kPolyFull>), dim3(gridLarge), dim3(64 * kPolyWavesPerGroup),
                           (size_t)ldsLarge * kPolyWavesPerGroup + 64, stream, p, nFrames, ldsLarge);
    return hipGetLastError();
}

} // namespace cape
