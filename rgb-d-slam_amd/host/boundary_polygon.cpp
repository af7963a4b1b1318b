#include "boundary_polygon.hpp"

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <limits>
#include <stdexcept>

namespace rgbd_slam::utils {

namespace {

inline double dot(const vector3& a, const vector3& b) { return (a[0] * b[0] + a[1] * b[1]) + a[2] * b[2]; }
inline vector3 cross(const vector3& a, const vector3& b)
{
    return {a[1] * b[2] - a[2] * b[1], a[2] * b[0] - a[0] * b[2], a[0] * b[1] - a[1] * b[0]};
}
inline double norm(const vector3& a) { return std::sqrt(dot(a, a)); }
inline vector3 normalized(const vector3& a)
{
    const double n = norm(a);
    return n > 0 ? vector3 {a[0] / n, a[1] / n, a[2] / n} : a;
}
inline bool double_equal(double a, double b, double eps = std::numeric_limits<double>::epsilon()) { return std::abs(a - b) <= eps; }

// select_correct_transform, polygon.cpp:50-68
vector3 select_correct_transform(const vector3& normal)
{
    const double distX = std::abs(normal[0]), distY = std::abs(normal[1]), distZ = std::abs(normal[2]);
    const double res = std::min(distX, std::min(distY, distZ));
    if (double_equal(res, distX, 0.1))
        return {1, 0, 0};
    if (double_equal(res, distY, 0.1))
        return {0, 1, 0};
    if (double_equal(res, distZ, 0.1))
        return {0, 0, 1};
    return normalized({normal[2], normal[0], normal[1]});
}

inline double cross2(const vector2& o, const vector2& a, const vector2& b)
{
    return (a[0] - o[0]) * (b[1] - o[1]) - (a[1] - o[1]) * (b[0] - o[0]);
}

// proper or touching intersection of the open segments (a1,a2) and (b1,b2); shared endpoints do not count
bool segments_intersect(const vector2& a1, const vector2& a2, const vector2& b1, const vector2& b2)
{
    // separated bounding boxes cannot cross or touch: settles nearly every pair of a ring's edges with four compares (the
    // simple-ring test is quadratic in the ring size and ran ~3x per polygon)
    if (std::max(a1[0], a2[0]) < std::min(b1[0], b2[0]) || std::max(b1[0], b2[0]) < std::min(a1[0], a2[0]) ||
        std::max(a1[1], a2[1]) < std::min(b1[1], b2[1]) || std::max(b1[1], b2[1]) < std::min(a1[1], a2[1]))
        return false;
    auto same = [](const vector2& p, const vector2& q) { return p[0] == q[0] && p[1] == q[1]; };
    if (same(a1, b1) || same(a1, b2) || same(a2, b1) || same(a2, b2))
        return false;
    const double d1 = cross2(b1, b2, a1), d2 = cross2(b1, b2, a2), d3 = cross2(a1, a2, b1), d4 = cross2(a1, a2, b2);
    if (((d1 > 0 && d2 < 0) || (d1 < 0 && d2 > 0)) && ((d3 > 0 && d4 < 0) || (d3 < 0 && d4 > 0)))
        return true;
    auto on = [](const vector2& p, const vector2& q, const vector2& r) {
        return std::min(p[0], q[0]) <= r[0] && r[0] <= std::max(p[0], q[0]) && std::min(p[1], q[1]) <= r[1] &&
               r[1] <= std::max(p[1], q[1]);
    };
    if (d1 == 0 && on(b1, b2, a1)) return true;
    if (d2 == 0 && on(b1, b2, a2)) return true;
    if (d3 == 0 && on(a1, a2, b1)) return true;
    if (d4 == 0 && on(a1, a2, b2)) return true;
    return false;
}

// crossing-number test; points on the boundary count as inside when `closed`
bool point_in_ring(const vector2& p, const std::vector<vector2>& ring, bool closed)
{
    const size_t n = ring.size();
    bool inside = false;
    for (size_t i = 0, j = n - 1; i < n; j = i++)
    {
        const vector2 &a = ring[i], &b = ring[j];
        if (cross2(a, b, p) == 0 && std::min(a[0], b[0]) <= p[0] && p[0] <= std::max(a[0], b[0]) &&
            std::min(a[1], b[1]) <= p[1] && p[1] <= std::max(a[1], b[1]))
            return closed;
        if (((a[1] > p[1]) != (b[1] > p[1])) && (p[0] < (b[0] - a[0]) * (p[1] - a[1]) / (b[1] - a[1]) + a[0]))
            inside = !inside;
    }
    return inside;
}

double ring_area_signed(const std::vector<vector2>& r)
{
    double s = 0;
    for (size_t i = 0, j = r.size() - 1; i < r.size(); j = i++)
        s += (r[j][0] * r[i][1] - r[i][0] * r[j][1]);
    return 0.5 * s;
}

// ---- third_party/concave_fitting.cpp:186-201: every comparison of the hull carries a DBL_EPSILON slack.  Restated literally:
//      IEEE additions and comparisons, so the device hull (csrc/cape_polygon.hip) evaluates the very same expressions.
constexpr double kHullEps = std::numeric_limits<double>::epsilon();
inline bool eq_eps(double a, double b) { return std::abs(a - b) <= kHullEps; }
inline bool zero_eps(double a) { return std::abs(a) <= kHullEps; }
inline bool lt_eps(double a, double b) { return a < (b - kHullEps); }
inline bool le_eps(double a, double b) { return a <= (b + kHullEps); }
inline bool gt_eps(double a, double b) { return a > (b + kHullEps); }
inline bool points_equal(const vector2& a, const vector2& b) { return eq_eps(a[0], b[0]) && eq_eps(a[1], b[1]); }

// Intersects, concave_fitting.cpp:426-463: the crossing point of the two carrier lines, then four bounding tests with the
// slack.  Parallel (and collinear) segments never intersect (:446-449).  The early-out settles nearly every pair without
// the two divisions: boxes further apart than 1e-9 cannot both hold the crossing point, whose tests allow ~4e-16.
bool hull_edges_intersect(const vector2& a1p, const vector2& a2p, const vector2& b1p, const vector2& b2p)
{
    const double ax1 = a1p[0], ay1 = a1p[1], ax2 = a2p[0], ay2 = a2p[1];
    const double bx1 = b1p[0], by1 = b1p[1], bx2 = b2p[0], by2 = b2p[1];
    const double aminx = std::min(ax1, ax2), amaxx = std::max(ax1, ax2), aminy = std::min(ay1, ay2), amaxy = std::max(ay1, ay2);
    const double bminx = std::min(bx1, bx2), bmaxx = std::max(bx1, bx2), bminy = std::min(by1, by2), bmaxy = std::max(by1, by2);
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

// PointInPolygon, concave_fitting.cpp:393-423, over the hull's vertex list as the walk left it (closed or not; consecutive
// pairs only, no wrap) -- with its quirk: a point whose ray towards +x crosses NO edge counts as inside (:416-417), so a
// hull may leave points out on its right and still pass.
bool point_in_hull(const vector2& p, const std::vector<vector2>& pts, const std::vector<size_t>& hull)
{
    if (hull.size() <= 2)
        return false;
    const double x = p[0], y = p[1];
    int inout = 0;
    for (size_t v = 0; v + 1 < hull.size(); ++v)
    {
        const vector2 &q0 = pts[hull[v]], &q1 = pts[hull[v + 1]];
        if (((le_eps(q0[1], y) && lt_eps(y, q1[1])) || (le_eps(q1[1], y) && lt_eps(y, q0[1]))) && !zero_eps(q1[1] - q0[1]) &&
            lt_eps(x, q0[0] + ((q1[0] - q0[0]) * (y - q0[1]) / (q1[1] - q0[1]))))
            inout++;
    }
    if (inout == 0)
        return true;
    return inout % 2 != 0;
}

// Two directions closer than 2 DBL_EPSILON radians are ONE direction.  The reference compares `-atan2` angles in [0, 2 pi) with a
// slack of DBL_EPSILON (GreaterThan, concave_fitting.cpp:200; SortByAngle :296-315); the angles themselves resolve 2.2e-16 ..
// 8.9e-16 rad depending on their size, so two directions a few 1e-16 apart come out as the same double, or one ulp apart and "equal"
// by the rounding of `b + DBL_EPSILON`.  What that makes of a neighbour on the line of the previous edge is the angle 0 (or pi), not
// 2 pi - 1e-16, and two neighbours on one ray from the current point stay nearest first.  Every such case met in 2.8e5 planes
// (profiles/r06_polygon_cpu_sweep.txt) lies below 4e-16 and is decided like the reference decides it with this threshold; an exact
// cross-product sign decided three of them the other way.  cr = ax by - ay bx; the test is sin^2 <= (2 eps)^2 in products only.
// The cheap first test makes the products rare: it cannot hide a case as long as |a| |b| < 2.2e12 (millimetres: vectors < 1.4 km).
inline bool same_direction_line(double cr, double ax, double ay, double bx, double by)
{
    constexpr double kEps2 = 1.9721522630525295135293214132069655741830160877724e-31; // 2^-102 = (2 DBL_EPSILON)^2
    return std::abs(cr) < 1e-3 && cr * cr <= kEps2 * ((ax * ax + ay * ay) * (bx * bx + by * by));
}

// FindMinYPoint, concave_fitting.cpp:231-243: std::min_element under (y ascending, then x DESCENDING), comparisons with the slack
size_t find_min_y_point(const std::vector<vector2>& pts)
{
    size_t smallest = 0;
    for (size_t i = 1; i < pts.size(); ++i)
    {
        const vector2 &a = pts[i], &b = pts[smallest];
        const bool less = eq_eps(a[1], b[1]) ? gt_eps(a[0], b[0]) : lt_eps(a[1], b[1]);
        if (less)
            smallest = i;
    }
    return smallest;
}

// One run of the Moreira-Santos k-nearest-neighbours walk, ConcaveHull of concave_fitting.cpp:93-183, for one k.  `h` receives
// the hull as point indices exactly as the reference's vector holds it: it ends with the start point again when the walk came
// back to it, and holds every point once when it ran out of points first (:126: `hull.size() != pointList.size()`).
// Two restatements, both forced by what the device can reproduce bit for bit:
//  * FLANN's approximate search over randomized kd-trees (:109-111, :258-288) is the EXACT k nearest visible points, nearest
//    first, ordered by (squared distance, point index): an exact tie goes to the smaller index.  (Through round 5 distances
//    within 2^-42 of each other went to the smaller index too, because the device selected on a 64-bit key that carried the
//    index in the distance's ten lowest mantissa bits; that moved 1 hull in ~10 000 away from the oracle's, and the device now
//    re-selects such a step on the full bit patterns.)  The start point re-enters the index at step 4 under the id n like the
//    reference's copy (:131): in an exact distance tie it comes last.
//  * SortByAngle (:296-315) orders the candidates by `-atan2` angles, descending, with the slack.  Here the clockwise turn
//    from the previous edge is never computed as an angle: a class (same direction / less than half a turn / opposite / more)
//    from the signs of one cross and one dot product, and inside a class one more cross product -- additions,
//    multiplications and comparisons only, scanned nearest-first (the reference's sort of <= 16 candidates is an insertion
//    sort: equal angles stay nearest-first there too).  An atan2 from glibc and one from ocml need not agree on two
//    candidates a rounding error apart; these predicates do.  Directions within 2 DBL_EPSILON of each other count as equal
//    (same_direction_line), which is what the reference's slack makes of them.  prevAngle = 0 (:122) is the +x direction, and so is the
//    "direction" of a duplicate of the current point (atan2(+0, +0) = 0).
bool concave_hull_k(const std::vector<vector2>& pts, size_t first, size_t k, std::vector<size_t>& h)
{
    const size_t n = pts.size();
    h.clear();
    if (n < 3)
        return true; // (:97-100)
    if (n == 3)
    {
        h = {0, 1, 2}; // (:101-105)
        return true;
    }
    std::vector<char> removed(n, 0);
    h.push_back(first);
    removed[first] = 1;
    size_t current = first;
    double prevX = 1.0, prevY = 0.0;
    size_t step = 1;
    std::vector<std::pair<uint64_t, size_t>> cand; // reused from step to step
    cand.reserve(n);
    while ((!points_equal(pts[current], pts[first]) || step == 1) && h.size() != n)
    {
        if (step == 4)
            removed[first] = 0; // the start point is put back into the index once the hull has three edges (:128-134)
        // k nearest visible neighbours of the current point (itself removed; a duplicate of it is a neighbour at distance 0)
        cand.clear();
        for (size_t i = 0; i < n; ++i)
            if (!removed[i])
            {
                const double dx = pts[current][0] - pts[i][0], dy = pts[current][1] - pts[i][1];
                const double d2 = dx * dx + dy * dy;
                uint64_t bits;
                std::memcpy(&bits, &d2, sizeof bits);
                cand.emplace_back(bits, (i == first && step >= 4) ? n : i); // the re-inserted start point carries the id n (:131)
            }
        const size_t kk = std::min(k, cand.size());
        std::partial_sort(cand.begin(), cand.begin() + kk, cand.end());
        cand.resize(kk);
        auto turn_class = [&](double vx, double vy) {
            const double cr = prevX * vy - prevY * vx, dt = prevX * vx + prevY * vy;
            if (same_direction_line(cr, prevX, prevY, vx, vy))
                return dt > 0 ? 0 : 2;
            return cr < 0 ? 1 : 3;
        };
        struct Cand
        {
            size_t idx;
            double vx, vy;
            int cls;
        };
        Cand cs[32];
        const size_t kc = cand.size();
        for (size_t c = 0; c < kc; ++c)
        {
            const size_t i = cand[c].second == n ? first : cand[c].second;
            double vx = pts[i][0] - pts[current][0], vy = pts[i][1] - pts[current][1];
            if (vx == 0 && vy == 0)
                vx = 1.0;
            cs[c] = {i, vx, vy, turn_class(vx, vy)};
        }
        auto turns_further = [](const Cand& a, const Cand& b) {
            if (a.cls != b.cls)
                return a.cls > b.cls;
            if (a.cls == 0 || a.cls == 2)
                return false;
            const double cr = b.vx * a.vy - b.vy * a.vx;
            return cr < 0 && !same_direction_line(cr, b.vx, b.vy, a.vx, a.vy); // a lies clockwise of b inside the same open half turn
        };
        unsigned tried = 0;
        bool found = false;
        size_t next = 0;
        for (size_t t = 0; t < kc && !found; ++t)
        {
            int best = -1;
            for (size_t c = 0; c < kc; ++c)
            {
                if ((tried >> c) & 1u)
                    continue;
                if (best < 0 || turns_further(cs[c], cs[best]))
                    best = static_cast<int>(c);
            }
            tried |= 1u << best;
            const size_t cnd = cs[best].idx;
            // (:146-163) the candidate edge against the hull edges (h[e], h[e+1]), e = h.size()-3 .. lastPoint: not the edge that
            // ends at the current point, and not the first edge when the candidate is the start point
            const size_t lastPoint = points_equal(pts[cnd], pts[first]) ? 1 : 0;
            bool its = false;
            for (size_t e = lastPoint; e + 2 < h.size() && !its; ++e)
                its = hull_edges_intersect(pts[current], pts[cnd], pts[h[e]], pts[h[e + 1]]);
            if (!its)
            {
                found = true;
                next = cnd;
            }
        }
        if (!found)
            return false; // every candidate crosses the hull (:166-169), or none is left
        prevX = pts[current][0] - pts[next][0]; // Angle(hull[step], hull[step - 1]) (:174): looking back along the new edge
        prevY = pts[current][1] - pts[next][1];
        if (prevX == 0 && prevY == 0)
            prevX = 1.0;
        current = next;
        h.push_back(current);
        removed[current] = 1;
        ++step;
    }
    // every point that is not a hull vertex must pass PointInPolygon (:176-182)
    std::vector<char> onHull(n, 0);
    for (size_t i : h)
        onHull[i] = 1;
    for (size_t i = 0; i < n; ++i)
        if (!onHull[i] && !point_in_hull(pts[i], pts, h))
            return false;
    return true;
}

// squared distance of p to the SEGMENT (a, b): boost::geometry's projected_point strategy in its comparable form, the
// measure of its Douglas-Peucker (strategy::simplify::douglas_peucker)
double segment_distance2(const vector2& p, const vector2& a, const vector2& b)
{
    const double vx = b[0] - a[0], vy = b[1] - a[1], wx = p[0] - a[0], wy = p[1] - a[1];
    const double c1 = wx * vx + wy * vy;
    if (c1 <= 0)
        return wx * wx + wy * wy;
    const double c2 = vx * vx + vy * vy;
    if (c2 <= c1)
    {
        const double ux = p[0] - b[0], uy = p[1] - b[1];
        return ux * ux + uy * uy;
    }
    const double t = c1 / c2;
    const double qx = a[0] + t * vx, qy = a[1] + t * vy;
    return (p[0] - qx) * (p[0] - qx) + (p[1] - qy) * (p[1] - qy);
}

// boost::geometry::simplify's Douglas-Peucker: the farthest point from the segment between the two kept ends (squared
// distances, the first of equal maxima), kept when it lies further than the threshold
void douglas_peucker(const std::vector<vector2>& in, size_t a, size_t b, double eps2, std::vector<char>& keep)
{
    if (b <= a + 1)
        return;
    double dmax = -1;
    size_t idx = a;
    for (size_t i = a + 1; i < b; ++i)
    {
        const double d = segment_distance2(in[i], in[a], in[b]);
        if (d > dmax)
        {
            dmax = d;
            idx = i;
        }
    }
    if (dmax > eps2)
    {
        keep[idx] = 1;
        douglas_peucker(in, a, idx, eps2, keep);
        douglas_peucker(in, idx, b, eps2, keep);
    }
}

bool ring_is_simple(const std::vector<vector2>& r)
{
    const size_t n = r.size();
    if (n < 3)
        return false;
    for (size_t i = 0; i < n; ++i)
        for (size_t j = i + 1; j < n; ++j)
        {
            if (j == i + 1 || (i == 0 && j == n - 1))
                continue; // adjacent edges share a vertex
            if (segments_intersect(r[i], r[(i + 1) % n], r[j], r[(j + 1) % n]))
                return false;
        }
    return std::abs(ring_area_signed(r)) > 0;
}

// The first pair (i, j), i < j, of non-adjacent edges of an open ring that share a point; `proper`: they cross at a point
// interior to both (the only kind dissolve_crossings undoes).
bool first_contact(const std::vector<vector2>& r, size_t& ci, size_t& cj, bool& proper)
{
    const size_t n = r.size();
    for (size_t i = 0; i < n; ++i)
        for (size_t j = i + 1; j < n; ++j)
        {
            if (j == i + 1 || (i == 0 && j == n - 1))
                continue;
            const vector2 &a1 = r[i], &a2 = r[(i + 1) % n], &b1 = r[j], &b2 = r[(j + 1) % n];
            if (!segments_intersect(a1, a2, b1, b2))
                continue;
            const double d1 = cross2(b1, b2, a1), d2 = cross2(b1, b2, a2), d3 = cross2(a1, a2, b1), d4 = cross2(a1, a2, b2);
            proper = ((d1 > 0 && d2 < 0) || (d1 < 0 && d2 > 0)) && ((d3 > 0 && d4 < 0) || (d3 < 0 && d4 > 0));
            ci = i;
            cj = j;
            return true;
        }
    return false;
}

// What the reference's repair does with a hull that CROSSES itself (its own Intersects misses a crossing with an axis-parallel
// edge by a rounding error: about one plane in a hundred): third_party/correct_boost_polygon.hpp:127-160 makes the crossing point
// a pseudo-vertex of both edges, :199-330 traces the ring apart there -- the ring that runs on past the crossing and the loop it
// cuts off -- and the constructor keeps result[0] (polygon.cpp:209-211); the pieces are ordered by decreasing |area| (:371-375),
// so that is the bigger one.  Restated for PROPER crossings, one at a time (the first in edge order), at most eight; a ring that
// merely touches itself, or still crosses after eight cuts, goes to the convex hull.  The device runs the same statements.
bool dissolve_crossings(std::vector<vector2>& r)
{
    for (int cut = 0; cut < 8; ++cut)
    {
        size_t i = 0, j = 0;
        bool proper = false;
        if (!first_contact(r, i, j, proper))
            return r.size() >= 3 && std::abs(ring_area_signed(r)) > 0;
        if (!proper)
            return false;
        const size_t n = r.size();
        const vector2 a = r[i], b = r[(i + 1) % n], c = r[j], d = r[(j + 1) % n];
        const double rx = b[0] - a[0], ry = b[1] - a[1], sx = d[0] - c[0], sy = d[1] - c[1];
        const double t = ((c[0] - a[0]) * sy - (c[1] - a[1]) * sx) / (rx * sy - ry * sx);
        const vector2 x {a[0] + t * rx, a[1] + t * ry};
        std::vector<vector2> outer, loop; // r[0..i], X, r[j+1..]   and   X, r[i+1..j]
        for (size_t k = 0; k <= i; ++k)
            outer.push_back(r[k]);
        outer.push_back(x);
        for (size_t k = j + 1; k < n; ++k)
            outer.push_back(r[k]);
        loop.push_back(x);
        for (size_t k = i + 1; k <= j; ++k)
            loop.push_back(r[k]);
        r = std::abs(ring_area_signed(loop)) > std::abs(ring_area_signed(outer)) ? loop : outer;
        if (r.size() >= 3 && ring_area_signed(r) > 0)
            std::reverse(r.begin() + 1, r.end()); // clockwise (correct_boost_polygon.hpp:358-369)
    }
    return false;
}

// Area of the intersection of two simple rings.  The plane is cut into vertical slabs at every vertex and every
// edge-edge crossing; inside a slab no two edges cross, so each ring is a stack of trapezoids ordered by y and the
// overlap of the two stacks is a sum of trapezoids.
double rings_inter_area(const std::vector<vector2>& A, const std::vector<vector2>& B)
{
    if (A.size() < 3 || B.size() < 3)
        return 0.0;
    struct Edge { vector2 a, b; }; // a.x < b.x
    auto edges_of = [](const std::vector<vector2>& r) {
        std::vector<Edge> e;
        for (size_t i = 0, j = r.size() - 1; i < r.size(); j = i++)
        {
            vector2 p = r[j], q = r[i];
            if (p[0] == q[0])
                continue; // vertical edges bound no area in x
            if (p[0] > q[0])
                std::swap(p, q);
            e.push_back({p, q});
        }
        return e;
    };
    const std::vector<Edge> ea = edges_of(A), eb = edges_of(B);
    std::vector<double> xs;
    for (const auto& p : A) xs.push_back(p[0]);
    for (const auto& p : B) xs.push_back(p[0]);
    for (const Edge& e : ea)
        for (const Edge& f : eb)
        {
            const double d1x = e.b[0] - e.a[0], d1y = e.b[1] - e.a[1], d2x = f.b[0] - f.a[0], d2y = f.b[1] - f.a[1];
            const double den = d1x * d2y - d1y * d2x;
            if (den == 0)
                continue; // parallel / collinear: no isolated crossing
            const double t = ((f.a[0] - e.a[0]) * d2y - (f.a[1] - e.a[1]) * d2x) / den;
            const double u = ((f.a[0] - e.a[0]) * d1y - (f.a[1] - e.a[1]) * d1x) / den;
            if (t > 0 && t < 1 && u > 0 && u < 1)
                xs.push_back(e.a[0] + t * d1x);
        }
    std::sort(xs.begin(), xs.end());
    xs.erase(std::unique(xs.begin(), xs.end()), xs.end());
    auto y_at = [](const Edge& e, double x) { return e.a[1] + (e.b[1] - e.a[1]) * ((x - e.a[0]) / (e.b[0] - e.a[0])); };
    double area = 0.0;
    for (size_t s = 0; s + 1 < xs.size(); ++s)
    {
        const double x0 = xs[s], x1 = xs[s + 1], xm = 0.5 * (x0 + x1);
        if (!(x1 > x0))
            continue;
        auto stack = [&](const std::vector<Edge>& es) {
            std::vector<std::pair<double, const Edge*>> st;
            for (const Edge& e : es)
                if (e.a[0] <= x0 && e.b[0] >= x1)
                    st.emplace_back(y_at(e, xm), &e);
            std::sort(st.begin(), st.end(), [](const auto& l, const auto& r) { return l.first < r.first; });
            return st;
        };
        const auto sa = stack(ea), sb = stack(eb);
        for (size_t i = 0; i + 1 < sa.size(); i += 2)
            for (size_t j = 0; j + 1 < sb.size(); j += 2)
            {
                const auto& lo = (sa[i].first > sb[j].first) ? sa[i] : sb[j];
                const auto& hi = (sa[i + 1].first < sb[j + 1].first) ? sa[i + 1] : sb[j + 1];
                if (hi.first <= lo.first)
                    continue;
                const double h0 = y_at(*hi.second, x0) - y_at(*lo.second, x0);
                const double h1 = y_at(*hi.second, x1) - y_at(*lo.second, x1);
                area += 0.5 * (h0 + h1) * (x1 - x0);
            }
    }
    return area;
}

} // namespace

std::pair<vector3, vector3> get_plane_coordinate_system(const vector3& normal)
{
    if (!double_equal(norm(normal), 1.0, 1e-9))
        throw std::invalid_argument("get_plane_coordinate_system: The normal should have a norm of 1");
    const vector3 r = select_correct_transform(normal);
    const vector3 xAxis = normalized(cross(normal, r));
    const vector3 yAxis = normalized(cross(normal, xAxis));
    return {xAxis, yAxis};
}

vector2 get_projected_plan_coordinates(const vector3& p, const vector3& c, const vector3& xAxis, const vector3& yAxis)
{
    const vector3 d {p[0] - c[0], p[1] - c[1], p[2] - c[2]};
    return {dot(xAxis, d), dot(yAxis, d)};
}

vector3 get_point_from_plane_coordinates(const vector2& p, const vector3& c, const vector3& xAxis, const vector3& yAxis)
{
    return {c[0] + p[0] * xAxis[0] + p[1] * yAxis[0], c[1] + p[0] * xAxis[1] + p[1] * yAxis[1],
            c[2] + p[0] * xAxis[2] + p[1] * yAxis[2]};
}

std::vector<vector2> Polygon::compute_convex_hull(const std::vector<vector2>& in) noexcept
{
    std::vector<vector2> p = in;
    std::sort(p.begin(), p.end());
    p.erase(std::unique(p.begin(), p.end()), p.end());
    if (p.size() < 3)
        return p;
    std::vector<vector2> h(2 * p.size());
    size_t k = 0;
    for (size_t i = 0; i < p.size(); ++i)
    {
        while (k >= 2 && cross2(h[k - 2], h[k - 1], p[i]) <= 0) k--;
        h[k++] = p[i];
    }
    for (size_t i = p.size() - 1, t = k + 1; i > 0; --i)
    {
        while (k >= t && cross2(h[k - 2], h[k - 1], p[i - 1]) <= 0) k--;
        h[k++] = p[i - 1];
    }
    h.resize(k - 1);
    std::reverse(h.begin() + 1, h.end()); // clockwise like boost's default polygon, still starting at the leftmost point
    return h;
}

// Polygon::compute_concave_hull (polygon.cpp:283-318) over ::polygon::compute_concave_hull (concave_fitting.cpp:69-90, the
// overload that takes a non-const vector: RemoveDuplicates of the other one does NOT run on this path) plus the repair the
// constructor applies to its result (polygon.cpp:195-226 with third_party/correct_boost_polygon.hpp:188-195, :172-186): the
// walk leaves a counter-clockwise ring that starts -- and, when it closed, ends -- at the lowest point; Boost wants it closed
// and clockwise.  Returned open (no repeated vertex), clockwise, starting at the walk's start point; empty when no rung of
// the ladder has a hull.
std::vector<vector2> Polygon::compute_concave_hull(const std::vector<vector2>& pts) noexcept
{
    std::vector<vector2> hull;
    if (pts.size() < 3)
        return hull;
    // k = 3, then the prime ladder, at most 8 attempts (:72-88)
    static const size_t ladder[] = {3, 3, 5, 7, 11, 13, 17, 21};
    const size_t first = find_min_y_point(pts);
    std::vector<size_t> h;
    bool ok = false;
    for (size_t a = 0; a < 8 && !ok; ++a)
    {
        ok = concave_hull_k(pts, first, ladder[a], h);
        if (!ok && a + 1 < 8 && ladder[a + 1] > pts.size())
            break; // (:86-87: the next k exceeds the point count)
    }
    if (!ok)
        return hull;
    if (h.size() > 1 && points_equal(pts[h.back()], pts[h.front()]))
        h.pop_back(); // the closing vertex
    hull.reserve(h.size());
    for (size_t i : h)
        hull.push_back(pts[i]);
    if (hull.size() >= 3 && ring_area_signed(hull) > 0)
        std::reverse(hull.begin() + 1, hull.end()); // clockwise, still starting at the walk's start point
    return hull;
}

Polygon::Polygon(const std::vector<vector3>& points, const vector3& normal, const vector3& center) : _center(center)
{
    if (!double_equal(norm(normal), 1.0, 1e-9))
        throw std::invalid_argument("Polygon: normal norm should be 1");
    if (points.size() < 3)
        throw std::invalid_argument("Polygon: need at least 3 points to fit a polygon");
    const auto axes = get_plane_coordinate_system(normal);
    _xAxis = axes.first;
    _yAxis = axes.second;
    std::vector<vector2> projected;
    projected.reserve(points.size());
    for (auto it = points.rbegin(); it != points.rend(); ++it) // the reference projects in reverse order
        projected.push_back(get_projected_plan_coordinates(*it, _center, _xAxis, _yAxis));
    _ring = compute_concave_hull(projected);
    // A hull that crosses itself is cut apart at its crossings like the reference's repair does (dissolve_crossings); one that
    // merely touches itself -- the reference re-unites those pieces with Boost set operations -- goes the way of a failed hull:
    // the convex hull (polygon.cpp:200-207)
    if (!is_valid() && !(_ring.size() >= 4 && dissolve_crossings(_ring) && is_valid()))
        _ring = compute_convex_hull(projected);
    _area = area();
    simplify();
}

Polygon::Polygon(const std::vector<vector2>& ring, const vector3& xAxis, const vector3& yAxis, const vector3& center) :
    _ring(ring),
    _center(center),
    _xAxis(xAxis),
    _yAxis(yAxis)
{
    if (_ring.size() > 1 && _ring.front() == _ring.back())
        _ring.pop_back();
    if (ring_area_signed(_ring) > 0) // boost::geometry::correct: clockwise outer ring
        std::reverse(_ring.begin(), _ring.end());
    _area = area();
}

Polygon::Polygon(OpenRing, const std::vector<vector2>& ring, const vector3& xAxis, const vector3& yAxis, const vector3& center) :
    _ring(ring),
    _center(center),
    _xAxis(xAxis),
    _yAxis(yAxis)
{
    if (ring_area_signed(_ring) > 0)
        std::reverse(_ring.begin(), _ring.end());
    _area = area();
}

Polygon Polygon::project(const vector3& nextNormal, const vector3& nextCenter) const
{
    const auto axes = get_plane_coordinate_system(nextNormal);
    return project(axes.first, axes.second, nextCenter);
}

Polygon Polygon::project(const vector3& nextXAxis, const vector3& nextYAxis, const vector3& nextCenter) const
{
    std::vector<vector2> ring;
    ring.reserve(_ring.size());
    for (const vector2& p : _ring)
        ring.push_back(get_projected_plan_coordinates(get_point_from_plane_coordinates(p, _center, _xAxis, _yAxis), nextCenter,
                                                      nextXAxis, nextYAxis));
    Polygon out(OpenRing {}, ring, nextXAxis, nextYAxis, nextCenter);
    for (const std::vector<vector2>& hole : _inners)
    {
        std::vector<vector2> h;
        h.reserve(hole.size());
        for (const vector2& p : hole)
            h.push_back(get_projected_plan_coordinates(get_point_from_plane_coordinates(p, _center, _xAxis, _yAxis), nextCenter,
                                                       nextXAxis, nextYAxis));
        out.add_hole(h);
    }
    return out;
}

// interior ring, stored counter-clockwise; the cached area follows
void Polygon::add_hole(std::vector<vector2> hole)
{
    if (hole.size() < 3)
        return;
    if (ring_area_signed(hole) < 0)
        std::reverse(hole.begin(), hole.end());
    _inners.push_back(std::move(hole));
    _area = area();
}

namespace {
// area of (A minus its holes) n (B minus its holes): the holes lie inside their outer rings, so inclusion-exclusion over
// the rings is exact
double polygons_inter_area(const std::vector<vector2>& A, const std::vector<std::vector<vector2>>& holesA,
                           const std::vector<vector2>& B, const std::vector<std::vector<vector2>>& holesB)
{
    double a = rings_inter_area(A, B);
    for (const auto& h : holesA)
        a -= rings_inter_area(h, B);
    for (const auto& h : holesB)
        a -= rings_inter_area(A, h);
    for (const auto& ha : holesA)
        for (const auto& hb : holesB)
            a += rings_inter_area(ha, hb);
    return a < 0 ? 0.0 : a;
}
} // namespace

double Polygon::inter_area(const Polygon& other) const
{
    const Polygon o = other.project(_xAxis, _yAxis, _center);
    return polygons_inter_area(_ring, _inners, o._ring, o._inners);
}

double Polygon::union_area(const Polygon& other) const
{
    const Polygon o = other.project(_xAxis, _yAxis, _center);
    return area() + o.area() - polygons_inter_area(_ring, _inners, o._ring, o._inners);
}

double Polygon::inter_over_union(const Polygon& other) const
{
    const Polygon o = other.project(_xAxis, _yAxis, _center);
    const double inter = polygons_inter_area(_ring, _inners, o._ring, o._inners);
    const double uni = area() + o.area() - inter;
    return (uni <= 0 || inter <= 0) ? 0.0 : inter / uni;
}

bool Polygon::is_valid() const noexcept
{
    if (!ring_is_simple(_ring))
        return false;
    for (const auto& h : _inners)
        if (!ring_is_simple(h) || !point_in_ring(h.front(), _ring, true))
            return false;
    return true;
}

bool Polygon::is_valid(std::string& reason) const noexcept
{
    if (_ring.size() < 3)
        reason = "Geometry has too few points";
    else if (!ring_is_simple(_ring))
        reason = "Geometry has invalid self-intersections";
    else if (!is_valid())
        reason = "Geometry has interior rings defined outside the outer boundary";
    else
        reason = "Geometry is valid";
    return is_valid();
}

double Polygon::area() const noexcept
{
    if (_ring.size() < 3)
        return 0.0;
    double a = std::abs(ring_area_signed(_ring));
    for (const auto& h : _inners)
        a -= std::abs(ring_area_signed(h));
    return a < 0 ? 0.0 : a;
}

bool Polygon::contains(const vector2& point) const noexcept
{
    if (_ring.size() < 3 || !point_in_ring(point, _ring, false))
        return false;
    for (const auto& h : _inners)
        if (point_in_ring(point, h, true)) // in a hole or on its edge: not within
            return false;
    return true;
}

vector3 Polygon::get_normal() const noexcept { return cross(_xAxis, _yAxis); }

namespace {
inline vector3 apply44(const double* T, const vector3& p)
{
    return {((T[0] * p[0] + T[1] * p[1]) + T[2] * p[2]) + T[3], ((T[4] * p[0] + T[5] * p[1]) + T[6] * p[2]) + T[7],
            ((T[8] * p[0] + T[9] * p[1]) + T[10] * p[2]) + T[11]};
}
inline vector3 rotate44(const double* T, const vector3& p)
{
    return {(T[0] * p[0] + T[1] * p[1]) + T[2] * p[2], (T[4] * p[0] + T[5] * p[1]) + T[6] * p[2], (T[8] * p[0] + T[9] * p[1]) + T[10] * p[2]};
}
} // namespace

void plane_to_camera(const double* normal, double d, const double* T, double* normalOut, double* dOut)
{
    const vector3 n {normal[0], normal[1], normal[2]};
    const vector3 rn = rotate44(T, n);
    const vector3 nn = normalized(rn);
    // last row of the plane matrix: -t^T R
    const double t0 = T[3], t1 = T[7], t2 = T[11];
    const double m0 = -((t0 * T[0] + t1 * T[4]) + t2 * T[8]), m1 = -((t0 * T[1] + t1 * T[5]) + t2 * T[9]),
                 m2 = -((t0 * T[2] + t1 * T[6]) + t2 * T[10]);
    normalOut[0] = nn[0];
    normalOut[1] = nn[1];
    normalOut[2] = nn[2];
    *dOut = ((m0 * n[0] + m1 * n[1]) + m2 * n[2]) + d;
}

Polygon Polygon::to_camera_space(const double* T) const
{
    const vector3 newCenter = apply44(T, _center);
    const vector3 newX = normalized(rotate44(T, _xAxis)), newY = normalized(rotate44(T, _yAxis));
    auto move_ring = [&](const std::vector<vector2>& in) {
        std::vector<vector2> ring;
        ring.reserve(in.size());
        for (const vector2& q : in)
            ring.push_back(get_projected_plan_coordinates(apply44(T, get_point_from_plane_coordinates(q, _center, _xAxis, _yAxis)), newCenter,
                                                          newX, newY));
        return ring;
    };
    Polygon out(OpenRing {}, move_ring(_ring), newX, newY, newCenter);
    for (const auto& h : _inners)
        out.add_hole(move_ring(h));
    return out;
}

Polygon Polygon::transform(const vector3& nextNormal, const vector3& nextCenter) const
{
    const auto axes = get_plane_coordinate_system(nextNormal);
    return transform(axes.first, axes.second, nextCenter);
}

Polygon Polygon::transform(const vector3& nextXAxis, const vector3& nextYAxis, const vector3& nextCenter) const
{
    auto cross = [](const vector3& a, const vector3& b) {
        return vector3 {a[1] * b[2] - a[2] * b[1], a[2] * b[0] - a[0] * b[2], a[0] * b[1] - a[1] * b[0]};
    };
    // R_from = [x y x^y] (orthonormal: inverse = transpose), R_to likewise: R = R_to R_from^T
    const vector3 zF = cross(_xAxis, _yAxis), zT = cross(nextXAxis, nextYAxis);
    const vector3 from[3] = {_xAxis, _yAxis, zF}, to[3] = {nextXAxis, nextYAxis, zT};
    auto move_ring = [&](const std::vector<vector2>& in) {
        std::vector<vector2> ring;
        ring.reserve(in.size());
        for (const vector2& q : in)
        {
            const vector3 p3 = get_point_from_plane_coordinates(q, _center, _xAxis, _yAxis);
            vector3 moved {nextCenter[0] - _center[0], nextCenter[1] - _center[1], nextCenter[2] - _center[2]};
            for (int k = 0; k < 3; ++k)
            {
                const double coord = from[k][0] * p3[0] + from[k][1] * p3[1] + from[k][2] * p3[2]; // (R_from^T p)_k
                for (int i = 0; i < 3; ++i)
                    moved[i] += to[k][i] * coord;
            }
            ring.push_back(get_projected_plan_coordinates(moved, nextCenter, nextXAxis, nextYAxis));
        }
        return ring;
    };
    Polygon out(OpenRing {}, move_ring(_ring), nextXAxis, nextYAxis, nextCenter);
    for (const auto& h : _inners)
        out.add_hole(move_ring(h));
    return out;
}

namespace {

// Outer boundary of the union of two simple rings (any orientation), counter-clockwise; empty if degenerate.
// holes (optional): the bounded faces of the arrangement that lie in neither ring, i.e. the regions the two outlines
// enclose without covering them (what boost::geometry::union_ returns as interior rings).
// `faceRule` selects which bounded faces of the arrangement of the two rings go to `holes`: 0 = inside neither operand (the holes of
// the union), 1 = inside A and outside B (the pieces of A \ B), 2 = inside both (the pieces of A n B).  Two faces that share an edge
// differ in their membership of that edge's ring, so every face is a whole piece.
std::vector<vector2> rings_union_outer(const std::vector<vector2>& A, const std::vector<vector2>& B,
                                       std::vector<std::vector<vector2>>* holes = nullptr, int faceRule = 0)
{
    double scale = 1.0;
    for (const auto* r : {&A, &B})
        for (const vector2& p : *r)
            scale = std::max(scale, std::max(std::abs(p[0]), std::abs(p[1])));
    const double eps = 1e-9 * scale;
    auto same = [&](const vector2& a, const vector2& b) { return std::abs(a[0] - b[0]) <= eps && std::abs(a[1] - b[1]) <= eps; };
    auto cross2 = [](const vector2& o, const vector2& a, const vector2& b) {
        return (a[0] - o[0]) * (b[1] - o[1]) - (a[1] - o[1]) * (b[0] - o[0]);
    };
    struct Seg
    {
        vector2 a, b;
        std::vector<double> cuts; // parameters in (0, 1) where the segment is split
    };
    std::vector<Seg> segs;
    for (const auto* r : {&A, &B})
        for (size_t i = 0; i < r->size(); ++i)
        {
            const vector2 &a = (*r)[i], &b = (*r)[(i + 1) % r->size()];
            if (!same(a, b))
                segs.push_back({a, b, {}});
        }
    const size_t nA = [&] {
        size_t n = 0;
        for (size_t i = 0; i < A.size(); ++i)
            n += !same(A[i], A[(i + 1) % A.size()]);
        return n;
    }();
    auto param_on = [&](const Seg& s, const vector2& p, double& t) {
        // is p on segment s (within eps)?  t = position along it
        const double dx = s.b[0] - s.a[0], dy = s.b[1] - s.a[1];
        const double len2 = dx * dx + dy * dy;
        const double c = cross2(s.a, s.b, p);
        if (std::abs(c) > eps * std::sqrt(len2))
            return false;
        t = ((p[0] - s.a[0]) * dx + (p[1] - s.a[1]) * dy) / len2;
        return t > 0 && t < 1 && !same(p, s.a) && !same(p, s.b);
    };
    for (size_t i = 0; i < nA; ++i)
        for (size_t j = nA; j < segs.size(); ++j)
        {
            Seg &s = segs[i], &u = segs[j];
            double t;
            // endpoints lying on the other segment (T junctions, collinear overlaps)
            if (param_on(s, u.a, t))
                s.cuts.push_back(t);
            if (param_on(s, u.b, t))
                s.cuts.push_back(t);
            if (param_on(u, s.a, t))
                u.cuts.push_back(t);
            if (param_on(u, s.b, t))
                u.cuts.push_back(t);
            // proper crossing
            const double d1 = cross2(u.a, u.b, s.a), d2 = cross2(u.a, u.b, s.b);
            const double d3 = cross2(s.a, s.b, u.a), d4 = cross2(s.a, s.b, u.b);
            if (((d1 > 0 && d2 < 0) || (d1 < 0 && d2 > 0)) && ((d3 > 0 && d4 < 0) || (d3 < 0 && d4 > 0)))
            {
                const double ts = d1 / (d1 - d2), tu = d3 / (d3 - d4);
                const vector2 x {s.a[0] + ts * (s.b[0] - s.a[0]), s.a[1] + ts * (s.b[1] - s.a[1])};
                if (!same(x, s.a) && !same(x, s.b))
                    s.cuts.push_back(ts);
                if (!same(x, u.a) && !same(x, u.b))
                    u.cuts.push_back(tu);
            }
        }
    // nodes and undirected edges of the arrangement
    std::vector<vector2> nodes;
    auto node_of = [&](const vector2& p) {
        for (size_t k = 0; k < nodes.size(); ++k)
            if (same(nodes[k], p))
                return k;
        nodes.push_back(p);
        return nodes.size() - 1;
    };
    std::vector<std::vector<size_t>> adj;
    auto link = [&](size_t a, size_t b) {
        if (a == b)
            return;
        if (adj.size() < nodes.size())
            adj.resize(nodes.size());
        if (std::find(adj[a].begin(), adj[a].end(), b) == adj[a].end())
        {
            adj[a].push_back(b);
            adj[b].push_back(a);
        }
    };
    for (Seg& s : segs)
    {
        std::sort(s.cuts.begin(), s.cuts.end());
        size_t prev = node_of(s.a);
        for (const double t : s.cuts)
        {
            const size_t cur = node_of({s.a[0] + t * (s.b[0] - s.a[0]), s.a[1] + t * (s.b[1] - s.a[1])});
            link(prev, cur);
            prev = cur;
        }
        link(prev, node_of(s.b));
    }
    if (nodes.size() < 3)
        return {};
    adj.resize(nodes.size());
    // walk the outer face counter-clockwise from the lowest of the leftmost nodes, always taking the sharpest right turn
    size_t start = 0;
    for (size_t k = 1; k < nodes.size(); ++k)
        if (nodes[k][0] < nodes[start][0] - eps || (std::abs(nodes[k][0] - nodes[start][0]) <= eps && nodes[k][1] < nodes[start][1]))
            start = k;
    auto next_of = [&](size_t v, const vector2& back) -> size_t {
        // first neighbour met when rotating counter-clockwise from direction `back` (the way we came from)
        const double ba = std::atan2(back[1], back[0]);
        size_t best = v;
        double bestAngle = 1e300;
        for (const size_t w : adj[v])
        {
            double ang = std::atan2(nodes[w][1] - nodes[v][1], nodes[w][0] - nodes[v][0]) - ba;
            while (ang <= 1e-12)
                ang += 2 * M_PI; // going straight back is the last resort (angle 2 pi)
            if (ang < bestAngle)
            {
                bestAngle = ang;
                best = w;
            }
        }
        return best;
    };
    // the face that the directed edge (from -> to) has on its right, walked until it closes; empty if it does not
    auto walk_face = [&](size_t from, size_t to, std::vector<std::pair<size_t, size_t>>* edges) {
        std::vector<vector2> ring;
        size_t cur = from, nxt = to;
        for (size_t guard = 0; guard < 4 * nodes.size() + 8; ++guard)
        {
            ring.push_back(nodes[cur]);
            if (edges)
                edges->emplace_back(cur, nxt);
            const vector2 back {nodes[cur][0] - nodes[nxt][0], nodes[cur][1] - nodes[nxt][1]};
            const size_t after = next_of(nxt, back);
            cur = nxt;
            nxt = after;
            if (cur == from && nxt == to)
                return ring;
        }
        return std::vector<vector2> {}; // did not close: degenerate input
    };
    const size_t first = next_of(start, vector2 {0.0, 1.0}); // we reach the leftmost node heading south
    if (first == start)
        return {};
    std::vector<std::pair<size_t, size_t>> outerEdges;
    const std::vector<vector2> ring = walk_face(start, first, &outerEdges);
    if (holes && !ring.empty())
    {
        // every other face of the arrangement: bounded, walked clockwise; a hole of the union is one whose inside belongs to
        // neither operand
        std::vector<std::pair<size_t, size_t>> seen = outerEdges;
        auto was_seen = [&](size_t a, size_t b) { return std::find(seen.begin(), seen.end(), std::make_pair(a, b)) != seen.end(); };
        for (size_t a = 0; a < nodes.size(); ++a)
            for (const size_t b : adj[a])
            {
                if (was_seen(a, b))
                    continue;
                std::vector<std::pair<size_t, size_t>> faceEdges;
                std::vector<vector2> face = walk_face(a, b, &faceEdges);
                seen.insert(seen.end(), faceEdges.begin(), faceEdges.end());
                if (face.size() < 3 || ring_area_signed(face) >= 0) // not a bounded face (or a degenerate spur)
                    continue;
                // a point strictly inside the face: just right of the middle of one of its edges
                bool found = false;
                vector2 probe {0, 0};
                for (size_t i = 0; i < face.size() && !found; ++i)
                {
                    const vector2 &p = face[i], &q = face[(i + 1) % face.size()];
                    const double dx = q[0] - p[0], dy = q[1] - p[1], len = std::hypot(dx, dy);
                    if (len <= eps)
                        continue;
                    for (double off = 1e-3; off >= 1e-7 && !found; off *= 0.1)
                    {
                        probe = {0.5 * (p[0] + q[0]) + off * len * (dy / len), 0.5 * (p[1] + q[1]) - off * len * (dx / len)};
                        found = point_in_ring(probe, face, false);
                    }
                }
                if (!found)
                    continue;
                const bool inA = point_in_ring(probe, A, true), inB = point_in_ring(probe, B, true);
                if ((faceRule == 0 && !inA && !inB) || (faceRule == 1 && inA && !inB) || (faceRule == 2 && inA && inB))
                    holes->push_back(face);
            }
    }
    return ring;
}

// drop vertices that lie on the segment joining their neighbours (repeated until stable)
void drop_collinear(std::vector<vector2>& r)
{
    bool changed = true;
    while (changed && r.size() > 3)
    {
        changed = false;
        for (size_t i = 0; i < r.size() && r.size() > 3; ++i)
        {
            const vector2 &a = r[(i + r.size() - 1) % r.size()], &b = r[i], &c = r[(i + 1) % r.size()];
            const double cr = (b[0] - a[0]) * (c[1] - a[1]) - (b[1] - a[1]) * (c[0] - a[0]);
            const double len = std::hypot(c[0] - a[0], c[1] - a[1]);
            if (std::abs(cr) <= 1e-9 * std::max(1.0, len * len))
            {
                r.erase(r.begin() + static_cast<long>(i));
                changed = true;
                --i;
            }
        }
    }
}

} // namespace

bool Polygon::merge_union(const Polygon& other)
{
    const Polygon o = other.project(_xAxis, _yAxis, _center);
    if (_ring.size() < 3 || o._ring.size() < 3)
        return false;
    std::vector<std::vector<vector2>> holes;
    std::vector<vector2> outer = rings_union_outer(_ring, o._ring, &holes);
    drop_collinear(outer);
    const double outerArea = outer.size() >= 3 ? std::abs(ring_area_signed(outer)) : 0.0;
    // two disjoint pieces: union_one keeps the biggest one of the multi-polygon (polygon.cpp:474-492)
    const double areaA = std::abs(ring_area_signed(_ring)), areaB = std::abs(ring_area_signed(o._ring));
    bool disjoint = false;
    if (outerArea + 1e-9 * std::max(areaA, areaB) < std::max(areaA, areaB))
    {
        disjoint = true;
        if (area() >= o.area())
        {
            // this polygon is the biggest piece of the multi-polygon: union_one returns it as is, and merge_union
            // still simplifies what it assigned (polygon.cpp:325-336)
            simplify();
            return true;
        }
        outer = o._ring;
    }
    if (outer.size() < 3 || !ring_is_simple(outer))
        return false; // "Merge of two polygons produces no overlaps, returning without merge operation"
    Polygon merged(outer, _xAxis, _yAxis, _center);
    if (disjoint)
    {
        for (const auto& h : o._inners)
            merged.add_hole(h);
    }
    else
    {
        for (auto& h : holes)
        {
            drop_collinear(h);
            if (h.size() >= 3 && ring_is_simple(h))
                merged.add_hole(h);
        }
        // A hole an operand already had (boost::geometry::union_ of polygons with interior rings, polygon.cpp:463-470): what the
        // other operand leaves uncovered of it stays a hole -- the whole of it if the other stays clear, nothing if the other
        // covers it, and otherwise the pieces of (hole \ other's outer ring) plus the pieces of (hole n other's own holes).
        // (Through round 5 a partly re-covered hole counted as filled.)
        const double tiny = 1e-12 * std::max(areaA, areaB);
        // (the uncovered set is the disjoint union of  holeA \ outerB,  holeA n holeB  and  holeB \ outerA: the middle term is
        // taken from this polygon's side only)
        auto carry_hole = [&](const std::vector<vector2>& h, const std::vector<vector2>& otherRing,
                              const std::vector<std::vector<vector2>>& otherHoles, const bool withOverlaps) {
            const double covered = polygons_inter_area(h, {}, otherRing, otherHoles);
            if (covered <= tiny)
            {
                merged.add_hole(h);
                return;
            }
            std::vector<std::vector<vector2>> pieces;
            if (rings_inter_area(h, otherRing) < std::abs(ring_area_signed(h)) - tiny)
                (void)rings_union_outer(h, otherRing, &pieces, 1); // hole \ other
            for (const auto& oh : otherHoles)
                if (withOverlaps && rings_inter_area(h, oh) > tiny)
                    (void)rings_union_outer(h, oh, &pieces, 2);    // hole n (a hole of the other)
            for (auto& piece : pieces)
            {
                drop_collinear(piece);
                if (piece.size() >= 3 && ring_is_simple(piece) && std::abs(ring_area_signed(piece)) > tiny)
                    merged.add_hole(piece);
            }
        };
        for (const auto& h : _inners)
            carry_hole(h, o._ring, o._inners, true);
        for (const auto& h : o._inners)
            carry_hole(h, _ring, _inners, false);
    }
    *this = merged;
    simplify();
    return true;
}

std::vector<vector3> Polygon::get_unprojected_boundary() const
{
    std::vector<vector3> out;
    out.reserve(_ring.size());
    for (const vector2& p : _ring)
        out.push_back(get_point_from_plane_coordinates(p, _center, _xAxis, _yAxis));
    return out;
}

void Polygon::simplify(const double distanceThreshold) noexcept
{
    _area = area(); // total area: outer ring minus the interior rings
    if (_ring.size() < 4)
        return;
    const double eps = std::max(_area / 1e5, distanceThreshold);
    // Douglas-Peucker on a closed ring: opened at vertex 0 and closed back onto it
    auto simplified = [eps](const std::vector<vector2>& ring) {
        std::vector<vector2> closed = ring;
        closed.push_back(ring.front());
        std::vector<char> keep(closed.size(), 0);
        keep.front() = keep.back() = 1;
        douglas_peucker(closed, 0, closed.size() - 1, eps * eps, keep);
        std::vector<vector2> out;
        for (size_t i = 0; i + 1 < closed.size(); ++i)
            if (keep[i])
                out.push_back(closed[i]);
        return out;
    };
    // the reference simplifies the WHOLE polygon into a temporary, keeps it only if that temporary is valid and its total
    // area (holes included) stays above 75 % of the old total area (polygon.cpp:577-598): all rings or none
    Polygon cand = *this;
    cand._ring = simplified(_ring);
    bool valid = cand._ring.size() >= 3 && ring_is_simple(cand._ring);
    for (size_t k = 0; valid && k < _inners.size(); ++k)
    {
        if (_inners[k].size() < 4)
            continue; // a triangle has nothing to drop
        cand._inners[k] = simplified(_inners[k]);
        valid = cand._inners[k].size() >= 3 && ring_is_simple(cand._inners[k]);
        for (size_t i = 0; valid && i < cand._inners[k].size(); ++i)
            valid = point_in_ring(cand._inners[k][i], cand._ring, true);
    }
    if (!valid)
        return; // "could not optimize polygon boundary": unchanged
    const double newArea = cand.area();
    if (newArea > _area * 0.75)
    {
        _ring = std::move(cand._ring);
        _inners = std::move(cand._inners);
        _area = newArea;
    }
}

} // namespace rgbd_slam::utils
