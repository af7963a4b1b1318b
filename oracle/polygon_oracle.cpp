// =====================================================================================================
//  POLYGON ORACLE  --  TEST INFRASTRUCTURE ONLY.  NOT PRODUCT CODE.
//
//  CPU restatement of the reference's boundary-polygon path ("next" rows N1 / N2 of SURVEY.md 8f), written from the
//  reference's sources and NOT from this repo's product classes (rgb-d-slam_amd/host/boundary_polygon.cpp and
//  csrc/cape_polygon.hip are what it CHECKS):
//     third_party/concave_fitting.cpp      Moreira-Santos k-nearest-neighbours concave hull, its k ladder, its
//                                          DBL_EPSILON comparisons, `-atan2` candidate ordering, Intersects,
//                                          PointInPolygon (zero-crossings quirk included)
//     src/utils/polygon.cpp                plane frame, projection, Polygon(points, normal, center), explicit-ring
//                                          constructor, project, transform, area, contains, inter_area, union_area,
//                                          simplify
//     third_party/correct_boost_polygon.hpp the repair step of the constructor (close + orient; see `repair_ring`)
//     src/coordinates/polygon_coordinates.cpp:135-165  WorldPolygon::to_camera_space
//     src/coordinates/plane_coordinates.cpp:20-24 + src/utils/camera_transformation.cpp:53-71  plane through a pose
//  Only tests/, bench.py's checker legs and profiles/ report scripts may load this library.
//
//  PARITY STATUS: pinned by the reference's own polygon tests -- tests/test_polygons.cpp:6-89 (SquareTests.SimpleFitting:
//  boundary length, area, containment, inter / union area with itself and with the flipped polygon, project and transform
//  results) is replayed on this library by tests/test_polygon_oracle.py; so are -- round 5 -- the reference's
//  CoordinateSystemChangeTests (tests/test_coordinate_systems.cpp:23-160: the transformation matrix Polygon::transform moves its
//  vertices with) and PlaneCoordinateSystemTests (:700-793: a plane through compute_plane_camera_to_world_matrix and back through
//  compute_plane_world_to_camera_matrix, i.e. the reference's two 4x4 inversions, which polyref_plane_to_camera now follows
//  operation by operation; the closed form of round 4 stays as polyref_plane_to_camera_analytic, the variant).  How much the
//  unpinned choices below can matter is MEASURED by oracle/polygon_variants.py (POLY_VAR_* switches; profiles/r05_polygon_variants.txt:
//  no validity, fallback or match decision depends on any of them).  What stays UNPINNED (the libraries are absent
//  from this image and from /root/reference; their published algorithms are restated and each restatement says so):
//    * FLANN 1.9 `Index<L2<double>>(KDTreeIndexParams(4))::knnSearch(..., SearchParams(128))` -- four RANDOMIZED kd-trees,
//      an approximate search: the reference's hull is not reproducible run to run.  Restated as the EXACT k nearest
//      neighbours, ascending squared distance, ties by index (what the approximate search converges to).
//    * Boost.Geometry `is_valid`, `correct`, `area`, `within`, `simplify` (Douglas-Peucker), `intersection`, `convex_hull`:
//      restated from their documented behaviour.  The intersection area is computed with an algorithm that shares nothing
//      with the product's (boundary integration over the pieces of each outline that lie inside the other polygon; the
//      product cuts vertical slabs).
// =====================================================================================================
// ---- switches of oracle/polygon_variants.py: each flips ONE of the third-party behaviours this file could only restate from
//      documentation, so that the sweep can measure how much of the polygons / matches depends on it (default 0 = the restatement
//      described above)
#ifndef POLY_VAR_KNN
#define POLY_VAR_KNN 0      // 1: equal squared distances ordered by DESCENDING id (FLANN's order among ties is arbitrary)
#endif
#ifndef POLY_VAR_SIMPLIFY
#define POLY_VAR_SIMPLIFY 0 // 1: Douglas-Peucker distance to the carrier LINE instead of the segment; 2: `>=` instead of `>` against
#endif                      //    the threshold; 3: the ring is rotated to the vertex farthest from its first one before it is simplified
                            //    (what newer Boost releases do with rings)
#ifndef POLY_VAR_VALID
#define POLY_VAR_VALID 0    // 1: a ring that merely TOUCHES itself (vertex on another edge) counts as valid; only proper crossings do not
#endif
#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>

namespace {

using std::size_t;

// ---- third_party/concave_fitting.hpp:16-36 -----------------------------------------------------------------------
struct Point
{
    double x = 0.0, y = 0.0;
    uint64_t id = 0;
};
struct PointValue
{
    Point point;
    double distance = 0.0;
    double angle = 0.0;
};
using PointVector = std::vector<Point>;

// ---- third_party/concave_fitting.cpp:186-201 : every comparison of the hull carries a DBL_EPSILON slack -----------
bool Equal(double a, double b) { return std::fabs(a - b) <= DBL_EPSILON; }
bool Zero(double a) { return std::fabs(a) <= DBL_EPSILON; }
bool LessThan(double a, double b) { return a < (b - DBL_EPSILON); }
bool LessThanOrEqual(double a, double b) { return a <= (b + DBL_EPSILON); }
bool GreaterThan(double a, double b) { return a > (b + DBL_EPSILON); }
bool PointsEqual(const Point& a, const Point& b) { return Equal(a.x, b.x) && Equal(a.y, b.y); }

// concave_fitting.cpp:231-243
Point FindMinYPoint(const PointVector& points)
{
    auto itr = std::min_element(points.begin(), points.end(), [](const Point& a, const Point& b) {
        if (Equal(a.y, b.y))
            return GreaterThan(a.x, b.x);
        return LessThan(a.y, b.y);
    });
    return *itr;
}

// concave_fitting.cpp:326-332
double NormaliseAngle(double radians)
{
    if (radians < 0.0)
        return radians + M_PI + M_PI;
    return radians;
}
// concave_fitting.cpp:318-323 : clockwise angle from the +x axis
double Angle(const Point& a, const Point& b)
{
    const double angle = -std::atan2(b.y - a.y, b.x - a.x);
    return NormaliseAngle(angle);
}

// concave_fitting.cpp:296-315 (std::sort, not stable_sort, like the reference: the same libstdc++ introsort on the same
// input order gives the same permutation)
PointVector SortByAngle(std::vector<PointValue>& values, const Point& from, double prevAngle)
{
    for (PointValue& to : values)
        to.angle = NormaliseAngle(Angle(from, to.point) - prevAngle);
    std::sort(values.begin(), values.end(), [](const PointValue& a, const PointValue& b) { return GreaterThan(a.angle, b.angle); });
    PointVector angled(values.size());
    std::transform(values.begin(), values.end(), angled.begin(), [](const PointValue& pv) { return pv.point; });
    return angled;
}

// concave_fitting.cpp:426-463
bool Intersects(const Point& a1p, const Point& a2p, const Point& b1p, const Point& b2p)
{
    const double ax1 = a1p.x, ay1 = a1p.y, ax2 = a2p.x, ay2 = a2p.y;
    const double bx1 = b1p.x, by1 = b1p.y, bx2 = b2p.x, by2 = b2p.y;
    const double a1 = ay2 - ay1;
    const double b1 = ax1 - ax2;
    const double c1 = a1 * ax1 + b1 * ay1;
    const double a2 = by2 - by1;
    const double b2 = bx1 - bx2;
    const double c2 = a2 * bx1 + b2 * by1;
    const double det = a1 * b2 - a2 * b1;
    if (Zero(det))
        return false;
    const double x = (b2 * c1 - b1 * c2) / det;
    const double y = (a1 * c2 - a2 * c1) / det;
    bool on_both = true;
    on_both = on_both && LessThanOrEqual(std::min(ax1, ax2), x) && LessThanOrEqual(x, std::max(ax1, ax2));
    on_both = on_both && LessThanOrEqual(std::min(ay1, ay2), y) && LessThanOrEqual(y, std::max(ay1, ay2));
    on_both = on_both && LessThanOrEqual(std::min(bx1, bx2), x) && LessThanOrEqual(x, std::max(bx1, bx2));
    on_both = on_both && LessThanOrEqual(std::min(by1, by2), y) && LessThanOrEqual(y, std::max(by1, by2));
    return on_both;
}

// concave_fitting.cpp:393-423 -- including its quirk: a point whose ray crosses NO edge counts as inside
bool PointInPolygon(const Point& p, const PointVector& list)
{
    if (list.size() <= 2)
        return false;
    const double x = p.x, y = p.y;
    int inout = 0;
    size_t v0 = 0, v1 = 1;
    while (v1 != list.size())
    {
        const Point &q0 = list[v0], &q1 = list[v1];
        if (((LessThanOrEqual(q0.y, y) && LessThan(y, q1.y)) || (LessThanOrEqual(q1.y, y) && LessThan(y, q0.y))) && !Zero(q1.y - q0.y) &&
            LessThan(x, q0.x + ((q1.x - q0.x) * (y - q0.y) / (q1.y - q0.y))))
            inout++;
        v0 = v1;
        v1++;
    }
    if (inout == 0)
        return true;
    if (inout % 2 == 0)
        return false;
    return true;
}

// The FLANN index of concave_fitting.cpp:109-111 as the hull uses it: points 0..n-1 plus, from step 4 on, the first point
// again under id n (:128-134); removePoint(id) hides a point from later searches.  knnSearch (:258-288) restated as the
// EXACT k nearest visible points (flann::L2<double>: ((0 + dx*dx) + dy*dy)), ascending distance, ties by id.
struct NeighbourIndex
{
    std::vector<Point> pts;
    std::vector<char> removed;
    explicit NeighbourIndex(const PointVector& list) : pts(list), removed(list.size(), 0)
    {
        for (size_t i = 0; i < pts.size(); ++i)
            pts[i].id = i; // a FLANN id is the row of the matrix
    }
    void removePoint(uint64_t id)
    {
        if (id < removed.size())
            removed[id] = 1;
    }
    void addPoint(const Point& p)
    {
        pts.push_back(p);
        pts.back().id = pts.size() - 1;
        removed.push_back(0);
    }
    std::vector<PointValue> knn(const Point& q, size_t k) const
    {
        std::vector<PointValue> all;
        all.reserve(pts.size());
        for (size_t i = 0; i < pts.size(); ++i)
        {
            if (removed[i])
                continue;
            const double dx = q.x - pts[i].x, dy = q.y - pts[i].y;
            PointValue v;
            v.point = pts[i];
            v.distance = dx * dx + dy * dy;
            all.push_back(v);
        }
        const size_t kk = std::min(k, all.size());
        std::partial_sort(all.begin(), all.begin() + kk, all.end(), [](const PointValue& a, const PointValue& b) {
#if POLY_VAR_KNN == 1
            return a.distance < b.distance || (a.distance == b.distance && a.point.id > b.point.id);
#else
            return a.distance < b.distance || (a.distance == b.distance && a.point.id < b.point.id);
#endif
        });
        all.resize(kk);
        return all;
    }
};

// concave_fitting.cpp:93-183 -- one run of the walk for a given k
bool ConcaveHull(const PointVector& pointList, size_t k, PointVector& hull)
{
    hull.clear();
    if (pointList.size() < 3)
        return true;
    if (pointList.size() == 3)
    {
        hull = pointList;
        return true;
    }
    NeighbourIndex flannIndex(pointList);
    Point firstPoint = FindMinYPoint(pointList);
    hull.push_back(firstPoint);
    Point currentPoint = firstPoint;
    flannIndex.removePoint(firstPoint.id);
    double prevAngle = 0.0;
    int step = 1;
    while ((!PointsEqual(currentPoint, firstPoint) || step == 1) && hull.size() != pointList.size())
    {
        if (step == 4)
        {
            firstPoint.id = pointList.size();
            flannIndex.addPoint(firstPoint);
        }
        std::vector<PointValue> kNearestNeighbours = flannIndex.knn(currentPoint, k);
        PointVector cPoints = SortByAngle(kNearestNeighbours, currentPoint, prevAngle);
        bool its = true;
        size_t i = 0;
        while (its && i < cPoints.size())
        {
            i++;
            size_t lastPoint = 0;
            if (PointsEqual(cPoints[i - 1], firstPoint))
                lastPoint = 1;
            size_t j = 2;
            its = false;
            while (!its && j < hull.size() - lastPoint)
            {
                its = Intersects(hull[step - 1], cPoints[i - 1], hull[step - j - 1], hull[step - j]);
                j++;
            }
        }
        if (its)
            return false;
        if (i == 0)
            return false; // (no neighbour left: the reference would index cPoints[-1]; unreachable while hull.size() != n)
        currentPoint = cPoints[i - 1];
        hull.push_back(currentPoint);
        prevAngle = Angle(hull[step], hull[step - 1]);
        flannIndex.removePoint(currentPoint.id);
        step++;
    }
    // every point that is not on the hull must be enclosed (:176-182; RemoveHull by id, :335-349)
    std::vector<uint64_t> ids(hull.size());
    std::transform(hull.begin(), hull.end(), ids.begin(), [](const Point& p) { return p.id; });
    std::sort(ids.begin(), ids.end());
    for (const Point& p : pointList)
    {
        if (std::binary_search(ids.begin(), ids.end(), p.id))
            continue;
        if (!PointInPolygon(p, hull))
            return false;
    }
    return true;
}

// concave_fitting.cpp:69-90 (the overload that takes a non-const PointVector&: Polygon::compute_concave_hull passes one,
// polygon.cpp:303, so RemoveDuplicates of the const overload (:61-67) does NOT run)
bool compute_concave_hull(const PointVector& points, PointVector& hull, uint8_t maxIterations, int* kUsed)
{
    static const unsigned possible[] = {3, 5, 7, 11, 13, 17, 21, 23, 29, 31, 37, 41, 43, 47, 53, 59, 61, 67, 71, 73, 79, 83, 89, 97};
    const uint8_t trueMax = std::min<uint8_t>(maxIterations, 24);
    unsigned nearest = possible[0];
    for (uint8_t iteration = 0; iteration < trueMax; ++iteration)
    {
        hull.clear();
        if (ConcaveHull(points, nearest, hull))
        {
            if (kUsed)
                *kUsed = (int)nearest;
            return true;
        }
        nearest = possible[iteration];
        if (nearest > points.size())
            break;
    }
    return false;
}

// ---- 3-vectors the way Eigen evaluates them (fixed-size: (a0*b0 + a1*b1) + a2*b2; normalized() = v / sqrt(squaredNorm)) ----
struct V3
{
    double v[3];
    double operator[](int i) const { return v[i]; }
    double& operator[](int i) { return v[i]; }
};
double dot3(const V3& a, const V3& b) { return (a[0] * b[0] + a[1] * b[1]) + a[2] * b[2]; }
V3 cross3(const V3& a, const V3& b) { return {{a[1] * b[2] - a[2] * b[1], a[2] * b[0] - a[0] * b[2], a[0] * b[1] - a[1] * b[0]}}; }
double norm3(const V3& a) { return std::sqrt(dot3(a, a)); }
V3 normalized3(const V3& a)
{
    const double n2 = dot3(a, a);
    if (n2 > 0)
    {
        const double n = std::sqrt(n2);
        return {{a[0] / n, a[1] / n, a[2] / n}};
    }
    return a;
}
// src/utils/distance_utils.cpp:11 (default epsilon = numeric_limits<double>::epsilon(), distance_utils.hpp:24-26)
bool double_equal(double a, double b, double eps = DBL_EPSILON) { return std::fabs(a - b) <= eps; }

enum : int
{
    F_VALID = 1,           // boost::geometry::is_valid of the final polygon (what add_planes_to_primitives tests, primitive_detection.cpp:624)
    F_CONVEX_FALLBACK = 2, // polygon.cpp:196-207
    F_SIMPLIFIED = 4,      // simplify() replaced the ring (polygon.cpp:589-596)
    F_THREW = 16,          // the reference constructor throws (normal / axes not unit within DBL_EPSILON, < 3 points)
    F_NEEDS_DISSOLVE = 32, // the hull touches or crosses itself: the reference dissolves it with Boost set operations
                           // (correct_boost_polygon.hpp:229-330), which this oracle does not restate -- reported, never guessed
    F_HULL_FAILED = 64,    // no rung of the ladder produced a hull (concave_fitting.cpp:89)
    F_DISSOLVED = 128,     // the hull crossed itself and was cut apart at its crossings (dissolve_proper_crossings)
};

// polygon.cpp:50-68
V3 select_correct_transform(const V3& normal)
{
    const double distX = std::fabs(normal[0]), distY = std::fabs(normal[1]), distZ = std::fabs(normal[2]);
    const double res = std::min(distX, std::min(distY, distZ));
    if (double_equal(res, distX, 0.1))
        return {{1, 0, 0}};
    if (double_equal(res, distY, 0.1))
        return {{0, 1, 0}};
    if (double_equal(res, distZ, 0.1))
        return {{0, 0, 1}};
    return normalized3({{normal[2], normal[0], normal[1]}});
}

// polygon.cpp:74-115 ; false = the reference throws
bool get_plane_coordinate_system(const V3& normal, V3& xAxis, V3& yAxis)
{
    if (!double_equal(norm3(normal), 1.0))
        return false;
    const V3 r = select_correct_transform(normal);
    if (!double_equal(norm3(r), 1.0))
        return false;
    xAxis = normalized3(cross3(normal, r));
    yAxis = normalized3(cross3(normal, xAxis));
    if (!double_equal(norm3(xAxis), 1.0) || !double_equal(norm3(yAxis), 1.0))
        return false;
    if (std::fabs(dot3(xAxis, normal)) > .01 || std::fabs(dot3(yAxis, xAxis)) > .01 || std::fabs(dot3(yAxis, normal)) > .01)
        return false;
    return true;
}

struct P2
{
    double x, y;
};
using Ring = std::vector<P2>; // CLOSED ring (front == back), clockwise: boost::geometry::model::polygon<point_xy<double>> defaults

// polygon.cpp:125-144 / :146-166
bool frame_ok(const V3& xAxis, const V3& yAxis)
{
    return double_equal(norm3(xAxis), 1.0) && double_equal(norm3(yAxis), 1.0) && !(std::fabs(dot3(yAxis, xAxis)) > .01);
}
P2 get_projected_plan_coordinates(const V3& p, const V3& c, const V3& xAxis, const V3& yAxis)
{
    const V3 reduced {{p[0] - c[0], p[1] - c[1], p[2] - c[2]}};
    return {dot3(xAxis, reduced), dot3(yAxis, reduced)};
}
V3 get_point_from_plane_coordinates(const P2& p, const V3& c, const V3& xAxis, const V3& yAxis)
{
    // planeCenter + x * xAxis + y * yAxis, left to right
    return {{(c[0] + p.x * xAxis[0]) + p.y * yAxis[0], (c[1] + p.x * xAxis[1]) + p.y * yAxis[1], (c[2] + p.x * xAxis[2]) + p.y * yAxis[2]}};
}

// boost::geometry::area of a clockwise closed ring (strategy::area::cartesian: sum of (x1 + x2) * (y1 - y2), halved) -- UNPINNED
double ring_area(const Ring& r)
{
    if (r.size() < 3)
        return 0.0;
    double sum = 0.0;
    for (size_t i = 0; i + 1 < r.size(); ++i)
        sum += (r[i].x + r[i + 1].x) * (r[i].y - r[i + 1].y);
    return 0.5 * sum;
}

double orient(const P2& a, const P2& b, const P2& c)
{
    // sign of the turn a -> b -> c, evaluated twice with the roles swapped so that a rounding tie is reported as 0
    const double l = (b.x - a.x) * (c.y - a.y), r = (b.y - a.y) * (c.x - a.x);
    return l - r;
}
bool on_segment(const P2& a, const P2& b, const P2& p)
{
    return std::min(a.x, b.x) <= p.x && p.x <= std::max(a.x, b.x) && std::min(a.y, b.y) <= p.y && p.y <= std::max(a.y, b.y);
}
bool same(const P2& a, const P2& b) { return a.x == b.x && a.y == b.y; }

// do the closed segments (a1,a2) and (b1,b2) share a point?
bool segments_touch(const P2& a1, const P2& a2, const P2& b1, const P2& b2)
{
    const double d1 = orient(b1, b2, a1), d2 = orient(b1, b2, a2), d3 = orient(a1, a2, b1), d4 = orient(a1, a2, b2);
    if (((d1 > 0 && d2 < 0) || (d1 < 0 && d2 > 0)) && ((d3 > 0 && d4 < 0) || (d3 < 0 && d4 > 0)))
        return true;
    if (d1 == 0 && on_segment(b1, b2, a1))
        return true;
    if (d2 == 0 && on_segment(b1, b2, a2))
        return true;
    if (d3 == 0 && on_segment(a1, a2, b1))
        return true;
    if (d4 == 0 && on_segment(a1, a2, b2))
        return true;
    return false;
}

// A closed ring is "simple" when no two non-adjacent edges share a point and adjacent edges share only their common vertex
// (no spike).  This is what boost::geometry::is_valid demands of a ring besides closure, size and orientation -- UNPINNED.
bool ring_is_simple(const Ring& r)
{
    const size_t n = r.size() - 1; // edges
    if (r.size() < 4)
        return false;
    for (size_t i = 0; i < n; ++i)
    {
        if (same(r[i], r[i + 1]))
            continue; // boost tolerates consecutive duplicates
        for (size_t j = i + 1; j < n; ++j)
        {
            if (same(r[j], r[j + 1]))
                continue;
            const bool adjacent = (j == i + 1) || (i == 0 && j == n - 1);
            if (adjacent)
            {
                // a spike: the two edges are collinear and fold back onto each other
                const P2 &a = (j == i + 1) ? r[i] : r[j], &b = (j == i + 1) ? r[i + 1] : r[0], &c = (j == i + 1) ? r[j + 1] : r[1];
                if (orient(a, b, c) == 0 && ((b.x - a.x) * (c.x - b.x) + (b.y - a.y) * (c.y - b.y)) < 0)
                    return false;
                continue;
            }
#if POLY_VAR_VALID == 1
            {
                const double d1 = orient(r[j], r[j + 1], r[i]), d2 = orient(r[j], r[j + 1], r[i + 1]), d3 = orient(r[i], r[i + 1], r[j]),
                             d4 = orient(r[i], r[i + 1], r[j + 1]);
                if (((d1 > 0 && d2 < 0) || (d1 < 0 && d2 > 0)) && ((d3 > 0 && d4 < 0) || (d3 < 0 && d4 > 0)))
                    return false;
            }
#else
            if (segments_touch(r[i], r[i + 1], r[j], r[j + 1]))
                return false;
#endif
        }
    }
    return true;
}

// boost::geometry::is_valid(polygon) for an outer ring without holes -- UNPINNED restatement: closed, >= 4 points, simple,
// clockwise with a non-zero area.
bool polygon_is_valid(const Ring& r)
{
    if (r.size() < 4 || !same(r.front(), r.back()))
        return false;
    if (!ring_is_simple(r))
        return false;
    return ring_area(r) > 0;
}

// geometry::correct of third_party/correct_boost_polygon.hpp:547-558 for an outer ring WITHOUT self-intersections:
// impl::correct(ring, clockwise) = close (:188-195), reverse if its area is negative (:172-186), no self turns (:127-160)
// -> the ring itself if |area| > 0 (:349-356), else nothing.  needsDissolve: the ring touches or crosses itself, the
// reference then traces sub-rings and unions them with Boost -- proper crossings are restated below; a mere touch is not, and
// tests/test_polygon_oracle.py::test_hull_that_touches_itself_is_flagged_and_measured forces one and measures what the product's
// convex fallback costs there.
// dissolve (correct_boost_polygon.hpp:127-160, :199-330) for PROPER crossings: the crossing point becomes a pseudo-vertex of both
// edges (:146-157), the trace from a start key follows the ring up to the crossing, takes the by-pass to the other edge and
// runs on (:286-300): the ring comes apart into the part that runs on past the crossing and the loop it cuts off.  The
// constructor keeps result[0] (polygon.cpp:209-211) of pieces ordered by decreasing |area| (fill_non_zero_winding :371-375; a
// loop cut off by a crossing winds the other way and lies outside the rest, so it is not subtracted from it: covered_by :383-392
// fails) -- UNPINNED where it leans on Boost (the turn point's coordinates, the order of the union's output): restated as "cut at
// the first crossing in edge order, keep the piece of greater |area|, orient it clockwise (:358-369), repeat".  A ring that merely
// touches itself needs Boost's union of the traced pieces: not restated.  Only a touch along COLLINEAR points can come out of the
// walk (a vertex on another edge makes Intersects reject the candidate: same test); flagged F_NEEDS_DISSOLVE, pinned by that test.
bool first_contact(const Ring& r, size_t& ci, size_t& cj, bool& proper)
{
    const size_t n = r.size() - 1; // closed ring: n edges
    for (size_t i = 0; i < n; ++i)
        for (size_t j = i + 1; j < n; ++j)
        {
            if (j == i + 1 || (i == 0 && j == n - 1))
                continue;
            if (same(r[i], r[i + 1]) || same(r[j], r[j + 1]))
                continue;
            if (!segments_touch(r[i], r[i + 1], r[j], r[j + 1]))
                continue;
            const double d1 = orient(r[j], r[j + 1], r[i]), d2 = orient(r[j], r[j + 1], r[i + 1]), d3 = orient(r[i], r[i + 1], r[j]),
                         d4 = orient(r[i], r[i + 1], r[j + 1]);
            proper = ((d1 > 0 && d2 < 0) || (d1 < 0 && d2 > 0)) && ((d3 > 0 && d4 < 0) || (d3 < 0 && d4 > 0));
            ci = i;
            cj = j;
            return true;
        }
    return false;
}
bool dissolve_proper_crossings(Ring& ring) // closed, clockwise in; closed, clockwise, free of crossings out
{
    for (int cut = 0; cut < 8; ++cut)
    {
        size_t i = 0, j = 0;
        bool proper = false;
        if (!first_contact(ring, i, j, proper))
            return true;
        if (!proper)
            return false;
        const P2 a = ring[i], b = ring[i + 1], c = ring[j], d = ring[j + 1];
        const double den = (b.x - a.x) * (d.y - c.y) - (b.y - a.y) * (d.x - c.x);
        const double t = ((c.x - a.x) * (d.y - c.y) - (c.y - a.y) * (d.x - c.x)) / den;
        const P2 x {a.x + t * (b.x - a.x), a.y + t * (b.y - a.y)};
        Ring outer(ring.begin(), ring.begin() + (long)i + 1), loop;
        outer.push_back(x);
        outer.insert(outer.end(), ring.begin() + (long)j + 1, ring.end()); // ... up to the closing vertex = ring[0]
        loop.push_back(x);
        loop.insert(loop.end(), ring.begin() + (long)i + 1, ring.begin() + (long)j + 1);
        loop.push_back(x);
        ring = std::fabs(ring_area(loop)) > std::fabs(ring_area(outer)) ? loop : outer;
        if (ring_area(ring) < 0)
        {
            // keep the start vertex: open, reverse the rest, close again
            ring.pop_back();
            std::reverse(ring.begin() + 1, ring.end());
            ring.push_back(ring.front());
        }
    }
    return false;
}

bool repair_ring(Ring ring, Ring& out, bool& needsDissolve, bool* dissolved = nullptr)
{
    needsDissolve = false;
    out.clear();
    if (ring.size() < 3)
        return false;
    if (!same(ring.front(), ring.back()))
        ring.push_back(ring.front());
    if (ring_area(ring) < 0)
    {
        // (reversed around its first vertex: a closed ring reversed as a whole starts at the same vertex again)
        std::reverse(ring.begin(), ring.end());
    }
    if (!ring_is_simple(ring))
    {
        if (ring.size() >= 5 && dissolve_proper_crossings(ring) && ring_is_simple(ring))
        {
            if (dissolved)
                *dissolved = true;
        }
        else
        {
            needsDissolve = true;
            return false;
        }
    }
    if (!(std::fabs(ring_area(ring)) > 0.0))
        return false;
    out = ring;
    return true;
}

// boost::geometry::convex_hull of a multi_point into a clockwise closed polygon (polygon.cpp:268-281) -- UNPINNED:
// Andrew's monotone chain, collinear points dropped.
Ring compute_convex_hull(const std::vector<P2>& in)
{
    std::vector<P2> p = in;
    std::sort(p.begin(), p.end(), [](const P2& a, const P2& b) { return a.x < b.x || (a.x == b.x && a.y < b.y); });
    p.erase(std::unique(p.begin(), p.end(), [](const P2& a, const P2& b) { return same(a, b); }), p.end());
    Ring out;
    if (p.size() < 3)
    {
        out = p;
        if (!out.empty())
            out.push_back(out.front());
        return out;
    }
    std::vector<P2> h(2 * p.size());
    size_t k = 0;
    for (size_t i = 0; i < p.size(); ++i)
    {
        while (k >= 2 && orient(h[k - 2], h[k - 1], p[i]) <= 0)
            k--;
        h[k++] = p[i];
    }
    for (size_t i = p.size() - 1, t = k + 1; i > 0; --i)
    {
        while (k >= t && orient(h[k - 2], h[k - 1], p[i - 1]) <= 0)
            k--;
        h[k++] = p[i - 1];
    }
    h.resize(k); // counter-clockwise, closed (last == first)
    std::reverse(h.begin(), h.end());
    return h;
}

// boost::geometry::simplify (strategy::simplify::douglas_peucker over strategy::distance::projected_point: the distance of
// a point to the SEGMENT between the two kept ends, compared squared) applied to the closed ring as a range from its
// first to its last point -- UNPINNED; newer Boost releases first rotate a ring to another start vertex, so the simplified
// VERTICES are not a parity target, the simplified AREA (within the threshold's reach) and validity are.
double seg_dist2(const P2& p, const P2& a, const P2& b)
{
    const double vx = b.x - a.x, vy = b.y - a.y, wx = p.x - a.x, wy = p.y - a.y;
    const double c1 = wx * vx + wy * vy;
#if POLY_VAR_SIMPLIFY == 1
    const double c2line = vx * vx + vy * vy;
    if (c2line > 0)
    {
        const double cr = vx * wy - vy * wx;
        return cr * cr / c2line;
    }
#endif
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
void douglas_peucker(const Ring& in, size_t a, size_t b, double maxDist2, std::vector<char>& keep)
{
    if (b <= a + 1)
        return;
    double best = -1.0;
    size_t idx = a;
    for (size_t i = a + 1; i < b; ++i)
    {
        const double d = seg_dist2(in[i], in[a], in[b]);
        if (d > best)
        {
            best = d;
            idx = i;
        }
    }
#if POLY_VAR_SIMPLIFY == 2
    if (best >= maxDist2)
#else
    if (best > maxDist2)
#endif
    {
        keep[idx] = 1;
        douglas_peucker(in, a, idx, maxDist2, keep);
        douglas_peucker(in, idx, b, maxDist2, keep);
    }
}
Ring simplify_ring(const Ring& ringIn, double maxDist)
{
    if (ringIn.size() <= 4) // core_detail::closure::minimum_ring_size<closed> = 4: nothing to drop
        return ringIn;
#if POLY_VAR_SIMPLIFY == 3
    // open the ring at the vertex farthest from its first one
    Ring ring;
    {
        const size_t n = ringIn.size() - 1;
        size_t far = 0;
        double best = -1.0;
        for (size_t i = 0; i < n; ++i)
        {
            const double dx = ringIn[i].x - ringIn[0].x, dy = ringIn[i].y - ringIn[0].y;
            if (dx * dx + dy * dy > best)
                best = dx * dx + dy * dy, far = i;
        }
        for (size_t i = 0; i <= n; ++i)
            ring.push_back(ringIn[(far + i) % n]);
    }
#else
    const Ring& ring = ringIn;
#endif
    std::vector<char> keep(ring.size(), 0);
    keep.front() = keep.back() = 1;
    douglas_peucker(ring, 0, ring.size() - 1, maxDist * maxDist, keep);
    Ring out;
    for (size_t i = 0; i < ring.size(); ++i)
        if (keep[i])
            out.push_back(ring[i]);
    return out;
}

struct Poly
{
    Ring ring; // closed, clockwise
    V3 center {{0, 0, 0}}, xAxis {{1, 0, 0}}, yAxis {{0, 1, 0}};
    double area = 0.0;
    int flags = 0;
    int kUsed = 0;
};

// Polygon::simplify, polygon.cpp:578-601
void polygon_simplify(Poly& p, double distanceThreshold = 10)
{
    p.area = ring_area(p.ring);
    const double distanceThres = std::max(p.area / 1e5, distanceThreshold);
    const Ring out = simplify_ring(p.ring, distanceThres);
    if (polygon_is_valid(out))
    {
        const double newArea = ring_area(out);
        if (newArea > p.area * 0.75)
        {
            if (out.size() != p.ring.size())
                p.flags |= F_SIMPLIFIED;
            p.area = newArea;
            p.ring = out;
        }
    }
}

// Polygon::Polygon(points, normal, center), polygon.cpp:168-229
Poly polygon_from_points(const std::vector<V3>& points, const V3& normal, const V3& center)
{
    Poly P;
    P.center = center;
    if (!double_equal(norm3(normal), 1.0) || points.size() < 3 || !get_plane_coordinate_system(normal, P.xAxis, P.yAxis) || !frame_ok(P.xAxis, P.yAxis))
    {
        P.flags = F_THREW;
        return P;
    }
    // :187-192 -- projected in REVERSE order
    std::vector<P2> boundaryPoints;
    boundaryPoints.reserve(points.size());
    for (auto it = points.rbegin(); it != points.rend(); ++it)
        boundaryPoints.push_back(get_projected_plan_coordinates(*it, P.center, P.xAxis, P.yAxis));
    // compute_concave_hull, :283-318 (ids in input order, 8 rungs)
    PointVector pv(boundaryPoints.size());
    for (size_t i = 0; i < pv.size(); ++i)
    {
        pv[i].x = boundaryPoints[i].x;
        pv[i].y = boundaryPoints[i].y;
        pv[i].id = i;
    }
    PointVector hull;
    Ring poly;
    if (compute_concave_hull(pv, hull, 8, &P.kUsed))
        for (const Point& q : hull)
            poly.push_back({q.x, q.y});
    else
        P.flags |= F_HULL_FAILED;
    // :195-226
    if (!polygon_is_valid(poly))
    {
        Ring repaired;
        bool needsDissolve = false, dissolved = false;
        if (repair_ring(poly, repaired, needsDissolve, &dissolved))
        {
            poly = repaired;
            if (dissolved)
                P.flags |= F_DISSOLVED;
        }
        else if (needsDissolve)
        {
            P.flags |= F_NEEDS_DISSOLVE;
            P.ring = poly;
            return P;
        }
        else
        {
            poly = compute_convex_hull(boundaryPoints);
            P.flags |= F_CONVEX_FALLBACK;
        }
    }
    P.ring = poly;
    P.area = ring_area(P.ring); // :228 (Polygon::area, :453-461: 0 below 3 points)
    polygon_simplify(P);        // :231
    if (polygon_is_valid(P.ring))
        P.flags |= F_VALID;
    return P;
}

// Polygon::Polygon(boundaryPoints, xAxis, yAxis, center), polygon.cpp:236-266: assign_points + boost::geometry::correct
// (closes the ring, reverses it when its area is negative) -- UNPINNED for `correct`
Poly polygon_from_ring(const std::vector<P2>& boundary, const V3& xAxis, const V3& yAxis, const V3& center)
{
    Poly P;
    P.center = center;
    P.xAxis = xAxis;
    P.yAxis = yAxis;
    if (!frame_ok(xAxis, yAxis))
    {
        P.flags = F_THREW;
        return P;
    }
    P.ring = boundary;
    if (!P.ring.empty() && !same(P.ring.front(), P.ring.back()))
        P.ring.push_back(P.ring.front());
    if (ring_area(P.ring) < 0)
        std::reverse(P.ring.begin(), P.ring.end());
    P.area = ring_area(P.ring);
    if (polygon_is_valid(P.ring))
        P.flags |= F_VALID;
    return P;
}

// Eigen isApprox (fuzzy compare, precision 1e-12): |a - b|^2 <= prec^2 * min(|a|^2, |b|^2)
bool isApprox3(const V3& a, const V3& b)
{
    const V3 d {{a[0] - b[0], a[1] - b[1], a[2] - b[2]}};
    return dot3(d, d) <= 1e-12 * 1e-12 * std::min(dot3(a, a), dot3(b, b));
}

// Polygon::project, polygon.cpp:351-382
Poly polygon_project(const Poly& p, const V3& nextX, const V3& nextY, const V3& nextCenter)
{
    if (isApprox3(p.center, nextCenter) && isApprox3(p.xAxis, nextX) && isApprox3(p.yAxis, nextY))
        return p;
    std::vector<P2> nb;
    nb.reserve(p.ring.size());
    for (const P2& q : p.ring)
        nb.push_back(get_projected_plan_coordinates(get_point_from_plane_coordinates(q, p.center, p.xAxis, p.yAxis), nextCenter, nextX, nextY));
    return polygon_from_ring(nb, nextX, nextY, nextCenter);
}

// 4x4 affine applied to a point ((T * p.homogeneous()).head<3>(), Eigen's fixed-size product: each row a left-to-right sum)
V3 apply44(const double* T, const V3& p)
{
    V3 r;
    for (int i = 0; i < 3; ++i)
        r[i] = ((T[4 * i + 0] * p[0] + T[4 * i + 1] * p[1]) + T[4 * i + 2] * p[2]) + T[4 * i + 3] * 1.0;
    return r;
}
V3 rotate44(const double* T, const V3& p)
{
    V3 r;
    for (int i = 0; i < 3; ++i)
        r[i] = (T[4 * i + 0] * p[0] + T[4 * i + 1] * p[1]) + T[4 * i + 2] * p[2];
    return r;
}

// Polygon::transform_boundary, polygon.cpp:430-451, and WorldPolygon::to_camera_space, polygon_coordinates.cpp:135-165
Poly polygon_to_camera_space(const Poly& p, const double* worldToCamera)
{
    const V3 newCenter = apply44(worldToCamera, p.center);
    const V3 newX = normalized3(rotate44(worldToCamera, p.xAxis));
    const V3 newY = normalized3(rotate44(worldToCamera, p.yAxis));
    if (!frame_ok(newX, newY))
    {
        Poly bad;
        bad.flags = F_THREW;
        return bad;
    }
    std::vector<P2> nb;
    nb.reserve(p.ring.size());
    for (const P2& q : p.ring)
    {
        const V3 retro = get_point_from_plane_coordinates(q, p.center, p.xAxis, p.yAxis);
        nb.push_back(get_projected_plan_coordinates(apply44(worldToCamera, retro), newCenter, newX, newY));
    }
    return polygon_from_ring(nb, newX, newY, newCenter);
}

// Polygon::transform, polygon.cpp:384-428 with get_transformation_matrix, point_coordinates.cpp:24-70:
// linear part = R_to * R_from^-1 (both orthonormal bases [x y x^y]), translation = centerTo - centerFrom
Poly polygon_transform(const Poly& p, const V3& nextX, const V3& nextY, const V3& nextCenter)
{
    if (isApprox3(p.center, nextCenter) && isApprox3(p.xAxis, nextX) && isApprox3(p.yAxis, nextY))
        return p;
    const V3 zF = cross3(p.xAxis, p.yAxis), zT = cross3(nextX, nextY);
    const V3 from[3] = {p.xAxis, p.yAxis, zF}, to[3] = {nextX, nextY, zT};
    double T[16] = {0};
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j)
            T[4 * i + j] = (to[0][i] * from[0][j] + to[1][i] * from[1][j]) + to[2][i] * from[2][j]; // R_to * R_from^T
    for (int i = 0; i < 3; ++i)
        T[4 * i + 3] = nextCenter[i] - p.center[i];
    T[15] = 1.0;
    std::vector<P2> nb;
    for (const P2& q : p.ring)
    {
        const V3 retro = get_point_from_plane_coordinates(q, p.center, p.xAxis, p.yAxis);
        nb.push_back(get_projected_plan_coordinates(apply44(T, retro), nextCenter, nextX, nextY));
    }
    return polygon_from_ring(nb, nextX, nextY, nextCenter);
}

// boost::geometry::within(point, polygon): strictly inside -- UNPINNED
bool polygon_contains(const Ring& r, const P2& p)
{
    if (r.size() < 4)
        return false;
    bool inside = false;
    for (size_t i = 0; i + 1 < r.size(); ++i)
    {
        const P2 &a = r[i], &b = r[i + 1];
        if (orient(a, b, p) == 0 && on_segment(a, b, p))
            return false; // on the boundary
        if ((a.y > p.y) != (b.y > p.y))
        {
            const double xi = a.x + (b.x - a.x) * ((p.y - a.y) / (b.y - a.y));
            if (p.x < xi)
                inside = !inside;
        }
    }
    return inside;
}
// closed containment with a three-way answer: 1 inside, 0 on the boundary, -1 outside
int locate(const Ring& r, const P2& p, size_t* onEdge = nullptr)
{
    bool inside = false;
    for (size_t i = 0; i + 1 < r.size(); ++i)
    {
        const P2 &a = r[i], &b = r[i + 1];
        if (same(a, b))
            continue;
        if (orient(a, b, p) == 0 && on_segment(a, b, p))
        {
            if (onEdge)
                *onEdge = i;
            return 0;
        }
        if ((a.y > p.y) != (b.y > p.y))
        {
            const double xi = a.x + (b.x - a.x) * ((p.y - a.y) / (b.y - a.y));
            if (p.x < xi)
                inside = !inside;
        }
    }
    return inside ? 1 : -1;
}

// Area of the intersection of two simple clockwise rings (the sum of boost::geometry::area over the pieces of
// boost::geometry::intersection, polygon.cpp:525-545) by BOUNDARY INTEGRATION: the outline of A n B is made of the pieces of
// A's outline that lie inside B and the pieces of B's outline that lie inside A.  Every edge is cut at its crossings with
// the other ring; a piece is kept when its midpoint is strictly inside the other polygon, or -- for A's pieces only -- when
// it runs along an edge of B in the same direction (shared outline, interiors on the same side).  The area is the
// trapezoid sum of the kept pieces.  Shares nothing with the product's vertical-slab decomposition.
struct SharedStretch
{
    double lo, hi;
    bool sameDirection;
};
void cut_parameters(const P2& a, const P2& b, const Ring& other, std::vector<double>& ts, std::vector<SharedStretch>& shared)
{
    ts.clear();
    shared.clear();
    ts.push_back(0.0);
    ts.push_back(1.0);
    const double dx = b.x - a.x, dy = b.y - a.y;
    const double len2 = dx * dx + dy * dy;
    for (size_t j = 0; j + 1 < other.size(); ++j)
    {
        const P2 &c = other[j], &d = other[j + 1];
        if (same(c, d))
            continue;
        const double d1 = orient(c, d, a), d2 = orient(c, d, b);
        if (d1 == 0 && d2 == 0)
        {
            // the two edges lie on one line: the stretch they share, in this edge's parameter
            const double tc = ((c.x - a.x) * dx + (c.y - a.y) * dy) / len2, td = ((d.x - a.x) * dx + (d.y - a.y) * dy) / len2;
            const double lo = std::max(0.0, std::min(tc, td)), hi = std::min(1.0, std::max(tc, td));
            if (hi > lo)
            {
                shared.push_back({lo, hi, tc < td});
                if (lo > 0)
                    ts.push_back(lo);
                if (hi < 1)
                    ts.push_back(hi);
            }
            continue;
        }
        const double d3 = orient(a, b, c), d4 = orient(a, b, d);
        if (((d1 >= 0 && d2 <= 0) || (d1 <= 0 && d2 >= 0)) && ((d3 >= 0 && d4 <= 0) || (d3 <= 0 && d4 >= 0)))
        {
            const double t = d1 / (d1 - d2);
            if (t > 0 && t < 1)
                ts.push_back(t);
        }
    }
    std::sort(ts.begin(), ts.end());
    ts.erase(std::unique(ts.begin(), ts.end()), ts.end());
}
// crossing number of a point that is known not to lie on the outline
bool strictly_inside(const Ring& r, const P2& p)
{
    bool inside = false;
    for (size_t i = 0; i + 1 < r.size(); ++i)
    {
        const P2 &a = r[i], &b = r[i + 1];
        if ((a.y > p.y) != (b.y > p.y))
        {
            const double xi = a.x + (b.x - a.x) * ((p.y - a.y) / (b.y - a.y));
            if (p.x < xi)
                inside = !inside;
        }
    }
    return inside;
}
double boundary_inside(const Ring& A, const Ring& B, bool keepShared)
{
    double sum = 0.0;
    std::vector<double> ts;
    std::vector<SharedStretch> shared;
    for (size_t i = 0; i + 1 < A.size(); ++i)
    {
        const P2 &a = A[i], &b = A[i + 1];
        if (same(a, b))
            continue;
        cut_parameters(a, b, B, ts, shared);
        for (size_t s = 0; s + 1 < ts.size(); ++s)
        {
            const double tm = 0.5 * (ts[s] + ts[s + 1]);
            bool keep;
            const SharedStretch* on = nullptr;
            for (const SharedStretch& st : shared)
                if (tm > st.lo && tm < st.hi)
                    on = &st;
            if (on)
                keep = keepShared && on->sameDirection; // a shared stretch of outline counts once, when the interiors lie on the same side
            else
                keep = strictly_inside(B, {a.x + tm * (b.x - a.x), a.y + tm * (b.y - a.y)});
            if (!keep)
                continue;
            const P2 p {a.x + ts[s] * (b.x - a.x), a.y + ts[s] * (b.y - a.y)};
            const P2 q {a.x + ts[s + 1] * (b.x - a.x), a.y + ts[s + 1] * (b.y - a.y)};
            sum += (p.x + q.x) * (p.y - q.y);
        }
    }
    return sum;
}
double rings_inter_area(const Ring& A, const Ring& B)
{
    if (A.size() < 4 || B.size() < 4)
        return 0.0;
    const double s = boundary_inside(A, B, true) + boundary_inside(B, A, false);
    const double area = 0.5 * s;
    return area > 0 ? area : 0.0;
}

// Polygon::inter_area, polygon.cpp:525-545
double polygon_inter_area(const Poly& self, const Poly& other)
{
    const Poly o = polygon_project(other, self.xAxis, self.yAxis, self.center);
    return rings_inter_area(self.ring, o.ring);
}

// ---- marshalling ---------------------------------------------------------------------------------------------------
Poly poly_in(const double* ring, int n, const double* xAxis, const double* yAxis, const double* center)
{
    // ring: n open vertices (no closing duplicate), clockwise, as the product stores them
    Poly p;
    for (int i = 0; i < n; ++i)
        p.ring.push_back({ring[2 * i], ring[2 * i + 1]});
    if (!p.ring.empty() && !same(p.ring.front(), p.ring.back()))
        p.ring.push_back(p.ring.front());
    for (int k = 0; k < 3; ++k)
    {
        p.xAxis[k] = xAxis[k];
        p.yAxis[k] = yAxis[k];
        p.center[k] = center[k];
    }
    p.area = ring_area(p.ring);
    return p;
}
int poly_out(const Poly& p, double* ringOut, int cap, int* count, double* area, double* xAxis, double* yAxis, double* center, int* flags, int* kUsed)
{
    const int n = p.ring.empty() ? 0 : (int)p.ring.size() - (same(p.ring.front(), p.ring.back()) && p.ring.size() > 1 ? 1 : 0);
    if (count)
        *count = n;
    if (ringOut)
        for (int i = 0; i < n && i < cap; ++i)
        {
            ringOut[2 * i] = p.ring[(size_t)i].x;
            ringOut[2 * i + 1] = p.ring[(size_t)i].y;
        }
    if (area)
        *area = p.area;
    for (int k = 0; k < 3; ++k)
    {
        if (xAxis)
            xAxis[k] = p.xAxis[k];
        if (yAxis)
            yAxis[k] = p.yAxis[k];
        if (center)
            center[k] = p.center[k];
    }
    if (flags)
        *flags = p.flags;
    if (kUsed)
        *kUsed = p.kUsed;
    return n > cap ? -1 : 0;
}

} // namespace

extern "C" {

// Polygon(points, normal, center).  ring_out: open clockwise ring (the closing vertex is not repeated).
int polyref_build(const double* points3, int n, const double* normal, const double* center, double* ring_out, int cap, int* count, double* area,
                  double* x_axis, double* y_axis, int* flags, int* k_used)
{
    std::vector<V3> pts((size_t)n);
    for (int i = 0; i < n; ++i)
        pts[(size_t)i] = {{points3[3 * i], points3[3 * i + 1], points3[3 * i + 2]}};
    const Poly p = polygon_from_points(pts, {{normal[0], normal[1], normal[2]}}, {{center[0], center[1], center[2]}});
    return poly_out(p, ring_out, cap, count, area, x_axis, y_axis, nullptr, flags, k_used);
}

// The unsimplified hull of a 2-D point set for one k (ConcaveHull) or over the ladder (k = 0): 1 = success.  hull_out closed
// or not exactly as the walk left it.
int polyref_concave_hull(const double* xy, int n, int k, double* hull_out, int cap, int* count, int* k_used)
{
    PointVector pv((size_t)n);
    for (int i = 0; i < n; ++i)
    {
        pv[(size_t)i].x = xy[2 * i];
        pv[(size_t)i].y = xy[2 * i + 1];
        pv[(size_t)i].id = (uint64_t)i;
    }
    PointVector hull;
    int used = k;
    const bool ok = k > 0 ? ConcaveHull(pv, (size_t)k, hull) : compute_concave_hull(pv, hull, 8, &used);
    if (count)
        *count = (int)hull.size();
    if (k_used)
        *k_used = used;
    for (size_t i = 0; i < hull.size() && (int)i < cap; ++i)
    {
        hull_out[2 * i] = hull[i].x;
        hull_out[2 * i + 1] = hull[i].y;
    }
    return ok ? 1 : 0;
}

double polyref_area(const double* ring, int n)
{
    const double x[3] = {1, 0, 0}, y[3] = {0, 1, 0}, c[3] = {0, 0, 0};
    return poly_in(ring, n, x, y, c).area;
}
int polyref_is_valid(const double* ring, int n)
{
    const double x[3] = {1, 0, 0}, y[3] = {0, 1, 0}, c[3] = {0, 0, 0};
    return polygon_is_valid(poly_in(ring, n, x, y, c).ring) ? 1 : 0;
}
int polyref_contains(const double* ring, int n, double px, double py)
{
    const double x[3] = {1, 0, 0}, y[3] = {0, 1, 0}, c[3] = {0, 0, 0};
    return polygon_contains(poly_in(ring, n, x, y, c).ring, {px, py}) ? 1 : 0;
}
// closed containment: 1 inside, 0 on the outline, -1 outside
int polyref_locate(const double* ring, int n, double px, double py)
{
    const double x[3] = {1, 0, 0}, y[3] = {0, 1, 0}, c[3] = {0, 0, 0};
    return locate(poly_in(ring, n, x, y, c).ring, {px, py});
}

// 0 for a point inside or on the outline, else its distance to the outline (how far a simplified outline left a candidate out)
double polyref_distance_outside(const double* ring, int n, double px, double py)
{
    const double x[3] = {1, 0, 0}, y[3] = {0, 1, 0}, c[3] = {0, 0, 0};
    const Poly p = poly_in(ring, n, x, y, c);
    if (p.ring.size() < 2)
        return HUGE_VAL;
    if (locate(p.ring, {px, py}) >= 0)
        return 0.0;
    double best = HUGE_VAL;
    for (size_t i = 0; i + 1 < p.ring.size(); ++i)
        best = std::min(best, seg_dist2({px, py}, p.ring[i], p.ring[i + 1]));
    return std::sqrt(best);
}

// detected.inter_area(projected): `b` is projected into the frame of `a` first
double polyref_inter_area(const double* ring_a, int na, const double* xa, const double* ya, const double* ca, const double* ring_b, int nb,
                          const double* xb, const double* yb, const double* cb)
{
    return polygon_inter_area(poly_in(ring_a, na, xa, ya, ca), poly_in(ring_b, nb, xb, yb, cb));
}
// area of the intersection of two rings given in the SAME frame
double polyref_rings_inter_area(const double* ring_a, int na, const double* ring_b, int nb)
{
    const double x[3] = {1, 0, 0}, y[3] = {0, 1, 0}, c[3] = {0, 0, 0};
    Poly A = poly_in(ring_a, na, x, y, c), B = poly_in(ring_b, nb, x, y, c);
    if (ring_area(A.ring) < 0)
        std::reverse(A.ring.begin(), A.ring.end());
    if (ring_area(B.ring) < 0)
        std::reverse(B.ring.begin(), B.ring.end());
    return rings_inter_area(A.ring, B.ring);
}

// Polygon::project(nextNormal, nextCenter) (polygon.cpp:338-349) / project(xAxis, yAxis, center); transform likewise.
// mode 0: project on a normal, 1: project on explicit axes (next_a = x axis, next_b = y axis), 2: transform on a normal,
// 3: transform on explicit axes, 4: to_camera_space (next_a = 16 doubles, row-major worldToCamera)
int polyref_move(int mode, const double* ring, int n, const double* x, const double* y, const double* c, const double* next_a, const double* next_b,
                 const double* next_center, double* ring_out, int cap, int* count, double* area, double* x_out, double* y_out, double* c_out, int* flags)
{
    const Poly p = poly_in(ring, n, x, y, c);
    Poly out;
    if (mode == 4)
        out = polygon_to_camera_space(p, next_a);
    else
    {
        V3 nx, ny;
        if (mode == 0 || mode == 2)
        {
            if (!get_plane_coordinate_system({{next_a[0], next_a[1], next_a[2]}}, nx, ny))
            {
                if (flags)
                    *flags = F_THREW;
                return 0;
            }
        }
        else
        {
            nx = {{next_a[0], next_a[1], next_a[2]}};
            ny = {{next_b[0], next_b[1], next_b[2]}};
        }
        const V3 nc {{next_center[0], next_center[1], next_center[2]}};
        out = (mode <= 1) ? polygon_project(p, nx, ny, nc) : polygon_transform(p, nx, ny, nc);
    }
    return poly_out(out, ring_out, cap, count, area, x_out, y_out, c_out, flags, nullptr);
}

// ---- plane through a pose: plane_coordinates.cpp:15-24 over camera_transformation.cpp:19-71 --------------------------------
// The reference builds the 4x4 plane matrix with TWO general matrix inversions (its own TODO says so, :62):
//     compute_plane_world_to_camera_matrix(W) = inverse( compute_plane_camera_to_world_matrix( inverse(W) ) )
//     compute_plane_camera_to_world_matrix(C) = [ R_C 0 ; -p_C^T R_C  1 ]                                  (:53-60)
// and multiplies the (normal, d) 4-vector by it; the PlaneCoordinates constructor normalises the normal (plane_coordinates.hpp:19-22).
// `matrix44::inverse()` is Eigen 3.4.0's fixed-size 4x4 inverse, restated below from its scalar cofactor form
// (Eigen/src/LU/InverseImpl.h: cofactor_4x4 over general_det3_helper, then result /= (col(0) . row(0)^T).sum()) -- UNPINNED: Eigen is
// absent from this image, and a vectorised build of the reference takes Eigen's SSE path for the same inverse (another summation
// order).  polyref_plane_to_camera_analytic keeps the closed form [R 0; -t^T R 1] for W = [R t] -- what the product computes -- as
// the variant: tests/test_polygon_oracle.py bounds the difference and shows that no match decision depends on it.
double det3_helper(const double* m, int i1, int i2, int i3, int j1, int j2, int j3)
{
    return m[4 * i1 + j1] * (m[4 * i2 + j2] * m[4 * i3 + j3] - m[4 * i2 + j3] * m[4 * i3 + j2]);
}
double cofactor44(const double* m, int i, int j)
{
    const int i1 = (i + 1) % 4, i2 = (i + 2) % 4, i3 = (i + 3) % 4, j1 = (j + 1) % 4, j2 = (j + 2) % 4, j3 = (j + 3) % 4;
    return (det3_helper(m, i1, i2, i3, j1, j2, j3) + det3_helper(m, i2, i3, i1, j1, j2, j3)) + det3_helper(m, i3, i1, i2, j1, j2, j3);
}
void inverse44(const double* m, double* r)
{
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j)
        {
            const double c = cofactor44(m, j, i); // result(i, j) = +-cofactor(j, i)
            r[4 * i + j] = ((i + j) & 1) ? -c : c;
        }
    // det = matrix.col(0) . result.row(0)
    const double det = ((m[0] * r[0] + m[4] * r[1]) + m[8] * r[2]) + m[12] * r[3];
    for (int k = 0; k < 16; ++k)
        r[k] /= det;
}
// compute_plane_camera_to_world_matrix, camera_transformation.cpp:53-60
void plane_camera_to_world_matrix(const double* c2w, double* out)
{
    for (int i = 0; i < 3; ++i)
    {
        for (int j = 0; j < 3; ++j)
            out[4 * i + j] = c2w[4 * i + j];
        out[4 * i + 3] = 0.0;
    }
    for (int j = 0; j < 3; ++j) // -position^T * rotation
        out[12 + j] = ((-c2w[3]) * c2w[0 + j] + (-c2w[7]) * c2w[4 + j]) + (-c2w[11]) * c2w[8 + j];
    out[15] = 1.0;
}
// compute_plane_world_to_camera_matrix, :62-71
void plane_world_to_camera_matrix(const double* w2c, double* out)
{
    double c2w[16], pc2w[16];
    inverse44(w2c, c2w);                     // compute_camera_to_world_transform(worldToCamera), :19-24
    plane_camera_to_world_matrix(c2w, pc2w);
    inverse44(pc2w, out);
}
// PlaneCoordinates(matrix * parametrization): 4x4 times 4-vector, then the normal normalised
void plane_through(const double* M, const double* plane, double* out)
{
    double v[4];
    for (int i = 0; i < 4; ++i)
        v[i] = ((M[4 * i] * plane[0] + M[4 * i + 1] * plane[1]) + M[4 * i + 2] * plane[2]) + M[4 * i + 3] * plane[3];
    const V3 nn = normalized3({{v[0], v[1], v[2]}});
    out[0] = nn[0];
    out[1] = nn[1];
    out[2] = nn[2];
    out[3] = v[3];
}

// PlaneWorldCoordinates::to_camera_coordinates (plane_coordinates.cpp:20-24) through compute_plane_world_to_camera_matrix, the
// reference's operation sequence.  plane = (nx, ny, nz, d).
void polyref_plane_to_camera(const double* plane, const double* worldToCamera, double* out)
{
    double M[16];
    plane_world_to_camera_matrix(worldToCamera, M);
    plane_through(M, plane, out);
}
// PlaneCameraCoordinates::to_world_coordinates (plane_coordinates.cpp:15-18) with compute_plane_camera_to_world_matrix(cameraToWorld)
void polyref_plane_to_world(const double* plane, const double* cameraToWorld, double* out)
{
    double M[16];
    plane_camera_to_world_matrix(cameraToWorld, M);
    plane_through(M, plane, out);
}
// the variant: the closed form of the same matrix, inverse([R^T 0; t^T 1]) = [R 0; -t^T R 1] for worldToCamera = [R t]
void polyref_plane_to_camera_analytic(const double* plane, const double* worldToCamera, double* out)
{
    const V3 n {{plane[0], plane[1], plane[2]}};
    const V3 rn = rotate44(worldToCamera, n);
    const V3 t {{worldToCamera[3], worldToCamera[7], worldToCamera[11]}};
    // last row of the plane matrix: (-t^T R) . n + 1 * d
    V3 mtR;
    for (int j = 0; j < 3; ++j)
        mtR[j] = -((t[0] * worldToCamera[0 + j] + t[1] * worldToCamera[4 + j]) + t[2] * worldToCamera[8 + j]);
    const double d = ((mtR[0] * n[0] + mtR[1] * n[1]) + mtR[2] * n[2]) + plane[3];
    const V3 nn = normalized3(rn);
    out[0] = nn[0];
    out[1] = nn[1];
    out[2] = nn[2];
    out[3] = d;
}
// matrix44::inverse() as restated above (compute_world_to_camera_transform / compute_camera_to_world_transform, :19-24, :39-44)
void polyref_inverse44(const double* m, double* out) { inverse44(m, out); }
// utils::get_transformation_matrix(quaternion, position) (camera_transformation.hpp:16-19): Eigen's Quaternion::toRotationMatrix
// (the quaternion is used AS GIVEN -- the reference's tests pass unnormalised ones) beside the position.  q = (w, x, y, z).
void polyref_transform_from_quaternion(const double* q, const double* position, double* out)
{
    const double w = q[0], x = q[1], y = q[2], z = q[3];
    const double tx = 2 * x, ty = 2 * y, tz = 2 * z;
    const double twx = tx * w, twy = ty * w, twz = tz * w, txx = tx * x, txy = ty * x, txz = tz * x, tyy = ty * y, tyz = tz * y, tzz = tz * z;
    const double R[9] = {1 - (tyy + tzz), txy - twz, txz + twy, txy + twz, 1 - (txx + tzz), tyz - twx, txz - twy, tyz + twx, 1 - (txx + tyy)};
    for (int i = 0; i < 3; ++i)
    {
        for (int j = 0; j < 3; ++j)
            out[4 * i + j] = R[3 * i + j];
        out[4 * i + 3] = position[i];
        out[12 + i] = 0.0;
    }
    out[15] = 1.0;
}
// get_transformation_matrix(xFrom, yFrom, centerFrom, xTo, yTo, centerTo) (point_coordinates.cpp:24-70) as polygon_transform above
// restates it: linear part R_to * R_from^T over the bases [x y x^y], translation centerTo - centerFrom.  0 if the reference throws.
int polyref_transformation_matrix(const double* xFrom, const double* yFrom, const double* cFrom, const double* xTo, const double* yTo,
                                  const double* cTo, double* out)
{
    const V3 xf {{xFrom[0], xFrom[1], xFrom[2]}}, yf {{yFrom[0], yFrom[1], yFrom[2]}}, xt {{xTo[0], xTo[1], xTo[2]}}, yt {{yTo[0], yTo[1], yTo[2]}};
    if (!frame_ok(xf, yf) || !frame_ok(xt, yt))
        return 0;
    const V3 zF = cross3(xf, yf), zT = cross3(xt, yt);
    const V3 from[3] = {xf, yf, zF}, to[3] = {xt, yt, zT};
    std::memset(out, 0, 16 * sizeof(double));
    for (int i = 0; i < 3; ++i)
    {
        for (int j = 0; j < 3; ++j)
            out[4 * i + j] = (to[0][i] * from[0][j] + to[1][i] * from[1][j]) + to[2][i] * from[2][j];
        out[4 * i + 3] = cTo[i] - cFrom[i];
    }
    out[15] = 1.0;
    return 1;
}

} // extern "C"
