// Mirrors the parts of the reference's tests/test_polygons.cpp (SquareTests.SimpleFitting, :6-43) that concern the
// polygon built from plane boundary points, plus shape checks for concave inputs.  Exit code 0 = all passed.
#include <cmath>
#include <cstdio>
#include <random>
#include <vector>

#include "../../rgb-d-slam_amd/host/boundary_polygon.hpp"

using namespace rgbd_slam::utils;

static int failures = 0;
#define EXPECT(cond)                                              \
    do                                                            \
    {                                                             \
        if (!(cond))                                              \
        {                                                         \
            std::printf("FAILED %s:%d  %s\n", __FILE__, __LINE__, #cond); \
            ++failures;                                           \
        }                                                         \
    } while (0)

int main()
{
    {
        // SquareTests.SimpleFitting
        const std::vector<vector3> points {{-1000.0, 1000.0, 0.0}, {1000.0, 1000.0, 0.0}, {-1000.0, -1000.0, 0.0}, {1000.0, -1000.0, 0.0}};
        const vector3 normal {0, 0, 1}, center {0, 0, 0};
        Polygon polygon(points, normal, center);
        EXPECT(polygon.get_center() == center);
        const vector3 n = polygon.get_normal();
        EXPECT(n[0] == 0 && n[1] == 0 && n[2] == 1); // x_axis.cross(y_axis) == normal
        EXPECT(polygon.boundary_length() == 4);
        EXPECT(std::abs(polygon.get_area() - 4e6) < 0.1);
        EXPECT(polygon.is_valid());
        const double in[][2] = {{0, 0}, {-999.99, 999.99}, {-999.99, -999.99}, {999.99, -999.99}, {999.99, 999.99},
                                {0, 999.99}, {0, -999.99}, {999.99, 0}, {-999.99, 0}};
        for (const auto& p : in)
            EXPECT(polygon.contains({p[0], p[1]}));
        EXPECT(!polygon.contains({1000.01, 0}) && !polygon.contains({0, -1500}));
        // same polygon with flipped normal
        const vector3 neg {0, 0, -1};
        Polygon inv(points, neg, center);
        const vector3 ni = inv.get_normal();
        EXPECT(ni[0] == 0 && ni[1] == 0 && ni[2] == -1);
        EXPECT(std::abs(inv.get_area() - 4e6) < 0.1);
        // SquareTests.Unions: diamond area
        const std::vector<vector3> diamond {{-1000.0, 0.0, 0.0}, {1000.0, 0.0, 0.0}, {0.0, -1000.0, 0.0}, {0.0, 1000.0, 0.0}};
        EXPECT(Polygon(diamond, normal, center).area() == 2e6);
    }
    {
        // tilted plane: axes orthonormal, round trip through the plane frame
        vector3 nrm {0.3, -0.2, -0.933};
        const double l = std::sqrt(nrm[0] * nrm[0] + nrm[1] * nrm[1] + nrm[2] * nrm[2]);
        nrm = {nrm[0] / l, nrm[1] / l, nrm[2] / l};
        const auto ax = get_plane_coordinate_system(nrm);
        const vector3 &x = ax.first, &y = ax.second;
        EXPECT(std::abs(x[0] * y[0] + x[1] * y[1] + x[2] * y[2]) < 1e-12);
        EXPECT(std::abs(x[0] * nrm[0] + x[1] * nrm[1] + x[2] * nrm[2]) < 1e-12);
        const vector3 c {10, 20, 2000};
        const vector3 p = get_point_from_plane_coordinates({123.0, -45.0}, c, x, y);
        const vector2 q = get_projected_plan_coordinates(p, c, x, y);
        EXPECT(std::abs(q[0] - 123.0) < 1e-9 && std::abs(q[1] + 45.0) < 1e-9);
    }
    {
        // L-shaped ring of cell centres (what the boundary candidates of an L-shaped plane look like): the hull must be
        // concave -- its area stays close to the L (3/4 of the bounding square), not the convex hull's 7/8
        std::vector<vector3> pts;
        const double s = 72.0; // ~20 px at 2 m
        for (int i = 0; i <= 20; ++i)
            for (int j = 0; j <= 20; ++j)
            {
                const bool inL = !(i > 10 && j > 10);
                if (!inL)
                    continue;
                const bool edge = i == 0 || j == 0 || (i == 20 && j <= 10) || (j == 20 && i <= 10) || (i == 10 && j >= 10) || (j == 10 && i >= 10);
                if (edge)
                    pts.push_back({i * s, j * s, 0.0});
            }
        Polygon poly(pts, {0, 0, 1}, {0, 0, 0});
        EXPECT(poly.is_valid());
        const double full = 20 * s * 20 * s;
        EXPECT(poly.get_area() > 0.70 * full && poly.get_area() < 0.80 * full);
        auto inPlane = [&](double X, double Y) {
            return get_projected_plan_coordinates({X, Y, 0.0}, poly.get_center(), poly.get_x_axis(), poly.get_y_axis());
        };
        EXPECT(poly.contains(inPlane(5 * s, 5 * s)) && poly.contains(inPlane(15 * s, 5 * s)) && poly.contains(inPlane(5 * s, 15 * s)));
        EXPECT(!poly.contains(inPlane(15 * s, 15 * s)));
        EXPECT(poly.boundary_length() <= 8); // simplified to the 6 corners (+ tolerance)
    }
    {
        // random blobs: always a valid ring that holds every input point
        std::mt19937 rng(5);
        std::uniform_real_distribution<double> U(-1, 1);
        for (int trial = 0; trial < 200; ++trial)
        {
            std::vector<vector3> pts;
            const int n = 5 + trial % 120;
            for (int i = 0; i < n; ++i)
            {
                const double a = U(rng) * 3.14159, r = 500 + 400 * U(rng);
                pts.push_back({r * std::cos(a), r * std::sin(a), 0});
            }
            Polygon poly(pts, {0, 0, 1}, {0, 0, 0});
            EXPECT(poly.is_valid());
            EXPECT(poly.get_area() > 0);
        }
    }
    {
        // SquareTests.SimpleFitting :36-47 and SquareTests.Unions (areas): union / intersection areas
        const std::vector<vector3> sq {{-1000.0, 1000.0, 0.0}, {1000.0, 1000.0, 0.0}, {-1000.0, -1000.0, 0.0}, {1000.0, -1000.0, 0.0}};
        const vector3 normal {0, 0, 1}, center {0, 0, 0};
        Polygon polygon(sq, normal, center);
        EXPECT(std::abs(polygon.get_area() - polygon.union_area(polygon)) < 0.1);
        EXPECT(std::abs(polygon.get_area() - polygon.inter_area(polygon)) < 0.1);
        Polygon inv(sq, {0, 0, -1}, center);
        EXPECT(std::abs(polygon.get_area() - polygon.union_area(inv)) < 0.1);
        EXPECT(std::abs(polygon.get_area() - polygon.inter_area(inv)) < 0.1);
        const std::vector<vector3> di {{-1000.0, 0.0, 0.0}, {1000.0, 0.0, 0.0}, {0.0, -1000.0, 0.0}, {0.0, 1000.0, 0.0}};
        Polygon diamond(di, normal, center);
        EXPECT(std::abs(polygon.inter_area(diamond) - 2e6) < 0.1 && std::abs(polygon.union_area(diamond) - 4e6) < 0.1);
        // the diamond shifted by half a length: half of it sticks out (Unions: area 4e6 + 1e6)
        Polygon shifted = diamond.project(normal, {-1000.0, 0.0, 0.0}); // same boundary seen from a frame moved to -1000
        std::vector<vector3> di2;
        for (const auto& v : di)
            di2.push_back({v[0] + 1000.0, v[1], v[2]});
        Polygon diamondRight(di2, normal, center);
        EXPECT(std::abs(polygon.union_area(diamondRight) - 5e6) < 0.1);
        EXPECT(std::abs(polygon.inter_area(diamondRight) - 1e6) < 0.1);
        EXPECT(std::abs(polygon.inter_over_union(diamondRight) - 0.2) < 1e-6);
        EXPECT(std::abs(shifted.get_area() - 2e6) < 0.1);
        // projection on a perpendicular plane has zero area; on a 45 degree plane it shrinks by cos(45)
        EXPECT(polygon.project({1, 0, 0}, {1000, 0, 0}).get_area() < 1e-6);
        const double c45 = std::sqrt(0.5);
        const double a45 = polygon.project({c45, 0, c45}, {1000, 0, 0}).get_area();
        EXPECT(a45 > 0 && a45 < polygon.get_area() && std::abs(a45 - 4e6 * c45) < 1.0);
        // disjoint and concave cases
        std::vector<vector3> far;
        for (const auto& v : sq)
            far.push_back({v[0] + 5000.0, v[1], v[2]});
        EXPECT(polygon.inter_area(Polygon(far, normal, center)) == 0.0);
    }
    {
        // SquareTests.SimpleFitting, second half (reference tests/test_polygons.cpp:44-66): the mirrored polygon and transform()
        const std::vector<vector3> points {{-1000.0, 1000.0, 0.0}, {1000.0, 1000.0, 0.0}, {-1000.0, -1000.0, 0.0}, {1000.0, -1000.0, 0.0}};
        const vector3 normal {0, 0, 1}, minusNormal {0, 0, -1}, center {0, 0, 0};
        auto eq3 = [](const vector3& a, const vector3& b) { return std::abs(a[0] - b[0]) + std::abs(a[1] - b[1]) + std::abs(a[2] - b[2]) < 1e-12; };
        Polygon polygon(points, normal, center), polygonInverse(points, minusNormal, center);
        EXPECT(eq3(polygonInverse.get_normal(), minusNormal));
        EXPECT(std::abs(polygon.get_area() - polygon.union_area(polygonInverse)) < 0.1);
        EXPECT(std::abs(polygon.get_area() - polygon.inter_area(polygonInverse)) < 0.1);
        const Polygon inversed = polygon.transform(minusNormal, center);
        EXPECT(eq3(inversed.get_normal(), minusNormal) && eq3(inversed.get_center(), center));
        const Polygon shiftedT = polygon.transform(normal, {500.0, 500.0, 0.0});
        EXPECT(eq3(shiftedT.get_normal(), normal) && eq3(shiftedT.get_center(), {500.0, 500.0, 0.0}));
        EXPECT(shiftedT.get_area() == polygon.get_area());
        const Polygon turned = polygon.transform({1.0, 0.0, 0.0}, center);
        EXPECT(eq3(turned.get_normal(), {1.0, 0.0, 0.0}) && eq3(turned.get_center(), center) && turned.boundary_length() == 4);
        const Polygon shiftedP = polygon.project(normal, {500.0, 500.0, 0.0});
        EXPECT(eq3(shiftedP.get_normal(), normal) && eq3(shiftedP.get_center(), {500.0, 500.0, 0.0}) && shiftedP.get_area() == polygon.get_area());
        const Polygon turnedP = polygon.project({1.0, 0.0, 0.0}, {1000.0, 0.0, 0.0});
        EXPECT(eq3(turnedP.get_normal(), {1.0, 0.0, 0.0}) && turnedP.boundary_length() == 4 && turnedP.get_area() == 0.0);
    }
    {
        // SquareTests.Unions (reference tests/test_polygons.cpp:91-152), step for step
        const vector3 normal {0, 0, 1}, center {0, 0, 0};
        Polygon rectangle({{-1000.0, 1000.0, 0.0}, {1000.0, 1000.0, 0.0}, {-1000.0, -1000.0, 0.0}, {1000.0, -1000.0, 0.0}}, normal, center);
        Polygon diamond({{-1000.0, 0.0, 0.0}, {1000.0, 0.0, 0.0}, {0.0, -1000.0, 0.0}, {0.0, 1000.0, 0.0}}, normal, center);
        auto near = [](double a, double b) { return std::abs(a - b) < 1e-3; };
        EXPECT(near(diamond.area(), 2e6));
        EXPECT(rectangle.merge_union(diamond));          // diamond inside, its corners ON the rectangle's edges
        EXPECT(near(rectangle.area(), 4e6) && rectangle.boundary_length() == 4);
        diamond = diamond.transform(normal, {1000.0, 0.0, 0.0}); // shift the shape by half a length
        EXPECT(near(diamond.area(), 2e6));
        EXPECT(rectangle.merge_union(diamond));
        EXPECT(rectangle.boundary_length() == 5 && near(rectangle.area(), 4e6 + 1e6));
        diamond = diamond.transform(normal, {-1000.0, 0.0, 0.0}); // offset to the left
        EXPECT(rectangle.merge_union(diamond));
        EXPECT(rectangle.boundary_length() == 6 && near(rectangle.area(), 4e6 + 2e6));
        diamond = diamond.transform(normal, {0.0, 1000.0, 0.0});  // offset to the top
        EXPECT(rectangle.merge_union(diamond));
        EXPECT(rectangle.boundary_length() == 5 && near(rectangle.area(), 4e6 + 3e6)); // reduced by the simplification
        diamond = diamond.transform(normal, {0.0, -1000.0, 0.0}); // offset to the bottom
        EXPECT(rectangle.merge_union(diamond));
        EXPECT(rectangle.boundary_length() == 4 && near(rectangle.area(), 4e6 + 4e6));
        EXPECT(rectangle.is_valid() && rectangle.contains({0, 1900}) && !rectangle.contains({1500, 1500}));

        // beyond the reference's cases: proper crossings, containment both ways, disjoint, random agreement with union_area
        Polygon a({{0, 0, 0}, {400, 0, 0}, {400, 400, 0}, {0, 400, 0}}, normal, center);
        Polygon b({{200, 200, 0}, {600, 200, 0}, {600, 600, 0}, {200, 600, 0}}, normal, center);
        Polygon u = a;
        EXPECT(u.merge_union(b) && near(u.area(), 2 * 160000.0 - 40000.0) && u.boundary_length() == 8);
        Polygon big({{-100, -100, 0}, {700, -100, 0}, {700, 700, 0}, {-100, 700, 0}}, normal, center);
        u = a;
        EXPECT(u.merge_union(big) && near(u.area(), 640000.0) && u.boundary_length() == 4);
        u = big;
        EXPECT(u.merge_union(a) && near(u.area(), 640000.0) && u.boundary_length() == 4);
        Polygon farAway({{5000, 0, 0}, {5100, 0, 0}, {5100, 100, 0}, {5000, 100, 0}}, normal, center);
        u = a;
        EXPECT(u.merge_union(farAway) && near(u.area(), 160000.0)); // disjoint: the bigger piece stays (polygon.cpp:474-492)
        u = farAway;
        EXPECT(u.merge_union(a) && near(u.area(), 160000.0));
        std::mt19937 gen(7);
        std::uniform_real_distribution<double> d(-300.0, 300.0), sz(150.0, 500.0);
        for (int rep = 0; rep < 200; ++rep)
        {
            // a random convex quadrilateral around a random centre against the fixed square: area(outer boundary of the
            // union) must equal area(a) + area(q) - inter_area, as long as the union encloses no hole (convex U convex)
            const double cx = d(gen) + 200, cy = d(gen) + 200, r = sz(gen);
            std::vector<vector3> pts;
            for (int k = 0; k < 4; ++k)
            {
                const double ang = (k + 0.15 * (d(gen) / 300.0)) * M_PI / 2 + 0.4;
                pts.push_back({cx + r * std::cos(ang), cy + r * std::sin(ang), 0.0});
            }
            Polygon q(pts, normal, center);
            if (!q.is_valid() || a.inter_area(q) <= 0)
                continue;
            Polygon m = a;
            EXPECT(m.merge_union(q));
            EXPECT(std::abs(m.area() - a.union_area(q)) < 0.02 * a.union_area(q)); // simplify() may shave slivers
            EXPECT(m.is_valid());
        }
    }
    {
        // interior rings: a union whose outlines enclose a region they do not cover (boost::geometry::union_ returns it as
        // an inner ring, which the reference assigns to its polygon: polygon.cpp:325-336, :463-470)
        const vector3 normal {0.0, 0.0, 1.0}, center {0.0, 0.0, 0.0};
        const auto axes = get_plane_coordinate_system(normal);
        auto near = [](double a, double b) { return std::abs(a - b) < 1e-3 * std::max(1.0, std::abs(b)); };
        // a "C" open to the right (3000 x 3000 with the notch x in [1000, 3000], y in [1000, 2000] missing) ...
        Polygon cShape(std::vector<vector2> {{0, 0}, {3000, 0}, {3000, 1000}, {1000, 1000}, {1000, 2000}, {3000, 2000}, {3000, 3000}, {0, 3000}},
                       axes.first, axes.second, center);
        EXPECT(cShape.is_valid() && near(cShape.area(), 9e6 - 2e6));
        // ... closed by a bar x in [2500, 4000]: the union is 4000 x 3000 with the hole x in [1000, 2500], y in [1000, 2000]
        Polygon bar(std::vector<vector2> {{2500, 0}, {4000, 0}, {4000, 3000}, {2500, 3000}}, axes.first, axes.second, center);
        Polygon ring = cShape;
        EXPECT(ring.merge_union(bar));
        EXPECT(ring.is_valid() && ring.interior_rings().size() == 1);
        EXPECT(near(ring.area(), 12e6 - 1.5e6) && near(ring.get_area(), ring.area()));
        EXPECT(!ring.contains({1700, 1500}) && !ring.contains({1000, 1500}));             // in the hole / on its edge
        EXPECT(ring.contains({500, 500}) && ring.contains({3000, 1500}) && ring.contains({2700, 1500}));
        EXPECT(!ring.contains({4500, 1500}));
        // the other way round gives the same region
        Polygon ring2 = bar;
        EXPECT(ring2.merge_union(cShape) && ring2.interior_rings().size() == 1 && near(ring2.area(), ring.area()));
        // areas against a probe that covers the hole: x in [500, 3000], y in [500, 2500]
        Polygon probe(std::vector<vector2> {{500, 500}, {3000, 500}, {3000, 2500}, {500, 2500}}, axes.first, axes.second, center);
        EXPECT(near(ring.inter_area(probe), 5e6 - 1.5e6) && near(probe.inter_area(ring), 5e6 - 1.5e6));
        EXPECT(near(ring.union_area(probe), 10.5e6 + 5e6 - 3.5e6));
        EXPECT(near(ring.inter_over_union(probe), 3.5e6 / 12e6));
        EXPECT(near(ring.inter_area(ring2), ring.area()) && near(ring.inter_over_union(ring2), 1.0));
        // a rigid move and a re-projection keep the hole
        const Polygon moved = ring.transform(normal, {100.0, -50.0, 20.0});
        EXPECT(moved.interior_rings().size() == 1 && near(moved.area(), ring.area()) && moved.is_valid());
        const Polygon same = ring.project(normal, center);
        EXPECT(same.interior_rings().size() == 1 && near(same.area(), ring.area()) && !same.contains({1700, 1500}));
        // a later union that stays clear of the hole keeps it; one that covers it fills it
        Polygon clear(std::vector<vector2> {{3500, 2500}, {5000, 2500}, {5000, 3500}, {3500, 3500}}, axes.first, axes.second, center);
        Polygon keep = ring;
        EXPECT(keep.merge_union(clear) && keep.interior_rings().size() == 1 && !keep.contains({1700, 1500}));
        EXPECT(near(keep.area(), 10.5e6 + 1.5e6 - 0.25e6));
        Polygon plug(std::vector<vector2> {{900, 900}, {2600, 900}, {2600, 2100}, {900, 2100}}, axes.first, axes.second, center);
        Polygon filled = ring;
        EXPECT(filled.merge_union(plug) && filled.interior_rings().empty() && near(filled.area(), 12e6) && filled.contains({1700, 1500}));
        // a later union that covers the hole PARTLY leaves the rest of it a hole (boost::geometry::union_ of a polygon with an
        // interior ring; through round 5 such a hole counted as filled): the plug covers x < 1750 of the hole [1000, 2500] x [1000, 2000]
        Polygon halfPlug(std::vector<vector2> {{900, 900}, {1750, 900}, {1750, 2100}, {900, 2100}}, axes.first, axes.second, center);
        Polygon half = ring;
        EXPECT(half.merge_union(halfPlug) && half.interior_rings().size() == 1 && near(half.area(), 12e6 - 0.75e6));
        EXPECT(half.contains({1500, 1500}) && !half.contains({2000, 1500}) && half.is_valid());
        // the other way round: the plain rectangle takes the ring in
        Polygon half2 = halfPlug;
        EXPECT(half2.merge_union(ring) && half2.interior_rings().size() == 1 && near(half2.area(), 12e6 - 0.75e6) && !half2.contains({2000, 1500}));
        // a bar through the middle of the hole cuts it in two: y in [1400, 1600] covered, two holes of 1500 x 400 remain
        Polygon midBar(std::vector<vector2> {{500, 1400}, {3500, 1400}, {3500, 1600}, {500, 1600}}, axes.first, axes.second, center);
        Polygon cut = ring;
        EXPECT(cut.merge_union(midBar) && cut.interior_rings().size() == 2 && near(cut.area(), 12e6 - 2 * 0.6e6));
        EXPECT(cut.contains({1700, 1500}) && !cut.contains({1700, 1200}) && !cut.contains({1700, 1800}));
        // an operand that brings a hole of its own over the first one's: only the overlap of the two holes stays open
        Polygon frame2 = cShape;
        EXPECT(frame2.merge_union(bar));                                   // hole [1000, 2500] x [1000, 2000]
        // the same figure 500 further along x: hole [1500, 3000] x [1000, 2000]
        Polygon shifted(std::vector<vector2> {{500, 0}, {3500, 0}, {3500, 1000}, {1500, 1000}, {1500, 2000}, {3500, 2000}, {3500, 3000}, {500, 3000}},
                        axes.first, axes.second, center);
        Polygon bar2(std::vector<vector2> {{3000, 0}, {4500, 0}, {4500, 3000}, {3000, 3000}}, axes.first, axes.second, center);
        EXPECT(shifted.merge_union(bar2) && shifted.interior_rings().size() == 1);
        Polygon both = frame2;
        EXPECT(both.merge_union(shifted) && both.interior_rings().size() == 1);
        EXPECT(near(both.area(), 4500.0 * 3000.0 - 1000.0 * 1000.0) && !both.contains({2000, 1500}) && both.contains({1200, 1500}) && both.contains({2800, 1500}));
        // two disjoint operands: the bigger piece stays, with its hole
        Polygon farAway(std::vector<vector2> {{9000, 0}, {9500, 0}, {9500, 500}, {9000, 500}}, axes.first, axes.second, center);
        Polygon big = ring;
        EXPECT(big.merge_union(farAway) && big.interior_rings().size() == 1 && near(big.area(), ring.area()));
        Polygon small = farAway;
        EXPECT(small.merge_union(ring) && small.interior_rings().size() == 1 && near(small.area(), ring.area()));
    }
    std::printf(failures ? "%d FAILURES\n" : "all polygon tests passed\n", failures);
    return failures ? 1 : 0;
}
