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
    std::printf(failures ? "%d FAILURES\n" : "all polygon tests passed\n", failures);
    return failures ? 1 : 0;
}
