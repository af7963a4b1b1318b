"""The polygon oracle (oracle/polygon_oracle.cpp, TEST INFRASTRUCTURE) against the reference's own polygon tests.

SquareTests.SimpleFitting (reference tests/test_polygons.cpp:6-89) is replayed assertion by assertion: that is the pin of the
oracle for rows N1 / N2.  The rest checks the restated pieces against closed forms and against an independent method
(Monte-Carlo / shapely-free inclusion-exclusion on convex shapes), and records the reference's quirks the oracle keeps."""
import math

import numpy as np
import pytest


@pytest.fixture(scope="module")
def P():
    import polygon_oracle_py

    polygon_oracle_py.build()
    return polygon_oracle_py


SQUARE = np.array([(-1000.0, 1000.0, 0.0), (1000.0, 1000.0, 0.0), (-1000.0, -1000.0, 0.0), (1000.0, -1000.0, 0.0)])


def test_reference_square_simple_fitting(P):
    """tests/test_polygons.cpp:6-89, line by line."""
    normal, center = np.array([0.0, 0.0, 1.0]), np.zeros(3)
    polygon = P.Polygon.from_points(SQUARE, normal, center)
    assert not polygon.threw
    assert np.array_equal(polygon.center, center)                                   # :19
    assert np.array_equal(polygon.get_normal(), normal)                             # :20-21
    assert polygon.boundary_length() == 4                                           # :23
    assert abs(polygon.area - 4e6) <= 0.1                                           # :25
    for x, y in ((0, 0), (-999.99, 999.99), (-999.99, -999.99), (999.99, -999.99), (999.99, 999.99),
                 (0, 999.99), (0, -999.99), (999.99, 0), (-999.99, 0)):            # :28-37
        assert polygon.contains(x, y), (x, y)
    assert abs(polygon.area - polygon.union_area(polygon)) <= 0.1                   # :40
    assert abs(polygon.area - polygon.inter_area(polygon)) <= 0.1                   # :41
    inverse = P.Polygon.from_points(SQUARE, -normal, center)                        # :44
    assert np.array_equal(inverse.get_normal(), -normal)                            # :45
    assert abs(polygon.area - polygon.union_area(inverse)) <= 0.1                   # :48
    assert abs(polygon.area - polygon.inter_area(inverse)) <= 0.1                   # :49
    inversed = polygon.transform(-normal, center)                                   # :52-54
    assert np.array_equal(inversed.get_normal(), -normal) and np.array_equal(inversed.center, center)
    shifted = polygon.transform(normal, [500.0, 500.0, 0.0])                        # :58-61
    assert np.array_equal(shifted.get_normal(), normal) and np.array_equal(shifted.center, [500.0, 500.0, 0.0])
    assert shifted.area == polygon.area
    turned = polygon.transform([1.0, 0.0, 0.0], center)                             # :64-68
    assert np.array_equal(turned.get_normal(), [1.0, 0.0, 0.0]) and np.array_equal(turned.center, center)
    assert turned.boundary_length() == 4
    sp = polygon.project(normal, [500.0, 500.0, 0.0])                               # :70-73
    assert np.array_equal(sp.get_normal(), normal) and np.array_equal(sp.center, [500.0, 500.0, 0.0])
    assert sp.area == polygon.area
    tp = polygon.project([1.0, 0.0, 0.0], [1000.0, 0.0, 0.0])                       # :76-80
    assert np.array_equal(tp.get_normal(), [1.0, 0.0, 0.0]) and np.array_equal(tp.center, [1000.0, 0.0, 0.0])
    assert tp.boundary_length() == 4 and tp.area == 0.0
    n45 = np.array([0.5, 0.0, 0.5]) / np.linalg.norm([0.5, 0.0, 0.5])
    semi = polygon.project(n45, [1000.0, 0.0, 0.0])                                 # :82-88
    assert np.allclose(semi.get_normal(), n45, rtol=1e-12, atol=0) and np.array_equal(semi.center, [1000.0, 0.0, 0.0])
    assert semi.boundary_length() == 4 and 0.0 < semi.area < polygon.area


def test_reference_diamond_areas(P):
    """tests/test_polygons.cpp:104-111 and :121-122 (the constructor / transform half of SquareTests.Unions; merge_union is
    not on the device path and is held by tests/host/test_polygon.cpp)."""
    normal, center = np.array([0.0, 0.0, 1.0]), np.zeros(3)
    diamond = P.Polygon.from_points([(-1000.0, 0, 0), (1000.0, 0, 0), (0, -1000.0, 0), (0, 1000.0, 0)], normal, center)
    assert diamond.area == 2e6
    moved = diamond.transform(normal, [1000.0, 0.0, 0.0])
    assert moved.area == 2e6
    rect = P.Polygon.from_points(SQUARE, normal, center)
    assert abs(rect.inter_area(diamond) - 2e6) <= 1e-6       # the diamond lies inside the rectangle
    assert abs(rect.inter_area(moved) - 1e6) <= 1e-6         # half of it once shifted by half a length (:124-127: union 5e6)
    assert abs(rect.union_area(moved) - 5e6) <= 1e-6


def test_hull_walk_orientation_ladder_and_no_dedupe(P):
    """The walk leaves a COUNTER-clockwise closed hull that starts at the lowest point (concave_fitting.cpp:114-173); the
    constructor's repair step turns it clockwise (correct_boost_polygon.hpp:172-186).  Duplicates are NOT removed on this call
    path (polygon.cpp:303 binds the non-const overload, concave_fitting.cpp:69)."""
    rng = np.random.default_rng(3)
    g = np.array([(i * 40.0, j * 40.0) for i in range(9) for j in range(9)]) + rng.normal(0, 1.0, (81, 2))
    ok, hull, k = P.concave_hull(g)
    assert ok and k == 3 and np.array_equal(hull[0], hull[-1])
    assert hull[0, 1] == g[:, 1].min()
    x, y = hull[:, 0], hull[:, 1]
    assert 0.5 * np.sum(x[:-1] * y[1:] - x[1:] * y[:-1]) > 0  # counter-clockwise
    pts3 = np.concatenate([g, np.zeros((81, 1))], 1)
    pol = P.Polygon.from_points(pts3, [0, 0, 1.0], [0, 0, 0.0])
    assert pol.valid and pol.area > 0 and pol.is_valid()
    # whatever the frame, every input point is inside the outline, on it, or within Douglas-Peucker's reach of it
    ring = hull[:-1][::-1]
    raw = P.Polygon(ring, [1, 0, 0], [0, 1, 0], [0, 0, 0])
    assert all(raw.locate(*q) >= 0 for q in g)
    for q in pts3:
        d = q - pol.center
        assert pol.distance_outside(float(pol.x_axis @ d), float(pol.y_axis @ d)) <= pol.simplify_reach()
    # k larger than the point count ends the ladder (concave_fitting.cpp:86-87)
    ok5, _, _ = P.concave_hull(g[:4], k=0)
    assert ok5


def test_point_in_polygon_zero_crossings_quirk(P):
    """concave_fitting.cpp:416-417: a point whose +x ray crosses no hull edge counts as INSIDE.  A hull that leaves points out
    on its right is therefore accepted -- the oracle keeps the quirk (a correct containment test would climb the ladder)."""
    # a 'C' opened to the right plus one point far to the right of everything
    pts = [(0, 0), (100, 0), (100, 20), (20, 20), (20, 80), (100, 80), (100, 100), (0, 100), (0, 50), (60, 50)]
    ok, hull, k = P.concave_hull(pts, k=3)
    if ok:
        ring = hull[:-1] if np.array_equal(hull[0], hull[-1]) else hull
        pol = P.Polygon(ring[::-1], [1, 0, 0], [0, 1, 0], [0, 0, 0])
        outside = [q for q in pts if pol.locate(*q) < 0]
        # whatever is outside lies to the right of every edge it could cross
        for q in outside:
            assert all(not ((a[1] <= q[1] < b[1]) or (b[1] <= q[1] < a[1])) or q[0] >= min(a[0], b[0]) for a, b in zip(hull[:-1], hull[1:]))


def test_inter_area_against_closed_forms(P):
    """Boundary integration vs closed forms: shifted squares, a square and a rotated square, disjoint, nested, shared edges."""
    def sq(x0, y0, s):
        return np.array([(x0, y0), (x0, y0 + s), (x0 + s, y0 + s), (x0 + s, y0)])  # clockwise
    a = sq(0, 0, 100)
    assert abs(P.rings_inter_area(a, a) - 1e4) < 1e-9
    assert abs(P.rings_inter_area(a, a[::-1]) - 1e4) < 1e-9                      # orientation is normalised
    assert abs(P.rings_inter_area(a, sq(50, 50, 100)) - 2500) < 1e-9
    assert abs(P.rings_inter_area(a, sq(100, 0, 100))) < 1e-9                    # touching along an edge from outside
    assert abs(P.rings_inter_area(a, sq(0, 0, 50)) - 2500) < 1e-9                # nested, two shared edges
    assert abs(P.rings_inter_area(a, sq(25, 25, 50)) - 2500) < 1e-9              # strictly nested
    assert P.rings_inter_area(a, sq(300, 300, 10)) == 0.0
    # an octagon: the square [-50,50]^2 against itself rotated by 45 degrees
    r = 50 * math.sqrt(2)
    dia = np.array([(0, -r), (-r, 0), (0, r), (r, 0)])
    s = sq(-50, -50, 100)
    want = 1e4 - 4 * 0.5 * (r - 50) * 2 * (r - 50)
    assert abs(P.rings_inter_area(s, dia) - want) < 1e-8
    # concave against convex: an L (area 7500) against the upper-right square it does not fill
    L = np.array([(0, 0), (0, 100), (50, 100), (50, 50), (100, 50), (100, 0)])
    assert abs(P.rings_inter_area(L, sq(50, 50, 50))) < 1e-9
    assert abs(P.rings_inter_area(L, sq(25, 25, 50)) - (2500 - 625)) < 1e-9


def test_inter_area_randomised_against_pixel_count(P):
    """Random star-shaped outlines: boundary integration vs a brute-force count on a fine lattice (an estimate, 1 % bar)."""
    rng = np.random.default_rng(11)
    for _ in range(12):
        rings = []
        for _k in range(2):
            m = int(rng.integers(5, 40))
            t = np.sort(rng.uniform(0, 2 * np.pi, m))[::-1]
            r = rng.uniform(40, 100, m)
            c = rng.uniform(-30, 30, 2)
            rings.append(np.stack([c[0] + r * np.cos(t), c[1] + r * np.sin(t)], 1))
        got = P.rings_inter_area(rings[0], rings[1])
        xs = np.arange(-130, 130, 0.5) + 0.25
        X, Y = np.meshgrid(xs, xs)

        def inside(ring):
            ins = np.zeros(X.shape, bool)
            for (ax, ay), (bx, by) in zip(ring, np.roll(ring, -1, 0)):
                cond = (ay > Y) != (by > Y)
                with np.errstate(divide="ignore", invalid="ignore"):
                    xi = ax + (bx - ax) * (Y - ay) / (by - ay)
                ins ^= cond & (X < xi)
            return ins
        est = float(np.count_nonzero(inside(rings[0]) & inside(rings[1]))) * 0.25
        assert abs(got - est) <= 0.01 * max(est, 100.0) + 30.0, (got, est)


def test_plane_and_polygon_through_a_pose(P):
    """to_camera_coordinates / to_camera_space under a rigid motion: the plane equation holds for the moved polygon's points,
    areas are preserved, and the identity leaves everything in place."""
    rng = np.random.default_rng(5)
    normal = np.array([0.2, -0.3, 0.9])
    normal /= np.linalg.norm(normal)
    normal /= np.linalg.norm(normal)
    d = -1800.0
    center = normal * (-d)
    a = np.cross(normal, [1.0, 0, 0])
    a /= np.linalg.norm(a)
    b = np.cross(normal, a)
    uv = rng.uniform(-400, 400, (40, 2))
    pts = center + uv[:, :1] * a + uv[:, 1:] * b
    pol = P.Polygon.from_points(pts, normal, center)
    assert pol.valid
    ang = 0.1
    R = np.array([[math.cos(ang), 0, math.sin(ang)], [0, 1, 0], [-math.sin(ang), 0, math.cos(ang)]])
    T = np.eye(4)
    T[:3, :3], T[:3, 3] = R, [30.0, -12.0, 55.0]
    n2, d2 = P.plane_to_camera(normal, d, T)
    moved = pol.to_camera_space(T)
    assert abs(moved.area - pol.area) <= 1e-9 * pol.area
    for q in moved.ring:
        p3 = moved.center + q[0] * moved.x_axis + q[1] * moved.y_axis
        assert abs(n2 @ p3 + d2) < 1e-6
    same = pol.to_camera_space(np.eye(4))
    assert np.allclose(same.ring, pol.ring, rtol=0, atol=1e-9) and np.array_equal(same.center, pol.center)
    n0, d0 = P.plane_to_camera(normal, d, np.eye(4))
    assert np.allclose(n0, normal, rtol=0, atol=1e-15) and d0 == d


def test_find_matches_selection_loop(P):
    """MapPlane::find_matches on two hand-made frames: gates, best area, one detected plane per map plane, the index-0 quirk."""
    def wall(z, x0, x1, nrm=(0.0, 0.0, 1.0)):
        pts = np.array([(x, y, z) for x in np.linspace(x0, x1, 6) for y in np.linspace(-300, 300, 6)])
        n = np.asarray(nrm, float)
        return n, -z, P.Polygon.from_points(pts, n, n * z)
    prev = [wall(2000, -500, 0), wall(2000, 100, 600), wall(3500, -500, 500)]
    cur = [wall(2010, -480, 20), wall(2005, 90, 610), wall(3490, -500, 500)]
    m, inter = P.find_matches(prev, cur)
    assert m == [-1, 1, 2]              # the best candidate of map plane 0 is detected plane 0: `selectedIndex <= 0` drops it
    assert inter[0, 0] > 0 and inter[0, 2] == -1.0   # 1 500 mm apart: the distance gate
    m2, _ = P.find_matches(prev, cur, allow_index0=True)
    assert m2 == [0, 1, 2]
