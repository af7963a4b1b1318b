"""ctypes binding of the polygon oracle (oracle/libpolygon_oracle.so) + the reference's plane-matching loop on top of it.

TEST INFRASTRUCTURE ONLY: importable from tests/, bench.py's checker legs and profiles/ report scripts.  The product
package never imports this module.  What the oracle restates, what pins it (the reference's tests/test_polygons.cpp) and
what stays unpinned (FLANN's randomized search, Boost.Geometry) is in the header of oracle/polygon_oracle.cpp.
"""
import ctypes as C
import math
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libpolygon_oracle.so")

VALID, CONVEX_FALLBACK, SIMPLIFIED, THREW, NEEDS_DISSOLVE, HULL_FAILED, DISSOLVED = 1, 2, 4, 16, 32, 64, 128

_lib = None


def build(force=False):
    src = os.path.join(_HERE, "polygon_oracle.cpp")
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(src) > os.path.getmtime(_LIB_PATH):
        subprocess.check_call(["make", "-C", _HERE, "-s", "libpolygon_oracle.so"])
    return _LIB_PATH


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_LIB_PATH)
        vp, ip, dp = C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_double)
        L.polyref_build.argtypes = [vp, C.c_int, vp, vp, vp, C.c_int, ip, dp, vp, vp, ip, ip]
        L.polyref_concave_hull.argtypes = [vp, C.c_int, C.c_int, vp, C.c_int, ip, ip]
        L.polyref_area.argtypes = [vp, C.c_int]
        L.polyref_area.restype = C.c_double
        L.polyref_is_valid.argtypes = [vp, C.c_int]
        L.polyref_contains.argtypes = [vp, C.c_int, C.c_double, C.c_double]
        L.polyref_locate.argtypes = [vp, C.c_int, C.c_double, C.c_double]
        L.polyref_distance_outside.argtypes = [vp, C.c_int, C.c_double, C.c_double]
        L.polyref_distance_outside.restype = C.c_double
        L.polyref_inter_area.argtypes = [vp, C.c_int, vp, vp, vp, vp, C.c_int, vp, vp, vp]
        L.polyref_inter_area.restype = C.c_double
        L.polyref_rings_inter_area.argtypes = [vp, C.c_int, vp, C.c_int]
        L.polyref_rings_inter_area.restype = C.c_double
        L.polyref_move.argtypes = [C.c_int, vp, C.c_int, vp, vp, vp, vp, vp, vp, vp, C.c_int, ip, dp, vp, vp, vp, ip]
        for name in ("polyref_plane_to_camera", "polyref_plane_to_camera_analytic", "polyref_plane_to_world", "polyref_transform_from_quaternion"):
            getattr(L, name).argtypes = [vp, vp, vp]
            getattr(L, name).restype = None
        L.polyref_inverse44.argtypes = [vp, vp]
        L.polyref_inverse44.restype = None
        L.polyref_transformation_matrix.argtypes = [vp] * 7
        _lib = L
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def _f64(a, shape=None):
    a = np.ascontiguousarray(a, np.float64)
    return a.reshape(shape) if shape is not None else a


class Polygon:
    """utils::Polygon of the reference (src/utils/polygon.hpp): an open clockwise ring in a plane frame."""

    def __init__(self, ring, x_axis, y_axis, center, area=None, flags=0, k_used=0):
        self.ring = _f64(ring, (-1, 2))
        self.x_axis, self.y_axis, self.center = _f64(x_axis), _f64(y_axis), _f64(center)
        self.flags, self.k_used = int(flags), int(k_used)
        self.area = float(lib().polyref_area(_p(self.ring), len(self.ring))) if area is None else float(area)

    # Polygon(points, normal, center), polygon.cpp:168-229
    @classmethod
    def from_points(cls, points3, normal, center):
        pts, nrm, ctr = _f64(points3, (-1, 3)), _f64(normal), _f64(center)
        cap = max(4, len(pts) + 2)
        ring = np.zeros((cap, 2))
        xa, ya = np.zeros(3), np.zeros(3)
        cnt, flags, k = C.c_int(0), C.c_int(0), C.c_int(0)
        area = C.c_double(0)
        rc = lib().polyref_build(_p(pts), len(pts), _p(nrm), _p(ctr), _p(ring), cap, C.byref(cnt), C.byref(area), _p(xa), _p(ya),
                                 C.byref(flags), C.byref(k))
        assert rc == 0
        return cls(ring[: cnt.value].copy(), xa, ya, ctr, area.value, flags.value, k.value)

    @property
    def threw(self):
        return bool(self.flags & THREW)

    @property
    def valid(self):
        return bool(self.flags & VALID) if self.flags else bool(lib().polyref_is_valid(_p(self.ring), len(self.ring)))

    def is_valid(self):
        return bool(lib().polyref_is_valid(_p(self.ring), len(self.ring)))

    def boundary_length(self):
        return len(self.ring)

    def get_normal(self):
        return np.cross(self.x_axis, self.y_axis)

    def contains(self, x, y):  # boost within: strictly inside
        return bool(lib().polyref_contains(_p(self.ring), len(self.ring), float(x), float(y)))

    def locate(self, x, y):  # 1 inside, 0 on the outline, -1 outside
        return int(lib().polyref_locate(_p(self.ring), len(self.ring), float(x), float(y)))

    def distance_outside(self, x, y):  # 0 inside or on the outline, else the distance to it
        return float(lib().polyref_distance_outside(_p(self.ring), len(self.ring), float(x), float(y)))

    def simplify_reach(self):  # Polygon::simplify's threshold (polygon.cpp:582): how far Douglas-Peucker may leave a point out
        return max(self.area / 1e5, 10.0)

    def inter_area(self, other):  # polygon.cpp:525-545 (other is projected into this frame first)
        return float(lib().polyref_inter_area(_p(self.ring), len(self.ring), _p(self.x_axis), _p(self.y_axis), _p(self.center),
                                              _p(other.ring), len(other.ring), _p(other.x_axis), _p(other.y_axis), _p(other.center)))

    def union_area(self, other):  # polygon.cpp:547-560: area of the union = sum of the areas minus the intersection
        o = other.project(self.x_axis, self.y_axis, self.center)
        return self.area + o.area - self.inter_area(other)

    def _move(self, mode, a, b, c):
        cap = len(self.ring) + 2
        ring = np.zeros((cap, 2))
        xa, ya, ca = np.zeros(3), np.zeros(3), np.zeros(3)
        cnt, flags = C.c_int(0), C.c_int(0)
        area = C.c_double(0)
        a, b, c = _f64(a), (_f64(b) if b is not None else np.zeros(3)), (_f64(c) if c is not None else np.zeros(3))
        lib().polyref_move(mode, _p(self.ring), len(self.ring), _p(self.x_axis), _p(self.y_axis), _p(self.center), _p(a), _p(b), _p(c),
                           _p(ring), cap, C.byref(cnt), C.byref(area), _p(xa), _p(ya), _p(ca), C.byref(flags))
        return Polygon(ring[: cnt.value].copy(), xa, ya, ca, area.value, flags.value)

    def project(self, a, b, c=None):
        """project(normal, center) or project(x_axis, y_axis, center), polygon.cpp:338-382."""
        return self._move(0, a, None, b) if c is None else self._move(1, a, b, c)

    def transform(self, a, b, c=None):
        """transform(normal, center) or transform(x_axis, y_axis, center), polygon.cpp:384-428."""
        return self._move(2, a, None, b) if c is None else self._move(3, a, b, c)

    def to_camera_space(self, world_to_camera):  # polygon_coordinates.cpp:135-165
        return self._move(4, _f64(world_to_camera, (16,)), None, None)


def concave_hull(xy, k=0):
    """(ok, hull points as the walk left them, k used): ConcaveHull for one k, or the ladder of compute_concave_hull (k = 0)."""
    xy = _f64(xy, (-1, 2))
    out = np.zeros((len(xy) + 2, 2))
    cnt, used = C.c_int(0), C.c_int(0)
    ok = lib().polyref_concave_hull(_p(xy), len(xy), k, _p(out), len(out), C.byref(cnt), C.byref(used))
    return bool(ok), out[: cnt.value].copy(), used.value


def rings_inter_area(a, b):
    a, b = _f64(a, (-1, 2)), _f64(b, (-1, 2))
    return float(lib().polyref_rings_inter_area(_p(a), len(a), _p(b), len(b)))


def plane_to_camera(normal, d, world_to_camera, analytic=False):
    """PlaneWorldCoordinates::to_camera_coordinates (plane_coordinates.cpp:20-24): (normal, d) seen from the camera, through the
    plane matrix the reference builds with two 4x4 inversions (camera_transformation.cpp:62-71); analytic=True: the closed form of
    that matrix (the variant the product computes)."""
    plane = _f64(list(normal) + [d])
    out = np.zeros(4)
    fn = lib().polyref_plane_to_camera_analytic if analytic else lib().polyref_plane_to_camera
    fn(_p(plane), _p(_f64(world_to_camera, (16,))), _p(out))
    return out[:3].copy(), float(out[3])


def plane_to_world(normal, d, camera_to_world):
    """PlaneCameraCoordinates::to_world_coordinates (plane_coordinates.cpp:15-18) with compute_plane_camera_to_world_matrix."""
    plane = _f64(list(normal) + [d])
    out = np.zeros(4)
    lib().polyref_plane_to_world(_p(plane), _p(_f64(camera_to_world, (16,))), _p(out))
    return out[:3].copy(), float(out[3])


def inverse44(m):
    """matrix44::inverse() as the oracle restates it (Eigen's fixed-size cofactor form)."""
    out = np.zeros(16)
    lib().polyref_inverse44(_p(_f64(m, (16,))), _p(out))
    return out.reshape(4, 4)


def transform_from_quaternion(wxyz, position):
    """utils::get_transformation_matrix(quaternion, position) (camera_transformation.hpp:16-19); the quaternion as given."""
    out = np.zeros(16)
    lib().polyref_transform_from_quaternion(_p(_f64(wxyz, (4,))), _p(_f64(position, (3,))), _p(out))
    return out.reshape(4, 4)


def transformation_matrix(x_from, y_from, c_from, x_to, y_to, c_to):
    """get_transformation_matrix(xFrom, yFrom, centerFrom, xTo, yTo, centerTo) (point_coordinates.cpp:24-70); None where it throws."""
    out = np.zeros(16)
    ok = lib().polyref_transformation_matrix(*[_p(_f64(v, (3,))) for v in (x_from, y_from, c_from, x_to, y_to, c_to)], _p(out))
    return out.reshape(4, 4) if ok else None


# ---- MapPlane::find_matches (src/map_management/map_features/map_primitive.cpp:91-161) --------------------------------------
MAX_ANGLE_D = 20.0        # parameters::matching::maximumAngleForPlaneMatch_d, src/parameters.hpp:92-93
MAX_DISTANCE_MM = 100.0   # maximumDistanceForPlaneMatch_mm, :94-95
MIN_OVERLAP = float(np.float32(0.4))  # minimumPlaneOverlapToConsiderMatch (a float constant), :90-91


def find_matches(map_planes, detected, world_to_camera=None, advanced=False, allow_index0=False):
    """The reference's matching loop as Feature_Map::get_matches drives it (feature_map.hpp:647-670): every map plane in turn
    calls MapPlane::find_matches against the detected planes, a detected plane can be taken once.

    map_planes / detected: lists of (normal[3], d, Polygon); the map planes live in the "world" = the previous camera frame,
    world_to_camera = 4x4 (None: identity).  Returns (match[j] = detected index or -1, inter[j][i] = area or -1 where the
    distance / normal gates reject the pair)."""
    T = np.eye(4) if world_to_camera is None else _f64(world_to_camera, (4, 4))
    min_cos = abs(math.cos(MAX_ANGLE_D * math.pi / 180.0))
    thr = MIN_OVERLAP / 2 if advanced else MIN_OVERLAP
    matched = [False] * len(detected)
    match = [-1] * len(map_planes)
    inter = np.full((len(map_planes), len(detected)), -1.0)
    for j, (mn, md, mpoly) in enumerate(map_planes):
        pn, pd = plane_to_camera(mn, md, T)
        ppoly = mpoly.to_camera_space(T)
        if ppoly.area <= 0.0:
            continue
        selected, greatest = -1, 0.0
        for i, (dn, dd, dpoly) in enumerate(detected):
            if matched[i]:
                continue
            cos = (dn[0] * pn[0] + dn[1] * pn[1]) + dn[2] * pn[2]  # get_cos_angle, plane_coordinates.hpp:60-63
            if not (abs(dd - pd) < MAX_DISTANCE_MM) or not (abs(cos) > min_cos):
                continue
            ia = dpoly.inter_area(ppoly)
            inter[j, i] = ia
            if ia > greatest and ia / dpoly.area >= thr:
                selected, greatest = i, ia
        if selected < 0 or (selected == 0 and not allow_index0):  # the reference's `selectedIndex <= 0` (map_primitive.cpp:146)
            continue
        match[j] = selected
        matched[selected] = True
    return match, inter
