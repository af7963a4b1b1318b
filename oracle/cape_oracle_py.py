"""ctypes binding of the CPU oracle (oracle/libcape_oracle.so).

TEST INFRASTRUCTURE ONLY: importable from tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.
The product package never imports this module.  Parity status of the oracle itself: **parity unpinned**
(see oracle/cape_oracle.hpp).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libcape_oracle.so")


def build(force=False):
    """Compile the oracle with the committed Makefile (g++ only)."""
    if force or not os.path.exists(_LIB_PATH) or any(
        os.path.getmtime(os.path.join(_HERE, f)) > os.path.getmtime(_LIB_PATH)
        for f in ("cape_oracle.cpp", "cape_oracle.hpp")
    ):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return _LIB_PATH


_libs = {}


def lib(path=None):
    """The oracle library (default build), or a variant build at `path` (oracle/variants.py: risk measurement only)."""
    key = path or _LIB_PATH
    if key not in _libs:
        if path is None:
            build()
        L = C.CDLL(key)
        L.cape_oracle_create.restype = C.c_void_p
        L.cape_oracle_create.argtypes = [C.c_int, C.c_int, C.c_double, C.c_double, C.c_double, C.c_double, C.c_int]
        L.cape_oracle_destroy.argtypes = [C.c_void_p]
        L.cape_oracle_set_rng_seed.argtypes = [C.c_void_p, C.c_uint]
        L.cape_oracle_cells.argtypes = [C.c_void_p]
        L.cape_oracle_run.argtypes = [C.c_void_p, C.c_void_p]
        L.cape_oracle_run_many.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
        L.cape_oracle_get_cloud.argtypes = [C.c_void_p, C.c_void_p]
        L.cape_oracle_rectify.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.cape_oracle_get_cell_stats.argtypes = [C.c_void_p] + [C.c_void_p] * 10
        L.cape_oracle_get_labels.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.cape_oracle_num_seeds.argtypes = [C.c_void_p]
        L.cape_oracle_get_seeds.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.cape_oracle_log_counts.argtypes = [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int)]
        L.cape_oracle_num_plane_segments.argtypes = [C.c_void_p]
        L.cape_oracle_get_plane_segments.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.cape_oracle_num_planes.argtypes = [C.c_void_p]
        L.cape_oracle_get_planes.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.cape_oracle_get_boundary.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        L.cape_oracle_num_cylinders.argtypes = [C.c_void_p]
        L.cape_oracle_get_cylinders.argtypes = [C.c_void_p, C.c_void_p]
        L.cape_oracle_depth_quantization.restype = C.c_double
        L.cape_oracle_depth_quantization.argtypes = [C.c_double]
        L.cape_oracle_eigen3.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.cape_oracle_mt19937_double.restype = C.c_double
        L.cape_oracle_mt19937_double.argtypes = [C.c_uint, C.c_int]
        L.cape_oracle_back_project.argtypes = [C.c_void_p, C.c_double, C.c_double, C.c_double, C.c_void_p]
        L.cape_oracle_ransac_max_iterations.restype = C.c_uint
        L.cape_oracle_cos_merge_angle.restype = C.c_double
        L.cape_oracle_sin_merge_angle.restype = C.c_float
        _libs[key] = L
    return _libs[key]


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


class OracleResult:
    """All observables of one frame (SURVEY.md 8a 'parity observables')."""

    pass


class Oracle:
    def __init__(self, width=640, height=480, fx=550.0, fy=550.0, cx=320.0, cy=240.0, cylinders=True, lib_path=None):
        self.L = lib(lib_path)
        self.width, self.height = width, height
        self.h = self.L.cape_oracle_create(width, height, fx, fy, cx, cy, 1 if cylinders else 0)
        self.cells = self.L.cape_oracle_cells(self.h)

    def set_rng_seed(self, seed):
        """utils::Random::_seed of a reference build without MAKE_DETERMINISTIC (random.hpp:59-64): every frame restarts there."""
        self.L.cape_oracle_set_rng_seed(self.h, int(seed) & 0xFFFFFFFF)

    def __del__(self):
        try:
            if self.h:
                self.L.cape_oracle_destroy(self.h)
                self.h = None
        except Exception:
            pass

    def run_many(self, depth_frames):
        """Timing loop used by bench.py cpu_baseline: no Python between frames."""
        d = np.ascontiguousarray(depth_frames, dtype=np.float32)
        n = d.shape[0]
        tot = C.c_longlong(0)
        self.L.cape_oracle_run_many(self.h, _p(d), n, C.byref(tot))
        return tot.value

    def run(self, depth):
        d = np.ascontiguousarray(depth, dtype=np.float32)
        assert d.shape == (self.height, self.width)
        self.L.cape_oracle_run(self.h, _p(d))
        r = OracleResult()
        n = self.cells
        r.n = np.zeros(n, np.uint32)
        r.planar = np.zeros(n, np.uint8)
        r.sums = np.zeros((n, 9), np.float64)
        r.centroid = np.zeros((n, 3), np.float64)
        r.normal = np.zeros((n, 3), np.float64)
        r.d = np.zeros(n, np.float64)
        r.mse = np.zeros(n, np.float64)
        r.score = np.zeros(n, np.float64)
        r.tol = np.zeros(n, np.float32)
        r.bins = np.zeros(n, np.int32)
        self.L.cape_oracle_get_cell_stats(self.h, _p(r.n), _p(r.planar), _p(r.sums), _p(r.centroid), _p(r.normal),
                                          _p(r.d), _p(r.mse), _p(r.score), _p(r.tol), _p(r.bins))
        r.plane_labels = np.zeros(n, np.int32)
        r.cyl_labels = np.zeros(n, np.int32)
        self.L.cape_oracle_get_labels(self.h, _p(r.plane_labels), _p(r.cyl_labels))
        ns = self.L.cape_oracle_num_seeds(self.h)
        r.seeds = np.zeros(ns, np.int32)
        r.seed_outcome = np.zeros(ns, np.int32)
        r.seed_activated = np.zeros(ns, np.uint32)
        if ns:
            self.L.cape_oracle_get_seeds(self.h, _p(r.seeds), _p(r.seed_outcome), _p(r.seed_activated))
        a, b = C.c_int(0), C.c_int(0)
        self.L.cape_oracle_log_counts(self.h, C.byref(a), C.byref(b))
        r.log_invalid_seed, r.log_not_planar_after_merge = a.value, b.value  # the reference's log lines, counted
        P = self.L.cape_oracle_num_plane_segments(self.h)
        r.segments = np.zeros((P, 20), np.float64)
        r.merge_labels = np.zeros(P, np.uint32)
        if P:
            self.L.cape_oracle_get_plane_segments(self.h, _p(r.segments), _p(r.merge_labels))
        Q = self.L.cape_oracle_num_planes(self.h)
        r.planes = np.zeros((Q, 20), np.float64)
        nb = np.zeros(Q, np.int32)
        if Q:
            self.L.cape_oracle_get_planes(self.h, _p(r.planes), _p(nb))
        r.boundary = []
        for i in range(Q):
            b = np.zeros((nb[i], 3), np.float64)
            self.L.cape_oracle_get_boundary(self.h, i, _p(b))
            r.boundary.append(b)
        K = self.L.cape_oracle_num_cylinders(self.h)
        r.cylinders = np.zeros((K, 4), np.float64)
        if K:
            self.L.cape_oracle_get_cylinders(self.h, _p(r.cylinders))
        return r

    def rectify(self, depth, T):
        d = np.ascontiguousarray(depth, np.float32)
        T = np.ascontiguousarray(T, np.float64).reshape(16)
        out = np.zeros_like(d)
        self.L.cape_oracle_rectify(self.h, _p(d), _p(T), _p(out))
        return out

    def cloud(self):
        c = np.zeros((3, self.width * self.height), np.float32)
        self.L.cape_oracle_get_cloud(self.h, _p(c))
        return c

    def back_project(self, col, row, z):
        out = np.zeros(3, np.float64)
        self.L.cape_oracle_back_project(self.h, float(col), float(row), float(z), _p(out))
        return out


def depth_quantization(z):
    return lib().cape_oracle_depth_quantization(float(z))


def eigen3(m):
    m = np.ascontiguousarray(m, np.float64)
    ev = np.zeros(3, np.float64)
    vec = np.zeros((3, 3), np.float64)
    it = C.c_int(0)
    lib().cape_oracle_eigen3(_p(m), _p(ev), _p(vec), C.byref(it))
    return ev, vec, it.value


def mt19937_double(seed, index):
    return lib().cape_oracle_mt19937_double(seed, index)
