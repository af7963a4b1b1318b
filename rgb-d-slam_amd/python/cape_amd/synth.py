"""Seeded synthetic 640x480 / 1280x960 depth streams: room, tumlike, tunnel.

The datasets the reference is run on (TUM RGB-D, CAPE 'tunnel') are not in the image, so the bench and the
parity tests use these stand-ins.  Everything is float64 numpy arithmetic restricted to + - * / sqrt and a
counter-based RNG read through Philox.random_raw, so a (scene, seed, frame) triple names the same bytes on
every box; tests/golden stores a SHA-256 of each generated input next to the expected outputs.

Units: depth in millimetres, float32, row-major H x W, 0 = invalid (what Depth_Map_Transformation::
get_organized_cloud_array receives, reference src/features/primitives/depth_map_transformation.cpp:89).
"""
import hashlib

import numpy as np

DEFAULT_INTRINSICS = dict(fx=550.0, fy=550.0, cx=320.0, cy=240.0)  # reference src/parameters.cpp:59-74
TUM_FR1_INTRINSICS = dict(fx=517.3, fy=516.5, cx=318.6, cy=255.3)  # reference examples/configuration_example.yaml:13-17


def _quant(z):
    # same model as reference src/utils/covariances.cpp:12-19 (used only to size the synthetic noise)
    return np.maximum(-0.53 + 0.74e-3 * z + 2.73e-6 * z * z, 0.5)


class _Rng:
    """Uniform / approximately normal variates from raw Philox words (bit-stable across numpy versions)."""

    def __init__(self, *key):
        k = np.array([int(x) & 0xFFFFFFFFFFFFFFFF for x in key], dtype=np.uint64)
        kk = np.zeros(2, np.uint64)
        kk[: min(2, k.size)] = k[:2]
        if k.size > 2:
            kk[1] = kk[1] * np.uint64(1000003) + k[2]
        self.bg = np.random.Philox(key=kk)

    def uniform(self, shape):
        n = int(np.prod(shape))
        raw = self.bg.random_raw(n)
        return ((raw >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)).reshape(shape)

    def normal(self, shape):
        # Irwin-Hall(4) rescaled to unit variance: only exact arithmetic, no libm
        n = int(np.prod(shape))
        raw = self.bg.random_raw(2 * n)
        lo = (raw & np.uint64(0xFFFFFFFF)).astype(np.float64) * (1.0 / 4294967296.0)
        hi = (raw >> np.uint64(32)).astype(np.float64) * (1.0 / 4294967296.0)
        s = (lo[:n] + hi[:n] + lo[n:] + hi[n:]) - 2.0
        return (s * np.sqrt(3.0)).reshape(shape)


def _rotation(ty, tp):
    """R = Ry(yaw) * Rx(pitch) from half-angle tangents (no trig calls)."""
    cy, sy = (1 - ty * ty) / (1 + ty * ty), 2 * ty / (1 + ty * ty)
    cp, sp = (1 - tp * tp) / (1 + tp * tp), 2 * tp / (1 + tp * tp)
    Ry = np.array([[cy, 0, sy], [0, 1, 0], [-sy, 0, cy]])
    Rx = np.array([[1, 0, 0], [0, cp, -sp], [0, sp, cp]])
    return Ry @ Rx


def _rays(W, H, fx, fy, cx, cy, R):
    u = (np.arange(W, dtype=np.float64) - cx) / fx
    v = (np.arange(H, dtype=np.float64) - cy) / fy
    X, Y = np.meshgrid(u, v)
    d = np.stack([X, Y, np.ones_like(X)], axis=-1)
    return d @ R.T  # world-frame direction of each pixel ray; camera-frame z of a hit at parameter t is t


def _hit_plane(o, dw, n, c):
    """Ray parameter of o + t*dw with plane n.p = c (inf where no forward hit)."""
    den = dw @ n
    num = c - o @ n
    with np.errstate(divide="ignore", invalid="ignore"):
        t = num / den
    return np.where((den != 0) & (t > 0), t, np.inf)


def _hit_box_inside(o, dw, lo, hi):
    """Camera inside an axis-aligned box: first wall hit."""
    t = np.full(dw.shape[:2], np.inf)
    for ax in range(3):
        n = np.zeros(3)
        n[ax] = 1.0
        t = np.minimum(t, _hit_plane(o, dw, n, lo[ax]))
        t = np.minimum(t, _hit_plane(o, dw, n, hi[ax]))
    return t


def _hit_box_outside(o, dw, lo, hi):
    """Slab test for a box seen from outside (inf where missed)."""
    with np.errstate(divide="ignore", invalid="ignore"):
        t0 = (lo - o) / dw
        t1 = (hi - o) / dw
    tn = np.minimum(t0, t1).max(axis=-1)
    tf = np.maximum(t0, t1).min(axis=-1)
    return np.where((tn <= tf) & (tn > 0), tn, np.inf)


def _hit_cylinder_inside(o, dw, p0, a, r):
    """Infinite cylinder of radius r around the line p0 + s*a (|a|=1), camera inside: far root."""
    oc = o - p0
    dperp = dw - (dw @ a)[..., None] * a
    operp = oc - (oc @ a) * a
    A = (dperp * dperp).sum(-1)
    B = 2.0 * (dperp @ operp)
    Cc = operp @ operp - r * r
    disc = B * B - 4 * A * Cc
    with np.errstate(divide="ignore", invalid="ignore"):
        t = (-B + np.sqrt(np.maximum(disc, 0.0))) / (2 * A)
    return np.where((disc >= 0) & (A > 0) & (t > 0), t, np.inf)


def _finish(t, rng, mode, hole_frac, noise_scale=0.5):
    """noise sigma = noise_scale*quant(z), sensor quantisation, random holes -> float32 mm."""
    z = np.where(np.isfinite(t), t, 0.0)
    valid = z > 0
    z = z + noise_scale * _quant(z) * rng.normal(z.shape)
    if mode == "mm":  # integer millimetres (CAPE / generic sensor)
        out = np.rint(z).astype(np.float32)
    elif mode == "tum":  # raw uint16 = round(5000 * metres); float mm = raw * (1/5) in f32 (cv convertTo)
        raw = np.clip(np.rint(5.0 * z), 0, 65535).astype(np.uint16)
        out = raw.astype(np.float32) * np.float32(1.0 / 5.0)
    else:
        raise ValueError(mode)
    holes = rng.uniform(z.shape) < hole_frac
    out[holes | ~valid | (out <= 0)] = 0.0
    return out


def _tri(x):
    """Triangle wave in [-1,1] with period 1 (smooth-enough seeded trajectories without trig)."""
    f = x - np.floor(x)
    return 4.0 * np.abs(f - 0.5) - 1.0


def room(seed=0, frame=0, width=640, height=480, intr=None):
    """Planar box room 4 x 3 x 5 m seen from a slowly panning camera (BASELINE.json configs[1])."""
    s = width / 640.0
    intr = intr or {k: v * s for k, v in DEFAULT_INTRINSICS.items()}
    rng0 = _Rng(0x726F6F6D, seed)
    ph = rng0.uniform((4,))
    ty = 0.18 * _tri(ph[0] + frame / 257.0) + 0.07
    tp = 0.08 * _tri(ph[1] + frame / 181.0) - 0.03
    o = np.array([300.0 * _tri(ph[2] + frame / 409.0), -150.0 + 100.0 * _tri(ph[3] + frame / 331.0), 0.0])
    R = _rotation(ty, tp)
    dw = _rays(width, height, intr["fx"], intr["fy"], intr["cx"], intr["cy"], R)
    t = _hit_box_inside(o, dw, np.array([-2000.0, -1500.0, -1500.0]), np.array([2000.0, 1500.0, 3500.0]))
    return _finish(t, _Rng(0x726F6F6D, seed, frame + 1), "mm", 0.02)


def tumlike(seed=1, frame=0, width=640, height=480, intr=None):
    """Desk-like Kinect frame: floor, tilted back wall, side wall, a 0.6 x 0.4 m box (configs[0], [3])."""
    s = width / 640.0
    intr = intr or {k: v * s for k, v in TUM_FR1_INTRINSICS.items()}
    rng0 = _Rng(0x74756D, seed)
    ph = rng0.uniform((4,))
    ty = 0.10 * _tri(ph[0] + frame / 193.0) + 0.105  # ~12 deg yaw: walls never fronto-parallel
    tp = 0.05 * _tri(ph[1] + frame / 149.0) - 0.12   # looking slightly down (y axis points down)
    o = np.array([100.0 * _tri(ph[2] + frame / 233.0), -400.0, 0.0])
    R = _rotation(ty, tp)
    dw = _rays(width, height, intr["fx"], intr["fy"], intr["cx"], intr["cy"], R)
    t = _hit_plane(o, dw, np.array([0.0, 1.0, 0.0]), 600.0)            # floor 1 m below the camera (y down)
    t = np.minimum(t, _hit_plane(o, dw, np.array([0.0, 0.0, 1.0]), 2500.0))  # back wall
    t = np.minimum(t, _hit_plane(o, dw, np.array([1.0, 0.0, 0.0]), 1600.0))  # side wall
    t = np.minimum(t, _hit_box_outside(o, dw, np.array([-500.0, 200.0, 1300.0]), np.array([100.0, 600.0, 1700.0])))
    z = _finish(t, _Rng(0x74756D, seed, frame + 1), "tum", 0.08)
    z[:, : int(12 * s)] = 0.0  # Kinect shadow band
    return z


def tunnel(seed=0, frame=0, width=640, height=480, intr=None):
    """Cylinder of radius 1.2 m around a slightly tilted axis with a flat floor chord (configs[2])."""
    s = width / 640.0
    intr = intr or {k: v * s for k, v in DEFAULT_INTRINSICS.items()}
    rng0 = _Rng(0x74756E6E, seed)
    ph = rng0.uniform((4,))
    ty = 0.06 * _tri(ph[0] + frame / 211.0) + 0.02
    tp = 0.04 * _tri(ph[1] + frame / 173.0) - 0.01
    o = np.array([150.0 * _tri(ph[2] + frame / 307.0), 100.0 * _tri(ph[3] + frame / 263.0), 0.0])
    R = _rotation(ty, tp)
    dw = _rays(width, height, intr["fx"], intr["fy"], intr["cx"], intr["cy"], R)
    a = np.array([0.05, 0.03, 1.0])
    a = a / np.sqrt(a @ a)
    t = _hit_cylinder_inside(o, dw, np.zeros(3), a, 1200.0)
    t = np.minimum(t, _hit_plane(o, dw, np.array([0.0, 1.0, 0.0]), 900.0))  # floor chord
    t = np.where(t > 8000.0, np.inf, t)  # sensor range
    return _finish(t, _Rng(0x74756E6E, seed, frame + 1), "mm", 0.02)


def facets(seed=0, frame=0, width=640, height=480, intr=None, n_planes=9, max_tilt=1.1):
    """Faceted surface (lower envelope of random planes) + a few foreground slabs: creases of every angle, depth jumps.

    Stress input for region growing / plane merging: neighbouring facets under and over the 18 degree merge angle,
    facets smaller than the 4-cell activation threshold, regions that touch after being grown from different seeds,
    low-score regions that go to the cylinder branch."""
    s = width / 640.0
    intr = intr or {k: v * s for k, v in DEFAULT_INTRINSICS.items()}
    rng = _Rng(0x6661636574, seed, frame + 1)
    u = (np.arange(width, dtype=np.float64) - intr["cx"]) / intr["fx"]
    v = (np.arange(height, dtype=np.float64) - intr["cy"]) / intr["fy"]
    X, Y = np.meshgrid(u, v)
    par = rng.uniform((n_planes, 3))
    t = np.full((height, width), np.inf)
    for k in range(n_planes):
        nx, ny = max_tilt * (2 * par[k, 0] - 1), max_tilt * (2 * par[k, 1] - 1)
        d = 1800.0 + 2200.0 * par[k, 2]
        den = 1.0 + nx * X + ny * Y  # plane: z * (1 + nx x + ny y) = d
        with np.errstate(divide="ignore", invalid="ignore"):
            z = d / den
        t = np.minimum(t, np.where(den > 0.05, z, np.inf))
    # foreground slabs: tilted rectangles in front of the envelope (depth discontinuities, small regions)
    slab = rng.uniform((4, 7))
    for k in range(4):
        x0, y0 = int(slab[k, 0] * width * 0.8), int(slab[k, 1] * height * 0.8)
        w, h = int((0.08 + 0.25 * slab[k, 2]) * width), int((0.08 + 0.25 * slab[k, 3]) * height)
        nx, ny = 0.6 * (2 * slab[k, 4] - 1), 0.6 * (2 * slab[k, 5] - 1)
        d = 900.0 + 700.0 * slab[k, 6]
        den = 1.0 + nx * X + ny * Y
        z = d / np.maximum(den, 0.05)
        region = np.zeros_like(t, dtype=bool)
        region[y0:y0 + h, x0:x0 + w] = True
        t = np.where(region & (z < t), z, t)
    t = np.where(t > 9000.0, np.inf, t)
    return _finish(t, rng, "mm", 0.03)


SCENES = {"room": room, "tumlike": tumlike, "tunnel": tunnel, "facets": facets}


def stream(scene, seed, n_frames, width=640, height=480, start=0):
    f = SCENES[scene]
    return np.stack([f(seed=seed, frame=start + i, width=width, height=height) for i in range(n_frames)])


def sha256(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()
