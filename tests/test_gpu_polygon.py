"""N1 on the device (cape_build_polygons): every polygon against the host class of this repo, vertex for vertex.

The host class (rgb-d-slam_amd/host/boundary_polygon.cpp, reached through the test hook cape_host_polygon of
libcape_primitives.so) is the checker: it replays the reference's polygon contract (tests/host/test_polygon.cpp after the
reference's tests/test_polygons.cpp) and the device kernel runs the same statements -- plane frame, projection, sort +
duplicate removal, k-nearest-neighbours hull on the k ladder, convex fallback, Douglas-Peucker -- with + - x / sqrt only,
so the bar is equality of bits: vertex count, every vertex, area, both axes, validity."""
import ctypes as C
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def host_poly(host_binaries):
    import cape_amd

    cape_amd.load_library()  # torch's HIP runtime, then libcape_hip, BEFORE the host library pulls libcape_hip in: one runtime per process
    lib = C.CDLL(os.path.join(host_binaries, "libcape_primitives.so"))
    vp = C.c_void_p
    lib.cape_host_polygon.argtypes = [vp, C.c_int, vp, vp, vp, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_double), vp, vp, C.POINTER(C.c_int)]

    def run(points3, normal, center):
        pts = np.ascontiguousarray(points3, np.float64).reshape(-1, 3)
        nrm = np.ascontiguousarray(normal, np.float64)
        ctr = np.ascontiguousarray(center, np.float64)
        ring = np.zeros((max(1, len(pts)), 2), np.float64)
        cnt, valid, area = C.c_int(0), C.c_int(0), C.c_double(0)
        xa, ya = np.zeros(3), np.zeros(3)
        rc = lib.cape_host_polygon(pts.ctypes.data_as(vp), len(pts), nrm.ctypes.data_as(vp), ctr.ctypes.data_as(vp), ring.ctypes.data_as(vp),
                                   len(ring), C.byref(cnt), C.byref(area), xa.ctypes.data_as(vp), ya.ctypes.data_as(vp), C.byref(valid))
        return dict(threw=rc != 0, ring=ring[: cnt.value], area=area.value, x_axis=xa, y_axis=ya, valid=bool(valid.value))

    return run


def _center(s):
    """Plane_Segment::get_center() (plane_segment.hpp:90 -> plane_coordinates.hpp:52): normal * (-d), the origin the reference
    gives its polygons (primitive_detection.cpp:622) -- NOT the centroid."""
    return np.asarray(s["normal"], np.float64) * (-np.float64(s["d"]))


def _bits(a):
    return np.ascontiguousarray(a, np.float64).view(np.uint64)


def _same(pol, verts, ref, what):
    import cape_amd

    if ref["threw"]:
        assert pol["flags"] & cape_amd.POLY_REJECTED, what
        return
    assert int(pol["vertex_count"]) == len(ref["ring"]), f"{what}: vertex count {int(pol['vertex_count'])} != {len(ref['ring'])}"
    assert np.array_equal(_bits(verts), _bits(ref["ring"])), f"{what}: vertices differ"
    assert _bits(pol["area"]) == _bits(ref["area"]), f"{what}: area"
    assert np.array_equal(_bits(pol["x_axis"]), _bits(ref["x_axis"])) and np.array_equal(_bits(pol["y_axis"]), _bits(ref["y_axis"])), what
    assert bool(pol["flags"] & cape_amd.POLY_VALID) == ref["valid"], f"{what}: validity"


@pytest.mark.parametrize("scene,cyl,n", [("room", False, 24), ("tumlike", True, 24), ("tunnel", True, 8)])
def test_polygons_of_extracted_planes(host_poly, scene, cyl, n):
    """Every output plane of device-rendered frames: the polygon built on the device from the plane's boundary candidates ==
    the host class fed with the same points, normal and centre."""
    import torch
    from cape_amd import Extractor, synth, synth_gpu

    intr = synth.TUM_FR1_INTRINSICS if scene == "tumlike" else synth.DEFAULT_INTRINSICS
    dev = synth_gpu.stream(scene, 55, n, start=300, device="cuda", chunk=8)
    ex = Extractor(640, 480, cylinders=cyl, max_batch=n, **intr)
    st = torch.cuda.current_stream().cuda_stream
    ex.extract_device(dev.data_ptr(), n, st)
    ex.build_polygons(n, st)
    res = ex.results(n)
    pol, ver = ex.polygons(n)
    planes = simplified = 0
    for f in range(n):
        segs = res.segments(f)
        for i, s in enumerate(segs):
            p = pol[f, i]
            if not s["is_output"]:
                assert p["flags"] == 0 and p["vertex_count"] == 0
                continue
            pts = res.boundary_points(f, s)
            ref = host_poly(pts, s["normal"], _center(s))
            o, c = int(p["vertex_offset"]), int(p["vertex_count"])
            assert o == int(s["boundary_offset"]) and p["segment"] == i
            _same(p, ver[f, o:o + c], ref, f"{scene} frame {f} segment {i} ({len(pts)} points)")
            assert np.array_equal(_bits(p["center"]), _bits(_center(s)))
            planes += 1
            simplified += int(bool(p["flags"] & 4))
    assert planes >= n, "the streams show planes"
    assert simplified > 0, "Douglas-Peucker must have replaced at least one ring"
    # results of the extraction are untouched by the polygon pass
    again = ex.results(n)
    assert again.records.tobytes() == res.records.tobytes()
    ex.close()


def test_polygon_shapes_and_degenerate_inputs(host_poly):
    """Point sets the scenes do not produce: concave outlines (L, U, star), duplicates, collinear points (no hull: the
    convex fallback degenerates as on the host), a dense disc (k ladder beyond 3), tilted plane frames, too few points,
    a normal that is not unit."""
    from cape_amd import Extractor, synth

    rng = np.random.default_rng(5)
    ex = Extractor(640, 480, max_batch=1, **synth.DEFAULT_INTRINSICS)

    def lift(xy, normal, center):
        # put 2-D samples on the plane through `center` with the given normal (any two spanning vectors do)
        nrm = np.asarray(normal, float)
        a = np.cross(nrm, [0.3, -0.5, 0.8])
        a /= np.linalg.norm(a)
        b = np.cross(nrm, a)
        return np.asarray(center) + xy[:, :1] * a + xy[:, 1:] * b

    def grid(mask_fn, step=40.0, n=18):
        g = np.array([(i * step, j * step) for i in range(n) for j in range(n) if mask_fn(i, j)], float)
        return g + rng.normal(0, 1.5, g.shape)

    shapes = {
        "square": grid(lambda i, j: True, n=10),
        "L": grid(lambda i, j: i < 6 or j < 6),
        "U": grid(lambda i, j: not (5 <= i < 13 and j >= 6)),
        "ring": grid(lambda i, j: (i - 8.5) ** 2 + (j - 8.5) ** 2 > 16),
        "star": np.array([(r * np.cos(t), r * np.sin(t)) for k, t in enumerate(np.linspace(0, 2 * np.pi, 40, endpoint=False))
                          for r in ((300, 120)[k % 2] * np.linspace(0.2, 1, 5))]),
        "disc": rng.normal(0, 200, (600, 2)),
        "duplicates": np.repeat(grid(lambda i, j: True, n=6), 3, axis=0),
        "collinear": np.stack([np.linspace(0, 900, 30), np.linspace(0, 900, 30) * 0.5], 1),
        "three": np.array([(0, 0), (100, 0), (0, 100.0)]),
        "two": np.array([(0, 0), (100, 0.0)]),
        "thin": np.stack([np.linspace(0, 2000, 80), rng.normal(0, 0.5, 80)], 1),
    }
    normals = [(0, 0, 1.0), (0, 0.6, 0.8), (0.48, 0.6, 0.64), (1.0, 0, 0), (0.7071067811865476, 0.7071067811865475, 0)]
    fallbacks = 0
    for name, xy in shapes.items():
        for nrm in normals:
            nrm = np.asarray(nrm) / np.linalg.norm(nrm)
            center = np.array([120.0, -340.0, 2100.0])
            pts = lift(np.asarray(xy, float), nrm, center)
            ref = host_poly(pts, nrm, center)
            pol, verts = ex.debug_polygon(pts, nrm, center)
            _same(pol, verts, ref, f"{name} normal {nrm}")
            fallbacks += int(bool(pol["flags"] & 2))
    assert fallbacks > 0, "at least one shape must take the convex-hull fallback"
    # a normal that is not unit: the host constructor throws, the device says REJECTED
    pts = lift(shapes["square"], np.array([0, 0, 1.0]), np.zeros(3))
    pol, _ = ex.debug_polygon(pts, [0, 0, 1.1], [0, 0, 0])
    assert pol["flags"] & 16 and host_poly(pts, [0, 0, 1.1], [0, 0, 0])["threw"]
    ex.close()


def test_polygon_random_point_sets_property(host_poly):
    """Randomised: clustered, gridded and noisy point sets of 3 to 700 points on random planes."""
    from cape_amd import Extractor, synth

    rng = np.random.default_rng(77)
    ex = Extractor(640, 480, max_batch=1, **synth.DEFAULT_INTRINSICS)
    for trial in range(60):
        n = int(rng.integers(3, 700))
        kind = trial % 3
        if kind == 0:
            xy = rng.uniform(-800, 800, (n, 2))
        elif kind == 1:
            xy = np.round(rng.uniform(-20, 20, (n, 2))) * 45.0 + rng.normal(0, 2.0, (n, 2))  # a cell grid seen at an angle
        else:
            c = rng.uniform(-600, 600, (4, 2))
            xy = c[rng.integers(0, 4, n)] + rng.normal(0, 90, (n, 2))
        nrm = rng.normal(0, 1, 3)
        nrm /= np.linalg.norm(nrm)
        nrm /= np.linalg.norm(nrm)
        center = rng.uniform(-500, 500, 3) + [0, 0, 2000]
        a = np.cross(nrm, [0.2, 0.9, -0.4])
        a /= np.linalg.norm(a)
        b = np.cross(nrm, a)
        pts = center + xy[:, :1] * a + xy[:, 1:] * b
        ref = host_poly(pts, nrm, center)
        pol, verts = ex.debug_polygon(pts, nrm, center)
        _same(pol, verts, ref, f"trial {trial} ({n} points, kind {kind})")
    ex.close()


def test_polygons_1280x960_and_one_frame_handle(host_poly):
    """The 64x48 grid (planes with several hundred boundary candidates: the 1 024-point instance on real data) and a
    one-frame handle, whose records, boundary points and polygons live in pinned host memory."""
    import torch
    import cape_amd
    from cape_amd import Extractor, synth, synth_gpu

    intr = {k: v * 2.0 for k, v in synth.DEFAULT_INTRINSICS.items()}
    n = 6
    dev = synth_gpu.stream("room", 91, n, width=1280, height=960, start=40, device="cuda", chunk=2)
    st = torch.cuda.current_stream().cuda_stream
    big = 0
    for max_batch, frames in ((n, n), (1, 1)):
        ex = Extractor(1280, 960, cylinders=True, max_batch=max_batch, **intr)
        ex.extract_device(dev.data_ptr(), frames, st)
        ex.build_polygons(frames, st)
        res = ex.results(frames)
        pol, ver = ex.polygons(frames)
        for f in range(frames):
            for i, s in enumerate(res.segments(f)):
                if not s["is_output"]:
                    continue
                p = pol[f, i]
                pts = res.boundary_points(f, s)
                if len(pts) > 1024:
                    assert p["flags"] & cape_amd.POLY_OVERFLOW
                    continue
                big += int(len(pts) > 256)
                o, c = int(p["vertex_offset"]), int(p["vertex_count"])
                _same(p, ver[f, o:o + c], host_poly(pts, s["normal"], _center(s)), f"1280x960 frame {f} segment {i} ({len(pts)} points)")
        ex.close()
    assert big > 0, "the wide grid must produce planes beyond the small instance's 256 points"


def test_polygon_pass_without_planes_terminates():
    """The task kernel's waiting waves are sent home by the wave that finishes the last plane: a batch WITHOUT any plane (empty
    frames), and one-frame / few-plane batches, must return all the same."""
    import torch
    from cape_amd import Extractor, synth, synth_gpu

    st = torch.cuda.current_stream().cuda_stream
    for n in (1, 5, 64):
        empty = torch.zeros((n, 480, 640), dtype=torch.float32, device="cuda")
        ex = Extractor(640, 480, cylinders=True, max_batch=max(n, 9), **synth.DEFAULT_INTRINSICS)
        ex.extract_device(empty.data_ptr(), n, st)
        ex.build_polygons(n, st)
        ex.match_polygons(n, 0, st)
        pol, _ = ex.polygons(n)
        assert not pol["flags"].any() and not pol["vertex_count"].any()
        # the same handle, next batch: one frame with planes among empty ones
        one = synth_gpu.stream("room", 8, 1, start=3, device="cuda", chunk=1)
        mixed = empty.clone()
        mixed[n // 2] = one[0]
        ex.extract_device(mixed.data_ptr(), n, st)
        ex.build_polygons(n, st)
        pol, _ = ex.polygons(n)
        assert (pol["flags"][n // 2] & 1).any() and not np.delete(pol["flags"], n // 2, axis=0).any()
        ex.close()


def _ragged_batch(n, seed=3):
    """Device-rendered room frames with dropped pixels, noise and dropped blocks: ragged outlines whose rung 0 often has no hull."""
    import torch
    from cape_amd import synth_gpu

    gen = torch.Generator(device="cuda").manual_seed(seed)
    dev = synth_gpu.stream("room", seed, n, start=17, device="cuda", chunk=16).clone()
    q = n // 3
    dev[:q][torch.rand(dev[:q].shape, device="cuda", generator=gen) < 0.05] = 0
    dev[q:2 * q] += torch.randn(dev[q:2 * q].shape, device="cuda", generator=gen) * 3.0 * (dev[q:2 * q] > 0)
    for _ in range(6):
        y, x = int(torch.randint(0, 400, (1,), generator=gen, device="cuda")), int(torch.randint(0, 560, (1,), generator=gen, device="cuda"))
        dev[2 * q:, y:y + 80, x:x + 80] = 0
    return dev


def _check_batch_against_host(ex, n, host_poly, what):
    res = ex.results(n)
    pol, ver = ex.polygons(n)
    planes = 0
    for f in range(n):
        for i, s in enumerate(res.segments(f)):
            if not s["is_output"]:
                continue
            p = pol[f, i]
            pts = res.boundary_points(f, s)
            o, c = int(p["vertex_offset"]), int(p["vertex_count"])
            _same(p, ver[f, o:o + c], host_poly(pts, s["normal"], _center(s)), f"{what} frame {f} segment {i} ({len(pts)} points)")
            planes += 1
    return planes


def test_polygon_task_queue_holds_every_task(host_poly):
    """the queue of spawned (plane, rung) tasks is sized for what a batch can spawn (six rungs per plane + a quit mark
    per wave), and the memset, the tickets and the spawns are bounded by the SAME length -- also when a small batch follows a big
    one on a handle made for more frames (a ticket must never read a slot an earlier call left behind)."""
    import torch
    from cape_amd import Extractor, synth

    st = torch.cuda.current_stream().cuda_stream
    ex = Extractor(640, 480, cylinders=False, max_batch=96, **synth.DEFAULT_INTRINSICS)
    dev = _ragged_batch(96)
    spawned = 0
    for n in (96, 6, 48, 1):
        ex.extract_device(dev.data_ptr(), n, st)
        ex.build_polygons(n, st)
        reserved, tickets, slots = ex.polygon_queue()
        pool = ex.spill_info()[1]  # (the records of the spill pool count as frames: their planes spawn rungs like any other)
        assert slots == (n + pool) * 6 * 64 + 8192, "one slot per rung a plane can spawn + one quit mark per wave of the grid"
        assert reserved <= slots and tickets <= reserved, (reserved, tickets, slots)
        planes = int(ex.results(n).records["header"]["n_planes"].sum())
        assert reserved - min(reserved, 8192) <= 6 * max(planes, 1)
        spawned += reserved
        if n in (6, 1):
            assert _check_batch_against_host(ex, n, host_poly, f"batch of {n}") > 0
    assert spawned > 4 * 1024, "the ragged frames must make planes climb the ladder (spawned rungs)"
    ex.close()


def test_polygon_task_queue_overflow_worker(host_poly):
    """Runs only inside test_polygon_task_queue_overflow's subprocess, on the twin library whose queue has 16 slots."""
    if not os.environ.get("CAPE_EXPECT_QUEUE_OVERFLOW"):
        pytest.skip("driven by test_polygon_task_queue_overflow")
    import torch
    from cape_amd import Extractor, synth

    st = torch.cuda.current_stream().cuda_stream
    n = 48
    ex = Extractor(640, 480, cylinders=False, max_batch=n, **synth.DEFAULT_INTRINSICS)
    dev = _ragged_batch(n)
    for rep in range(3):
        ex.extract_device(dev.data_ptr(), n, st)
        ex.build_polygons(n, st)
        reserved, tickets, slots = ex.polygon_queue()
        assert slots == 16 and reserved > slots, f"the twin's queue must overflow: reserved {reserved}, slots {slots}"
    assert _check_batch_against_host(ex, n, host_poly, "overflowing queue") > n
    ex.close()


def test_polygon_task_queue_overflow():
    """through round 4 a spawned rung that did not fit the queue left its reserved slot unwritten behind the
    moved tail, and the ticket that landed on that slot would have spun for ever.  The twin library (16 slots for any batch)
    overflows on every call: every reserved slot below the capacity is written, the rungs beyond it are walked by the wave that
    could not enqueue them, tickets beyond it leave -- the polygons stay bit-identical to the host class and nothing hangs."""
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    twin = os.path.join(root, "rgb-d-slam_amd", "lib", "libcape_hip_cyl_exact.so")
    if not os.path.exists(twin):
        subprocess.check_call(["make", "-C", os.path.join(root, "rgb-d-slam_amd", "csrc"), "variants"])
    env = dict(os.environ, CAPE_HIP_LIB=twin, CAPE_EXPECT_QUEUE_OVERFLOW="1")
    out = subprocess.run([sys.executable, "-m", "pytest", os.path.abspath(__file__), "-q", "-x", "-p", "no:cacheprovider", "-k",
                          "queue_overflow_worker or polygons_of_extracted_planes or pass_without_planes_terminates"],
                         env=env, capture_output=True, text=True, cwd=root, timeout=600)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    assert " passed" in out.stdout and "failed" not in out.stdout and "skipped" not in out.stdout
