"""GPU parity: the HIP path (through the C ABI) against the CPU oracle on the same seeded inputs.

Contract (SURVEY.md 8a 'parity observables'): integers / labels / flags bit-exact; every f64/f32 observable
bit-exact against the oracle as well (same IEEE operation sequence, contraction off) -- tolerance 0.
"""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

SCENES = [("room", 0, 0), ("room", 3, 17), ("tumlike", 1, 0), ("tumlike", 2, 5), ("tunnel", 0, 0), ("tunnel", 4, 9)]


def _intr(scene, scale=1.0):
    from cape_amd import synth

    base = synth.TUM_FR1_INTRINSICS if scene == "tumlike" else synth.DEFAULT_INTRINSICS
    return {k: v * scale for k, v in base.items()}


def _bits(a):
    a = np.ascontiguousarray(a)
    return a.view(np.uint64 if a.dtype == np.float64 else np.uint32)


def compare_frame(orc_res, ex, res, f, check_cells=True):
    """orc_res: OracleResult ; ex: Extractor ; res: FrameResults ; f: frame index in the batch."""
    if check_cells:
        cs = ex.cell_stats(f)
        assert np.array_equal(cs["planar"], orc_res.planar), "planar flags"
        assert np.array_equal(cs["point_count"], orc_res.n), "point counts"
        assert np.array_equal(_bits(cs["sums"]), _bits(orc_res.sums)), "cell sums (bitwise)"
        assert np.array_equal(_bits(cs["centroid"]), _bits(orc_res.centroid)), "centroid"
        assert np.array_equal(_bits(cs["normal"]), _bits(orc_res.normal)), "normal"
        assert np.array_equal(_bits(cs["d"]), _bits(orc_res.d)), "d"
        assert np.array_equal(_bits(cs["mse"]), _bits(orc_res.mse)), "mse"
        assert np.array_equal(_bits(cs["score"]), _bits(orc_res.score)), "score"
        assert np.array_equal(_bits(cs["tol"]), _bits(orc_res.tol)), "tolerances (f32 bitwise)"
        assert np.array_equal(cs["bin"], orc_res.bins), "histogram bins"
    hdr = res.records["header"][f]
    assert hdr["n_seeds"] == len(orc_res.seeds), "seed loop length"
    # the reference's hot-path log lines travel in the status word (cape_set_log_callback): as many as the oracle counted
    assert bool(hdr["status"] & (1 << 7)) == (orc_res.log_invalid_seed > 0), "log: invalid seed (primitive_detection.cpp:302)"
    assert ((int(hdr["status"]) >> 8) & 0xFF) == min(255, orc_res.log_not_planar_after_merge), "log: not planar after merge (:374, :497)"
    assert np.array_equal(ex.seed_sequence(f), orc_res.seeds), "seed sequence (cells in the order they were tried)"
    assert np.array_equal(res.plane_labels[f], orc_res.plane_labels), "plane label grid (bit-exact)"
    assert np.array_equal(res.cyl_labels[f], orc_res.cyl_labels), "cylinder label grid (bit-exact)"
    segs = res.segments(f)
    assert len(segs) == len(orc_res.segments)
    if len(segs):
        o = orc_res.segments
        assert np.array_equal(segs["merge_label"], orc_res.merge_labels), "planeMergeLabels"
        assert np.array_equal(_bits(segs["normal"]), _bits(o[:, 0:3])), "segment normal"
        assert np.array_equal(_bits(segs["d"]), _bits(o[:, 3])), "segment d"
        assert np.array_equal(_bits(segs["centroid"]), _bits(o[:, 4:7])), "segment centroid"
        assert np.array_equal(_bits(segs["mse"]), _bits(o[:, 7])), "segment mse"
        assert np.array_equal(_bits(segs["score"]), _bits(o[:, 8])), "segment score"
        assert np.array_equal(_bits(segs["sums"]), _bits(o[:, 9:18])), "segment sums"
        assert np.array_equal(segs["point_count"], o[:, 18].astype(np.uint32)), "segment n"
        assert np.array_equal(segs["planar"], o[:, 19].astype(np.uint32)), "segment planar"
    # cylinders (axis bitwise vs the oracle; radius NaN by the reference's quirk)
    assert hdr["n_cylinders"] == len(orc_res.cylinders), "cylinder container size"
    kept = res.cylinder_labels(f)
    assert len(kept) == hdr["n_cylinder_labels"]
    kept = kept[kept["kept"] == 1]
    assert len(kept) == len(orc_res.cylinders)
    if len(kept):
        assert np.array_equal(_bits(kept["axis"]), _bits(orc_res.cylinders[:, 0:3])), "cylinder axis"
        assert np.isnan(kept["radius"]).all()
    planes = res.planes(f)
    assert len(planes) == len(orc_res.planes) == hdr["n_planes"]
    bounds = res.plane_boundaries(f)  # (follows the frame's spill records, if it has more than 64 segments)
    for k, pl in enumerate(planes):
        o = orc_res.planes[k]
        assert np.array_equal(_bits(pl["out_normal"]), _bits(o[0:3])), "plane normal"
        assert _bits(pl["d"]) == _bits(o[3:4])[0], "plane d"
        assert np.array_equal(_bits(pl["cov"]), _bits(o[10:19])), "point cloud covariance"
        b_gpu = bounds[k]
        b_orc = orc_res.boundary[k]
        # the reference's point order is nondeterministic (parallel forEach): compare as sets
        assert sorted(map(tuple, _bits(b_gpu).tolist())) == sorted(map(tuple, _bits(b_orc).tolist())), "boundary points"


def _scan_boundary_frames():
    """Depth jumps placed on single steps of the continuity cross scan (plane_segment.cpp:62-100): the horizontal scan walks local row 10,
    idx 200..219, seeded with max(z[200], z[201]); the vertical one local column 10, idx 10, 30, .., 370 -- it stops BEFORE idx 390 --,
    seeded with max(z[10], z[30]).  A jump on the last horizontal step breaks the cell, one on idx 390 must not; jumps / holes on the
    seed pixels and on the step behind a hole exercise `last`."""
    from cape_amd import synth

    base = synth.room(seed=5, frame=4).astype(np.float32)
    rng = np.random.default_rng(23)
    frames = []
    spots = {
        "h_last": (10, 19), "v_ignored": (19, 10), "v_last": (18, 10), "h_seed0": (10, 0), "h_seed1": (10, 1),
        "v_seed0": (0, 10), "v_seed1": (1, 10), "centre": (10, 10), "h_mid": (10, 7), "v_mid": (7, 10),
    }
    for kind in ("jump", "hole", "hole_then_jump"):
        f = base.copy()
        for ci, (name, (r, c)) in enumerate(spots.items()):
            # a different block of cells for every spot, a few cells each
            for k in range(6):
                cy, cx = (2 * ci + k // 3) % 24, (3 * ci + 5 * k + ci // 4) % 32
                y, x = cy * 20 + r, cx * 20 + c
                if kind == "jump":
                    f[y, x] = f[y, x] + 400.0 + 50.0 * rng.random()
                elif kind == "hole":
                    f[y, x] = 0.0
                else:
                    f[y, x] = 0.0
                    yy, xx = (y + 1, x) if name.startswith("v") and r < 19 else (y, min(x + 1, cx * 20 + 19))
                    f[yy, xx] = f[yy, xx] + 300.0
        frames.append(f)
    return np.stack(frames)


@pytest.mark.parametrize("batch", [1, 3])
def test_continuity_scan_boundary_steps(oracle_mod, batch, monkeypatch):
    """Every step of both scans against the oracle, on the one-frame instances (strips / bands) and on the batch kernels."""
    from cape_amd import Extractor

    frames = _scan_boundary_frames()
    intr = _intr("room")
    orc = oracle_mod.Oracle(640, 480, cylinders=False, **intr)
    refs = [orc.run(f) for f in frames]
    # the spots really decide cells: the three frames do not give the same validity map
    assert not np.array_equal(refs[0].planar, refs[1].planar) or not np.array_equal(refs[0].n, refs[1].n)
    for mode in (("strips", "bands") if batch == 1 else ("bands",)):
        monkeypatch.setenv("CAPE_STAGE_A", mode)
        ex = Extractor(640, 480, cylinders=False, max_batch=(1 if batch == 1 else 16), **intr)
        if batch == 1:
            for i, f in enumerate(frames):
                n = ex.extract_host(f)
                compare_frame(refs[i], ex, ex.results(n), 0)
        else:
            n = ex.extract_host(frames)
            res = ex.results(n)
            for i in range(n):
                compare_frame(refs[i], ex, res, i)
        ex.close()


@pytest.fixture(params=["strips", "bands"])
def stage_a(request, monkeypatch):
    """One-frame handles (max_batch <= 8) run the ONE-FRAME CHAIN: stage A as one launch of strip workgroups
    (cape_cell_strip_kernel: the strip's workgroup sums, scans and fits its cells, the frame's last workgroup evaluates the
    cell edges) and ONE grow kernel, the 64-segment instance on every frame, whose last wave signals the host.
    CAPE_STAGE_A=bands keeps the classic chain on such a handle: the two streaming kernels of stage A, the 32-segment grow
    kernel, the redo pass for frames with more segments and the one-thread signal kernel.  The edge-case tests run on both
    (the knob is read at cape_create)."""
    monkeypatch.setenv("CAPE_STAGE_A", request.param)
    return request.param



@pytest.mark.parametrize("scene,seed,frame", SCENES)
def test_frame_parity_640(oracle_mod, scene, seed, frame, stage_a):
    from cape_amd import Extractor, synth

    depth = synth.SCENES[scene](seed=seed, frame=frame)
    intr = _intr(scene)
    orc = oracle_mod.Oracle(640, 480, cylinders=False, **intr)
    r = orc.run(depth)
    ex = Extractor(640, 480, cylinders=False, max_batch=4, **intr)
    n = ex.extract_host(depth)
    res = ex.results(n)
    compare_frame(r, ex, res, 0)
    ex.close()


def test_batch_matches_single_frames(oracle_mod, stage_a):
    """A batch is frames processed independently: every frame of a mixed batch equals its own oracle run."""
    from cape_amd import Extractor, synth

    frames = np.stack([synth.room(seed=5, frame=i * 7) for i in range(6)])
    frames[2] = 0.0                      # empty frame (all invalid)
    frames[4, :, 320:] = 0.0             # half-empty frame
    intr = _intr("room")
    orc = oracle_mod.Oracle(640, 480, cylinders=False, **intr)
    ex = Extractor(640, 480, cylinders=False, max_batch=8, **intr)
    n = ex.extract_host(frames)
    res = ex.results(n)
    for f in range(n):
        compare_frame(orc.run(frames[f]), ex, res, f)
    assert res.records["header"]["n_plane_segments"][2] == 0
    ex.close()


def test_edge_inputs(oracle_mod, stage_a):
    """NaN / negative / huge depths, ragged holes, constant depth (degenerate scatter -> rejected cells)."""
    from cape_amd import Extractor, synth

    rng = np.random.default_rng(7)
    base = synth.tumlike(seed=3, frame=1)
    a = base.copy()
    a[rng.random(a.shape) < 0.35] = 0.0          # ragged: 35 % holes
    b = base.copy()
    b[100:140, 200:260] = np.nan
    b[300:330, 50:90] = -5.0
    b[10:12, :] = 65535.0
    c = np.full_like(base, 1500.0)               # fronto-parallel, noise-free: det == 0 -> no planar cell
    d = base.copy()
    d[:, ::2] *= 1.6                             # violent discontinuities in every cell
    frames = np.stack([a, b, c, d])
    intr = _intr("tumlike")
    orc = oracle_mod.Oracle(640, 480, cylinders=False, **intr)
    ex = Extractor(640, 480, cylinders=False, max_batch=4, **intr)
    n = ex.extract_host(frames)
    res = ex.results(n)
    for f in range(n):
        compare_frame(orc.run(frames[f]), ex, res, f)
    ex.close()


def test_sparse_invalid_patterns_inside_valid_cells(oracle_mod, stage_a):
    """The streaming kernel accumulates pixels without the reference's `z > 0` test and leaves every cell that holds a
    bit pattern the test would reject (negative, -0, NaN) or that cannot be summed exactly (+inf) to the in-order pass
    of the per-cell kernel.  Sprinkle such values thinly over cells that stay valid (>= 200 good pixels, continuous):
    every observable has to match the oracle bit for bit, counts included."""
    from cape_amd import Extractor, synth

    rng = np.random.default_rng(19)
    base = synth.room(seed=8, frame=3)
    frames = []
    for value, density in ((-1200.0, 0.02), (-0.0, 0.05), (np.nan, 0.01), (np.inf, 0.004), (-np.inf, 0.01)):
        f = base.copy()
        mask = rng.random(f.shape) < density
        mask[:, 10::20] = False   # keep the centre row / column samples of the continuity scan clean
        mask[10::20, :] = False
        f[mask] = value
        frames.append(f)
    mixed = base.copy()
    for value in (-3.0, -0.0, np.nan):
        m = rng.random(mixed.shape) < 0.01
        m[:, 10::20] = False
        m[10::20, :] = False
        mixed[m] = value
    frames.append(mixed)
    frames = np.stack(frames).astype(np.float32)
    intr = _intr("room")
    orc = oracle_mod.Oracle(640, 480, cylinders=False, **intr)
    ex = Extractor(640, 480, cylinders=False, max_batch=len(frames), **intr)
    n = ex.extract_host(frames)
    res = ex.results(n)
    for f in range(n):
        r = orc.run(frames[f])
        if f != 3:  # (+inf is a valid depth for the reference: its sums are inf / NaN, compared as labels and counts)
            compare_frame(r, ex, res, f)
        cs = ex.cell_stats(f)
        assert np.array_equal(cs["point_count"], r.n) and np.array_equal(cs["planar"], r.planar)
        assert np.array_equal(res.plane_labels[f], r.plane_labels)
    # the damaged cells really took the in-order pass (CAPE_FRAME_INORDER_CELLS), and planes were still found around them
    assert (res.records["header"]["status"][[0, 1, 2, 4, 5]] & 0x10).all()
    assert (res.records["header"]["n_planes"][[0, 1, 2, 4, 5]] >= 1).all()
    ex.close()


def test_parity_1280x960(oracle_mod, stage_a):
    from cape_amd import Extractor, synth

    intr = _intr("room", 2.0)
    depth = synth.room(seed=2, frame=11, width=1280, height=960)
    orc = oracle_mod.Oracle(1280, 960, cylinders=False, **intr)
    ex = Extractor(1280, 960, cylinders=False, max_batch=2, **intr)
    n = ex.extract_host(depth)
    res = ex.results(n)
    compare_frame(orc.run(depth), ex, res, 0)
    ex.close()


def test_inorder_guard_path(oracle_mod, stage_a):
    """Intrinsics with a near-zero column factor force the exactness guard onto the in-order path; still bit-exact."""
    from cape_amd import Extractor, synth

    intr = dict(fx=550.0, fy=550.0, cx=320.0000001, cy=240.0000001)
    depth = synth.room(seed=9, frame=3)
    orc = oracle_mod.Oracle(640, 480, cylinders=False, **intr)
    ex = Extractor(640, 480, cylinders=False, max_batch=1, **intr)
    n = ex.extract_host(depth)
    res = ex.results(n)
    cs = ex.cell_stats(0)
    assert cs["inorder"].sum() > 0
    compare_frame(orc.run(depth), ex, res, 0)
    ex.close()


GOLDEN_PLANE_ONLY = ["tumlike_s1_planeonly", "room_s0_f0_planeonly", "room_s3_f17_planeonly", "room_1280_s2_f11_planeonly",
                     "tunnel_s0_f0_cyl", "tunnel_s4_f9_cyl", "tumlike_s2_f5_cyl"]


@pytest.mark.parametrize("name", GOLDEN_PLANE_ONLY)
def test_gpu_matches_committed_golden(name):
    """HIP path vs the committed fixtures (no oracle library involved at run time)."""
    import os
    import sys

    here = os.path.dirname(os.path.abspath(__file__))
    sys.path.insert(0, os.path.join(here, "golden"))
    import make_golden
    from cape_amd import Extractor, synth

    case = [c for c in make_golden.CASES if c[0] == name][0]
    _, scene, seed, frame, w, h, cyl = case
    g = np.load(os.path.join(here, "golden", name + ".npz"))
    depth = synth.SCENES[scene](seed=seed, frame=frame, width=w, height=h)
    assert synth.sha256(depth) == str(g["sha256"])
    ex = Extractor(w, h, cylinders=cyl, max_batch=1, **make_golden.intrinsics(scene, w))
    n = ex.extract_host(depth)
    res = ex.results(n)
    cs = ex.cell_stats(0)
    assert np.array_equal(cs["planar"], g["planar"]) and np.array_equal(cs["point_count"], g["point_count"])
    assert np.array_equal(cs["bin"], g["bins"])
    assert np.array_equal(cs["tol"].view(np.uint32), g["tol"].view(np.uint32))
    assert np.array_equal(res.plane_labels[0], g["plane_labels"])
    assert np.array_equal(res.cyl_labels[0], g["cyl_labels"])
    segs = res.segments(0)
    assert np.array_equal(segs["merge_label"], g["merge_labels"])
    assert np.array_equal(_bits(segs["normal"]), _bits(g["segments"][:, 0:3]))
    assert np.array_equal(_bits(segs["d"]), _bits(g["segments"][:, 3]))
    planes = res.planes(0)
    assert len(planes) == len(g["planes"])
    # the contract vs the real reference: normals / d within 1e-5 ; vs the oracle it is bitwise
    assert np.abs(planes["out_normal"] - g["planes"][:, 0:3]).max() <= 1e-5
    assert np.array_equal(_bits(planes["out_normal"]), _bits(g["planes"][:, 0:3]))
    ex.close()


def test_raw_u16_fixture_on_gpu():
    import os

    from cape_amd import Extractor, synth

    here = os.path.dirname(os.path.abspath(__file__))
    raw = np.load(os.path.join(here, "golden", "tumlike_s1_raw_u16.npz"))["raw"]
    depth = raw.astype(np.float32) * np.float32(0.2)
    g = np.load(os.path.join(here, "golden", "tumlike_s1_planeonly.npz"))
    ex = Extractor(640, 480, cylinders=False, max_batch=1, **synth.TUM_FR1_INTRINSICS)
    ex.extract_host(depth)
    res = ex.results(1)
    assert np.array_equal(res.plane_labels[0], g["plane_labels"])
    assert np.array_equal(_bits(res.segments(0)["normal"]), _bits(g["segments"][:, 0:3]))
    ex.close()


def test_streamed_batch_properties():
    """Full-size property checks (BASELINE.json configs[1] batch): determinism across launches and independence of
    frames from their batch neighbours -- size-independent, no oracle needed."""
    import torch
    from cape_amd import Extractor, synth

    U, B = 8, 512
    unique = synth.stream("room", seed=42, n_frames=U)
    depth = torch.from_numpy(unique).cuda().repeat(B // U, 1, 1).contiguous()
    ex = Extractor(640, 480, cylinders=False, max_batch=B)
    s = torch.cuda.current_stream().cuda_stream
    ex.extract_device(depth.data_ptr(), B, s)
    r1 = ex.results(B, with_boundary=False)
    ex.extract_device(depth.data_ptr(), B, s)
    r2 = ex.results(B, with_boundary=False)
    assert np.array_equal(r1.plane_labels, r2.plane_labels)
    assert r1.records.tobytes() == r2.records.tobytes()
    lab = r1.plane_labels.reshape(B // U, U, -1)
    assert (lab == lab[0]).all(), "identical frames at different batch positions must give identical labels"
    hdr = r1.records["header"].reshape(B // U, U)
    assert (hdr["n_planes"] == hdr["n_planes"][0]).all() and (hdr["n_planes"] >= 2).all()
    assert (r1.records["header"]["status"] & 0x7).max() == 0  # no overflow bits
    ex.close()


CYL_SCENES = [("tunnel", 0, 0), ("tunnel", 4, 9), ("tunnel", 7, 123), ("tumlike", 2, 5), ("room", 1, 4)]


@pytest.mark.parametrize("scene,seed,frame", CYL_SCENES)
def test_frame_parity_with_cylinders(oracle_mod, scene, seed, frame, stage_a):
    """BASELINE.json configs[2]: planes + cylinder RANSAC, RNG stream restarted per frame (mt19937(0))."""
    from cape_amd import Extractor, synth

    depth = synth.SCENES[scene](seed=seed, frame=frame)
    intr = _intr(scene)
    orc = oracle_mod.Oracle(640, 480, cylinders=True, **intr)
    r = orc.run(depth)
    ex = Extractor(640, 480, cylinders=True, max_batch=2, **intr)
    n = ex.extract_host(np.stack([depth, depth]))
    res = ex.results(n)
    compare_frame(r, ex, res, 0)
    compare_frame(r, ex, res, 1)  # same frame twice in one batch: per-frame RNG restart
    if scene == "tunnel":
        assert res.records["header"]["n_cylinders"][0] >= 1
    ex.close()


def test_cylinder_ordered_sum_fallback():
    """The cylinder instance decides `dist < minHypothesisDist` (cylinder_segment.cpp:296) and `plane MSE < cylinder MSE`
    (primitive_detection.cpp:444) on brackets around tree-order sums and runs the reference's ordered sums only when the
    brackets overlap -- which the real bound (2^-40) never produces on these scenes.  The twin library built with
    -DCAPE_CYL_EPS=0.25 takes that fallback on most comparisons; it has to reproduce the oracle bit for bit as well."""
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    twin = os.path.join(root, "rgb-d-slam_amd", "lib", "libcape_hip_cyl_exact.so")
    if not os.path.exists(twin):
        subprocess.check_call(["make", "-C", os.path.join(root, "rgb-d-slam_amd", "csrc"), "variants"])
    env = dict(os.environ, CAPE_HIP_LIB=twin)
    out = subprocess.run([sys.executable, "-m", "pytest", os.path.abspath(__file__), "-q", "-x", "-p", "no:cacheprovider", "-k",
                          "frame_parity_with_cylinders or cylinders_noisy_and_1280 or cylinder_schedules_agree"],
                         env=env, capture_output=True, text=True, cwd=root)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    assert " passed" in out.stdout and "failed" not in out.stdout


def test_cylinders_noisy_and_1280(oracle_mod, stage_a):
    """Harder RANSAC inputs: bumpy tunnel (several sub-segments / plane-vs-cylinder model selection) and 1280x960."""
    from cape_amd import Extractor, synth

    rng = np.random.default_rng(11)
    d = synth.tunnel(seed=3, frame=40)
    d2 = d.copy()
    d2[:, 200:440] *= np.float32(1.04)      # a second, slightly larger "pipe" section
    d3 = d.copy()
    d3 += (rng.standard_normal(d.shape) * 6).astype(np.float32) * (d > 0)
    frames = np.stack([d, d2, d3])
    intr = _intr("tunnel")
    orc = oracle_mod.Oracle(640, 480, cylinders=True, **intr)
    ex = Extractor(640, 480, cylinders=True, max_batch=4, **intr)
    n = ex.extract_host(frames)
    res = ex.results(n)
    for f in range(n):
        compare_frame(orc.run(frames[f]), ex, res, f)
    ex.close()
    big = synth.tunnel(seed=1, frame=2, width=1280, height=960)
    intr2 = _intr("tunnel", 2.0)
    orc2 = oracle_mod.Oracle(1280, 960, cylinders=True, **intr2)
    ex2 = Extractor(1280, 960, cylinders=True, max_batch=1, **intr2)
    ex2.extract_host(big)
    compare_frame(orc2.run(big), ex2, ex2.results(1), 0)
    ex2.close()


def test_packed_payload_visible_to_torch_without_copy():
    """The gather payload: torch wraps libcape_hip's packed staging slot through __cuda_array_interface__."""
    import torch
    from cape_amd import Extractor, synth
    from cape_amd.dist import Shard

    class DevMem:
        def __init__(self, ptr, n):
            self.__cuda_array_interface__ = {"shape": (n,), "typestr": "|u1", "data": (ptr, False), "version": 2}

    frames = synth.stream("tumlike", seed=1, n_frames=3)
    ex = Extractor(640, 480, cylinders=False, max_batch=3, **synth.TUM_FR1_INTRINSICS)
    lay = ex.gather_configure(3, 64, 64)
    ex.extract_host(frames)
    res = ex.results(3)
    ptr = ex.pack(3)
    torch.cuda.synchronize()
    t = torch.as_tensor(DevMem(ptr, lay["bytes_per_rank"]), device="cuda")
    sh = Shard(t.cpu().numpy(), lay)
    assert np.array_equal(sh.frames["n_planes"], res.records["header"]["n_planes"])
    for f in range(3):
        pl = res.planes(f)
        got = sh.frame_planes(f)
        assert len(got) == len(pl)  # nothing truncated, however many planes a frame holds
        assert np.array_equal(got["normal"], pl["out_normal"]) and np.array_equal(got["d"], pl["d"])
        assert np.array_equal(got["sums"], pl["sums"]) and np.array_equal(got["point_count"], pl["point_count"])
    # the payload depends on the current frames only: an empty batch after a busy one leaves nothing behind
    ex.extract_host(np.zeros_like(frames))
    ex.pack(3)
    ex.pack(3)  # both staging slots
    raw = ex.packed_host()
    assert not raw[lay["frames_offset"]:].any(), "stale primitives in the gather payload"
    ex.close()


def test_raw_uint16_input_path(oracle_mod, stage_a):
    """N4: cape_extract_u16 (device-side convertTo(CV_32F, 1/5)) equals the float path and the oracle, bit for bit."""
    import os

    import torch
    from cape_amd import Extractor, synth

    here = os.path.dirname(os.path.abspath(__file__))
    raw0 = np.load(os.path.join(here, "golden", "tumlike_s1_raw_u16.npz"))["raw"]
    raws = [raw0]
    for f in (3, 8):
        d = synth.tumlike(seed=4, frame=f)
        r = np.rint(d.astype(np.float64) * 5.0).astype(np.uint16)
        assert np.array_equal(r.astype(np.float32) * np.float32(0.2), d)
        raws.append(r)
    raw = np.stack(raws)
    depth = raw.astype(np.float32) * np.float32(0.2)
    intr = synth.TUM_FR1_INTRINSICS
    ex = Extractor(640, 480, cylinders=True, max_batch=4, **intr)
    t = torch.from_numpy(raw.view(np.int16)).cuda()
    ex.extract_device_u16(t.data_ptr(), 0.2, len(raw), torch.cuda.current_stream().cuda_stream)
    res = ex.results(len(raw))
    orc = oracle_mod.Oracle(640, 480, cylinders=True, **intr)
    for f in range(len(raw)):
        compare_frame(orc.run(depth[f]), ex, res, f)
    # the same frames from host memory (cape_extract_u16_host): pageable, then pinned (three frames are read in place over PCIe)
    first = res.records.tobytes()
    ex.extract_host_u16(raw, 0.2, torch.cuda.current_stream().cuda_stream)
    assert ex.results(len(raw)).records.tobytes() == first
    pinned = ex.host_alloc(raw.shape, np.uint16)
    pinned[...] = raw
    ex.extract_host_u16(pinned, 0.2, torch.cuda.current_stream().cuda_stream)
    assert ex.results(len(raw)).records.tobytes() == first
    ex.host_free(pinned)
    ex.close()


FACET_CASES = [(11, 33, True), (22, 66, True), (0, 0, True), (13, 39, True), (4, 12, False), (9, 27, False),
               (31, 93, False), (34, 102, True)]


def test_faceted_scenes_batch(oracle_mod, stage_a):
    """Faceted surfaces + foreground slabs: 5-9 regions per frame, failed seeds, cylinder branch, and (seeds 11, 22)
    plane segments that merge_planes() fuses -- the one path the room / tunnel scenes never reach."""
    from cape_amd import Extractor, synth

    intr = _intr("room")
    merges = 0
    for cyl in (True, False):
        cases = [c for c in FACET_CASES if c[2] == cyl]
        frames = np.stack([synth.facets(seed=s, frame=f) for s, f, _ in cases])
        orc = oracle_mod.Oracle(640, 480, cylinders=cyl, **intr)
        ex = Extractor(640, 480, cylinders=cyl, max_batch=len(cases), **intr)
        n = ex.extract_host(frames)
        res = ex.results(n)
        for k in range(n):
            r = orc.run(frames[k])
            compare_frame(r, ex, res, k)
            merges += int((r.merge_labels != np.arange(len(r.merge_labels))).sum())
        ex.close()
    assert merges >= 2, "the merge path was not exercised"


def test_random_frames_property(oracle_mod):
    """Randomised sweep: 24 frames across all generators and random corruptions, every observable bit-exact."""
    from cape_amd import Extractor, synth

    rng = np.random.default_rng(2024)
    names = ["room", "tumlike", "tunnel", "facets"]
    intr = _intr("room")
    frames = []
    for k in range(24):
        d = synth.SCENES[names[k % 4]](seed=int(rng.integers(0, 1000)), frame=int(rng.integers(0, 500)))
        mode = k % 6
        if mode == 1:
            d[rng.random(d.shape) < 0.2] = 0
        elif mode == 2:
            d += (rng.standard_normal(d.shape) * 5).astype(np.float32) * (d > 0)
        elif mode == 3:
            y, x = int(rng.integers(0, 400)), int(rng.integers(0, 560))
            d[y:y + 80, x:x + 80] *= np.float32(0.5)
        elif mode == 4:
            d = np.ascontiguousarray(d[:, ::-1])
        frames.append(d)
    frames = np.stack(frames)
    for cyl in (False, True):
        orc = oracle_mod.Oracle(640, 480, cylinders=cyl, **intr)
        ex = Extractor(640, 480, cylinders=cyl, max_batch=len(frames), **intr)
        n = ex.extract_host(frames)
        res = ex.results(n)
        for k in range(n):
            compare_frame(orc.run(frames[k]), ex, res, k, check_cells=(not cyl))
        ex.close()


def test_pathological_values_terminate_and_match_labels(oracle_mod, stage_a):
    """+inf / NaN / denormal / 1e30 depths: both paths terminate; integer observables still agree (float payloads of
    NaN sums are not compared)."""
    from cape_amd import Extractor, synth

    d = synth.room(seed=6, frame=2)
    a = d.copy()
    a[40:60, 100:140] = np.inf
    a[200:220, 300:340] = np.nan
    a[300:320, 20:60] = 1e-42      # denormal
    a[400:420, 500:540] = 1e30
    a[100:120, 400:440] = -np.inf
    b = np.full_like(d, np.inf)
    c = np.full_like(d, np.nan)
    frames = np.stack([a, b, c])
    intr = _intr("room")
    for cyl in (False, True):
        orc = oracle_mod.Oracle(640, 480, cylinders=cyl, **intr)
        ex = Extractor(640, 480, cylinders=cyl, max_batch=3, **intr)
        ex.extract_host(frames)
        res = ex.results(3)
        for f in range(3):
            r = orc.run(frames[f])
            cs = ex.cell_stats(f)
            assert np.array_equal(cs["planar"], r.planar)
            assert np.array_equal(cs["point_count"], r.n)
            assert np.array_equal(res.plane_labels[f], r.plane_labels)
            assert np.array_equal(res.cyl_labels[f], r.cyl_labels)
            assert res.records["header"]["n_plane_segments"][f] == len(r.segments)
        ex.close()


def test_rectify_depth_1280x960(oracle_mod):
    """N3 at the wide geometry (bands of 8 target rows: the LDS keys of a band are 8 B per target)."""
    import torch
    from cape_amd import Extractor, synth

    W, H = 1280, 960
    intr = {k: v * 2 for k, v in synth.DEFAULT_INTRINSICS.items()}
    frames = np.stack([synth.tunnel(seed=4, frame=3, width=W, height=H, intr=intr), synth.room(seed=9, frame=2, width=W, height=H, intr=intr)])
    a = np.deg2rad(0.7)
    T = np.array([[np.cos(a), 0, np.sin(a), -50.0], [0, 1, 0, 2.5], [-np.sin(a), 0, np.cos(a), 3.0], [0, 0, 0, 1]])
    ex = Extractor(W, H, cylinders=False, max_batch=2, **intr)
    orc = oracle_mod.Oracle(W, H, cylinders=False, **intr)
    din = torch.from_numpy(frames).cuda()
    dout = torch.empty_like(din)
    ex.rectify_device(din.data_ptr(), dout.data_ptr(), 2, T, torch.cuda.current_stream().cuda_stream)
    got = dout.cpu().numpy()
    for f in range(2):
        assert np.array_equal(got[f].view(np.uint32), orc.rectify(frames[f], T).view(np.uint32))
    assert ex.rectify_flagged() == 0
    ex.close()


@pytest.mark.parametrize("band", ["8", "32", "64"])
def test_rectify_band_override_never_exceeds_lds(oracle_mod, monkeypatch, band):
    """CAPE_RECTIFY_BAND asks for a band height; 32 rows x 640 x 8 B is exactly the 160 KB a gfx950 workgroup may have and does not fit
    beside the kernel's static LDS (the launch used to abort the queue): the launcher has to take the next smaller band, same bits."""
    import torch
    from cape_amd import Extractor, synth

    monkeypatch.setenv("CAPE_RECTIFY_BAND", band)
    intr = _intr("room")
    frames = np.stack([synth.room(seed=2, frame=i) for i in range(2)]).astype(np.float32)
    a = np.deg2rad(0.4)
    T = np.eye(4)
    T[:3, :3] = [[np.cos(a), 0, np.sin(a)], [0, 1, 0], [-np.sin(a), 0, np.cos(a)]]
    T[:3, 3] = [-20.0, 2.0, 0.5]
    orc = oracle_mod.Oracle(640, 480, cylinders=False, **intr)
    ex = Extractor(640, 480, cylinders=False, max_batch=2, **intr)
    din = torch.from_numpy(frames).cuda()
    dout = torch.empty_like(din)
    ex.rectify_device(din.data_ptr(), dout.data_ptr(), 2, T, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    got = dout.cpu().numpy()
    for f in range(2):
        assert np.array_equal(got[f].view(np.uint32), orc.rectify(frames[f], T).view(np.uint32))
    assert ex.rectify_flagged() == 0
    ex.close()


def test_rectify_depth_parity(oracle_mod):
    """N3: device rectify_depth == oracle (deterministic last-writer-wins), then the rectified image through the path."""
    import torch
    from cape_amd import Extractor, synth

    intr = synth.DEFAULT_INTRINSICS
    frames = np.stack([synth.room(seed=3, frame=1), synth.tunnel(seed=2, frame=5), synth.facets(seed=7, frame=0)])
    # a realistic Kinect-style extrinsic: 25 mm baseline, ~1 degree rotation about y, small z offset
    a = np.deg2rad(1.0)
    T = np.array([[np.cos(a), 0, np.sin(a), -25.0], [0, 1, 0, 1.5], [-np.sin(a), 0, np.cos(a), 4.0], [0, 0, 0, 1]])
    ex = Extractor(640, 480, cylinders=False, max_batch=3, **intr)
    orc = oracle_mod.Oracle(640, 480, cylinders=False, **intr)
    din = torch.from_numpy(frames).cuda()
    dout = torch.empty_like(din)
    s = torch.cuda.current_stream().cuda_stream
    for TT in (np.eye(4), T):
        ex.rectify_device(din.data_ptr(), dout.data_ptr(), 3, TT, s)
        got = dout.cpu().numpy()
        for f in range(3):
            ref = orc.rectify(frames[f], TT)
            assert np.array_equal(got[f].view(np.uint32), ref.view(np.uint32)), "rectified depth differs"
        # second call reuses the (cleared) key buffer
        ex.rectify_device(din.data_ptr(), dout.data_ptr(), 3, TT, s)
        assert np.array_equal(dout.cpu().numpy().view(np.uint32), got.view(np.uint32))
    # rigs the band kernel's row-displacement prediction treats differently: a pitch (every row moves by ~19: the bands scan
    # an offset window), a small roll (the displacement varies by ~17 rows over the image: wide windows), a big roll (more than
    # the bands can afford: every frame goes to the general kernels), and a scene closer than the depths the prediction samples
    # with a vertical baseline (its pixels escape the windows: that frame alone is flagged and redone)
    def rot(axis, deg, t):
        a = np.deg2rad(deg)
        c, s_ = np.cos(a), np.sin(a)
        R = {"x": [[1, 0, 0], [0, c, -s_], [0, s_, c]], "y": [[c, 0, s_], [0, 1, 0], [-s_, 0, c]], "z": [[c, -s_, 0], [s_, c, 0], [0, 0, 1]]}[axis]
        M = np.eye(4)
        M[:3, :3] = R
        M[:3, 3] = t
        return M

    near = frames.copy()
    near[1] *= np.float32(0.08)  # 80 .. 400 mm
    cases = [("pitch 2 deg", rot("x", 2.0, [-25.0, 1.5, 4.0]), frames, 0), ("roll 1.5 deg", rot("z", 1.5, [-25.0, 0.0, 0.0]), frames, 0),
             ("roll 6 deg", rot("z", 6.0, [10.0, -3.0, 0.0]), frames, 3), ("near scene, 30 mm vertical baseline", rot("y", 0.3, [0.0, 30.0, 0.0]), near, 1)]
    for name, TT, src, flagged in cases:
        dsrc = torch.from_numpy(src).cuda()
        ex.rectify_device(dsrc.data_ptr(), dout.data_ptr(), 3, TT, s)
        got = dout.cpu().numpy()
        for f in range(3):
            ref = orc.rectify(src[f], TT)
            assert np.array_equal(got[f].view(np.uint32), ref.view(np.uint32)), f"rectified depth differs: {name}, frame {f}"
            assert (ref > 0).mean() > 0.2, name
        assert ex.rectify_flagged() == flagged, f"{name}: {ex.rectify_flagged()} frames went to the general kernels, expected {flagged}"
    ex.rectify_device(din.data_ptr(), dout.data_ptr(), 3, T, s)
    # rectified frames feed the extractor like in examples/main_CAPE.cpp:186
    ex.extract_device(dout.data_ptr(), 3, s)
    res = ex.results(3)
    rect = dout.cpu().numpy()
    for f in range(3):
        compare_frame(orc.run(rect[f]), ex, res, f)
    ex.close()


def test_large_mixed_batch_labels(oracle_mod):
    """256 different frames in one launch (several independent frame-waves per workgroup): every label grid, plane count
    equals the frame's own oracle run."""
    from cape_amd import Extractor, synth

    names = ["room", "facets", "tunnel", "facets"]
    frames = np.stack([synth.SCENES[names[i % 4]](seed=1000 + i, frame=3 * i) for i in range(64)])
    frames = np.concatenate([frames, frames[::-1], frames[:, :, ::-1], frames[::-1, :, ::-1]])  # 256 frames
    frames = np.ascontiguousarray(frames)
    intr = _intr("room")
    for cyl in (False, True):
        ex = Extractor(640, 480, cylinders=cyl, max_batch=len(frames), **intr)
        ex.extract_host(frames)
        res = ex.results(len(frames), with_boundary=False)
        orc = oracle_mod.Oracle(640, 480, cylinders=cyl, **intr)
        for f in range(len(frames)):
            r = orc.run(frames[f])
            assert np.array_equal(res.plane_labels[f], r.plane_labels), f
            assert np.array_equal(res.cyl_labels[f], r.cyl_labels), f
            assert res.records["header"]["n_planes"][f] == len(r.planes)
            assert res.records["header"]["n_cylinders"][f] == len(r.cylinders)
        ex.close()


@pytest.mark.parametrize("w,h", [(320, 240), (800, 600), (1280, 720), (160, 120), (1000, 40)])
def test_other_grid_shapes(oracle_mod, w, h, stage_a):
    """Grids that are not 32x24 / 64x48: partial band segments (40 = 32 + 8 cells), odd band counts, u64 rows with
    fewer than 64 columns, a single cell row."""
    from cape_amd import Extractor, synth

    s = w / 640.0
    intr = dict(fx=550.0 * s, fy=550.0 * s, cx=w / 2.0 + 0.25, cy=h / 2.0 - 0.5)
    frames = np.stack([synth.facets(seed=21, frame=1, width=w, height=h, intr=intr),
                       synth.tunnel(seed=5, frame=2, width=w, height=h, intr=intr)])
    for cyl in (False, True):
        orc = oracle_mod.Oracle(w, h, cylinders=cyl, **intr)
        ex = Extractor(w, h, cylinders=cyl, max_batch=2, **intr)
        ex.extract_host(frames)
        res = ex.results(2)
        for f in range(2):
            compare_frame(orc.run(frames[f]), ex, res, f)
        ex.close()


def test_cylinder_schedules_agree(oracle_mod):
    """With cylinders on, a handle first grows every frame with the plane-only kernel and hands the frames that reach
    the cylinder branch to the cylinder kernel (two-pass); once most frames of a call were handed over it switches to
    the cylinder kernel alone.  Both schedules must give the same bits, call after call, for all-cylinder, no-cylinder
    and mixed batches."""
    import torch
    from cape_amd import Extractor, synth

    intr = _intr("room")
    tunnel = np.stack([synth.tunnel(seed=3, frame=f) for f in range(8)])
    room = np.stack([synth.room(seed=3, frame=f) for f in range(8)])
    mixed = np.concatenate([tunnel[:3], room[:3], tunnel[3:5]])
    orc = oracle_mod.Oracle(640, 480, cylinders=True, **intr)
    ex = Extractor(640, 480, cylinders=True, max_batch=8, **intr)
    expected = {}
    for call, (name, frames) in enumerate([("t", tunnel), ("t", tunnel), ("t", tunnel), ("t", tunnel), ("r", room),
                                           ("m", mixed), ("t", tunnel), ("m", mixed)]):
        n = ex.extract_host(frames)
        torch.cuda.synchronize()   # lets the handle read the hand-over count of this call before the next one
        res = ex.results(n)
        if name not in expected:
            expected[name] = [orc.run(frames[k]) for k in range(n)]
        for k in range(n):
            compare_frame(expected[name][k], ex, res, k, check_cells=(call == 0))
    ex.close()


@pytest.mark.parametrize("mode", ["wave", "group", "off"])
@pytest.mark.parametrize("w,h,n", [(640, 480, 48), (1280, 960, 12), (320, 240, 24)])  # (320x240: 192 cells, fewer than one pass of the workgroup finisher's register cache)
def test_cylinder_resume_modes(oracle_mod, monkeypatch, mode, w, h, n):
    """A frame that reaches a cylinder candidate is parked by the plane-only pass and finished by one wavefront (the RESUME
    instance of the grow kernel), by one workgroup (cape_resume.hip) or -- CAPE_RESUME=off, the round-2 schedule -- grown
    again from scratch by the cylinder kernel.  Three implementations of cylinder_fitting's bookkeeping around the same
    arithmetic: every one must give the oracle's bits, on both mask widths, on scenes with small and wall-sized candidates."""
    import torch
    from cape_amd import Extractor, synth_gpu

    monkeypatch.setenv("CAPE_RESUME", mode)
    monkeypatch.setenv("CAPE_SCHEDULE", "two")
    scale = w / 640.0
    for scene in ("room", "tunnel", "tumlike"):
        intr = _intr(scene, scale)
        dev = synth_gpu.stream(scene, 31, n, width=w, height=h, start=200, device="cuda", chunk=8)
        frames = dev.cpu().numpy()
        orc = oracle_mod.Oracle(w, h, cylinders=True, **intr)
        ex = Extractor(w, h, cylinders=True, max_batch=n, **intr)
        for _ in range(2):
            ex.extract_device(dev.data_ptr(), n, torch.cuda.current_stream().cuda_stream)
            res = ex.results(n)
        cyl_frames = 0
        for f in range(n):
            r = orc.run(frames[f])
            compare_frame(r, ex, res, f, check_cells=False)
            cyl_frames += int(len(r.cylinders) > 0)
        ex.close()
        if scene == "tunnel":
            assert cyl_frames >= n // 2, "the tunnel stream must exercise cylinder_fitting"


def test_async_second_pass_two_handles(oracle_mod):
    """CAPE_FLAG_ASYNC_SECOND_PASS: the cylinder second pass runs on a stream of the handle's own and the caller's stream
    does not wait for it; two handles fed alternately overlap one batch's streaming kernels with the other's slow tail.
    Whatever touches a handle next (the next extract, a results copy, the packing, the polygon pass) must find the
    second pass done: results bit-exact for both handles, call after call."""
    import torch
    from cape_amd import Extractor, synth_gpu

    n = 96
    intr = _intr("room")
    st = torch.cuda.current_stream().cuda_stream
    streams = {sc: synth_gpu.stream(sc, 71, n, start=10, device="cuda", chunk=32) for sc in ("room", "tunnel")}
    orc = oracle_mod.Oracle(640, 480, cylinders=True, **intr)
    want = {sc: [orc.run(fr) for fr in streams[sc][::8].cpu().numpy()] for sc in streams}
    pair = [Extractor(640, 480, cylinders=True, max_batch=n, async_second_pass=True, **intr) for _ in range(2)]
    order = ["room", "tunnel", "tunnel", "room", "room", "tunnel"]
    last = [None, None]
    for i, sc in enumerate(order):
        pair[i & 1].extract_device(streams[sc].data_ptr(), n, st)   # returns while the previous handle's tail is still running
        last[i & 1] = sc
        if i == 3:
            pair[1].build_polygons(n, st)                            # an entry point right behind an asynchronous second pass
    for k, ex in enumerate(pair):
        res = ex.results(n)
        for j, f in enumerate(range(0, n, 8)):
            compare_frame(want[last[k]][j], ex, res, f, check_cells=False)
    # raw device pointers: cape_sync_results orders a consumer stream behind the second pass
    pair[0].extract_device(streams["room"].data_ptr(), n, st)
    pair[0].sync_results(st)
    rec_ptr, pl_ptr, _, _ = pair[0].device_pointers()
    lab = torch.empty(n * pair[0].cells, dtype=torch.int32, device="cuda")
    import ctypes
    hip = ctypes.CDLL("libamdhip64.so")
    assert hip.hipMemcpyAsync(ctypes.c_void_p(lab.data_ptr()), ctypes.c_void_p(pl_ptr), ctypes.c_size_t(lab.numel() * 4), 3, ctypes.c_void_p(st)) == 0
    torch.cuda.synchronize()
    assert np.array_equal(lab.cpu().numpy().reshape(n, -1)[0], want["room"][0].plane_labels)
    for ex in pair:
        ex.close()


@pytest.mark.parametrize("cyl", [False, True])
def test_sub_batch_pipeline_matches_single_chain(oracle_mod, cyl):
    """cape_config.sub_batches > 1 cuts a batch into sub-batches that alternate between two internal streams; results
    (records, label grids, boundary points, summaries order) must not depend on it."""
    from cape_amd import Extractor, synth

    intr = _intr("room")
    frames = np.stack([synth.tunnel(seed=5, frame=f) if f % 3 == 0 else synth.room(seed=5, frame=f) for f in range(22)])
    ref = Extractor(640, 480, cylinders=cyl, max_batch=len(frames), **intr)
    n = ref.extract_host(frames)
    want = ref.results(n)
    ref.close()
    for sb in (2, 3, 5):
        ex = Extractor(640, 480, cylinders=cyl, max_batch=len(frames), sub_batches=sb, **intr)
        for _ in range(2):  # second call: the pipeline's events and streams are reused
            assert ex.extract_host(frames) == n
            got = ex.results(n)
            assert np.array_equal(got.plane_labels, want.plane_labels), f"sub_batches={sb}"
            assert np.array_equal(got.cyl_labels, want.cyl_labels), f"sub_batches={sb}"
            assert np.array_equal(got.records["header"], want.records["header"]), f"sub_batches={sb}"
            for f in range(n):
                k = int(want.records["header"]["n_plane_segments"][f])
                assert got.records["segments"][f][:k].tobytes() == want.records["segments"][f][:k].tobytes(), (sb, f)
                nb = int(want.records["header"]["n_boundary_points"][f])
                assert np.array_equal(got.boundary[f][:nb], want.boundary[f][:nb])
        ex.close()
    orc = oracle_mod.Oracle(640, 480, cylinders=cyl, **intr)
    for f in (0, 1, 21):
        ex = Extractor(640, 480, cylinders=cyl, max_batch=len(frames), sub_batches=3, **intr)
        ex.extract_host(frames)
        compare_frame(orc.run(frames[f]), ex, ex.results(n), f, check_cells=False)
        ex.close()


def test_more_than_32_plane_segments(oracle_mod, stage_a):
    """The everyday kernels keep 32 plane segments in LDS; a frame that needs more is redone by the 64-segment
    instance.  This 1280x960 room frame (depth scaled x2.13: noisy far walls, dozens of cylinder-branch regions whose
    inliers fit planes better) yields 34 segments in the reference algorithm; it sits between ordinary frames."""
    from cape_amd import Extractor, synth

    W, H = 1280, 960
    intr = _intr("room", 2.0)
    big = synth.room(seed=85969, frame=1635, width=W, height=H, intr=intr) * np.float32(2.128972746525244)
    frames = np.stack([synth.room(seed=1, frame=0, width=W, height=H, intr=intr), big,
                       synth.tunnel(seed=1, frame=0, width=W, height=H, intr=intr), big])
    orc = oracle_mod.Oracle(W, H, cylinders=True, **intr)
    want = [orc.run(f) for f in frames]
    assert len(want[1].segments) == 34
    ex = Extractor(W, H, cylinders=True, max_batch=len(frames), **intr)
    for _ in range(2):
        n = ex.extract_host(frames)
        res = ex.results(n)
        assert int(res.records["header"]["status"][1]) & 1 == 0, "frame flagged as truncated"
        assert int(res.records["header"]["n_plane_segments"][1]) == 34
        for k in range(n):
            compare_frame(want[k], ex, res, k, check_cells=False)
    ex.close()


@pytest.mark.parametrize("scene,cyl", [("room", False), ("tumlike", True), ("tunnel", True)])
def test_device_rendered_streams_parity(oracle_mod, scene, cyl):
    """bench.py renders its streams on the GPU (cape_amd.synth_gpu, torch): those frames are fresh inputs, not fixtures --
    the extractor must agree with the oracle on them like on any other frame, float32 and raw uint16 alike."""
    import torch
    from cape_amd import Extractor, synth_gpu

    n = 6
    intr = _intr(scene)
    dev = synth_gpu.stream(scene, 77, n, start=1000, device="cuda", chunk=4)
    frames = dev.cpu().numpy()
    assert frames.dtype == np.float32 and (frames > 0).mean() > 0.5
    assert len({frames[i].tobytes() for i in range(n)}) == n, "every frame of a rendered stream is distinct"
    orc = oracle_mod.Oracle(640, 480, cylinders=cyl, **intr)
    ex = Extractor(640, 480, cylinders=cyl, max_batch=n, **intr)
    ex.extract_device(dev.data_ptr(), n, torch.cuda.current_stream().cuda_stream)
    res = ex.results(n)
    for f in range(n):
        compare_frame(orc.run(frames[f]), ex, res, f, check_cells=(f == 0))
    raw = synth_gpu.stream(scene, 77, 2, start=1000, device="cuda", raw_u16=True)
    ex.extract_device_u16(raw.data_ptr(), 0.2, 2, torch.cuda.current_stream().cuda_stream)
    res = ex.results(2)
    as_f32 = raw.cpu().numpy().view(np.uint16).astype(np.float32) * np.float32(0.2)
    for f in range(2):
        compare_frame(orc.run(as_f32[f]), ex, res, f, check_cells=False)
    ex.close()


def test_stage_a_instances_agree_on_pinned_and_device_input(oracle_mod, monkeypatch):
    """The latency instance of stage A (strips) and the throughput kernels (bands) leave the same bits for the grow kernel,
    whichever way the frame arrives: pageable (staged copy -> strips), pinned (read over the link -> bands unless forced),
    resident in HBM.  Checked on the cell statistics, the edge predicates as the seed sequence / label grids show them, and
    against the oracle; several calls in a row (the per-frame strip counter is handed back at zero)."""
    import torch
    from cape_amd import Extractor, synth

    intr = _intr("tumlike")
    frames = np.stack([synth.tumlike(seed=4, frame=i * 3) for i in range(3)])
    frames[1, 200:260, 100:400] = 0.0
    orc = oracle_mod.Oracle(640, 480, cylinders=True, **intr)
    want = [orc.run(f) for f in frames]
    for mode in (None, "strips", "bands"):
        if mode:
            monkeypatch.setenv("CAPE_STAGE_A", mode)
        else:
            monkeypatch.delenv("CAPE_STAGE_A", raising=False)
        ex = Extractor(640, 480, cylinders=True, max_batch=2, **intr)
        pinned = ex.host_alloc((2, 480, 640))
        dev = torch.from_numpy(frames).cuda()
        for rep in range(2):
            for k in range(3):
                # pageable
                n = ex.extract_host(frames[k])
                compare_frame(want[k], ex, ex.results(n), 0)
                # pinned: two frames in one call
                pinned[0] = frames[k]
                pinned[1] = frames[(k + 1) % 3]
                n = ex.extract_host(pinned)
                res = ex.results(n)
                compare_frame(want[k], ex, res, 0)
                compare_frame(want[(k + 1) % 3], ex, res, 1)
                # resident in HBM
                ex.extract_device(dev[k].data_ptr(), 1, torch.cuda.current_stream().cuda_stream)
                torch.cuda.synchronize()
                compare_frame(want[k], ex, ex.results(1), 0)
        ex.host_free(pinned)
        ex.close()


def test_one_frame_chain_many_calls_in_a_row(monkeypatch):
    """The one-frame chain hands work between workgroups inside a launch (a strip's planes -> the frame's last workgroup, across
    the XCDs' L2 caches: release fence + counter + acquire fence) and signals the host from the grow kernel's last wave.  2 000
    calls in a row on one handle, alternating frames of different scenes and one- and three-frame calls: every call's label
    grids, seed counts and segment records equal the classic chain's (CAPE_STAGE_A=bands) for the same frame, bit for bit."""
    from cape_amd import Extractor, synth

    intr = _intr("room")
    frames = np.stack([synth.room(seed=2, frame=5), synth.tumlike(seed=1, frame=3), synth.tunnel(seed=0, frame=2), synth.facets(seed=4, frame=0),
                       synth.room(seed=7, frame=40)])
    frames[3, 100:180, 300:420] = 0.0
    monkeypatch.setenv("CAPE_STAGE_A", "bands")
    ref = Extractor(640, 480, cylinders=True, max_batch=8, **intr)
    n = ref.extract_host(frames)
    want = ref.results(n)
    want_labels = want.plane_labels.copy()
    want_cyl = want.cyl_labels.copy()
    want_rec = want.records.copy()
    ref.close()
    monkeypatch.delenv("CAPE_STAGE_A")
    ex = Extractor(640, 480, cylinders=True, max_batch=3, **intr)

    def same(res, f, k):
        assert np.array_equal(res.plane_labels[f], want_labels[k]) and np.array_equal(res.cyl_labels[f], want_cyl[k])
        assert res.records["header"][f].tobytes() == want_rec["header"][k].tobytes()
        ns = int(want_rec["header"][k]["n_plane_segments"])
        assert res.records["segments"][f][:ns].tobytes() == want_rec["segments"][k][:ns].tobytes()

    for call in range(2000):
        k = (call * 3 + call // 7) % 5
        if call % 5 == 4:
            ks = [k, (k + 1) % 5, (k + 3) % 5]
            n = ex.extract_host(frames[ks])
            res = ex.results(n, with_boundary=False)
            for f, kk in enumerate(ks):
                same(res, f, kk)
        else:
            n = ex.extract_host(frames[k])
            same(ex.results(n, with_boundary=False), 0, k)
    ex.close()


def _checkerboard_of_facets(W=1280, H=960, tile=100, seed=3):
    """Tilted facets in a checkerboard -- four orientations, no two neighbours alike, a depth step between neighbours: every facet is a
    plane segment of its own (116 of them at 1280x960 with 100-px tiles)."""
    from cape_amd import synth

    s = W / 640.0
    intr = {k: v * s for k, v in synth.DEFAULT_INTRINSICS.items()}
    u = (np.arange(W) - intr["cx"]) / intr["fx"]
    v = (np.arange(H) - intr["cy"]) / intr["fy"]
    X, Y = np.meshgrid(u, v)
    rng = np.random.default_rng(seed)
    tilts = [(0.5, 0.0), (-0.5, 0.0), (0.0, 0.5), (0.0, -0.5)]
    z = np.zeros((H, W))
    for ty in range(0, H, tile):
        for tx in range(0, W, tile):
            nx, ny = tilts[((tx // tile) % 2) + 2 * ((ty // tile) % 2)]
            d = 2000.0 + 120.0 * (((tx // tile) * 7 + (ty // tile) * 13) % 9)
            sl = (slice(ty, min(ty + tile, H)), slice(tx, min(tx + tile, W)))
            z[sl] = d / (1.0 + nx * X[sl] + ny * Y[sl])
    z += rng.normal(0, 0.6, z.shape)
    return np.round(z).astype(np.float32), intr


@pytest.mark.parametrize("cyl", [False, True])
def test_more_than_64_plane_segments_spill_into_a_record_chain(oracle_mod, cyl):
    """`_planeSegments` is an unbounded vector in the reference (primitive_detection.hpp:206); a frame record holds CAPE_MAX_PLANES = 64.
    A checkerboard of 116 facets (the oracle: 116 plane segments) must come back WHOLE: the 64-segment instance hands the frame to the
    general instance (cape_grow_general.hip), which writes a chain of records -- the frame's own and one of the handle's spill pool
    (header.next_record) -- and every observable of compare_frame equals the oracle's, for the big frames and for the frames around
    them in the batch, call after call.  No capacity warning reaches the log callback."""
    from cape_amd import Extractor, synth

    W, H = 1280, 960
    big, intr = _checkerboard_of_facets(W, H)
    orc = oracle_mod.Oracle(W, H, cylinders=cyl, **intr)
    want_big = orc.run(big)
    assert len(want_big.segments) == 116
    room = synth.room(seed=1, frame=0, width=W, height=H, intr=intr)
    tunnel = synth.tunnel(seed=1, frame=0, width=W, height=H, intr=intr)
    frames = np.stack([room, big, tunnel, big])
    want = {0: orc.run(room), 1: want_big, 2: orc.run(tunnel), 3: want_big}
    ex = Extractor(W, H, cylinders=cyl, max_batch=len(frames), **intr)
    lines = []
    ex.set_log_callback(lambda level, msg, frame: lines.append((level, msg, frame)))
    for rep in range(2):
        n = ex.extract_host(frames)
        res = ex.results(n)
        hdr = res.records["header"]
        assert ex.spill_info() == (2, 8, 2), "two frames through the general instance, one spill record each"
        for f in range(4):
            assert int(hdr["status"][f]) & 0x7 == 0, "no capacity flag"
            assert (int(hdr["next_record"][f]) >= len(frames)) == (f in (1, 3))
            compare_frame(want[f], ex, res, f, check_cells=False)
        for f in (1, 3):
            assert int(hdr["n_plane_segments"][f]) == 116 and int(res.plane_labels[f].max()) == 116
            chain = res.chain(f)
            assert len(chain) == 2 and int(chain[1][0]["header"]["segment_base"]) == 64
            assert int(chain[1][0]["header"]["n_plane_segments"]) == 116 - 64 and int(chain[1][0]["header"]["next_record"]) == -1
    assert not [ln for ln in lines if "per-frame capacity exceeded" in ln[1]]
    ex.close()


def test_spill_pool_exhausted_is_flagged_not_silent(oracle_mod):
    """The pool of spill records is a memory budget (cape_config.spill_records), not an algorithmic limit: with ONE spill record and two
    116-segment frames in the batch, one of them gets the record and equals the oracle, the other is flagged CAPE_FRAME_PLANE_OVERFLOW,
    truncated at its own record's 64 segments, and warned about through the log callback; the frames around them are untouched."""
    from cape_amd import Extractor, synth

    W, H = 1280, 960
    big, intr = _checkerboard_of_facets(W, H)
    orc = oracle_mod.Oracle(W, H, cylinders=False, **intr)
    want_big = orc.run(big)
    room = synth.room(seed=1, frame=0, width=W, height=H, intr=intr)
    frames = np.stack([big, room, big])
    ex = Extractor(W, H, cylinders=False, max_batch=len(frames), spill_records=1, **intr)
    lines = []
    ex.set_log_callback(lambda level, msg, frame: lines.append((level, msg, frame)))
    n = ex.extract_host(frames)
    res = ex.results(n)
    hdr = res.records["header"]
    flagged = [f for f in (0, 2) if int(hdr["status"][f]) & 1]
    whole = [f for f in (0, 2) if not int(hdr["status"][f]) & 1]
    assert len(flagged) == 1 and len(whole) == 1
    compare_frame(want_big, ex, res, whole[0], check_cells=False)
    f = flagged[0]
    assert int(hdr["n_plane_segments"][f]) == 64 and int(hdr["next_record"][f]) == -1
    assert np.array_equal(res.plane_labels[f], want_big.plane_labels), "the label grid is the whole frame's even when the records ran out"
    compare_frame(orc.run(room), ex, res, 1, check_cells=False)
    warned = [ln for ln in lines if "per-frame capacity exceeded" in ln[1]]
    assert [ln[2] for ln in warned] == [f]
    ex.close()


def test_record_window_of_more_than_64_regions(oracle_mod):
    """A frame whose seed loop records 66 regions of which 59 become plane segments: it overflows the 32-segment instance and is redone
    by the 64-segment one, whose record window used to hold MAXP + 1 = 65 slots while the regions' plane fits run one LANE each -- the
    65th record was converted with whatever its slot held (found by the round-6 fuzz sweep: an extra segment in about half of 16 copies
    of such a frame).  Sixteen copies in one batch, all equal to the oracle."""
    from cape_amd import Extractor, synth

    W, H, tile = 1280, 960, 120
    intr = {k: v * 2.0 for k, v in synth.DEFAULT_INTRINSICS.items()}
    u = (np.arange(W) - intr["cx"]) / intr["fx"]
    v = (np.arange(H) - intr["cy"]) / intr["fy"]
    X, Y = np.meshgrid(u, v)
    rng = np.random.default_rng(5)
    tilts = [(0.5, 0.0), (-0.5, 0.0), (0.0, 0.5), (0.0, -0.5)]
    z = np.zeros((H, W))
    for ty in range(0, H, tile):
        for tx in range(0, W, tile):
            nx, ny = tilts[((tx // tile) % 2) + 2 * ((ty // tile) % 2)]
            d = 2000.0 + 120.0 * (((tx // tile) * 7 + (ty // tile) * 13) % 9)
            sl = (slice(ty, min(ty + tile, H)), slice(tx, min(tx + tile, W)))
            sigma = 15.0 if ((tx // tile) + (ty // tile)) % 3 == 0 else 0.6  # a third of the facets: too noisy for a plane
            z[sl] = d / (1.0 + nx * X[sl] + ny * Y[sl]) + rng.normal(0, sigma, z[sl].shape)
    frame = np.round(z).astype(np.float32)
    orc = oracle_mod.Oracle(W, H, cylinders=False, **intr)
    want = orc.run(frame)
    recorded = int(np.isin(want.seed_outcome, (1, 2, 3)).sum())
    assert recorded >= 65 and 32 < len(want.segments) <= 64, (recorded, len(want.segments))
    ex = Extractor(W, H, cylinders=False, max_batch=16, **intr)
    for rep in range(2):
        n = ex.extract_host(np.stack([frame] * 16))
        res = ex.results(n)
        for f in range(n):
            compare_frame(want, ex, res, f, check_cells=False)
    ex.close()
