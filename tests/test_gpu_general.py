"""The general grow instance (csrc/cape_grow_general.hip): any cell grid, any number of plane segments.

The reference's detector takes any image size (primitive_detection.cpp:26-67) and keeps its segments in an unbounded vector
(primitive_detection.hpp:206).  The fast kernels hold a grid row in one 64-bit mask and a frame's segments in 64 slots; whatever is
beyond either goes through the general instance -- and must equal the oracle bit for bit like everything else.
"""
import numpy as np
import pytest

from test_gpu_parity import compare_frame

pytestmark = pytest.mark.gpu


def _intr(width):
    from cape_amd import synth

    return {k: v * width / 640.0 for k, v in synth.DEFAULT_INTRINSICS.items()}


def _mix(W, H, n, intr):
    from cape_amd import synth

    gens = (synth.room, synth.tunnel, synth.tumlike)
    return np.stack([gens[k % 3](seed=3 + k % 5, frame=7 * k, width=W, height=H, intr=intr) for k in range(n)])


@pytest.mark.parametrize("cyl", [False, True])
@pytest.mark.parametrize("size", [(640, 480), (1280, 960), (100, 60), (1280, 20)])
def test_general_instance_equals_the_oracle_on_the_fast_kernels_grids(oracle_mod, monkeypatch, size, cyl):
    """CAPE_GROW=general (read at cape_create) sends EVERY frame of a handle through the general instance: on the grids the fast kernels
    serve -- 32 x 24, 64 x 48, a 5 x 3 toy grid, a single cell row -- it must reproduce the oracle like they do."""
    from cape_amd import Extractor

    W, H = size
    intr = _intr(W)
    frames = _mix(W, H, 6, intr)
    orc = oracle_mod.Oracle(W, H, cylinders=cyl, **intr)
    monkeypatch.setenv("CAPE_GROW", "general")
    ex = Extractor(W, H, cylinders=cyl, max_batch=len(frames), **intr)
    monkeypatch.delenv("CAPE_GROW")
    for rep in range(2):
        n = ex.extract_host(frames)
        res = ex.results(n)
        assert ex.spill_info()[2] == n, "every frame went through the general instance"
        for f in range(n):
            compare_frame(orc.run(frames[f]), ex, res, f, check_cells=(rep == 0))
    ex.close()


@pytest.mark.parametrize("cyl", [False, True])
@pytest.mark.parametrize("size", [(1920, 1080, "wide"), (1920, 1080, "general"), (2560, 1280, "wide"), (1080, 1920, "general"), (2560, 1440, "general"),
                                  (1300, 1300, "general")])
def test_grids_beyond_64_cells_equal_the_oracle(oracle_mod, monkeypatch, size, cyl):
    """1920 x 1080 (96 x 54 cells) and 2560 x 1280 (128 x 64): the fast kernels' two-word rows (Mask128) -- and 1920 x 1080 once more
    through the general instance (CAPE_GROW=general); the portrait twin (54 x 96), 2560 x 1440 (128 x 72) and 65 x 65 cells: more rows
    than a wave has lanes, the general instance.  A room / tunnel / desk mix, every observable of compare_frame incl. the per-cell fits
    of stage A on these widths."""
    from cape_amd import Extractor

    W, H, instance = size
    intr = _intr(W)
    frames = _mix(W, H, 6, intr)
    orc = oracle_mod.Oracle(W, H, cylinders=cyl, **intr)
    if instance == "general":
        monkeypatch.setenv("CAPE_GROW", "general")  # (read at cape_create; a no-op on the grids only that instance serves)
    ex = Extractor(W, H, cylinders=cyl, max_batch=len(frames), **intr)
    for rep in range(2):
        n = ex.extract_host(frames)
        res = ex.results(n)
        assert ex.spill_info()[2] == (n if instance == "general" else 0), "which grow instance served the frames"
        for f in range(n):
            compare_frame(orc.run(frames[f]), ex, res, f, check_cells=(rep == 0))
    # one frame at a time on a handle of its own (results in pinned host memory, the general kernel signals the host)
    one = Extractor(W, H, cylinders=cyl, max_batch=1, **intr)
    for f in (1, 4):
        n = one.extract_host(frames[f])
        compare_frame(orc.run(frames[f]), one, one.results(n), 0, check_cells=False)
    one.close()
    ex.close()


def test_one_frame_handle_follows_a_spilled_frame(oracle_mod):
    """The reference's call pattern -- one frame per call, results in pinned host memory, the host spinning on the chain's completion
    word: a 116-segment frame makes the 64-segment instance hand over to the general kernel, whose last wave signals instead."""
    from cape_amd import Extractor, synth
    from test_gpu_parity import _checkerboard_of_facets

    W, H = 1280, 960
    big, intr = _checkerboard_of_facets(W, H)
    room = synth.room(seed=2, frame=3, width=W, height=H, intr=intr)
    orc = oracle_mod.Oracle(W, H, cylinders=True, **intr)
    want = {id(big): orc.run(big), id(room): orc.run(room)}
    ex = Extractor(W, H, cylinders=True, max_batch=1, **intr)
    for fr in (room, big, room, big, big, room):
        n = ex.extract_host(fr)
        res = ex.results(n)
        compare_frame(want[id(fr)], ex, res, 0, check_cells=False)
        assert (int(res.records["header"]["next_record"][0]) >= 1) == (fr is big)
    ex.close()


def test_more_than_64_cylinder_labels(oracle_mod):
    """cylinder2regionMap is as unbounded as _planeSegments: a field of thin pipes gives more cylinder labels than a record's 64 --
    if the oracle finds that many on this scene the chain must carry them (else the test only pins equality)."""
    from cape_amd import Extractor

    W, H = 1280, 960
    intr = _intr(W)
    u = (np.arange(W) - intr["cx"]) / intr["fx"]
    v = (np.arange(H) - intr["cy"]) / intr["fy"]
    X, Y = np.meshgrid(u, v)
    rng = np.random.default_rng(11)
    z = np.full((H, W), 6000.0)
    pitch, rad = 160, 70.0
    for k, x0 in enumerate(range(pitch // 2, W, pitch)):
        for j, y0 in enumerate(range(0, H, 240)):
            # a vertical pipe segment: depth bulges towards the camera across its width, phase-shifted from row block to row block
            cx = x0 + (20 if j % 2 else -20)
            dx = (np.arange(W) - cx)
            bulge = np.sqrt(np.clip(rad * rad - dx * dx, 0, None))
            sl = slice(y0, min(y0 + 240, H))
            zz = 2500.0 + 150.0 * ((k + 2 * j) % 5) - 6.0 * bulge
            m = np.abs(dx) < rad
            z[sl, m] = zz[m]
    z += rng.normal(0, 0.5, z.shape)
    frame = np.round(z).astype(np.float32)
    orc = oracle_mod.Oracle(W, H, cylinders=True, **intr)
    want = orc.run(frame)
    ex = Extractor(W, H, cylinders=True, max_batch=2, **intr)
    n = ex.extract_host(np.stack([frame, frame]))
    res = ex.results(n)
    for f in range(n):
        compare_frame(want, ex, res, f, check_cells=False)
    ex.close()


def test_next_rows_on_a_1920x1080_grid(oracle_mod):
    """The rows around the path on a grid of 96-cell rows: N3 rectify_depth and N4 raw uint16 input bit-exact against the oracle,
    N1 / N2 (device polygons + polygon matches) against the oracle of the reference's polygon algorithm, the cell-mask pre-filter
    against match_oracle -- every piece that takes a width must take this one."""
    import torch
    import cape_amd
    import match_oracle
    import polygon_oracle_py as P
    from cape_amd import Extractor, synth
    from test_gpu_polygon_oracle import _center, compare_plane, new_stats

    P.build()
    W, H = 1920, 1080
    intr = _intr(W)
    frames = np.stack([synth.room(seed=5, frame=10 + k, width=W, height=H, intr=intr) for k in range(4)])
    orc = oracle_mod.Oracle(W, H, cylinders=True, **intr)
    ex = Extractor(W, H, cylinders=True, max_batch=len(frames), **intr)
    st = torch.cuda.current_stream().cuda_stream
    # N3
    a = np.deg2rad(1.0)
    T = np.array([[np.cos(a), 0, np.sin(a), -25.0], [0, 1, 0, 1.5], [-np.sin(a), 0, np.cos(a), 4.0], [0, 0, 0, 1]])
    din = torch.from_numpy(frames).cuda()
    dout = torch.empty_like(din)
    ex.rectify_device(din.data_ptr(), dout.data_ptr(), len(frames), T, st)
    got = dout.cpu().numpy()
    for f in range(len(frames)):
        assert np.array_equal(got[f].view(np.uint32), orc.rectify(frames[f], T).view(np.uint32)), "rectified depth differs"
    # N4: millimetre depths as raw sensor units with scale 1 (examples/main_CAPE.cpp:58-59)
    raw = frames.astype(np.uint16)
    assert np.array_equal(raw.astype(np.float32), frames)
    t16 = torch.from_numpy(raw.view(np.int16)).cuda()
    ex.extract_device_u16(t16.data_ptr(), 1.0, len(frames), st)
    res = ex.results(len(frames))
    want = [orc.run(f) for f in frames]
    for f in range(len(frames)):
        compare_frame(want[f], ex, res, f, check_cells=(f == 0))
    # N1 / N2
    ex.build_polygons(len(frames), st)
    ex.match_polygons(len(frames), 0, st)
    ex.match_consecutive(len(frames), 0, st)
    pol, ver = ex.polygons(len(frames))
    got_m = ex.polygon_matches(len(frames))
    stats = new_stats()
    kept = []
    for f in range(len(frames)):
        planes = []
        for i, s in enumerate(res.segments(f)):
            if not s["is_output"]:
                continue
            p = pol[f, i]
            o, c = int(p["vertex_offset"]), int(p["vertex_count"])
            ref = compare_plane(P, p, ver[f, o:o + c], res.boundary_points(f, s), s["normal"], _center(s), f"1920x1080 frame {f} segment {i}", stats)
            if ref is not None and ref.valid and ref.boundary_length() >= 3:
                planes.append((i, np.asarray(s["out_normal"], np.float64), float(s["d"]), ref))
        kept.append(planes)
    assert stats["planes"] >= 8 and stats["vertex_identical"] >= 0.9 * (stats["planes"] - stats["threw"] - len(stats["dissolve"]) - stats.get("degenerate", 0)), stats
    for f in range(1, len(frames)):
        if got_m[f]["flags"] & cape_amd.MATCH_EXACT_OVERFLOW or len(stats["dissolve"]):
            continue
        prev, cur = kept[f - 1], kept[f]
        wantm, _ = P.find_matches([q[1:] for q in prev], [q[1:] for q in cur], None, advanced=False, allow_index0=False)
        assert list(got_m[f]["match"][: len(prev)]) == wantm
    # the cell-mask pre-filter against match_oracle
    mm = ex.matches(len(frames))

    def per(r):
        roots = r.planes[:, 19].astype(int) if len(r.planes) else np.zeros(0, int)
        is_out = np.zeros(len(r.merge_labels), bool)
        is_out[roots] = True
        masks, _ = match_oracle.plane_masks(r.plane_labels, r.segments, r.merge_labels, is_out)
        return {"masks": masks, "normals": r.planes[:, 0:3], "d": r.planes[:, 3]}

    fr = [per(r) for r in want]
    for k in range(1, len(frames)):
        m, ap, ac, inter = match_oracle.match_frame(fr[k - 1], fr[k], advanced=False, allow_index0=False)
        g = mm[k]
        assert g["n_prev"] == len(ap) and g["n_cur"] == len(ac) and list(g["match"][: len(ap)]) == m
        assert np.array_equal(g["inter"][: len(ap), : len(ac)], inter)
    ex.close()
