"""Behaviour of the C ABI around the kernels: the handle's device is used whatever device the caller has current
(and the caller gets its device back), one stream in flight per handle, seed-sequence export, error codes."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _intr():
    from cape_amd import synth

    return dict(synth.DEFAULT_INTRINSICS)


def test_current_device_is_preserved(oracle_mod):
    """Every entry point selects the handle's device itself and restores the caller's."""
    import torch
    from cape_amd import Extractor, synth

    torch.cuda.set_device(0)
    depth = synth.room(seed=2, frame=4)
    ex = Extractor(640, 480, cylinders=False, device=0, max_batch=2, **_intr())
    n = ex.extract_host(depth)
    res = ex.results(n)
    assert torch.cuda.current_device() == 0
    ref = oracle_mod.Oracle(640, 480, cylinders=False, **_intr()).run(depth)
    assert np.array_equal(res.plane_labels[0], ref.plane_labels)
    ex.close()


def test_handle_on_another_device_than_the_current_one(oracle_mod):
    """a handle created for device 1 while device 0 is current must allocate, copy, launch and synchronise on
    device 1 in every entry point (extract_host staging, rectify scratch, cell stats, timings)."""
    import torch
    from cape_amd import Extractor, synth

    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs")
    torch.cuda.set_device(0)
    depth = synth.tumlike(seed=1, frame=0)
    intr = dict(synth.TUM_FR1_INTRINSICS)
    ex = Extractor(640, 480, cylinders=True, device=1, max_batch=2, **intr)
    ex.enable_timing(True)
    n = ex.extract_host(depth)
    res = ex.results(n)
    assert torch.cuda.current_device() == 0
    orc = oracle_mod.Oracle(640, 480, cylinders=True, **intr)
    ref = orc.run(depth)
    assert np.array_equal(res.plane_labels[0], ref.plane_labels)
    assert np.array_equal(ex.cell_stats(0)["bin"], ref.bins)
    assert ex.timings()["calls"] == 1
    with torch.cuda.device(1):
        src = torch.from_numpy(depth).cuda()
        dst = torch.empty_like(src)
    ex.rectify_device(src.data_ptr(), dst.data_ptr(), 1, np.eye(4))
    with torch.cuda.device(1):
        torch.cuda.synchronize()
    assert np.array_equal(dst.cpu().numpy(), orc.rectify(depth, np.eye(4)))
    assert torch.cuda.current_device() == 0
    ex.close()


def test_stream_switch_is_ordered(oracle_mod):
    """One stream in flight per handle: a call on a second stream first waits for the handle's work on the first one,
    so the shared scratch (depth staging, per-cell buffers, records) is never overwritten under a running kernel."""
    import torch
    from cape_amd import Extractor, synth

    frames = np.stack([synth.room(seed=9, frame=i) for i in range(8)])
    other = np.stack([synth.tumlike(seed=3, frame=i) for i in range(8)])
    ex = Extractor(640, 480, cylinders=False, max_batch=8, **_intr())
    orc = oracle_mod.Oracle(640, 480, cylinders=False, **_intr())
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    for _ in range(3):
        ex.extract_host(other, s1.cuda_stream)      # long-ish work on stream 1 ...
        ex.extract_host(frames, s2.cuda_stream)     # ... then the same handle on stream 2 right away
    res = ex.results(8)
    for f in (0, 3, 7):
        assert np.array_equal(res.plane_labels[f], orc.run(frames[f]).plane_labels)
    ex.close()


def test_seed_sequence_entry(oracle_mod):
    from cape_amd import Extractor, synth

    depth = synth.tumlike(seed=2, frame=3)
    intr = dict(synth.TUM_FR1_INTRINSICS)
    ex = Extractor(640, 480, cylinders=True, max_batch=1, **intr)
    ex.extract_host(depth)
    ref = oracle_mod.Oracle(640, 480, cylinders=True, **intr).run(depth)
    seeds = ex.seed_sequence(0)
    assert len(seeds) > 0 and np.array_equal(seeds, ref.seeds)
    # capacity smaller than the sequence: truncated copy, full length reported
    out = np.zeros(2, np.int32)
    n = C.c_int32(0)
    assert ex.L.cape_copy_seed_sequence(ex.h, 0, out.ctypes.data_as(C.c_void_p), 2, C.byref(n)) == 0
    assert n.value == len(ref.seeds) and np.array_equal(out, ref.seeds[:2])
    assert ex.L.cape_copy_seed_sequence(ex.h, 5, out.ctypes.data_as(C.c_void_p), 2, C.byref(n)) == -1  # frame >= max_batch
    ex.close()


def test_previous_stream_may_be_destroyed(oracle_mod):
    """the handle never touches the stream of an earlier call again -- a caller may synchronise a temporary
    stream, destroy it, and go on with another one (the order is kept by a handle-owned event)."""
    import ctypes

    from cape_amd import Extractor, synth

    hip = ctypes.CDLL("libamdhip64.so")
    frames = np.stack([synth.room(seed=4, frame=i) for i in range(4)])
    ex = Extractor(640, 480, cylinders=False, max_batch=4, **_intr())
    orc = oracle_mod.Oracle(640, 480, cylinders=False, **_intr())
    for rounds in range(3):
        st = ctypes.c_void_p()
        assert hip.hipStreamCreate(ctypes.byref(st)) == 0
        ex.extract_host(frames, st.value)
        assert hip.hipStreamSynchronize(st) == 0
        assert hip.hipStreamDestroy(st) == 0          # legal: its work is done
        ex.extract_host(frames[::-1].copy(), 0)       # next call on the null stream: must not fail or hang
    res = ex.results(4)
    for f in range(4):
        assert np.array_equal(res.plane_labels[f], orc.run(frames[3 - f]).plane_labels)
    buf = ex.host_alloc((2, 480, 640))                # a pinned buffer left to close()
    buf[:] = frames[:2]
    ex.extract_host(buf)
    assert np.array_equal(ex.results(2).plane_labels[1], orc.run(frames[1]).plane_labels)
    ex.close()


def test_count_primitives_entry(oracle_mod):
    from cape_amd import Extractor, synth

    frames = np.stack([synth.room(seed=6, frame=i) for i in range(12)])
    ex = Extractor(640, 480, cylinders=False, max_batch=12, **_intr())
    ex.extract_host(frames)
    planes, cyls, most = ex.count_primitives(12)
    hdr = ex.results(12, with_boundary=False).records["header"]
    assert planes == int(hdr["n_planes"].sum()) and cyls == int(hdr["n_cylinders"].sum()) and most == int(hdr["n_planes"].max())
    assert planes > 0
    ex.close()


def test_effective_flags_tell_whether_the_async_second_pass_is_active():
    """CAPE_FLAG_ASYNC_SECOND_PASS is only usable with cylinders on, max_batch > 8 and no sub-batches; the layout
    says what is in force."""
    import cape_amd
    from cape_amd import Extractor

    on = Extractor(640, 480, cylinders=True, max_batch=32, async_second_pass=True, **_intr())
    assert on.effective_flags & cape_amd.CAPE_FLAG_ASYNC_SECOND_PASS and on.effective_flags & cape_amd.CAPE_FLAG_CYLINDERS
    on.close()
    for kw in (dict(cylinders=False, max_batch=32), dict(cylinders=True, max_batch=4), dict(cylinders=True, max_batch=32, sub_batches=2)):
        ex = Extractor(640, 480, async_second_pass=True, **kw, **_intr())
        assert not (ex.effective_flags & cape_amd.CAPE_FLAG_ASYNC_SECOND_PASS), kw
        ex.close()


def test_count_primitives_waits_for_the_async_second_pass():
    """cape_count_primitives read the records while the side stream's second pass was still writing them."""
    import torch
    from cape_amd import Extractor, synth, synth_gpu

    n = 512
    dev = synth_gpu.stream("room", 12, n, device="cuda", chunk=64)
    st = torch.cuda.current_stream().cuda_stream
    ref = Extractor(640, 480, cylinders=True, max_batch=n, **_intr())
    ref.extract_device(dev.data_ptr(), n, st)
    want = ref.count_primitives(n)
    ref.close()
    ex = Extractor(640, 480, cylinders=True, max_batch=n, async_second_pass=True, **_intr())
    for _ in range(5):
        ex.extract_device(dev.data_ptr(), n, st)
        assert ex.count_primitives(n) == want
    ex.close()


def test_polygon_reads_refuse_a_stale_batch():
    """after a new cape_extract the polygons / polygon matches on the device belong to the previous batch."""
    import torch
    from cape_amd import CapeError, Extractor, synth_gpu

    n = 4
    dev = synth_gpu.stream("room", 3, n, device="cuda", chunk=4)
    st = torch.cuda.current_stream().cuda_stream
    ex = Extractor(640, 480, max_batch=n, **_intr())
    ex.extract_device(dev.data_ptr(), n, st)
    ex.build_polygons(n, st)
    ex.match_polygons(n, 0, st)
    ex.polygons(n), ex.polygon_matches(n)
    ex.extract_device(dev.data_ptr(), n, st)
    with pytest.raises(CapeError):
        ex.polygons(n)
    with pytest.raises(CapeError):
        ex.polygon_matches(n)
    ex.build_polygons(2, st)
    ex.polygons(2)
    with pytest.raises(CapeError):
        ex.polygons(3)
    ex.close()


@pytest.mark.parametrize("cyl,batch", [(False, 96), (True, 96), (True, 1)])
def test_timings_carry_the_references_five_buckets(cyl, batch):
    """cape_get_timings maps the kernels onto the reference's stage buckets (find_primitives,
    primitive_detection.cpp:126-160): reset (nothing to reset: 0), init = A1 + A2, and the stage-B kernels' event time split into
    grow / merge / refine by the shader-clock ticks every frame's wave books in each.  Same bits with timing on (the parity suite
    runs with it off)."""
    import torch
    from cape_amd import Extractor, synth, synth_gpu

    dev = synth_gpu.stream("room", 12, batch, start=5, device="cuda", chunk=16)
    st = torch.cuda.current_stream().cuda_stream
    ex = Extractor(640, 480, cylinders=cyl, max_batch=batch, **_intr())
    ex.extract_device(dev.data_ptr(), batch, st)
    ref = ex.results(batch)
    assert ex.timings()["calls"] == 0
    ex.enable_timing(True)
    for _ in range(3):
        ex.extract_device(dev.data_ptr(), batch, st)
    t = ex.timings()
    assert t["calls"] == 3 and t["frames"] == 3 * batch
    assert t["reset_s"] == 0.0 and t["init_s"] == t["cell_fit_s"] > 0
    assert t["grow_phase_s"] > 0 and t["merge_s"] > 0 and t["refine_s"] > 0
    assert abs(t["grow_phase_s"] + t["merge_s"] + t["refine_s"] - t["grow_s"]) <= 1e-9 * t["grow_s"]
    assert t["merge_s"] < t["grow_phase_s"], "merge_planes is the short one of the three"
    got = ex.results(batch)
    assert got.records.tobytes() == ref.records.tobytes() and np.array_equal(got.plane_labels, ref.plane_labels)
    ex.reset_timings()
    z = ex.timings()
    assert z["calls"] == 0 and z["grow_phase_s"] == 0 and z["refine_s"] == 0
    ex.enable_timing(False)
    ex.close()


def test_log_callback_gets_the_batch_once(oracle_mod):
    """cape_set_log_callback: the lines arrive when a batch's records first reach the host, once per extracted batch.  Frames that
    overflow the boundary capacity (a handle made with a tiny one) produce the library's capacity warning; the reference's own lines
    (invalid seed / not planar after merge) are compared with the oracle's counts by every parity test (compare_frame) and, as host
    logic, by tests/test_abi.py::test_log_lines_from_frame_records."""
    import ctypes as C

    import cape_amd
    from cape_amd import synth

    frames = np.stack([synth.room(seed=2, frame=i) for i in range(3)])
    lines = []
    for max_batch in (1, 16):  # results in pinned host memory / in HBM
        L = cape_amd.load_library()
        cfg = cape_amd.cape_config(640, 480, 550.0, 550.0, 320.0, 240.0, 0, 0, max_batch, 8, 0)  # boundary_capacity = 8 points
        h = C.c_void_p()
        assert L.cape_create(C.byref(cfg), C.byref(h)) == 0
        cb = cape_amd.LOG_FN(lambda level, msg, frame, _u: lines.append((max_batch, int(level), msg.decode(), int(frame))))
        assert L.cape_set_log_callback(h, cb, None) == 0
        n = min(max_batch, 3)
        d = np.ascontiguousarray(frames[:n])
        assert L.cape_extract_host(h, d.ctypes.data_as(C.c_void_p), n, None) == 0
        rec = np.zeros(n, cape_amd.FRAME_RECORD_DTYPE)
        for _ in range(2):  # the second copy of the same batch stays silent
            assert L.cape_copy_results(h, n, rec.ctypes.data_as(C.c_void_p), None, None, None) == 0
        assert (rec["header"]["status"] & 2).all(), "boundary capacity of 8 points must overflow on a room frame"
        got = [ln for ln in lines if ln[0] == max_batch]
        assert got == [(max_batch, 1, "find_primitives: per-frame capacity exceeded, primitive list truncated", f) for f in range(n)]
        assert L.cape_extract_host(h, d.ctypes.data_as(C.c_void_p), n, None) == 0  # the next batch speaks again
        assert L.cape_copy_results(h, n, rec.ctypes.data_as(C.c_void_p), None, None, None) == 0
        assert len([ln for ln in lines if ln[0] == max_batch]) == 2 * n
        assert L.cape_set_log_callback(h, cape_amd.LOG_FN(0), None) == 0
        assert L.cape_extract_host(h, d.ctypes.data_as(C.c_void_p), n, None) == 0
        assert L.cape_copy_results(h, n, rec.ctypes.data_as(C.c_void_p), None, None, None) == 0
        assert len([ln for ln in lines if ln[0] == max_batch]) == 2 * n
        L.cape_destroy(h)


@pytest.mark.parametrize("max_batch", [1, 6])
def test_random_seed_of_a_non_deterministic_reference_build(oracle_mod, max_batch):
    """SURVEY section 5, determinism: the reference seeds its RANSAC engine with 0 under MAKE_DETERMINISTIC and with time(0) of the
    process otherwise (random.hpp:59-64); every frame restarts there.  cape_set_rng_seed regenerates the handle's draw table: the
    device must follow the oracle bit for bit for ANY seed, on both kinds of handle (results in pinned host memory / in HBM), and go
    back to the deterministic mode with seed 0."""
    from cape_amd import Extractor, synth
    from test_gpu_parity import compare_frame

    rng = np.random.default_rng(2)
    base = synth.tunnel(seed=3, frame=40)
    frames = [base, base * np.float32(1.03), synth.room(seed=4, frame=9)]
    frames.append(frames[0] + (rng.standard_normal(base.shape) * 5).astype(np.float32) * (base > 0))
    frames = np.stack(frames)[: max(1, min(4, max_batch))]
    intr = _intr()
    orc = oracle_mod.Oracle(640, 480, cylinders=True, **intr)
    ex = Extractor(640, 480, cylinders=True, max_batch=max_batch, **intr)
    seen = []
    for seed in (0, 1742515200, 7, 0):
        orc.set_rng_seed(seed)
        ex.set_rng_seed(seed)
        n = ex.extract_host(frames)
        res = ex.results(n)
        for f in range(n):
            compare_frame(orc.run(frames[f]), ex, res, f, check_cells=False)
        seen.append(res.cyl_labels.copy())
    assert np.array_equal(seen[0], seen[3]), "seed 0 again: the deterministic mode"
    assert res.records["header"]["n_cylinders"].sum() >= 1
    ex.close()
