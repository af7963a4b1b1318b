"""GPU parity at BASELINE.json's full batch sizes (configs[1], [2], [4]'s one-GPU leg).

The oracle runs at ~500 frames/s per core, so a 4 096-frame batch is checked the way the contract allows: the CPU oracle
on every 16th frame (all observables of `compare_frame`, bit for bit), and size-independent properties on ALL frames --
every frame terminated without a capacity/seed-limit status, counts are self-consistent with the label grids, and the
batch result of a frame equals what the same frame gives alone in a batch of one (scheduling independence: which
wave / kernel instance / redo list a frame lands in must not change a bit of it).
"""
import numpy as np
import pytest

from test_gpu_parity import _intr, compare_frame

pytestmark = pytest.mark.gpu

BAD_STATUS = 1 | 2 | 4 | 32 | 64  # plane / boundary / cylinder overflow, RNG exhausted, seed limit


def _properties_all_frames(res, n, cells):
    hdr = res.records["header"][:n]
    assert not (hdr["status"] & BAD_STATUS).any(), "no frame may hit a capacity or iteration guard on these streams"
    assert (hdr["n_plane_segments"] >= hdr["n_planes"]).all()
    # label grids: every plane label in [0, n_plane_segments], every cylinder label in [0, n_cylinder_labels]
    assert (res.plane_labels[:n].min(axis=1) >= 0).all()
    assert (res.plane_labels[:n].max(axis=1) <= hdr["n_plane_segments"]).all()
    assert (res.cyl_labels[:n].max(axis=1) <= hdr["n_cylinder_labels"]).all()
    # a cell is never both a plane cell and a cylinder cell
    assert not ((res.plane_labels[:n] > 0) & (res.cyl_labels[:n] > 0)).any()


def _check_stream(oracle_mod, scene, cyl, n, W=640, H=480, stride=16, match=False, chunk=64, keep=()):
    import torch
    from cape_amd import Extractor, synth_gpu

    scale = W / 640.0
    intr = _intr(scene, scale)
    dev = synth_gpu.stream(scene, 100, n, width=W, height=H, start=0, device="cuda", chunk=chunk)
    ex = Extractor(W, H, cylinders=cyl, max_batch=n, **intr)
    stream = torch.cuda.current_stream().cuda_stream
    ex.extract_device(dev.data_ptr(), n, stream)
    if match:
        ex.match_consecutive(n, 0, stream)
        # the reference's own measure on the same batch: boundary polygons + their intersection areas (rows N1 / N2 on the device)
        ex.build_polygons(n, stream)
        ex.match_polygons(n, 0, stream)
    res = ex.results(n, with_boundary=True)
    _properties_all_frames(res, n, ex.cells)
    orc = oracle_mod.Oracle(W, H, cylinders=cyl, **intr)
    picked = list(range(0, n, stride))
    frames = dev[picked].cpu().numpy()
    n_planes = n_cyl = 0
    for k, f in enumerate(picked):
        r = orc.run(frames[k])
        compare_frame(r, ex, res, f, check_cells=(k % 8 == 0))
        n_planes += len(r.planes)
        n_cyl += len(r.cylinders)
    # scheduling independence on a few frames the oracle did NOT see: alone in a batch of one == inside the big batch
    ex1 = Extractor(W, H, cylinders=cyl, max_batch=1, **intr)
    for f in (1, n // 3 + 1, n - 3):
        ex1.extract_device(dev[f:f + 1].data_ptr(), 1, stream)
        one = ex1.results(1)
        assert np.array_equal(one.plane_labels[0], res.plane_labels[f])
        assert np.array_equal(one.cyl_labels[0], res.cyl_labels[f])
        assert one.records["header"][0].tobytes() == res.records["header"][f].tobytes()
        ns = int(one.records["header"]["n_plane_segments"][0])
        assert one.records["segments"][0][:ns].tobytes() == res.records["segments"][f][:ns].tobytes()
    ex1.close()
    matches = ex.matches(n) if match else None
    if match:
        import cape_amd

        pol, _ = ex.polygons(n)
        exact = ex.polygon_matches(n)
        out = res.records["segments"]["is_output"][:n] == 1
        # no BASELINE config may need the host class: no plane beyond the device hull's capacity, no frame beyond the matcher's
        assert not (pol["flags"][out] & cape_amd.POLY_OVERFLOW).any(), "CAPE_POLY_OVERFLOW on a BASELINE stream"
        assert not (exact["flags"] & cape_amd.MATCH_EXACT_OVERFLOW).any(), "CAPE_MATCH_EXACT_OVERFLOW on a BASELINE stream"
        # (the tunnel shows one plane per frame, index 0, which the reference's `selectedIndex <= 0` never returns: areas, not matches)
        assert (exact["inter_area"][1:, 0, 0] > 0).mean() > 0.9
    ex.close()
    kept = dev[list(keep)].cpu().numpy() if keep else None  # the very frames of the batch (a re-rendered stream may chunk its noise differently)
    return n_planes, n_cyl, res, (matches, kept) if match else None


def test_configs1_room_4096_plane_only(oracle_mod):
    """BASELINE.json configs[1] at the size bench.py times: 4 096 distinct device-rendered room frames, planes only."""
    n_planes, _, res, _ = _check_stream(oracle_mod, "room", False, 4096)
    assert n_planes >= 256 * 2, "a room shows walls"
    assert (res.records["header"]["n_planes"] > 0).mean() > 0.95


def test_configs2_tunnel_2048_cylinders(oracle_mod):
    """BASELINE.json configs[2]: 2 048 tunnel frames, planes + cylinder RANSAC."""
    _, n_cyl, res, _ = _check_stream(oracle_mod, "tunnel", True, 2048)
    assert n_cyl >= 64, "the tunnel is a cylinder"
    assert (res.records["header"]["n_cylinders"] > 0).mean() > 0.5


def test_configs4_1280x960_tunnel_1024_cylinders_match(oracle_mod):
    """BASELINE.json configs[4], one-GPU leg: 1 024 frames of 1280x960, planes + cylinders + consecutive-frame matching."""
    import match_oracle
    n = 1024
    _, n_cyl, res, (matches, dev) = _check_stream(oracle_mod, "tunnel", True, n, W=1280, H=960, stride=32, match=True, chunk=16,
                                                  keep=(510, 511, 512, 513))
    assert n_cyl >= 16
    # the matcher on two consecutive pairs, against match_oracle fed with the oracle's own frames
    intr = _intr("tunnel", 2.0)
    orc = oracle_mod.Oracle(1280, 960, cylinders=True, **intr)

    def per(depth):
        r = orc.run(depth)
        roots = r.planes[:, 19].astype(int) if len(r.planes) else np.zeros(0, int)
        is_out = np.zeros(len(r.merge_labels), bool)
        is_out[roots] = True
        masks, _ = match_oracle.plane_masks(r.plane_labels, r.segments, r.merge_labels, is_out)
        return {"masks": masks, "normals": r.planes[:, 0:3], "d": r.planes[:, 3]}

    fr = [per(d) for d in dev]
    for k in (1, 2, 3):
        m, ap, ac, inter = match_oracle.match_frame(fr[k - 1], fr[k], advanced=False, allow_index0=False)
        g = matches[510 + k]
        assert g["n_prev"] == len(ap) and g["n_cur"] == len(ac)
        assert list(g["match"][: len(ap)]) == m
        assert np.array_equal(g["inter"][: len(ap), : len(ac)], inter)


def _every_frame(oracle_mod, scene, cyl, n, chunk=256, W=640, H=480, dev=None, after_extract=None):
    """ALL n frames of a batch against the oracle (the full-size batches were checked by sampling): label grids,
    counts, the seed-loop length, the log-line bits of the status word, every plane segment record and every output plane bit for
    bit, cylinder axes.  The oracle runs on a pool of threads (one Oracle object each; ctypes releases the GIL in the C call)."""
    import concurrent.futures as cf
    import os
    import threading

    import torch
    from cape_amd import Extractor, synth_gpu

    intr = _intr(scene, W / 640.0)
    if dev is None:
        dev = synth_gpu.stream(scene, 100, n, start=0, device="cuda", chunk=64)
    ex = Extractor(W, H, cylinders=cyl, max_batch=n, **intr)
    ex.extract_device(dev.data_ptr(), n, torch.cuda.current_stream().cuda_stream)
    extra = after_extract(ex, n) if after_extract else None  # (polygons / matches of the same batch, checked by the caller)
    res = ex.results(n, with_boundary=after_extract is not None)
    local = threading.local()

    def bits(a):
        return np.ascontiguousarray(a).view(np.uint64)

    def check(args):
        f, depth = args
        if not hasattr(local, "orc"):
            local.orc = oracle_mod.Oracle(W, H, cylinders=cyl, **intr)
        r = local.orc.run(depth)
        hdr = res.records["header"][f]
        ok = (np.array_equal(res.plane_labels[f], r.plane_labels) and np.array_equal(res.cyl_labels[f], r.cyl_labels)
              and hdr["n_seeds"] == len(r.seeds) and hdr["n_plane_segments"] == len(r.segments) and hdr["n_planes"] == len(r.planes)
              and hdr["n_cylinders"] == len(r.cylinders)
              and bool(hdr["status"] & (1 << 7)) == (r.log_invalid_seed > 0) and ((int(hdr["status"]) >> 8) & 0xFF) == min(255, r.log_not_planar_after_merge))
        segs = res.segments(f)
        if ok and len(segs):
            o = r.segments
            ok = (np.array_equal(bits(segs["normal"]), bits(o[:, 0:3])) and np.array_equal(bits(segs["d"]), bits(o[:, 3]))
                  and np.array_equal(bits(segs["centroid"]), bits(o[:, 4:7])) and np.array_equal(bits(segs["mse"]), bits(o[:, 7]))
                  and np.array_equal(bits(segs["score"]), bits(o[:, 8])) and np.array_equal(bits(segs["sums"]), bits(o[:, 9:18]))
                  and np.array_equal(segs["merge_label"], r.merge_labels))
        planes = res.planes(f)
        if ok and len(planes):
            ok = (np.array_equal(bits(planes["out_normal"]), bits(r.planes[:, 0:3])) and np.array_equal(bits(planes["d"]), bits(r.planes[:, 3]))
                  and np.array_equal(bits(planes["cov"]).reshape(len(planes), 9), bits(r.planes[:, 10:19])))
        kept = res.cylinder_labels(f)
        kept = kept[kept["kept"] == 1]
        if ok and len(kept):
            ok = np.array_equal(bits(kept["axis"]), bits(r.cylinders[:, 0:3])) and bool(np.isnan(kept["radius"]).all())
        return f if not ok else -1, len(r.planes), len(r.cylinders)

    bad, n_planes, n_cyl = [], 0, 0
    with cf.ThreadPoolExecutor(max(2, min(16, os.cpu_count() or 2))) as pool:
        for c0 in range(0, n, chunk):
            host = dev[c0:c0 + chunk].cpu().numpy()
            for f, npl, ncy in pool.map(check, [(c0 + k, host[k]) for k in range(len(host))]):
                n_planes += npl
                n_cyl += ncy
                if f >= 0:
                    bad.append(f)
    if after_extract is None:
        ex.close()
    assert not bad, f"{len(bad)} of {n} frames differ from the oracle, first: {bad[:8]}"
    return (n_planes, n_cyl) if after_extract is None else (n_planes, n_cyl, ex, res, extra)


def test_configs1_every_frame_of_the_4096_batch(oracle_mod):
    """BASELINE.json configs[1], the very batch bench.py times: EVERY one of its 4 096 frames against the oracle."""
    n_planes, _ = _every_frame(oracle_mod, "room", False, 4096)
    assert n_planes > 8000


@pytest.mark.parametrize("scene,n", [("room", 4096), ("tumlike", 2048), ("tunnel", 2048)])
def test_reference_faithful_mode_every_frame(oracle_mod, scene, n):
    """The streams of BASELINE.json configs[1] - [3] with the reference's unconditional cylinder branch on (two-pass schedule, parked
    frames, finisher; the tunnel: the cylinder kernel alone): every frame of the batch against the oracle."""
    n_planes, n_cyl = _every_frame(oracle_mod, scene, True, n)
    assert n_planes + n_cyl > n // 2
    if scene != "tumlike":
        assert n_cyl > 50


def test_configs4_1280x960_every_frame_of_a_room_tunnel_mix(oracle_mod):
    """BASELINE.json configs[4], one-GPU leg, at the standard of the 640 x 480 batches: 1 024 frames of 1280 x 960,
    a room / tunnel MIX (blocks of 64 consecutive frames of either trajectory), planes + cylinders + polygons + polygon
    matches -- EVERY frame against the extraction oracle, every output plane's polygon and every frame pair's match decisions against
    the oracle of the reference's polygon algorithm."""
    import concurrent.futures as cf
    import os

    import torch
    import cape_amd
    import polygon_oracle_py as P
    from cape_amd import synth_gpu
    from test_gpu_polygon_oracle import AREA_RTOL, _center, compare_plane, new_stats

    P.build()
    n, W, H, blk = 1024, 1280, 960, 64
    room = synth_gpu.stream("room", 100, n // 2, width=W, height=H, start=0, device="cuda", chunk=16)
    tun = synth_gpu.stream("tunnel", 100, n // 2, width=W, height=H, start=0, device="cuda", chunk=16)
    dev = torch.empty((n, H, W), dtype=room.dtype, device="cuda")
    for b in range(n // blk):
        src = room if b % 2 == 0 else tun
        dev[b * blk:(b + 1) * blk] = src[(b // 2) * blk:(b // 2 + 1) * blk]
    del room, tun

    def after(ex, m):
        st = torch.cuda.current_stream().cuda_stream
        ex.build_polygons(m, st)
        ex.match_polygons(m, 0, st)
        return ex.polygons(m) + (ex.polygon_matches(m),)

    n_planes, n_cyl, ex, res, (pol, ver, got) = _every_frame(oracle_mod, "room", True, n, chunk=128, W=W, H=H, dev=dev, after_extract=after)
    assert n_planes > n and n_cyl > 32, (n_planes, n_cyl)
    _properties_all_frames(res, n, ex.cells)
    out = res.records["segments"]["is_output"][:n] == 1
    assert not (pol["flags"][out] & cape_amd.POLY_OVERFLOW).any() and not (got["flags"] & cape_amd.MATCH_EXACT_OVERFLOW).any()

    # ---- every output plane's polygon against the oracle of the reference's algorithm (threads: ctypes releases the GIL)
    def planes_of(f):
        st = new_stats()
        keep = []
        for i, s in enumerate(res.segments(f)):
            if not s["is_output"]:
                continue
            p = pol[f, i]
            o, c = int(p["vertex_offset"]), int(p["vertex_count"])
            ref = compare_plane(P, p, ver[f, o:o + c], res.boundary_points(f, s), s["normal"], _center(s), f"frame {f} segment {i}", st)
            if ref is None:
                keep = None  # a dissolved / degenerate / rejected hull: the frame's kept-plane list is not compared
            elif keep is not None and ref.valid and ref.boundary_length() >= 3:
                keep.append((i, np.asarray(s["out_normal"], np.float64), float(s["d"]), ref))
        return st, keep

    stats, kept = new_stats(), []
    with cf.ThreadPoolExecutor(max(2, min(16, os.cpu_count() or 2))) as pool:
        for st, keep in pool.map(planes_of, range(n)):
            kept.append(keep)
            for k in ("planes", "threw", "vertex_identical", "left_out_by_reference_rule"):
                stats[k] += st[k]
            stats["dissolve"] += st["dissolve"]
            stats["degenerate"] = stats.get("degenerate", 0) + st.get("degenerate", 0)
    compared = stats["planes"] - stats["threw"] - len(stats["dissolve"]) - stats.get("degenerate", 0)
    assert compared > n and len(stats["dissolve"]) <= 0.08 * stats["planes"], stats
    assert stats["vertex_identical"] >= 0.98 * compared, stats
    # ---- every frame pair's match decisions against MapPlane::find_matches run by the oracle on its own polygons
    pairs = decided = 0
    for f in range(1, n):
        if kept[f] is None or kept[f - 1] is None:
            continue
        prev, cur = kept[f - 1], kept[f]
        assert [q[0] for q in prev] == list(got[f]["seg_prev"][: len(prev)]) and [q[0] for q in cur] == list(got[f]["seg_cur"][: len(cur)]), f
        want, inter = P.find_matches([q[1:] for q in prev], [q[1:] for q in cur], None, advanced=False, allow_index0=False)
        assert list(got[f]["match"][: len(prev)]) == want, f"frame {f}: {list(got[f]['match'][:len(prev)])} vs {want}"
        decided += sum(1 for m in want if m >= 0)
        for j in range(len(prev)):
            for i in range(len(cur)):
                a, b = float(got[f]["inter_area"][j][i]), float(inter[j, i])
                if b < 0:
                    continue
                assert a >= 0 and abs(a - b) <= AREA_RTOL * max(b, float(cur[i][3].area)) + 1e-6, f"frame {f} pair ({j},{i}): {a} vs {b}"
                pairs += 1
    assert pairs > n // 2 and decided > n // 8, (pairs, decided)
    ex.close()
