"""N2 with the reference's own area measure (cape_match_polygons): MapPlane::find_matches (map_primitive.cpp:91-161)
between consecutive frames, on the device's boundary polygons.

Checker: the host class of this repo (Polygon::inter_area behind the test hook cape_host_polygon_inter_area of
libcape_primitives.so; tests/host/test_polygon.cpp holds it to the reference's polygon tests) for the areas -- compared
BIT FOR BIT -- and the reference's selection loop restated below in Python for the matches."""
import ctypes as C
import math
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def host_inter(host_binaries):
    import cape_amd

    cape_amd.load_library()  # one HIP runtime per process: torch's, then libcape_hip, before the host library
    lib = C.CDLL(os.path.join(host_binaries, "libcape_primitives.so"))
    vp = C.c_void_p
    lib.cape_host_polygon_inter_area.argtypes = [vp, C.c_int, vp, vp, vp, vp, C.c_int, vp, vp, vp, C.POINTER(C.c_double), C.POINTER(C.c_double)]
    lib.cape_host_polygon_inter_area.restype = C.c_double

    def run(ring_a, pa, ring_b, pb):
        ra = np.ascontiguousarray(ring_a, np.float64)
        rb = np.ascontiguousarray(ring_b, np.float64)
        arrs = [np.ascontiguousarray(pa[k], np.float64) for k in ("x_axis", "y_axis", "center")] + \
               [np.ascontiguousarray(pb[k], np.float64) for k in ("x_axis", "y_axis", "center")]
        aa, ab = C.c_double(0), C.c_double(0)
        v = lib.cape_host_polygon_inter_area(ra.ctypes.data_as(vp), len(ra), *[a.ctypes.data_as(vp) for a in arrs[:3]],
                                             rb.ctypes.data_as(vp), len(rb), *[a.ctypes.data_as(vp) for a in arrs[3:]],
                                             C.byref(aa), C.byref(ab))
        return v, aa.value, ab.value

    return run


def _bits(a):
    return np.ascontiguousarray(a, np.float64).view(np.uint64)


def _kept(res, pol, f):
    """Segment indices of the planes Primitive_Detection keeps (primitive_detection.cpp:623-631), in order."""
    import cape_amd

    segs = res.segments(f)
    return [i for i, s in enumerate(segs) if s["is_output"] and (pol[f, i]["flags"] & cape_amd.POLY_VALID) and pol[f, i]["vertex_count"] >= 3]


def _expected(res, pol, ver, f, host_inter, flags):
    """find_matches of every kept plane of frame f-1 (identity pose) against the kept planes of frame f."""
    import cape_amd

    prev, cur = _kept(res, pol, f - 1), _kept(res, pol, f)
    sp, sc = res.segments(f - 1), res.segments(f)
    min_cos = abs(math.cos(20.0 * math.pi / 180.0))
    overlap = float(np.float32(0.4))
    if flags & cape_amd.MATCH_ADVANCED:
        overlap /= 2
    inter = np.full((len(prev), len(cur)), -1.0)
    matched = [False] * len(cur)
    match = []
    for j, sj in enumerate(prev):
        q, pq = sp[sj], pol[f - 1, sj]
        ring_q = ver[f - 1, pq["vertex_offset"]: pq["vertex_offset"] + pq["vertex_count"]]
        selected, greatest = -1, 0.0
        for i, si in enumerate(cur):
            s, ps = sc[si], pol[f, si]
            nq, ns = q["out_normal"], s["out_normal"]
            cos = (ns[0] * nq[0] + ns[1] * nq[1]) + ns[2] * nq[2]
            if not (abs(s["d"] - q["d"]) < 100.0 and abs(cos) > min_cos):
                continue
            ring_s = ver[f, ps["vertex_offset"]: ps["vertex_offset"] + ps["vertex_count"]]
            ia, area_s, area_q = host_inter(ring_s, ps, ring_q, pq)
            # the explicit-ring constructor recomputes the areas from the rings: the same numbers the device stored
            assert _bits(area_s) == _bits(ps["area"]) and _bits(area_q) == _bits(pq["area"])
            inter[j, i] = ia
            if matched[i] or not pq["area"] > 0.0:
                continue
            if ia > greatest and ia / ps["area"] >= overlap:
                selected, greatest = i, ia
        if selected <= 0 and not (flags & cape_amd.MATCH_ALLOW_INDEX0):
            selected = -1
        if selected >= 0:
            matched[selected] = True
        match.append(selected)
    return prev, cur, inter, match


def _check(ex, res, pol, ver, got, n, host_inter, flags):
    import cape_amd

    pairs = matches = 0
    assert got[0]["n_prev"] == 0 and np.all(got[0]["match"] == -1)
    for f in range(1, n):
        prev, cur, inter, match = _expected(res, pol, ver, f, host_inter, flags)
        g = got[f]
        assert g["n_prev"] == len(prev) and g["n_cur"] == len(cur), f"frame {f}: plane counts"
        if len(prev) > cape_amd.MATCH_MAX_PLANES or len(cur) > cape_amd.MATCH_MAX_PLANES:
            assert g["flags"] & cape_amd.MATCH_EXACT_OVERFLOW
            continue
        assert list(g["seg_prev"][: len(prev)]) == prev and list(g["seg_cur"][: len(cur)]) == cur
        assert not (g["flags"] & cape_amd.MATCH_EXACT_OVERFLOW), f"frame {f}: capacity exceeded"
        gi = g["inter_area"][: len(prev), : len(cur)]
        bad = np.argwhere(_bits(gi) != _bits(inter))
        assert len(bad) == 0, f"frame {f}: intersection areas differ at {bad[:4].tolist()}: {gi[tuple(bad[0])]!r} vs {inter[tuple(bad[0])]!r}"
        assert list(g["match"][: len(prev)]) == match, f"frame {f}: matches {list(g['match'][:len(prev)])} vs {match}"
        assert np.all(g["match"][len(prev):] == -1)
        pairs += int((inter >= 0).sum())
        matches += sum(1 for m in match if m >= 0)
    return pairs, matches


@pytest.mark.parametrize("scene,cyl,n,flags", [("tumlike", False, 48, 0), ("tumlike", True, 24, 1), ("room", False, 32, 2), ("room", False, 16, 3),
                                                  ("tumlike", False, 256, 0)])
def test_polygon_matches_of_a_stream(host_inter, scene, cyl, n, flags):
    """A moving-camera stream: every gated pair's intersection area is bit-identical to the host class, the selection equals
    the reference's loop (including its never-returns-index-0 quirk unless ALLOW_INDEX0)."""
    import torch
    from cape_amd import Extractor, synth, synth_gpu

    intr = synth.TUM_FR1_INTRINSICS if scene == "tumlike" else synth.DEFAULT_INTRINSICS
    dev = synth_gpu.stream(scene, 77, n, start=120, device="cuda", chunk=8)
    ex = Extractor(640, 480, cylinders=cyl, max_batch=n, **intr)
    st = torch.cuda.current_stream().cuda_stream
    ex.extract_device(dev.data_ptr(), n, st)
    ex.build_polygons(n, st)
    ex.match_polygons(n, flags, st)
    res = ex.results(n)
    pol, ver = ex.polygons(n)
    got = ex.polygon_matches(n)
    pairs, matches = _check(ex, res, pol, ver, got, n, host_inter, flags)
    assert pairs > n and matches > 0, (pairs, matches)
    ex.close()


def test_polygon_matches_identical_frames(host_inter):
    """The same frame twice: every plane intersects itself with its own area (up to the slab sum's rounding) and, with
    ALLOW_INDEX0, matches itself."""
    import torch
    import cape_amd
    from cape_amd import Extractor, synth, synth_gpu

    one = synth_gpu.stream("room", 55, 1, start=300, device="cuda", chunk=1)
    dev = one.repeat(4, 1, 1).contiguous()
    ex = Extractor(640, 480, cylinders=False, max_batch=4, **synth.DEFAULT_INTRINSICS)
    st = torch.cuda.current_stream().cuda_stream
    ex.extract_device(dev.data_ptr(), 4, st)
    ex.build_polygons(4, st)
    ex.match_polygons(4, cape_amd.MATCH_ALLOW_INDEX0, st)
    res = ex.results(4)
    pol, ver = ex.polygons(4)
    got = ex.polygon_matches(4)
    _check(ex, res, pol, ver, got, 4, host_inter, cape_amd.MATCH_ALLOW_INDEX0)
    for f in range(1, 4):
        g = got[f]
        k = int(g["n_cur"])
        assert k >= 1 and list(g["match"][:k]) == list(range(k))
        for i in range(k):
            area = pol[f, g["seg_cur"][i]]["area"]
            assert abs(g["inter_area"][i, i] - area) <= 1e-9 * area
    ex.close()


def test_polygon_matches_argument_checks():
    import torch
    import cape_amd
    from cape_amd import Extractor, synth, synth_gpu

    dev = synth_gpu.stream("room", 5, 2, start=0, device="cuda", chunk=2)
    ex = Extractor(640, 480, cylinders=False, max_batch=2, **synth.DEFAULT_INTRINSICS)
    st = torch.cuda.current_stream().cuda_stream
    ex.extract_device(dev.data_ptr(), 2, st)
    with pytest.raises(cape_amd.CapeError):  # polygons of this batch not built
        ex.match_polygons(2, 0, st)
    ex.build_polygons(2, st)
    with pytest.raises(cape_amd.CapeError):
        ex.match_polygons(2, 8, st)  # unknown flag
    ex.match_polygons(2, 0, st)
    ex.extract_device(dev.data_ptr(), 2, st)  # a new batch invalidates the polygons
    with pytest.raises(cape_amd.CapeError):
        ex.match_polygons(2, 0, st)
    ex.close()


def test_polygon_matches_1280x960_outlines_beyond_128_vertices(host_inter):
    """BASELINE.json configs[4] geometry: the 64 x 48 cell grid shows outlines of more than 128 vertices (207 on this stream,
    profiles/r04_capacity_probe.txt) -- the fourth capacity tier of the intersection kernel (512 vertices, 2 048 slab
    boundaries).  No frame may fall back to the host class, and the areas stay bit-identical to it."""
    import torch
    import cape_amd
    from cape_amd import Extractor, synth, synth_gpu

    n = 64
    intr = {k: v * 2.0 for k, v in synth.TUM_FR1_INTRINSICS.items()}
    dev = synth_gpu.stream("tumlike", 17, n, width=1280, height=960, start=256, device="cuda", chunk=4)
    ex = Extractor(1280, 960, cylinders=True, max_batch=n, **intr)
    st = torch.cuda.current_stream().cuda_stream
    ex.extract_device(dev.data_ptr(), n, st)
    ex.build_polygons(n, st)
    ex.match_polygons(n, 0, st)
    res = ex.results(n)
    pol, ver = ex.polygons(n)
    got = ex.polygon_matches(n)
    assert int(pol["vertex_count"].max()) > 128, "the stream must show an outline the 128-vertex tiers cannot hold"
    assert not (pol["flags"] & cape_amd.POLY_OVERFLOW).any()
    assert not (got["flags"] & cape_amd.MATCH_EXACT_OVERFLOW).any()
    pairs, matches = _check(ex, res, pol, ver, got, n, host_inter, 0)
    assert pairs > n and matches > 0
    ex.close()
