"""N1 / N2 on the device against the ORACLE OF THE REFERENCE'S ALGORITHM (oracle/polygon_oracle.cpp: an independent
restatement of third_party/concave_fitting.cpp + src/utils/polygon.cpp, pinned by the reference's tests/test_polygons.cpp,
see tests/test_polygon_oracle.py) -- not against this repo's host class, which tests/test_gpu_polygon.py keeps as the
bit-for-bit regression.

Bars: for every output plane -- validity identical, area within 1e-9 relative, every boundary
candidate inside the outline or within Douglas-Peucker's reach of it, IoU(device, oracle) >= 0.999; for every frame pair --
the matcher's decisions identical and its areas within 1e-9 relative.  A plane whose hull the reference would DISSOLVE with
Boost set operations (correct_boost_polygon.hpp:229-330: the walk crossed itself; the oracle does not restate Boost's overlay)
is counted and listed, never averaged in."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

AREA_RTOL = 1e-9
IOU_MIN = 0.999


@pytest.fixture(scope="module")
def P():
    import polygon_oracle_py

    polygon_oracle_py.build()
    return polygon_oracle_py


def _center(s):
    return np.asarray(s["normal"], np.float64) * (-np.float64(s["d"]))


def compare_plane(P, pol, verts, pts, normal, center, what, stats):
    """One plane: the device's record + ring against the oracle fed with the same boundary candidates."""
    import cape_amd

    ref = P.Polygon.from_points(pts, normal, center)
    stats["planes"] += 1
    if ref.threw:
        assert pol["flags"] & cape_amd.POLY_REJECTED, what
        stats["threw"] += 1
        return None
    if ref.flags & P.NEEDS_DISSOLVE:
        stats["dissolve"].append(what)
        return None
    # a point set that is a line up to rounding noise has no outline to agree on: whether its sliver of a hull "touches itself"
    # is decided by the last bit of a cross product (Boost's own verdict on it is not known either) -- counted, not compared
    q = (np.asarray(pts, np.float64) - np.asarray(center, np.float64)) @ np.stack([ref.x_axis, ref.y_axis], 1)
    extent = float(np.max(q.max(0) - q.min(0)))
    if ref.area <= 1e-9 * extent * extent:
        stats["degenerate"] = stats.get("degenerate", 0) + 1
        return None
    dev_valid = bool(pol["flags"] & cape_amd.POLY_VALID) and int(pol["vertex_count"]) >= 3
    assert dev_valid == (ref.valid and ref.boundary_length() >= 3), f"{what}: validity {dev_valid} vs oracle flags {ref.flags}"
    assert bool(pol["flags"] & cape_amd.POLY_CONVEX_FALLBACK) == bool(ref.flags & P.CONVEX_FALLBACK), f"{what}: convex fallback"
    assert np.allclose(pol["x_axis"], ref.x_axis, rtol=0, atol=1e-15) and np.allclose(pol["y_axis"], ref.y_axis, rtol=0, atol=1e-15), what
    if not dev_valid:
        return ref
    assert abs(float(pol["area"]) - ref.area) <= AREA_RTOL * ref.area, f"{what}: area {float(pol['area'])} vs {ref.area}"
    dev = P.Polygon(verts, pol["x_axis"], pol["y_axis"], pol["center"])
    inter = dev.inter_area(ref)
    iou = inter / (dev.area + ref.area - inter)
    stats["worst_iou"] = min(stats["worst_iou"], iou)
    stats["worst_area"] = max(stats["worst_area"], abs(float(pol["area"]) / ref.area - 1))
    assert iou >= IOU_MIN, f"{what}: IoU {iou}"
    same = len(verts) == len(ref.ring) and np.array_equal(np.asarray(verts), ref.ring)
    stats["vertex_identical"] += int(same)
    stats["k_used"][ref.k_used] = stats["k_used"].get(ref.k_used, 0) + 1
    # every boundary candidate is inside the outline, on it, or within the simplification's reach of it
    d = np.asarray(pts, np.float64) - np.asarray(center, np.float64)
    reach = dev.simplify_reach() * (1 + 1e-9)
    for q in d:
        x, y = float(np.dot(pol["x_axis"], q)), float(np.dot(pol["y_axis"], q))
        far = dev.distance_outside(x, y)
        if far > reach:
            # the reference's own containment rule lets a hull leave points out on its right (concave_fitting.cpp:416-417):
            # the oracle's outline must leave the same point out
            assert ref.distance_outside(x, y) > reach * 0.999, f"{what}: candidate {far:.2f} mm outside the device outline only"
            stats["left_out_by_reference_rule"] += 1
    return ref


def new_stats():
    return dict(planes=0, threw=0, dissolve=[], vertex_identical=0, worst_iou=1.0, worst_area=0.0, k_used={}, left_out_by_reference_rule=0)


@pytest.mark.parametrize("scene,cyl,n", [("room", False, 48), ("tumlike", True, 48), ("tunnel", True, 16)])
def test_device_polygons_against_the_reference_algorithm(P, scene, cyl, n):
    import torch
    from cape_amd import Extractor, synth, synth_gpu

    intr = synth.TUM_FR1_INTRINSICS if scene == "tumlike" else synth.DEFAULT_INTRINSICS
    dev = synth_gpu.stream(scene, 21, n, start=100, device="cuda", chunk=8)
    ex = Extractor(640, 480, cylinders=cyl, max_batch=n, **intr)
    st = torch.cuda.current_stream().cuda_stream
    ex.extract_device(dev.data_ptr(), n, st)
    ex.build_polygons(n, st)
    res = ex.results(n)
    pol, ver = ex.polygons(n)
    stats = new_stats()
    for f in range(n):
        for i, s in enumerate(res.segments(f)):
            if not s["is_output"]:
                continue
            p = pol[f, i]
            o, c = int(p["vertex_offset"]), int(p["vertex_count"])
            compare_plane(P, p, ver[f, o:o + c], res.boundary_points(f, s), s["normal"], _center(s), f"{scene} frame {f} segment {i}", stats)
    compared = stats["planes"] - stats["threw"] - len(stats["dissolve"])
    assert compared >= n // 2
    assert len(stats["dissolve"]) <= 0.08 * stats["planes"], stats["dissolve"]
    # the exact turn predicates order like the reference's angles: the hulls are not merely close, they are the same vertices
    assert stats["vertex_identical"] >= 0.98 * compared, stats
    ex.close()


def test_hand_made_shapes_against_the_reference_algorithm(P):
    """Concave outlines, a dense disc (the ladder beyond k = 3), duplicates (kept: the reference's call path removes none),
    collinear points (no hull: convex fallback), tilted frames."""
    from cape_amd import Extractor, synth

    rng = np.random.default_rng(5)
    ex = Extractor(640, 480, max_batch=1, **synth.DEFAULT_INTRINSICS)

    def lift(xy, normal, center):
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
        "disc": rng.normal(0, 200, (600, 2)),
        "duplicates": np.repeat(grid(lambda i, j: True, n=6), 3, axis=0),
        "collinear": np.stack([np.linspace(0, 900, 30), np.linspace(0, 900, 30) * 0.5], 1),
        "three": np.array([(0, 0), (100, 0), (0, 100.0)]),
        "thin": np.stack([np.linspace(0, 2000, 80), rng.normal(0, 0.5, 80)], 1),
    }
    stats = new_stats()
    for name, xy in shapes.items():
        for nrm in [(0, 0, 1.0), (0, 0.6, 0.8), (0.48, 0.6, 0.64), (1.0, 0, 0)]:
            nrm = np.asarray(nrm) / np.linalg.norm(nrm)
            nrm = nrm / np.linalg.norm(nrm)
            center = np.array([120.0, -340.0, 2100.0])
            pts = lift(np.asarray(xy, float), nrm, center)
            pol, verts = ex.debug_polygon(pts, nrm, center)
            compare_plane(P, pol, verts, pts, nrm, center, f"{name} normal {nrm}", stats)
    assert stats["planes"] - stats["threw"] - len(stats["dissolve"]) >= 24, stats
    ex.close()


def test_polygon_matches_against_the_reference_algorithm(P):
    """cape_match_polygons against MapPlane::find_matches run by the oracle on ORACLE-built polygons: same decisions, areas
    within 1e-9 relative, for every consecutive frame pair of a moving stream (identity pose: what the entry point assumes)."""
    import torch
    import cape_amd
    from cape_amd import Extractor, synth, synth_gpu

    n = 24
    for scene, flags in (("tumlike", 0), ("room", cape_amd.MATCH_ADVANCED | cape_amd.MATCH_ALLOW_INDEX0)):
        intr = synth.TUM_FR1_INTRINSICS if scene == "tumlike" else synth.DEFAULT_INTRINSICS
        dev = synth_gpu.stream(scene, 9, n, start=10, device="cuda", chunk=8)
        ex = Extractor(640, 480, cylinders=True, max_batch=n, **intr)
        st = torch.cuda.current_stream().cuda_stream
        ex.extract_device(dev.data_ptr(), n, st)
        ex.build_polygons(n, st)
        ex.match_polygons(n, flags, st)
        res = ex.results(n)
        pol, ver = ex.polygons(n)
        got = ex.polygon_matches(n)
        kept = []  # per frame: [(segment, normal, d, oracle polygon)] in the order Primitive_Detection keeps them
        skip = set()
        for f in range(n):
            planes = []
            for i, s in enumerate(res.segments(f)):
                if not s["is_output"]:
                    continue
                ref = P.Polygon.from_points(res.boundary_points(f, s), s["normal"], _center(s))
                if ref.flags & P.NEEDS_DISSOLVE:
                    skip.add(f)
                    continue
                if ref.valid and ref.boundary_length() >= 3:
                    planes.append((i, np.asarray(s["out_normal"], np.float64), float(s["d"]), ref))
            kept.append(planes)
        pairs = decided = 0
        for f in range(1, n):
            if f in skip or (f - 1) in skip or (got[f]["flags"] & cape_amd.MATCH_EXACT_OVERFLOW):
                continue
            prev, cur = kept[f - 1], kept[f]
            assert [q[0] for q in prev] == list(got[f]["seg_prev"][: len(prev)]) and [q[0] for q in cur] == list(got[f]["seg_cur"][: len(cur)])
            want, inter = P.find_matches([q[1:] for q in prev], [q[1:] for q in cur], None, advanced=bool(flags & cape_amd.MATCH_ADVANCED),
                                         allow_index0=bool(flags & cape_amd.MATCH_ALLOW_INDEX0))
            assert list(got[f]["match"][: len(prev)]) == want, f"{scene} frame {f}: {list(got[f]['match'][:len(prev)])} vs {want}"
            decided += sum(1 for m in want if m >= 0)
            for j in range(len(prev)):
                for i in range(len(cur)):
                    a, b = float(got[f]["inter_area"][j][i]), float(inter[j, i])
                    if b < 0:
                        # the oracle skips planes already matched (their area is never computed); the device computes every gated pair
                        continue
                    assert a >= 0 and abs(a - b) <= AREA_RTOL * max(b, float(cur[i][3].area)) + 1e-6, f"{scene} frame {f} pair ({j},{i}): {a} vs {b}"
                    pairs += 1
        assert pairs > n and decided > n // 2, (pairs, decided)
        ex.close()


def test_device_polygons_1280x960_against_the_reference_algorithm(P):
    """BASELINE.json configs[4] geometry: the 64 x 48 cell grid, whose planes have several hundred boundary candidates -- the
    1 024-point instance of the device hull (the whole ladder in one wave) -- against the oracle of the reference's algorithm."""
    import torch
    from cape_amd import Extractor, synth, synth_gpu

    n = 12
    stats = new_stats()
    big = 0
    for scene in ("room", "tumlike"):
        base = synth.TUM_FR1_INTRINSICS if scene == "tumlike" else synth.DEFAULT_INTRINSICS
        intr = {k: v * 2.0 for k, v in base.items()}
        dev = synth_gpu.stream(scene, 33, n, width=1280, height=960, start=20, device="cuda", chunk=4)
        ex = Extractor(1280, 960, cylinders=True, max_batch=n, **intr)
        st = torch.cuda.current_stream().cuda_stream
        ex.extract_device(dev.data_ptr(), n, st)
        ex.build_polygons(n, st)
        res = ex.results(n)
        pol, ver = ex.polygons(n)
        for f in range(n):
            for i, s in enumerate(res.segments(f)):
                if not s["is_output"]:
                    continue
                p = pol[f, i]
                o, c = int(p["vertex_offset"]), int(p["vertex_count"])
                big += int(s["boundary_count"] > 256)
                compare_plane(P, p, ver[f, o:o + c], res.boundary_points(f, s), s["normal"], _center(s), f"1280x960 {scene} frame {f} segment {i}", stats)
        ex.close()
    compared = stats["planes"] - stats["threw"] - len(stats["dissolve"])
    assert big >= 4, "the wide grid must show planes beyond the 256-point task kernel"
    assert compared >= 2 * n and stats["vertex_identical"] >= 0.98 * compared, stats


def test_polygons_of_a_spilled_frame_against_the_reference_algorithm(P):
    """A frame of 116 plane segments is a chain of two records; cape_build_polygons builds the polygons of the spill record's
    planes with the batch's (its polygon row / vertex slab sit at the record's own index): every plane of the chain is compared
    with the oracle of the reference's algorithm, the matcher leaves such a frame to the host class (CAPE_MATCH_EXACT_OVERFLOW)."""
    import torch
    from cape_amd import Extractor, synth, MATCH_EXACT_OVERFLOW
    from test_gpu_parity import _checkerboard_of_facets

    W, H = 1280, 960
    big, intr = _checkerboard_of_facets(W, H)
    room = synth.room(seed=1, frame=0, width=W, height=H, intr=intr)
    frames = np.stack([room, big, big, room])
    dev = torch.from_numpy(frames).cuda()
    ex = Extractor(W, H, cylinders=True, max_batch=len(frames), **intr)
    st = torch.cuda.current_stream().cuda_stream
    n = len(frames)
    ex.extract_device(dev.data_ptr(), n, st)
    ex.build_polygons(n, st)
    ex.match_polygons(n, 0, st)
    res = ex.results(n)
    pol, ver = ex.polygons(n)
    used = ex.spill_info()[0]
    assert used == 2
    spol, sver = ex.spill_polygons(0, used)
    stats = new_stats()
    in_spill = 0
    for f in range(n):
        chain = res.chain(f)
        assert len(chain) == (2 if f in (1, 2) else 1)
        for part, (rec, slab) in enumerate(chain):
            k = int(res.records["header"]["next_record"][f]) - ex.max_batch if part else None
            prow, vslab = (pol[f], ver[f]) if part == 0 else (spol[k], sver[k])
            for i in range(min(64, int(rec["header"]["n_plane_segments"]))):
                s = rec["segments"][i]
                if not s["is_output"]:
                    continue
                p = prow[i]
                o, c = int(p["vertex_offset"]), int(p["vertex_count"])
                pts = slab[int(s["boundary_offset"]): int(s["boundary_offset"]) + int(s["boundary_count"])]
                compare_plane(P, p, vslab[o:o + c], pts, s["normal"], _center(s), f"frame {f} record {part} segment {i}", stats)
                in_spill += part
    assert in_spill >= 40, "the planes of the spill records were compared too"
    compared = stats["planes"] - stats["threw"] - len(stats["dissolve"])
    assert stats["vertex_identical"] >= 0.98 * compared, stats
    m = ex.polygon_matches(n)
    assert [bool(int(m["flags"][f]) & MATCH_EXACT_OVERFLOW) for f in range(n)] == [False, True, True, True]
    ex.close()


def test_hull_that_touches_itself_on_the_device(P):
    """The comb of tests/test_polygon_oracle.py::test_hull_that_touches_itself_is_flagged_and_measured on the device.  The oracle flags
    NEEDS_DISSOLVE (its validity test counts a contact along collinear points; the reference would re-unite the traced pieces with
    Boost).  The device -- same statements as the host class -- looks for PROPER crossings only, finds none, keeps the walk's ring and
    simplifies it (the doubled stretch along the tooth goes with Douglas-Peucker): a valid polygon whose area is within 2 % of the region
    the walk's ring covers by the non-zero winding rule.  That is the measured size of the one N1 divergence that is left."""
    import cape_amd
    from cape_amd import Extractor, synth
    from test_polygon_oracle import _comb_touching_itself, winding_area

    pts = _comb_touching_itself()
    nrm, ctr = np.array([0.0, 0.0, 1.0]), np.zeros(3)
    ref = P.Polygon.from_points(pts, nrm, ctr)
    assert ref.flags & P.NEEDS_DISSOLVE
    covered = winding_area(ref.ring)
    ex = Extractor(640, 480, max_batch=1, **synth.DEFAULT_INTRINSICS)
    pol, verts = ex.debug_polygon(pts, nrm, ctr)
    assert pol["flags"] & cape_amd.POLY_VALID and not pol["flags"] & cape_amd.POLY_OVERFLOW, hex(int(pol["flags"]))
    assert abs(float(pol["area"]) / covered - 1.0) < 0.02, (float(pol["area"]), covered)
    ex.close()


def test_neighbours_at_nearly_equal_distances_on_the_device(P):
    """tests/golden/polygon_near_ties.npz through cape_debug_polygon.  Distances: the walk's fast key orders two neighbours whose
    squared distances agree in their upper 54 bits by index; the kernel notices (`sameBucket`) and selects that step again on the
    full bit patterns (the re-inserted start point under the id n).  Directions: closer than 2 DBL_EPSILON they are one direction
    (same_direction_line).  Either way the ring is the oracle's."""
    import cape_amd
    from cape_amd import Extractor, synth
    from test_polygon_oracle import _near_ties

    ex = Extractor(640, 480, max_batch=1, **synth.DEFAULT_INTRINSICS)
    for name, pts, nrm, ctr, ring, area, flags, k in _near_ties():
        pol, verts = ex.debug_polygon(pts, nrm, ctr)
        assert bool(pol["flags"] & cape_amd.POLY_VALID) == bool(flags & P.VALID), name
        assert bool(pol["flags"] & cape_amd.POLY_DISSOLVED) == bool(flags & P.DISSOLVED), name
        assert np.array_equal(verts, ring), name
    ex.close()


def test_direction_gray_zone_divergence_on_the_device(P):
    """tests/test_polygon_oracle.py::test_direction_gray_zone_divergence_is_pinned on the device: the same 10-vertex ring as the host
    class (the oracle's has 7), IoU > 0.995."""
    import os

    from cape_amd import Extractor, synth

    d = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "polygon_gray_zone_device_dump.npz"))
    ex = Extractor(640, 480, max_batch=1, **synth.DEFAULT_INTRINSICS)
    pol, verts = ex.debug_polygon(d["pts"], d["normal"], d["center"])
    ex.close()
    assert np.array_equal(verts, d["product_ring"])
    ref = P.Polygon.from_points(d["pts"], d["normal"], d["center"])
    mine = P.Polygon(verts, pol["x_axis"], pol["y_axis"], pol["center"])
    inter = mine.inter_area(ref)
    assert inter / (mine.area + ref.area - inter) > 0.995
