#!/usr/bin/env python3
"""Long GPU-vs-host-class sweep of the two polygon rows (not part of the test suite): every boundary polygon of device-rendered
streams (row N1, cape_build_polygons) against the host class vertex for vertex, and every gated pair of consecutive frames (row
N2, cape_match_polygons) against Polygon::inter_area + the reference's selection loop.  Noise, holes and dropped blocks are
added to the frames so that ragged outlines, concave hulls and deep edge stacks occur.
usage: fuzz_polygons.py [frames_per_scene=512] [seed=3]"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in ("rgb-d-slam_amd/python", "tests"):
    sys.path.insert(0, os.path.join(ROOT, p))
import numpy as np
import torch
import cape_amd
from cape_amd import Extractor, synth, synth_gpu
import test_gpu_match_polygon as TM
import test_gpu_polygon as TP

N = int(sys.argv[1]) if len(sys.argv) > 1 else 512
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 3
cape_amd.load_library()
lib = C.CDLL(os.path.join(ROOT, "rgb-d-slam_amd", "lib", "libcape_primitives.so"))
vp = C.c_void_p
lib.cape_host_polygon.argtypes = [vp, C.c_int, vp, vp, vp, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_double), vp, vp, C.POINTER(C.c_int)]
lib.cape_host_polygon_inter_area.argtypes = [vp, C.c_int, vp, vp, vp, vp, C.c_int, vp, vp, vp, C.POINTER(C.c_double), C.POINTER(C.c_double)]
lib.cape_host_polygon_inter_area.restype = C.c_double


def host_poly(points3, normal, center):
    pts = np.ascontiguousarray(points3, np.float64).reshape(-1, 3)
    nrm, ctr = np.ascontiguousarray(normal, np.float64), np.ascontiguousarray(center, np.float64)
    ring = np.zeros((max(1, len(pts)), 2), np.float64)
    cnt, valid, area = C.c_int(0), C.c_int(0), C.c_double(0)
    xa, ya = np.zeros(3), np.zeros(3)
    rc = lib.cape_host_polygon(pts.ctypes.data_as(vp), len(pts), nrm.ctypes.data_as(vp), ctr.ctypes.data_as(vp), ring.ctypes.data_as(vp),
                               len(ring), C.byref(cnt), C.byref(area), xa.ctypes.data_as(vp), ya.ctypes.data_as(vp), C.byref(valid))
    return dict(threw=rc != 0, ring=ring[: cnt.value], area=area.value, x_axis=xa, y_axis=ya, valid=bool(valid.value))


def host_inter(ring_a, pa, ring_b, pb):
    ra, rb = np.ascontiguousarray(ring_a, np.float64), np.ascontiguousarray(ring_b, np.float64)
    arrs = [np.ascontiguousarray(pa[k], np.float64) for k in ("x_axis", "y_axis", "center")] + \
           [np.ascontiguousarray(pb[k], np.float64) for k in ("x_axis", "y_axis", "center")]
    aa, ab = C.c_double(0), C.c_double(0)
    v = lib.cape_host_polygon_inter_area(ra.ctypes.data_as(vp), len(ra), *[a.ctypes.data_as(vp) for a in arrs[:3]],
                                         rb.ctypes.data_as(vp), len(rb), *[a.ctypes.data_as(vp) for a in arrs[3:]], C.byref(aa), C.byref(ab))
    return v, aa.value, ab.value


gen = torch.Generator(device="cuda").manual_seed(seed)
tot = dict(planes=0, polygon_mismatches=0, pairs=0, matches=0, match_mismatches=0, overflow_frames=0, convex_fallbacks=0, simplified=0)
B = 64
for scene, cyl in (("room", False), ("tumlike", False), ("tumlike", True), ("tunnel", True)):
    intr = synth.TUM_FR1_INTRINSICS if scene == "tumlike" else synth.DEFAULT_INTRINSICS
    ex = Extractor(640, 480, cylinders=cyl, max_batch=B, **intr)
    st = torch.cuda.current_stream().cuda_stream
    for start in range(0, N, B):
        dev = synth_gpu.stream(scene, seed + start, B, start=start, device="cuda", chunk=16).clone()
        mode = (start // B) % 4
        if mode == 1:   # speckle holes
            dev[torch.rand(dev.shape, device="cuda", generator=gen) < 0.05] = 0
        elif mode == 2:  # depth noise
            dev += torch.randn(dev.shape, device="cuda", generator=gen) * 3.0 * (dev > 0)
        elif mode == 3:  # dropped blocks: ragged, concave outlines
            for _ in range(6):
                y, x = int(torch.randint(0, 400, (1,), generator=gen, device="cuda")), int(torch.randint(0, 560, (1,), generator=gen, device="cuda"))
                dev[:, y:y + 80, x:x + 80] = 0
        ex.extract_device(dev.data_ptr(), B, st)
        ex.build_polygons(B, st)
        ex.match_polygons(B, 0, st)
        res = ex.results(B)
        pol, ver = ex.polygons(B)
        got = ex.polygon_matches(B)
        for f in range(B):
            for i, s in enumerate(res.segments(f)):
                if not s["is_output"]:
                    continue
                p = pol[f, i]
                ref = host_poly(res.boundary_points(f, s), s["normal"], TP._center(s))
                o, c = int(p["vertex_offset"]), int(p["vertex_count"])
                tot["planes"] += 1
                tot["convex_fallbacks"] += int(bool(p["flags"] & 2))
                tot["simplified"] += int(bool(p["flags"] & 4))
                try:
                    TP._same(p, ver[f, o:o + c], ref, "")
                except AssertionError as e:
                    tot["polygon_mismatches"] += 1
                    print("POLYGON MISMATCH", scene, start, f, i, e)
        for f in range(1, B):
            g = got[f]
            if g["flags"] & 1:
                tot["overflow_frames"] += 1
                continue
            prev, cur, inter, match = TM._expected(res, pol, ver, f, host_inter, 0)
            gi = g["inter_area"][: len(prev), : len(cur)]
            ok = (g["n_prev"] == len(prev) and g["n_cur"] == len(cur) and np.array_equal(gi.view(np.uint64), inter.view(np.uint64))
                  and list(g["match"][: len(prev)]) == match)
            tot["pairs"] += int((inter >= 0).sum())
            tot["matches"] += sum(1 for m in match if m >= 0)
            if not ok:
                tot["match_mismatches"] += 1
                print("MATCH MISMATCH", scene, start, f)
    ex.close()
    print(scene, "cylinders" if cyl else "planes only", "->", tot, flush=True)
print("RESULT", "OK" if tot["polygon_mismatches"] == 0 and tot["match_mismatches"] == 0 else "MISMATCH", tot)
