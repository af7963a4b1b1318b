#!/usr/bin/env python3
"""How often do the device polygon / polygon-matcher capacities bite on BASELINE.json configs[4] (1280x960, planes + cylinders +
matching) and on the 640x480 streams?  Counts CAPE_POLY_OVERFLOW planes (more boundary candidates than the device hull takes),
frames with more kept planes than the matcher's table, and CAPE_MATCH_EXACT_OVERFLOW frames (with the reason).
usage: capacity_probe.py [frames=512]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "rgb-d-slam_amd", "python"))
import numpy as np
import torch
import cape_amd
from cape_amd import Extractor, synth, synth_gpu

N = int(sys.argv[1]) if len(sys.argv) > 1 else 512
for W, H, scene, cyl in ((1280, 960, "tunnel", True), (1280, 960, "room", True), (1280, 960, "tumlike", True), (640, 480, "room", True),
                         (640, 480, "tumlike", True), (640, 480, "tunnel", True)):
    s = W / 640.0
    base = synth.TUM_FR1_INTRINSICS if scene == "tumlike" else synth.DEFAULT_INTRINSICS
    intr = {k: v * s for k, v in base.items()}
    B = 64
    ex = Extractor(W, H, cylinders=cyl, max_batch=B, **intr)
    st = torch.cuda.current_stream().cuda_stream
    tot = dict(frames=0, planes=0, poly_overflow=0, max_candidates=0, planes_over_256=0, planes_over_1024=0, max_kept_planes=0, frames_over_16_planes=0,
               match_overflow_frames=0, pair_capacity_frames=0, convex_fallback=0, invalid=0, max_vertices=0, pairs_beyond_ring=0, pairs_beyond_slabs=0, pairs_beyond_stack=0)
    for start in range(0, N, B):
        dev = synth_gpu.stream(scene, 17, B, width=W, height=H, start=start, device="cuda", chunk=4)
        ex.extract_device(dev.data_ptr(), B, st)
        ex.build_polygons(B, st)
        ex.match_polygons(B, 0, st)
        res = ex.results(B, with_boundary=False)
        pol, _ = ex.polygons(B)
        got = ex.polygon_matches(B)
        segs = res.records["segments"]
        out = segs["is_output"] == 1
        bc = segs["boundary_count"][out]
        fl = pol["flags"][out]
        tot["frames"] += B
        tot["planes"] += int(out.sum())
        tot["poly_overflow"] += int(((fl & cape_amd.POLY_OVERFLOW) != 0).sum())
        tot["convex_fallback"] += int(((fl & cape_amd.POLY_CONVEX_FALLBACK) != 0).sum())
        tot["invalid"] += int(((fl & cape_amd.POLY_VALID) == 0).sum())
        tot["max_candidates"] = max(tot["max_candidates"], int(bc.max()) if len(bc) else 0)
        tot["planes_over_256"] += int((bc > 256).sum())
        tot["planes_over_1024"] += int((bc > 1024).sum())
        kept = (out & ((pol["flags"] & cape_amd.POLY_VALID) != 0) & (pol["vertex_count"] >= 3)).sum(1)
        tot["max_kept_planes"] = max(tot["max_kept_planes"], int(kept.max()))
        tot["frames_over_16_planes"] += int((kept > cape_amd.MATCH_MAX_PLANES).sum())
        ov = (got["flags"] & cape_amd.MATCH_EXACT_OVERFLOW) != 0
        tot["match_overflow_frames"] += int(ov.sum())
        tot["pair_capacity_frames"] += int(np.isnan(got["inter_area"]).any(axis=(1, 2)).sum())
        codes = got["inter_area"].view(np.uint64)[np.isnan(got["inter_area"])] & np.uint64(7)
        tot["pairs_beyond_ring"] += int((codes == 1).sum())
        tot["pairs_beyond_slabs"] += int((codes == 2).sum())
        tot["pairs_beyond_stack"] += int((codes == 3).sum())
        tot["max_vertices"] = max(tot["max_vertices"], int(pol["vertex_count"].max()))
        big = np.argwhere(pol["vertex_count"] > 128)
        if len(big):
            print("   outlines of more than 128 vertices: frames", sorted(set(int(start + f) for f, _ in big)))
    ex.close()
    print(f"{W}x{H} {scene} cylinders={cyl}: {tot}", flush=True)
