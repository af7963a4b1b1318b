#!/usr/bin/env python3
"""cape_match_polygons by the capacities of the intersection kernel's first tier: time per batch and how many pairs leave each tier for
which reason (cape_debug_match_lists).  Run once per library variant (CAPE_HIP_LIB=rgb-d-slam_amd/lib/exp/libcape_mp_*.so,
profiles/build_tu_variant.sh mp_<name> cape_match_polygon.hip -DCAPE_MP_T0_STACK=.. -DCAPE_MP_T0_XS=.. -DCAPE_MP_T0_GROUPS=..)."""
import os
import sys

sys.path.insert(0, "rgb-d-slam_amd/python")
import torch
from cape_amd import Extractor, synth, synth_gpu

name = os.path.basename(os.environ.get("CAPE_HIP_LIB", "default"))
for scene, B in (("room", 4096), ("tumlike", 2048)):
    intr = synth.TUM_FR1_INTRINSICS if scene == "tumlike" else synth.DEFAULT_INTRINSICS
    dev = synth_gpu.stream(scene, 100, B, device="cuda")
    ex = Extractor(640, 480, max_batch=B, cylinders=True, **intr)
    st = torch.cuda.current_stream().cuda_stream
    ex.extract_device(dev.data_ptr(), B, st)
    ex.build_polygons(B, st)
    for _ in range(2):
        ex.match_polygons(B, 0, st)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        ex.match_polygons(B, 0, st)
    e1.record()
    torch.cuda.synchronize()
    got = ex.polygon_matches(B)
    print(f"{name:28s} {scene:8s} {e0.elapsed_time(e1) / 10:.3f} ms  lists {ex.match_lists()}  matches {int((got['match'] >= 0).sum())} flagged {int((got['flags'] & 1).astype(bool).sum())}")
    ex.close()
