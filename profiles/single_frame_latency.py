#!/usr/bin/env python3
"""Latency of the drop-in call pattern: ONE host frame in, primitives out (cape_extract_host + cape_copy_results), the
way rgbd_slam.cpp:291-297 calls find_primitives.  Host-side wall clock, PCIe both ways included."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "rgb-d-slam_amd", "python"))
import numpy as np
from cape_amd import Extractor, synth

for scene, cyl in (("room", False), ("room", True), ("tumlike", True), ("tunnel", True)):
    intr = synth.TUM_FR1_INTRINSICS if scene == "tumlike" else synth.DEFAULT_INTRINSICS
    frames = [np.ascontiguousarray(getattr(synth, scene)(seed=7, frame=f)) for f in range(16)]
    ex = Extractor(640, 480, cylinders=cyl, max_batch=1, **intr)
    for f in frames[:4]:
        ex.extract_host(f[None])
        ex.results(1)
    ts = []
    for rep in range(20):
        for f in frames:
            t0 = time.perf_counter()
            ex.extract_host(f[None])
            ex.results(1)
            ts.append(time.perf_counter() - t0)
    ts = np.array(ts) * 1e6
    print(f"{scene:8s} cylinders={int(cyl)}  single-frame latency: median {np.median(ts):7.1f} us  p90 {np.percentile(ts, 90):7.1f} us"
          f"  => {1e6 / np.median(ts):7.0f} frames/s one at a time")
    ex.close()
