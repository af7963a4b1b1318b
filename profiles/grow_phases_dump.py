#!/usr/bin/env python3
"""Dumps the per-frame phase ticks of a -DCAPE_B_PROFILE build (see grow_phases.py) to an .npz for offline analysis."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "rgb-d-slam_amd", "python"))
import numpy as np
import torch
from cape_amd import Extractor, synth, synth_gpu

B = int(sys.argv[1]); scene = sys.argv[2]; out = sys.argv[3]
W = int(sys.argv[4]) if len(sys.argv) > 4 else 640
H = int(sys.argv[5]) if len(sys.argv) > 5 else 480
d = synth_gpu.stream(scene, 100, B, width=W, height=H, device="cuda", chunk=64 if W <= 640 else 16)
intr = synth.TUM_FR1_INTRINSICS if scene == "tumlike" else synth.DEFAULT_INTRINSICS
intr = {k: v * W / 640.0 for k, v in intr.items()}
ex = Extractor(W, H, max_batch=B, cylinders=True, **intr)
ex.extract_device(d.data_ptr(), B, torch.cuda.current_stream().cuda_stream)
cyc = ex.debug_cycles(B)
res = ex.results(B, False)
np.savez_compressed(out, cyc=cyc, header=res.records["header"])
