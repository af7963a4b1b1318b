#!/usr/bin/env python3
"""Where does the plane-only first pass stop paying?  Batches with a given fraction of all-cylinder (tunnel) frames among
room frames, grow-kernel time with the schedule forced either way (CAPE_SCHEDULE=two|single, a debug knob)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1:
    sys.path.insert(0, os.path.join(ROOT, "rgb-d-slam_amd", "python"))
    import numpy as np
    import torch
    from cape_amd import Extractor, synth

    frac = float(sys.argv[1])
    B = 2048
    room = synth.stream("room", seed=100, n_frames=16)
    tun = synth.stream("tunnel", seed=100, n_frames=16)
    k = int(round(frac * 32))
    uniq = np.concatenate([tun[: (k + 1) // 2], room[: 16 - (k + 1) // 2], tun[: k // 2], room[: 16 - k // 2]])
    d = torch.from_numpy(uniq).cuda().repeat(B // len(uniq), 1, 1).contiguous()
    ex = Extractor(640, 480, cylinders=True, max_batch=B, **synth.DEFAULT_INTRINSICS)
    st = torch.cuda.current_stream().cuda_stream
    for _ in range(3):
        ex.extract_device(d.data_ptr(), B, st)
    torch.cuda.synchronize()
    ex.enable_timing(True)
    ex.reset_timings()
    for _ in range(10):
        ex.extract_device(d.data_ptr(), B, st)
    torch.cuda.synchronize()
    tm = ex.timings()
    print(f"{tm['grow_s'] / tm['calls'] * 1e3:.3f}")
else:
    print("tunnel fraction   two-pass ms   single-pass ms   auto ms   (grow kernels, 2048 frames)")
    for frac in (0.0, 0.25, 0.5, 0.625, 0.75, 0.875, 1.0):
        r = []
        for sched in ("two", "single", "auto"):
            env = dict(os.environ, CAPE_SCHEDULE=sched)
            r.append(subprocess.run([sys.executable, __file__, str(frac)], env=env, capture_output=True, text=True).stdout.strip().splitlines()[-1])
        print(f"{frac:14.3f}   {r[0]:>11s}   {r[1]:>14s}   {r[2]:>7s}")
