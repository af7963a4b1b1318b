#!/usr/bin/env python3
"""Row N3 (cape_rectify_depth) on a device-resident batch: time per batch, frames/s and the HBM rate against the algorithmic
bytes (4 B read per source pixel, 4 B written per target pixel: the collision keys live in LDS).
usage: rectify_bench.py [frames=1024] [scene=room]"""
import sys
sys.path.insert(0, "rgb-d-slam_amd/python")
import numpy as np, torch
from cape_amd import Extractor, synth, synth_gpu
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
scene = sys.argv[2] if len(sys.argv) > 2 else "room"
intr = synth.TUM_FR1_INTRINSICS if scene == "tumlike" else synth.DEFAULT_INTRINSICS
dev = synth_gpu.stream(scene, 100, B, device="cuda")
out = torch.empty_like(dev)
ex = Extractor(640, 480, max_batch=B, **intr)
st = torch.cuda.current_stream().cuda_stream
# a depth camera 25 mm beside the colour camera, half a degree of yaw: the usual RGB-D rig
a = np.deg2rad(0.5)
T = np.eye(4)
T[:3, :3] = [[np.cos(a), 0, np.sin(a)], [0, 1, 0], [-np.sin(a), 0, np.cos(a)]]
T[:3, 3] = [-25.0, 1.5, 0.8]
for name, M in (("identity", np.eye(4)), ("25 mm baseline, 0.5 deg yaw", T)):
    ex.rectify_device(dev.data_ptr(), out.data_ptr(), B, M, st)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 10
    e0.record()
    for _ in range(reps):
        ex.rectify_device(dev.data_ptr(), out.data_ptr(), B, M, st)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    valid = float((dev > 0).float().mean())
    px = B * 640 * 480
    alg = px * (4 + 4)
    print("flagged frames (general kernels): %d" % ex.rectify_flagged())
    print("%-28s %8.3f ms per %d frames = %.2f M frames/s ; algorithmic %.1f MB -> %.0f GB/s (%.1f %% of 8 TB/s); valid source pixels %.1f %%, targets hit %.1f %%" % (
        name, ms, B, B / ms / 1e3, alg / 1e6, alg / ms / 1e6, 100 * alg / ms / 1e6 / 8000, 100 * valid, 100 * float((out > 0).float().mean())))
if "--chunks" in sys.argv:  # does the key buffer's residency between the two kernels matter?  the same batch in sub-batches
    fb = 640 * 480 * 4
    for C in (1024, 256, 64, 32, 16, 8, 4):
        if C > B:
            continue
        def run():
            for s in range(0, B, C):
                ex.rectify_device(dev.data_ptr() + s * fb, out.data_ptr() + s * fb, min(C, B - s), T, st)
        run(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            run()
        e1.record(); torch.cuda.synchronize()
        print("  sub-batches of %4d frames: %.3f ms per %d frames" % (C, e0.elapsed_time(e1) / 5, B))
