"""Stage A of a one-frame call with the frame already in HBM: the strip kernel (one launch) against the two throughput kernels
(CAPE_STAGE_A=bands), by the library's own HIP events.  usage: python profiles/stage_a_latency.py"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "rgb-d-slam_amd", "python"))
import numpy as np
import torch
from cape_amd import Extractor, synth

def run(mode, scene, cyl):
    if mode:
        os.environ["CAPE_STAGE_A"] = mode
    else:
        os.environ.pop("CAPE_STAGE_A", None)
    intr = synth.DEFAULT_INTRINSICS
    frames = [getattr(synth, scene)(seed=0, frame=s) for s in range(8)]
    dev = torch.from_numpy(np.stack(frames)).cuda()
    ex = Extractor(640, 480, cylinders=cyl, max_batch=1, **intr)
    st = torch.cuda.current_stream().cuda_stream
    for k in range(16):
        ex.extract_device(dev[k % 8].data_ptr(), 1, st)
    torch.cuda.synchronize()
    ex.reset_timings()
    ex.enable_timing(True)
    t0 = time.perf_counter()
    n = 400
    for k in range(n):
        ex.extract_device(dev[k % 8].data_ptr(), 1, st)
        torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / n
    t = ex.timings()
    c = max(1, t["calls"])
    print(f"{scene:8s} cyl={int(cyl)} stage A = {mode or 'strips':6s}: A (moments+plane) {1e6 * (t['cell_moments_s'] + t['cell_plane_s']) / c:6.1f} us   "
          f"grow {1e6 * t['grow_s'] / c:6.1f} us   call+sync {1e6 * wall:6.1f} us")
    ex.close()

for scene, cyl in (("room", False), ("tumlike", True)):
    for mode in (None, "bands"):
        run(mode, scene, cyl)
