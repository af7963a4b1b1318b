#!/usr/bin/env python3
"""PCIe-inclusive rate of the C ABI: cape_extract_host on batches that live in host memory (pageable numpy, and pinned
memory from cape_host_alloc), against the same batch already resident in HBM.  usage: host_input_rate.py [frames=256]"""
import sys, time
sys.path.insert(0, "rgb-d-slam_amd/python")
import numpy as np, torch
from cape_amd import Extractor, synth, synth_gpu
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
dev = synth_gpu.stream("room", 100, B, device="cuda")
host = dev.cpu().numpy()
ex = Extractor(640, 480, max_batch=B, **synth.DEFAULT_INTRINSICS)
pinned = ex.host_alloc(host.shape)
pinned[...] = host
st = torch.cuda.current_stream().cuda_stream
def timed(fn, reps=8):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps
mb = host.nbytes / 1e6
for name, fn in (("resident in HBM", lambda: ex.extract_device(dev.data_ptr(), B, st)), ("pageable host memory", lambda: ex.extract_host(host, st)),
                 ("pinned host memory (cape_host_alloc)", lambda: ex.extract_host(pinned, st))):
    t = timed(fn)
    print("%-40s %8.3f ms per %d frames = %9.0f frames/s (%.1f GB/s of depth)" % (name, 1e3 * t, B, B / t, mb / t / 1e3))
raw = synth_gpu.stream("room", 100, B, device="cuda", raw_u16=True).cpu().numpy().view(np.uint16)
t = timed(lambda: ex.extract_host_u16(raw, 0.2, st))
print("%-40s %8.3f ms per %d frames = %9.0f frames/s (%.1f GB/s of depth)" % ("raw uint16 in pageable host memory", 1e3 * t, B, B / t, raw.nbytes / 1e6 / t / 1e3))
