import sys, time
sys.path.insert(0,'rgb-d-slam_amd/python')
import numpy as np, torch
from cape_amd import Extractor, synth, synth_gpu
for n in (64, 512, 4096):
    dev = synth_gpu.stream("room", 100, n, device="cuda", chunk=64)
    ex = Extractor(640, 480, cylinders=False, max_batch=n, **synth.DEFAULT_INTRINSICS)
    st = torch.cuda.current_stream().cuda_stream
    ex.extract_device(dev.data_ptr(), n, st)
    ex.build_polygons(n, st); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3): ex.build_polygons(n, st)
    e1.record(); torch.cuda.synchronize()
    print(n, "frames: polygon pass ms", e0.elapsed_time(e1)/3, flush=True)
    ex.close()
