import sys
sys.path.insert(0,'rgb-d-slam_amd/python')
import numpy as np, torch
from cape_amd import Extractor, synth, synth_gpu
n=24
dev = synth_gpu.stream("room", 55, n, start=300, device="cuda", chunk=8)
ex = Extractor(640, 480, cylinders=False, max_batch=n, **synth.DEFAULT_INTRINSICS)
st = torch.cuda.current_stream().cuda_stream
ex.extract_device(dev.data_ptr(), n, st)
for rep in range(3):
    ex.build_polygons(n, st)
    res = ex.results(n); pol, ver = ex.polygons(n)
    for f in range(n):
        for i, s in enumerate(res.segments(f)):
            if s["is_output"] and s["boundary_count"]>=125:
                fl=int(pol[f,i]["flags"])
                print(rep, f,i,int(s["boundary_count"]),'verts',int(pol[f,i]["vertex_count"]),'flags',fl&0xff,'winner',((fl>>8)&15)-1,'by rung',(fl>>12)&15,'state done %02x hulls %02x'%((fl>>16)&0xff,(fl>>24)&0xff))
