import sys
sys.path.insert(0,'rgb-d-slam_amd/python')
import torch
from cape_amd import Extractor, synth, synth_gpu
out=[]
for scene,n in (("room",4096),("tumlike",2048)):
    intr = synth.TUM_FR1_INTRINSICS if scene=="tumlike" else synth.DEFAULT_INTRINSICS
    dev = synth_gpu.stream(scene, 100, n, device="cuda", chunk=64)
    ex = Extractor(640, 480, cylinders=False, max_batch=n, **intr)
    st = torch.cuda.current_stream().cuda_stream
    ex.extract_device(dev.data_ptr(), n, st); ex.build_polygons(n, st); ex.match_polygons(n, 0, st); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(8): ex.match_polygons(n, 0, st)
    e1.record(); torch.cuda.synchronize()
    out.append("%s %d: match %.3f ms" % (scene, n, e0.elapsed_time(e1)/8))
    ex.close()
print(" | ".join(out))
