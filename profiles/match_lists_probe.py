import sys
sys.path.insert(0, "rgb-d-slam_amd/python")
import numpy as np, torch
from cape_amd import Extractor, synth, synth_gpu
for scene, B, cyl in (("room", 4096, True), ("tumlike", 2048, True)):
    intr = synth.TUM_FR1_INTRINSICS if scene == "tumlike" else synth.DEFAULT_INTRINSICS
    dev = synth_gpu.stream(scene, 100, B, device="cuda")
    ex = Extractor(640, 480, max_batch=B, cylinders=cyl, **intr)
    st = torch.cuda.current_stream().cuda_stream
    ex.extract_device(dev.data_ptr(), B, st); ex.build_polygons(B, st); ex.match_polygons(B, 0, st)
    print(scene, ex.match_lists())
    pol, _ = ex.polygons(B)
    vc = pol["vertex_count"][pol["vertex_count"] > 0]
    print("  ring sizes: mean %.1f p90 %d p99 %d max %d ; >32: %.2f %%" % (vc.mean(), np.percentile(vc, 90), np.percentile(vc, 99), vc.max(), 100 * (vc > 32).mean()))
    ex.close()
