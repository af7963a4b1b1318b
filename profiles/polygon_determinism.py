#!/usr/bin/env python3
"""cape_build_polygons on one small batch, many times: every run must leave the same bytes.  usage: polygon_determinism.py [runs=300]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "rgb-d-slam_amd", "python"))
import numpy as np, torch
from cape_amd import Extractor, synth
runs = int(sys.argv[1]) if len(sys.argv) > 1 else 300
intr = synth.DEFAULT_INTRINSICS
names = ["room", "tunnel", "facets", "room", "tunnel", "facets", "room"]
frames = np.stack([synth.SCENES[n](seed=11 + i, frame=2 * i) for i, n in enumerate(names)])
dev = torch.from_numpy(frames).cuda()
n = len(frames)
st = torch.cuda.current_stream().cuda_stream
ref = None
bad = 0
for r in range(runs):
    ex = Extractor(640, 480, cylinders=True, max_batch=64, **intr)
    ex.extract_device(dev.data_ptr(), n, st)
    ex.build_polygons(n, st)
    pol, ver = ex.polygons(n)
    res = ex.results(n)
    if ref is None:
        ref = (pol.copy(), ver.copy(), res.records.tobytes())
    else:
        if res.records.tobytes() != ref[2]:
            print("run", r, "EXTRACTION differs")
            bad += 1
        elif pol.tobytes() != ref[0].tobytes() or ver.tobytes() != ref[1].tobytes():
            bad += 1
            for f in range(n):
                for i in range(64):
                    a, b = ref[0][f, i], pol[f, i]
                    va = ref[1][f, a["vertex_offset"]: a["vertex_offset"] + a["vertex_count"]]
                    vb = ver[f, b["vertex_offset"]: b["vertex_offset"] + b["vertex_count"]]
                    if a.tobytes() != b.tobytes() or va.tobytes() != vb.tobytes():
                        seg = res.segments(f)[i] if i < len(res.segments(f)) else None
                        print("run", r, "frame", f, "segment", i, "flags", int(a["flags"]), int(b["flags"]), "count", int(a["vertex_count"]), int(b["vertex_count"]),
                              "area", a["area"], b["area"], "boundary points", int(seg["boundary_count"]) if seg is not None else -1)
                        if a["vertex_count"] == b["vertex_count"]:
                            d = np.argwhere((va != vb).any(1)).ravel()
                            print("   differing vertices", d[:10].tolist(), va[d[:2]].tolist(), vb[d[:2]].tolist())
    ex.close()
print("runs", runs, "differing", bad)
