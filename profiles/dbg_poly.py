import sys, os, ctypes as C
sys.path.insert(0, "rgb-d-slam_amd/python")
import numpy as np
import cape_amd
from cape_amd import Extractor, synth
cape_amd.load_library()
ex = Extractor(640, 480, max_batch=1, **synth.DEFAULT_INTRINSICS)
xy = np.stack([np.linspace(0, 900, 30), np.linspace(0, 900, 30) * 0.5], 1)
for nrm in [(0, 0, 1.0), (0, 0.6, 0.8), (0.48, 0.6, 0.64), (1.0, 0, 0)]:
    nrm = np.asarray(nrm) / np.linalg.norm(nrm)
    center = np.array([120.0, -340.0, 2100.0])
    a = np.cross(nrm, [0.3, -0.5, 0.8]); a /= np.linalg.norm(a); b = np.cross(nrm, a)
    pts = center + xy[:, :1] * a + xy[:, 1:] * b
    pol, verts = ex.debug_polygon(pts, nrm, center)
    print(nrm, "count", pol["vertex_count"], "flags", pol["flags"], "area", pol["area"])
    print(verts[:8])
