#!/usr/bin/env python3
"""Boundary-polygon pass (cape_build_polygons) on a device-rendered stream: time per batch, statistics of the planes (boundary
points, flags, vertices), the latency of small batches and -- with CAPE_POLY_PHASES=1 and a -DCAPE_POLY_PROFILE library
(profiles/build_variant.sh polyprof -DCAPE_POLY_PROFILE; CAPE_HIP_LIB=.../libcape_polyprof.so) -- the phase ticks per plane.
usage: polygon_phases.py [frames=1024] [scene=room]"""
import sys, time, os
sys.path.insert(0, "rgb-d-slam_amd/python")
import numpy as np, torch
import cape_amd
from cape_amd import Extractor, synth, synth_gpu
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
scene = sys.argv[2] if len(sys.argv) > 2 else "room"
intr = synth.TUM_FR1_INTRINSICS if scene == "tumlike" else synth.DEFAULT_INTRINSICS
dev = synth_gpu.stream(scene, 100, B, device="cuda")
ex = Extractor(640, 480, max_batch=B, **intr)
st = torch.cuda.current_stream().cuda_stream
ex.extract_device(dev.data_ptr(), B, st)
ex.build_polygons(B, st)
torch.cuda.synchronize()
t0 = time.perf_counter(); ex.build_polygons(B, st); torch.cuda.synchronize(); print("ms", 1e3 * (time.perf_counter() - t0))
res = ex.results(B, with_boundary=False)
pol, _ = ex.polygons(B)
segs = res.records["segments"]
out = segs["is_output"] == 1
bc = segs["boundary_count"][out]
fl = pol["flags"][out]
vc = pol["vertex_count"][out]
print("planes", out.sum(), "boundary points: mean %.1f p50 %d p90 %d p99 %d max %d" % (bc.mean(), *np.percentile(bc, [50, 90, 99]), bc.max()))
print("flags: valid %.3f convex_fallback %.3f simplified %.3f overflow %d" % ((fl & 1).astype(bool).mean(), (fl & 2).astype(bool).mean(), (fl & 4).astype(bool).mean(), (fl & 8).astype(bool).sum()))
print("vertices: mean %.1f max %d" % (vc.mean(), vc.max()))
# latency of a lone plane-wave: one frame
for nb in (1, 64, 512):
    ex1 = Extractor(640, 480, max_batch=max(nb, 16), **intr)
    ex1.extract_device(dev.data_ptr(), nb, st)
    ex1.build_polygons(nb, st); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        ex1.build_polygons(nb, st)
    e1.record(); torch.cuda.synchronize()
    print("frames", nb, "polygon pass ms", e0.elapsed_time(e1) / 10)
    ex1.close()
if os.environ.get("CAPE_POLY_PHASES"):
    ex.build_polygons(B, st); torch.cuda.synchronize()
    cyc = ex.debug_cycles(B).astype(np.float64).sum(0)
    planes = cyc[11]
    names = ["projection + sort + dedupe", "hull walks", "simple-ring test of hulls", "orientation + re-check (+ convex fallback)", "area + simplify", "final validity + stores"]
    tot = cyc[:6].sum()
    print("per plane: %.0f ticks; attempts %.2f; vertices before simplify %.1f; distinct points %.1f" % (tot / planes, cyc[8] / planes, cyc[9] / planes, cyc[10] / planes))
    for k, nm in enumerate(names):
        print("  %-45s %8.0f ticks  %5.1f %%" % (nm, cyc[k] / planes, 100 * cyc[k] / tot))
if os.environ.get("CAPE_POLY_PHASES"):
    raw = ex.debug_cycles(B).astype(np.float64)
    print("task kernel: static tasks %d (%.0f ticks each to get), spawned / idle-exit acquisitions %d (%.0f ticks each)" % (
        raw[0, 26], raw[0, 24] / max(1, raw[0, 26]), raw[0, 27], raw[0, 25] / max(1, raw[0, 27])))
    t = ex.debug_cycles(B)[0].astype(np.int64)
    print("   timeline (us after the first wave's start, wall_clock64): planes of the batch handed out %.1f, last polygon %.1f, last wave leaves %.1f; busy %.0f shader ticks per wave (%d waves)" % (
        (t[7] - t[6]) / 100.0, (t[23] - t[6]) / 100.0, (t[31] - t[6]) / 100.0, float(t[28]) / (ex.compute_units * 16), ex.compute_units * 16))
if os.environ.get("CAPE_POLY_PHASES"):
    rawi = ex.debug_cycles(B).astype(np.int64)
    has = rawi[:, 22] > 0
    t0 = rawi[0, 6]
    end = (rawi[:, 22] - t0) / 100.0  # wall_clock64 ticks of 10 ns -> us
    print("   frames' last polygon finished at (us after the kernel's first wave started): percentiles", {q: int(np.percentile(end[has], q)) for q in (10, 50, 90, 99, 99.9, 100)})
    for f in np.argsort(np.where(has, end, 0))[-8:]:
        print("      frame %d: last polygon at %.1f us, %d candidates, winning rung %d (0 = convex fallback)" % (f, end[f], rawi[f, 21], rawi[f, 20] - 1))
if os.environ.get("CAPE_POLY_PHASES"):
    per = ex.debug_cycles(B).astype(np.float64)
    t = per[:, :6].sum(1)
    print("per-frame ticks percentiles:", {q: int(np.percentile(t, q)) for q in (50, 90, 99, 99.9, 100)})
    w = np.argsort(t)[-5:]
    for f in w:
        print("  frame", f, "ticks", int(t[f]), "planes", int(per[f, 11]), "attempts", int(per[f, 8]), "hull ticks", int(per[f, 1]), "points", int(per[f, 10]))
if os.environ.get("CAPE_POLY_PHASES"):
    lad = per[:, 12:23].sum(0)
    nl = lad[0] + lad[2:8].sum()
    if nl > 0:
        print("ladder kernel: %d planes (%.1f %% of all); winning rung k=5: %d  k=7: %d  k=11: %d  k=13: %d  k=17: %d  k=21: %d  none (convex fallback): %d" % (
            nl, 100 * nl / per[:, 11].sum(), *lad[2:8], lad[0]))
        print("  points per ladder plane %.1f ; ticks per ladder plane (both stages, the three waves side by side) mean %.0f ; slowest plane %d ticks" % (
            lad[8] / nl, lad[9] / nl, per[:, 22].max()))
        sl = np.sort(per[:, 22])[::-1][:8]
        print("  the eight slowest frames' slowest ladder plane:", [int(v) for v in sl])
