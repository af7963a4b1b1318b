#!/usr/bin/env python3
"""Exact polygon matcher (cape_match_polygons) on device-rendered streams: time per batch beside the cell-mask matcher and the
polygon pass, the share of frames / pairs beyond the kernel's capacities, matches per frame and how often the two matchers
agree.  usage: match_polygons_bench.py [frames=2048] [scene=tumlike]"""
import sys
sys.path.insert(0, "rgb-d-slam_amd/python")
import numpy as np, torch
import cape_amd
from cape_amd import Extractor, synth, synth_gpu
B = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
scene = sys.argv[2] if len(sys.argv) > 2 else "tumlike"
intr = synth.TUM_FR1_INTRINSICS if scene == "tumlike" else synth.DEFAULT_INTRINSICS
dev = synth_gpu.stream(scene, 100, B, device="cuda")
ex = Extractor(640, 480, max_batch=B, **intr)
st = torch.cuda.current_stream().cuda_stream
ex.extract_device(dev.data_ptr(), B, st)
ex.build_polygons(B, st)
ex.match_polygons(B, 0, st)
ex.match_consecutive(B, 0, st)
torch.cuda.synchronize()
def timed(fn, reps=10):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
print("scene", scene, "frames", B)
print("  polygon pass        ms %.3f" % timed(lambda: ex.build_polygons(B, st)))
print("  polygon matcher     ms %.3f" % timed(lambda: ex.match_polygons(B, 0, st)))
print("  cell-mask matcher   ms %.3f" % timed(lambda: ex.match_consecutive(B, 0, st)))
got = ex.polygon_matches(B)
cells = ex.matches(B)
ia = got["inter_area"]
nan = np.isnan(ia)
codes = ia.view(np.uint64)[nan] & 0xF
print("  frames flagged OVERFLOW %d of %d; NaN pairs %d (ring %d, slabs %d, stack %d); gated pairs %d" % (
    int((got["flags"] & 1).astype(bool).sum()), B, int(nan.sum()), int((codes == 1).sum()), int((codes == 2).sum()), int((codes == 3).sum()),
    int((ia >= 0).sum())))
print("  kept planes per frame: mean %.2f max %d ; matches per frame %.2f" % (got["n_cur"].mean(), got["n_cur"].max(), (got["match"] >= 0).sum() / B))
if "--ticks" in sys.argv:  # a -DCAPE_MP_PROFILE library (profiles/build_variant.sh mpprof -DCAPE_MP_PROFILE): ticks of every pair
    rows = []
    for f in range(1, B):
        g = got[f]
        for j in range(min(8, int(g["n_prev"]))):
            for i in range(min(8, int(g["n_cur"]))):
                if not (ia[f, j, i] >= 0 or np.isnan(ia[f, j, i])):
                    continue
                a, b, c = ia[f, j + 8, i + 8], ia[f, j + 8, i], ia[f, j, i + 8]
                if not a > 0:
                    continue
                tot, na_, nb_, nx0 = a % 1e6, (a // 1e6) % 1e3, (a // 1e9) % 1e3, a // 1e12
                cross, sort = b % 1e6, b // 1e6
                slabs, nx1, tier = c % 1e6, (c // 1e6) % 1e6, c // 1e12
                rows.append((tot, na_, nb_, nx0, nx1, cross, sort, slabs, tier))
    r = np.array(rows)
    print("  pairs with ticks", len(r), "(units of 16 s_memtime ticks); last tier that ran the pair:", {int(t): int((r[:, 8] == t).sum()) for t in np.unique(r[:, 8])})
    for q in (50, 90, 99, 100):
        k = np.argsort(r[:, 0])[min(len(r) - 1, int(len(r) * q / 100))]
        print("   p%-3d total %6d ticks: na %3d nb %3d boundaries %4d -> %4d slabs; crossings %6d sort+unique %6d slabs %6d (tier %d)" % (q, *r[k]))
    print("   sum of ticks %.0f ; mean %.0f" % (r[:, 0].sum(), r[:, 0].mean()))
    sys.exit(0)
# agreement with the cell-mask matcher (its indices count OUTPUT planes; map through the segment lists)
res = ex.results(B, with_boundary=False)
segs = res.records["segments"]
agree = total = 0
for f in range(1, B):
    g = got[f]
    if g["flags"] & 1:
        continue
    outs_prev = [i for i in range(64) if segs[f - 1, i]["is_output"]][: 64]
    outs_cur = [i for i in range(64) if segs[f, i]["is_output"]]
    c = cells[f]
    for j in range(min(int(g["n_prev"]), 16)):
        sj = int(g["seg_prev"][j])
        if sj not in outs_prev:
            continue
        cj = outs_prev.index(sj)
        mc = int(c["match"][cj]) if cj < len(c["match"]) else -1
        mp = int(g["match"][j])
        seg_c = outs_cur[mc] if 0 <= mc < len(outs_cur) else -1
        seg_p = int(g["seg_cur"][mp]) if mp >= 0 else -1
        total += 1
        agree += seg_c == seg_p
print("  same decision as the cell-mask matcher for %d of %d previous planes (%.1f %%)" % (agree, total, 100.0 * agree / max(1, total)))
