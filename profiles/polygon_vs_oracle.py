#!/usr/bin/env python3
"""Device polygons (row N1, cape_build_polygons) and the device polygon matcher (row N2, cape_match_polygons) against the ORACLE
OF THE REFERENCE'S ALGORITHM (oracle/polygon_oracle.cpp) -- a long sweep over device-rendered streams with noise, holes and
dropped blocks (ragged, concave outlines), not part of the test suite.  Every disagreement is LISTED, nothing is averaged
away: per plane validity / convex-fallback verdict / area ratio / IoU / vertex identity; the planes whose hull the reference
would dissolve with Boost set operations (not restated); per frame pair the matcher's decisions and areas.
usage: polygon_vs_oracle.py [frames_per_scene=256] [seed=3]   (POLY_DUMP=dir: the candidates of every plane that is not vertex-identical)"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in ("rgb-d-slam_amd/python", "tests", "oracle"):
    sys.path.insert(0, os.path.join(ROOT, p))
import numpy as np
import torch
import cape_amd
import polygon_oracle_py as P
from cape_amd import Extractor, synth, synth_gpu

N = int(sys.argv[1]) if len(sys.argv) > 1 else 256
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 3
P.build()
gen = torch.Generator(device="cuda").manual_seed(seed)
B = 64
tot = dict(planes=0, compared=0, vertex_identical=0, validity_mismatch=0, fallback_mismatch=0, iou_below_0999=0, area_beyond_1e9=0,
           reference_would_dissolve=0, cut_at_crossings=0, cut_verdict_mismatch=0, threw=0, convex_fallbacks=0, simplified=0, climbed_ladder=0, candidates_left_out_by_reference_rule=0,
           pairs=0, pair_area_beyond_1e9=0, frames_matched=0, decision_mismatch_frames=0, decisions=0, overflow_frames=0, skipped_frames=0)
worst = dict(iou=1.0, area=0.0, pair_area=0.0)
listing = []
t0 = time.time()


def center(s):
    return np.asarray(s["normal"], np.float64) * (-np.float64(s["d"]))


for scene, cyl in (("room", False), ("tumlike", False), ("tumlike", True), ("tunnel", True)):
    intr = synth.TUM_FR1_INTRINSICS if scene == "tumlike" else synth.DEFAULT_INTRINSICS
    ex = Extractor(640, 480, cylinders=cyl, max_batch=B, **intr)
    st = torch.cuda.current_stream().cuda_stream
    for start in range(0, N, B):
        dev = synth_gpu.stream(scene, seed + start, B, start=start, device="cuda", chunk=16).clone()
        mode = (start // B) % 4
        if mode == 1:
            dev[torch.rand(dev.shape, device="cuda", generator=gen) < 0.05] = 0
        elif mode == 2:
            dev += torch.randn(dev.shape, device="cuda", generator=gen) * 3.0 * (dev > 0)
        elif mode == 3:
            for _ in range(6):
                y, x = int(torch.randint(0, 400, (1,), generator=gen, device="cuda")), int(torch.randint(0, 560, (1,), generator=gen, device="cuda"))
                dev[:, y:y + 80, x:x + 80] = 0
        ex.extract_device(dev.data_ptr(), B, st)
        ex.build_polygons(B, st)
        ex.match_polygons(B, 0, st)
        res = ex.results(B)
        pol, ver = ex.polygons(B)
        got = ex.polygon_matches(B)
        kept, skip = [], set()
        for f in range(B):
            planes = []
            for i, s in enumerate(res.segments(f)):
                if not s["is_output"]:
                    continue
                what = f"{scene}{'+cyl' if cyl else ''} batch {start} frame {f} segment {i}"
                p = pol[f, i]
                pts = res.boundary_points(f, s)
                ref = P.Polygon.from_points(pts, s["normal"], center(s))
                tot["planes"] += 1
                tot["convex_fallbacks"] += int(bool(p["flags"] & cape_amd.POLY_CONVEX_FALLBACK))
                tot["simplified"] += int(bool(p["flags"] & cape_amd.POLY_SIMPLIFIED))
                if ref.threw:
                    tot["threw"] += 1
                    continue
                if ref.flags & P.NEEDS_DISSOLVE:
                    tot["reference_would_dissolve"] += 1
                    skip.add(f)
                    listing.append(f"DISSOLVE   {what}: {len(pts)} candidates, the walk's hull (k = {ref.k_used}) crosses itself; device: "
                                   f"{'convex fallback' if p['flags'] & cape_amd.POLY_CONVEX_FALLBACK else 'flags %d' % p['flags']}")
                    continue
                tot["compared"] += 1
                tot["cut_at_crossings"] += int(bool(ref.flags & P.DISSOLVED))
                if bool(ref.flags & P.DISSOLVED) != bool(p["flags"] & cape_amd.POLY_DISSOLVED):
                    tot["cut_verdict_mismatch"] += 1
                    listing.append(f"CUT        {what}: device flags {p['flags']}, oracle flags {ref.flags}")
                tot["climbed_ladder"] += int(ref.k_used > 3)
                dev_valid = bool(p["flags"] & cape_amd.POLY_VALID) and int(p["vertex_count"]) >= 3
                ref_valid = ref.valid and ref.boundary_length() >= 3
                if ref_valid:
                    planes.append((i, np.asarray(s["out_normal"], np.float64), float(s["d"]), ref))
                if dev_valid != ref_valid:
                    tot["validity_mismatch"] += 1
                    listing.append(f"VALIDITY   {what}: device {dev_valid}, oracle {ref_valid} (flags {ref.flags})")
                    skip.add(f)
                    continue
                if bool(p["flags"] & cape_amd.POLY_CONVEX_FALLBACK) != bool(ref.flags & P.CONVEX_FALLBACK):
                    tot["fallback_mismatch"] += 1
                    listing.append(f"FALLBACK   {what}: device flags {p['flags']}, oracle flags {ref.flags}")
                if not dev_valid:
                    continue
                o, c = int(p["vertex_offset"]), int(p["vertex_count"])
                verts = ver[f, o:o + c]
                d = P.Polygon(verts, p["x_axis"], p["y_axis"], p["center"])
                inter = d.inter_area(ref)
                iou = inter / (d.area + ref.area - inter)
                ratio = float(p["area"]) / ref.area
                same = len(verts) == len(ref.ring) and np.array_equal(verts, ref.ring)
                tot["vertex_identical"] += int(same)
                if not same and os.environ.get("POLY_DUMP"):  # the plane's candidates, for a stand-alone reproduction on the CPU
                    np.savez(os.path.join(os.environ["POLY_DUMP"], f"poly_diff_{seed}_{scene}_{int(cyl)}_{start}_{f}_{i}.npz"), pts=np.asarray(pts, np.float64),
                             normal=np.asarray(s["normal"], np.float64), center=center(s), device_ring=verts, oracle_ring=ref.ring)
                worst["iou"] = min(worst["iou"], iou)
                worst["area"] = max(worst["area"], abs(ratio - 1))
                if iou < 0.999:
                    tot["iou_below_0999"] += 1
                    listing.append(f"IOU        {what}: IoU {iou:.6f}, area ratio {ratio:.6f}, {len(pts)} candidates, vertices {c} vs {len(ref.ring)}, k {ref.k_used}")
                if abs(ratio - 1) > 1e-9:
                    tot["area_beyond_1e9"] += 1
                    if iou >= 0.999:
                        listing.append(f"AREA       {what}: area ratio {ratio:.12f} (IoU {iou:.6f})")
                q = np.asarray(pts, np.float64) - center(s)
                reach = d.simplify_reach() * (1 + 1e-9)
                for v in q:
                    x, y = float(np.dot(p["x_axis"], v)), float(np.dot(p["y_axis"], v))
                    if d.distance_outside(x, y) > reach:
                        tot["candidates_left_out_by_reference_rule"] += 1
            kept.append(planes)
        for f in range(1, B):
            if got[f]["flags"] & cape_amd.MATCH_EXACT_OVERFLOW:
                tot["overflow_frames"] += 1
                continue
            if f in skip or (f - 1) in skip:
                tot["skipped_frames"] += 1
                continue
            prev, cur = kept[f - 1], kept[f]
            want, inter = P.find_matches([q[1:] for q in prev], [q[1:] for q in cur])
            tot["frames_matched"] += 1
            tot["decisions"] += sum(1 for m in want if m >= 0)
            if list(got[f]["match"][: len(prev)]) != want or [q[0] for q in prev] != list(got[f]["seg_prev"][: len(prev)]):
                tot["decision_mismatch_frames"] += 1
                listing.append(f"MATCH      {scene}{'+cyl' if cyl else ''} batch {start} frame {f}: device {list(got[f]['match'][:len(prev)])} oracle {want}")
            for j in range(len(prev)):
                for i in range(len(cur)):
                    b = float(inter[j, i])
                    if b < 0:
                        continue
                    a = float(got[f]["inter_area"][j][i])
                    tot["pairs"] += 1
                    rel = abs(a - b) / max(b, float(cur[i][3].area))
                    worst["pair_area"] = max(worst["pair_area"], rel)
                    if rel > 1e-9:
                        tot["pair_area_beyond_1e9"] += 1
                        listing.append(f"PAIR AREA  {scene} batch {start} frame {f} pair ({j},{i}): device {a:.6f} oracle {b:.6f}")
    ex.close()
    print(scene, "cylinders" if cyl else "planes only", "->", {k: v for k, v in tot.items() if v}, flush=True)
print()
print(f"planes {tot['planes']}: compared {tot['compared']} -- of them {tot['cut_at_crossings']} hulls that cross themselves and are cut apart at the crossing like the "
      f"reference's repair does (verdict mismatches {tot['cut_verdict_mismatch']}); left out: {tot['reference_would_dissolve']} hulls that merely touch themselves "
      f"(the reference re-unites those with Boost set operations: not restated), constructor throws {tot['threw']}")
print(f"vertex-identical to the oracle: {tot['vertex_identical']} of {tot['compared']} compared "
      f"({100.0 * tot['vertex_identical'] / max(1, tot['compared']):.3f} %)")
print(f"worst IoU {worst['iou']:.12f}, worst |area ratio - 1| {worst['area']:.3e}, planes with IoU < 0.999: {tot['iou_below_0999']}, "
      f"validity mismatches {tot['validity_mismatch']}, convex-fallback verdict mismatches {tot['fallback_mismatch']}")
print(f"planes that climbed the ladder (k > 3) {tot['climbed_ladder']}, convex fallbacks {tot['convex_fallbacks']}, simplified {tot['simplified']}, "
      f"boundary candidates left outside by the reference's own containment rule {tot['candidates_left_out_by_reference_rule']}")
print(f"matcher: {tot['frames_matched']} frame pairs ({tot['decisions']} matches), decision mismatches {tot['decision_mismatch_frames']}, "
      f"{tot['pairs']} intersected pairs, worst relative area difference {worst['pair_area']:.3e}, beyond 1e-9: {tot['pair_area_beyond_1e9']}; "
      f"skipped (a plane of the pair dissolves / validity differs) {tot['skipped_frames']}, device overflow frames {tot['overflow_frames']}")
print(f"elapsed {time.time() - t0:.0f} s")
print()
print("LISTING (every disagreement and every plane the comparison leaves out):")
for line in listing:
    print(" ", line)
bad = tot["cut_verdict_mismatch"] + tot["validity_mismatch"] + tot["iou_below_0999"] + tot["decision_mismatch_frames"] + tot["pair_area_beyond_1e9"] + tot["area_beyond_1e9"]
print("RESULT", "OK" if bad == 0 else "DISAGREEMENTS", tot)
