#!/usr/bin/env python3
"""Per-phase shader-clock breakdown of the grow kernel (needs a -DCAPE_B_PROFILE build: CAPE_HIP_LIB=...libcape_prof.so)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "rgb-d-slam_amd", "python"))
import numpy as np
import torch
from cape_amd import Extractor, synth

NAMES = ["hist + edge-mask prologue", "hist argmax", "candidate scan", "seed pick", "propagation", "list build", "ordered accum",
         "hist removal + record", "region fits (lane parallel)", "seed-loop tail", "merge", "boundary+records", "cyl: cov+eigen", "cyl: projection", "cyl: RANSAC", "",
         "cyl: LLS + merged-plane sums (one ordered pass)", "cyl: ids compaction", "cyl: MSE addends (parallel)", "cyl: MSE decision", "",
         "cyl: plane fit", "cyl: select+labels", "cov staged: consume + other sets", "#RANSAC rounds", "#cells in region", "#inliers", "#hypotheses", "cyl: cov pass 1", "cyl: cov pass 2", "cov staged: hand-over of set a (vmcnt wait)", "cov staged: request of set a"]
B = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
scene = sys.argv[2] if len(sys.argv) > 2 else "room"
W = int(sys.argv[4]) if len(sys.argv) > 4 else 640
H = int(sys.argv[5]) if len(sys.argv) > 5 else 480
if os.environ.get("CAPE_PHASES_16"):
    u = synth.stream(scene, seed=100, n_frames=16, width=W, height=H)  # the round-2 input: 16 distinct frames, tiled
    d = torch.from_numpy(u).cuda().repeat(B // 16, 1, 1).contiguous()
else:
    from cape_amd import synth_gpu
    d = synth_gpu.stream(scene, 100, B, width=W, height=H, device="cuda", chunk=64 if W <= 640 else 16)  # bench.py's stream: every frame distinct
intr = synth.TUM_FR1_INTRINSICS if scene == "tumlike" else synth.DEFAULT_INTRINSICS
intr = {k: v * W / 640.0 for k, v in intr.items()}
cyl = len(sys.argv) > 3 and sys.argv[3] == "cyl"
ex = Extractor(W, H, max_batch=B, cylinders=cyl, **intr)
ex.extract_device(d.data_ptr(), B, torch.cuda.current_stream().cuda_stream)
cyc = ex.debug_cycles(B).astype(np.float64)
tot = cyc[:, :23].sum(1) + cyc[:, 28:30].sum(1)  # slots 23, 30, 31 are sub-intervals of the covariance passes
if os.environ.get("CAPE_PHASES_GROUP"):  # sub-intervals of the group kernel's phases reuse the seed-loop slots (it has no seed loop)
    for k, nm in {1: "group RANSAC: scoring", 2: "group RANSAC: wave reductions", 3: "group RANSAC: barrier", 4: "group RANSAC: replay",
                  5: "group LLS: consume (wave 0)", 6: "group LLS: wait for producers"}.items():
        NAMES[k] = nm
print(f"frames {B}: mean ticks/frame {tot.mean():.0f} (s_memtime = shader clock, ~2.3 GHz => {tot.mean() / 2300:.1f} us)  seeds/frame {ex.results(B, False).records['header']['n_seeds'].mean():.1f}")
for k, nm in enumerate(NAMES):
    if nm.startswith("#"):
        print(f"  {nm:22s} {cyc[:, k].mean():10.1f} per frame")
    elif nm and cyc[:, k].any():
        print(f"  {nm:22s} {cyc[:, k].mean():10.0f} ticks  {100 * cyc[:, k].mean() / tot.mean():5.1f} %")

# the tail: a second-pass round lasts as long as its slowest frame
order = np.argsort(tot)
for label, sel in (("slowest 1 %", order[-max(1, B // 100):]), ("median 10 %", order[int(B * 0.45):int(B * 0.55)])):
    sub = cyc[sel]
    st = tot[sel]
    print(f"{label}: mean ticks/frame {st.mean():.0f} (max {st.max():.0f})")
    for k, nm in enumerate(NAMES):
        if nm.startswith("#"):
            print(f"    {nm:22s} {sub[:, k].mean():10.1f} per frame")
        elif nm and sub[:, k].mean() > 0.01 * st.mean():
            print(f"    {nm:22s} {sub[:, k].mean():10.0f} ticks  {100 * sub[:, k].mean() / st.mean():5.1f} %")
print("percentiles of ticks/frame:", {q: int(np.percentile(tot, q)) for q in (50, 75, 90, 99, 100)})
