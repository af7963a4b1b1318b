#!/usr/bin/env python3
"""Randomised sweep of row N3 (cape_rectify_depth) against the oracle's deterministic rectify_depth: random rigs (rotation about a random
axis by up to 4 degrees, baselines up to 80 mm, now and then a big roll or a near scene: the frames the band kernel hands to the general
kernels), device-rendered frames with holes and noise, both image sizes.  Every pixel's bit pattern must agree.  Not part of the suite.
usage: fuzz_rectify.py [trials=200] [seed=1]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in ("rgb-d-slam_amd/python", "oracle", "tests"):
    sys.path.insert(0, os.path.join(ROOT, p))
import numpy as np
import torch
import cape_oracle_py as O
from cape_amd import Extractor, synth

trials = int(sys.argv[1]) if len(sys.argv) > 1 else 200
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
B = 4
bad = flagged_total = frames_total = 0
rigs = {}
for W, H in ((640, 480), (1280, 960)):
    intr = {k: v * W / 640.0 for k, v in synth.DEFAULT_INTRINSICS.items()}
    rigs[(W, H)] = (intr, Extractor(W, H, cylinders=False, max_batch=B, **intr), O.Oracle(W, H, cylinders=False, **intr))
names = ["room", "tumlike", "tunnel", "facets"]
st = torch.cuda.current_stream().cuda_stream
for t in range(trials):
    W, H = (640, 480) if t % 4 else (1280, 960)
    intr, ex, orc = rigs[(W, H)]
    frames = np.stack([synth.SCENES[names[int(rng.integers(0, 4))]](seed=int(rng.integers(0, 10000)), frame=int(rng.integers(0, 900)), width=W, height=H, intr=intr)
                       for _ in range(B)])
    mode = int(rng.integers(0, 6))
    if mode == 1:
        frames[rng.random(frames.shape) < 0.1] = 0
    elif mode == 2:
        frames += (rng.standard_normal(frames.shape) * 4.0).astype(np.float32) * (frames > 0)
    elif mode == 3:
        frames[int(rng.integers(0, B))] *= np.float32(rng.uniform(0.05, 0.3))  # a near scene
    axis = rng.standard_normal(3)
    axis /= np.linalg.norm(axis)
    ang = np.deg2rad(rng.uniform(0, 8.0) if mode == 4 else rng.uniform(0, 4.0) * rng.random())
    K = np.array([[0, -axis[2], axis[1]], [axis[2], 0, -axis[0]], [-axis[1], axis[0], 0]])
    T = np.eye(4)
    T[:3, :3] = np.eye(3) + np.sin(ang) * K + (1 - np.cos(ang)) * (K @ K)
    T[:3, 3] = rng.uniform(-80, 80, 3) * np.array([1.0, 0.5, 0.2])
    din = torch.from_numpy(frames).cuda()
    dout = torch.empty_like(din)
    ex.rectify_device(din.data_ptr(), dout.data_ptr(), B, T, st)
    got = dout.cpu().numpy()
    flagged_total += ex.rectify_flagged()
    for f in range(B):
        frames_total += 1
        if not np.array_equal(got[f].view(np.uint32), orc.rectify(frames[f], T).view(np.uint32)):
            bad += 1
            print(f"MISMATCH trial {t} frame {f} ({W}x{H}, mode {mode}, angle {np.rad2deg(ang):.2f} deg)", flush=True)
    if (t + 1) % 20 == 0:
        print(f"{t + 1} trials, {frames_total} frames, flagged for the general kernels {flagged_total}, mismatches {bad}", flush=True)
print("RESULT", "OK" if bad == 0 else f"{bad} MISMATCHES", {"frames": frames_total, "flagged": flagged_total})
sys.exit(1 if bad else 0)
