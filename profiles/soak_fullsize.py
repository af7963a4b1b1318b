#!/usr/bin/env python3
"""Soak of the BATCH path at full size (not part of the test suite): 4 096-frame batches of device-rendered streams with fresh seeds,
a third of the frames damaged on the device (dropped pixels, a dropped block, a scaled block, extra noise), EVERY frame of every
batch against the oracle -- the hand-over lists between the grow instances (redo / resume classes / spill) see thousands of frames
per launch here, which the 64-frame batches of fuzz_parity.py do not give them.
usage: soak_fullsize.py [seconds=300] [first_seed=1000] [frames=4096] [width=640] [height=480]   (CAPE_GROW=general in the environment: every
frame through the general grow instance)"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in ("rgb-d-slam_amd/python", "oracle", "tests"):
    sys.path.insert(0, os.path.join(ROOT, p))
import torch
import cape_oracle_py as O
from cape_amd import synth_gpu
from test_gpu_fullsize import _every_frame

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 300.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
n = int(sys.argv[3]) if len(sys.argv) > 3 else 4096
W = int(sys.argv[4]) if len(sys.argv) > 4 else 640
H = int(sys.argv[5]) if len(sys.argv) > 5 else 480
t0 = time.time()
batches = frames = planes = cyls = 0
bad = 0
while time.time() - t0 < budget:
    scene = ("room", "tumlike", "tunnel")[seed % 3]
    cyl = (seed // 3) % 2 == 0
    dev = synth_gpu.stream(scene, seed, n, width=W, height=H, start=(seed * 37) % 500, device="cuda", chunk=64 if W <= 640 else 16)
    g = torch.Generator(device="cuda").manual_seed(seed)
    # damage on the device: frames 0 mod 3 lose pixels, 1 mod 6 lose a block and get a scaled block, 4 mod 6 get noise
    k = torch.arange(n, device="cuda")
    drop = torch.rand(dev.shape, device="cuda", generator=g) < 0.08
    dev[(k % 3 == 0)] = torch.where(drop[(k % 3 == 0)], torch.zeros((), device="cuda"), dev[(k % 3 == 0)])
    del drop
    blk = dev[(k % 6 == 1)]
    blk[:, 100:180, 200:330] = 0.0
    blk[:, 300:380, 400:520] *= 0.7
    dev[(k % 6 == 1)] = blk
    noisy = dev[(k % 6 == 4)]
    noisy += torch.randn(noisy.shape, device="cuda", generator=g) * 3.0 * (noisy > 0)
    dev[(k % 6 == 4)] = noisy
    del blk, noisy
    try:
        p_, c_ = _every_frame(O, scene, cyl, n, W=W, H=H, dev=dev)
        planes += p_
        cyls += c_
    except AssertionError as e:
        bad += 1
        print(f"MISMATCH seed {seed} {scene} cylinders={cyl}: {str(e)[:200]}", flush=True)
    batches += 1
    frames += n
    seed += 1
    print(f"{batches} batches ({frames} frames) checked in {time.time() - t0:.0f} s, mismatching batches {bad}", flush=True)
print(f"coverage: batches {batches}, frames {frames}, planes {planes}, cylinders {cyls}")
print("RESULT", "OK" if bad == 0 else f"{bad} BATCHES WITH MISMATCHES")
sys.exit(1 if bad else 0)
