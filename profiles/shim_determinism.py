#!/usr/bin/env python3
"""Runs tests/host/test_shim.exe many times on the batch of tests/test_gpu_host_shim.py::test_sharded_batch_matches_oracle and
compares the outputs: any difference between two runs of the same input is a race.  usage: shim_determinism.py [runs=200] [shards=0]"""
import os, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "rgb-d-slam_amd", "python"))
import numpy as np
from cape_amd import synth
runs = int(sys.argv[1]) if len(sys.argv) > 1 else 200
shards = sys.argv[2] if len(sys.argv) > 2 else "0"
exe = os.path.join(ROOT, "rgb-d-slam_amd", "lib", "test_shim.exe")
intr = synth.DEFAULT_INTRINSICS
names = ["room", "tunnel", "facets", "room", "tunnel", "facets", "room"]
frames = np.stack([synth.SCENES[n](seed=11 + i, frame=2 * i) for i, n in enumerate(names)])
path = os.path.join(tempfile.mkdtemp(), "batch.f32")
frames.tofile(path)
args = [exe, path, "640", "480", str(intr["fx"]), str(intr["fy"]), str(intr["cx"]), str(intr["cy"]), str(len(frames) - 1), shards]
ref = None
bad = 0
for r in range(runs):
    out = subprocess.run(args, capture_output=True, text=True, timeout=120)
    if out.returncode != 0:
        print("run", r, "returned", out.returncode, out.stderr[-400:])
        bad += 1
        continue
    if ref is None:
        ref = out.stdout
    elif out.stdout != ref:
        bad += 1
        a, b = ref.splitlines(), out.stdout.splitlines()
        diff = [(x, y) for x, y in zip(a, b) if x != y][:6]
        print("run", r, "differs:", len(a), len(b), diff, [ln for ln in out.stderr.splitlines() if "polygon" in ln][:4])
print("runs", runs, "differing / failing", bad)
