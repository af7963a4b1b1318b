#!/usr/bin/env python3
"""Latency of the drop-in call pattern: ONE host frame in, primitives out, the way rgbd_slam.cpp:291-297 calls
find_primitives.  The measurement itself is C++ (profiles/single_frame_latency.cpp -> rgb-d-slam_amd/lib/latency_bench.exe);
this script renders the frames and runs it per scene.  usage: single_frame_latency.py > profiles/rNN_single_frame_latency.txt"""
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "rgb-d-slam_amd", "python"))
import numpy as np
from cape_amd import synth

exe = os.path.join(ROOT, "rgb-d-slam_amd", "lib", "latency_bench.exe")
for scene, cyl in (("room", 0), ("room", 1), ("tumlike", 1), ("tunnel", 1)):
    intr = synth.TUM_FR1_INTRINSICS if scene == "tumlike" else synth.DEFAULT_INTRINSICS
    frames = np.stack([getattr(synth, scene)(seed=7, frame=f) for f in range(16)])
    with tempfile.NamedTemporaryFile(suffix=".f32") as tf:
        frames.tofile(tf.name)
        print(f"== {scene}, cylinders={cyl} (abi rows; the overlay always runs the cylinder branch)", flush=True)
        out = subprocess.run([exe, tf.name, str(len(frames)), "640", "480", str(intr["fx"]), str(intr["fy"]), str(intr["cx"]),
                              str(intr["cy"]), str(cyl)], capture_output=True, text=True)
        print(out.stdout, end="")
        if out.returncode:
            print("FAILED", out.returncode, out.stderr[-500:])
