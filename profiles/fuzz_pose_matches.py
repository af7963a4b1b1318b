#!/usr/bin/env python3
"""Sweep of cape_match_polygons_pose against MapPlane::find_matches run by the polygon oracle (oracle polygons, oracle intersection,
oracle to_camera_space) over strided device-rendered streams with their true relative poses -- the loop of
tests/test_gpu_match_pose.py::test_pose_matches_equal_the_reference_algorithm, counting instead of asserting (not part of the suite).
usage: fuzz_pose_matches.py [runs=24] [first_seed=100] [frames_per_run=48]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in ("rgb-d-slam_amd/python", "tests", "oracle"):
    sys.path.insert(0, os.path.join(ROOT, p))
import numpy as np
import cape_amd
import polygon_oracle_py as P
from test_gpu_match_pose import _run

runs = int(sys.argv[1]) if len(sys.argv) > 1 else 24
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 100
n = int(sys.argv[3]) if len(sys.argv) > 3 else 48
P.build()
tot = dict(frame_pairs=0, decision_mismatch_frames=0, decisions=0, pairs=0, area_beyond_1e9=0, skipped=0, overflow=0)
worst = 0.0
for r in range(runs):
    seed = seed0 + r
    scene = ("room", "tumlike", "tunnel")[r % 3]
    stride = (3, 7, 11, 17, 25)[r % 5]
    flags = (0, cape_amd.MATCH_ADVANCED, cape_amd.MATCH_ALLOW_INDEX0, cape_amd.MATCH_ADVANCED | cape_amd.MATCH_ALLOW_INDEX0)[r % 4]
    res, pol, ver, got, T = _run(scene, seed, 20 + 13 * r, stride, n, flags=flags, cyl=(r % 2 == 0))
    kept, skip = [], set()
    for f in range(n):
        planes = []
        for i, s in enumerate(res.segments(f)):
            if not s["is_output"]:
                continue
            c0 = np.asarray(s["normal"], np.float64) * (-np.float64(s["d"]))
            ref = P.Polygon.from_points(res.boundary_points(f, s), s["normal"], c0)
            if ref.flags & P.NEEDS_DISSOLVE:
                skip.add(f)
            elif ref.valid and ref.boundary_length() >= 3:
                planes.append((i, np.asarray(s["out_normal"], np.float64), float(s["d"]), ref))
        kept.append(planes)
    for f in range(1, n):
        if got[f]["flags"] & cape_amd.MATCH_EXACT_OVERFLOW:
            tot["overflow"] += 1
            continue
        if f in skip or (f - 1) in skip:
            tot["skipped"] += 1
            continue
        prev, cur = kept[f - 1], kept[f]
        want, inter = P.find_matches([q[1:] for q in prev], [q[1:] for q in cur], T[f], advanced=bool(flags & cape_amd.MATCH_ADVANCED),
                                     allow_index0=bool(flags & cape_amd.MATCH_ALLOW_INDEX0))
        tot["frame_pairs"] += 1
        tot["decisions"] += sum(1 for m in want if m >= 0)
        if list(got[f]["match"][: len(prev)]) != want:
            tot["decision_mismatch_frames"] += 1
            print(f"MISMATCH run {r} ({scene} seed {seed} stride {stride} flags {flags}) frame {f}: device {list(got[f]['match'][:len(prev)])} oracle {want}", flush=True)
        for j in range(len(prev)):
            for i in range(len(cur)):
                b = float(inter[j, i])
                if b < 0:
                    continue
                a = float(got[f]["inter_area"][j][i])
                tot["pairs"] += 1
                rel = abs(a - b) / max(b, cur[i][3].area)
                worst = max(worst, rel)
                if rel > 1e-9:
                    tot["area_beyond_1e9"] += 1
                    print(f"AREA run {r} ({scene} seed {seed}) frame {f} pair ({j},{i}): device {a} oracle {b}", flush=True)
    print(f"run {r + 1} of {runs}: {tot}", flush=True)
print("worst relative area difference", worst)
print("RESULT", "OK" if tot["decision_mismatch_frames"] == 0 and tot["area_beyond_1e9"] == 0 else "DISAGREEMENTS", tot)
