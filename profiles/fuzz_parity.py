#!/usr/bin/env python3
"""Long randomised GPU-vs-oracle parity sweep (not part of the test suite: minutes, not seconds).
usage: fuzz_parity.py [n_frames=512] [seed=1] [width=640] [height=480]   -- every observable of every frame must be bit-identical.
FUZZ_BATCH=n: frames per call (n <= 8: one-frame handles, i.e. the latency instance of stage A and results in pinned memory);
FUZZ_CELLS=1: the per-cell statistics are compared as well; FUZZ_BIG=1: one frame in eight is a checkerboard of tilted facets with a
random tile size (frames of more than 64 plane segments: record chains through the general grow instance); CAPE_GROW=general in the
environment sends every frame through that instance; FUZZ_U16=1: the frames go in as raw uint16 sensor units (row N4, cape_extract_u16_host,
scale 0.2 like a TUM depth PNG) and the oracle sees float(raw) * 0.2f."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in ("rgb-d-slam_amd/python", "oracle", "tests"):
    sys.path.insert(0, os.path.join(ROOT, p))
import numpy as np
import cape_oracle_py as O
from cape_amd import Extractor, synth
from test_gpu_parity import compare_frame

n_total = int(sys.argv[1]) if len(sys.argv) > 1 else 512
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
W = int(sys.argv[3]) if len(sys.argv) > 3 else 640
H = int(sys.argv[4]) if len(sys.argv) > 4 else 480
names = ["room", "tumlike", "tunnel", "facets", "facets", "tunnel"]
intr = {k: v * W / 640.0 for k, v in synth.DEFAULT_INTRINSICS.items()}
B = int(os.environ.get("FUZZ_BATCH", 64 if W * H <= 640 * 480 else 16))
check_cells = os.environ.get("FUZZ_CELLS") == "1"
big = os.environ.get("FUZZ_BIG") == "1"
as_u16 = os.environ.get("FUZZ_U16") == "1"


def checkerboard(tile, seed):
    """tests/test_gpu_parity.py::_checkerboard_of_facets with the frame's own intrinsics"""
    u = (np.arange(W) - intr["cx"]) / intr["fx"]
    v = (np.arange(H) - intr["cy"]) / intr["fy"]
    X, Y = np.meshgrid(u, v)
    r = np.random.default_rng(seed)
    tilts = [(0.5, 0.0), (-0.5, 0.0), (0.0, 0.5), (0.0, -0.5)]
    z = np.zeros((H, W))
    for ty in range(0, H, tile):
        for tx in range(0, W, tile):
            nx, ny = tilts[((tx // tile) % 2) + 2 * ((ty // tile) % 2)]
            d = 2000.0 + 120.0 * (((tx // tile) * 7 + (ty // tile) * 13) % 9)
            sl = (slice(ty, min(ty + tile, H)), slice(tx, min(tx + tile, W)))
            z[sl] = d / (1.0 + nx * X[sl] + ny * Y[sl])
    z += r.normal(0, 0.6, z.shape)
    return np.round(z).astype(np.float32)
bad = 0
stats = {"cyl_labels": 0, "planes": 0, "merged": 0, "cyl_frames": 0}
orc = {c: O.Oracle(W, H, cylinders=c, **intr) for c in (False, True)}
ex = {c: Extractor(W, H, cylinders=c, max_batch=B, **intr) for c in (False, True)}
done = 0
while done < n_total:
    frames = []
    for k in range(B):
        if big and int(rng.integers(0, 8)) == 0:
            d = checkerboard(20 * int(rng.integers(3, 9)), int(rng.integers(0, 100000)))
        else:
            d = synth.SCENES[names[int(rng.integers(0, len(names)))]](seed=int(rng.integers(0, 100000)), frame=int(rng.integers(0, 2000)), width=W, height=H, intr=intr)
        mode = int(rng.integers(0, 8))
        if mode == 1:
            d[rng.random(d.shape) < rng.uniform(0.02, 0.3)] = 0
        elif mode == 2:
            d += (rng.standard_normal(d.shape) * rng.uniform(0.5, 8)).astype(np.float32) * (d > 0)
        elif mode == 3:
            bh, bw = min(80, H // 2), min(80, W // 2)
            y, x = int(rng.integers(0, H - bh)), int(rng.integers(0, W - bw))
            d[y:y + bh, x:x + bw] *= np.float32(rng.uniform(0.3, 0.9))
        elif mode == 4:
            d = np.ascontiguousarray(d[:, ::-1])
        elif mode == 5:
            d = np.ascontiguousarray(d[::-1, :])
        elif mode == 6:
            d *= np.float32(rng.uniform(0.3, 3.0))
        frames.append(d)
    frames = np.stack(frames)
    if as_u16:
        raw = np.clip(np.nan_to_num(np.round(frames * 5.0), nan=0.0, posinf=65535.0, neginf=0.0), 0, 65535).astype(np.uint16)
        frames = raw.astype(np.float32) * np.float32(0.2)  # what cv::Mat::convertTo(CV_32F, 0.2) leaves (examples/main_TUM.cpp:242)
    for cyl in (True, False):
        n = ex[cyl].extract_host_u16(raw, 0.2) if as_u16 else ex[cyl].extract_host(frames)
        res = ex[cyl].results(n)
        for k in range(n):
            r = orc[cyl].run(frames[k])
            if int(res.records["header"]["status"][k]) & 0x7:
                # the pool of spill records (or the boundary slab) ran out: counted -- and a mismatch below, since round 6 removed
                # the fixed per-frame capacities
                stats["capacity_flagged"] = stats.get("capacity_flagged", 0) + 1
            stats["chained"] = stats.get("chained", 0) + int(int(res.records["header"]["next_record"][k]) >= 0)
            stats["most_segments"] = max(stats.get("most_segments", 0), len(r.segments))
            try:
                compare_frame(r, ex[cyl], res, k, check_cells=check_cells)
            except AssertionError as e:
                bad += 1
                print(f"MISMATCH batch@{done} frame {k} cylinders={cyl}: {str(e)[:200]}", flush=True)
                if os.environ.get("FUZZ_DUMP"):  # the offending frame, for a stand-alone reproduction
                    np.save(os.path.join(os.environ["FUZZ_DUMP"], f"fuzz_fail_{W}x{H}_{done}_{k}_{int(cyl)}.npy"), frames[k])
            if cyl:
                stats["cyl_labels"] += int(r.cyl_labels.max() > 0)
                stats["planes"] += len(r.planes)
                stats["merged"] += int((r.merge_labels != np.arange(len(r.merge_labels))).sum())
                stats["cyl_frames"] += int((r.seed_outcome == 2).any())
    done += B
    stats["general_frames"] = stats.get("general_frames", 0) + ex[True].spill_info()[2] + ex[False].spill_info()[2]
    print(f"{done} frames x 2 modes checked, mismatches so far {bad}", flush=True)
print("coverage:", stats)
print("RESULT", "OK" if bad == 0 else f"{bad} MISMATCHES")
sys.exit(1 if bad else 0)
