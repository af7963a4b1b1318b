#!/usr/bin/env python3
"""Regenerates tests/golden/*.npz from the CPU oracle (the reference has no golden vectors for this path and cannot be
built here -- SURVEY.md 8c -- so these pin the ORACLE's behaviour over time and across boxes, and give the GPU tests
fixed targets).  Inputs are named by (scene, seed, frame) + SHA-256 of the generated depth; one full TUM-like frame is
also stored as raw uint16 (what a TUM depth PNG holds) so that one fixture does not depend on the generator at all.

usage: python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "rgb-d-slam_amd", "python"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))

import cape_oracle_py as O  # noqa: E402
from cape_amd import synth  # noqa: E402

CASES = [
    # name, scene, seed, frame, width, height, cylinders
    ("tumlike_s1_planeonly", "tumlike", 1, 0, 640, 480, False),   # BASELINE.json configs[0] stand-in
    ("room_s0_f0_planeonly", "room", 0, 0, 640, 480, False),      # configs[1]
    ("room_s3_f17_planeonly", "room", 3, 17, 640, 480, False),
    ("tunnel_s0_f0_cyl", "tunnel", 0, 0, 640, 480, True),         # configs[2]
    ("tunnel_s4_f9_cyl", "tunnel", 4, 9, 640, 480, True),
    ("tumlike_s2_f5_cyl", "tumlike", 2, 5, 640, 480, True),
    ("room_1280_s2_f11_planeonly", "room", 2, 11, 1280, 960, False),  # configs[4] size
]


def intrinsics(scene, width):
    base = synth.TUM_FR1_INTRINSICS if scene == "tumlike" else synth.DEFAULT_INTRINSICS
    return {k: v * (width / 640.0) for k, v in base.items()}


def run_case(scene, seed, frame, width, height, cylinders, depth=None):
    if depth is None:
        depth = synth.SCENES[scene](seed=seed, frame=frame, width=width, height=height)
    orc = O.Oracle(width, height, cylinders=cylinders, **intrinsics(scene, width))
    r = orc.run(depth)
    out = dict(
        sha256=np.array(synth.sha256(depth)), planar=r.planar, point_count=r.n, bins=r.bins,
        tol=r.tol, cell_mse=r.mse, plane_labels=r.plane_labels, cyl_labels=r.cyl_labels, seeds=r.seeds,
        seed_outcome=r.seed_outcome, seed_activated=r.seed_activated, segments=r.segments,
        merge_labels=r.merge_labels, planes=r.planes, cylinders=r.cylinders,
        boundary_counts=np.array([len(b) for b in r.boundary], np.int32),
        boundary=np.concatenate(r.boundary) if r.boundary else np.zeros((0, 3)),
    )
    return depth, out


def main():
    for name, scene, seed, frame, w, h, cyl in CASES:
        depth, out = run_case(scene, seed, frame, w, h, cyl)
        np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
        print(name, "segments", len(out["segments"]), "planes", len(out["planes"]), "cyl", len(out["cylinders"]))
    # one self-contained fixture: raw uint16 depth (TUM convention: 5000 units per metre)
    depth = synth.tumlike(seed=1, frame=0)
    raw = np.rint(depth.astype(np.float64) * 5.0).astype(np.uint16)
    assert np.array_equal(raw.astype(np.float32) * np.float32(0.2), depth)
    np.savez_compressed(os.path.join(HERE, "tumlike_s1_raw_u16.npz"), raw=raw)
    print("raw fixture bytes", os.path.getsize(os.path.join(HERE, "tumlike_s1_raw_u16.npz")))


if __name__ == "__main__":
    main()
