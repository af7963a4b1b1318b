#!/usr/bin/env python3
"""Regenerates tests/golden/polygon_near_ties.npz: boundary-candidate sets whose k-nearest-neighbour walk meets a near-tie that the
product used to decide differently from the polygon oracle (round 6 sweeps, profiles/r06_polygon_cpu_sweep.txt):
  * sets 0-3 and the first device dump: two neighbours at squared distances within 2^-42 of each other -- a selection on the
    54-bit-distance + index key (the device's fast key; through round 5 also the host class's order) against (distance, index);
  * sets 4-5 and the second device dump: two directions less than 4e-16 rad apart (a neighbour on the line of the previous edge,
    two neighbours on one ray): the reference's atan2 angles + DBL_EPSILON slack call them equal, an exact cross product does not.
The numbered sets come out of the CPU extraction oracle on perturbed synthetic frames; the dumps were met on an MI355X
(profiles/polygon_vs_oracle.py 1024 31 / 32 with POLY_DUMP: device-rendered frames, so their candidates are kept as data in
polygon_near_tie_device_dump*.npz).  Expected outputs: the polygon oracle's ring, area, flags and k for every set.

usage: python tests/golden/make_polygon_near_ties.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "rgb-d-slam_amd", "python"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))

import cape_oracle_py as O  # noqa: E402
import polygon_oracle_py as P  # noqa: E402
from cape_amd import synth  # noqa: E402

# (scene, first seed of the run, frame number in the run, output plane): frame k of a run is scene(seed = first + k // 16,
# frame = 7 k mod 900), damaged by mode k mod 4 with ONE generator per run (so the draws of the frames before it are replayed)
CASES = [("room", 17, 343, 1), ("room", 2017, 329, 0), ("tumlike", 6017, 45, 3), ("facets", 6017, 36, 1),
         ("facets", 51017, 681, 2), ("facets", 52017, 1383, 4)]


def frame_of(scene, seed0, k_wanted):
    rng = np.random.default_rng(seed0)
    shape = (480, 640)
    for k in range(k_wanted + 1):
        mode = k % 4
        d = synth.SCENES[scene](seed=seed0 + k // 16, frame=(k * 7) % 900) if k == k_wanted else None
        if mode == 1:
            drop = rng.random(shape) < 0.05
            if d is not None:
                d[drop] = 0
        elif mode == 2:
            noise = (rng.standard_normal(shape) * 3.0).astype(np.float32)
            if d is not None:
                d += noise * (d > 0)
        elif mode == 3:
            for _ in range(6):
                y, x = int(rng.integers(0, 400)), int(rng.integers(0, 560))
                if d is not None:
                    d[y:y + 80, x:x + 80] = 0
    return d


def main():
    P.build()
    out = {}
    sets = []
    for scene, seed0, k, i in CASES:
        intr = synth.TUM_FR1_INTRINSICS if scene == "tumlike" else synth.DEFAULT_INTRINSICS
        r = O.Oracle(640, 480, cylinders=False, **intr).run(frame_of(scene, seed0, k))
        nrm = r.planes[i, 0:3].copy()
        sets.append((f"{scene}_{seed0}_{k}_{i}", r.boundary[i].copy(), nrm, nrm * (-r.planes[i, 3])))
    for name, path in (("device_tunnel_31_768_30_0", "polygon_near_tie_device_dump.npz"), ("device_room_32_768_24_0", "polygon_near_tie_device_dump2.npz")):
        dump = np.load(os.path.join(HERE, path))
        sets.append((name, dump["pts"], dump["normal"], dump["center"]))
    out["names"] = np.array([s[0] for s in sets])
    for j, (name, pts, nrm, ctr) in enumerate(sets):
        ref = P.Polygon.from_points(pts, nrm, ctr)
        out[f"pts{j}"], out[f"normal{j}"], out[f"center{j}"] = np.asarray(pts, np.float64), np.asarray(nrm, np.float64), np.asarray(ctr, np.float64)
        out[f"ring{j}"], out[f"area{j}"], out[f"flags{j}"], out[f"k{j}"] = ref.ring, ref.area, ref.flags, ref.k_used
        print(name, len(pts), "candidates -> ring of", len(ref.ring), "k", ref.k_used, "flags", ref.flags)
    np.savez_compressed(os.path.join(HERE, "polygon_near_ties.npz"), **out)


if __name__ == "__main__":
    main()
