#!/usr/bin/env python3
"""How much of the oracle's output depends on the third-party choices it could NOT pin against the real reference?

TEST INFRASTRUCTURE (see cape_oracle.hpp).  The reference cannot be built in this image (no Eigen / OpenCV / Boost), so
the oracle restates Eigen's fixed-size reductions, its dynamic GEMM summation order, SelfAdjointEigenSolver, the 3x3
determinant, normalize() and glibc's acos / atan2 from their published algorithms.  Each of those
choices is a compile-time switch of cape_oracle.cpp (CAPE_VAR_*).  This script builds one library per variant under
oracle/_variants/ (git-ignored), runs the same randomised frames (the generator of profiles/fuzz_parity.py) through the
default build and every variant, and reports per variant

    frames whose plane / cylinder label grid changes, cells changed, frames whose plane count changes,
    and over the frames whose labels are unchanged: max |delta normal| (per component), max |delta d| (mm).

usage: variants.py [n_frames=2048] [seed=1] [workers=8] [width=640] [height=480]
       -> prints a markdown table (DESIGN.md section 2 keeps a copy; profiles/r02_oracle_variants.txt, r05_oracle_variants.txt the logs)
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
for p in (os.path.join(ROOT, "rgb-d-slam_amd", "python"), HERE):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np  # noqa: E402

VARIANTS = {
    "dot_order": ("-DCAPE_VAR_DOT_ORDER=1", "3-vector reductions a0+(a1+a2) instead of (a0+a1)+a2 (A.2)"),
    "normalize_recip": ("-DCAPE_VAR_NORMALIZE=1", "normalize() multiplies by 1/sqrt instead of dividing (A.2)"),
    "eigen_jacobi": ("-DCAPE_VAR_EIGEN=1", "another backward-stable eigen-solver (cyclic Jacobi) (A.1)"),
    "eigen_direct": ("-DCAPE_VAR_EIGEN=2", "closed-form computeDirect-style eigen-solver (A.1, pessimistic)"),
    "det_form": ("-DCAPE_VAR_DET=1", "Eigen's bruteforce_det3_helper association (A.2)"),
    "gemm_blocked": ("-DCAPE_VAR_GEMM=1", "cylinder covariance summed in depth blocks of 256"),
    "gemm_descending": ("-DCAPE_VAR_GEMM=2", "cylinder covariance summed in descending k"),
    "gemm_two_lanes": ("-DCAPE_VAR_GEMM=3", "cylinder covariance with two interleaved accumulators"),
    "libm_plus_ulp": ("-DCAPE_VAR_LIBM=1", "acos / atan2 one ulp higher (glibc vs another libm)"),
    "libm_minus_ulp": ("-DCAPE_VAR_LIBM=2", "acos / atan2 one ulp lower"),
    "all_at_once": ("-DCAPE_VAR_DOT_ORDER=1 -DCAPE_VAR_NORMALIZE=1 -DCAPE_VAR_EIGEN=1 -DCAPE_VAR_DET=1 -DCAPE_VAR_GEMM=3 -DCAPE_VAR_LIBM=1",
                    "every switch flipped together (Jacobi solver)"),
}
CXXFLAGS = "-O2 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -w"


def build_variants(names=None):
    out_dir = os.path.join(HERE, "_variants")
    os.makedirs(out_dir, exist_ok=True)
    src = os.path.join(HERE, "cape_oracle.cpp")
    paths = {}
    for name in (names or VARIANTS):
        so = os.path.join(out_dir, f"libcape_oracle_{name}.so")
        if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
            subprocess.check_call(f"g++ {CXXFLAGS} {VARIANTS[name][0]} -shared -o {so} {src}", shell=True)
        paths[name] = so
    return paths


def fuzz_frame(rng, W=640, H=480):
    """One frame of the randomised sweep (same recipe as profiles/fuzz_parity.py)."""
    from cape_amd import synth

    names = ["room", "tumlike", "tunnel", "facets", "facets", "tunnel"]
    intr = {k: v * W / 640.0 for k, v in synth.DEFAULT_INTRINSICS.items()}
    d = synth.SCENES[names[int(rng.integers(0, len(names)))]](seed=int(rng.integers(0, 100000)), frame=int(rng.integers(0, 2000)),
                                                              width=W, height=H, intr=intr)
    mode = int(rng.integers(0, 8))
    if mode == 1:
        d[rng.random(d.shape) < rng.uniform(0.02, 0.3)] = 0
    elif mode == 2:
        d += (rng.standard_normal(d.shape) * rng.uniform(0.5, 8)).astype(np.float32) * (d > 0)
    elif mode == 3:
        y, x = int(rng.integers(0, H - 80)), int(rng.integers(0, W - 80))
        d[y:y + 80, x:x + 80] *= np.float32(rng.uniform(0.3, 0.9))
    elif mode == 4:
        d = np.ascontiguousarray(d[:, ::-1])
    elif mode == 5:
        d = np.ascontiguousarray(d[::-1, :])
    elif mode == 6:
        d *= np.float32(rng.uniform(0.3, 3.0))
    return d, intr


def compare(base, var):
    """Differences between two OracleResults of the same frame."""
    out = dict(label_cells=int((base.plane_labels != var.plane_labels).sum()), cyl_cells=int((base.cyl_labels != var.cyl_labels).sum()),
               count_changed=int(len(base.planes) != len(var.planes) or len(base.cylinders) != len(var.cylinders)),
               bins_changed=int((base.bins != var.bins).sum()), dn=0.0, dd=0.0, daxis=0.0)
    if out["label_cells"] == 0 and out["cyl_cells"] == 0 and not out["count_changed"]:
        if len(base.planes):
            out["dn"] = float(np.abs(base.planes[:, 0:3] - var.planes[:, 0:3]).max())
            out["dd"] = float(np.abs(base.planes[:, 3] - var.planes[:, 3]).max())
        if len(base.cylinders):
            a, b = base.cylinders[:, 0:3], var.cylinders[:, 0:3]
            out["daxis"] = float(np.minimum(np.abs(a - b).max(axis=1), np.abs(a + b).max(axis=1)).max())  # axis sign is free
    return out


def _work(args):
    seed, n, paths, W, H = args
    import cape_oracle_py as O

    rng = np.random.default_rng(seed)
    acc = {name: dict(frames=0, label_frames=0, label_cells=0, cyl_frames=0, count_frames=0, bins_cells=0, dn=0.0, dd=0.0, daxis=0.0)
           for name in paths}
    oracles = {}
    n_planes = 0
    for _ in range(n):
        d, intr = fuzz_frame(rng, W, H)
        key = tuple(sorted(intr.items()))
        if key not in oracles:
            oracles[key] = (O.Oracle(W, H, cylinders=True, **intr),
                            {nm: O.Oracle(W, H, cylinders=True, lib_path=p, **intr) for nm, p in paths.items()})
        base_o, var_o = oracles[key]
        base = base_o.run(d)
        n_planes += len(base.planes)
        for nm, vo in var_o.items():
            c = compare(base, vo.run(d))
            a = acc[nm]
            a["frames"] += 1
            a["label_frames"] += int(c["label_cells"] > 0)
            a["label_cells"] += c["label_cells"]
            a["cyl_frames"] += int(c["cyl_cells"] > 0)
            a["count_frames"] += c["count_changed"]
            a["bins_cells"] += c["bins_changed"]
            a["dn"] = max(a["dn"], c["dn"])
            a["dd"] = max(a["dd"], c["dd"])
            a["daxis"] = max(a["daxis"], c["daxis"])
    return acc, n_planes


def run(n_frames, seed=1, workers=8, names=None, width=640, height=480):
    import multiprocessing as mp

    paths = build_variants(names)
    per = -(-n_frames // workers)
    jobs = [(seed * 1000 + w, per, paths, width, height) for w in range(workers)]
    with mp.get_context("spawn").Pool(workers) as pool:
        parts = pool.map(_work, jobs)
    total = {nm: dict(frames=0, label_frames=0, label_cells=0, cyl_frames=0, count_frames=0, bins_cells=0, dn=0.0, dd=0.0, daxis=0.0) for nm in paths}
    planes = 0
    for acc, npl in parts:
        planes += npl
        for nm, a in acc.items():
            t = total[nm]
            for k in ("frames", "label_frames", "label_cells", "cyl_frames", "count_frames", "bins_cells"):
                t[k] += a[k]
            for k in ("dn", "dd", "daxis"):
                t[k] = max(t[k], a[k])
    return total, planes


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    workers = int(sys.argv[3]) if len(sys.argv) > 3 else 8
    W = int(sys.argv[4]) if len(sys.argv) > 4 else 640
    H = int(sys.argv[5]) if len(sys.argv) > 5 else 480
    total, planes = run(n, seed, workers, width=W, height=H)
    frames = next(iter(total.values()))["frames"]
    print(f"# {frames} randomised {W}x{H} frames (cylinders on), {planes} output planes in the default build, seed {seed}")
    print("| variant | what changes | frames with a plane-label change | cells changed | frames with a cylinder-label change | "
          "frames with another primitive count | histogram bins changed (cells) | max abs delta normal | max abs delta d (mm) | max abs delta cylinder axis |")
    print("|---|---|---|---|---|---|---|---|---|---|")
    for nm, t in total.items():
        print(f"| `{nm}` | {VARIANTS[nm][1]} | {t['label_frames']} | {t['label_cells']} | {t['cyl_frames']} | {t['count_frames']} | "
              f"{t['bins_cells']} | {t['dn']:.3g} | {t['dd']:.3g} | {t['daxis']:.3g} |")


if __name__ == "__main__":
    main()
