#!/usr/bin/env python3
"""How much of the polygon oracle's output depends on the Boost / FLANN behaviours it could NOT pin against the real libraries?

TEST INFRASTRUCTURE (see polygon_oracle.cpp).  The reference's polygon path leans on FLANN's randomized k-nearest search and on
Boost.Geometry's simplify / is_valid; neither library exists in this image, so the oracle restates their documented behaviour.
Every such choice is a compile-time switch of polygon_oracle.cpp (POLY_VAR_*); this script builds one library per variant under
oracle/_variants/ (git-ignored) and runs the SAME planes -- the boundary candidates the extraction oracle (cape_oracle) finds on
randomised synthetic streams, CPU only -- through the default build and every variant:

    planes whose validity / convex-fallback verdict changes, planes whose area moves by more than 1e-9 (relative) and by how much,
    worst IoU against the default polygon, and -- on consecutive frames of the streams -- MapPlane::find_matches decisions that change.

usage: polygon_variants.py [frames_per_scene=48] [seed=3]   -> a markdown table (profiles/r05_polygon_variants.txt)
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
    "knn_ties_descending": ("-DPOLY_VAR_KNN=1", "equal squared distances in DESCENDING point order (FLANN leaves ties in arbitrary order)"),
    "simplify_line_distance": ("-DPOLY_VAR_SIMPLIFY=1", "Douglas-Peucker measures the distance to the carrier line, not to the segment"),
    "simplify_threshold_ge": ("-DPOLY_VAR_SIMPLIFY=2", "a point AT the threshold distance is kept (>= instead of >)"),
    "simplify_rotated_start": ("-DPOLY_VAR_SIMPLIFY=3", "the ring is opened at the vertex farthest from its first one (newer Boost) before it is simplified"),
    "valid_if_only_touching": ("-DPOLY_VAR_VALID=1", "is_valid accepts a ring that touches itself without crossing"),
}
CXXFLAGS = "-O2 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -w"


def build_variants():
    out_dir = os.path.join(HERE, "_variants")
    os.makedirs(out_dir, exist_ok=True)
    src = os.path.join(HERE, "polygon_oracle.cpp")
    paths = {}
    for name, (flag, _) in VARIANTS.items():
        so = os.path.join(out_dir, f"libpolygon_oracle_{name}.so")
        if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
            subprocess.check_call(f"g++ {CXXFLAGS} {flag} -shared -o {so} {src}", shell=True)
        paths[name] = so
    return paths


def planes_of_stream(scene, n, seed, mode):
    """[(frame, [(normal, d, boundary points)])]: the extraction oracle's output planes on n consecutive frames"""
    import cape_oracle_py as O
    from cape_amd import synth

    intr = synth.TUM_FR1_INTRINSICS if scene == "tumlike" else synth.DEFAULT_INTRINSICS
    orc = O.Oracle(640, 480, cylinders=False, **intr)
    rng = np.random.default_rng(seed)
    frames = synth.stream(scene, seed=seed, n_frames=n, start=11)
    out = []
    for f in range(n):
        d = frames[f].copy()
        if mode == 1:
            d[rng.random(d.shape) < 0.05] = 0
        elif mode == 2:
            d += (rng.standard_normal(d.shape) * 3.0).astype(np.float32) * (d > 0)
        elif mode == 3:
            for _ in range(6):
                y, x = int(rng.integers(0, 400)), int(rng.integers(0, 560))
                d[y:y + 80, x:x + 80] = 0
        r = orc.run(d)
        planes = []
        for k in range(len(r.planes)):
            seg = r.segments[int(r.planes[k][19])]
            planes.append((np.array(r.planes[k][0:3]), float(r.planes[k][3]), np.array(seg[0:3]), float(seg[3]), np.asarray(r.boundary[k])))
        out.append(planes)
    return out


def polygons(P, planes):
    """per frame: [(out normal, d, Polygon or None)] as primitive_detection.cpp:620-631 builds them (segment normal, centre = normal * -d)"""
    res = []
    for frame in planes:
        cur = []
        for (on, od, sn, sd, pts) in frame:
            pol = P.Polygon.from_points(pts, sn, sn * (-sd))
            ok = (not pol.threw) and pol.valid and pol.boundary_length() >= 3
            cur.append((on, od, pol if ok else None, pol))
        res.append(cur)
    return res


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 48
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 3
    import importlib

    import polygon_oracle_py as P

    P.build()
    paths = build_variants()
    streams = []
    for scene in ("room", "tumlike"):
        for mode in range(4):
            streams.append(planes_of_stream(scene, n, seed + 17 * mode, mode))
    base = [polygons(P, s) for s in streams]
    n_planes = sum(len(f) for s in base for f in s)
    n_valid = sum(1 for s in base for f in s for q in f if q[2] is not None)
    base_matches = []
    for s in base:
        for f in range(1, len(s)):
            prev = [(q[0], q[1], q[2]) for q in s[f - 1] if q[2] is not None]
            cur = [(q[0], q[1], q[2]) for q in s[f] if q[2] is not None]
            base_matches.append(P.find_matches(prev, cur)[0])
    print(f"# {len(streams)} streams x {n} frames (room / TUM-like; clean, 5 % dropped pixels, 3 mm noise, dropped blocks), seed {seed}: "
          f"{n_planes} output planes, {n_valid} with a valid polygon in the default build, {len(base_matches)} frame pairs, "
          f"{sum(1 for m in base_matches for x in m if x >= 0)} matches")
    print("| variant | what changes | planes whose validity changes | planes whose convex-fallback verdict changes | planes with another vertex list | "
          "planes whose area moves > 1e-9 rel | worst area ratio - 1 | worst IoU vs default | frame pairs with another match decision |")
    print("|---|---|---|---|---|---|---|---|---|")
    for name, path in paths.items():
        V = importlib.reload(importlib.import_module("polygon_oracle_py"))
        V._LIB_PATH = path
        V._lib = None
        V.build = lambda force=False: path
        var = [polygons(V, s) for s in streams]
        validity = fallback = verts = area_moved = 0
        worst_area, worst_iou = 0.0, 1.0
        for sb, sv in zip(base, var):
            for fb, fv in zip(sb, sv):
                for qb, qv in zip(fb, fv):
                    validity += int((qb[2] is None) != (qv[2] is None))
                    fallback += int(bool(qb[3].flags & P.CONVEX_FALLBACK) != bool(qv[3].flags & P.CONVEX_FALLBACK))
                    if qb[2] is None or qv[2] is None:
                        continue
                    same = len(qb[2].ring) == len(qv[2].ring) and np.array_equal(qb[2].ring, qv[2].ring)
                    verts += int(not same)
                    if not same:
                        rel = abs(qv[2].area / qb[2].area - 1.0)
                        area_moved += int(rel > 1e-9)
                        worst_area = max(worst_area, rel)
                        inter = qb[2].inter_area(qv[2])
                        worst_iou = min(worst_iou, inter / (qb[2].area + qv[2].area - inter))
        decisions = 0
        k = 0
        for s in var:
            for f in range(1, len(s)):
                prev = [(q[0], q[1], q[2]) for q in s[f - 1] if q[2] is not None]
                cur = [(q[0], q[1], q[2]) for q in s[f] if q[2] is not None]
                decisions += int(V.find_matches(prev, cur)[0] != base_matches[k])
                k += 1
        print(f"| `{name}` | {VARIANTS[name][1]} | {validity} | {fallback} | {verts} | {area_moved} | {worst_area:.3g} | {worst_iou:.4f} | {decisions} |")


if __name__ == "__main__":
    main()
