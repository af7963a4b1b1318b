#!/usr/bin/env python3
"""Compare a dump of the REAL reference (ref_dump.cpp, see README.md) with the oracle on the same frame.
usage: compare_ref_dump.py out.bin depth.f32 width height fx fy cx cy      TEST INFRASTRUCTURE ONLY."""
import os
import struct
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cape_oracle_py as O  # noqa: E402


def main():
    dump, depth_path = sys.argv[1], sys.argv[2]
    W, H = int(sys.argv[3]), int(sys.argv[4])
    fx, fy, cx, cy = (float(v) for v in sys.argv[5:9])
    raw = open(dump, "rb").read()
    assert raw[:8] == b"CAPEREF1"
    hC, vC, nP, nC = struct.unpack_from("<4i", raw, 8)
    cells = hC * vC
    off = 24
    pl = np.frombuffer(raw, "<i4", cells, off); off += 4 * cells
    cl = np.frombuffer(raw, "<i4", cells, off); off += 4 * cells
    cell_dt = np.dtype([("planar", "u1"), ("n", "<u4"), ("normal", "<f8", 3), ("d", "<f8"), ("mse", "<f8"), ("score", "<f8"), ("tol", "<f4")])
    cs = np.frombuffer(raw, cell_dt, cells, off); off += cell_dt.itemsize * cells
    planes = np.frombuffer(raw, "<f8", nP * 13, off).reshape(nP, 13); off += 8 * 13 * nP
    cyls = np.frombuffer(raw, "<f8", nC * 4, off).reshape(nC, 4)
    depth = np.fromfile(depth_path, np.float32).reshape(H, W)
    r = O.Oracle(W, H, fx, fy, cx, cy, cylinders=True).run(depth)
    rep = []
    rep.append(("plane label grid", "EQUAL" if np.array_equal(pl, r.plane_labels) else f"{int((pl != r.plane_labels).sum())} cells differ"))
    rep.append(("cylinder label grid", "EQUAL" if np.array_equal(cl, r.cyl_labels) else f"{int((cl != r.cyl_labels).sum())} cells differ"))
    rep.append(("cell planar flags", "EQUAL" if np.array_equal(cs["planar"], r.planar) else f"{int((cs['planar'] != r.planar).sum())} differ"))
    rep.append(("cell point counts", "EQUAL" if np.array_equal(cs["n"], r.n) else "differ"))
    both = (cs["planar"] == 1) & (r.planar == 1)
    rep.append(("cell normals max |delta|", f"{np.abs(cs['normal'][both] - r.normal[both]).max() if both.any() else 0:.3g}"))
    rep.append(("cell d max |delta|", f"{np.abs(cs['d'][both] - r.d[both]).max() if both.any() else 0:.3g}"))
    rep.append(("cell tolerances", "EQUAL (bitwise)" if np.array_equal(cs["tol"].view(np.uint32), r.tol.view(np.uint32)) else
                f"max |delta| {np.abs(cs['tol'] - r.tol).max():.3g}"))
    if nP == len(r.planes):
        rep.append(("plane normals max |delta|", f"{np.abs(planes[:, 0:3] - r.planes[:, 0:3]).max() if nP else 0:.3g}"))
        rep.append(("plane d max |delta|", f"{np.abs(planes[:, 3] - r.planes[:, 3]).max() if nP else 0:.3g}"))
    else:
        rep.append(("plane count", f"reference {nP} vs oracle {len(r.planes)}"))
    if nC == len(r.cylinders) and nC:
        a, b = cyls[:, 0:3], r.cylinders[:, 0:3]
        rep.append(("cylinder axis max |delta| (sign free)", f"{np.minimum(np.abs(a - b).max(1), np.abs(a + b).max(1)).max():.3g}"))
    for k, v in rep:
        print(f"{k:42s} {v}")


if __name__ == "__main__":
    main()
