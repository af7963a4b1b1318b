"""The oracle against the committed golden fixtures (tests/golden/, produced by tests/golden/make_golden.py)."""
import os
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))
import make_golden  # noqa: E402


@pytest.mark.parametrize("case", make_golden.CASES, ids=[c[0] for c in make_golden.CASES])
def test_oracle_matches_golden(oracle_mod, case):
    name, scene, seed, frame, w, h, cyl = case
    g = np.load(os.path.join(HERE, "golden", name + ".npz"))
    depth, out = make_golden.run_case(scene, seed, frame, w, h, cyl)
    assert str(g["sha256"]) == str(out["sha256"]), "synthetic generator no longer reproduces the fixture input"
    for k in g.files:
        if k == "sha256":
            continue
        a, b = g[k], out[k]
        assert a.shape == b.shape, k
        if a.dtype.kind == "f":
            assert np.array_equal(np.ascontiguousarray(a).view(np.uint8), np.ascontiguousarray(b).view(np.uint8)), k
        else:
            assert np.array_equal(a, b), k


def test_raw_u16_fixture(oracle_mod):
    """Self-contained fixture: raw uint16 TUM-style depth -> float mm (x 1/5 in f32) -> golden labels."""
    from cape_amd import synth

    raw = np.load(os.path.join(HERE, "golden", "tumlike_s1_raw_u16.npz"))["raw"]
    depth = raw.astype(np.float32) * np.float32(0.2)
    g = np.load(os.path.join(HERE, "golden", "tumlike_s1_planeonly.npz"))
    r = oracle_mod.Oracle(640, 480, cylinders=False, **synth.TUM_FR1_INTRINSICS).run(depth)
    assert np.array_equal(r.plane_labels, g["plane_labels"])
    assert np.array_equal(r.segments.view(np.uint64), g["segments"].view(np.uint64))
