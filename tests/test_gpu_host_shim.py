"""The C++ host-side mirror of the reference interface (rgb-d-slam_amd/host) driven like src/rgbd_slam.cpp drives
the reference's primitives library; output compared with the CPU oracle (cylinder branch on, as in the reference)."""
import os
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("scene,seed,frame", [("tumlike", 1, 0), ("tunnel", 0, 0)])
def test_shim_matches_oracle(oracle_mod, host_binaries, tmp_path, scene, seed, frame):
    from cape_amd import synth

    exe = os.path.join(ROOT, "rgb-d-slam_amd", "lib", "test_shim.exe")
    assert os.path.exists(exe), "build with make -C rgb-d-slam_amd/csrc host"
    intr = synth.TUM_FR1_INTRINSICS if scene == "tumlike" else synth.DEFAULT_INTRINSICS
    depth = synth.SCENES[scene](seed=seed, frame=frame)
    path = tmp_path / "depth.f32"
    depth.tofile(path)
    out = subprocess.run([exe, str(path), "640", "480", str(intr["fx"]), str(intr["fy"]), str(intr["cx"]), str(intr["cy"])],
                         capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr
    lines = out.stdout.strip().splitlines()
    r = oracle_mod.Oracle(640, 480, cylinders=True, **intr).run(depth)
    assert lines[0] == f"planes {len(r.planes)} cylinders {len(r.cylinders)}"
    P = [ln.split()[1:] for ln in lines if ln.startswith("P ")]
    for k, p in enumerate(P):
        vals = np.array([float.fromhex(v) for v in p[:4]])
        assert np.array_equal(vals.view(np.uint64), np.ascontiguousarray(r.planes[k, 0:4]).view(np.uint64))
        assert int(p[4]) == len(r.boundary[k])
        # N1: the host-side boundary polygon is a valid ring spanning the candidate points (area in mm^2)
        assert 3 <= int(p[5]) <= len(r.boundary[k]) and float(p[6]) > 1e4
    Cc = [ln.split()[1:] for ln in lines if ln.startswith("C ")]
    for k, c in enumerate(Cc):
        vals = np.array([float.fromhex(v) for v in c])
        assert np.array_equal(vals.view(np.uint64), np.ascontiguousarray(r.cylinders[k, 0:3]).view(np.uint64))
    # N2: every plane matches itself, except index 0 which the reference's `selectedIndex <= 0` test can never return
    M = {int(ln.split()[1]): int(ln.split()[2]) for ln in lines if ln.startswith("M ")}
    assert len(M) == len(P) and all(M[i] == (i if i > 0 else -1) for i in M)
    # N2 device part on a duplicated frame: same verdicts from the cell masks
    D = [int(v) for v in [ln for ln in lines if ln.startswith("D ")][0].split()[1:]]
    assert D[0] == D[1] == len(r.planes) and D[2:] == [(i if i > 0 else -1) for i in range(len(r.planes))]
    # rectify_depth through the mirror class == oracle rectify (identity transform), and its frame still yields planes
    R = [ln.split()[1:] for ln in lines if ln.startswith("R ")][0]
    ref_rect = oracle_mod.Oracle(640, 480, cylinders=True, **intr).rectify(depth, np.eye(4))
    assert int(R[0]) == int((ref_rect > 0).sum())
    assert int(R[1]) == len(oracle_mod.Oracle(640, 480, cylinders=True, **intr).run(ref_rect).planes)
    assert "Mean primitive extraction time" in out.stderr
