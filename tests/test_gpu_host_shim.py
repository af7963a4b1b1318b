"""The C++ host-side replacement of the reference's `primitives` library (rgb-d-slam_amd/host/overlay) driven like
src/rgbd_slam.cpp drives the original; output compared with the CPU oracle (cylinder branch on, as in the reference)."""
import os
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bits(v):
    return np.ascontiguousarray(v, dtype=np.float64).view(np.uint64)


def _check_frame(lines, tag, r, min_area=1e3):
    """lines of one frame (prefix `tag`) against an OracleResult."""
    assert lines[0] == f"{tag}planes {len(r.planes)} cylinders {len(r.cylinders)}"
    P = [ln[len(tag):].split()[1:] for ln in lines if ln.startswith(tag + "P ")]
    assert len(P) == len(r.planes)
    for k, p in enumerate(P):
        vals = np.array([float.fromhex(v) for v in p[:4]])
        # Plane::get_normal / get_d: the host re-normalises like Plane(planeSeg, polygon) does -- same bits as the oracle
        assert np.array_equal(_bits(vals), _bits(r.planes[k, 0:4]))
        # N1: the host-side boundary polygon is a valid ring spanning the candidate points (area in mm^2)
        assert 3 <= int(p[4]) <= len(r.boundary[k]) and float(p[5]) > min_area
        cov = np.array([float.fromhex(p[6]), float.fromhex(p[7])])
        assert np.array_equal(_bits(cov), _bits(r.planes[k, [10, 17]])), "get_point_cloud_covariance (0,0) and (2,1)"
    Cc = [ln[len(tag):].split()[1:] for ln in lines if ln.startswith(tag + "C ")]
    assert len(Cc) == len(r.cylinders)
    for k, c in enumerate(Cc):
        vals = np.array([float.fromhex(v) for v in c[:3]])
        assert np.array_equal(_bits(vals), _bits(r.cylinders[k, 0:3]))
        assert c[3] == "1", "Cylinder::_radius is NaN, as in the reference"


@pytest.mark.parametrize("scene,seed,frame", [("tumlike", 1, 0), ("tunnel", 0, 0)])
def test_shim_matches_oracle(oracle_mod, host_binaries, tmp_path, scene, seed, frame):
    from cape_amd import synth

    exe = os.path.join(host_binaries, "test_shim.exe")
    intr = synth.TUM_FR1_INTRINSICS if scene == "tumlike" else synth.DEFAULT_INTRINSICS
    depth = synth.SCENES[scene](seed=seed, frame=frame)
    path = tmp_path / "depth.f32"
    depth.tofile(path)
    out = subprocess.run([exe, str(path), "640", "480", str(intr["fx"]), str(intr["fy"]), str(intr["cx"]), str(intr["cy"])],
                         capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr
    lines = out.stdout.strip().splitlines()
    orc = oracle_mod.Oracle(640, 480, cylinders=True, **intr)
    r = orc.run(depth)
    head = [ln for ln in lines if ln.startswith(("planes", "P ", "C "))]
    _check_frame(head, "", r)
    n_planes = len(r.planes)
    # N2: every plane matches itself, except index 0 which the reference's `selectedIndex <= 0` test can never return
    M = {int(ln.split()[1]): int(ln.split()[2]) for ln in lines if ln.startswith("M ")}
    assert len(M) == n_planes and all(M[i] == (i if i > 0 else -1) for i in M)
    # N2 device part on a duplicated frame: same verdicts from the cell masks
    D = [int(v) for v in [ln for ln in lines if ln.startswith("D ")][0].split()[1:]]
    assert D[0] == D[1] == n_planes and D[2:] == [(i if i > 0 else -1) for i in range(n_planes)]
    # rectify_depth through the class == oracle rectify (identity transform), and its frame still yields planes
    R = [ln.split()[1:] for ln in lines if ln.startswith("R ")][0]
    ref_rect = orc.rectify(depth, np.eye(4))
    assert int(R[0]) == int((ref_rect > 0).sum())
    assert int(R[1]) == len(orc.run(ref_rect).planes)
    assert "Mean primitive extraction time" in out.stderr


@pytest.mark.parametrize("shards", [1, 3, 0])
def test_sharded_batch_matches_oracle(oracle_mod, host_binaries, tmp_path, shards):
    """find_primitives_batch cuts the batch in contiguous blocks over `shards` handles (one host thread each; shard i on
    device i % device_count, so three shards also run on a one-GPU box; 0 = one per visible device) and returns the
    containers in frame order."""
    from cape_amd import synth

    exe = os.path.join(host_binaries, "test_shim.exe")
    intr = synth.DEFAULT_INTRINSICS
    names = ["room", "tunnel", "facets", "room", "tunnel", "facets", "room"]
    frames = np.stack([synth.SCENES[n](seed=11 + i, frame=2 * i) for i, n in enumerate(names)])
    path = tmp_path / "batch.f32"
    frames.tofile(path)
    out = subprocess.run([exe, str(path), "640", "480", str(intr["fx"]), str(intr["fy"]), str(intr["cx"]), str(intr["cy"]),
                          str(len(frames) - 1), str(shards)], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr
    lines = out.stdout.strip().splitlines()
    S = [ln for ln in lines if ln.startswith("S ")][0].split()
    assert int(S[2]) == len(frames)
    if shards:
        assert int(S[1]) == shards
    orc = oracle_mod.Oracle(640, 480, cylinders=True, **intr)
    for k in range(len(frames)):
        tag = f"B{k} "
        _check_frame([ln for ln in lines if ln.startswith(tag)], tag, orc.run(frames[k]))
        # boundary polygons: built on the device (B lines) == built by the host class (H lines), summary and every vertex
        dev = [ln[len(tag):] for ln in lines if ln.startswith(tag)]
        host = [ln[len(tag):] for ln in lines if ln.startswith(f"H{k} ")]
        assert dev == host and len(dev) > 0
        assert f"V{k} 1" in lines
    # the batch as raw uint16 images (find_primitives_batch(const uint16_t*, scale, ...)) == the quantised depths as float32
    U = [ln.split() for ln in lines if ln.startswith("U ")][0]
    assert U[1] == "1" and int(U[2]) >= len(frames)


def test_polygon_matcher_equals_host_selection(host_binaries, tmp_path):
    """Row N2 through the overlay: Primitive_Detection::match_consecutive_polygons (device, cape_match_polygons) between the
    frames of a moving-camera batch == find_plane_match (host, MapPlane::find_matches on CameraPolygons) run on the
    containers the same batch returned, previous plane by previous plane, matched flags carried along."""
    from cape_amd import synth

    exe = os.path.join(host_binaries, "test_shim.exe")
    intr = synth.TUM_FR1_INTRINSICS
    frames = np.stack([synth.SCENES["tumlike"](seed=21, frame=40 + i) for i in range(10)])
    path = tmp_path / "stream.f32"
    frames.tofile(path)
    out = subprocess.run([exe, str(path), "640", "480", str(intr["fx"]), str(intr["fy"]), str(intr["cx"]), str(intr["cy"]),
                          str(len(frames) - 1), "1"], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr
    X = [int(v) for v in [ln for ln in out.stdout.splitlines() if ln.startswith("X ")][0].split()[1:]]
    n, previous, matched, mismatches = X
    assert n == len(frames) and previous >= 2 * (n - 1) and matched >= n - 1, X
    assert mismatches == 0, X
    # set_batch_matching: the same batch over three shards, the two shard boundaries (and frame 0) stitched by the host class,
    # gives the decisions of one shard matched on the device -- with a camera pose in both (cape_match_polygons_pose)
    Y = [int(v) for v in [ln for ln in out.stdout.splitlines() if ln.startswith("Y ")][0].split()[1:]]
    n, host_one, host_three, matched, mismatches = Y
    assert n == len(frames) and host_one == 1 and host_three == 3 and matched >= n - 1 and mismatches == 0, Y


@pytest.mark.parametrize("size", [(1280, 960, 100), (1920, 1080, 160)])
def test_overlay_follows_record_chains_and_wide_grids(oracle_mod, host_binaries, tmp_path, size):
    """Through the C++ overlay (find_primitives_batch): a frame of more than 64 plane segments -- the checkerboard of facets, a chain
    of records on the device -- between ordinary frames, at 1280 x 960 (fast kernels + the general instance for the one frame) and at
    1920 x 1080 (96 x 54 cells: the general instance for every frame).  Containers == oracle, device polygons == host class for every
    plane incl. those of the spill records."""
    from cape_amd import synth
    from test_gpu_parity import _checkerboard_of_facets

    W, H, tile = size  # (a facet must reach uint(0.0065 cells) cells to become a region: bigger tiles on the bigger grid)
    exe = os.path.join(host_binaries, "test_shim.exe")
    big, intr = _checkerboard_of_facets(W, H, tile=tile)
    frames = np.stack([synth.room(seed=4, frame=1, width=W, height=H, intr=intr), big,
                       synth.tunnel(seed=2, frame=6, width=W, height=H, intr=intr), big])
    path = tmp_path / "batch.f32"
    frames.tofile(path)
    out = subprocess.run([exe, str(path), str(W), str(H), str(intr["fx"]), str(intr["fy"]), str(intr["cx"]), str(intr["cy"]),
                          str(len(frames) - 1), "1"], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = out.stdout.strip().splitlines()
    orc = oracle_mod.Oracle(W, H, cylinders=True, **intr)
    want = [orc.run(f) for f in frames]
    assert len(want[1].segments) > 64
    for k in range(len(frames)):
        tag = f"B{k} "
        # (a facet cut by the image border keeps a few nearly collinear candidates: a sliver of a polygon, no area to speak of)
        _check_frame([ln for ln in lines if ln.startswith(tag)], tag, want[k], min_area=0.0)
        dev = [ln[len(tag):] for ln in lines if ln.startswith(tag)]
        host = [ln[len(tag):] for ln in lines if ln.startswith(f"H{k} ")]
        assert dev == host and len(dev) > 0
        assert f"V{k} 1" in lines
