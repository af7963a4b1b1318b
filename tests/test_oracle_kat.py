"""Known-answer tests that pin what CAN be pinned of the CPU oracle without the reference binary:
closed-form values of the formulas on the path, libstdc++ RNG values, an independent eigen-solver, analytic planes,
and the reference's own ScreenToCameraToScreen round-trip test (tests/test_coordinate_systems.cpp:161-209)."""
import numpy as np
import pytest


def test_depth_quantization_values(oracle_mod):
    # covariances.cpp:12-19 with parameters.hpp:16-18 ; values quoted in SURVEY.md 8(c)
    q = oracle_mod.depth_quantization
    assert q(500.0) == pytest.approx(0.5225, abs=1e-12)
    assert q(1000.0) == pytest.approx(2.94, abs=1e-12)
    assert q(2000.0) == pytest.approx(11.87, abs=1e-12)
    assert q(4000.0) == pytest.approx(46.11, abs=1e-12)
    assert q(0.0) == 0.5 and q(100.0) == 0.5  # floor at 0.5 mm
    # exact operation order ((a + b*z) + c*(z*z)) with the constants as the compiler folds them
    z = 1234.5
    b, c = 0.74 / 1000.0, 2.73 * ((1.0 / 1000.0) * (1.0 / 1000.0))
    assert q(z) == (-0.53 + b * z) + c * (z * z)


def test_thresholds_and_constants(oracle_mod):
    L = oracle_mod.lib()
    assert L.cape_oracle_ransac_max_iterations() == 43                 # cylinder_segment.cpp:132
    assert L.cape_oracle_cos_merge_angle() == 0.95105651629515353       # plane_segment.cpp:324
    assert abs(L.cape_oracle_sin_merge_angle() - 0.309017003) < 1e-8    # primitive_detection.cpp:189-190
    assert int(0.008 * 768) == 6 and int(0.0065 * 768) == 4             # seed / activation thresholds (640x480)
    assert int(0.008 * 3072) == 24 and int(0.0065 * 3072) == 19         # 1280x960
    assert int(np.floor(np.float32(400) * np.float32(0.7))) == 280      # _minZeroPointCount


def test_mt19937_known_answers(oracle_mod):
    # libstdc++ mt19937(0) + uniform_real_distribution<double>(0,1) (random.hpp:17-30)
    assert oracle_mod.mt19937_double(0, 0) == 0.59284461651668263
    assert oracle_mod.mt19937_double(0, 1) == 0.84426574425659828
    assert oracle_mod.mt19937_double(0, 2) == 0.85794561998982988
    # generate_canonical<double,53>: (lo + hi * 2^32) / 2^64 from the known raw words 2357136044, 2546248239
    assert oracle_mod.mt19937_double(0, 0) == (2357136044 + 2546248239 * 2.0 ** 32) / 2.0 ** 64


def test_eigen3_against_numpy(oracle_mod):
    rng = np.random.default_rng(0)
    worst_val, worst_res, max_it = 0.0, 0.0, 0
    for i in range(3000):
        p = rng.standard_normal((400, 3)) * rng.uniform(0.01, 100, 3)
        c = p.T @ p
        ev, vec, it = oracle_mod.eigen3(c)
        w = np.linalg.eigvalsh(c)
        worst_val = max(worst_val, np.abs(ev - w).max() / np.abs(w).max())
        worst_res = max(worst_res, np.abs(c @ vec - vec * ev).max() / np.abs(w).max())
        max_it = max(max_it, it)
        assert np.all(np.diff(ev) >= 0)
    assert worst_val < 1e-14 and worst_res < 1e-14 and max_it <= 12


def test_eigen3_special_cases(oracle_mod):
    ev, vec, it = oracle_mod.eigen3(np.diag([3.0, 1.0, 2.0]))
    assert np.array_equal(ev, [1.0, 2.0, 3.0]) and it == 0
    assert np.array_equal(np.abs(vec), np.array([[0, 0, 1], [1, 0, 0], [0, 1, 0]], float))
    ev, vec, it = oracle_mod.eigen3(np.zeros((3, 3)))
    assert np.array_equal(ev, [0, 0, 0]) and np.array_equal(vec, np.eye(3))


def test_back_projection_round_trip(oracle_mod):
    """Reference test PointCoordinateSystemTests.ScreenToCameraToScreen: screen -> camera -> screen at 1e-3."""
    orc = oracle_mod.Oracle(640, 480)  # default intrinsics, parameters.cpp:59-74
    for u in range(0, 640, 40):
        for v in range(0, 480, 40):
            for z in (1.0, 500.0, 4000.0):
                p = orc.back_project(u, v, z)
                assert p[2] == z
                assert abs(550.0 * p[0] / p[2] + 320.0 - u) < 1e-3
                assert abs(550.0 * p[1] / p[2] + 240.0 - v) < 1e-3
    # K^-1 structure: x = z * fl(fl(k00*u) + k02)
    invdet = 1.0 / (550.0 * 550.0)
    k00, k02 = 550.0 * invdet, -(320.0 * 550.0) * invdet
    assert orc.back_project(17, 5, 1234.0)[0] == 1234.0 * (k00 * 17.0 + k02)


def _plane_depth(W, H, fx, fy, cx, cy, n, d):
    """depth of the plane n.p + d = 0 along each pixel ray (mm)."""
    u = (np.arange(W) - cx) / fx
    v = (np.arange(H) - cy) / fy
    X, Y = np.meshgrid(u, v)
    return (-d / (n[0] * X + n[1] * Y + n[2])).astype(np.float32)


def test_analytic_tilted_plane(oracle_mod):
    n = np.array([0.3, -0.2, -1.0])
    n /= np.linalg.norm(n)
    d = 2000.0
    depth = _plane_depth(640, 480, 550, 550, 320, 240, n, d)
    r = oracle_mod.Oracle(640, 480, cylinders=False).run(depth)
    assert r.planar.all() and (r.n == 400).all()
    # every cell recovers the plane: normal faces the camera (d > 0), float32 depth rounding only
    assert np.abs(r.normal - n).max() < 2e-4 and np.abs(r.d - d).max() < 0.5
    assert len(r.segments) == 1 and len(r.planes) == 1 and (r.plane_labels == 1).all()
    assert np.abs(r.planes[0, 0:3] - n).max() < 1e-5 and abs(r.planes[0, 3] - d) < 0.05
    assert abs(np.linalg.norm(r.planes[0, 0:3]) - 1) < 1e-15


def test_fronto_parallel_noise_free_plane(oracle_mod):
    """Degenerate scatter (all z equal): zz = Szs - Sz^2/n is 0 or rounding noise, so a cell is either rejected by the
    det ~ 0 test (plane_segment.cpp:245-248) or fitted with a ~0 MSE and the optical axis as normal."""
    r = oracle_mod.Oracle(640, 480, cylinders=False).run(np.full((480, 640), 1500.0, np.float32))
    ok = r.planar.astype(bool)
    assert np.abs(r.normal[ok] - np.array([0, 0, -1.0])).max() < 1e-4  # x, y carry f32 rounding of z*a
    assert (r.mse[ok] < 1e-6).all() and (r.mse[~ok] == np.finfo(np.float64).max).all()


def test_cell_rejection_rules(oracle_mod):
    n = np.array([0.1, 0.15, -1.0])
    n /= np.linalg.norm(n)
    depth = _plane_depth(640, 480, 550, 550, 320, 240, n, 1500.0)
    orc = oracle_mod.Oracle(640, 480, cylinders=False)
    base = orc.run(depth)
    assert base.planar.all()
    d = depth.copy()
    d[0:20, 0:20][10, :2] = 0      # cell 0: both start pixels of the centre row invalid -> rejected, sums cleared
    d[0:20, 20:40][:, 10][[0, 1]] = 0  # cell 1: both start pixels of the centre column invalid
    c2 = d[0:20, 40:60]
    c2[:, 14:] = 0                 # cell 2: 280 valid pixels: kept (>= 280); invalid pixels are skipped by the scans
    c3 = d[0:20, 60:80]
    c3[:, 14:] = 0
    c3[6, 0] = 0                   # cell 3: 279 valid pixels: sums kept but not planar
    d[0:20, 80:100][10, 5] += 200  # cell 4: depth jump on the centre row
    c5 = d[0:20, 100:120]
    c5[:, :] = 0
    c5[8:13, :] = depth[8:13, 100:120]
    c5[:, 8:13] = depth[0:20, 108:113]  # cell 5: continuous cross but < 200 valid pixels
    r = orc.run(d)
    assert not r.planar[0] and r.n[0] == 0 and not r.sums[0].any()
    assert not r.planar[1] and r.n[1] == 0
    assert r.planar[2] and r.n[2] == 280
    assert not r.planar[3] and r.n[3] == 279 and r.sums[3].all() and r.mse[3] == np.finfo(np.float64).max
    assert not r.planar[4] and r.n[4] == 0
    assert not r.planar[5] and r.n[5] == 0
    assert r.planar[6:].all()


def test_sums_follow_float_product_rule(oracle_mod):
    """x*x, x*y ... are float32 products widened to double (types.hpp:84 SQR on the operand's own type)."""
    depth = _plane_depth(640, 480, 550, 550, 320, 240, np.array([0.2, 0.1, -0.97]) / np.linalg.norm([0.2, 0.1, -0.97]), 1800.0)
    orc = oracle_mod.Oracle(640, 480, cylinders=False)
    r = orc.run(depth)
    cloud = orc.cloud()
    x, y, z = (cloud[k, :400] for k in range(3))  # cell 0 is rows 0..399 of each block
    S = [x.astype(np.float64).sum(), y.astype(np.float64).sum(), z.astype(np.float64).sum(),
         (x * x).astype(np.float64).sum(), (y * y).astype(np.float64).sum(), (z * z).astype(np.float64).sum(),
         (x * y).astype(np.float64).sum(), (y * z).astype(np.float64).sum(), (x * z).astype(np.float64).sum()]
    assert np.array_equal(r.sums[0], np.array(S))  # exact: the addends span < 2^20, any order gives the same f64


def test_histogram_quirk_and_seed_stream(oracle_mod):
    """remove_point() re-bins to 1 (histogram.hpp:112): seeds may be re-picked from bin 1 and burn iterations."""
    from cape_amd import synth

    r = oracle_mod.Oracle(640, 480, cylinders=False, **synth.TUM_FR1_INTRINSICS).run(synth.tumlike(seed=1, frame=0))
    assert len(r.seeds) >= len(r.segments)
    assert set(np.unique(r.seed_outcome)) <= {0, 1, 3, 4}
    # every labelled cell is a planar cell, labels are 1..P, merge labels point at roots
    assert r.planar[r.plane_labels > 0].all()
    assert set(np.unique(r.plane_labels)) - {0} == set(range(1, len(r.segments) + 1))
    assert all(r.merge_labels[m] == m for m in r.merge_labels)


def test_plane_only_mode_is_a_subset(oracle_mod):
    from cape_amd import synth

    d = synth.tunnel(seed=0, frame=0)
    a = oracle_mod.Oracle(640, 480, cylinders=True).run(d)
    b = oracle_mod.Oracle(640, 480, cylinders=False).run(d)
    assert a.cyl_labels.any() and not b.cyl_labels.any()
    assert np.array_equal(a.seeds, b.seeds)  # the seed sequence does not depend on the cylinder branch
    assert len(a.cylinders) >= 1 and np.isnan(a.cylinders[:, 3]).all()  # radius NaN quirk
    assert abs(np.linalg.norm(a.cylinders[0, :3]) - 1) < 1e-12
    assert abs(a.cylinders[0, 2]) > 0.99  # tunnel axis ~ optical axis


def test_rectify_depth_identity_and_shift(oracle_mod):
    """Depth_Map_Transformation::rectify_depth (N3).  With the identity transform a pixel re-projects to u +- 1e-5 (the
    pre-factors are float32), so floor() sends it to its own or to the previous column/row; row/col 0 are dropped by the
    `> 0` test (depth_map_transformation.cpp:62-63).  A pure +x translation moves columns by fx*tx/z."""
    orc = oracle_mod.Oracle(640, 480)
    flat = np.full((480, 640), 2000.0, np.float32)
    r = orc.rectify(flat, np.eye(4))
    assert (r[0, :] == 0).all() and (r[:, 0] == 0).all()
    # targets are hit by their own or their right/lower neighbour; rows/columns whose re-projection rounds just below
    # the integer on both sides stay empty -- a property of the reference algorithm, reproduced as is
    assert set(np.unique(r)) == {0.0, 2000.0} and (r > 0).mean() > 0.5
    ramp = np.tile(np.arange(640, dtype=np.float32) + 1000.0, (480, 1))
    rr = orc.rectify(ramp, np.eye(4))
    hit = rr[5:-5, 5:-5] > 0
    d = (rr[5:-5, 5:-5] - ramp[5:-5, 5:-5])[hit]
    assert set(np.unique(d)) <= {0.0, 1.0}  # value of column u or u+1
    T = np.eye(4)
    T[0, 3] = 100.0  # 100 mm to the right: du = 550 * 100 / z
    r2 = orc.rectify(flat, T)
    shift = int(np.floor(550.0 * 100.0 / 2000.0))
    cols = np.flatnonzero((r2 > 0).any(axis=0))
    assert abs(int(cols.min()) - shift) <= 1 and set(np.unique(r2)) == {0.0, 2000.0}


def test_random_seed_restarts_every_frame_and_selects_the_sequence(oracle_mod):
    """utils::Random::_seed (random.hpp:59-64): 0 under MAKE_DETERMINISTIC, time(0) of the process otherwise; the engine is
    thread_local and find_primitives runs on a fresh thread per frame, so every frame restarts at the seed.  The oracle with
    another seed: still reproducible frame after frame, and the RANSAC draws really come from that seed (known answers of
    libstdc++'s mt19937 + uniform_real_distribution)."""
    from cape_amd import synth

    assert abs(oracle_mod.mt19937_double(0, 0) - 0.5928446165166826) < 1e-15  # (the KAT above pins the whole sequence)
    assert oracle_mod.mt19937_double(12345, 0) != oracle_mod.mt19937_double(0, 0)
    d = synth.tunnel(seed=3, frame=40)
    intr = dict(synth.DEFAULT_INTRINSICS)
    orc = oracle_mod.Oracle(640, 480, cylinders=True, **intr)
    a0 = orc.run(d)
    orc.set_rng_seed(20250321)
    b0, b1 = orc.run(d), orc.run(d)
    assert np.array_equal(b0.cyl_labels, b1.cyl_labels) and np.array_equal(b0.plane_labels, b1.plane_labels)  # restart per frame
    orc.set_rng_seed(0)
    a1 = orc.run(d)
    assert np.array_equal(a0.cyl_labels, a1.cyl_labels) and np.array_equal(a0.cylinders, a1.cylinders, equal_nan=True)
