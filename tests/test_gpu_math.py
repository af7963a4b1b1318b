"""Device scalar math vs the CPU oracle / host libm, bit for bit (through cape_debug_eval of the C ABI)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _bits(a):
    return np.ascontiguousarray(a, np.float64).view(np.uint64)


def test_f64_sqrt_div_correctly_rounded():
    from cape_amd import debug_eval

    rng = np.random.default_rng(1)
    a = np.abs(rng.standard_normal(200000)) * 10.0 ** rng.integers(-8, 9, 200000)
    b = rng.standard_normal(200000) * 10.0 ** rng.integers(-8, 9, 200000)
    assert np.array_equal(_bits(debug_eval("sqrt", a)), _bits(np.sqrt(a)))
    assert np.array_equal(_bits(debug_eval("div", a, b)), _bits(a / b))
    f = (np.abs(rng.standard_normal(100000)) * 1e4).astype(np.float32)
    assert np.array_equal(debug_eval("sqrtf", f.astype(np.float64)).astype(np.float32).view(np.uint32),
                          np.sqrt(f).view(np.uint32))


def test_quantization_bitwise(oracle_mod):
    from cape_amd import debug_eval

    z = np.concatenate([np.linspace(0, 10000, 5001), [500.0, 1000.0, 2000.0, 4000.0]])
    ref = np.array([oracle_mod.depth_quantization(v) for v in z])
    assert np.array_equal(_bits(debug_eval("quant", z)), _bits(ref))


def test_acos_atan2_bins_match_host_libm():
    """ocml vs glibc may differ in the last ulp; what must agree is the histogram bin (histogram.hpp:48-54)."""
    from cape_amd import debug_eval

    rng = np.random.default_rng(2)
    n = rng.standard_normal((300000, 3))
    n /= np.linalg.norm(n, axis=1)[:, None]
    th_d = debug_eval("acos", -n[:, 2])
    ph_d = debug_eval("atan2", n[:, 0], n[:, 1])
    th_h, ph_h = np.arccos(-n[:, 2]), np.arctan2(n[:, 0], n[:, 1])
    ulp = np.abs(_bits(th_d).astype(np.int64) - _bits(th_h).astype(np.int64))
    assert ulp.max() <= 2
    xq_d, xq_h = np.floor(19 * th_d / np.pi), np.floor(19 * th_h / np.pi)
    yq_d, yq_h = np.floor(19 * (ph_d + np.pi) / (2 * np.pi)), np.floor(19 * (ph_h + np.pi) / (2 * np.pi))
    assert np.array_equal(xq_d, xq_h) and np.array_equal(yq_d, yq_h)
    # exact special values
    sp = debug_eval("acos", np.array([1.0, -1.0, 0.0]))
    assert np.array_equal(_bits(sp), _bits(np.arccos(np.array([1.0, -1.0, 0.0]))))


def test_eigen3_bitwise_vs_oracle(oracle_mod):
    from cape_amd import debug_eval

    rng = np.random.default_rng(3)
    N = 4000
    mats = np.zeros((N, 6))
    refs = np.zeros((N, 12))
    for i in range(N):
        p = rng.standard_normal((400, 3)) * rng.uniform(0.01, 100, 3)
        if i % 7 == 0:
            p[:, 2] = 0.0  # degenerate direction
        c = p.T @ p
        mats[i] = [c[0, 0], c[1, 0], c[1, 1], c[2, 0], c[2, 1], c[2, 2]]
        ev, vec, _ = oracle_mod.eigen3(c)
        refs[i, :3] = ev
        refs[i, 3:] = vec.ravel()
    out = debug_eval("eigen3", mats)
    assert np.array_equal(_bits(out), _bits(refs))
