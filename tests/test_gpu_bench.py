"""bench.py keeps its contract: ONE JSON line with the driver's keys plus `roofline` (and `gather` on the sharded
workload), for the default single-GPU workload (BASELINE.json configs[1]) and for the configs[3] code path that the driver
runs at N > 1 -- exercised here on one GPU (world 1, CAPE_BENCH_FORCE_GATHER=1) with both transports."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KEYS = {"metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
        "dtype", "data", "config", "roofline"}


def _run(args, env=None):
    e = dict(os.environ, **(env or {}))
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, capture_output=True, text=True, timeout=600, env=e)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, "exactly one JSON line"
    return json.loads(lines[0])


def test_bench_line_single_gpu():
    d = _run(["--frames", "256", "--steps", "3", "--warmup", "1", "--no-cpu-baseline"])
    assert KEYS <= set(d)
    assert d["n_gpus"] == 1 and d["steps"] == 3 and d["warmup"] == 1 and d["unit"] == "frames/s" and d["vs_baseline"] is None
    assert "configs[1]" in d["config"]["workload"] and d["config"]["unique_frames_per_gpu"] == 256
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0 and 0 < r["frac"] < 1
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9
    assert d["value"] > 1e5


@pytest.mark.parametrize("gather", ["native", "torch"])
def test_bench_sharded_workload_on_one_gpu(gather):
    d = _run(["--gpus", "1", "--frames", "128", "--steps", "3", "--warmup", "1", "--no-cpu-baseline", "--gather", gather],
             env={"CAPE_BENCH_FORCE_GATHER": "1", "MASTER_PORT": str(29700 + os.getpid() % 200)})
    assert KEYS <= set(d) and "gather" in d
    assert "configs[3]" in d["config"]["workload"] and d["config"]["scene"] == "tumlike"
    g = d["gather"]
    assert g["ok"] and g["frames"] == 128 and g["planes"] > 128 and g["overflow"] == 0
    assert d["config"]["stream_frames"] == 128 and d["scaling"] == "weak"
