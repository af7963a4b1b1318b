"""bench.py end to end on the GPU box: the launcher paths the driver uses, and the in-run proof of work."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, env_extra=None, timeout=900):
    env = dict(os.environ)
    env.update(env_extra or {})
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, capture_output=True, text=True, timeout=timeout)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, f"stdout must carry exactly one JSON line, got {len(lines)}: {p.stdout[-1000:]}"
    return json.loads(lines[0])


def test_self_spawned_launcher_with_gather():
    """`python bench.py --gpus N` with no launcher around it starts its own ranks under torch.distributed.run.  One GPU
    here, so the path runs at N = 1 (--spawn) with the multi-GPU exchange forced on: native RCCL communicator, device
    packing, one collective per step, the budget measured on the stream."""
    out = _run(["--gpus", "1", "--spawn", "--steps", "3", "--warmup", "1", "--frames", "256", "--no-cpu-baseline"],
               {"CAPE_BENCH_FORCE_GATHER": "1"})
    assert out["n_gpus"] == 1 and out["steps"] == 3
    assert out["ranks"]["launcher"].startswith("self-spawned")
    assert len(out["ranks"]["ms_per_step"]) == 1
    g = out["gather"]
    assert g["ok"] and g["overflow"] == 0 and g["frames"] == 256
    assert g["path"].startswith("native")
    assert g["payload_bytes_per_frame"] <= 1229, g  # <= 1.2 KB per frame on the TUM-like stream
    assert "exposed_ms_per_step" in g
    pc = out["parity_check"]
    assert pc["frames"] == 64 and pc["labels_equal"] and pc["counts_equal"] and pc["segments_bitwise"] and pc["all_ranks_ok"]


def test_gather_to_root_path():
    out = _run(["--gpus", "1", "--spawn", "--steps", "2", "--warmup", "1", "--frames", "128", "--no-cpu-baseline", "--gather-root"],
               {"CAPE_BENCH_FORCE_GATHER": "1"})
    assert out["gather"]["ok"] and "root" in out["gather"]["path"]


def test_default_workload_proves_its_work():
    """The N = 1 line (reduced batch to keep the test short): parity_check against the oracle inside the run, the
    reference-faithful cylinders-on leg next to the plane-only value."""
    out = _run(["--gpus", "1", "--steps", "3", "--warmup", "1", "--frames", "512", "--no-cpu-baseline"])
    assert out["ranks"]["launcher"] == "single process"
    pc = out["parity_check"]
    assert pc["frames"] == 64 and pc["labels_equal"] and pc["segments_bitwise"] and pc["planes"] > 0
    cyl = out["cylinders_on"]
    assert cyl["value"] > 0 and cyl["parity_check"]["labels_equal"] and cyl["parity_check"]["cylinders_bitwise"]
    assert set(cyl["kernel_ms"]) == {"cape_cell_moments_kernel", "cape_cell_plane_kernel", "cape_grow_kernel"}
    assert out["roofline"]["frac"] > 0.05 and out["roofline"]["bound"] == "hbm"
