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
    # first-contact evidence for an N-GPU run: what RCCL itself reports for every rank's communicator (cape_comm_info)
    assert g["native_rccl"] and not g["torch_fallback_taken"]
    assert g["rccl_ranks_seen"] == [1] and g["rccl_devices_seen"] == [[0, 0]] and g["rccl_comm"][0]["has_comm"] == 1, g
    assert g["payload_bytes_per_frame"] <= 1229, g  # <= 1.2 KB per frame on the TUM-like stream
    assert "exposed_ms_per_step" in g
    pc = out["parity_check"]
    assert pc["frames"] == 256 and pc["labels_equal"] and pc["counts_equal"] and pc["segments_bitwise"] and pc["all_ranks_ok"]


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
    assert pc["frames"] == 512 and pc["labels_equal"] and pc["segments_bitwise"] and pc["planes"] > 0
    cyl = out["cylinders_on"]
    assert cyl["value"] > 0 and cyl["parity_check"]["labels_equal"] and cyl["parity_check"]["cylinders_bitwise"]
    assert set(cyl["kernel_ms"]) == {"cape_cell_moments_kernel", "cape_cell_plane_kernel", "cape_grow_kernel"}
    assert out["roofline"]["frac"] > 0.05 and out["roofline"]["bound"] == "hbm"
    # round 6: the legs beyond the fast kernels' fixed shapes, each checked against the oracle inside the run
    b = out["beyond_fixed_capacities"]
    assert b["wide_grid"]["general_instance_frames"] == 0 and b["wide_grid"]["frames"] == 256 and b["wide_grid"]["parity_check"]["labels_equal"]
    assert b["record_chain"]["frames_of_more_than_64_segments"] == 2 and b["record_chain"]["spill_records_used"] == 2
    assert b["record_chain"]["most_segments_in_a_frame"] == 116 and b["record_chain"]["parity_check"]["segments_bitwise"]


@pytest.mark.parametrize("scaling", ["weak", "strong"])
def test_two_ranks_share_the_gpu_through_gloo(scaling):
    """the N > 1 branches of bench.py (shard arithmetic, the budget all-reduce, the torch gather, max over
    ranks, per-rank parity, weak / strong stream sizes) had never run with world > 1.  CAPE_BENCH_BACKEND=gloo lets two ranks
    share the one GPU of this box (RCCL refuses that); the line is a dry run of the code paths, not a scaling number."""
    out = _run(["--gpus", "2", "--gather", "torch", "--scaling", scaling, "--steps", "3", "--warmup", "1", "--frames", "128",
                "--no-cpu-baseline"], {"CAPE_BENCH_BACKEND": "gloo"})
    assert out["n_gpus"] == 2 and out["scaling"] == scaling
    total = 8 * 128 if scaling == "strong" else 2 * 128
    assert out["config"]["stream_frames"] == total and out["config"]["frames_per_step_per_gpu"] == total // 2
    assert "gloo" in out["config"]["backend"]
    assert out["ranks"]["launcher"].startswith("self-spawned") and len(out["ranks"]["ms_per_step"]) == 2
    assert out["ranks"]["ms_per_step_max"] >= out["ranks"]["ms_per_step_min"] > 0
    g = out["gather"]
    assert g["ok"] and g["frames"] == total and g["overflow"] == 0 and g["path"].startswith("torch")
    assert not g["native_rccl"] and isinstance(g["rccl_comm"], str) and "absent by design" in g["rccl_comm"]
    assert "exposed_ms_per_step" in g and g["planes"] > 0
    pc = out["parity_check"]
    assert pc["ranks_checked"] == 2 and pc["all_ranks_ok"] and pc["labels_equal"] and pc["segments_bitwise"]
    assert abs(out["value"] - total * out["steps"] / (out["ms_per_step"] * 1e-3 * out["steps"])) < 1e-6 * out["value"]
    assert "cpu_baseline" not in out and "cylinders_on" not in out  # rank 0 at N = 1 only


def test_default_line_names_the_bound_that_holds():
    """the driver-visible line says that A1 sits at about half the HBM peak AND at about its VALU-issue floor,
    with the raw-uint16 launch (half the bytes, same time) beside it."""
    out = _run(["--gpus", "1", "--steps", "3", "--warmup", "1", "--no-cpu-baseline", "--no-polygons", "--no-parity-check"])
    r = out["roofline"]
    assert r["bound"] == "hbm" and r["kernel"] == "cape_cell_moments_kernel"
    v = r["valu_issue"]
    assert v is not None and 0.5 < v["frac_of_issue_floor"] <= 1.05, v
    assert v["f64_rate_insts_per_launch"] > v["other_valu_insts_per_launch"] * 0.5
    u = r["u16_launch"]
    assert 0.7 < u["launch_ms"] / r["launch_ms"] < 1.4, "half the bytes in about the same time"
    assert u["frac"] < 0.75 * r["frac"]
