"""N>1 path on CPU: world_size-2 gloo processes run the sharding + gather plumbing of cape_amd.dist on packed
cape_primitive_summary records (the records themselves come from the oracle here; on the GPU box they come from
libcape_hip and the same calls run over RCCL)."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_shard_range_covers_everything():
    from cape_amd.dist import shard_range

    for n in (0, 1, 7, 8, 4096, 4099):
        for world in (1, 2, 3, 8):
            spans = [shard_range(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1


def _make_summaries(frames_idx):
    """Deterministic fake extraction of frame ids -> SUMMARY records (content is a pure function of the id)."""
    from cape_amd import SUMMARY_DTYPE

    out = np.zeros(len(frames_idx), SUMMARY_DTYPE)
    for k, f in enumerate(frames_idx):
        out["n_planes"][k] = f % 5
        out["n_plane_segments"][k] = f % 7
        out["planes"]["d"][k, 0] = 1000.0 + f
        out["planes"]["normal"][k, 0] = (0.0, 0.0, -1.0)
    return out


def _worker(rank, world, port, n_frames, q):
    for p in (os.path.join(ROOT, "rgb-d-slam_amd", "python"),):
        sys.path.insert(0, p)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from cape_amd import SUMMARY_DTYPE
    from cape_amd.dist import gather_ragged, gather_summaries, shard_range, summaries_from_bytes

    a, b = shard_range(n_frames, rank, world)
    local = _make_summaries(range(a, b))
    t = torch.from_numpy(local.view(np.uint8).copy())
    counts = [shard_range(n_frames, r, world)[1] - shard_range(n_frames, r, world)[0] for r in range(world)]
    if len(set(counts)) == 1:
        g = gather_summaries(t, world).reshape(-1)
    else:
        g = gather_ragged(t, counts, SUMMARY_DTYPE.itemsize)
    allrec = summaries_from_bytes(g.numpy().tobytes())
    ok = len(allrec) == n_frames and np.array_equal(allrec["planes"]["d"][:, 0], 1000.0 + np.arange(n_frames))
    ok = ok and np.array_equal(allrec["n_planes"], np.arange(n_frames) % 5)
    # max-over-ranks timing reduction used by bench.py
    el = torch.tensor([float(rank + 1)], dtype=torch.float64)
    dist.all_reduce(el, op=dist.ReduceOp.MAX)
    ok = ok and el.item() == world
    dist.barrier()
    dist.destroy_process_group()
    q.put((rank, bool(ok)))


def _run(n_frames, world=2):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000) + n_frames % 13
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_frames, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert sorted(res) == [(r, True) for r in range(world)]


def test_gloo_gather_even_shards():
    _run(64)


def test_gloo_gather_ragged_shards():
    _run(37)
