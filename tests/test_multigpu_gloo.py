"""N>1 path on CPU: world_size-2 gloo processes shard a TUM-like stream (BASELINE.json configs[3]) by contiguous
frame blocks, each runs ITS frames through the oracle, packs the primitive lists in the wire format of
include/cape_hip.h (cape_packed_*), all-gathers the fixed-size buffers and parses them with the product's parser
(cape_amd.dist).  The gathered lists must equal the unsharded run frame for frame.  On the GPU box the same bytes come
from cape_pack_primitives and travel through ONE ncclAllGather issued by libcape_hip (tests/test_gpu_gather.py)."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
N_FRAMES_EVEN, N_FRAMES_RAGGED = 8, 7
W, H = 640, 480


def test_shard_range_covers_everything():
    from cape_amd.dist import largest_shard, shard_range

    for n in (0, 1, 7, 8, 4096, 4099, 8 * 2048):
        for world in (1, 2, 3, 8):
            spans = [shard_range(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1
            assert max(sizes) == largest_shard(n, world)


def test_packed_layout_matches_struct_sizes():
    from cape_amd.dist import packed_layout

    lay = packed_layout(4, 768, 16, 8, labels=True)
    assert lay["frames_offset"] == 48
    assert lay["planes_offset"] == 48 + 4 * 24
    assert lay["cylinders_offset"] == lay["planes_offset"] + 64 * 152
    assert lay["plane_labels_offset"] == lay["cylinders_offset"] + 32 * 32
    assert lay["bytes_per_rank"] == lay["cyl_labels_offset"] + 4 * 768
    assert all(lay[k] % 16 == 0 for k in lay if k.endswith("_offset") or k == "bytes_per_rank")


def _oracle_stream(frame_ids):
    import cape_oracle_py as O
    from cape_amd import synth

    intr = dict(synth.TUM_FR1_INTRINSICS)
    orc = O.Oracle(W, H, cylinders=True, **intr)
    out = []
    for f in frame_ids:
        # a TUM-like stream with a tunnel frame mixed in so that cylinders travel too
        depth = synth.tunnel(seed=2, frame=f) if f % 4 == 3 else synth.tumlike(seed=2, frame=f)
        out.append(orc.run(depth))
    return out


def _worker(rank, world, port, n_frames, q):
    for p in (os.path.join(ROOT, "rgb-d-slam_amd", "python"), os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
        sys.path.insert(0, p)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from cape_amd.dist import (all_gather_bytes, broadcast_unique_id, largest_shard, packed_layout, primitives_by_frame,
                               shard_range, unpack_gathered)
    from packing import pack_oracle

    ok = True
    # the communicator id travels exactly like this on the GPU box (128 opaque bytes from rank 0)
    uid = broadcast_unique_id(lambda: bytes(range(128)), rank)
    ok = ok and uid == bytes(range(128))

    cells = (W // 20) * (H // 20)
    lay = packed_layout(largest_shard(n_frames, world), cells, planes_per_frame=16, cylinders_per_frame=8, labels=True)
    a, b = shard_range(n_frames, rank, world)
    mine = _oracle_stream(range(a, b))
    local = pack_oracle(mine, a, lay, labels=True)
    gathered = all_gather_bytes(torch.from_numpy(local), world).numpy()
    shards = unpack_gathered(gathered, world, lay)
    ok = ok and [int(s.header["n_frames"]) for s in shards] == [shard_range(n_frames, r, world)[1] - shard_range(n_frames, r, world)[0] for r in range(world)]
    ok = ok and all(int(s.header["overflow"]) == 0 for s in shards)
    by_frame = primitives_by_frame(shards)
    ok = ok and sorted(by_frame) == list(range(n_frames))
    # the unsharded run (every rank recomputes it: the check needs no communication)
    whole = _oracle_stream(range(n_frames))
    n_cyl = 0
    for f, r in enumerate(whole):
        planes, cyls = by_frame[f]
        ok = ok and len(planes) == len(r.planes) and len(cyls) == len(r.cylinders)
        if len(r.planes):
            ok = ok and np.array_equal(planes["normal"].view(np.uint64), np.ascontiguousarray(r.planes[:, 0:3]).view(np.uint64))
            ok = ok and np.array_equal(planes["d"].view(np.uint64), np.ascontiguousarray(r.planes[:, 3]).view(np.uint64))
            ok = ok and np.array_equal(planes["point_count"], r.planes[:, 9].astype(np.uint32))
        if len(r.cylinders):
            ok = ok and np.array_equal(cyls["axis"].view(np.uint64), np.ascontiguousarray(r.cylinders[:, 0:3]).view(np.uint64))
            n_cyl += len(r.cylinders)
    ok = ok and n_cyl > 0
    for s in shards:
        for k in range(len(s.frames)):
            g = s.first_frame + k
            ok = ok and np.array_equal(s.plane_labels[k], whole[g].plane_labels.astype(np.uint8))
            ok = ok and np.array_equal(s.cyl_labels[k], whole[g].cyl_labels.astype(np.uint8))
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
    res = [q.get(timeout=300) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert sorted(res) == [(r, True) for r in range(world)]


def test_gloo_gather_even_shards(oracle_mod):
    _run(N_FRAMES_EVEN)


def test_gloo_gather_ragged_shards(oracle_mod):
    _run(N_FRAMES_RAGGED)
