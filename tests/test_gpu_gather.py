"""Multi-GPU exchange on the device side (BASELINE.json configs[3], [4]): the packed primitive lists
written by cape_pack_primitives equal the ones built from ORACLE results byte for byte, and they travel unchanged
through (a) ONE ncclAllGather issued by libcape_hip itself (cape_comm_init + cape_gather_primitives, RCCL resolved with
dlopen) and (b) torch.distributed's nccl (= RCCL) group.  The box has one GPU, so the world is 1 here; the N-rank
arithmetic (sharding, ragged shards, parsing in rank order) is covered by tests/test_multigpu_gloo.py."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


class _DevMem:
    def __init__(self, ptr, nbytes):
        self.__cuda_array_interface__ = {"shape": (nbytes,), "typestr": "|u1", "data": (ptr, False), "version": 2}


def _stream(n, seed=2):
    from cape_amd import synth

    return np.stack([synth.tunnel(seed=seed, frame=f) if f % 4 == 3 else synth.tumlike(seed=seed, frame=f) for f in range(n)])


def _setup(n, labels=True, planes_per_frame=0, cylinders_per_frame=0, frames_capacity=None):
    from cape_amd import Extractor, synth

    intr = dict(synth.TUM_FR1_INTRINSICS)
    ex = Extractor(640, 480, cylinders=True, max_batch=max(n, frames_capacity or 0), **intr)
    lay = ex.gather_configure(frames_capacity or n, planes_per_frame, cylinders_per_frame, labels=labels)
    return ex, lay, intr


def test_packed_lists_equal_oracle_bytes(oracle_mod):
    from cape_amd.dist import Shard, packed_layout
    from packing import pack_oracle

    n = 10
    frames = _stream(n)
    ex, lay, intr = _setup(n, labels=True, frames_capacity=12)  # two unused frame slots: must come out zeroed
    assert lay == packed_layout(12, 768, 16, 8, labels=True)
    ex.extract_host(frames)
    ex.pack(n, first_frame=40)
    got = ex.packed_host()
    res = ex.results(n, with_boundary=False)
    orc = oracle_mod.Oracle(640, 480, cylinders=True, **intr)
    refs = [orc.run(frames[f]) for f in range(n)]
    want = pack_oracle(refs, 40, lay, labels=True, status=res.records["header"]["status"])
    sh = Shard(got, lay)
    assert int(sh.header["n_frames"]) == n and sh.first_frame == 40 and int(sh.header["overflow"]) == 0
    assert int(sh.header["n_planes_total"]) == sum(len(r.planes) for r in refs) > 0
    assert int(sh.header["n_cylinders_total"]) == sum(len(r.cylinders) for r in refs) > 0
    assert np.array_equal(got, want), "packed bytes differ from the oracle-built payload"
    # packing again (other staging slot) gives the same bytes: what travels depends on the frames only
    ex.pack(n, first_frame=40)
    assert np.array_equal(ex.packed_host(), want)
    ex.close()


def test_overflow_is_reported_not_silent(oracle_mod):
    from cape_amd import PACKED_PLANES_DROPPED
    from cape_amd.dist import Shard

    n = 6
    frames = _stream(n)
    ex, lay, intr = _setup(n, labels=False, planes_per_frame=1, cylinders_per_frame=1)
    ex.extract_host(frames)
    ex.pack(n)
    sh = Shard(ex.packed_host(), lay)
    orc = oracle_mod.Oracle(640, 480, cylinders=True, **intr)
    total = sum(len(orc.run(frames[f]).planes) for f in range(n))
    assert total > lay["planes_capacity"] == n
    assert int(sh.header["n_planes_total"]) == total
    assert int(sh.header["overflow"]) & PACKED_PLANES_DROPPED
    assert [int(f["n_planes"]) for f in sh.frames] == [len(orc.run(frames[f]).planes) for f in range(n)]
    # a budget of CAPE_MAX_PLANES per frame can never overflow
    lay = ex.gather_configure(n, 64, 64)
    ex.pack(n)
    sh = Shard(ex.packed_host(), lay)
    assert int(sh.header["overflow"]) == 0 and int(sh.header["n_planes_total"]) == total
    ex.close()


def test_native_rccl_gather(oracle_mod):
    """cape_comm_unique_id -> cape_comm_init -> cape_gather_primitives: the C layer calls librccl itself."""
    import torch
    from cape_amd.dist import primitives_by_frame, unpack_gathered

    n = 8
    frames = _stream(n, seed=5)
    ex, lay, intr = _setup(n, labels=True)
    uid = ex.comm_unique_id()
    assert len(uid) == 128
    assert ex.comm_info()["has_comm"] == 0 and ex.comm_info()["nranks"] == -1
    ex.comm_init(uid, 0, 1)
    info = ex.comm_info()  # what RCCL itself reports (ncclCommCount / ncclCommUserRank / ncclCommCuDevice)
    assert info["has_comm"] == 1 and (info["nranks"], info["rank"], info["device"]) == (1, 0, 0), info
    assert (info["init_nranks"], info["init_rank"], info["handle_device"]) == (1, 0, 0)
    recv = torch.zeros(lay["bytes_per_rank"], dtype=torch.uint8, device="cuda")
    stream = torch.cuda.current_stream().cuda_stream
    for rep in range(3):  # both staging slots get reused
        ex.extract_host(frames, stream)
        ex.gather(n, 0, recv.data_ptr(), stream)
    ex.gather_wait(host_sync=True)
    got = recv.cpu().numpy()
    assert np.array_equal(got, ex.packed_host()), "gathered bytes differ from the packed staging slot"
    orc = oracle_mod.Oracle(640, 480, cylinders=True, **intr)
    by_frame = primitives_by_frame(unpack_gathered(got, 1, lay))
    for f in range(n):
        r = orc.run(frames[f])
        planes, cyls = by_frame[f]
        assert len(planes) == len(r.planes) and len(cyls) == len(r.cylinders)
        if len(planes):
            assert np.array_equal(planes["normal"].view(np.uint64), np.ascontiguousarray(r.planes[:, 0:3]).view(np.uint64))
            assert np.array_equal(planes["d"].view(np.uint64), np.ascontiguousarray(r.planes[:, 3]).view(np.uint64))
    # stream-ordered wait instead of a host wait
    ex.extract_host(frames, stream)
    ex.gather(n, 0, recv.data_ptr(), stream)
    ex.gather_wait(stream, host_sync=False)
    torch.cuda.synchronize()
    assert np.array_equal(recv.cpu().numpy(), got)
    ex.comm_destroy()
    ex.close()


def test_torch_nccl_group_gather(oracle_mod):
    """The same payload through torch.distributed's nccl (= RCCL) process group, zero-copy view of the library's slot."""
    import torch
    import torch.distributed as dist
    from cape_amd.dist import all_gather_bytes, unpack_gathered

    n = 5
    frames = _stream(n, seed=7)
    ex, lay, intr = _setup(n, labels=False)
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", str(29600 + os.getpid() % 300))
    created = not dist.is_initialized()
    if created:
        dist.init_process_group("nccl", rank=0, world_size=1)
    try:
        stream = torch.cuda.current_stream().cuda_stream
        ex.extract_host(frames, stream)
        ptr = ex.pack(n, 0, stream)
        local = torch.as_tensor(_DevMem(ptr, lay["bytes_per_rank"]), device="cuda")
        out = all_gather_bytes(local, 1)
        torch.cuda.synchronize()
        got = out.cpu().numpy()
        assert np.array_equal(got, ex.packed_host())
        sh = unpack_gathered(got, 1, lay)[0]
        orc = oracle_mod.Oracle(640, 480, cylinders=True, **intr)
        assert [int(f["n_planes"]) for f in sh.frames] == [len(orc.run(frames[f]).planes) for f in range(n)]
    finally:
        if created:
            dist.destroy_process_group()
    ex.close()


def _two_proc_worker(rank, world, port, n_frames, q):
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for p in (os.path.join(root, "rgb-d-slam_amd", "python"), os.path.join(root, "oracle"), os.path.join(root, "tests")):
        sys.path.insert(0, p)
    import torch
    import torch.distributed as dist

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import cape_oracle_py as O
    from cape_amd import Extractor, synth
    from cape_amd.dist import all_gather_bytes, largest_shard, primitives_by_frame, shard_range, unpack_gathered

    ok = True
    intr = dict(synth.TUM_FR1_INTRINSICS)
    a, b = shard_range(n_frames, rank, world)
    frames = _stream(n_frames)
    ex = Extractor(640, 480, cylinders=True, device=0, max_batch=largest_shard(n_frames, world), **intr)
    lay = ex.gather_configure(largest_shard(n_frames, world), 16, 8, labels=True)
    ex.extract_host(frames[a:b])
    ex.pack(b - a, first_frame=a)
    local = torch.from_numpy(ex.packed_host().copy())          # the device-packed shard of THIS process
    gathered = all_gather_bytes(local, world).numpy()
    shards = unpack_gathered(gathered, world, lay)
    by_frame = primitives_by_frame(shards)
    ok = ok and sorted(by_frame) == list(range(n_frames))
    orc = O.Oracle(640, 480, cylinders=True, **intr)
    for f in range(n_frames):
        r = orc.run(frames[f])
        planes, cyls = by_frame[f]
        ok = ok and len(planes) == len(r.planes) and len(cyls) == len(r.cylinders)
        if len(planes):
            ok = ok and np.array_equal(planes["normal"].view(np.uint64), np.ascontiguousarray(r.planes[:, 0:3]).view(np.uint64))
            ok = ok and np.array_equal(planes["d"].view(np.uint64), np.ascontiguousarray(r.planes[:, 3]).view(np.uint64))
    for s in shards:
        for k in range(len(s.frames)):
            ok = ok and np.array_equal(s.plane_labels[k], orc.run(frames[s.first_frame + k]).plane_labels.astype(np.uint8))
    ex.close()
    dist.barrier()
    dist.destroy_process_group()
    q.put((rank, bool(ok)))


def test_two_process_sharded_gather_on_one_gpu(oracle_mod):
    """Two PROCESSES, each with its own handle, shard a 9-frame stream (ragged: 5 + 4), pack their shards on the device
    and all-gather the packed bytes -- through gloo, because RCCL refuses two ranks on one GPU and the box has one.  What is
    exercised beyond the world-1 tests: real device payloads of different shards, first_frame offsets, ragged frame
    counts, parsing in rank order.  The transport at N > 1 (ncclAllGather) is the one thing left to the driver's run."""
    import torch.multiprocessing as mp

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29800 + os.getpid() % 150
    procs = [ctx.Process(target=_two_proc_worker, args=(r, 2, port, 9, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert sorted(res) == [(0, True), (1, True)]


def test_packed_lists_of_a_frame_of_116_segments(oracle_mod):
    """A frame that continues in a spill record (more than 64 plane segments) is packed WHOLE: the pack kernel follows
    cape_frame_header.next_record, the planes of the second record land behind those of the first with their frame-wide segment
    index -- byte for byte what the oracle's lists give."""
    from cape_amd import Extractor, synth
    from cape_amd.dist import Shard
    from packing import pack_oracle
    from test_gpu_parity import _checkerboard_of_facets

    W, H = 1280, 960
    big, intr = _checkerboard_of_facets(W, H)
    frames = np.stack([synth.room(seed=1, frame=0, width=W, height=H, intr=intr), big,
                       synth.tunnel(seed=1, frame=0, width=W, height=H, intr=intr), big])
    n = len(frames)
    ex = Extractor(W, H, cylinders=True, max_batch=n, **intr)
    lay = ex.gather_configure(n, 80, 8, labels=True)
    ex.extract_host(frames)
    ex.pack(n, first_frame=8)
    got = ex.packed_host()
    res = ex.results(n, with_boundary=False)
    orc = oracle_mod.Oracle(W, H, cylinders=True, **intr)
    refs = [orc.run(f) for f in frames]
    assert len(refs[1].planes) > 64
    want = pack_oracle(refs, 8, lay, labels=True, status=res.records["header"]["status"])
    sh = Shard(got, lay)
    assert int(sh.header["overflow"]) == 0 and int(sh.header["n_planes_total"]) == sum(len(r.planes) for r in refs)
    assert np.array_equal(got, want), "packed bytes differ from the oracle-built payload"
    ex.close()


def test_configs3_gather_leg_every_frame_with_labels(oracle_mod):
    """BASELINE.json configs[3], the one-GPU leg of the sharded TUM-like stream at its full shard size (2 048 frames), label grids
    on: what the native RCCL gather delivers -- plane lists, cylinder lists, both label grids of EVERY frame -- against the oracle
    (threaded), not a sample."""
    import concurrent.futures as cf
    import threading

    import torch
    from cape_amd import Extractor, synth, synth_gpu
    from cape_amd.dist import unpack_gathered

    n = 2048
    intr = dict(synth.TUM_FR1_INTRINSICS)
    dev = synth_gpu.stream("tumlike", 2, n, start=0, device="cuda", chunk=64)
    ex = Extractor(640, 480, cylinders=True, max_batch=n, **intr)
    st = torch.cuda.current_stream().cuda_stream
    ex.extract_device(dev.data_ptr(), n, st)
    n_pl, n_cy, most = ex.count_primitives(n)
    lay = ex.gather_configure(n, planes_per_frame=int(np.ceil(1.15 * n_pl / n)) + 1, cylinders_per_frame=int(np.ceil(1.15 * n_cy / n)) + 1,
                              labels=True)
    ex.comm_init(ex.comm_unique_id(), 0, 1)
    recv = torch.zeros(lay["bytes_per_rank"], dtype=torch.uint8, device="cuda")
    ex.gather(n, 0, recv.data_ptr(), st)
    ex.gather_wait(host_sync=True)
    sh = unpack_gathered(recv.cpu().numpy(), 1, lay)[0]
    assert int(sh.header["overflow"]) == 0 and int(sh.header["n_frames"]) == n and int(sh.header["n_planes_total"]) == n_pl
    local = threading.local()

    def bits(a):
        return np.ascontiguousarray(a).view(np.uint64)

    def check(args):
        f, depth = args
        if not hasattr(local, "orc"):
            local.orc = oracle_mod.Oracle(640, 480, cylinders=True, **intr)
        r = local.orc.run(depth)
        planes, cyls = sh.frame_planes(f), sh.frame_cylinders(f)
        ok = (len(planes) == len(r.planes) and len(cyls) == len(r.cylinders) and int(sh.frames[f]["n_plane_segments"]) == len(r.segments)
              and np.array_equal(sh.plane_labels[f], r.plane_labels.astype(np.uint8)) and np.array_equal(sh.cyl_labels[f], r.cyl_labels.astype(np.uint8)))
        if ok and len(planes):
            ok = (np.array_equal(bits(planes["normal"]), bits(r.planes[:, 0:3])) and np.array_equal(bits(planes["d"]), bits(r.planes[:, 3]))
                  and np.array_equal(planes["segment"], r.planes[:, 19].astype(np.uint32)))
        if ok and len(cyls):
            ok = np.array_equal(bits(cyls["axis"]), bits(r.cylinders[:, 0:3]))
        return -1 if ok else f

    bad = []
    with cf.ThreadPoolExecutor(max(2, min(16, os.cpu_count() or 2))) as pool:
        for c0 in range(0, n, 256):
            host = dev[c0:c0 + 256].cpu().numpy()
            bad += [f for f in pool.map(check, [(c0 + k, host[k]) for k in range(len(host))]) if f >= 0]
    assert not bad, f"{len(bad)} of {n} gathered frames differ from the oracle, first: {bad[:8]}"
    ex.comm_destroy()
    ex.close()
