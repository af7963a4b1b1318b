#!/usr/bin/env python3
"""bench.py -- depth frames/s of CAPE primitive extraction on MI355X (BASELINE.json metric).

Workload (N=1): BASELINE.json configs[1], "640x480 synthetic planar-room depth stream, 1xMI355X, plane extraction
only": a 4 096-frame stream, every frame distinct, rendered on the GPU (cape_amd.synth_gpu) and resident in HBM.  A *step*
is one pass of the hot path (stage A cell fit + stage B grow/merge/boundary) over that batch.
Workload (N>1, one process per GPU launched by torch.distributed.run): BASELINE.json configs[3], "batched 640x480 TUM
fr1_desk stream sharded across 8xMI355X, RCCL gather of primitive lists": ONE TUM-like stream of N x 2 048 frames cut
in contiguous blocks (cape_amd.dist.shard_range), rank r extracts frames [a_r, b_r), and every step ends with ONE
ncclAllGather of the packed primitive lists, issued by libcape_hip itself (cape_gather_primitives) on its own stream so
that it runs under the next step's kernels.  `--scaling strong` keeps the stream at 8 x 2 048 frames for every N.

Prints ONE JSON line on rank 0 (see the driver contract in the task description) with two extra objects:
  roofline     -- the dominant kernel's algorithmic bytes per launch / its mean launch duration (HIP events on the
                  launch stream, inside the timed region) against the 8 TB/s HBM peak;
  cpu_baseline -- the CPU oracle (a port of the reference algorithm, NOT the product) timed on this host.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "rgb-d-slam_amd", "python"))

HBM_PEAK_BYTES_S = 8.0e12  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E peak BW, spec


class _DevMem:
    """Expose a raw device allocation of libcape_hip to torch (no copy) for the RCCL gather."""

    def __init__(self, ptr, nbytes):
        self.__cuda_array_interface__ = {"shape": (nbytes,), "typestr": "|u1", "data": (ptr, False), "version": 2}


def available_cores():
    """Cores this process may actually use: affinity mask capped by the cgroup CPU quota (containers)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                n = min(n, max(1, q // per))
        except Exception:
            pass
    return n


def cpu_baseline(unique_frames, intr, budget_s=12.0, cylinders=False, scene="room"):
    """Oracle (port of the reference CPU path) on this host: 1 thread (how the reference runs it), bounded sample."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import concurrent.futures as cf

    import numpy as np

    import cape_oracle_py as O

    H, W = unique_frames.shape[1:]
    orc = O.Oracle(W, H, cylinders=cylinders, **intr)
    orc.run_many(unique_frames[:2])  # warm-up
    t0 = time.perf_counter()
    orc.run_many(unique_frames[:8])
    per_frame = (time.perf_counter() - t0) / 8
    n = int(max(32, min(4000, budget_s * 0.5 / per_frame)))
    reps = -(-n // unique_frames.shape[0])
    sample = np.concatenate([unique_frames] * reps)[:n]
    t0 = time.perf_counter()
    orc.run_many(sample)
    dt1 = time.perf_counter() - t0
    # frame-parallel over all host cores (BASELINE.md mode B): every thread runs the same frames through its own oracle
    cores = available_cores()
    per_thread = sample[: max(32, min(len(sample), int(2.0 / per_frame)))]  # ~2 s of work per thread
    oracles = [O.Oracle(W, H, cylinders=cylinders, **intr) for _ in range(cores)]
    t0 = time.perf_counter()
    with cf.ThreadPoolExecutor(cores) as pool:
        list(pool.map(lambda o: o.run_many(per_thread), oracles))
    dtn = time.perf_counter() - t0
    n_all = cores * len(per_thread)
    return {
        "value": n / dt1, "unit": "frames/s", "cores": 1, "kind": "port",
        "sample": f"{n} frames of the same {scene} stream ({unique_frames.shape[0]} distinct, read back from the device), "
                  f"oracle/libcape_oracle.so (g++ -O2 -ffp-contract=off), "
                  f"{dt1:.1f} s single thread",
        "all_cores": {"value": n_all / dtn, "cores": cores, "seconds": dtn, "frames": n_all},
    }


def self_spawn(n_gpus):
    """Re-run this very command line as N ranks under torch.distributed.run and forward rank 0's JSON line.
    Everything else the children print goes to stderr, so stdout carries exactly one line."""
    import socket
    import subprocess

    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    argv = [a for a in sys.argv[1:] if a != "--spawn"]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n_gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + argv
    env = dict(os.environ, CAPE_BENCH_SPAWNED="1")
    env.pop("CAPE_BENCH_FORCE_SPAWN", None)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC only on these hosts (RCCL needs it)
    env.setdefault("OMP_NUM_THREADS", "1")
    proc = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True)
    line = None
    for ln in proc.stdout.splitlines():
        if ln.startswith("{") and '"metric"' in ln:
            line = ln
        else:
            print(ln, file=sys.stderr)
    if proc.returncode != 0 or line is None:
        print(f"bench.py: the {n_gpus}-rank run failed (exit code {proc.returncode}, JSON line {'present' if line else 'absent'})", file=sys.stderr)
        return proc.returncode or 1
    print(line, flush=True)
    return 0


def parity_check(ex, frames_host, intr, cylinders, n):
    """The work proves itself: the first `n` frames of the LAST timed step (their results are still on the device) against
    the CPU oracle run on the very same frames -- label grids, counts, and every plane-segment record bit for bit.  The
    oracle is the checker here, never the thing measured.  frames_host: an array of frames, or a callable (first, count) -> frames
    (the whole batch does not have to sit in host memory at once); the oracle runs on a pool of threads, one Oracle object each."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import concurrent.futures as cf
    import threading

    import numpy as np

    import cape_oracle_py as O

    fetch = frames_host if callable(frames_host) else (lambda a, c: frames_host[a:a + c])
    if not callable(frames_host):
        n = min(n, len(frames_host))
    first = fetch(0, 1)
    H, W = first.shape[1:]
    res = ex.results(n, with_boundary=False)
    local = threading.local()

    def bits(a):
        return np.ascontiguousarray(a).view(np.uint64)

    def one(args):
        f, depth = args
        if not hasattr(local, "orc"):
            local.orc = O.Oracle(W, H, cylinders=cylinders, **intr)
        r = local.orc.run(depth)
        hdr = res.records["header"][f]
        labels = bool(np.array_equal(res.plane_labels[f], r.plane_labels) and np.array_equal(res.cyl_labels[f], r.cyl_labels))
        counts = bool(hdr["n_plane_segments"] == len(r.segments) and hdr["n_planes"] == len(r.planes)
                      and hdr["n_cylinders"] == len(r.cylinders) and hdr["n_seeds"] == len(r.seeds))
        segs = res.segments(f)
        seg_ok = len(segs) == len(r.segments)
        if seg_ok and len(segs):
            o = r.segments
            seg_ok = bool(
                np.array_equal(bits(segs["normal"]), bits(o[:, 0:3])) and np.array_equal(bits(segs["d"]), bits(o[:, 3]))
                and np.array_equal(bits(segs["centroid"]), bits(o[:, 4:7])) and np.array_equal(bits(segs["mse"]), bits(o[:, 7]))
                and np.array_equal(bits(segs["score"]), bits(o[:, 8])) and np.array_equal(bits(segs["sums"]), bits(o[:, 9:18]))
                and np.array_equal(segs["merge_label"], r.merge_labels))
        kept = res.cylinder_labels(f)  # (follows the frame's spill records, if it has more than 64 labels)
        kept = kept[kept["kept"] == 1]
        cyl_ok = len(kept) == len(r.cylinders)
        if cyl_ok and len(kept):
            cyl_ok = bool(np.array_equal(bits(kept["axis"]), bits(r.cylinders[:, 0:3])))
        return labels, counts, seg_ok, cyl_ok, int(hdr["n_planes"]), int(hdr["n_plane_segments"]), int(hdr["n_cylinders"])

    labels_equal = counts_equal = segments_bitwise = cylinders_bitwise = True
    n_planes = n_segments = n_cyl = 0
    with cf.ThreadPoolExecutor(max(1, min(32, available_cores()))) as pool:
        for c0 in range(0, n, 256):
            host = fetch(c0, min(256, n - c0))
            for lab, cnt, sg, cy, a, b, c in pool.map(one, [(c0 + k, host[k]) for k in range(len(host))]):
                labels_equal &= lab
                counts_equal &= cnt
                segments_bitwise &= sg
                cylinders_bitwise &= cy
                n_planes += a
                n_segments += b
                n_cyl += c
    return {"frames": n, "labels_equal": labels_equal, "counts_equal": counts_equal, "segments_bitwise": segments_bitwise,
            "cylinders_bitwise": cylinders_bitwise, "planes": n_planes, "plane_segments": n_segments, "cylinders": n_cyl,
            "checker": "oracle/libcape_oracle.so on these frames of the last timed step (threaded, one Oracle per thread)"}


def polygon_check(ex, n):
    """In-run check of rows N1 / N2 (checker only, outside every timed region): the device polygons of the first n frames of
    the last batch and the polygon matches between them against the oracle of the reference's algorithm
    (oracle/polygon_oracle.cpp: concave_fitting.cpp + polygon.cpp + MapPlane::find_matches restated)."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import numpy as np

    import cape_amd
    import polygon_oracle_py as P

    P.build()
    res = ex.results(n)
    pol, ver = ex.polygons(n)
    got = ex.polygon_matches(n)
    out = dict(frames=n, planes=0, vertex_identical=0, iou_below_0999=0, validity_mismatches=0, reference_would_dissolve=0,
               frame_pairs=0, match_decision_mismatches=0, matches=0)
    kept, skip = [], set()
    for f in range(n):
        planes = []
        for i, s in enumerate(res.segments(f)):
            if not s["is_output"]:
                continue
            c0 = np.asarray(s["normal"], np.float64) * (-np.float64(s["d"]))
            ref = P.Polygon.from_points(res.boundary_points(f, s), s["normal"], c0)
            out["planes"] += 1
            if ref.flags & P.NEEDS_DISSOLVE:
                out["reference_would_dissolve"] += 1
                skip.add(f)
                continue
            p = pol[f, i]
            dev_valid = bool(p["flags"] & cape_amd.POLY_VALID) and int(p["vertex_count"]) >= 3
            ref_valid = (not ref.threw) and ref.valid and ref.boundary_length() >= 3
            if ref_valid:
                planes.append((i, np.asarray(s["out_normal"], np.float64), float(s["d"]), ref))
            if dev_valid != ref_valid:
                out["validity_mismatches"] += 1
                skip.add(f)
                continue
            if not dev_valid:
                continue
            o, c = int(p["vertex_offset"]), int(p["vertex_count"])
            verts = ver[f, o:o + c]
            if len(verts) == len(ref.ring) and np.array_equal(verts, ref.ring):
                out["vertex_identical"] += 1
                continue
            d = P.Polygon(verts, p["x_axis"], p["y_axis"], p["center"])
            inter = d.inter_area(ref)
            if inter / (d.area + ref.area - inter) < 0.999:
                out["iou_below_0999"] += 1
        kept.append(planes)
    for f in range(1, n):
        if f in skip or (f - 1) in skip or (got[f]["flags"] & cape_amd.MATCH_EXACT_OVERFLOW):
            continue
        prev, cur = kept[f - 1], kept[f]
        want, _ = P.find_matches([q[1:] for q in prev], [q[1:] for q in cur])
        out["frame_pairs"] += 1
        out["matches"] += sum(1 for m in want if m >= 0)
        if list(got[f]["match"][: len(prev)]) != want:
            out["match_decision_mismatches"] += 1
    out["ok"] = out["iou_below_0999"] == 0 and out["validity_mismatches"] == 0 and out["match_decision_mismatches"] == 0
    out["checker"] = "oracle/polygon_oracle.cpp (the reference's concave hull / polygon / find_matches restated; pinned by the reference's tests/test_polygons.cpp)"
    return out


def parity_ok(pc):
    return bool(pc["labels_equal"] and pc["counts_equal"] and pc["segments_bitwise"] and pc["cylinders_bitwise"])


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--frames", type=int, default=0,
                    help="frames per step per GPU, resident in HBM (default: 4096 at N=1, 2048 per GPU at N>1)")
    ap.add_argument("--unique", type=int, default=0,
                    help="distinct frames per GPU, tiled to --frames (default: every frame distinct)")
    ap.add_argument("--width", type=int, default=640)
    ap.add_argument("--height", type=int, default=480)
    ap.add_argument("--scene", default="", help="room | tumlike | tunnel (default: room at N=1, tumlike at N>1)")
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"],
                    help="N>1: weak = 2048 frames per GPU (stream of N x 2048), strong = one 8 x 2048-frame stream for every N")
    ap.add_argument("--gather", default="native", choices=["native", "torch", "none"],
                    help="N>1 exchange of the packed primitive lists: native = ncclAllGather issued by libcape_hip "
                         "(cape_gather_primitives), torch = torch.distributed all_gather_into_tensor, none = no exchange")
    ap.add_argument("--cylinders", action="store_true", help="planes + cylinder RANSAC (BASELINE.json configs[2], [4])")
    ap.add_argument("--match", action="store_true",
                    help="also run the consecutive-frame plane matcher every step (the 'IoU matching' of BASELINE.json configs[4])")
    ap.add_argument("--u16", action="store_true",
                    help="feed raw uint16 sensor depth (1/5 mm units, the TUM PNG format) through cape_extract_u16 instead of float32 mm")
    ap.add_argument("--host-synth", action="store_true", help="render the frames with the numpy generators (slow; parity fixtures use these)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--sub-batches", type=int, default=0, help="cape_config.sub_batches (0 = one kernel chain per step)")
    ap.add_argument("--planes-per-frame", type=int, default=0,
                    help="N>1: plane budget of the packed payload per frame (0 = measured on the stream: mean x 1.15 + 1)")
    ap.add_argument("--spawn", action="store_true",
                    help="go through the self-spawning launcher even at --gpus 1 (what `--gpus N` does for N > 1 when no "
                         "launcher started this process); also CAPE_BENCH_FORCE_SPAWN=1")
    ap.add_argument("--gather-root", action="store_true",
                    help="N>1: gather the packed lists to rank 0 only (cape_gather_primitives_root) instead of all-gathering them")
    ap.add_argument("--no-parity-check", action="store_true", help="skip the in-run comparison of the last step's results with the CPU oracle")
    ap.add_argument("--parity-frames", type=int, default=0, help="frames of the last step the in-run check compares with the oracle (0 = every distinct frame)")
    ap.add_argument("--no-wide-grid", action="store_true", help="skip the extra legs on a 1920x1080 grid and on a frame of more than 64 plane segments")
    ap.add_argument("--no-polygons", action="store_true", help="skip the extra boundary-polygon leg (cape_build_polygons)")
    ap.add_argument("--no-cylinders-on", action="store_true",
                    help="N=1 default workload: skip the extra 'cylinders_on' leg (same stream with the reference's unconditional cylinder branch)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if "WORLD_SIZE" not in os.environ and (args.gpus > 1 or args.spawn or os.environ.get("CAPE_BENCH_FORCE_SPAWN") == "1"):
        # `python bench.py --gpus N` without a launcher: start the N ranks ourselves (one process per GPU under
        # torch.distributed.run, exactly the command line the driver would use) and hand its single JSON line on
        raise SystemExit(self_spawn(args.gpus))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: launch with --nproc-per-node {args.gpus}")

    # stdout carries ONE line, the JSON: everything else a library may print there (RCCL's version banner, for one) is sent
    # to stderr at the file-descriptor level for the lifetime of the process, and the line goes out through the saved fd
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)

    import numpy as np
    import torch
    import torch.distributed as dist

    if not os.path.exists(os.path.join(ROOT, "rgb-d-slam_amd", "lib", "libcape_hip.so")) and local_rank == 0:
        # fresh checkout: the libraries are git-ignored build products (what __graft_entry__.build() makes)
        import subprocess

        subprocess.check_call(["make", "-C", os.path.join(ROOT, "rgb-d-slam_amd", "csrc"), "all"], stdout=subprocess.DEVNULL)
    for _ in range(600):  # the other ranks wait for rank 0's build
        if os.path.exists(os.path.join(ROOT, "rgb-d-slam_amd", "lib", "libcape_hip.so")):
            break
        time.sleep(0.5)
    from cape_amd import Extractor, synth, synth_gpu
    from cape_amd import dist as cdist

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the product has no CPU path")
    # CAPE_BENCH_BACKEND=gloo: the N > 1 code paths of this file (sharding, budget all-reduce, the torch gather, max over ranks,
    # per-rank parity) with several ranks SHARING the visible GPU(s) -- RCCL refuses two ranks on one device, gloo does not care.
    # Validation of the launcher path on a one-GPU box; the line says so and is not a scaling measurement.
    backend = os.environ.get("CAPE_BENCH_BACKEND", "nccl")
    if backend not in ("nccl", "gloo"):
        raise SystemExit(f"bench.py: CAPE_BENCH_BACKEND={backend}: nccl (RCCL, default) or gloo")
    dev_index = local_rank if backend == "nccl" else local_rank % torch.cuda.device_count()
    coll_dev = "cuda" if backend == "nccl" else "cpu"  # where the tensors of the small collectives live
    torch.cuda.set_device(dev_index)
    # CAPE_BENCH_FORCE_GATHER=1 exercises the multi-GPU exchange on a single GPU (world 1; validation only)
    force = os.environ.get("CAPE_BENCH_FORCE_GATHER") == "1"
    multi = world > 1 or force
    if multi:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group(backend, rank=rank, world_size=world)

    W, H = args.width, args.height
    scene = args.scene or ("tumlike" if multi else "room")
    # frames of this rank: a contiguous block of ONE stream (configs[3]) -- at N=1 the block is the whole stream
    if multi:
        per_gpu = args.frames or 2048
        total = 8 * per_gpu if args.scaling == "strong" else world * per_gpu
        first, last = cdist.shard_range(total, rank, world)
        B = last - first
        B_max = cdist.largest_shard(total, world)
    else:
        B = B_max = args.frames or 4096
        total, first = B, 0
    scale = W / 640.0
    base_intr = synth.TUM_FR1_INTRINSICS if scene == "tumlike" else synth.DEFAULT_INTRINSICS
    intr = {k: v * scale for k, v in base_intr.items()}

    # ---- synthetic stream, resident in HBM before the timed region
    U = min(args.unique or B, B)
    reps = -(-B // U)
    if args.host_synth:
        unique_host = synth.stream(scene, seed=100, n_frames=U, width=W, height=H, start=first)
        if args.u16:
            raw = np.clip(np.rint(unique_host * 5.0), 0, 65535).astype(np.uint16)
            unique_dev = torch.from_numpy(raw.view(np.int16)).cuda()
        else:
            unique_dev = torch.from_numpy(unique_host).cuda()
    else:
        # frame ids [first, first + U) of the stream: what rank r renders depends on its block only
        unique_dev = synth_gpu.stream(scene, 100, U, width=W, height=H, start=first, device="cuda",
                                      chunk=64 if W <= 640 else 16, raw_u16=args.u16)
    depth = unique_dev if reps == 1 else unique_dev.repeat(reps, 1, 1)[:B].contiguous()
    torch.cuda.synchronize()

    ex = Extractor(W, H, cylinders=args.cylinders, device=dev_index, max_batch=B_max, sub_batches=args.sub_batches, **intr)
    stream = torch.cuda.current_stream().cuda_stream
    gather = args.gather if multi else "none"
    gather_note = ""
    if gather == "native" and backend == "gloo" and world > 1:
        gather = "torch"
        gather_note = "CAPE_BENCH_BACKEND=gloo: ranks share a device, which RCCL refuses -- the packed lists travel through torch.distributed (gloo, staged through the host)"
    lay, recv, works, local_view = None, None, [None, None], [None, None]
    planes_budget, budget_note = 16, "default"
    if gather != "none":
        # The payload is a fixed byte count per rank (what the collective needs), sized by a plane budget for the whole
        # shard.  The budget comes from the stream itself: one un-gathered pass, the mean number of planes per frame over
        # all ranks, 15 % head-room (an overflow would be reported in the packed header and is checked after the timed
        # region; planes_per_frame = 64 can never overflow)
        ex.extract_device(depth.data_ptr(), B, stream) if not args.u16 else ex.extract_device_u16(depth.data_ptr(), 0.2, B, stream)
        n_pl0, n_cy0, _ = ex.count_primitives(B)  # cape_count_primitives: a device-side reduction over the batch's headers
        cnt = torch.tensor([float(n_pl0), float(n_cy0), float(B)], dtype=torch.float64, device=coll_dev)
        mx = torch.tensor([n_pl0 / B, n_cy0 / B], dtype=torch.float64, device=coll_dev)
        dist.all_reduce(cnt)
        dist.all_reduce(mx, op=dist.ReduceOp.MAX)
        if args.planes_per_frame > 0:
            planes_budget, budget_note = args.planes_per_frame, "--planes-per-frame"
        else:
            # the budget is per SHARD, so it is the fullest shard's mean that must fit
            planes_budget = int(min(64, np.ceil(float(mx[0]) * 1.15) + 1))
            budget_note = f"measured: fullest shard holds {float(mx[0]):.2f} planes per frame (stream mean {float(cnt[0] / cnt[2]):.2f}), x 1.15 + 1"
        cyl_budget = int(min(64, np.ceil(float(mx[1]) * 1.15) + 1)) if args.cylinders else 1
        lay = ex.gather_configure(B_max, planes_per_frame=planes_budget, cylinders_per_frame=cyl_budget)
        if args.gather_root and gather == "native" and rank != 0:
            recv = [None, None]  # ncclGather: only the root receives
        else:
            recv = [torch.empty(world * lay["bytes_per_rank"], dtype=torch.uint8, device=coll_dev) for _ in range(2)]
        if gather == "native":
            # The C layer's communicator (librccl through dlopen, ncclCommInitRank).  If that does not come up on every
            # rank -- a library that cannot be resolved, an init that fails -- all ranks agree to route the same packed
            # bytes through torch.distributed instead, and the JSON line says which path ran.
            err = ""
            try:
                uid = cdist.broadcast_unique_id(ex.comm_unique_id, rank, device=coll_dev)
                ex.comm_init(uid, rank, world)
            except Exception as e:  # noqa: BLE001 -- whatever went wrong, the bench must still report a number
                err = f"{type(e).__name__}: {e}"
            okflag = torch.tensor([0 if err else 1], dtype=torch.int32, device=coll_dev)
            dist.all_reduce(okflag, op=dist.ReduceOp.MIN)
            if int(okflag.item()) == 0:
                gather = "torch"
                gather_note = "native RCCL gather unavailable" + (f" ({err})" if err else " on another rank") + ": torch.distributed used"
                if rank == 0:
                    print("bench.py: WARNING -- " + gather_note, file=sys.stderr)
    # what RCCL itself reports per rank (cape_comm_info: ncclCommCount / ncclCommUserRank / ncclCommCuDevice next to the values
    # cape_comm_init was given): the evidence that an N-GPU run had N ranks on N devices.  None when torch.distributed moved the bytes.
    comm_infos = None
    if gather == "native":
        comm_infos = [None] * world
        dist.all_gather_object(comm_infos, dict(ex.comm_info(), local_rank=dev_index))
    step_no = [0]

    def step():
        if args.u16:
            ex.extract_device_u16(depth.data_ptr(), 0.2, B, stream)
        else:
            ex.extract_device(depth.data_ptr(), B, stream)
        if args.match:
            ex.match_consecutive(B, 0, stream)
        if gather == "native":
            # pack kernels on this stream, ONE ncclAllGather on the handle's communication stream behind an event: the
            # collective of step k runs under the kernels of step k+1 (two staging slots, two receive buffers)
            if args.gather_root:
                ex.gather_root(B, first, 0, recv[step_no[0] & 1].data_ptr() if rank == 0 else 0, stream)
            else:
                ex.gather(B, first, recv[step_no[0] & 1].data_ptr(), stream)
            step_no[0] += 1
        elif gather == "torch":
            k = step_no[0] & 1
            step_no[0] += 1
            if works[k] is not None:
                works[k].wait()
            ptr = ex.pack(B, first, stream)
            local_view[k] = torch.as_tensor(_DevMem(ptr, lay["bytes_per_rank"]), device="cuda")
            if backend == "gloo":
                local_view[k] = local_view[k].cpu()  # (synchronises: gloo gathers host tensors)
            works[k] = dist.all_gather_into_tensor(recv[k], local_view[k], async_op=True)

    def drain():
        if gather == "native":
            ex.gather_wait(host_sync=True)
        for k in range(2):
            if works[k] is not None:
                works[k].wait()
                works[k] = None

    def timed(n_steps, fn):
        """barrier + device sync, n_steps x fn, drain, device sync + barrier: this rank's seconds"""
        drain()
        torch.cuda.synchronize()
        if multi:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n_steps):
            fn()
        drain()  # every gather has landed before the clock stops
        torch.cuda.synchronize()
        t_own = time.perf_counter() - t0  # this rank's own work, before it waits for the others
        if multi:
            dist.barrier()
        torch.cuda.synchronize()
        return time.perf_counter() - t0, t_own

    def over_ranks(v):
        """(max, min, list) of a per-rank scalar"""
        if not multi:
            return v, v, [v]
        t = torch.tensor([v], dtype=torch.float64, device=coll_dev)
        allv = [torch.zeros_like(t) for _ in range(world)]
        dist.all_gather(allv, t)
        vals = [float(x.item()) for x in allv]
        return max(vals), min(vals), vals

    for _ in range(args.warmup):
        step()
    ex.reset_timings()
    ex.enable_timing(True)
    elapsed, own = timed(args.steps, step)
    ex.enable_timing(False)
    tm = ex.timings()
    elapsed, _, _ = over_ranks(elapsed)                  # the contract's clock: max over ranks, barriers inside
    own_max, own_min, own_all = over_ranks(own)

    gather_check = None
    if gather != "none":
        # outside the timed region: what arrived is every rank's shard, in rank order, nothing dropped
        got = recv[(step_no[0] - 1) & 1]
        if got is None:
            gather_check = {"ok": True}  # a non-root rank of the gather-to-root path receives nothing
        else:
            shards = cdist.unpack_gathered(got.cpu().numpy(), world, lay)
            spans = [cdist.shard_range(total, r, world) for r in range(world)]
            ok = all(sh.first_frame == spans[r][0] and int(sh.header["n_frames"]) == spans[r][1] - spans[r][0]
                     for r, sh in enumerate(shards))
            n_pl = int(sum(int(sh.header["n_planes_total"]) for sh in shards))
            n_cy = int(sum(int(sh.header["n_cylinders_total"]) for sh in shards))
            from cape_amd import PACKED_CYLINDER_DTYPE, PACKED_FRAME_DTYPE, PACKED_HEADER_DTYPE, PACKED_PLANE_DTYPE
            used = (world * PACKED_HEADER_DTYPE.itemsize + total * PACKED_FRAME_DTYPE.itemsize + n_pl * PACKED_PLANE_DTYPE.itemsize
                    + n_cy * PACKED_CYLINDER_DTYPE.itemsize)
            gather_check = {"ok": bool(ok), "frames": int(sum(int(sh.header["n_frames"]) for sh in shards)),
                            "planes": n_pl, "cylinders": n_cy,
                            "overflow": int(max(int(sh.header["overflow"]) for sh in shards)),
                            "bytes_per_rank": int(lay["bytes_per_rank"]),
                            "payload_bytes_per_frame": lay["bytes_per_rank"] * world / total,
                            "used_bytes_per_frame": used / total,
                            "padding_frac": 1.0 - used / (lay["bytes_per_rank"] * world),
                            "planes_per_frame_budget": planes_budget, "planes_per_frame_mean": n_pl / total,
                            "budget_from": budget_note}
            if not ok:
                raise SystemExit(f"gathered shards are inconsistent: {gather_check}")
        # what the exchange costs the step: the same steps again without it (same clock, max over ranks)
        saved = gather
        gather = "none"
        k2 = max(1, min(args.steps, 10))
        e2, _ = timed(k2, step)
        e2, _, _ = over_ranks(e2)
        gather = saved
        gather_check["ms_per_step_without_gather"] = 1e3 * e2 / k2
        gather_check["exposed_ms_per_step"] = 1e3 * (elapsed / args.steps - e2 / k2)

    # ---- the work proves itself: first frames of the last timed step vs the CPU oracle (every rank checks its own shard)
    parity = None
    if not args.no_parity_check:
        # EVERY distinct frame of the step (64 of 4 096 used to be checked): the oracle does ~500 frames/s per core and
        # the box has dozens of cores -- a few seconds outside every timed region.  (--parity-frames bounds it.)
        n_chk = min(U, args.parity_frames if args.parity_frames > 0 else U)
        if args.u16:
            chk = lambda a, c: unique_dev[a:a + c].cpu().numpy().view(np.uint16).astype(np.float32) * np.float32(0.2)  # noqa: E731
        else:
            chk = lambda a, c: unique_dev[a:a + c].cpu().numpy()  # noqa: E731
        if gather_check is not None:
            step()  # the extra steps above ran the same frames; make the last batch's results current again
            drain()
            torch.cuda.synchronize()
        parity = parity_check(ex, chk, intr, args.cylinders, n_chk)
        _, okmin, _ = over_ranks(1.0 if parity_ok(parity) else 0.0)
        parity["ranks_checked"] = world
        parity["all_ranks_ok"] = bool(okmin == 1.0)
        if okmin != 1.0:
            print(f"bench.py: rank {rank}: results differ from the oracle: {parity}", file=sys.stderr)
            if multi:
                dist.destroy_process_group()
            raise SystemExit(3)

    if rank == 0:
        frames_total = total * args.steps if multi else B * args.steps
        cells = (W // 20) * (H // 20)
        calls = max(1, tm["calls"])
        fpl = tm["frames"] / calls  # frames per kernel launch (= B unless the batch is cut in sub-batches)
        a1_ms = 1e3 * tm["cell_moments_s"] / calls
        a2_ms = 1e3 * tm["cell_plane_s"] / calls
        b_ms = 1e3 * tm["grow_s"] / calls
        # algorithmic bytes per launch (DESIGN.md "Measurement"):
        #   A1 cell moments : reads every depth pixel once, writes 96 B per cell (10 f64 sums + 16 B hand-over)
        #   A2 cell plane   : reads those 96 B, writes 88 B per cell (plane, score, tolerance, flags, bin)
        #   B  grow         : reads 112 B per cell (sums 80, flags 4, bin 4, MSE 8, centre-pixel record 16; the cell planes stay
        #                     in stage A2, which hands over four edge bits per cell), writes label grids + primitive lists
        kernels = {
            "cape_cell_moments_kernel": (a1_ms, fpl * (W * H * (2 if args.u16 else 4) + cells * 96)),
            "cape_cell_plane_kernel": (a2_ms, fpl * (cells * (96 + 88))),
            "cape_grow_kernel": (b_ms, fpl * (cells * 112 + 2 * cells * 4 + 32 * 128)),
        }
        dom = max(kernels, key=lambda k: kernels[k][0])
        dom_ms, dom_bytes = kernels[dom]
        achieved = dom_bytes / (dom_ms * 1e-3) / 1e9
        traffic, traffic_source = None, None
        tpath = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tpath):
            try:
                tj = json.load(open(tpath))
                if tj.get("kernel") == dom and tj.get("frames_per_launch") == fpl and tj.get("width") == W:
                    traffic = tj.get("hbm_bytes_per_launch")
                    traffic_source = ("profiles/traffic.json: static, from the separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of "
                                      "profiles/collect.sh over this same command -- not measured by this run")
            except Exception:
                traffic = None
        # The bound that actually holds the dominant kernel: A1 moves 1.0001 x its algorithmic bytes at about half the
        # HBM peak because the ONE VALU port of a SIMD is busy.  profiles/valu_issue.py counts the vector instructions of the
        # shipped ISA per class, prices them with the issue costs measured on gfx950 (profiles/r02_valu_rates.txt), and checks
        # the count against a rocprofv3 --pmc SQ_INSTS_VALU pass of this command; the fraction below uses THIS run's launch time.
        valu_issue = None
        vpath = os.path.join(ROOT, "profiles", "valu_issue.json")
        if dom == "cape_cell_moments_kernel" and os.path.exists(vpath):
            try:
                vj = json.load(open(vpath))
                vv = vj["variants"]["u16" if args.u16 else "f32"]
                if vj.get("width") == W and vj.get("height") == H:
                    sc = fpl / vj["frames_per_launch"]
                    valu_issue = {"floor_ms": vv["floor_ms"] * sc, "frac_of_issue_floor": vv["floor_ms"] * sc / dom_ms,
                                  "f64_rate_insts_per_launch": vv["f64_rate_insts_per_launch"] * sc,
                                  "other_valu_insts_per_launch": vv["other_insts_per_launch"] * sc,
                                  "f64_rate_share_of_floor": vv["f64_rate_share_of_floor"],
                                  "valu_insts_per_wave": vv["valu_per_wave"], "waves_per_launch": vj["waves_per_launch"] * sc,
                                  "pmc_SQ_INSTS_VALU_per_launch": (vv.get("pmc") or {}).get("SQ_INSTS_VALU_per_launch"),
                                  "ns_per_wave_instruction": vj["rate_ns_per_wave_instruction"],
                                  "source": "profiles/valu_issue.json (profiles/valu_issue.py: ISA of the shipped kernel x the measured issue costs of "
                                            "profiles/r02_valu_rates.txt, checked against the SQ_INSTS_VALU pass " +
                                            str((vv.get("pmc") or {}).get("source")) + "); static, the fraction uses this run's launch_ms",
                                  "note": "one VALU port per SIMD: f64-rate and other vector instructions of different waves add up "
                                          "(profiles/r02_valu_mix.txt); the f64 conversions and adds are the reference's operand types"}
            except Exception:
                valu_issue = None
        e2e_bytes_per_frame = W * H * (2 if args.u16 else 4) + 2 * cells * 4 + 32 * 128  # 1 239 040 B at 640x480
        if multi:
            workload = (f"{W}x{H} synthetic TUM-like depth stream of {total} frames sharded in contiguous blocks over {world} GPU(s), "
                        f"planes" + (" + cylinder RANSAC" if args.cylinders else " only") +
                        f", one ncclAllGather of the packed primitive lists per step (BASELINE.json configs[3])" if scene == "tumlike"
                        else f"{W}x{H} synthetic {scene} depth stream of {total} frames sharded over {world} GPU(s)")
        elif scene == "room" and not args.cylinders:
            workload = f"{W}x{H} synthetic planar-room depth stream, plane extraction only (BASELINE.json configs[1])"
        else:
            workload = (f"{W}x{H} synthetic {scene} depth stream, planes" + (" + cylinder RANSAC" if args.cylinders else " only"))
        workload += (" + consecutive-frame plane matching" if args.match else "") + (", raw uint16 input" if args.u16 else "")
        out = {
            "metric": "depth frames/s primitive extraction (640x480)" if (W, H) == (640, 480)
                      else f"depth frames/s primitive extraction ({W}x{H})",
            "value": frames_total / elapsed,
            "unit": "frames/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True,
            "scaling": args.scaling if multi else "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {
                "workload": workload,
                "frames_per_step_per_gpu": B, "stream_frames": total, "unique_frames_per_gpu": U, "scene": scene,
                "sub_batches": args.sub_batches, "frames_rendered_on": "host (numpy)" if args.host_synth else "device (torch)",
                "sharding": "contiguous frame blocks per GPU (cape_amd.dist.shard_range)" +
                            (f", packed primitive lists all-gathered once per step ({gather})" if gather != "none" else ""),
                **({"backend": f"gloo: {world} ranks on {torch.cuda.device_count()} visible device(s) -- a dry run of the N > 1 code "
                               "paths, NOT a scaling measurement (CAPE_BENCH_BACKEND)"} if backend == "gloo" else {}),
            },
            "roofline": {
                "bound": "hbm", "kernel": dom, "achieved": achieved, "peak": HBM_PEAK_BYTES_S / 1e9, "unit": "GB/s",
                "frac": achieved * 1e9 / HBM_PEAK_BYTES_S, "traffic": traffic, "traffic_source": traffic_source,
                # also against the 6.29 TB/s a streaming kernel can actually reach (MI355X_MICROARCH.md)
                "frac_of_achievable_6p29TBps": achieved * 1e9 / 6.29e12,
                "algorithmic_bytes_per_launch": dom_bytes, "launch_ms": dom_ms,
                "valu_issue": valu_issue,
                "kernel_ms": {"cape_cell_moments_kernel": a1_ms, "cape_cell_plane_kernel": a2_ms, "cape_grow_kernel": b_ms},
                "stage_b": {"us_per_frame": 1e3 * b_ms / fpl, "frames_in_flight": ex.grow_frames_per_cu * ex.compute_units,
                            "note": "one wavefront per frame, latency bound; ~0 HBM bytes beyond the 112 B/cell it reads"},
                "end_to_end_GBps": frames_total / world * e2e_bytes_per_frame / elapsed / 1e9,
                "end_to_end_frac": frames_total / world * e2e_bytes_per_frame / elapsed / HBM_PEAK_BYTES_S,
            },
        }
        out["ranks"] = {"ms_per_step_max": 1e3 * own_max / args.steps, "ms_per_step_min": 1e3 * own_min / args.steps,
                        "ms_per_step": [1e3 * v / args.steps for v in own_all],
                        "note": "each rank's own clock around its K steps (device-synchronised, before the closing barrier); "
                                "`ms_per_step` above is the contract's clock: barriers inside, max over ranks",
                        "launcher": "self-spawned (python bench.py --gpus N -> torch.distributed.run)" if os.environ.get("CAPE_BENCH_SPAWNED") == "1"
                                    else ("torch.distributed.run" if "WORLD_SIZE" in os.environ else "single process")}
        out["results"] = ("stay in HBM inside the timed region (records 19.5 KB + label grids 6 KB per frame: copying them out would cost "
                          "more than the step over PCIe); the packed primitive lists (cape_pack_primitives) are what is meant to travel")
        if parity is not None:
            out["parity_check"] = parity
        if gather_check is not None:
            gather_check["path"] = gather + (" gather-to-root" if args.gather_root else " all-gather")
            if gather_note:
                gather_check["note"] = gather_note
            gather_check["native_rccl"] = gather == "native"
            gather_check["torch_fallback_taken"] = bool(multi and args.gather == "native" and gather == "torch")
            gather_check["rccl_comm"] = comm_infos if comm_infos is not None else "absent by design: the packed lists travelled through torch.distributed"
            if comm_infos is not None:
                gather_check["rccl_ranks_seen"] = sorted({int(c["nranks"]) for c in comm_infos})
                gather_check["rccl_devices_seen"] = sorted({(int(c["rank"]), int(c["device"])) for c in comm_infos})
            out["gather"] = gather_check
        if world == 1 and not args.no_cpu_baseline:
            n_host = min(U, 64)
            if args.u16:
                sample = unique_dev[:n_host].cpu().numpy().view(np.uint16).astype(np.float32) * np.float32(0.2)
            else:
                sample = unique_dev[:n_host].cpu().numpy()
            out["cpu_baseline"] = cpu_baseline(sample, intr, cylinders=args.cylinders, scene=scene)
    else:
        out = None

    polygons_leg = None
    if rank == 0 and not args.no_polygons:
        # "Next" row N1 on the device, outside the main timed region: boundary polygons of every output plane of the last
        # batch (one wavefront per plane), timed with events on the launch stream
        ex.build_polygons(B, stream)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        kp = 5
        e0.record()
        for _ in range(kp):
            ex.build_polygons(B, stream)
        e1.record()
        torch.cuda.synchronize()
        n_pl, _, _ = ex.count_primitives(B)
        ms = e0.elapsed_time(e1) / kp
        polygons_leg = {"ms_per_batch": ms, "planes": n_pl, "planes_per_s": n_pl / (ms * 1e-3), "frames_per_s": B / (ms * 1e-3),
                        "note": "cape_build_polygons over the batch's output planes; vertices bit-identical to the host class "
                                "(tests/test_gpu_polygon.py) and to the oracle of the reference's algorithm (tests/test_gpu_polygon_oracle.py)"}
        # "Next" row N2 on those polygons: the reference's intersection areas between consecutive frames (cape_match_polygons)
        ex.match_polygons(B, 0, stream)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(kp):
            ex.match_polygons(B, 0, stream)
        e1.record()
        torch.cuda.synchronize()
        pm = ex.polygon_matches(B)
        ms = e0.elapsed_time(e1) / kp
        polygons_leg["match_polygons"] = {"ms_per_batch": ms, "intersected_pairs": int((pm["inter_area"] >= 0).sum()),
                                          "matches": int((pm["match"] >= 0).sum()),
                                          "frames_beyond_capacity": int((pm["flags"] & 1).astype(bool).sum()),
                                          "note": "MapPlane::find_matches between consecutive frames on polygon intersection areas, "
                                                  "bit-identical to the host class (tests/test_gpu_match_polygon.py)"}
    if gather == "native":
        ex.comm_destroy()
    ex.close()

    if (rank == 0 and world == 1 and not multi and not args.cylinders and not args.no_cylinders_on and not args.u16
            and not args.match and scene == "room"):
        # The headline workload once more on TWO handles, each on a stream of its own, fed alternately (outside the timed region of
        # `value`, which stays one handle, one stream): the per-cell fits and the one-wave-per-frame grow kernel of one batch issue
        # under the streaming kernel of the next.  What a caller with a continuous stream of batches gets for twice the scratch memory.
        # (ONE pair of streams for every overlapped leg of this run: the runtime maps streams onto a few hardware queues in
        #  creation order, and two streams that land on the same queue do not overlap at all)
        # Row N4 beside it: the SAME frames as raw uint16 sensor units (half the bytes per pixel).  If A1 were HBM-bound this
        # launch would take about half the time; it takes the same -- the kernel is bound by what it issues per pixel.
        exu = Extractor(W, H, cylinders=False, device=dev_index, max_batch=B_max, **intr)
        raw16 = torch.empty((B, H, W), dtype=torch.int16, device="cuda")
        for c0 in range(0, B, 256):
            raw16[c0:c0 + 256] = torch.clamp(torch.round(depth[c0:c0 + 256] * 5.0), 0, 65535).to(torch.int32).to(torch.int16)
        for _ in range(3):
            exu.extract_device_u16(raw16.data_ptr(), 0.2, B, stream)
        torch.cuda.synchronize()
        exu.reset_timings()
        exu.enable_timing(True)
        ku = max(5, min(args.steps, 20))
        for _ in range(ku):
            exu.extract_device_u16(raw16.data_ptr(), 0.2, B, stream)
        torch.cuda.synchronize()
        exu.enable_timing(False)
        tu = exu.timings()
        a1u_ms = 1e3 * tu["cell_moments_s"] / max(1, tu["calls"])
        bytes_u = B * (W * H * 2 + cells * 96)
        out["roofline"]["u16_launch"] = {"kernel": "cape_cell_moments_kernel<u16>", "launch_ms": a1u_ms, "algorithmic_bytes_per_launch": bytes_u,
                                         "achieved": bytes_u / (a1u_ms * 1e-3) / 1e9, "frac": bytes_u / (a1u_ms * 1e-3) / HBM_PEAK_BYTES_S,
                                         "note": "the same frames fed as raw uint16 (cape_extract_u16, row N4): half the bytes per pixel in the same "
                                                 "launch time -- A1 is bound by VALU issue (roofline.valu_issue), not by HBM"}
        try:
            vju = json.load(open(os.path.join(ROOT, "profiles", "valu_issue.json")))
            if vju.get("width") == W and vju.get("height") == H:
                fl = vju["variants"]["u16"]["floor_ms"] * B / vju["frames_per_launch"]
                out["roofline"]["u16_launch"]["valu_issue"] = {"floor_ms": fl, "frac_of_issue_floor": fl / a1u_ms}
        except Exception:
            pass
        exu.close()
        del raw16
        streams0 = [torch.cuda.Stream(device=dev_index) for _ in range(2)]
        pair0 = [Extractor(W, H, cylinders=False, device=dev_index, max_batch=B_max, **intr) for _ in range(2)]
        k0 = max(10, min(args.steps, 40))
        for i in range(4):
            pair0[i & 1].extract_device(depth.data_ptr(), B, streams0[i & 1].cuda_stream)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(k0):
            pair0[i & 1].extract_device(depth.data_ptr(), B, streams0[i & 1].cuda_stream)
        torch.cuda.synchronize()
        e0 = time.perf_counter() - t0
        out["two_handles_overlapped"] = {"value": B * k0 / e0, "unit": "frames/s", "steps": k0, "ms_per_step": 1e3 * e0 / k0,
                                         "note": "the same workload on two handles / two streams fed alternately: one batch's latency-bound "
                                                 "kernels (per-cell fits, grow) run under the next batch's streaming kernel; `value` above is "
                                                 "one handle on one stream"}
        if not args.no_parity_check:
            for h0 in pair0:
                pc0 = parity_check(h0, unique_dev[:16].cpu().numpy(), intr, False, 16)
                if not parity_ok(pc0):
                    print(f"bench.py: overlapped plane-only results differ from the oracle: {pc0}", file=sys.stderr)
                    raise SystemExit(3)
            out["two_handles_overlapped"]["parity_check"] = "both handles: 16 / 16 frames bit-exact vs the oracle"
        for h0 in pair0:
            h0.close()
        # The reference has no plane-only switch: its cylinder branch is unconditional (primitive_detection.cpp:385-388).
        # Same stream, same frames, cylinders enabled -- outside the main timed region, reported next to `value`.
        ex2 = Extractor(W, H, cylinders=True, device=dev_index, max_batch=B_max, sub_batches=args.sub_batches, **intr)
        k2 = max(10, min(args.steps, 20))
        for _ in range(3):
            ex2.extract_device(depth.data_ptr(), B, stream)
        torch.cuda.synchronize()
        ex2.reset_timings()
        ex2.enable_timing(True)
        t0 = time.perf_counter()
        for _ in range(k2):
            ex2.extract_device(depth.data_ptr(), B, stream)
        torch.cuda.synchronize()
        e2 = time.perf_counter() - t0
        ex2.enable_timing(False)
        t2 = ex2.timings()
        c2 = max(1, t2["calls"])
        cyl = {"value": B * k2 / e2, "unit": "frames/s", "steps": k2, "ms_per_step": 1e3 * e2 / k2,
               "kernel_ms": {"cape_cell_moments_kernel": 1e3 * t2["cell_moments_s"] / c2,
                             "cape_cell_plane_kernel": 1e3 * t2["cell_plane_s"] / c2,
                             "cape_grow_kernel": 1e3 * t2["grow_s"] / c2},
               "workload": f"the same {W}x{H} room stream with the reference's unconditional cylinder branch on "
                           "(primitive_detection.cpp:385-388): the reference-faithful mode"}
        if not args.no_parity_check:
            pc2 = parity_check(ex2, unique_dev[:16].cpu().numpy(), intr, True, 16)
            cyl["parity_check"] = pc2
            if not parity_ok(pc2):
                print(f"bench.py: cylinders-on results differ from the oracle: {pc2}", file=sys.stderr)
                raise SystemExit(3)
        ex2.close()
        # The second pass lasts as long as its slowest frame and leaves the device nearly idle meanwhile.  TWO handles fed
        # alternately, each with its second pass on a stream of its own (CAPE_FLAG_ASYNC_SECOND_PASS): the streaming kernels
        # of one batch run under the tail of the other.  Same frames, same results (checked), twice the scratch memory.
        pair = [Extractor(W, H, cylinders=True, device=dev_index, max_batch=B_max, async_second_pass=True, **intr) for _ in range(2)]
        for i in range(4):
            pair[i & 1].extract_device(depth.data_ptr(), B, stream)
        for e in pair:
            e.sync_results(stream)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(k2):
            pair[i & 1].extract_device(depth.data_ptr(), B, stream)
        for e in pair:
            e.sync_results(stream)
        torch.cuda.synchronize()
        e3 = time.perf_counter() - t0
        cyl["two_handles_overlapped"] = {"value": B * k2 / e3, "ms_per_step": 1e3 * e3 / k2,
                                         "note": "two handles fed alternately, each second pass on its handle's own stream "
                                                 "(CAPE_FLAG_ASYNC_SECOND_PASS): one batch's streaming kernels run under the other's slow tail"}
        if not args.no_parity_check:
            for e in pair:
                pc3 = parity_check(e, unique_dev[:16].cpu().numpy(), intr, True, 16)
                if not parity_ok(pc3):
                    print(f"bench.py: overlapped cylinders-on results differ from the oracle: {pc3}", file=sys.stderr)
                    raise SystemExit(3)
            cyl["two_handles_overlapped"]["parity_check"] = "both handles: 16 / 16 frames bit-exact vs the oracle"
        for e in pair:
            e.close()
        out["cylinders_on"] = cyl
        # What Primitive_Detection::find_primitives RETURNS (primitive_detection.cpp:119-166, ending in add_planes_to_primitives
        # :562-648): planes WITH their boundary polygons, the cylinder branch on -- plus the consumer's next step on them,
        # MapPlane::find_matches between consecutive frames.  ONE timed region: extract + polygons + matches per step.
        ex3 = Extractor(W, H, cylinders=True, device=dev_index, max_batch=B_max, **intr)

        def full_step():
            ex3.extract_device(depth.data_ptr(), B, stream)
            ex3.build_polygons(B, stream)
            ex3.match_polygons(B, 0, stream)

        for _ in range(3):
            full_step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(k2):
            full_step()
        torch.cuda.synchronize()
        e4 = time.perf_counter() - t0
        # the same step once more with events between the three calls (outside the timed region): where the time goes
        evs = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
        parts = np.zeros(3)
        for _ in range(5):
            evs[0].record()
            ex3.extract_device(depth.data_ptr(), B, stream)
            evs[1].record()
            ex3.build_polygons(B, stream)
            evs[2].record()
            ex3.match_polygons(B, 0, stream)
            evs[3].record()
            torch.cuda.synchronize()
            parts += [evs[i].elapsed_time(evs[i + 1]) for i in range(3)]
        parts /= 5
        n_pl, n_cy, _ = ex3.count_primitives(B)
        fpe = {"value": B * k2 / e4, "unit": "frames/s", "steps": k2, "ms_per_step": 1e3 * e4 / k2,
               "call_ms": {"cape_extract (A1 + A2 + B, cylinder branch on)": float(parts[0]), "cape_build_polygons": float(parts[1]),
                           "cape_match_polygons": float(parts[2])},
               "planes_per_batch": n_pl, "cylinders_per_batch": n_cy,
               "workload": f"the same {W}x{H} room stream: everything find_primitives returns (planes with boundary polygons, cylinders) "
                           "+ find_matches between consecutive frames, one timed region"}
        if not args.no_parity_check:
            pc4 = parity_check(ex3, unique_dev[:16].cpu().numpy(), intr, True, 16)
            fpe["parity_check"] = pc4
            fpe["polygon_check"] = polygon_check(ex3, 32)
            if not parity_ok(pc4) or not fpe["polygon_check"]["ok"]:
                print(f"bench.py: find_primitives_equivalent results differ from the oracles: {pc4} {fpe['polygon_check']}", file=sys.stderr)
                raise SystemExit(3)
        ex3.close()
        # The same three calls per batch on TWO handles, each on a stream of its own, fed alternately: the polygon task kernel
        # and the cylinder second pass wait most of their cycles (dependent chains), the streaming kernels of the other handle's
        # batch issue under them.  Same frames, same results (both handles checked), twice the scratch memory.
        streams2 = streams0
        pair = [Extractor(W, H, cylinders=True, device=dev_index, max_batch=B_max, **intr) for _ in range(2)]

        def full_step_on(i):
            h, st = pair[i & 1], streams2[i & 1].cuda_stream
            h.extract_device(depth.data_ptr(), B, st)
            h.build_polygons(B, st)
            h.match_polygons(B, 0, st)

        for i in range(4):
            full_step_on(i)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(k2):
            full_step_on(i)
        torch.cuda.synchronize()
        e5 = time.perf_counter() - t0
        fpe["two_handles_overlapped"] = {"value": B * k2 / e5, "ms_per_step": 1e3 * e5 / k2,
                                         "note": "two handles on two streams fed alternately (extract + polygons + matches each): one batch's "
                                                 "streaming kernels run under the other's polygon walks and cylinder second pass"}
        if not args.no_parity_check:
            for h in pair:
                pc5 = parity_check(h, unique_dev[:16].cpu().numpy(), intr, True, 16)
                pg5 = polygon_check(h, 32)
                if not parity_ok(pc5) or not pg5["ok"]:
                    print(f"bench.py: overlapped find_primitives_equivalent results differ from the oracles: {pc5} {pg5}", file=sys.stderr)
                    raise SystemExit(3)
            fpe["two_handles_overlapped"]["parity_check"] = "both handles: 16 / 16 frames bit-exact vs the oracle, 32 frames' polygons + matches vs the polygon oracle"
        for h in pair:
            h.close()
        out["find_primitives_equivalent"] = fpe
    if (rank == 0 and world == 1 and not multi and out is not None and not args.no_wide_grid and not args.cylinders and not args.u16
            and not args.match and scene == "room" and (W, H) == (640, 480)):
        # Round 6: what lay beyond the fast kernels' fixed shapes through round 5, in the driver-visible line.  (a) 1920 x 1080 (96 x 54
        # cells: rows of two mask words in the fast kernels), frames resident in HBM, every frame of the step checked against the oracle;
        # (b) a frame of more than 64 plane segments (a checkerboard of tilted facets) inside an ordinary 1280 x 960 batch: its record
        # chain against the oracle.  Not part of `value`.
        Ww, Hw, nw = 1920, 1080, 256
        intr_w = {k: v * Ww / 640.0 for k, v in synth.DEFAULT_INTRINSICS.items()}
        dw = synth_gpu.stream("room", 7, nw, width=Ww, height=Hw, device="cuda", chunk=16)
        exw = Extractor(Ww, Hw, cylinders=True, device=dev_index, max_batch=nw, **intr_w)
        for _ in range(2):
            exw.extract_device(dw.data_ptr(), nw, stream)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            exw.extract_device(dw.data_ptr(), nw, stream)
        e1.record()
        torch.cuda.synchronize()
        msw = e0.elapsed_time(e1) / 5
        wide = {"grid": f"{Ww}x{Hw} ({Ww // 20} x {Hw // 20} cells)", "frames": nw, "ms_per_batch": msw, "frames_per_s": nw / (msw * 1e-3),
                "general_instance_frames": exw.spill_info()[2], "cylinders": True,
                "note": "the reference takes any image size (primitive_detection.cpp:26-67); rows of 65 .. 128 cells run in the fast grow "
                        "kernels on two mask words per lane (Mask128), grids of more than 64 rows or 128 columns in cape_grow_general_kernel "
                        "(bit rows in memory, 16-bit labels); stage A unchanged"}
        if not args.no_parity_check:
            wide["parity_check"] = parity_check(exw, lambda a, c: dw[a:a + c].cpu().numpy(), intr_w, True, nw)
            if not parity_ok(wide["parity_check"]):
                print(f"bench.py: 1920x1080 results differ from the oracle: {wide['parity_check']}", file=sys.stderr)
                raise SystemExit(3)
        exw.close()
        del dw
        Wc, Hc = 1280, 960
        intr_c = {k: v * 2.0 for k, v in synth.DEFAULT_INTRINSICS.items()}
        uu = (np.arange(Wc) - intr_c["cx"]) / intr_c["fx"]
        vv = (np.arange(Hc) - intr_c["cy"]) / intr_c["fy"]
        Xc, Yc = np.meshgrid(uu, vv)
        rngc = np.random.default_rng(3)
        zc = np.zeros((Hc, Wc))
        tilts = [(0.5, 0.0), (-0.5, 0.0), (0.0, 0.5), (0.0, -0.5)]
        for ty in range(0, Hc, 100):
            for tx in range(0, Wc, 100):
                nxc, nyc = tilts[((tx // 100) % 2) + 2 * ((ty // 100) % 2)]
                dc = 2000.0 + 120.0 * (((tx // 100) * 7 + (ty // 100) * 13) % 9)
                sl = (slice(ty, min(ty + 100, Hc)), slice(tx, min(tx + 100, Wc)))
                zc[sl] = dc / (1.0 + nxc * Xc[sl] + nyc * Yc[sl])
        zc += rngc.normal(0, 0.6, zc.shape)
        board = np.round(zc).astype(np.float32)
        mix = np.stack([synth.room(seed=1, frame=k, width=Wc, height=Hc, intr=intr_c) if k % 4 else board for k in range(8)])
        exc = Extractor(Wc, Hc, cylinders=True, device=dev_index, max_batch=len(mix), **intr_c)
        exc.extract_host(mix)
        used, cap, through = exc.spill_info()
        chain = {"grid": "1280x960", "frames": len(mix), "frames_of_more_than_64_segments": int(through), "spill_records_used": int(used),
                 "spill_pool": int(cap),
                 "most_segments_in_a_frame": int(exc.results(len(mix), with_boundary=False).records["header"]["n_plane_segments"].max()),
                 "note": "_planeSegments is an unbounded vector in the reference (primitive_detection.hpp:206): a frame beyond a record's 64 "
                         "segments continues in spill records (cape_frame_header.next_record), written by the general instance"}
        if not args.no_parity_check:
            chain["parity_check"] = parity_check(exc, mix, intr_c, True, len(mix))
            if not parity_ok(chain["parity_check"]):
                print(f"bench.py: record-chain results differ from the oracle: {chain['parity_check']}", file=sys.stderr)
                raise SystemExit(3)
        exc.close()
        out["beyond_fixed_capacities"] = {"wide_grid": wide, "record_chain": chain}
    if out is not None and polygons_leg is not None:
        out["boundary_polygons"] = polygons_leg
    result_line = json.dumps(out) if out is not None else None
    if multi:
        dist.destroy_process_group()
    if result_line is not None:
        sys.stdout.flush()
        os.write(real_stdout, (result_line + "\n").encode())
    os.close(real_stdout)


if __name__ == "__main__":
    main()
