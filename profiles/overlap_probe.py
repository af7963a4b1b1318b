#!/usr/bin/env python3
"""why the latency-bound grow kernel of one batch does not hide under the streaming kernel (A1) of the next.

run   : the plane-only headline workload (4 096 room frames) on TWO handles / two streams fed alternately, like bench.py's
        `two_handles_overlapped` leg -- run this under `rocprofv3 --kernel-trace`;
report: from the trace's rocpd database: the resources every kernel was dispatched with (VGPRs, LDS, workgroup size), what that
        allows side by side on one SIMD / CU of gfx950 (512 VGPRs per SIMD lane, 160 KB of LDS per CU), and what the timeline shows --
        how long the grow / per-cell-fit kernels of one stream take when the other stream's A1 is running against when it is not.

usage: overlap_probe.py run [steps=24]          |          overlap_probe.py report <rocpd .db>"""
import os
import sqlite3
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "rgb-d-slam_amd", "python"))


def run(steps):
    import time

    import torch
    from cape_amd import Extractor, synth, synth_gpu

    B = 4096
    depth = synth_gpu.stream("room", 100, B, device="cuda", chunk=64)
    streams = [torch.cuda.Stream() for _ in range(2)]
    pair = [Extractor(640, 480, cylinders=False, max_batch=B, **synth.DEFAULT_INTRINSICS) for _ in range(2)]
    one = Extractor(640, 480, cylinders=False, max_batch=B, **synth.DEFAULT_INTRINSICS)
    st = torch.cuda.current_stream().cuda_stream
    for _ in range(3):
        one.extract_device(depth.data_ptr(), B, st)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        one.extract_device(depth.data_ptr(), B, st)
    torch.cuda.synchronize()
    t1 = time.perf_counter() - t0
    for i in range(4):
        pair[i & 1].extract_device(depth.data_ptr(), B, streams[i & 1].cuda_stream)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        pair[i & 1].extract_device(depth.data_ptr(), B, streams[i & 1].cuda_stream)
    torch.cuda.synchronize()
    t2 = time.perf_counter() - t0
    print(f"one handle {1e3 * t1 / steps:.3f} ms per 4096 frames; two handles alternating {1e3 * t2 / steps:.3f} ms ({100 * (t1 / t2 - 1):+.1f} %)")


def report(db):
    cur = sqlite3.connect(db).cursor()
    rows = cur.execute("select name, start, end, workgroup_x, grid_x, lds_size, vgpr_count, accum_vgpr_count, scratch_size, queue_id from kernels "
                       "where name like '%cape::%' order by start").fetchall()

    def short(n):
        return n.split("(")[0].replace("void ", "").replace("cape::", "")[:44]

    kinds = {}
    for r in rows:
        kinds.setdefault(short(r[0]), r)
    print("kernel                                        wg threads  VGPRs (+acc)  LDS/wg B   waves/SIMD by VGPRs   workgroups/CU by LDS")
    res = {}
    # rocprofv3's arch_vgpr_count column holds HALF the wave64 allocation on gfx950 (44 for A1, 116 for the grow kernel, where the
    # compiler's -Rpass-analysis=kernel-resource-usage says 83 and 229 VGPRs, i.e. 88 and 232 allocated): doubled here
    for k, r in kinds.items():
        vg = 2 * (int(r[6] or 0) + int(r[7] or 0))
        alloc = -(-max(vg, 1) // 8) * 8
        per_simd = 512 // alloc
        lds = int(r[5] or 0)
        per_cu = (160 * 1024) // lds if lds else 99
        res[k] = (alloc, lds, int(r[3]))
        print(f"{k:44s} {int(r[3]):6d}      {2 * int(r[6] or 0):4d} (+{2 * int(r[7] or 0):3d})   {lds:8d}   {per_simd:10d}           {per_cu if lds else '-':>10}")
    a1 = next((k for k in res if "cell_moments" in k), None)
    grow = next((k for k in res if "grow_kernel" in k), None)
    if a1 and grow:
        ra, rg = res[a1], res[grow]
        waves_a = min(512 // ra[0], 8)
        print(f"\nA1 resident: {ra[2] // 64} waves per workgroup, {min((160 * 1024) // max(ra[1], 1), 4)} workgroups per CU by LDS = "
              f"{(ra[2] // 64) * min((160 * 1024) // max(ra[1], 1), 4)} waves per CU = {(ra[2] // 64) * min((160 * 1024) // max(ra[1], 1), 4) / 4:.0f} per SIMD x {ra[0]} VGPRs = "
              f"{(ra[2] // 64) * min((160 * 1024) // max(ra[1], 1), 4) / 4 * ra[0]:.0f} of 512 -> {512 - (ra[2] // 64) * min((160 * 1024) // max(ra[1], 1), 4) / 4 * ra[0]:.0f} free per SIMD; "
              f"a grow wave needs {rg[0]}.  LDS left beside four A1 workgroups: {160 * 1024 - 4 * ra[1]} B; a grow workgroup asks for {rg[1]} B.")
        del waves_a
    # timeline: for every grow / plane kernel, was an A1 of ANOTHER queue running during it?
    a1s = [(r[1], r[2], r[9]) for r in rows if "cell_moments" in r[0]]
    for pat in ("grow_kernel", "cell_plane"):
        alone, under = [], []
        for r in rows:
            if pat not in r[0]:
                continue
            ov = sum(max(0, min(r[2], e) - max(r[1], s)) for s, e, q in a1s if q != r[9])
            (under if ov > 0.5 * (r[2] - r[1]) else alone).append((r[2] - r[1]) / 1e3)
        if alone or under:
            m = lambda v: sum(v) / len(v) if v else float("nan")
            print(f"{pat:12s}: {len(alone):3d} dispatches with the device to themselves, mean {m(alone):7.1f} us; {len(under):3d} under the other stream's A1, mean {m(under):7.1f} us")


if __name__ == "__main__":
    if sys.argv[1] == "run":
        run(int(sys.argv[2]) if len(sys.argv) > 2 else 24)
    else:
        report(sys.argv[2])
