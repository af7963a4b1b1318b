#!/usr/bin/env python3
"""The VALU-issue roofline of A1 (cape_cell_moments_kernel) from the ISA of the binary that ships -- VERDICT r4 item 3.

A1 moves 1.0001 x its algorithmic bytes at ~0.49 of the HBM peak: it is bound by the ONE VALU port of a SIMD, not by memory.
This script makes that number reproducible instead of prose:

  1. compiles csrc/cape_cell_moments.hip to gfx950 assembly with the Makefile's flags (hipcc -S, no GPU needed);
  2. cuts the kernel into its regions -- first (peeled) trip, the rolled trips of the row loop, last (peeled) trip, the tail
     behind the barrier -- and classifies every vector instruction by the issue cost MEASURED on gfx950
     (csrc/microbench/valu_rates.hip, profiles/r02_valu_rates.txt): f64-rate (v_*_f64, conversions from / to f64),
     double-rate (v_mul/add/sub_f32, v_add/sub_u32, v_and_b32, v_mov_b32: ~2.3 cycles), every other vector op (~4.3 cycles);
  3. prices one launch: instructions per wave x waves per SIMD x ns per wave-instruction;
  4. cross-checks the static count against the dynamic one of a rocprofv3 --pmc SQ_INSTS_VALU pass on the same binary
     (profiles/pmc_sq.sh), when its csv is given, and attributes the difference to the tail (whose blocks are executed by
     subsets of the workgroup's waves and cannot be counted statically).

usage: valu_issue.py [--pmc profiles/r05_pmc_sq_insts.csv] [--frames 4096] [--out profiles/valu_issue.json]
bench.py reads the json (like profiles/traffic.json) and puts `roofline.valu_issue` in its line, with
frac_of_issue_floor = floor_ms / the launch time measured by that run."""
import argparse
import csv
import json
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "rgb-d-slam_amd", "csrc")

# ns per wave-instruction per SIMD, profiles/r02_valu_rates.txt (valu_rates.exe on MI355X; 2.4 GHz shader clock)
RATE_NS = {"f64": 1.95, "other": 1.85, "double_rate": 0.95}
DOUBLE_RATE = {"v_mul_f32", "v_add_f32", "v_sub_f32", "v_subrev_f32", "v_add_u32", "v_sub_u32", "v_subrev_u32", "v_and_b32", "v_mov_b32"}


def classify(op):
    base = re.sub(r"_(e32|e64|dpp|sdwa)$", "", op)
    if "f64" in base or base.endswith("_b64") and base.startswith("v_mov"):
        return "f64"
    if base in DOUBLE_RATE:
        return "double_rate"
    return "other"


def kernel_regions(asm_lines, symbol):
    """[(region name, [opcodes])]: prologue (first trip), loop (one rolled trip), epilogue (last trip), tail (behind the barrier)"""
    start = next(i for i, ln in enumerate(asm_lines) if ln.startswith(symbol + ":"))
    end = next(i for i in range(start, len(asm_lines)) if "s_endpgm" in asm_lines[i])
    body = asm_lines[start + 1:end + 1]
    loop_head = next(i for i, ln in enumerate(body) if "Loop Header" in ln)
    loop_label = body[loop_head].split(":")[0]
    loop_end = next(i for i in range(loop_head, len(body)) if re.search(r"s_c?branch\w*\s+" + re.escape(loop_label) + r"\b", body[i]))
    barrier = next(i for i in range(loop_end, len(body)) if "s_barrier" in body[i])

    def ops(lo, hi):
        out = []
        for ln in body[lo:hi]:
            m = re.match(r"\s+([vs]_\w+|ds_\w+|global_\w+|buffer_\w+|flat_\w+)", ln)
            if m:
                out.append(m.group(1))
        return out

    return [("first_trip", ops(0, loop_head)), ("rolled_trip", ops(loop_head, loop_end + 1)), ("last_trip", ops(loop_end + 1, barrier)),
            ("tail", ops(barrier, len(body)))]


def count(ops):
    c = {"f64": 0, "double_rate": 0, "other": 0, "salu": 0, "lds": 0, "vmem": 0}
    for op in ops:
        if op.startswith("v_"):
            c[classify(op)] += 1
        elif op.startswith("s_"):
            c["salu"] += 1
        elif op.startswith("ds_"):
            c["lds"] += 1
        else:
            c["vmem"] += 1
    return c


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--pmc", default="")
    ap.add_argument("--frames", type=int, default=4096)
    ap.add_argument("--width", type=int, default=640)
    ap.add_argument("--height", type=int, default=480)
    ap.add_argument("--simds", type=int, default=1024)
    ap.add_argument("--out", default=os.path.join(ROOT, "profiles", "valu_issue.json"))
    args = ap.parse_args()

    asm_path = "/tmp/cape_cell_moments.gfx950.s"
    flags = "-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -fno-slp-vectorize".split()
    subprocess.check_call(["/opt/rocm/bin/hipcc"] + flags + ["-S", "--cuda-device-only", "-o", asm_path, os.path.join(CSRC, "cape_cell_moments.hip")],
                          stderr=subprocess.DEVNULL)
    asm = open(asm_path).read().splitlines()
    # launch shape (cape_cell_moments.hip): a 320-thread workgroup = 5 waves = two bands of 20 rows x 640 px; ten groups of two
    # rows, two groups per trip, first and last trip peeled
    rows_per_trip, cell = 4, 20
    trips = cell // rows_per_trip
    bands = (args.height // cell) * ((args.width + 639) // 640)
    waves_per_launch = args.frames * ((bands + 1) // 2) * 5
    out = {"kernel": "cape_cell_moments_kernel", "frames_per_launch": args.frames, "width": args.width, "height": args.height,
           "waves_per_launch": waves_per_launch, "simds": args.simds, "rate_ns_per_wave_instruction": RATE_NS,
           "rates_from": "profiles/r02_valu_rates.txt (csrc/microbench/valu_rates.hip on MI355X)", "variants": {}}
    for variant, symbol in (("f32", "_ZN4cape24cape_cell_moments_kernelILb0EEEvNS_12StageAParamsE"),
                            ("u16", "_ZN4cape24cape_cell_moments_kernelILb1EEEvNS_12StageAParamsE")):
        regs = dict(kernel_regions(asm, symbol))
        c = {k: count(v) for k, v in regs.items()}
        weights = {"first_trip": 1, "rolled_trip": trips - 2, "last_trip": 1}
        per_wave = {k: sum(c[r][k] * w for r, w in weights.items()) for k in ("f64", "double_rate", "other", "salu", "lds", "vmem")}
        tail_static = sum(c["tail"][k] for k in ("f64", "double_rate", "other"))
        v = {"static_per_wave": {"row_loop": per_wave, "tail_listed": c["tail"], "per_16_pixels_rolled_trip": c["rolled_trip"]}}
        loop_valu = per_wave["f64"] + per_wave["double_rate"] + per_wave["other"]
        tail_dyn = None
        if args.pmc:
            for row in csv.DictReader(open(args.pmc)):
                if ("cell_moments_kernel<false>" if variant == "f32" else "cell_moments_kernel<true>") in row["kernel"] and row["counter"] == "SQ_INSTS_VALU":
                    dyn = float(row["avg_value"])
                    v["pmc"] = {"SQ_INSTS_VALU_per_launch": dyn, "per_wave": dyn / waves_per_launch, "source": os.path.relpath(args.pmc, ROOT)}
                    tail_dyn = dyn / waves_per_launch - loop_valu
        if tail_dyn is None:
            tail_dyn = 0.5 * tail_static  # no counter pass given: half of the listed tail (its blocks run on subsets of the waves)
            v["tail_note"] = "no PMC pass given: the tail is priced at half its listed instructions"
        else:
            v["tail_note"] = ("tail = SQ_INSTS_VALU per wave minus the row loop's static count: its blocks (partial-sum reduce on wave 4, "
                              "continuity scans on waves 0-3) run on subsets of the workgroup's waves")
        # the tail's instructions are priced at the mix of its listing
        mix = {k: c["tail"][k] / max(1, tail_static) for k in ("f64", "double_rate", "other")}
        per_wave_ns = sum(per_wave[k] * RATE_NS[k] for k in RATE_NS) + tail_dyn * sum(mix[k] * RATE_NS[k] for k in RATE_NS)
        floor_ms = per_wave_ns * waves_per_launch / args.simds * 1e-6
        f64_ns = per_wave["f64"] * RATE_NS["f64"]
        v.update({"valu_per_wave": loop_valu + tail_dyn, "tail_valu_per_wave": tail_dyn, "issue_ns_per_wave": per_wave_ns, "floor_ms": floor_ms,
                  "f64_rate_share_of_floor": f64_ns / per_wave_ns,
                  "f64_rate_insts_per_launch": per_wave["f64"] * waves_per_launch,
                  "other_insts_per_launch": (per_wave["double_rate"] + per_wave["other"] + tail_dyn) * waves_per_launch})
        out["variants"][variant] = v
        print(f"{variant}: row loop per wave f64-rate {per_wave['f64']}, double-rate {per_wave['double_rate']}, other {per_wave['other']}; "
              f"tail {tail_dyn:.0f} (listed {tail_static}); issue floor {floor_ms:.3f} ms per {args.frames} frames "
              f"({100 * f64_ns / per_wave_ns:.0f} % of it the f64-rate instructions the reference's operand types dictate)", file=sys.stderr)
    json.dump(out, open(args.out, "w"), indent=1)
    print(args.out)


if __name__ == "__main__":
    main()
