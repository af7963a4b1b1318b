#!/usr/bin/env python3
"""profiles/traffic.json from the PMC passes of profiles/collect.sh: HBM bytes per launch of every cape kernel, with the
FETCH_SIZE / WRITE_SIZE counters scaled by the calibration run over a known byte count (MI355X_MICROARCH.md, HBM section:
FETCH_SIZE reports half the bytes of a wide coalesced read on gfx950).  usage: make_traffic.py <tag> [frames=4096] [width=640]"""
import csv
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
tag = sys.argv[1]
frames = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
width = int(sys.argv[3]) if len(sys.argv) > 3 else 640
height = width * 3 // 4
cells = (width // 20) * (height // 20)


def table(name):
    out = {}
    for row in csv.DictReader(open(os.path.join(HERE, f"{tag}_pmc_{name}.csv"))):
        key = (row["kernel"].split("<")[0].replace("cape::", ""), row["counter"])
        out[key] = out.get(key, 0.0) + float(row["avg_value"])  # template instances of one kernel (32 / 64 segments) add up
    return out


fetch, write = table("fetch"), table("write")
cf, cw = table("calibration_fetch"), table("calibration_write")
# stream_read.exe: stream_read_f4 reads 4 GiB, stream_write_f64 writes 512 MiB (KB counters)
fetch_cal = (4 * 2**30 / 1024) / cf[("stream_read_f4", "FETCH_SIZE")]
write_cal = (512 * 2**20 / 1024) / cw[("stream_write_f64", "WRITE_SIZE")]
alg = {
    "cape_cell_moments_kernel": frames * (width * height * 4 + cells * 96),
    "cape_cell_plane_kernel": frames * cells * (96 + 88),
    "cape_grow_kernel": frames * (cells * 112 + 2 * cells * 4 + 32 * 128),
}
kernels = {}
for k in alg:
    f = fetch.get((k, "FETCH_SIZE"), 0.0) * 1024 * fetch_cal
    w = write.get((k, "WRITE_SIZE"), 0.0) * 1024 * write_cal
    kernels[k] = {"hbm_bytes_per_launch": round(f + w), "fetched": round(f), "written": round(w),
                  "algorithmic_bytes_per_launch": alg[k], "ratio": round((f + w) / alg[k], 4)}
dom = "cape_cell_moments_kernel"
out = {"kernel": dom, "frames_per_launch": frames, "width": width, "hbm_bytes_per_launch": kernels[dom]["hbm_bytes_per_launch"],
       "algorithmic_bytes_per_launch": alg[dom], "fetch_calibration": fetch_cal, "write_calibration": write_cal,
       "kernels": kernels,
       "calibration": "rgb-d-slam_amd/csrc/microbench/stream_read.hip under the same rocprofv3: a 4 GiB float4 streaming read and a "
                      "512 MiB f64 streaming write of known size give the scale of FETCH_SIZE / WRITE_SIZE (gfx950: FETCH_SIZE reports "
                      "half the bytes of a wide coalesced read, MI355X_MICROARCH.md HBM section)",
       "source": f"profiles/{tag}_pmc_fetch.csv, profiles/{tag}_pmc_write.csv (separate --pmc passes), profiles/{tag}_pmc_calibration_*.csv"}
json.dump(out, open(os.path.join(HERE, "traffic.json"), "w"), indent=1)
print(json.dumps(out, indent=1))
