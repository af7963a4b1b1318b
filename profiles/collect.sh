#!/bin/bash
# Regenerates the committed evidence of a round on the GPU box (run through gpurun from the repo root):
#   <tag>_bench_n1.json               python bench.py (default workload, configs[1]; 4096 distinct frames)
#   <tag>_kernel_stats.csv            rocprofv3 --kernel-trace --stats of the same command
#   <tag>_pmc_fetch.csv / _pmc_write.csv          separate --pmc passes (FETCH_SIZE / WRITE_SIZE), never mixed with tracing
#   <tag>_pmc_calibration_fetch.csv / _write.csv  the same counters over a known byte count (microbench/stream_read.hip)
#   <tag>_cyl_kernel_stats.csv        configs[2] (tunnel, planes + cylinder RANSAC)
#   <tag>_1280_kernel_stats.csv       configs[4] geometry (1280x960, tunnel + cylinders + consecutive-frame matching)
#   <tag>_u16_kernel_stats.csv        raw uint16 input (row N4)
#   <tag>_room_cyl_kernel_stats.csv   configs[1]'s stream with the reference's unconditional cylinder branch on
#   <tag>_polygon_kernel_stats.csv    the boundary-polygon kernels (row N1 on the device)
#   <tag>_configs3_bench.json         configs[3] on one GPU (sharded TUM-like stream + native RCCL gather, world 1)
# Everything is first written under gpurun_out/ (the only directory that travels back); profiles/make_traffic.py then turns
# the PMC passes into profiles/traffic.json.
set -u
TAG=${1:-r04}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/collect_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
python $R/bench.py > $OUT/${TAG}_bench_n1.json 2> $OUT/bench.err
CAPE_BENCH_FORCE_GATHER=1 python $R/bench.py --gpus 1 --no-cpu-baseline > $OUT/${TAG}_configs3_bench.json 2> $OUT/bench3.err
BARGS="--no-cpu-baseline --steps 60 --warmup 3 --no-cylinders-on --no-polygons --no-parity-check --no-wide-grid"
stats() { # name, bench args...
    local name=$1; shift
    rocprofv3 --kernel-trace --stats -d $OUT/kt_$name -o k -- python $R/bench.py $BARGS "$@" > /dev/null 2> $OUT/kt_$name.err
    python $R/profiles/summarize_rocprof.py stats $(find $OUT/kt_$name -name "*.db" | head -1) | grep -v "at::\|vectorized_elementwise\|Memset\|fillBuffer\|Cijk\|rocblas\|copyBuffer\|elementwise_kernel\|distribution\|reduce_kernel\|CatArray\|index_\|philox\|triu\|arange" > $OUT/${TAG}_${name}kernel_stats.csv
    rm -rf $OUT/kt_$name
}
pmc() { # counter, file suffix, command...
    local ctr=$1 suf=$2; shift 2
    rocprofv3 --pmc $ctr -d $OUT/p_$suf -o k -- "$@" > /dev/null 2> $OUT/p_$suf.err
    python $R/profiles/summarize_rocprof.py pmc $(find $OUT/p_$suf -name "*.db" | head -1) | grep "kernel,counter\|cape::\|stream_\|^\"cape" > $OUT/${TAG}_pmc_$suf.csv
    rm -rf $OUT/p_$suf
}
stats ""
stats cyl_ --scene tunnel --cylinders --frames 2048
stats 1280_ --width 1280 --height 960 --frames 1024 --scene tunnel --cylinders --match
stats u16_ --u16
stats room_cyl_ --cylinders
rocprofv3 --kernel-trace --stats -d $OUT/kt_poly -o k -- python $R/bench.py --no-cpu-baseline --steps 3 --warmup 1 --no-cylinders-on --no-parity-check > /dev/null 2> $OUT/kt_poly.err
python $R/profiles/summarize_rocprof.py stats $(find $OUT/kt_poly -name "*.db" | head -1) | grep "kernel,\|polygon" > $OUT/${TAG}_polygon_kernel_stats.csv; rm -rf $OUT/kt_poly
pmc FETCH_SIZE fetch python $R/bench.py $BARGS
pmc WRITE_SIZE write python $R/bench.py $BARGS
pmc FETCH_SIZE calibration_fetch $R/rgb-d-slam_amd/lib/stream_read.exe
pmc WRITE_SIZE calibration_write $R/rgb-d-slam_amd/lib/stream_read.exe
ls -la $OUT
tail -1 $OUT/${TAG}_bench_n1.json | cut -c1-300
head -8 $OUT/${TAG}_kernel_stats.csv
cat $OUT/${TAG}_pmc_fetch.csv $OUT/${TAG}_pmc_write.csv $OUT/${TAG}_pmc_calibration_fetch.csv $OUT/${TAG}_pmc_calibration_write.csv
head -6 $OUT/${TAG}_cyl_kernel_stats.csv
