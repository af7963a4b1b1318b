#!/bin/bash
# Regenerates the committed evidence of a round on the GPU box (run through gpurun from the repo root):
#   profiles/<tag>_bench_n1.json            python bench.py  (default workload, configs[1])
#   profiles/<tag>_kernel_stats.csv         rocprofv3 --kernel-trace --stats of the same command
#   profiles/<tag>_pmc_fetch.csv / _write   separate --pmc passes (FETCH_SIZE / WRITE_SIZE), never mixed with tracing
#   profiles/<tag>_cyl_kernel_stats.csv     same for configs[2] (tunnel, planes + cylinder RANSAC)
# Everything is first written under gpurun_out/ (the only directory that travels back) and copied by the caller.
set -u
TAG=${1:-r01}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/collect_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
python $R/bench.py > $OUT/${TAG}_bench_n1.json 2> $OUT/bench.err
BARGS="--no-cpu-baseline --steps 20 --warmup 3"
rocprofv3 --kernel-trace --stats -d $OUT/kt -o k -- python $R/bench.py $BARGS > /dev/null 2> $OUT/kt.err
python $R/profiles/summarize_rocprof.py stats $(find $OUT/kt -name "*.db" | head -1) > $OUT/${TAG}_kernel_stats.csv
rocprofv3 --pmc FETCH_SIZE -d $OUT/pf -o k -- python $R/bench.py $BARGS > /dev/null 2> $OUT/pf.err
python $R/profiles/summarize_rocprof.py pmc $(find $OUT/pf -name "*.db" | head -1) > $OUT/${TAG}_pmc_fetch.csv
rocprofv3 --pmc WRITE_SIZE -d $OUT/pw -o k -- python $R/bench.py $BARGS > /dev/null 2> $OUT/pw.err
python $R/profiles/summarize_rocprof.py pmc $(find $OUT/pw -name "*.db" | head -1) > $OUT/${TAG}_pmc_write.csv
CARGS="--no-cpu-baseline --steps 20 --warmup 3 --scene tunnel --cylinders --frames 2048"
rocprofv3 --kernel-trace --stats -d $OUT/ktc -o k -- python $R/bench.py $CARGS > /dev/null 2> $OUT/ktc.err
python $R/profiles/summarize_rocprof.py stats $(find $OUT/ktc -name "*.db" | head -1) > $OUT/${TAG}_cyl_kernel_stats.csv
rm -rf $OUT/kt $OUT/pf $OUT/pw $OUT/ktc
ls -la $OUT
tail -1 $OUT/${TAG}_bench_n1.json | cut -c1-400
head -6 $OUT/${TAG}_kernel_stats.csv
grep -i "moments" $OUT/${TAG}_pmc_fetch.csv $OUT/${TAG}_pmc_write.csv
head -5 $OUT/${TAG}_cyl_kernel_stats.csv
