#!/bin/bash
# SQ counters of the polygon pass (cape_build_polygons on 4 096 room frames): where the wave-cycles go and the instruction mix.
# usage: profiles/pmc_polygon.sh <tag>   (through gpurun from the repo root; two separate --pmc passes, no tracing beside them)
set -u
TAG=${1:-r04}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/pmcpoly_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
BARGS="--no-cpu-baseline --steps 2 --warmup 1 --no-cylinders-on --no-parity-check"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVES GRBM_GUI_ACTIVE -d $OUT/p1 -o k -- python $R/bench.py $BARGS > /dev/null 2> $OUT/p1.err
python $R/profiles/summarize_rocprof.py pmc $(find $OUT/p1 -name "*.db" | head -1) | grep "kernel,counter\|polygon" > $OUT/${TAG}_polygon_pmc_sq_waits.csv
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE -d $OUT/p2 -o k -- python $R/bench.py $BARGS > /dev/null 2> $OUT/p2.err
python $R/profiles/summarize_rocprof.py pmc $(find $OUT/p2 -name "*.db" | head -1) | grep "kernel,counter\|polygon" > $OUT/${TAG}_polygon_pmc_sq_insts.csv
rm -rf $OUT/p1 $OUT/p2
cat $OUT/${TAG}_polygon_pmc_sq_waits.csv $OUT/${TAG}_polygon_pmc_sq_insts.csv | grep -v "inter_kernel\|gate\|select"
tail -2 $OUT/p1.err $OUT/p2.err
