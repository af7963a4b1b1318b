#!/bin/bash
# SQ counter passes for the three kernels of the default bench (run through gpurun from the repo root):
#   pass 1: where the wave-cycles go (parked / issue-stalled / issuing)      pass 2: VALU + LDS + memory issue detail
# usage: profiles/pmc_sq.sh <tag> [extra bench args]
set -u
TAG=${1:-r02}; shift || true
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
BARGS="--no-cpu-baseline --steps 6 --warmup 2 --unique 256 $*"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVES GRBM_GUI_ACTIVE -d $OUT/p1 -o k -- python $R/bench.py $BARGS > /dev/null 2> $OUT/p1.err
python $R/profiles/summarize_rocprof.py pmc $(find $OUT/p1 -name "*.db" | head -1) > $OUT/${TAG}_pmc_sq_waits.csv
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE -d $OUT/p2 -o k -- python $R/bench.py $BARGS > /dev/null 2> $OUT/p2.err
python $R/profiles/summarize_rocprof.py pmc $(find $OUT/p2 -name "*.db" | head -1) > $OUT/${TAG}_pmc_sq_insts.csv
rm -rf $OUT/p1 $OUT/p2
cat $OUT/${TAG}_pmc_sq_waits.csv $OUT/${TAG}_pmc_sq_insts.csv | grep -v "at::\|vectorized\|elementwise\|Memset\|fill\|copy" 
tail -3 $OUT/p1.err $OUT/p2.err
