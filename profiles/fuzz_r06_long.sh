#!/bin/bash
# Round-6 long sweep (~15 GPU-minutes): every observable of every frame against the oracle.
# usage: fuzz_r06_long.sh [seed_base=70] [output=gpurun_out/r06_fuzz_long.txt]   -- seeds seed_base+1 ... seed_base+10
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out
S=${1:-70}
O=${2:-gpurun_out/r06_fuzz_long.txt}
run() { echo "== $*" >> $O; ( "$@" ) 2>&1 | grep "MISMATCH batch\|coverage\|RESULT" | tail -6 >> $O; }
: > $O
run env FUZZ_BIG=1 python profiles/fuzz_parity.py 8192 $((S+1))
run env FUZZ_BIG=1 CAPE_RESUME=group python profiles/fuzz_parity.py 2048 $((S+2))
run env FUZZ_BIG=1 FUZZ_CELLS=1 python profiles/fuzz_parity.py 1024 $((S+3))
run env FUZZ_BIG=1 python profiles/fuzz_parity.py 1536 $((S+4)) 1280 960
run env FUZZ_BIG=1 python profiles/fuzz_parity.py 768 $((S+5)) 1920 1080
run env FUZZ_BIG=1 CAPE_GROW=general python profiles/fuzz_parity.py 256 $((S+6)) 1920 1080
run env FUZZ_BIG=1 python profiles/fuzz_parity.py 256 $((S+7)) 2560 1280
run env FUZZ_BIG=1 python profiles/fuzz_parity.py 256 $((S+8)) 1080 1920
run env FUZZ_BIG=1 FUZZ_BATCH=1 python profiles/fuzz_parity.py 512 $((S+9))
run python profiles/fuzz_polygons.py 2048 $((S+10))
cat $O
