#!/bin/bash
# Round-5 randomised sweeps of the final library (run through gpurun from the repo root): every observable of every frame against the oracle.
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out
O=gpurun_out/r05_fuzz_parity.txt
run() { echo "== $*" >> $O; ( "$@" ) 2>&1 | tail -3 >> $O; }
: > $O
run python profiles/fuzz_parity.py 8192 51
run env CAPE_RESUME=group python profiles/fuzz_parity.py 4096 52
run python profiles/fuzz_parity.py 1024 53 1280 960
run env FUZZ_CELLS=1 python profiles/fuzz_parity.py 2048 54
run env FUZZ_BATCH=1 python profiles/fuzz_parity.py 1024 55
run python profiles/fuzz_polygons.py 1024 56
cat $O
