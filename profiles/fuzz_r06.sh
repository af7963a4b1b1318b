#!/bin/bash
# Round-6 randomised sweeps of the final library (run through gpurun from the repo root): every observable of every frame against the
# oracle -- the fast kernels, the general grow instance on their grids (CAPE_GROW=general), record chains (FUZZ_BIG), the grids only
# the general instance serves, one-frame handles.
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out
O=gpurun_out/r06_fuzz_parity.txt
run() { echo "== $*" >> $O; ( "$@" ) 2>&1 | tail -3 >> $O; }
: > $O
run python profiles/fuzz_parity.py 4096 61
run env CAPE_GROW=general python profiles/fuzz_parity.py 2048 62
run env FUZZ_BIG=1 python profiles/fuzz_parity.py 2048 63
run env FUZZ_BIG=1 python profiles/fuzz_parity.py 512 64 1280 960
run env FUZZ_BIG=1 FUZZ_BATCH=1 python profiles/fuzz_parity.py 256 65 1280 960
run python profiles/fuzz_parity.py 384 66 1920 1080
run env FUZZ_BIG=1 python profiles/fuzz_parity.py 192 67 1080 1920
run env FUZZ_BATCH=1 python profiles/fuzz_parity.py 64 68 1920 1080
run python profiles/fuzz_polygons.py 1024 69
cat $O
