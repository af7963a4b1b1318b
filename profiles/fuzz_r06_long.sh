#!/bin/bash
# Round-6 long sweep (fresh seeds; ~15 GPU-minutes): every observable of every frame against the oracle.
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out
O=gpurun_out/r06_fuzz_long.txt
run() { echo "== $*" >> $O; ( "$@" ) 2>&1 | grep "MISMATCH batch\|coverage\|RESULT" | tail -6 >> $O; }
: > $O
run env FUZZ_BIG=1 python profiles/fuzz_parity.py 8192 71
run env FUZZ_BIG=1 CAPE_RESUME=group python profiles/fuzz_parity.py 2048 72
run env FUZZ_BIG=1 FUZZ_CELLS=1 python profiles/fuzz_parity.py 1024 73
run env FUZZ_BIG=1 python profiles/fuzz_parity.py 1536 74 1280 960
run env FUZZ_BIG=1 python profiles/fuzz_parity.py 768 75 1920 1080
run env FUZZ_BIG=1 CAPE_GROW=general python profiles/fuzz_parity.py 256 76 1920 1080
run env FUZZ_BIG=1 python profiles/fuzz_parity.py 256 77 2560 1280
run env FUZZ_BIG=1 python profiles/fuzz_parity.py 256 78 1080 1920
run env FUZZ_BIG=1 FUZZ_BATCH=1 python profiles/fuzz_parity.py 512 79
run python profiles/fuzz_polygons.py 2048 80
cat $O
