#!/bin/bash
# Round-5 extras beside profiles/collect.sh (run through gpurun from the repo root): the SQ passes the VALU-issue roofline is
# checked against, the host-input and overlay end-to-end rates on the final library, the single-frame latency, the polygon
# sweep against the oracle of the reference's algorithm.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/collect_r05x
mkdir -p $OUT
cd $R
bash profiles/pmc_sq.sh r05 > $OUT/pmc_sq.log 2>&1
cp gpurun_out/pmc_r05/r05_pmc_sq_insts.csv gpurun_out/pmc_r05/r05_pmc_sq_waits.csv $OUT/ 2>/dev/null
python profiles/host_input_rate.py 256 > $OUT/r05_host_input_rate.txt 2>&1
python profiles/overlay_batch_rate.py > $OUT/r05_overlay_batch_rate.txt 2>&1
python profiles/single_frame_latency.py > $OUT/r05_single_frame_latency.txt 2>&1
python profiles/polygon_vs_oracle.py 256 3 > $OUT/r05_polygon_vs_oracle.txt 2>&1
tail -3 $OUT/r05_host_input_rate.txt $OUT/r05_overlay_batch_rate.txt $OUT/r05_polygon_vs_oracle.txt
