#!/bin/bash
# Round-6 extras beside profiles/collect.sh r06 (run through gpurun from the repo root): the randomised sweeps incl. record chains and
# wide grids, the general grow instance's rates, the overlay's batch rates with the new default sharding, the single-frame latency,
# the matcher's ticket A/B (variant libraries built beforehand with profiles/build_tu_variant.sh).
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/collect_r06x
mkdir -p $OUT
cd $R
bash profiles/fuzz_r06.sh > $OUT/fuzz.log 2>&1
cp gpurun_out/r06_fuzz_parity.txt $OUT/
python profiles/general_instance_rate.py $OUT/r06_general_instance_rate.txt > $OUT/general.log 2>&1
python profiles/overlay_batch_rate.py 128:256 256:256 512:256 1024:256 2>&1 | grep -v "host class" > $OUT/r06_overlay_batch_rate.txt
python profiles/single_frame_latency.py > $OUT/r06_single_frame_latency.txt 2>&1
python profiles/host_input_rate.py 256 > $OUT/r06_host_input_rate.txt 2>&1
{
  for v in "" mp_nocoop mp_static mp_c0g8; do
    if [ -n "$v" ]; then export CAPE_HIP_LIB=$R/rgb-d-slam_amd/lib/exp/libcape_$v.so; else unset CAPE_HIP_LIB; fi
    [ -n "$v" ] && [ ! -f "$CAPE_HIP_LIB" ] && continue
    echo "variant ${v:-shipped (cooperative tiers 1..3, tickets)}"; python profiles/match_ms.py; python profiles/match_ms.py
  done
  unset CAPE_HIP_LIB
} 2>&1 | grep -v amdgpu.ids > $OUT/r06_match_tickets_raw.txt
tail -4 $OUT/r06_fuzz_parity.txt $OUT/r06_overlay_batch_rate.txt $OUT/r06_single_frame_latency.txt
