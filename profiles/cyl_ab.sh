#!/bin/bash
# A/B of the cylinder schedule on the GPU box: usage cyl_ab.sh <outdir> [label=libpath-or-ENV ...]
# every variant runs the three cylinder workloads (mixed room, tunnel, 1280x960 tunnel) and prints B's mean launch time
OUT=$1; shift
mkdir -p "$OUT"
run() { # label, env assignments...
  local label=$1; shift
  for wl in "room 4096 640 480" "tunnel 2048 640 480" "tumlike 2048 640 480" "tunnel 1024 1280 960"; do
    set -- $wl "$@"
    local scene=$1 frames=$2 w=$3 h=$4; shift 4
    env "$@" python bench.py --cylinders --scene $scene --frames $frames --width $w --height $h --steps 10 --warmup 3 \
        --no-cpu-baseline --no-parity-check > "$OUT/${label}_${scene}_${w}.json" 2> "$OUT/${label}_${scene}_${w}.err" || echo "FAILED $label $scene"
    python - "$OUT/${label}_${scene}_${w}.json" "$label" "$scene" "$w" <<'PY'
import json,sys
try:
    j=json.load(open(sys.argv[1])); k=j["roofline"]["kernel_ms"]
    print(f"{sys.argv[2]:10s} {sys.argv[3]:8s} {sys.argv[4]:5s} fps {j['value']:10.0f}  A1 {k['cape_cell_moments_kernel']:.3f} A2 {k['cape_cell_plane_kernel']:.3f} B {k['cape_grow_kernel']:.3f} ms")
except Exception as e:
    print(sys.argv[2], sys.argv[3], "no result", e)
PY
  done
}
for v in "$@"; do
  label=${v%%=*}; rest=${v#*=}
  run "$label" $rest
done
