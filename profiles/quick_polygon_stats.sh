#!/bin/bash
# quick: bench line + polygon kernel stats
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/q_$1
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
python $R/bench.py --no-cpu-baseline --steps 10 --warmup 3 > $OUT/bench.json 2> $OUT/bench.err
rocprofv3 --kernel-trace --stats -d $OUT/kt_poly -o k -- python $R/bench.py --no-cpu-baseline --steps 3 --warmup 1 --no-cylinders-on --no-parity-check > /dev/null 2> $OUT/kt_poly.err
python $R/profiles/summarize_rocprof.py stats $(find $OUT/kt_poly -name "*.db" | head -1) | grep "kernel,\|polygon" > $OUT/polygon_kernel_stats.csv; rm -rf $OUT/kt_poly
cat $OUT/polygon_kernel_stats.csv
python - <<PY
import json
d=json.loads(open("$OUT/bench.json").read().strip().splitlines()[-1])
print({k:d[k] for k in ("value","ms_per_step")})
print("cyl", d["cylinders_on"]["value"], d["cylinders_on"]["kernel_ms"])
print("fpe", json.dumps(d.get("find_primitives_equivalent"), indent=None)[:1500])
print("poly", d["boundary_polygons"])
PY
tail -5 $OUT/bench.err
