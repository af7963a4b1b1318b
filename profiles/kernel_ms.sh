#!/bin/bash
# one line per leg: the per-kernel event times bench.py reports (quick A/B of kernel changes); usage: kernel_ms.sh [bench args]
R=${GRAFT_REPO_ROOT:-$(pwd)}
python $R/bench.py --no-cpu-baseline --steps 20 --warmup 5 --no-polygons "$@" 2>/dev/null | python -c '
import json, sys
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print("value %.0f ms_per_step %.4f" % (d["value"], d["ms_per_step"]), "kernel_ms", {k: round(v, 4) for k, v in (d.get("kernel_ms") or d.get("roofline", {}).get("kernel_ms", {})).items()})
c = d.get("cylinders_on")
if c: print("cylinders_on %.0f" % c["value"], {k: round(v, 4) for k, v in c["kernel_ms"].items()})
f = d.get("find_primitives_equivalent")
if f: print("find_primitives_equivalent %.0f ms %.3f" % (f["value"], f["ms_per_step"]), {k: round(v, 3) for k, v in f["call_ms"].items()})
'
