python bench.py --no-cpu-baseline --steps 10 --warmup 3 --no-parity-check 2>/dev/null | python -c '
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); f=d["find_primitives_equivalent"]
print("fpe %.3f ms  overlapped %.3f ms  polygons alone %.3f" % (f["ms_per_step"], f["two_handles_overlapped"]["ms_per_step"], d["boundary_polygons"]["ms_per_batch"]))'
