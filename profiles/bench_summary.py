import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("value",round(d["value"]), "ms",round(d["ms_per_step"],4), "frac",round(d["roofline"]["frac"],4))
c=d.get("cylinders_on",{}); print("cylinders_on", round(c.get("value",0)), c.get("kernel_ms"))
f=d.get("find_primitives_equivalent",{}); print("fpe", round(f.get("value",0)), {k:v for k,v in f.items() if isinstance(v,(int,float))})
t=d.get("two_handles_overlapped",{}); print("two_handles", {k:(round(v) if isinstance(v,float) and v>1000 else v) for k,v in t.items() if isinstance(v,(int,float))})
