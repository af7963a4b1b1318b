#!/usr/bin/env python3
"""Tiny helper for A/B runs on the GPU box: runs bench.py with each argument set and prints value + kernel times."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for spec in sys.argv[1:]:
    env = dict(os.environ)
    args = []
    for tok in spec.split():
        if "=" in tok and not tok.startswith("--"):
            k, v = tok.split("=", 1)
            env[k] = v
        else:
            args.append(tok)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--no-cpu-baseline"] + args, env=env,
                         capture_output=True, text=True)
    line = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    if not line:
        print(spec, "FAILED", out.stderr[-400:])
        continue
    d = json.loads(line[-1])
    km = d["roofline"].get("kernel_ms", {})
    print(f"{spec:50s} value={d['value']:.0f} ms/step={d['ms_per_step']:.3f} "
          + " ".join(f"{k.replace('cape_', '').replace('_kernel', '')}={v:.3f}" for k, v in km.items())
          + f" frac={d['roofline']['frac']:.3f}", flush=True)
