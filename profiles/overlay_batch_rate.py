#!/usr/bin/env python3
"""Throughput of the overlay's batch entry (Primitive_Detection::find_primitives_batch: host frames in, plane / cylinder
containers out, boundary polygons included) by chunk size and shard count, through rgb-d-slam_amd/lib/latency_bench.exe
(profiles/single_frame_latency.cpp; CAPE_BENCH_BATCH / CAPE_BENCH_CHUNK).  usage: overlay_batch_rate.py"""
import sys,subprocess,os
sys.path.insert(0,"/root/repo/rgb-d-slam_amd/python")
import numpy as np
from cape_amd import synth
intr=synth.TUM_FR1_INTRINSICS
frames=np.stack([synth.tumlike(seed=7,frame=f) for f in range(16)])
frames.tofile("/tmp/f.f32")
exe="/root/repo/rgb-d-slam_amd/lib/latency_bench.exe"
for B,Cn in [(int(a.split(":")[0]),int(a.split(":")[1])) for a in sys.argv[1:]] or ((1024,64),(1024,128),(1024,256)):
    env=dict(os.environ,CAPE_BENCH_BATCH=str(B),CAPE_BENCH_CHUNK=str(Cn))
    out=subprocess.run([exe,"/tmp/f.f32","16","640","480",str(intr["fx"]),str(intr["fy"]),str(intr["cx"]),str(intr["cy"]),"1"],capture_output=True,text=True,env=env)
    print("batch",B,"chunk",Cn); print("\n".join(l for l in out.stdout.splitlines() if l.startswith("overlay batch")))
