import os, sys
sys.path.insert(0, "rgb-d-slam_amd/python"); sys.path.insert(0, "profiles")
import general_instance_rate as G
for scene in ("room", "tunnel"):
    for cyl in (False, True):
        r = G.run(1920, 1080, 1024, scene, cyl, False)
        print(os.environ.get("CAPE_A2_WIDE_BATCH", "default"), scene, cyl, {k: round(v, 3) for k, v in r.items()})
