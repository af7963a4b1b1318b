import sys, os
sys.path.insert(0, os.path.join(os.environ["GRAFT_REPO_ROOT"], "rgb-d-slam_amd", "python"))
from cape_amd import Extractor
ex = Extractor(640, 480, max_batch=4)
print(os.environ.get("CAPE_HIP_LIB","default").split("/")[-1], "CUs", ex.compute_units, "grow frames per CU", ex.grow_frames_per_cu)
