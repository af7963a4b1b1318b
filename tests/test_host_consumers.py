"""The reference's consumers of the primitives library's value types (map_primitive.cpp:102-153,217-278,
plane_with_tracking.cpp:33-48, matches_containers.hpp:50-60, rgbd_slam.cpp:291-297) restated in
tests/host/test_consumers.cpp and compiled against the overlay + compat types this repo ships.  Needs no GPU: it also
checks that a detector without a device yields no primitives instead of throwing or computing on the CPU."""
import os
import subprocess


def test_consumer_call_sites_compile_and_run(host_binaries):
    exe = os.path.join(host_binaries, "test_consumers.exe")
    env = dict(os.environ, HIP_VISIBLE_DEVICES="", ROCR_VISIBLE_DEVICES="")
    out = subprocess.run([exe], capture_output=True, text=True, timeout=120, env=env)
    assert out.returncode == 0, f"rc {out.returncode}\n{out.stdout}\n{out.stderr}"
    assert "consumer call sites ok" in out.stdout
