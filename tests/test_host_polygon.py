"""N1 (host-side boundary polygon): runs the C++ test program that mirrors the reference's tests/test_polygons.cpp."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_polygon_cpp_suite(tmp_path):
    exe = tmp_path / "test_polygon"
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-ffp-contract=off", "-o", str(exe), os.path.join(ROOT, "tests", "host", "test_polygon.cpp"),
                           os.path.join(ROOT, "rgb-d-slam_amd", "host", "boundary_polygon.cpp")])
    out = subprocess.run([str(exe)], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "all polygon tests passed" in out.stdout
