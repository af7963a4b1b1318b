"""The C-ABI library loads on a box without a GPU, exports every function include/cape_hip.h declares, and its
record layouts match the numpy mirrors used by the tests.  No compute call is made here."""
import ctypes as C
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_functions():
    src = open(os.path.join(ROOT, "include", "cape_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(cape_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol(hip_library):
    import cape_amd

    lib = cape_amd.load_library()
    declared = _declared_functions()
    assert len(declared) >= 14
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in cape_hip.h but not exported"
    assert set(cape_amd.EXPORTED_SYMBOLS) == set(declared)


def test_struct_sizes_match_header(hip_library):
    import cape_amd

    assert cape_amd.PLANE_SEGMENT_DTYPE.itemsize == 30 * 8 + 6 * 4
    assert cape_amd.CYLINDER_DTYPE.itemsize == 40
    assert cape_amd.HEADER_DTYPE.itemsize == 40  # ABI 2: + next_record, segment_base
    assert cape_amd.FRAME_RECORD_DTYPE.itemsize == 40 + 64 * 264 + 64 * 40
    assert cape_amd.PACKED_HEADER_DTYPE.itemsize == 48 and cape_amd.PACKED_FRAME_DTYPE.itemsize == 24
    assert cape_amd.PACKED_PLANE_DTYPE.itemsize == 152 and cape_amd.PACKED_CYLINDER_DTYPE.itemsize == 32
    assert cape_amd.CELL_STATS_DTYPE.itemsize == 18 * 8 + 6 * 4
    assert cape_amd.MATCH_DTYPE.itemsize == 8 + 64 * 4 + 2 * 64 * 2 + 64 * 64 * 2


def test_no_cpu_fallback(hip_library):
    """Without a HIP device cape_create must fail with CAPE_ERR_NO_DEVICE (-2), never compute on the CPU."""
    import pytest
    import torch

    import cape_amd

    if torch.cuda.is_available():
        pytest.skip("GPU present")
    lib = cape_amd.load_library()
    cfg = cape_amd.cape_config(640, 480, 550.0, 550.0, 320.0, 240.0, 0, 0, 1, 0, 0)
    h = C.c_void_p()
    assert lib.cape_create(C.byref(cfg), C.byref(h)) == -2
    assert b"no CPU fallback" in lib.cape_last_error()
    bad = cape_amd.cape_config(641, 480, 550.0, 550.0, 320.0, 240.0, 0, 0, 1, 0, 0)
    assert lib.cape_create(C.byref(bad), C.byref(h)) == -1


def test_create_rejects_unknown_flags_and_knob_values(hip_library):
    """an unknown CAPE_FLAG_* bit, or a debug knob with a value the library does not know, is an error of
    cape_create -- not a silent default.  (Argument checks run before the device probe: no GPU needed.)"""
    import subprocess
    import sys

    import cape_amd

    lib = cape_amd.load_library()
    h = C.c_void_p()
    bad = cape_amd.cape_config(640, 480, 550.0, 550.0, 320.0, 240.0, 1 << 7, 0, 1, 0, 0)
    assert lib.cape_create(C.byref(bad), C.byref(h)) == -1 and b"unknown CAPE_FLAG" in lib.cape_last_error()
    code = (
        "import ctypes as C, sys\n"
        "sys.path.insert(0, %r)\n"
        "import cape_amd\n"
        "L = cape_amd.load_library()\n"
        "cfg = cape_amd.cape_config(640, 480, 550.0, 550.0, 320.0, 240.0, 1, 0, 1, 0, 0)\n"
        "h = C.c_void_p()\n"
        "print(L.cape_create(C.byref(cfg), C.byref(h)), L.cape_last_error().decode())\n" % os.path.join(ROOT, "rgb-d-slam_amd", "python"))
    for knob in ("CAPE_RESUME", "CAPE_SCHEDULE", "CAPE_STAGE_A"):
        out = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, **{knob: "sideways"}), capture_output=True, text=True)
        assert out.returncode == 0, out.stderr[-800:]
        rc, msg = out.stdout.strip().split(" ", 1)
        assert int(rc) == -1 and knob in msg


def test_product_never_touches_oracle():
    """The product tree must not reference oracle/ (SURVEY / task rule: the oracle is the checker only)."""
    pkg = os.path.join(ROOT, "rgb-d-slam_amd")
    for dp, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".hpp", ".cpp", "Makefile")):
                txt = open(os.path.join(dp, f), errors="ignore").read()
                assert "cape_oracle" not in txt and "oracle/" not in txt, os.path.join(dp, f)


def test_unresolvable_rccl_is_an_error_not_a_crash(hip_library):
    """cape_comm_unique_id with a librccl that cannot be opened (CAPE_RCCL_LIB) returns CAPE_ERR_UNSUPPORTED with the
    loader's message; it used to dereference the second, null, dlerror().  Needs no GPU: the loader runs first."""
    import subprocess
    import sys

    code = (
        "import ctypes as C, sys\n"
        "sys.path.insert(0, %r)\n"
        "import cape_amd\n"
        "L = cape_amd.load_library()\n"
        "buf = (C.c_ubyte * cape_amd.COMM_ID_BYTES)()\n"
        "rc = L.cape_comm_unique_id(buf)\n"
        "print(rc, L.cape_last_error().decode())\n" % os.path.join(ROOT, "rgb-d-slam_amd", "python"))
    env = dict(os.environ, CAPE_RCCL_LIB="/nonexistent/librccl.so")
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True)
    assert out.returncode == 0, out.stderr[-800:]
    rc, msg = out.stdout.strip().split(" ", 1)
    assert int(rc) == -5 and "librccl.so not found" in msg and "/nonexistent/librccl.so" in msg


def test_log_lines_from_frame_records(hip_library):
    """cape_log_records / cape_set_log_callback: the lines the reference's find_primitives logs on the hot path
    (primitive_detection.cpp:302, :374 / :497, :618) are derived from the frame record -- status bits and segment fields --
    on the host.  Pure host code: crafted records, no GPU."""
    import numpy as np

    import cape_amd

    rec = np.zeros(4, cape_amd.FRAME_RECORD_DTYPE)
    # frame 0: nothing to say
    rec["header"]["n_plane_segments"][0] = 2
    rec["segments"][0]["planar"][:2] = 1
    rec["segments"][0]["merge_label"][:2] = [0, 1]
    rec["segments"][0]["boundary_count"][:2] = [40, 3]
    # frame 1: two "not planar after merge", then the seed loop ends on an invalid seed
    rec["header"]["status"][1] = cape_amd.FRAME_INVALID_SEED | (2 << 8) | (1 << 4)  # (+ an unrelated bit: CAPE_FRAME_INORDER_CELLS)
    # frame 2: a planar merge root with two boundary points (rejected), a merged-away segment and a non-planar one that say nothing
    rec["header"]["n_plane_segments"][2] = 3
    rec["segments"][2]["planar"][:3] = [1, 1, 0]
    rec["segments"][2]["merge_label"][:3] = [0, 0, 2]
    rec["segments"][2]["boundary_count"][:3] = [2, 0, 0]
    # frame 3: a capacity overflow (the library's own warning)
    rec["header"]["status"][3] = 1
    lines = cape_amd.log_records(rec)
    assert lines == [
        (0, "Plane segment is not planar after merge", 1), (0, "Plane segment is not planar after merge", 1),
        (1, "Could not find a single plane segment: invalid seed", 1),
        (1, "Could not find a correct boundary polygon, rejecting plane segment", 2),
        (1, "find_primitives: per-frame capacity exceeded, primitive list truncated", 3)]
    assert cape_amd.frame_not_planar_count(rec["header"]["status"][1]) == 2
    lib = cape_amd.load_library()
    assert lib.cape_log_records(None, 1, cape_amd.LOG_FN(0), None) == -1


def test_timings_struct_carries_the_five_buckets(hip_library):
    import cape_amd

    names = [f[0] for f in cape_amd.cape_timings._fields_]
    assert names[-5:] == ["reset_s", "init_s", "grow_phase_s", "merge_s", "refine_s"]  # primitive_detection.cpp:126-160
    assert C.sizeof(cape_amd.cape_timings) == 12 * 8
    src = open(os.path.join(ROOT, "include", "cape_hip.h")).read()
    assert "double reset_s, init_s, grow_phase_s, merge_s, refine_s;" in src
