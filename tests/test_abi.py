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


def test_record_chains_on_the_host(hip_library):
    """ABI 2: a frame of more than 64 plane segments is a chain of records (cape_frame_header.next_record).  Pure host logic on crafted
    records, no GPU: the binding's FrameResults walks the chain (segments, planes, boundary slabs, cylinder labels), cape_log_records
    follows it as far as the caller's array reaches and never loops on a zero-filled or backward link."""
    import numpy as np
    import pytest

    import cape_amd

    assert cape_amd.load_library().cape_abi_version() == cape_amd.CAPE_ABI_VERSION == 2
    max_batch, cap = 2, 16
    rec = np.zeros(max_batch, cape_amd.FRAME_RECORD_DTYPE)
    spill = np.zeros(2, cape_amd.FRAME_RECORD_DTYPE)
    rec["header"]["next_record"] = -1
    spill["header"]["next_record"] = -1
    # frame 0: 64 + 64 + 2 segments over three records; frame 1: three segments in its own record
    rec["header"]["n_plane_segments"][0], rec["header"]["next_record"][0] = 130, max_batch + 0
    spill["header"]["n_plane_segments"][0], spill["header"]["segment_base"][0], spill["header"]["next_record"][0] = 66, 64, max_batch + 1
    spill["header"]["n_plane_segments"][1], spill["header"]["segment_base"][1] = 2, 128
    for r, base in ((rec[0], 0), (spill[0], 64), (spill[1], 128)):
        n = min(64, int(r["header"]["n_plane_segments"]))
        r["segments"]["merge_label"][:n] = np.arange(base, base + n)
        r["segments"]["planar"][:n] = 1
        r["segments"]["is_output"][:n] = (np.arange(base, base + n) % 2 == 0)
        r["segments"]["boundary_offset"][:n] = np.arange(n) % 4 * 3
        r["segments"]["boundary_count"][:n] = 3
    rec["header"]["n_plane_segments"][1] = 3
    rec["segments"]["merge_label"][1][:3] = [0, 0, 2]
    rec["segments"]["planar"][1][:3] = [1, 1, 1]
    rec["segments"]["boundary_count"][1][:3] = [2, 0, 5]  # a planar merge root with two boundary points: the reference's warning
    rec["header"]["n_cylinder_labels"][0] = 65
    rec["cylinders"]["kept"][0][:64] = 1
    spill["header"]["n_cylinder_labels"][0] = 1
    spill["cylinders"]["kept"][0][0] = 1
    bd = np.arange(max_batch * cap * 3, dtype=np.float64).reshape(max_batch, cap, 3)
    sbd = -np.arange(2 * cap * 3, dtype=np.float64).reshape(2, cap, 3)
    res = cape_amd.FrameResults(rec, None, None, bd, max_batch, spill, sbd)
    segs = res.segments(0)
    assert len(segs) == 130 and np.array_equal(segs["merge_label"], np.arange(130)) and len(res.chain(0)) == 3 and len(res.chain(1)) == 1
    assert len(res.planes(0)) == 65 and len(res.cylinder_labels(0)) == 65 and len(res.segments(1)) == 3
    pb = res.plane_boundaries(0)
    assert len(pb) == 65 and pb[0].shape == (3, 3) and pb[0][0, 0] >= 0 and pb[32][0, 0] <= 0 and pb[64][0, 0] <= 0  # own slab, then the spill slabs
    # a chain whose spill records were not copied is an error, not a silent truncation
    with pytest.raises(cape_amd.CapeError):
        cape_amd.FrameResults(rec, None, None, bd, max_batch).segments(0)
    # the log lines: frame 1's rejected plane; the chain of frame 0 is followed through an array that holds batch + pool
    both = np.concatenate([rec, spill])
    spill2 = both[max_batch:]
    spill2["segments"]["boundary_count"][1][1] = 1  # segment 129 (a merge root, planar): rejected
    lines = cape_amd.log_records(both)
    assert lines.count((1, "Could not find a correct boundary polygon, rejecting plane segment", 0)) == 1
    assert lines.count((1, "Could not find a correct boundary polygon, rejecting plane segment", 1)) == 1
    # links that point backwards or at the record itself never loop
    both["header"]["next_record"][3] = 2
    both["header"]["next_record"][1] = 1
    assert isinstance(cape_amd.log_records(both), list)
