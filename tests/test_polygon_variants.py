"""Risk measurement for the unpinned parts of the polygon oracle: FLANN's tie order and Boost.Geometry's
simplify / is_valid semantics are restated from documentation; every such choice is a compile-time switch of
oracle/polygon_oracle.cpp (POLY_VAR_*).  The full sweep is profiles/r05_polygon_variants.txt (oracle/polygon_variants.py 96 3:
2 817 planes, 760 frame pairs, no validity / fallback / match decision depends on any switch); this test keeps the switches
compiling and the headline -- decisions do not depend on them -- on a small sample."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_match_decisions_do_not_depend_on_the_unpinned_choices(oracle_mod):
    out = subprocess.run([sys.executable, os.path.join(ROOT, "oracle", "polygon_variants.py"), "6", "3"], capture_output=True, text=True,
                         timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    rows = [ln for ln in out.stdout.splitlines() if ln.startswith("| `")]
    assert len(rows) == 5
    moved = 0
    for ln in rows:
        cells = [c.strip() for c in ln.strip("|").split("|")]
        validity, fallback, verts, decisions = int(cells[2]), int(cells[3]), int(cells[4]), int(cells[8])
        assert validity == 0 and fallback == 0 and decisions == 0, ln
        assert float(cells[7]) >= 0.95, ln  # worst IoU against the default polygon
        moved += verts
    assert moved > 0, "at least the rotated-start variant must move vertices (else the switches compile to the default)"
