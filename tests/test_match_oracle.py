"""Known answers for oracle/match_oracle.py (the CPU restatement of MapPlane::find_matches on cell masks) -- runs
without a GPU.  Hand-built masks pin the selection rules the device kernel is compared against."""
import numpy as np

import match_oracle as mo


def _planes(masks, normals=None, d=None):
    k = len(masks)
    return {"masks": masks, "normals": np.array(normals if normals is not None else [[0, 0, 1.0]] * k, float),
            "d": np.array(d if d is not None else [1000.0] * k, float)}


def _mask(cells, idx):
    m = np.zeros(cells, bool)
    m[list(idx)] = True
    return m


def test_best_overlap_wins_and_index0_quirk():
    prev = _planes([_mask(100, range(0, 40))])
    cur = _planes([_mask(100, range(0, 30)), _mask(100, range(30, 40)), _mask(100, range(60, 70))])
    # plane 0 has the largest intersection (30) but the reference can never return index 0 (map_primitive.cpp:146)
    assert mo.match_frame(prev, cur)[0] == [-1]
    assert mo.match_frame(prev, cur, allow_index0=True)[0] == [0]
    m, ap, ac, inter = mo.match_frame(prev, cur)
    assert ap == [40] and ac == [30, 10, 10] and inter.tolist() == [[30, 10, 0]]


def test_overlap_ratio_threshold_uses_detected_area():
    prev = _planes([_mask(100, range(0, 10)), _mask(100, range(50, 60))])
    # detected plane 1 covers 30 cells, shares 10 with previous plane 0: 10/30 < 0.4 but >= 0.2
    cur = _planes([_mask(100, range(90, 95)), _mask(100, range(0, 30))])
    assert mo.match_frame(prev, cur)[0] == [-1, -1]
    assert mo.match_frame(prev, cur, advanced=True)[0] == [1, -1]
    assert abs(mo.MIN_OVERLAP - 0.4) < 1e-7 and mo.MIN_OVERLAP != 0.4   # the float constant, widened


def test_gates_and_single_use_of_a_detected_plane():
    a, b = _mask(100, range(0, 20)), _mask(100, range(0, 20))
    cur = _planes([_mask(100, range(80, 90)), _mask(100, range(0, 20))])
    # two previous planes both overlap detected plane 1: the first takes it, the second finds it already matched
    assert mo.match_frame(_planes([a, b]), cur)[0] == [1, -1]
    # distance gate: |d - d'| must be < 100 mm (strict)
    assert mo.match_frame(_planes([a], d=[1100.0]), cur)[0] == [-1]
    assert mo.match_frame(_planes([a], d=[1099.999]), cur)[0] == [1]
    # angle gate: |cos| > cos 20 deg, sign of the normal irrelevant
    c19, c21 = np.deg2rad(19.0), np.deg2rad(21.0)
    assert mo.match_frame(_planes([a], normals=[[np.sin(c19), 0, -np.cos(c19)]]), cur)[0] == [1]
    assert mo.match_frame(_planes([a], normals=[[np.sin(c21), 0, np.cos(c21)]]), cur)[0] == [-1]


def test_tie_keeps_the_lowest_index_and_empty_inputs():
    prev = _planes([_mask(100, range(0, 40))])
    cur = _planes([_mask(100, range(90, 95)), _mask(100, range(0, 20)), _mask(100, range(20, 40))])
    assert mo.match_frame(prev, cur)[0] == [1]            # 20 == 20: strict `>` keeps the first
    assert mo.match_frame(_planes([]), cur)[0] == []
    assert mo.match_frame(prev, _planes([]))[0] == [-1]
    assert mo.match_frame(_planes([_mask(100, [])]), cur)[0] == [-1]   # projectedArea <= 0


def test_plane_masks_follow_merge_groups():
    labels = np.array([0, 1, 1, 2, 3, 3, 0, 2])
    merge = np.array([0, 1, 0])          # segment 2 merged into root 0
    masks, roots = mo.plane_masks(labels, None, merge, np.array([True, True, False]))
    assert roots == [0, 1]
    assert masks[0].tolist() == [False, True, True, False, True, True, False, False]
    assert masks[1].tolist() == [False, False, False, True, False, False, False, True]
