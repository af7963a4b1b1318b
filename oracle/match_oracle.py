"""CPU restatement of the plane-matching selection loop, on cell masks -- TEST INFRASTRUCTURE ONLY.

Follows MapPlane::find_matches (reference src/map_management/map_features/map_primitive.cpp:91-161) as it is driven by
Feature_Map::get_matches (src/map_management/feature_map.hpp:647-670), with the two substitutions the device
pre-filter makes: the "map" planes are the planes of the previous frame seen through the identity
pose, and areas are counted in cells of the label grid instead of polygon mm^2.  Plain Python loops; parity status
**parity unpinned** like the rest of oracle/ (the reference holds no test for find_matches).
"""
import math

import numpy as np

MAX_ANGLE_D = 20.0        # parameters::matching::maximumAngleForPlaneMatch_d, src/parameters.hpp:92-93
MAX_DISTANCE_MM = 100.0   # maximumDistanceForPlaneMatch_mm, :94-95
MIN_OVERLAP = float(np.float32(0.4))  # minimumPlaneOverlapToConsiderMatch (a float constant), :90-91


def plane_masks(plane_labels, segments, merge_labels, is_output):
    """add_planes_to_primitives' mask per output plane (primitive_detection.cpp:576-594): cells labelled j+1 for every
    j >= root with planeMergeLabels[j] == root.  Returns (list of boolean masks, list of root segment indices)."""
    masks, roots = [], []
    n = len(merge_labels)
    for r in range(n):
        if not is_output[r]:
            continue
        m = np.zeros(plane_labels.shape, bool)
        for j in range(r, n):
            if merge_labels[j] == merge_labels[r]:
                m |= plane_labels == (j + 1)
        masks.append(m)
        roots.append(r)
    return masks, roots


def match_frame(prev, cur, advanced=False, allow_index0=False):
    """prev / cur: dicts with masks (list of bool arrays), normals (k x 3), d (k).  Returns (match[j] for every
    previous plane, area_prev, area_cur, inter[j][i])."""
    min_cos = abs(math.cos(MAX_ANGLE_D * math.pi / 180.0))
    threshold = MIN_OVERLAP / 2 if advanced else MIN_OVERLAP
    n_prev, n_cur = len(prev["masks"]), len(cur["masks"])
    area_prev = [int(m.sum()) for m in prev["masks"]]
    area_cur = [int(m.sum()) for m in cur["masks"]]
    inter = np.zeros((n_prev, n_cur), np.int64)
    for j in range(n_prev):
        for i in range(n_cur):
            inter[j, i] = int((prev["masks"][j] & cur["masks"][i]).sum())
    is_matched = [False] * n_cur          # _isDetectedFeatureMatched, feature_map.hpp:648
    match = [-1] * n_prev
    for j in range(n_prev):               # map features in order, feature_map.hpp:652
        projected_area = float(area_prev[j])
        greatest = 0.0
        if projected_area <= 0.0:         # map_primitive.cpp:111-112
            continue
        selected = -1
        for i in range(n_cur):
            if is_matched[i]:
                continue
            a, b = cur["normals"][i], prev["normals"][j]
            cos_angle = (a[0] * b[0] + a[1] * b[1]) + a[2] * b[2]
            if not (abs(cur["d"][i] - prev["d"][j]) < MAX_DISTANCE_MM) or not (abs(cos_angle) > min_cos):
                continue
            new_area = float(area_cur[i])
            inter_area = float(inter[j, i])
            if new_area > 0 and inter_area > greatest and inter_area / new_area >= threshold:
                selected = i
                greatest = inter_area
        if selected <= 0 and not (allow_index0 and selected == 0):  # quirk, map_primitive.cpp:146
            continue
        match[j] = selected
        is_matched[selected] = True
    return match, area_prev, area_cur, inter
