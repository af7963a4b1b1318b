"""Test helper: the packed gather payload (include/cape_hip.h, cape_packed_*) built on the host from ORACLE results.
Restates what cape_pack_scan_kernel / cape_pack_copy_kernel write, so that the GPU's bytes can be compared with it and
the world_size-2 gloo test can ship real primitive lists without a GPU."""
import numpy as np


def pack_oracle(results, first_frame, layout, labels=False, status=None):
    from cape_amd import (PACKED_CYLINDER_DTYPE, PACKED_FRAME_DTYPE, PACKED_HEADER_DTYPE, PACKED_MAGIC, PACKED_PLANE_DTYPE,
                          PACKED_PLANES_DROPPED, PACKED_CYLINDERS_DROPPED, GATHER_LABELS)

    buf = np.zeros(layout["bytes_per_rank"], np.uint8)
    F, P, Cy, cells = layout["frames_capacity"], layout["planes_capacity"], layout["cylinders_capacity"], layout["cells"]
    hdr = buf[: PACKED_HEADER_DTYPE.itemsize].view(PACKED_HEADER_DTYPE)
    frames = buf[layout["frames_offset"]: layout["frames_offset"] + F * PACKED_FRAME_DTYPE.itemsize].view(PACKED_FRAME_DTYPE)
    planes = buf[layout["planes_offset"]: layout["planes_offset"] + P * PACKED_PLANE_DTYPE.itemsize].view(PACKED_PLANE_DTYPE)
    cyls = buf[layout["cylinders_offset"]: layout["cylinders_offset"] + Cy * PACKED_CYLINDER_DTYPE.itemsize].view(PACKED_CYLINDER_DTYPE)
    po = co = 0
    st_or = 0
    clipped = False
    for f, r in enumerate(results):
        st = int(status[f]) if status is not None else 0
        st_or |= st & 0xFF  # flag bits only: bits 8..15 of a frame's status are a count
        frames[f] = (po, len(r.planes), co, len(r.cylinders), st, len(r.segments))
        for k in range(len(r.planes)):
            o = r.planes[k]
            if po + k < P:
                q = planes[po + k]
                q["normal"] = o[0:3]
                q["d"] = o[3]
                q["centroid"] = o[4:7]
                q["mse"] = o[7]
                q["score"] = o[8]
                seg = int(o[19])
                q["sums"] = r.segments[seg, 9:18]
                q["point_count"] = int(o[9])
                q["segment"] = seg
        for k in range(len(r.cylinders)):
            if co + k < Cy:
                cyls[co + k]["axis"] = r.cylinders[k, 0:3]
                cyls[co + k]["radius"] = r.cylinders[k, 3]
        po += len(r.planes)
        co += len(r.cylinders)
        if labels:
            o1, o2 = layout["plane_labels_offset"], layout["cyl_labels_offset"]
            # one byte per cell on the wire: a label beyond 255 reads 255 (CAPE_PACKED_LABELS_CLIPPED in the header)
            buf[o1 + f * cells: o1 + (f + 1) * cells] = np.minimum(r.plane_labels, 255).astype(np.uint8)
            buf[o2 + f * cells: o2 + (f + 1) * cells] = np.minimum(r.cyl_labels, 255).astype(np.uint8)
            clipped = clipped or len(r.segments) > 255 or int(r.cyl_labels.max(initial=0)) > 255
    hdr[0] = (PACKED_MAGIC, len(results), first_frame, po, co, P, Cy,
              (PACKED_PLANES_DROPPED if po > P else 0) | (PACKED_CYLINDERS_DROPPED if co > Cy else 0) | (4 if clipped else 0), st_or, cells, F,
              GATHER_LABELS if labels else 0)
    return buf
