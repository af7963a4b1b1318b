"""Multi-GPU plumbing.  Frames shard by contiguous blocks, one process per GPU, no collective inside a
frame; once per batch the ranks all-gather their PACKED primitive lists (include/cape_hip.h: cape_packed_*), a fixed byte
count per rank.  On the GPU box the collective is ONE ncclAllGather issued by libcape_hip itself
(cape_gather_primitives, RCCL over xGMI); this module holds the host-side pieces around it: shard arithmetic, the
exchange of the communicator id, and the parser of the gathered bytes.  The CPU tests drive the same parser through a
world_size-2 gloo all-gather."""
import numpy as np


def shard_range(n_frames, rank, world):
    """Contiguous block of frames owned by `rank`: the first (n % world) ranks get one extra frame."""
    q, r = divmod(n_frames, world)
    start = rank * q + min(rank, r)
    return start, start + q + (1 if rank < r else 0)


def largest_shard(n_frames, world):
    return -(-n_frames // world)


def packed_layout(frames_capacity, cells, planes_per_frame=16, cylinders_per_frame=8, labels=False):
    """Host restatement of cape_gather_configure's layout arithmetic (sections on 16-byte boundaries)."""
    from . import PACKED_CYLINDER_DTYPE, PACKED_FRAME_DTYPE, PACKED_HEADER_DTYPE, PACKED_PLANE_DTYPE

    def a16(v):
        return (v + 15) & ~15

    lay = dict(frames_capacity=frames_capacity, planes_capacity=frames_capacity * planes_per_frame,
               cylinders_capacity=frames_capacity * cylinders_per_frame, cells=cells,
               plane_labels_offset=0, cyl_labels_offset=0)
    off = a16(PACKED_HEADER_DTYPE.itemsize)
    lay["frames_offset"] = off
    off = a16(off + frames_capacity * PACKED_FRAME_DTYPE.itemsize)
    lay["planes_offset"] = off
    off = a16(off + lay["planes_capacity"] * PACKED_PLANE_DTYPE.itemsize)
    lay["cylinders_offset"] = off
    off = a16(off + lay["cylinders_capacity"] * PACKED_CYLINDER_DTYPE.itemsize)
    if labels:
        lay["plane_labels_offset"] = off
        off = a16(off + frames_capacity * cells)
        lay["cyl_labels_offset"] = off
        off = a16(off + frames_capacity * cells)
    lay["bytes_per_rank"] = off
    return lay


class Shard:
    """One rank's packed buffer, parsed (views into the bytes, no copies)."""

    def __init__(self, buf, layout):
        from . import (PACKED_CYLINDER_DTYPE, PACKED_FRAME_DTYPE, PACKED_HEADER_DTYPE, PACKED_MAGIC, PACKED_PLANE_DTYPE)

        buf = np.ascontiguousarray(np.frombuffer(buf, dtype=np.uint8) if not isinstance(buf, np.ndarray) else buf)
        assert buf.size == layout["bytes_per_rank"], (buf.size, layout["bytes_per_rank"])
        self.header = buf[: PACKED_HEADER_DTYPE.itemsize].view(PACKED_HEADER_DTYPE)[0]
        if int(self.header["magic"]) != PACKED_MAGIC:
            raise ValueError("not a packed cape shard (bad magic)")
        F, P, Cy = layout["frames_capacity"], layout["planes_capacity"], layout["cylinders_capacity"]
        o = layout["frames_offset"]
        self.frames = buf[o: o + F * PACKED_FRAME_DTYPE.itemsize].view(PACKED_FRAME_DTYPE)[: int(self.header["n_frames"])]
        o = layout["planes_offset"]
        self.planes = buf[o: o + P * PACKED_PLANE_DTYPE.itemsize].view(PACKED_PLANE_DTYPE)
        o = layout["cylinders_offset"]
        self.cylinders = buf[o: o + Cy * PACKED_CYLINDER_DTYPE.itemsize].view(PACKED_CYLINDER_DTYPE)
        cells = layout["cells"]
        self.plane_labels = self.cyl_labels = None
        if layout["plane_labels_offset"]:
            o = layout["plane_labels_offset"]
            self.plane_labels = buf[o: o + F * cells].reshape(F, cells)[: len(self.frames)]
            o = layout["cyl_labels_offset"]
            self.cyl_labels = buf[o: o + F * cells].reshape(F, cells)[: len(self.frames)]

    @property
    def first_frame(self):
        return int(self.header["first_frame"])

    def frame_planes(self, k):
        """Planes of the shard's k-th frame (fewer than n_planes only if the header reports an overflow)."""
        fr = self.frames[k]
        a = int(fr["plane_offset"])
        b = min(a + int(fr["n_planes"]), len(self.planes))
        return self.planes[a:max(a, b)]

    def frame_cylinders(self, k):
        fr = self.frames[k]
        a = int(fr["cylinder_offset"])
        b = min(a + int(fr["n_cylinders"]), len(self.cylinders))
        return self.cylinders[a:max(a, b)]


def unpack_gathered(buf, world, layout):
    """world x bytes_per_rank gathered bytes -> list of Shard, rank order."""
    buf = np.ascontiguousarray(np.frombuffer(buf, dtype=np.uint8) if not isinstance(buf, np.ndarray) else buf).reshape(-1)
    n = layout["bytes_per_rank"]
    assert buf.size == world * n
    return [Shard(buf[r * n:(r + 1) * n], layout) for r in range(world)]


def primitives_by_frame(shards):
    """{global frame index: (planes, cylinders)} over all shards -- what one unsharded run would have produced."""
    out = {}
    for sh in shards:
        for k in range(len(sh.frames)):
            out[sh.first_frame + k] = (sh.frame_planes(k), sh.frame_cylinders(k))
    return out


def broadcast_unique_id(make_id, rank, group=None, device=None):
    """Rank 0 makes the RCCL unique id (cape_comm_unique_id); every rank ends up with the same 128 bytes.
    torch.distributed is only the messenger here (any byte transport would do)."""
    import torch
    import torch.distributed as dist

    t = torch.zeros(128, dtype=torch.uint8, device=device)
    if rank == 0:
        t.copy_(torch.frombuffer(bytearray(make_id()), dtype=torch.uint8))
    dist.broadcast(t, src=0, group=group)
    return bytes(t.cpu().numpy().tobytes())


def all_gather_bytes(local, world, group=None):
    """torch.distributed all-gather of equal-sized uint8 tensors (the CPU tests' stand-in for ncclAllGather)."""
    import torch
    import torch.distributed as dist

    out = torch.empty(world * local.numel(), dtype=torch.uint8, device=local.device)
    dist.all_gather_into_tensor(out, local.contiguous().view(-1), group=group)
    return out
