"""Multi-GPU plumbing: frames shard by contiguous blocks (one process per GPU, no data-path collective inside a
frame) and the per-frame primitive lists (cape_primitive_summary, 1296 B) are exchanged with ONE all-gather per
batch -- RCCL over xGMI on the GPU box (backend "nccl"), gloo in the CPU tests."""
import numpy as np


def shard_range(n_frames, rank, world):
    """Contiguous block of frames owned by `rank`: the first (n % world) ranks get one extra frame."""
    q, r = divmod(n_frames, world)
    start = rank * q + min(rank, r)
    return start, start + q + (1 if rank < r else 0)


def gather_summaries(local_bytes, world, group=None):
    """All-gather equal-sized uint8 tensors of packed cape_primitive_summary records; returns world x nbytes."""
    import torch
    import torch.distributed as dist

    out = torch.empty((world, local_bytes.numel()), dtype=torch.uint8, device=local_bytes.device)
    dist.all_gather_into_tensor(out.view(-1), local_bytes.contiguous().view(-1), group=group)
    return out


def gather_ragged(local_bytes, counts, record_bytes, group=None):
    """Shards of different length (n_frames % world != 0): pad to the longest shard, gather, strip the padding."""
    import torch

    world = len(counts)
    longest = max(counts)
    pad = torch.zeros(longest * record_bytes, dtype=torch.uint8, device=local_bytes.device)
    pad[: local_bytes.numel()] = local_bytes.view(-1)
    g = gather_summaries(pad, world, group)
    return torch.cat([g[r, : counts[r] * record_bytes] for r in range(world)])


def summaries_from_bytes(buf):
    from . import SUMMARY_DTYPE

    return np.frombuffer(bytes(buf), dtype=SUMMARY_DTYPE)
