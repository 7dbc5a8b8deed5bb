"""Frame-index sharding for multi-GPU runs (one process per GPU).

Frames are independent in the reference (one process per frame, appended to the
.yuv in order, tiff.cpp:440), so the only multi-GPU structure is *which rank
converts which frame* and *where its bytes land*: frame k goes to byte offset
k * frame_bytes of the output.  No data-path collective; the ranks only reduce
their counters at the end (RCCL all-reduce in bench.py, gloo in the CPU tests).
"""
from __future__ import annotations


def frames_for_rank(n_frames: int, rank: int, world: int) -> range:
    """Contiguous block [lo, hi) of frame indices owned by `rank`; blocks differ
    in size by at most one frame and cover 0..n_frames-1 exactly once."""
    if world < 1 or not (0 <= rank < world) or n_frames < 0:
        raise ValueError("bad shard arguments")
    base, extra = divmod(n_frames, world)
    lo = rank * base + min(rank, extra)
    hi = lo + base + (1 if rank < extra else 0)
    return range(lo, hi)


def frame_offset_bytes(frame_index: int, frame_bytes: int) -> int:
    """Byte offset of frame k in the shared .yuv (append order of the reference)."""
    return frame_index * frame_bytes


def reduce_counters(pixels: float, seconds: float, dist=None, device=None):
    """(sum of pixels over ranks, max of seconds over ranks)."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return pixels, seconds
    import torch

    t = torch.tensor([seconds], dtype=torch.float64, device=device)
    p = torch.tensor([pixels], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dist.all_reduce(p, op=dist.ReduceOp.SUM)
    return float(p.item()), float(t.item())
