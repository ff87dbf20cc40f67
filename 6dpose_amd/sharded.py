"""Multi-GPU template-bank sharding: one process per GPU, every rank holds the whole
frame and searches a contiguous slice of the selected template pyramids (independent units, no
data-path collective), then ONE exchange step: an all-gather of the per-rank match records over
RCCL/xGMI (`torch.distributed`, backend "nccl" on GPUs, "gloo" in the CPU tests), followed by the
canonical merge (LL.cpp:1771-1776 semantics) on every rank.

The records gathered are the PRE-unique lists: Match::operator== ignores template_id while the sort
key contains it (LL.h:234-246), so unique-ing per shard first could drop entries that are not
adjacent in the global order.  Messages are tiny (20 B per match), i.e. latency-bound; counts are
gathered first, then records padded to the maximum.
"""
from __future__ import annotations

from typing import Optional, Sequence

import numpy as np

import linemodLevelup_pybind as lm


def gather_records(local: np.ndarray, device=None, group=None, force: bool = False) -> np.ndarray:
    """all-gather of MATCH_DTYPE records from every rank (concatenated in rank order).  `force` runs
    the collective even with a single rank (used to exercise the RCCL path on a 1-GPU box)."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return local
    if dist.get_world_size(group) == 1 and not force:
        return local
    world = dist.get_world_size(group)
    dev = torch.device("cpu") if device is None else torch.device(device)
    n_local = torch.tensor([len(local)], dtype=torch.int64, device=dev)
    counts = [torch.zeros(1, dtype=torch.int64, device=dev) for _ in range(world)]
    dist.all_gather(counts, n_local, group=group)
    counts = [int(c.item()) for c in counts]
    nmax = max(counts)
    if nmax == 0:
        return np.zeros(0, lm.MATCH_DTYPE)
    words = lm.MATCH_DTYPE.itemsize // 4                     # 5 x 32-bit fields per record
    buf = np.zeros((nmax, words), np.int32)
    buf[:len(local)] = np.ascontiguousarray(local).view(np.int32).reshape(-1, words)
    mine = torch.from_numpy(buf).to(dev)
    parts = [torch.empty_like(mine) for _ in range(world)]
    dist.all_gather(parts, mine, group=group)
    out = [p.cpu().numpy()[:c].reshape(-1).view(lm.MATCH_DTYPE) for p, c in zip(parts, counts)]
    return np.concatenate(out)


def match_sharded(detector: "lm.Detector", sources, threshold: float, class_ids: Sequence[str] = (), masks=(),
                  device=None, group=None, resident: bool = False) -> np.ndarray:
    """Detector.match across all ranks of the process group: identical, canonically ordered result on
    every rank.  The detector must hold the full bank on every rank (template ids stay global)."""
    import torch.distributed as dist
    rank, world = 0, 1
    if dist.is_available() and dist.is_initialized():
        rank, world = dist.get_rank(group), dist.get_world_size(group)
    detector.setShard(rank, world)
    if not resident:
        detector.setFrame(sources, masks)
    # records without exact duplicates (dropped on the device: std::unique removes them on every rank's merge anyway)
    local = detector.matchResident(threshold, class_ids, sort_unique=False, distinct=True)
    allrec = gather_records(local, device=device, group=group)
    return lm.merge_matches(allrec)
