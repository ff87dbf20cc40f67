"""Multi-GPU template-bank sharding: one process per GPU, every rank holds the whole
frame and searches a contiguous slice of the selected template pyramids (independent units, no
data-path collective), then ONE exchange step: an all-gather of the per-rank match records over
RCCL/xGMI (`torch.distributed`, backend "nccl" on GPUs, "gloo" in the CPU tests), followed by the
canonical merge (LL.cpp:1771-1776 semantics) on every rank.

The records gathered are the PRE-unique lists: Match::operator== ignores template_id while the sort
key contains it (LL.h:234-246), so unique-ing per shard first could drop entries that are not
adjacent in the global order.  Messages are tiny (20 B per match), i.e. latency-bound; counts are
gathered first, then records padded to the maximum.

Two ways to do the exchange:
  * gather_records + lm.merge_matches: records through host memory, sort on every rank's host.  Simple, used by the CPU
    (gloo) tests and as the fall-back; its cost grows with the number of ranks (every host sorts every rank's records).
  * DeviceExchange: the exchange as device work — each rank sorts its records in LDS, one all_gather_into_tensor of
    fixed-size blocks on the detector's exchange stream (RCCL reads and writes HBM directly), a ranking merge kernel, and
    the host only copies the result out.  No host synchronisation between submit and collect, so with frames in flight it
    overlaps the matching kernels of the following frames.
"""
from __future__ import annotations

from typing import Optional, Sequence

import os

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


class DeviceExchange:
    """Sharded Detector.match with the exchange on the device (lm_detector_exchange_*; kernels in csrc/exchange.hip).

        ex = DeviceExchange(detector, device)            # after init_process_group; detector.setShard is done here
        ex.submit(threshold, class_ids)                  # up to lm_detector_max_in_flight() frames in flight
        records = ex.collect()                           # oldest frame: canonical list, identical on every rank

    collect() returns None when a rank had more distinct records than `capacity` or overflowed its candidate buffer —
    on every rank alike, so all of them can rerun that frame through match_sharded (host path); the capacity is doubled
    for the following frames when that helps.  With a backend that cannot gather device tensors (gloo in the tests)
    the blocks are staged through the host; the kernels are the same."""

    def __init__(self, detector: "lm.Detector", device, group=None, capacity: int = 4096, force: bool = False, shard: bool = True):
        import torch
        import torch.distributed as dist
        self.det, self.group_handle = detector, group
        self.dev = torch.device(device)
        self.world, self.rank = 1, 0
        self.dist = dist if (dist.is_available() and dist.is_initialized()) else None
        if self.dist is not None:
            self.world, self.rank = dist.get_world_size(group), dist.get_rank(group)
        self.collective = self.dist is not None and (self.world > 1 or force)
        self.device_collective = self.collective and dist.get_backend(group) == "nccl"
        # the all-gather itself: issued by the C library (lm_exchange_allgather: its own RCCL communicator on the exchange stream — what a C++ caller
        # of the library uses, no Python per group of frames on the data path beyond this call) when RCCL loads; else torch.distributed's
        self.comm = None
        if self.device_collective and os.environ.get("LM_EXCHANGE_COLLECTIVE", "c") != "torch" and lm.Comm.available():
            import torch as _t
            ident = [lm.Comm.unique_id() if self.rank == 0 else None]
            dist.broadcast_object_list(ident, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
            self.comm = lm.Comm(ident[0], self.rank, self.world, device=_t.device(device).index or 0)
        if shard:                      # shard=False: the caller partitioned the bank itself (e.g. whole classes per rank)
            detector.setShard(self.rank, self.world)
        detector.setAsyncCollect(False)   # the merged list of all ranks comes from the device exchange: no per-rank host list to prepare
        self.stream = torch.cuda.ExternalStream(detector.exchangeStream(), device=self.dev)
        self.slots = lm.load_library().lm_detector_max_in_flight()
        self.queue = []                 # numbers of the frames submitted whose exchange is not enqueued yet
        self.group = None
        self._alloc(capacity)

    def _alloc(self, capacity: int):
        import torch
        self.capacity = capacity
        nbytes = lm.load_library().lm_exchange_block_bytes(capacity)
        if nbytes == 0:
            raise ValueError("capacity must be a power of two in [256, %d]" % lm.load_library().lm_exchange_max_capacity())
        self.block_bytes = nbytes
        # one send / receive buffer per GROUP of frames in flight: the blocks of up to `group` consecutive frames travel in ONE all-gather.
        # The group size has to be the SAME on every rank (it sets the size and the number of the collectives): the smallest getBatch() of
        # all ranks, agreed once (a rank with another LM_FRAME_BATCH / setBatch would otherwise hang the all-gather: ADVICE r03); a later
        # setBatch() on the detector changes the kernel batches, not the exchange groups.
        if getattr(self, "group", None) is None:
            g = max(1, min(self.det.getBatch(), self.slots))
            if self.collective:
                import torch.distributed as dist
                t = torch.tensor([g], dtype=torch.int32, device=self.dev if self.device_collective else "cpu")
                dist.all_reduce(t, op=dist.ReduceOp.MIN, group=self.group_handle)
                g = int(t.item())
            self.group = g
        self.sets = self.slots // self.group + 2
        self.send = [torch.zeros(nbytes * self.group, dtype=torch.uint8, device=self.dev) for _ in range(self.sets)]
        self.recv = [torch.zeros(nbytes * self.group * self.world, dtype=torch.uint8, device=self.dev) for _ in range(self.sets)]
        # views for every group size, made once (slicing a tensor and asking for its data_ptr cost ~10 us per frame in the loop)
        self.views = [[(s[:n * nbytes], r[:n * nbytes * self.world]) for n in range(self.group + 1)] for s, r in zip(self.send, self.recv)]
        self.ptrs = [(s.data_ptr(), r.data_ptr()) for s, r in zip(self.send, self.recv)]
        self.next_set = 0

    def submit(self, threshold: float, class_ids: Sequence[str] = (), frame=None) -> None:
        """frame=None: the detector's current frame (setFrame / selectFrame); frame=(rgb, depth): a new host frame through
        the live-stream ingest (Detector.submitFrame).  Streamed frames share their kernel launches (a batch is launched when it
        is full or the GPU's queue runs short), and their exchange — sort, all-gather, merge on the exchange stream — is enqueued
        for a whole GROUP of consecutive frames at a time, once they have been launched, with one collective per group: `_pump`.
        Which frames form a group depends only on the sequence of submit / collect calls, never on GPU timing, so every rank
        issues the same collectives in the same order."""
        frame_no = self.det.framesSubmitted()
        if frame is None:
            self.det.submit(threshold, class_ids)
        else:
            self.det.submitFrame(frame, threshold, class_ids)
        self.queue.append(frame_no)
        self._pump(False)

    def _pump(self, force: bool) -> None:
        """Exchange of the frames waiting in self.queue, head first.  Not forced: every FULL group whose frames have all been
        launched.  Forced (collect() needs the oldest frame and it is still queued): ONE group from the head — full if that many
        frames have been submitted, else all there are.  Group boundaries so depend only on the numbers of frames submitted at the
        calls, which are the same on every rank; GPU timing only decides WHEN a rank enqueues a group."""
        while len(self.queue) >= self.group and self.queue[self.group - 1] < self.det.framesLaunched():
            self._exchange_group(self.group)
        if force and self.queue and self.queue[0] <= self.det.framesCollected():   # (still queued after the full groups above)
            n = min(self.group, len(self.queue))
            if self.queue[n - 1] >= self.det.framesLaunched():
                self.det.flush()
            self._exchange_group(n)

    def _exchange_group(self, n: int) -> None:
        import torch
        frames = self.queue[:n]
        del self.queue[:n]
        k = self.next_set
        self.next_set = (k + 1) % self.sets
        cap = self.capacity
        send, recv = self.views[k][n]
        send_ptr, recv_ptr = self.ptrs[k]
        if self.comm is not None:                                # pack, all-gather and merge enqueued by the library
            self.det.exchangeGroup(self.comm, frames[0], n, send_ptr, recv_ptr, cap)
            return
        self.det.exchangePackGroup(frames[0], n, send_ptr, cap)
        if not self.collective:
            recv_ptr = send_ptr
        elif self.device_collective:
            with torch.cuda.stream(self.stream):                 # RCCL orders itself after the pack kernels / before the merges
                self.dist.all_gather_into_tensor(recv, send, group=self.group_handle)
        else:                                                    # backend without device collectives: stage through the host
            with torch.cuda.stream(self.stream):
                mine = send.cpu()
                parts = [torch.empty_like(mine) for _ in range(self.world)]
                self.dist.all_gather(parts, mine, group=self.group_handle)
                recv.copy_(torch.cat(parts), non_blocking=False)
        self.det.exchangeMergeGroup(frames[0], n, recv_ptr, self.world, cap)

    def collect(self, into: Optional[np.ndarray] = None) -> Optional[np.ndarray]:
        """Oldest frame in flight.  `into` (MATCH_DTYPE, world * capacity records) avoids the allocation and a copy; the
        result is then a view of it."""
        self._pump(True)                                           # forced only if the oldest frame still waits for its group
        out, failed = self.det.exchangeCollect() if into is None else self.det.exchangeCollectInto(into)
        if out is None and failed > self.capacity:               # a run did not fit: larger blocks from the next submit on, if they exist
            cap = self.capacity
            while cap < failed:
                cap *= 2
            if cap <= lm.load_library().lm_exchange_max_capacity():
                self._pending_capacity = cap
            else:
                import warnings
                warnings.warn("DeviceExchange: a rank produced %d distinct records, more than the largest block (%d): such frames take the host path"
                              % (failed, lm.load_library().lm_exchange_max_capacity()))
        return out

    def grow_if_needed(self) -> None:
        """Call with no frame in flight after collect() returned None: applies the larger capacity (same on every rank)."""
        cap = getattr(self, "_pending_capacity", None)
        if cap:
            self._pending_capacity = None
            self._alloc(cap)


def match_sharded(detector: "lm.Detector", sources, threshold: float, class_ids: Sequence[str] = (), masks=(),
                  device=None, group=None, resident: bool = False, exchange: Optional[DeviceExchange] = None,
                  shard: bool = True) -> np.ndarray:
    """Detector.match across all ranks of the process group: identical, canonically ordered result on
    every rank.  The detector must hold the full bank on every rank (template ids stay global)."""
    import torch.distributed as dist
    rank, world = 0, 1
    if dist.is_available() and dist.is_initialized():
        rank, world = dist.get_rank(group), dist.get_world_size(group)
    if shard:                # shard=False: the caller partitioned the bank itself (whole classes per rank) — every rank searches
        detector.setShard(rank, world)   # all the templates it holds; class positions and template ids are global either way
    if not resident:
        detector.setFrame(sources, masks)
    if exchange is not None:                                     # sort / gather / merge on the device
        exchange.submit(threshold, class_ids)
        out = exchange.collect()
        if out is not None:
            return out
        exchange.grow_if_needed()                                # every rank takes the host path for this frame
    # records without exact duplicates (dropped on the device: std::unique removes them on every rank's merge anyway)
    local = detector.matchResident(threshold, class_ids, sort_unique=False, distinct=True)
    allrec = gather_records(local, device=device, group=group)
    return lm.merge_matches(allrec)
