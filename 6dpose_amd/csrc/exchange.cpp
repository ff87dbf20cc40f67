// Multi-GPU exchange of match records, host side (kernels in exchange.hip).  Per frame and rank:
//
//   lm_detector_submit                       matching kernels of the frame (streams `stream` / `mstream`)
//   lm_detector_exchange_pack(send, cap)     exchange stream: wait for the frame's records, sort them into this rank's block
//   <all-gather of the blocks>               the caller's collective (RCCL through torch.distributed), enqueued on the
//                                            exchange stream (lm_detector_exchange_stream) — no host synchronisation
//   lm_detector_exchange_merge(recv, W, cap) exchange stream: merge the W sorted runs, mark what std::unique drops, copy to
//                                            pinned memory
//   lm_detector_exchange_collect             host: wait, skip the marked records -> the frame's Detector::match result
//
// Nothing here blocks the host until collect, and the exchange stream only ever waits for the frame it works on, so with
// frames in flight the sort / gather / merge of frame k overlaps the matching kernels of frame k+1.
#include <string.h>

#include <algorithm>

#include "detector_internal.h"

using namespace lm;

static_assert(sizeof(lm_match) == 20, "lm_match is five 32-bit fields");

static int ensure_exchange(lm_detector* d) {
    if (d->xchg.done[0]) return LM_OK;                                               // the stream itself is created with the detector
    for (int a = 0; a < lm_detector::kSlots; ++a) HIP_TRY(hipEventCreateWithFlags(&d->xchg.done[a], hipEventDisableTiming));
    return LM_OK;
}

static bool valid_capacity(int capacity) {
    return capacity >= 256 && (uint32_t)capacity <= kXchgMaxCapacity && (capacity & (capacity - 1)) == 0;
}

extern "C" void* lm_detector_exchange_stream(lm_detector* d) {
    if (!d) { lm_set_error(LM_ERR_INVALID, "null detector"); return nullptr; }
    if (hipSetDevice(d->device) != hipSuccess || ensure_exchange(d) != LM_OK) return nullptr;
    return (void*)d->xchg.stream;
}

extern "C" int lm_exchange_max_capacity(void) { return (int)kXchgMaxCapacity; }

extern "C" size_t lm_exchange_block_bytes(int capacity) { return valid_capacity(capacity) ? 16 + (size_t)capacity * 16 : 0; }

extern "C" uint64_t lm_detector_frames_submitted(const lm_detector* d) { return d ? d->n_submitted : 0; }
extern "C" uint64_t lm_detector_frames_launched(const lm_detector* d) { return d ? d->n_launched : 0; }
extern "C" uint64_t lm_detector_frames_collected(const lm_detector* d) { return d ? d->n_collected : 0; }

// The exchange of a GROUP of n consecutive frames, first = the number of its first frame in submission order (sharded.DeviceExchange:
// one all-gather carries the blocks of the whole group).  The frames must have been LAUNCHED (streamed frames wait for their batch:
// lm_detector_frames_launched, lm_detector_flush) and not collected yet.  Frame first + i packs into send_blocks + i * block bytes; ONE
// launch of each kernel serves the group.
extern "C" int lm_detector_exchange_pack_group(lm_detector* d, uint64_t first, int n, void* send_blocks, int capacity) {
    if (!d || !send_blocks || n < 1 || n > kMaxBatch) return lm_set_error(LM_ERR_INVALID, "bad argument");
    if (!valid_capacity(capacity)) return lm_set_error(LM_ERR_INVALID, "capacity must be a power of two in [256, %u]", kXchgMaxCapacity);
    if (first < d->n_collected || first + (uint64_t)n > d->n_launched)
        return lm_set_error(LM_ERR_INVALID, "frames %llu..%llu are not all in flight on the device (collected %llu, launched %llu, submitted %llu)", (unsigned long long)first,
                            (unsigned long long)(first + n - 1), (unsigned long long)d->n_collected, (unsigned long long)d->n_launched, (unsigned long long)d->n_submitted);
    HIP_TRY(hipSetDevice(d->device));
    int rc = ensure_exchange(d);
    if (rc) return rc;
    if ((rc = d->xchg.d_runs.ensure((size_t)kXchgMaxCapacity * kMaxBatch))) return rc;
    const size_t blk = lm_exchange_block_bytes(capacity);
    hipStream_t xs = d->xchg.stream;
    XchgGroup G{};
    G.n = n;
    int last_leader = -1;
    for (int i = 0; i < n; ++i) {
        const int slot = (int)((first + (uint64_t)i) % lm_detector::kSlots);
        if (d->xchg.state[slot] != 0) return lm_set_error(LM_ERR_INVALID, "frame %llu was packed already", (unsigned long long)(first + i));
        const int leader = d->slot[slot].leader;
        if (leader != last_leader) { HIP_TRY(hipStreamWaitEvent(xs, d->slot[leader].done, 0)); last_leader = leader; }   // records + counters of the frame's batch are final
        G.f[i].keys = d->d_distinct_keys.p + (size_t)d->buf_cand_cap * slot;
        G.f[i].counters = d->d_final.p + 8 * (size_t)slot;
        G.f[i].runs = d->xchg.d_runs.p + (size_t)kXchgMaxCapacity * i;
        G.f[i].block = (uint32_t*)((uint8_t*)send_blocks + (size_t)i * blk);
    }
    launch_exchange_pack_group(G, d->buf_cand_cap, (uint32_t)capacity, xs);
    HIP_TRY(hipGetLastError());
    for (int i = 0; i < n; ++i) {
        const int slot = (int)((first + (uint64_t)i) % lm_detector::kSlots);
        d->xchg.state[slot] = 1;
        d->xchg.cap[slot] = capacity;
    }
    return LM_OK;
}

// recv_blocks: the group's blocks of rank 0, frame first + i at recv_blocks + i * block bytes; rank j's lie j * rank_stride_bytes further
// (0 = n * block bytes: every rank sent its n blocks back to back).
extern "C" int lm_detector_exchange_merge_group_strided(lm_detector* d, uint64_t first, int n, const void* recv_blocks, int world, int capacity, size_t rank_stride_bytes) {
    if (!d || !recv_blocks || world < 1 || n < 1 || n > kMaxBatch || (rank_stride_bytes & 3)) return lm_set_error(LM_ERR_INVALID, "bad argument");
    if (first < d->n_collected || first + (uint64_t)n > d->n_launched) return lm_set_error(LM_ERR_INVALID, "frames %llu.. are not in flight on the device", (unsigned long long)first);
    const size_t blk = lm_exchange_block_bytes(capacity);
    if (!blk) return lm_set_error(LM_ERR_INVALID, "bad capacity");
    if (!rank_stride_bytes) rank_stride_bytes = (size_t)n * blk;
    HIP_TRY(hipSetDevice(d->device));
    const size_t words = (size_t)kXchgHeaderWords + (size_t)world * capacity * 5;
    XchgGroup G{};
    G.n = n;
    for (int i = 0; i < n; ++i) {
        const int slot = (int)((first + (uint64_t)i) % lm_detector::kSlots);
        if (d->xchg.state[slot] != 1 || d->xchg.cap[slot] != capacity)
            return lm_set_error(LM_ERR_INVALID, "lm_detector_exchange_pack(capacity %d) has to precede the merge of a frame", capacity);
        int rc = d->xchg.d_merged[slot].ensure(words);                                   // the slot's previous frame was collected: buffers are idle
        if (rc) return rc;
        if (d->xchg.h_words[slot] < words) {
            if (d->xchg.h_merged[slot]) (void)hipHostFree(d->xchg.h_merged[slot]);
            d->xchg.h_merged[slot] = nullptr; d->xchg.h_words[slot] = 0;
            HIP_TRY(hipHostMalloc((void**)&d->xchg.h_merged[slot], words * sizeof(int32_t), hipHostMallocDefault));
            d->xchg.h_words[slot] = words;
        }
        G.f[i].recv = (const uint32_t*)((const uint8_t*)recv_blocks + (size_t)i * blk);
        G.f[i].merged = d->xchg.d_merged[slot].p;
        HIP_TRY(hipHostGetDevicePointer((void**)&G.f[i].host, d->xchg.h_merged[slot], 0));
    }
    hipStream_t xs = d->xchg.stream;
    launch_exchange_merge_group(G, world, (uint32_t)capacity, xs, (uint32_t)(rank_stride_bytes / 4));   // merge + copy-out to the pinned buffers
    HIP_TRY(hipGetLastError());
    for (int i = 0; i < n; ++i) {
        // the group's completion on EVERY member slot's own event (a record is cheap): with one event on the first slot, that slot's next
        // frame — 16 frames on — re-recorded it while the later members of this group were still uncollected, and their collect then waited
        // for the newer group
        const int slot = (int)((first + (uint64_t)i) % lm_detector::kSlots);
        HIP_TRY(hipEventRecord(d->xchg.done[slot], xs));
        d->xchg.state[slot] = 2;
        d->xchg.world[slot] = world;
        d->xchg.done_slot[slot] = slot;
    }
    return LM_OK;
}
extern "C" int lm_detector_exchange_merge_group(lm_detector* d, uint64_t first, int n, const void* recv_blocks, int world, int capacity) {
    return lm_detector_exchange_merge_group_strided(d, first, n, recv_blocks, world, capacity, 0);
}

// One frame: a group of one.
extern "C" int lm_detector_exchange_pack_frame(lm_detector* d, uint64_t frame_no, void* send_block, int capacity) {
    return lm_detector_exchange_pack_group(d, frame_no, 1, send_block, capacity);
}
extern "C" int lm_detector_exchange_merge_frame_strided(lm_detector* d, uint64_t frame_no, const void* recv_blocks, int world, int capacity, size_t rank_stride_bytes) {
    return lm_detector_exchange_merge_group_strided(d, frame_no, 1, recv_blocks, world, capacity, rank_stride_bytes ? rank_stride_bytes : lm_exchange_block_bytes(capacity));
}
extern "C" int lm_detector_exchange_merge_frame(lm_detector* d, uint64_t frame_no, const void* recv_blocks, int world, int capacity) {
    return lm_detector_exchange_merge_frame_strided(d, frame_no, recv_blocks, world, capacity, 0);
}

// The frame just submitted (launching whatever waits for its batch): the one-frame-at-a-time form of the two calls above.
extern "C" int lm_detector_exchange_pack(lm_detector* d, void* send_block, int capacity) {
    if (!d) return lm_set_error(LM_ERR_INVALID, "null argument");
    if (d->n_submitted == d->n_collected) return lm_set_error(LM_ERR_INVALID, "no frame in flight: call lm_detector_submit first");
    int rc = lm_launch_pending(d);                                                   // the exchange follows the frame's kernels in stream order
    if (rc) return rc;
    return lm_detector_exchange_pack_frame(d, d->n_submitted - 1, send_block, capacity);
}

extern "C" int lm_detector_exchange_merge(lm_detector* d, const void* recv_blocks, int world, int capacity) {
    if (!d) return lm_set_error(LM_ERR_INVALID, "bad argument");
    if (d->n_submitted == d->n_collected) return lm_set_error(LM_ERR_INVALID, "no frame in flight");
    return lm_detector_exchange_merge_frame(d, d->n_submitted - 1, recv_blocks, world, capacity);
}

static int exchange_collect(lm_detector* d, lm_match* dst, size_t dst_capacity, lm_match** out, size_t* n_out, int* failed) {
    *n_out = 0; *failed = 0;
    if (d->n_submitted == d->n_collected) return lm_set_error(LM_ERR_INVALID, "no frame in flight");
    const int slot = (int)(d->n_collected % lm_detector::kSlots);
    if (d->xchg.state[slot] != 2) return lm_set_error(LM_ERR_INVALID, "the oldest frame in flight was not exchanged (pack + merge)");
    HIP_TRY(hipSetDevice(d->device));
    HIP_TRY(hipEventSynchronize(d->xchg.done[d->xchg.done_slot[slot]]));             // the merged list (of the frame's whole group) is in pinned memory
    d->xchg.state[slot] = 0;
    const int rc = lm_collect_frame(d, -1, nullptr, nullptr);                        // retires the frame (timings, overflow bookkeeping)
    if (rc < 0) return rc;
    const int32_t* h = d->xchg.h_merged[slot];
    const uint32_t flags = (uint32_t)h[1];
    if (rc == 1 && !(flags & kXchgCandOverflow))
        return lm_set_error(LM_ERR_HIP, "candidate overflow seen by the host but not by the exchange");
    if (flags) {                                                                     // every rank reads the same flags: all fall back together
        int need = 0;
        for (int j = 0; j < h[2] && j < kXchgHeaderWords - 8; ++j) need = std::max(need, h[8 + j]);
        *failed = (flags & kXchgRunOverflow) ? std::max(need, 1) : -(int)flags;
        return LM_OK;
    }
    const size_t total = (size_t)h[0];
    if (total > (size_t)d->xchg.world[slot] * (size_t)d->xchg.cap[slot]) return lm_set_error(LM_ERR_HIP, "exchange header inconsistent");
    lm_match* res = dst;
    if (!res) {
        res = (lm_match*)malloc(std::max<size_t>(1, total) * sizeof(lm_match));
        if (!res) return lm_set_error(LM_ERR_INVALID, "out of host memory");
    } else if (dst_capacity < total) {
        return lm_set_error(LM_ERR_INVALID, "destination holds %zu records, the frame has up to %zu", dst_capacity, total);
    }
    const int32_t* rec = h + kXchgHeaderWords;
    size_t w = 0;
    for (size_t i = 0; i < total; ++i, rec += 5) {
        if (rec[3] < 0) continue;                                                    // what std::unique removes (LL.cpp:1772-1774)
        memcpy(&res[w++], rec, sizeof(lm_match));                                    // same five 32-bit fields, same order
    }
    if (out) *out = res;
    *n_out = w;
    return LM_OK;
}

extern "C" int lm_detector_exchange_collect(lm_detector* d, lm_match** out, size_t* n_out, int* failed) {
    if (!d || !out || !n_out || !failed) return lm_set_error(LM_ERR_INVALID, "null argument");
    *out = nullptr;
    return exchange_collect(d, nullptr, 0, out, n_out, failed);
}

extern "C" int lm_detector_exchange_collect_into(lm_detector* d, lm_match* dst, size_t capacity, size_t* n_out, int* failed) {
    if (!d || !dst || !n_out || !failed) return lm_set_error(LM_ERR_INVALID, "null argument");
    return exchange_collect(d, dst, capacity, nullptr, n_out, failed);
}
