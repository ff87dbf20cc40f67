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

extern "C" int lm_detector_exchange_pack(lm_detector* d, void* send_block, int capacity) {
    if (!d || !send_block) return lm_set_error(LM_ERR_INVALID, "null argument");
    if (!valid_capacity(capacity)) return lm_set_error(LM_ERR_INVALID, "capacity must be a power of two in [256, %u]", kXchgMaxCapacity);
    if (d->n_submitted == d->n_collected) return lm_set_error(LM_ERR_INVALID, "no frame in flight: call lm_detector_submit first");
    HIP_TRY(hipSetDevice(d->device));
    int rc = ensure_exchange(d);
    if (rc) return rc;
    const int slot = (int)((d->n_submitted - 1) % lm_detector::kSlots);            // the frame just submitted
    if (d->xchg.state[slot] != 0) return lm_set_error(LM_ERR_INVALID, "the frame in flight was packed already");
    if ((rc = lm_launch_pending(d))) return rc;                                     // the exchange follows the frame's kernels in stream order: no waiting for a batch to fill
    lm_detector::Slot& sl = d->slot[slot];
    hipStream_t xs = d->xchg.stream;
    HIP_TRY(hipStreamWaitEvent(xs, d->slot[sl.leader].done, 0));                    // records + counters of this frame (of its batch) are final
    if ((rc = d->xchg.d_runs.ensure(kXchgMaxCapacity))) return rc;
    launch_exchange_pack(d->d_distinct_keys.p + (size_t)d->buf_cand_cap * slot, d->d_final.p + 8 * (size_t)slot, d->buf_cand_cap,
                         (uint32_t)capacity, d->xchg.d_runs.p, (uint32_t*)send_block, xs);
    HIP_TRY(hipGetLastError());
    d->xchg.state[slot] = 1;
    d->xchg.cap[slot] = capacity;
    return LM_OK;
}

extern "C" int lm_detector_exchange_merge(lm_detector* d, const void* recv_blocks, int world, int capacity) {
    if (!d || !recv_blocks || world < 1) return lm_set_error(LM_ERR_INVALID, "bad argument");
    if (d->n_submitted == d->n_collected) return lm_set_error(LM_ERR_INVALID, "no frame in flight");
    const int slot = (int)((d->n_submitted - 1) % lm_detector::kSlots);
    if (d->xchg.state[slot] != 1 || d->xchg.cap[slot] != capacity)
        return lm_set_error(LM_ERR_INVALID, "lm_detector_exchange_pack(capacity %d) has to precede the merge of a frame", capacity);
    HIP_TRY(hipSetDevice(d->device));
    const size_t words = (size_t)kXchgHeaderWords + (size_t)world * capacity * 5;
    int rc = d->xchg.d_merged[slot].ensure(words);                                   // the slot's previous frame was collected: buffers are idle
    if (rc) return rc;
    if (d->xchg.h_words[slot] < words) {
        if (d->xchg.h_merged[slot]) (void)hipHostFree(d->xchg.h_merged[slot]);
        d->xchg.h_merged[slot] = nullptr; d->xchg.h_words[slot] = 0;
        HIP_TRY(hipHostMalloc((void**)&d->xchg.h_merged[slot], words * sizeof(int32_t), hipHostMallocDefault));
        d->xchg.h_words[slot] = words;
    }
    hipStream_t xs = d->xchg.stream;
    launch_exchange_merge((const uint32_t*)recv_blocks, world, (uint32_t)capacity, d->xchg.d_merged[slot].p, xs);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(d->xchg.h_merged[slot], d->xchg.d_merged[slot].p, words * sizeof(int32_t), hipMemcpyDeviceToHost, xs));
    HIP_TRY(hipEventRecord(d->xchg.done[slot], xs));
    d->xchg.state[slot] = 2;
    d->xchg.world[slot] = world;
    return LM_OK;
}

static int exchange_collect(lm_detector* d, lm_match* dst, size_t dst_capacity, lm_match** out, size_t* n_out, int* failed) {
    *n_out = 0; *failed = 0;
    if (d->n_submitted == d->n_collected) return lm_set_error(LM_ERR_INVALID, "no frame in flight");
    const int slot = (int)(d->n_collected % lm_detector::kSlots);
    if (d->xchg.state[slot] != 2) return lm_set_error(LM_ERR_INVALID, "the oldest frame in flight was not exchanged (pack + merge)");
    HIP_TRY(hipSetDevice(d->device));
    HIP_TRY(hipEventSynchronize(d->xchg.done[slot]));                                // the merged list is in pinned memory
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
