// On-device NMS + top-K over the matches of one frame: the caller-side loop of the
// reference driver (linemod_and_levelup_test.py:331-352 — boxes x..x+width, y..y+height of the matched
// template, numpy `nms` :34-61 with IoU > 0.5 suppressing, then the first few kept detections go to
// poseRefine) without the device-to-host copy of thousands of matches and the Python greedy loop.
//
// Semantics = lm_nms_boxes(canonical match list) stopped after top_k kept boxes, exactly:
//   * candidates are the entries of the list Detector::match returns (canonical sort + adjacent-unique,
//     SURVEY A12).  Nothing is sorted here: an entry R is removed by std::unique iff its immediate
//     predecessor in the canonical order has the same (x, y, similarity, class) — found with one
//     arg-max pass over the records, and only for the few entries that are about to be kept;
//   * visiting order of the greedy loop: similarity descending, ties by HIGHER canonical index first
//     (scores.argsort()[::-1], the rule lm_nms_boxes documents);
//   * IoU in double with the +1 pixel convention, suppressed iff !(IoU <= thresh).
// One 1024-thread workgroup.  Records identical in every field (several coarse candidates of a template refined
// to one position) are dropped first with a hash; the distinct ones (a few thousand) live in LDS.  A round =
// arg-max of the live records, predecessor test, suppression: three passes over LDS.
#include "lm_kernels.h"

namespace lm {

constexpr int kNmsWG = 1024;

// packed record: .x = x | y << 16 (int16 each), .y = w | h << 16, .z = similarity bits (valid: >= 0 float), .w = tid | cls << 24 | dead << 31
struct NmsKey {           // canonical order key of a record (ascending = earlier in the list) + its slot
    uint32_t sim;         // similarity bits (non-negative floats order like unsigned)
    int32_t tid, cls, y, x, slot;
};

static __device__ __forceinline__ NmsKey key_of(const int4 r, int slot) {
    NmsKey k;
    k.sim = (uint32_t)r.z;
    k.tid = r.w & 0xFFFFFF; k.cls = (r.w >> 24) & 0x7F;
    k.x = (int)(int16_t)(r.x & 0xFFFF); k.y = (int)(int16_t)((uint32_t)r.x >> 16);
    k.slot = slot;
    return k;
}
// a before b in the canonical order (similarity desc, template_id, class, y, x asc; slot separates identical records)
static __device__ __forceinline__ bool canon_before(const NmsKey& a, const NmsKey& b) {
    if (a.sim != b.sim) return a.sim > b.sim;
    if (a.tid != b.tid) return a.tid < b.tid;
    if (a.cls != b.cls) return a.cls < b.cls;
    if (a.y != b.y) return a.y < b.y;
    if (a.x != b.x) return a.x < b.x;
    return a.slot < b.slot;
}
// a is visited before b by the greedy loop: higher similarity, then the LATER canonical entry
static __device__ __forceinline__ bool visit_before(const NmsKey& a, const NmsKey& b) {
    if (a.sim != b.sim) return a.sim > b.sim;
    return canon_before(b, a);
}
// The two orders as 128-bit unsigned integers (compared hi word first, larger = preferred), so that a selection exchanges four
// dwords per step instead of six fields and a chain of comparisons:
//   visit order  (higher similarity, then the LATER canonical entry): (sim, tid, cls | y, x, slot), all ascending = preferred
//   canonical    (similarity desc, template_id, class, y, x, slot asc): the same with the similarity inverted; the LATEST
//                entry before R is the largest such key below R's.
struct Key128 { unsigned long long hi, lo; };
static __device__ __forceinline__ Key128 pack_key(const NmsKey& k, bool canonical) {
    Key128 p;
    const uint32_t sim = canonical ? ~k.sim : k.sim;
    p.hi = ((unsigned long long)sim << 32) | ((unsigned long long)(uint32_t)k.tid << 8) | (unsigned long long)(uint32_t)k.cls;
    p.lo = ((unsigned long long)(uint32_t)(k.y + 32768) << 48) | ((unsigned long long)(uint32_t)(k.x + 32768) << 32) | (unsigned long long)(uint32_t)k.slot;
    return p;
}
static __device__ __forceinline__ bool key_less(const Key128& a, const Key128& b) { return a.hi < b.hi || (a.hi == b.hi && a.lo < b.lo); }
static __device__ __forceinline__ NmsKey unpack_key(const Key128& p, bool canonical) {
    NmsKey k;
    const uint32_t sim = (uint32_t)(p.hi >> 32);
    k.sim = canonical ? ~sim : sim;
    k.tid = (int32_t)((p.hi >> 8) & 0xFFFFFF); k.cls = (int32_t)(p.hi & 0xFF);
    k.y = (int32_t)((p.lo >> 48) & 0xFFFF) - 32768; k.x = (int32_t)((p.lo >> 32) & 0xFFFF) - 32768; k.slot = (int32_t)(uint32_t)p.lo;
    return k;
}
// The larger of a lane's key and the key of the lane a DPP pattern pairs it with; after the four patterns (swap 1, swap 2, mirror of the
// half row, mirror of the row) every lane holds the maximum of its row of 16.  (Was: six butterfly steps through ds_bpermute, four dwords
// each — 24 trips through the LDS crossbar per selection, two selections per round of the greedy loop.)
template <int CTRL>
static __device__ __forceinline__ Key128 dpp_key128(const Key128& k) {
    auto mv = [](uint32_t v) { return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xF, 0xF, false); };
    Key128 o;
    o.hi = ((unsigned long long)mv((uint32_t)(k.hi >> 32)) << 32) | mv((uint32_t)k.hi);
    o.lo = ((unsigned long long)mv((uint32_t)(k.lo >> 32)) << 32) | mv((uint32_t)k.lo);
    return o;
}
static __device__ __forceinline__ Key128 row_max_key128(Key128 k) {
    Key128 o;
    o = dpp_key128<0xB1>(k); if (key_less(k, o)) k = o;
    o = dpp_key128<0x4E>(k); if (key_less(k, o)) k = o;
    o = dpp_key128<0x141>(k); if (key_less(k, o)) k = o;
    o = dpp_key128<0x140>(k); if (key_less(k, o)) k = o;
    return k;
}
static __device__ __forceinline__ Key128 readlane_key128(const Key128& k, const int l) {
    auto rl = [&](uint32_t v) { return (uint32_t)__builtin_amdgcn_readlane((int)v, l); };
    Key128 o;
    o.hi = ((unsigned long long)rl((uint32_t)(k.hi >> 32)) << 32) | rl((uint32_t)k.hi);
    o.lo = ((unsigned long long)rl((uint32_t)(k.lo >> 32)) << 32) | rl((uint32_t)k.lo);
    return o;
}

// Workgroup-wide maximum of the packed keys (`valid` = this thread has one); slot < 0 in the result = none.  Rows of 16 lanes by DPP, the four
// rows of a wave through readlane, the 16 wave results through LDS — every wave reduces them itself, so two barriers do.
static __device__ __forceinline__ NmsKey block_select_max(const NmsKey& mine, const bool canonical, Key128* s_keys /*[kNmsWG / 64]*/) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    Key128 k = pack_key(mine, canonical);
    const bool have = mine.slot >= 0;
    if (!have) { k.hi = 0; k.lo = 0; }
    // validity travels as an extra flag in bit 31 of lo's slot field: slots are < 2^31, the flag makes every valid key larger than "none"
    k.lo = have ? (k.lo | 0x80000000ull) : 0ull;
    k = row_max_key128(k);
    Key128 w = readlane_key128(k, 0);
#pragma unroll
    for (int r = 1; r < 4; ++r) { const Key128 o = readlane_key128(k, 16 * r); if (key_less(w, o)) w = o; }
    __syncthreads();                                                 // (the previous selection's results have been read)
    if (lane == 0) s_keys[wave] = w;
    __syncthreads();
    Key128 best = lane < kNmsWG / 64 ? s_keys[lane] : Key128{0, 0};
    best = readlane_key128(row_max_key128(best), 0);
    NmsKey r;
    if (!(best.lo & 0x80000000ull)) { r.slot = -1; r.sim = 0; r.tid = r.cls = r.x = r.y = 0; return r; }
    best.lo &= ~0x80000000ull;
    return unpack_key(best, canonical);
}

constexpr int kNmsLds = 8192;      // distinct records held in LDS (128 KiB); more than that stay in the HBM scratch

// Greedy loop over m distinct records (rec in LDS or HBM).
template <typename RecPtr>
static __device__ __forceinline__ int nms_rounds(RecPtr rec, const int m, const int top_k, const double thresh, Key128* s_keys,
                                                 TopkSel* __restrict__ sel) {
    const int tid = threadIdx.x;
    int kept = 0;
    while (kept < top_k) {
        // (1) next entry the greedy loop visits
        NmsKey mine; mine.slot = -1; mine.sim = 0; mine.tid = mine.cls = mine.x = mine.y = 0;
        for (int i = tid; i < m; i += kNmsWG) {
            const int4 r = rec[i];
            if (r.w < 0) continue;                                              // suppressed / removed
            const NmsKey k = key_of(r, i);
            if (mine.slot < 0 || visit_before(k, mine)) mine = k;
        }
        const NmsKey R = block_select_max(mine, false, s_keys);
        if (R.slot < 0) break;
        // (2) std::unique: R disappears iff its canonical predecessor (over ALL entries of the frame) equals it in (x, y, similarity, class)
        NmsKey pm; pm.slot = -1; pm.sim = 0; pm.tid = pm.cls = pm.x = pm.y = 0;
        for (int i = tid; i < m; i += kNmsWG) {
            const NmsKey k = key_of(rec[i], i);
            if (canon_before(k, R) && (pm.slot < 0 || canon_before(pm, k))) pm = k;
        }
        const NmsKey P = block_select_max(pm, true, s_keys);
        const bool dup = P.slot >= 0 && P.sim == R.sim && P.cls == R.cls && P.x == R.x && P.y == R.y;
        const int4 rr = rec[R.slot];
        __syncthreads();
        if (dup) {
            if (tid == 0) rec[R.slot].w = rr.w | (int)0x80000000;
            __syncthreads();
            continue;
        }
        // (3) keep R, suppress what overlaps it (R itself has IoU 1)
        const double x1 = (double)R.x, y1 = (double)R.y;
        const double x2 = x1 + (double)(rr.y & 0xFFFF), y2 = y1 + (double)((uint32_t)rr.y >> 16);
        const double ai = (x2 - x1 + 1) * (y2 - y1 + 1);
        for (int i = tid; i < m; i += kNmsWG) {
            const int4 r = rec[i];
            if (r.w < 0) continue;
            const double bx1 = (double)(int16_t)(r.x & 0xFFFF), by1 = (double)(int16_t)((uint32_t)r.x >> 16);
            const double bx2 = bx1 + (double)(r.y & 0xFFFF), by2 = by1 + (double)((uint32_t)r.y >> 16);
            const double xx1 = fmax(x1, bx1), yy1 = fmax(y1, by1), xx2 = fmin(x2, bx2), yy2 = fmin(y2, by2);
            const double w = fmax(0.0, xx2 - xx1 + 1), h = fmax(0.0, yy2 - yy1 + 1);
            const double inter = w * h;
            const double aj = (bx2 - bx1 + 1) * (by2 - by1 + 1);
            const double ovr = __ddiv_rn(inter, ai + aj - inter);
            if (!(ovr <= thresh)) rec[i].w = r.w | (int)0x80000000;
        }
        if (tid == 0) {
            TopkSel o;
            o.x = R.x; o.y = R.y; o.similarity = __int_as_float((int)R.sim); o.work = -1;
            o.class_index = R.cls; o.template_id = R.tid;
            o.width = rr.y & 0xFFFF; o.height = (int)((uint32_t)rr.y >> 16);
            sel[kept] = o;
        }
        ++kept;
        __syncthreads();
    }
    return kept;
}

// Pack + exact-duplicate removal (k_nms_pack, the whole chip), then the greedy loop (k_topk_nms, one workgroup).  Several coarse
// candidates of one template often refine to the same position: those records are identical in every output field, std::unique
// removes all but one, so they are dropped up front with an open-addressing hash on (x, y, template, class) (`table`: 64-bit
// slots preset to ~0).  A record is a chain of five dependent HBM reads and an HBM atomic: one workgroup walking 10k of them
// 1024 at a time took ~55 us; spread over the chip the chain is paid once.  The distinct records land in `rec` in whatever order
// the atomics resolve — the order does not matter: no two of them are equal in (x, y, template, class), so no comparison of
// the greedy loop ever gets as far as the slot.  ctr[0] = number of records - 1, ctr[1] = "bad field" marker (both preset to ~0).
__global__ void __launch_bounds__(256)
k_nms_pack(const Candidate* __restrict__ matches, const unsigned long long* __restrict__ counters, uint32_t cap,
           const int32_t* __restrict__ work_pyramids, const int32_t* __restrict__ work_cls, const int32_t* __restrict__ work_tid,
           const TemplEntry* __restrict__ entries, int levels, const int32_t* __restrict__ class_base, const int32_t* __restrict__ view_wh,
           int num_views, int4* __restrict__ rec, unsigned long long* __restrict__ table, uint32_t table_mask, uint32_t* __restrict__ ctr) {
    const int lane = threadIdx.x & 63;
    const unsigned long long nc = counters[0];
    const int n = (int)(nc < cap ? nc : cap);
    const int stride = gridDim.x * blockDim.x;
    for (int i0 = blockIdx.x * blockDim.x; i0 < n; i0 += stride) {
        const int i = i0 + threadIdx.x;
        bool keep = false;
        int4 r = make_int4(0, 0, 0, 0);
        if (i < n) {
            const Candidate c = matches[i];
            if (c.work >= 0) {
                const TemplEntry e = entries[(size_t)work_pyramids[c.work] * levels];  // level 0: the template's size
                const int t = work_tid[c.work], cl = work_cls[c.work];
                int bw = e.width, bh = e.height;                                      // NMS box: the template's size, or the
                if (class_base && view_wh && cl >= 0 && cl < 128) {                   // caller's aTemplateInfo width / height
                    const int base = class_base[cl];
                    const int v = base + t;
                    if (base >= 0 && v >= 0 && v < num_views && view_wh[2 * v] >= 0) { bw = view_wh[2 * v]; bh = view_wh[2 * v + 1]; }
                }
                if (t < 0 || t >= (1 << 24) || cl < 0 || cl >= 128 || c.x < -32768 || c.x > 32767 || c.y < -32768 || c.y > 32767 ||
                    bw < 0 || bw > 65535 || bh < 0 || bh > 65535 || c.score < 0.f || !(c.score == c.score)) {
                    ctr[1] = 0;
                } else {
                    r.x = (c.x & 0xFFFF) | (c.y << 16);
                    r.y = (bw & 0xFFFF) | (bh << 16);
                    r.z = __float_as_int(c.score);
                    r.w = t | (cl << 24);
                    const unsigned long long key = ((unsigned long long)(uint32_t)r.x << 32) | (uint32_t)r.w;
                    unsigned long long hsh = key * 0x9E3779B97F4A7C15ull;
                    uint32_t slot = (uint32_t)(hsh >> 40) & table_mask;
                    for (;;) {                                                     // linear probing; the table is never full (>= 2 x cap slots)
                        const unsigned long long prev = atomicCAS(&table[slot], ~0ull, key);
                        if (prev == ~0ull) { keep = true; break; }
                        if (prev == key) break;                                    // an identical record is already in
                        slot = (slot + 1) & table_mask;
                    }
                }
            }
        }
        const unsigned long long mask = __ballot(keep);                            // one HBM atomic per wave reserves the slots
        if (mask) {
            uint32_t base = 0;
            if (lane == 0) base = atomicAdd(&ctr[0], (uint32_t)__popcll(mask)) + 1u;   // the counter starts at ~0
            base = (uint32_t)__shfl((int)base, 0, 64);
            if (keep) rec[base + __popcll(mask & ((1ull << lane) - 1ull))] = r;
        }
    }
}

__global__ void __launch_bounds__(kNmsWG)
k_topk_nms(int top_k, double thresh, int4* __restrict__ rec, const uint32_t* __restrict__ ctr, TopkSel* __restrict__ sel,
           int32_t* __restrict__ nsel_status) {
    __shared__ int4 s_rec[kNmsLds];
    __shared__ Key128 s_keys[kNmsWG / 64];
    const int tid = threadIdx.x;
    if (ctr[1] == 0) {                                                             // a field does not fit the packed record
        if (tid == 0) { nsel_status[0] = 0; nsel_status[1] = 1; }
        return;
    }
    const int m = (int)(ctr[0] + 1u);
    if (m <= kNmsLds)
        for (int i = tid; i < m; i += kNmsWG) s_rec[i] = rec[i];
    __syncthreads();
    const int kept = m <= kNmsLds ? nms_rounds(s_rec, m, top_k, thresh, s_keys, sel) : nms_rounds(rec, m, top_k, thresh, s_keys, sel);
    if (tid == 0) { nsel_status[0] = kept; nsel_status[1] = 0; }
}

void launch_topk_nms(const Candidate* matches_dev, const unsigned long long* counters, uint32_t cap, const int32_t* work_pyramids,
                     const int32_t* work_cls, const int32_t* work_tid, const TemplEntry* entries, int levels,
                     const int32_t* class_base, const int32_t* view_wh, int num_views, int top_k,
                     double iou_thresh, void* scratch, TopkSel* sel, int32_t* nsel_status, hipStream_t s) {
    // scratch: cap int4 records, then the hash table (power of two >= 2 * cap 64-bit slots), then the two counters — one
    // memset presets table and counters to ~0
    size_t tsz = 1;
    while (tsz < 2 * (size_t)cap) tsz <<= 1;
    unsigned long long* table = reinterpret_cast<unsigned long long*>(reinterpret_cast<char*>(scratch) + (size_t)cap * 16);
    uint32_t* ctr = reinterpret_cast<uint32_t*>(table + tsz);
    (void)hipMemsetAsync(table, 0xFF, tsz * sizeof(unsigned long long) + 16, s);
    hipLaunchKernelGGL(k_nms_pack, dim3(64), dim3(256), 0, s, matches_dev, counters, cap, work_pyramids, work_cls, work_tid, entries, levels, class_base,
                       view_wh, num_views, reinterpret_cast<int4*>(scratch), table, (uint32_t)(tsz - 1), ctr);
    hipLaunchKernelGGL(k_topk_nms, dim3(1), dim3(kNmsWG), 0, s, top_k, iou_thresh, reinterpret_cast<int4*>(scratch), ctr, sel, nsel_status);
}

// Records identical in every field — several coarse candidates of one template refined to the same position — are
// adjacent in the canonical order and std::unique removes all but one (LL.cpp:1772-1774), so dropping them on the device
// changes nothing and shrinks what the host converts / sorts and what the multi-GPU all-gather carries (typically 5x).
// Persistent grid over the candidate slots; open-addressing hash on (x, y, work item); one atomic per wave appends the
// survivors to `distinct` (pinned host memory) and, as sort keys, to `distinct_keys` (HBM; exchange.hip).  counters[1] = distinct
// records, counters[2] = records alive before, counters[3] != 0: a record does not fit the key.
__global__ void __launch_bounds__(256)
k_dedupe(FrameBatch fb, uint32_t cap, uint32_t table_mask /*allocated slots - 1*/, const int32_t* __restrict__ work_cls,
         const int32_t* __restrict__ work_tid) {
    const FrameSlot& F = fb.f[blockIdx.y];                                           // grid (blocks, frames of the batch)
    const Candidate* __restrict__ matches = F.matches_dev;
    unsigned long long* __restrict__ counters = F.counters;
    unsigned long long* __restrict__ table = F.dedupe_table;
    Candidate* __restrict__ distinct = F.distinct;
    ulonglong2* __restrict__ distinct_keys = F.distinct_keys;
    unsigned long long* __restrict__ final_dev = F.final_dev;
    unsigned long long* __restrict__ final_host = F.final_host;
    __shared__ bool s_last;
    const unsigned long long nc = counters[0] & kCandMask;                           // low bits: candidates; above: tiles (k_coarse)
    const uint32_t n = (uint32_t)(nc < cap ? nc : cap);
    table_mask = dedupe_slots_for(n, table_mask + 1) - 1;                            // the slots k_local emptied for this frame
    const int lane = threadIdx.x & 63;
    for (uint32_t i0 = blockIdx.x * blockDim.x; i0 < n; i0 += gridDim.x * blockDim.x) {
        const uint32_t i = i0 + threadIdx.x;
        bool alive = false, keep = false;
        Candidate c{0, 0, 0.f, -1};
        if (i < n) {
            c = matches[i];
            alive = c.work >= 0;
            if (alive) {
                const unsigned long long key = ((unsigned long long)(uint32_t)c.work << 32) | ((uint32_t)(c.x & 0xFFFF) << 16) | (uint32_t)(c.y & 0xFFFF);
                const bool packable = c.x >= -32768 && c.x <= 32767 && c.y >= -32768 && c.y <= 32767;
                keep = !packable;                                                    // cannot be keyed: keep it (std::unique decides on the host)
                if (packable) {
                    unsigned long long hsh = key * 0x9E3779B97F4A7C15ull;
                    uint32_t slot = (uint32_t)(hsh >> 40) & table_mask;
                    for (;;) {                                                         // the table has >= 2 x cap slots: never full
                        const unsigned long long prev = atomicCAS(&table[slot], ~0ull, key);
                        if (prev == ~0ull) { keep = true; break; }
                        if (prev == key) break;
                        slot = (slot + 1) & table_mask;
                    }
                }
            }
        }
        const unsigned long long mk = __ballot(keep), ma = __ballot(alive);
        if (ma) {
            unsigned long long base = 0;
            if (lane == 0) {
                if (mk) base = atomicAdd(&counters[1], (unsigned long long)__popcll(mk));
                atomicAdd(&counters[2], (unsigned long long)__popcll(ma));
            }
            base = ((unsigned long long)(uint32_t)__shfl((int)(base >> 32), 0, 64) << 32) | (uint32_t)__shfl((int)(uint32_t)base, 0, 64);
            if (keep) {
                const unsigned long long at = base + __popcll(mk & ((1ull << lane) - 1ull));
                distinct[at] = c;
                if (distinct_keys) {                                                 // the record as a sort key for the multi-GPU exchange
                    const int cls = work_cls[c.work], tid = work_tid[c.work];
                    distinct_keys[at] = xchg_make_key(c.x, c.y, c.score, cls, tid);
                    if (!xchg_key_fits(c.x, c.y, cls, tid)) atomicOr(&counters[3], 1ull);
                }
            }
        }
    }
    // Publish + reset by the last block to finish (every block has read counters[0] by then).  The counter atomics are
    // performed at the device's coherence point, so all the ticket needs is that this block's atomics have completed — a wait
    // on the memory counter.  (__threadfence() here is an agent-scope release = a write-back of the XCD's L2: 9 -> 37 us.)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) s_last = atomicAdd(&counters[7], 1ull) == (unsigned long long)gridDim.x - 1ull;
    __syncthreads();
    if (s_last && threadIdx.x == 0) {
        const unsigned long long nd = atomicAdd(&counters[1], 0ull), na = atomicAdd(&counters[2], 0ull), bad = atomicAdd(&counters[3], 0ull);
        unsigned long long evals = 0, lbytes = 0;                                                        // the refinement's sharded statistics
        for (int q = 0; q < kStatShards; ++q) { evals += atomicAdd(&counters[8 + 2 * q], 0ull); lbytes += atomicAdd(&counters[9 + 2 * q], 0ull); }
        final_dev[0] = nc; final_dev[1] = nd; final_dev[2] = na; final_dev[3] = bad;
        final_host[0] = nc; final_host[1] = nd; final_host[2] = na; final_host[3] = bad;
        final_host[4] = atomicAdd(&counters[0], 0ull) >> kCandBits;                                       // tiles planned by k_coarse
        final_host[5] = evals; final_host[6] = lbytes;
        for (int q = 0; q < kCounterWords; ++q) counters[q] = 0;
    }
}

void launch_dedupe(const FrameBatch& fb, uint32_t cap, size_t table_slots, const int32_t* work_cls, const int32_t* work_tid, int blocks,
                   hipStream_t s) {
    if (fb.nb <= 0 || blocks <= 0) return;
    hipLaunchKernelGGL(k_dedupe, dim3(blocks, fb.nb), dim3(256), 0, s, fb, cap, (uint32_t)(table_slots - 1), work_cls, work_tid);
}

size_t dedupe_table_slots(uint32_t cap) {
    size_t t = 1;
    while (t < 2 * (size_t)cap) t <<= 1;
    return t;
}

size_t topk_nms_scratch_bytes(uint32_t cap) {
    size_t tsz = 1;
    while (tsz < 2 * (size_t)cap) tsz <<= 1;
    return (size_t)cap * 16 + tsz * sizeof(unsigned long long) + 16;
}

}  // namespace lm
