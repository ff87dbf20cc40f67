// Multi-GPU exchange of match records on the device.  gfx950 only.
//
// Every rank refines its shard of the template bank; Detector::match then sorts the union of all records and removes
// adjacent duplicates (LL.cpp:1771-1776).  Doing that on the host costs every rank a sort of EVERY rank's records per
// frame — more than the matching kernels take at 8 GPUs — so it is split the way a distributed merge sort is:
//   k_exchange_pack   each rank turns its distinct records (k_dedupe) into 128-bit sort keys that encode every field of a
//                     match in canonical order (similarity desc, template id, class position, y, x — SURVEY A12) and sorts
//                     them in LDS (bitonic, one workgroup): a sorted RUN in a fixed-capacity block [header | keys]
//   (RCCL all-gather of the blocks, enqueued by the caller on the same stream)
//   k_exchange_merge  every rank merges the W runs by RANKING: the final position of a key is its index in its own run
//                     plus, for every other run, the number of keys below it (binary search in LDS tiles; keys of
//                     different runs never compare equal because template ids are partitioned over the ranks), and it
//                     is dropped by std::unique exactly when its predecessor in the merged order agrees on
//                     (x, y, similarity, class) (Match::operator==, LL.h:243-246) — the predecessor is the largest of the
//                     W per-run predecessors, so no second pass over the merged list is needed.
// The records land in canonical order with a drop mark; the host only skips the marked ones while copying out.
#include "lm_kernels.h"

namespace lm {

namespace {

struct Key128 { unsigned long long hi, lo; };

__device__ __forceinline__ bool key_less(const Key128& a, const Key128& b) { return a.hi < b.hi || (a.hi == b.hi && a.lo < b.lo); }

// float -> unsigned that orders like the float, inverted so that larger similarities come first
__device__ __forceinline__ uint32_t sim_desc_bits(float f) {
    uint32_t u = __float_as_uint(f);
    u ^= (u >> 31) ? 0xFFFFFFFFu : 0x80000000u;
    return ~u;
}
__device__ __forceinline__ float sim_from_desc_bits(uint32_t v) {
    uint32_t u = ~v;
    u ^= (u >> 31) ? 0x80000000u : 0xFFFFFFFFu;
    return __uint_as_float(u);
}

constexpr int kPackWG = 1024;
constexpr int kMergeWG = 256;
constexpr int kTile = 2048;          // keys of another run staged in LDS at a time (32 KB)

}  // namespace

// One workgroup.  counters: [0] coarse candidates, [1] distinct records of this frame (k_dedupe).  block: XHeader + cap keys.
__global__ void __launch_bounds__(kPackWG)
k_exchange_pack(const Candidate* __restrict__ distinct, const unsigned long long* __restrict__ counters, uint32_t cand_cap,
                const int32_t* __restrict__ work_cls, const int32_t* __restrict__ work_tid, uint32_t cap, uint32_t* __restrict__ block) {
    extern __shared__ unsigned long long s_keys[];        // [P2] hi then [P2] lo
    __shared__ uint32_t s_flags;
    const unsigned long long ncand = counters[0];
    const unsigned long long nd = counters[1];
    if (threadIdx.x == 0) s_flags = (ncand > cand_cap ? kXchgCandOverflow : 0u) | (nd > cap ? kXchgRunOverflow : 0u);
    __syncthreads();
    if (s_flags) {                                         // uniform: nothing to send, every rank learns why
        if (threadIdx.x == 0) { block[0] = (uint32_t)(nd > 0xFFFFFFFFull ? 0xFFFFFFFFu : nd); block[1] = s_flags; block[2] = cap; block[3] = 0; }
        return;
    }
    const uint32_t n = (uint32_t)nd;
    uint32_t P2 = 1;
    while (P2 < n) P2 <<= 1;
    unsigned long long* s_hi = s_keys;
    unsigned long long* s_lo = s_keys + P2;
    bool bad = false;
    for (uint32_t i = threadIdx.x; i < P2; i += kPackWG) {
        unsigned long long hi = ~0ull, lo = ~0ull;         // padding sorts last
        if (i < n) {
            const Candidate c = distinct[i];
            const int32_t cls = work_cls[c.work], tid = work_tid[c.work];
            bad |= c.x < -32768 || c.x > 32767 || c.y < -32768 || c.y > 32767 || cls < 0 || tid < 0;
            hi = ((unsigned long long)sim_desc_bits(c.score) << 32) | (uint32_t)tid;
            lo = ((unsigned long long)(uint32_t)cls << 32) | ((unsigned long long)(uint16_t)(c.y + 32768) << 16) | (uint16_t)(c.x + 32768);
        }
        s_hi[i] = hi; s_lo[i] = lo;
    }
    if (bad) atomicOr(&s_flags, kXchgFieldOverflow);
    __syncthreads();
    for (uint32_t k = 2; k <= P2; k <<= 1)
        for (uint32_t j = k >> 1; j > 0; j >>= 1) {
            for (uint32_t t = threadIdx.x; t < P2; t += kPackWG) {
                const uint32_t l = t ^ j;
                if (l > t) {
                    const Key128 a{s_hi[t], s_lo[t]}, b{s_hi[l], s_lo[l]};
                    const bool up = (t & k) == 0;
                    if (key_less(b, a) == up) { s_hi[t] = b.hi; s_lo[t] = b.lo; s_hi[l] = a.hi; s_lo[l] = a.lo; }
                }
            }
            __syncthreads();
        }
    ulonglong2* out = reinterpret_cast<ulonglong2*>(block + 4);
    for (uint32_t i = threadIdx.x; i < n; i += kPackWG) out[i] = make_ulonglong2(s_hi[i], s_lo[i]);
    if (threadIdx.x == 0) { block[0] = n; block[1] = s_flags; block[2] = cap; block[3] = 0; }
}

// grid (cap / kMergeWG, W).  blocks: W blocks of block_words uint32 each.  merged: XchgResultHeader words, then 5-word records
// (x, y, similarity, class position — or -1 - class position when std::unique drops the record —, template id).
__global__ void __launch_bounds__(kMergeWG)
k_exchange_merge(const uint32_t* __restrict__ blocks, int W, uint32_t cap, uint32_t block_words, int32_t* __restrict__ merged) {
    __shared__ ulonglong2 s_tile[kTile];
    __shared__ ulonglong2 s_first[kMergeWG], s_last[kMergeWG];   // first / last key of every (run, tile), kMergeWG entries at a time
    __shared__ ulonglong2 s_chunk[2];
    __shared__ uint32_t s_or, s_total;
    const int i = blockIdx.y;
    const uint32_t b = blockIdx.x;
    const int tiles = (int)((cap + kTile - 1) / kTile);
    if (threadIdx.x == 0) { s_or = 0; s_total = 0; }
    __syncthreads();
    {
        uint32_t f = 0, cnt = 0;
        for (int j = threadIdx.x; j < W; j += kMergeWG) {
            const uint32_t* h = blocks + (size_t)j * block_words;
            f |= h[1] | (h[2] != cap ? kXchgRunOverflow : 0u);
            cnt += h[0];
        }
        if (f) atomicOr(&s_or, f);
        if (cnt) atomicAdd(&s_total, cnt);
    }
    __syncthreads();
    const uint32_t flags = s_or;
    if (b == 0 && i == 0) {
        if (threadIdx.x == 0) { merged[0] = flags ? 0 : (int32_t)s_total; merged[1] = (int32_t)flags; merged[2] = W; merged[3] = (int32_t)cap; }
        for (int j = threadIdx.x; j < W && j < kXchgHeaderWords - 8; j += kMergeWG) merged[8 + j] = (int32_t)blocks[(size_t)j * block_words];
    }
    if (flags) return;
    const uint32_t* mine = blocks + (size_t)i * block_words;
    const uint32_t n_i = mine[0];
    if (b * kMergeWG >= n_i) return;
    const ulonglong2* run_i = reinterpret_cast<const ulonglong2*>(mine + 4);
    const uint32_t idx = b * kMergeWG + threadIdx.x;
    const bool valid = idx < n_i;
    const ulonglong2 kv = valid ? run_i[idx] : make_ulonglong2(~0ull, ~0ull);
    const Key128 key{kv.x, kv.y};
    if (threadIdx.x == 0) s_chunk[0] = kv;
    if (idx == min(n_i, (b + 1) * kMergeWG) - 1) s_chunk[1] = kv;
    uint32_t pos = idx;
    bool has_pred = false;
    Key128 pred{0, 0};
    if (valid && idx > 0) { const ulonglong2 p = run_i[idx - 1]; pred = Key128{p.x, p.y}; has_pred = true; }
    __syncthreads();
    const Key128 cfirst{s_chunk[0].x, s_chunk[0].y}, clast{s_chunk[1].x, s_chunk[1].y};
    auto take_pred = [&](const Key128& c) { if (!has_pred || key_less(pred, c)) { pred = c; has_pred = true; } };
    for (int j0 = 0; j0 < W; j0 += kMergeWG / tiles) {            // groups of runs whose tile boundaries fit the two LDS arrays
        const int jn = min(W - j0, kMergeWG / tiles);
        __syncthreads();
        for (int q = threadIdx.x; q < jn * tiles; q += kMergeWG) {
            const int j = j0 + q / tiles, t = q % tiles;
            const uint32_t* h = blocks + (size_t)j * block_words;
            const uint32_t n_j = h[0];
            const uint32_t t0 = (uint32_t)t * kTile;
            if (t0 < n_j) {
                const ulonglong2* run_j = reinterpret_cast<const ulonglong2*>(h + 4);
                s_first[q] = run_j[t0];
                s_last[q] = run_j[min(n_j, t0 + kTile) - 1];
            }
        }
        __syncthreads();
        for (int jj = 0; jj < jn; ++jj) {
            const int j = j0 + jj;
            if (j == i) continue;
            const uint32_t* h = blocks + (size_t)j * block_words;
            const uint32_t n_j = h[0];
            const ulonglong2* run_j = reinterpret_cast<const ulonglong2*>(h + 4);
            for (int t = 0; t < tiles; ++t) {
                const uint32_t t0 = (uint32_t)t * kTile;
                if (t0 >= n_j) break;
                const uint32_t len = min(n_j - t0, (uint32_t)kTile);
                const Key128 tl{s_last[jj * tiles + t].x, s_last[jj * tiles + t].y};
                if (key_less(tl, cfirst)) {                       // the whole tile is below every key of this chunk
                    if (valid) { pos += len; take_pred(tl); }
                    continue;
                }
                const Key128 tf{s_first[jj * tiles + t].x, s_first[jj * tiles + t].y};
                if (key_less(clast, tf)) break;                   // this tile and the rest of the run are above the chunk
                __syncthreads();
                for (uint32_t q = threadIdx.x; q < len; q += kMergeWG) s_tile[q] = run_j[t0 + q];
                __syncthreads();
                if (valid) {
                    uint32_t lo = 0, hi = len;                    // number of tile keys below `key`
                    while (lo < hi) {
                        const uint32_t mid = (lo + hi) >> 1;
                        const ulonglong2 m = s_tile[mid];
                        if (key_less(Key128{m.x, m.y}, key)) lo = mid + 1; else hi = mid;
                    }
                    pos += lo;
                    if (lo > 0) { const ulonglong2 m = s_tile[lo - 1]; take_pred(Key128{m.x, m.y}); }
                }
            }
        }
    }
    if (!valid) return;
    const bool dup = has_pred && (pred.hi >> 32) == (key.hi >> 32) && pred.lo == key.lo;   // same x, y, similarity, class
    const int32_t cls = (int32_t)(uint32_t)(key.lo >> 32);
    int32_t* o = merged + kXchgHeaderWords + (size_t)pos * 5;
    o[0] = (int32_t)(uint32_t)(key.lo & 0xFFFFu) - 32768;
    o[1] = (int32_t)(uint32_t)((key.lo >> 16) & 0xFFFFu) - 32768;
    o[2] = (int32_t)__float_as_uint(sim_from_desc_bits((uint32_t)(key.hi >> 32)));
    o[3] = dup ? -1 - cls : cls;
    o[4] = (int32_t)(uint32_t)(key.hi & 0xFFFFFFFFu);
}

size_t exchange_pack_lds_bytes(uint32_t cap) { return (size_t)cap * 16; }

int launch_exchange_pack(const Candidate* distinct, const unsigned long long* counters, uint32_t cand_cap, const int32_t* work_cls,
                         const int32_t* work_tid, uint32_t cap, uint32_t* block, hipStream_t s) {
    static size_t configured = 0;
    const size_t lds = exchange_pack_lds_bytes(cap);
    if (lds > configured) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(k_exchange_pack), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return -1;
        configured = lds;
    }
    hipLaunchKernelGGL(k_exchange_pack, dim3(1), dim3(kPackWG), lds, s, distinct, counters, cand_cap, work_cls, work_tid, cap, block);
    return 0;
}

void launch_exchange_merge(const uint32_t* blocks, int world, uint32_t cap, int32_t* merged, hipStream_t s) {
    const uint32_t block_words = 4 + cap * 4;
    hipLaunchKernelGGL(k_exchange_merge, dim3((cap + kMergeWG - 1) / kMergeWG, world), dim3(kMergeWG), 0, s, blocks, world, cap, block_words, merged);
}

}  // namespace lm
