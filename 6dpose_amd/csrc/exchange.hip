// Multi-GPU exchange of match records on the device.  gfx950 only.
//
// Every rank refines its shard of the template bank; Detector::match then sorts the union of all records and removes
// adjacent duplicates (LL.cpp:1771-1776).  Doing that on the host costs every rank a sort of EVERY rank's records per
// frame — more than the matching kernels take at 8 GPUs — so it is split the way a distributed merge sort is.  The
// records travel as 128-bit keys (xchg_make_key: every field of a match, ascending key order = canonical order of SURVEY
// A12), written by k_dedupe next to the distinct records.
//   k_exchange_sort256   chunks of 256 keys, one workgroup each: bitonic network in LDS -> sorted runs of 256
//   k_exchange_merge256  the rank's runs merged by RANKING into one sorted run inside its block [header | keys]
//   (RCCL all-gather of the W blocks, enqueued by the caller on the same stream)
//   k_exchange_merge     every rank merges the W runs by ranking and writes the records in canonical order, marking the
//                        ones std::unique drops
// Ranking: the final position of a key is its index in its own run plus, for every other run, the number of keys below it
// (binary search in an LDS copy of the part of that run that overlaps the workgroup's 256 keys; keys never compare equal:
// k_dedupe removed exact duplicates within a rank and template ids are partitioned over the ranks).  A record is dropped by
// std::unique exactly when its predecessor in the merged order agrees on (x, y, similarity, class) (Match::operator==,
// LL.h:243-246); the predecessor is the largest of the per-run predecessors, which the same searches deliver, so no pass
// over the merged list is needed.  The host only skips the marked records while copying out.
#include "lm_kernels.h"

namespace lm {

namespace {

constexpr int kWG = 256;             // threads per workgroup = keys ranked per workgroup = length of the first-level runs
constexpr int kTile = 2048;          // keys of another run staged in LDS at a time (32 KB)

__device__ __forceinline__ bool key_less(const ulonglong2& a, const ulonglong2& b) { return a.x < b.x || (a.x == b.x && a.y < b.y); }

__device__ __forceinline__ float sim_from_key(unsigned long long hi) {
    uint32_t u = ~(uint32_t)(hi >> 32);
    u ^= (u >> 31) ? 0x80000000u : 0xFFFFFFFFu;
    return __uint_as_float(u);
}

struct RankLds {
    ulonglong2 tile[kTile];
    ulonglong2 first[kWG], last[kWG];     // first / last key of every (run, tile), kWG entries at a time
    ulonglong2 chunk[2];                  // first / last key of this workgroup's chunk
};

// Position of `key` (element idx of run `self`, valid lanes only) in the merge of R sorted runs, and its predecessor there.
// run(j) -> pointer to run j's keys, count(j) -> its length (<= run_cap).  All lanes of the workgroup call this.
template <class RunFn, class CountFn>
__device__ __forceinline__ void rank_in_runs(RankLds& L, const ulonglong2& key, bool valid, uint32_t idx, int self, int R, uint32_t run_cap,
                                             RunFn run, CountFn count, uint32_t& pos, ulonglong2& pred, bool& has_pred) {
    const int tile_len = (int)min(run_cap, (uint32_t)kTile);
    const int tiles = (int)((run_cap + tile_len - 1) / tile_len);
    const uint32_t n_self = count(self);
    if (threadIdx.x == 0) L.chunk[0] = key;
    if (valid && (idx + 1 == n_self || threadIdx.x == kWG - 1)) L.chunk[1] = key;
    pos = idx;
    has_pred = false;
    pred = make_ulonglong2(0, 0);
    if (valid && idx > 0) { pred = run(self)[idx - 1]; has_pred = true; }
    auto take_pred = [&](const ulonglong2& c) { if (!has_pred || key_less(pred, c)) { pred = c; has_pred = true; } };
    const int group = kWG / tiles;                                       // runs whose tile boundaries fit first[] / last[]
    for (int j0 = 0; j0 < R; j0 += group) {
        const int jn = min(R - j0, group);
        __syncthreads();
        for (int q = threadIdx.x; q < jn * tiles; q += kWG) {
            const int j = j0 + q / tiles, t = q % tiles;
            const uint32_t n_j = count(j), t0 = (uint32_t)t * tile_len;
            if (t0 < n_j) { L.first[q] = run(j)[t0]; L.last[q] = run(j)[min(n_j, t0 + tile_len) - 1]; }
        }
        __syncthreads();
        const ulonglong2 cfirst = L.chunk[0], clast = L.chunk[1];
        for (int jj = 0; jj < jn; ++jj) {
            const int j = j0 + jj;
            if (j == self) continue;
            const uint32_t n_j = count(j);
            const ulonglong2* rj = run(j);
            for (int t = 0; t < tiles; ++t) {
                const uint32_t t0 = (uint32_t)t * tile_len;
                if (t0 >= n_j) break;
                const uint32_t len = min(n_j - t0, (uint32_t)tile_len);
                const ulonglong2 tl = L.last[jj * tiles + t];
                if (key_less(tl, cfirst)) {                               // the whole tile is below every key of this chunk
                    if (valid) { pos += len; take_pred(tl); }
                    continue;
                }
                if (key_less(clast, L.first[jj * tiles + t])) break;      // this tile and the rest of the run are above the chunk
                __syncthreads();
                for (uint32_t q = threadIdx.x; q < len; q += kWG) L.tile[q] = rj[t0 + q];
                __syncthreads();
                if (valid) {
                    uint32_t lo = 0, hi = len;                            // number of tile keys below `key`
                    while (lo < hi) {
                        const uint32_t mid = (lo + hi) >> 1;
                        if (key_less(L.tile[mid], key)) lo = mid + 1; else hi = mid;
                    }
                    pos += lo;
                    if (lo > 0) take_pred(L.tile[lo - 1]);
                }
            }
        }
    }
}

__device__ __forceinline__ uint32_t pack_flags(const unsigned long long* counters, uint32_t cand_cap, uint32_t cap) {
    return (counters[0] > cand_cap ? kXchgCandOverflow : 0u) | (counters[1] > cap ? kXchgRunOverflow : 0u) | (counters[3] ? kXchgFieldOverflow : 0u);
}

}  // namespace

// grid cap / 256.  runs[b * 256 ..] = sorted chunk b of the rank's keys, padded with ~0 keys (which sort last).
// Workgroup 0 writes the block header {count, flags, capacity, 0}.
__global__ void __launch_bounds__(kWG)
k_exchange_sort256(XchgGroup G, uint32_t cand_cap, uint32_t cap) {
    __shared__ ulonglong2 s[kWG];
    const XchgFrame& F = G.f[blockIdx.y];                                 // grid (chunks, frames of the group)
    const ulonglong2* __restrict__ keys = F.keys;
    const unsigned long long* __restrict__ counters = F.counters;
    ulonglong2* __restrict__ runs = F.runs;
    uint32_t* __restrict__ block = F.block;
    const uint32_t flags = pack_flags(counters, cand_cap, cap);
    const unsigned long long nd = counters[1];
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        block[0] = (uint32_t)(nd > 0xFFFFFFFFull ? 0xFFFFFFFFull : nd); block[1] = flags; block[2] = cap; block[3] = 0;
    }
    if (flags) return;                                                    // uniform: nothing to send, every rank learns why
    const uint32_t n = (uint32_t)nd;
    if (blockIdx.x * kWG >= n) return;
    const uint32_t i = blockIdx.x * kWG + threadIdx.x;
    s[threadIdx.x] = i < n ? keys[i] : make_ulonglong2(~0ull, ~0ull);
    __syncthreads();
    for (uint32_t k = 2; k <= kWG; k <<= 1)
        for (uint32_t j = k >> 1; j > 0; j >>= 1) {
            const uint32_t t = threadIdx.x, l = t ^ j;
            if (l > t) {
                const ulonglong2 a = s[t], b = s[l];
                if (key_less(b, a) == ((t & k) == 0)) { s[t] = b; s[l] = a; }
            }
            __syncthreads();
        }
    runs[i] = s[threadIdx.x];
}

// grid cap / 256.  The rank's sorted chunks -> one sorted run in block + 4 words.
__global__ void __launch_bounds__(kWG)
k_exchange_merge256(XchgGroup G, uint32_t cand_cap, uint32_t cap) {
    __shared__ RankLds L;
    const XchgFrame& F = G.f[blockIdx.y];
    const ulonglong2* __restrict__ runs = F.runs;
    const unsigned long long* __restrict__ counters = F.counters;
    uint32_t* __restrict__ block = F.block;
    if (pack_flags(counters, cand_cap, cap)) return;
    const uint32_t n = (uint32_t)counters[1];
    const int self = blockIdx.x;
    if ((uint32_t)self * kWG >= n) return;
    const int R = (int)((n + kWG - 1) / kWG);
    auto run = [&](int j) { return runs + (size_t)j * kWG; };
    auto count = [&](int j) { return min(n - (uint32_t)j * kWG, (uint32_t)kWG); };
    const uint32_t idx = threadIdx.x;
    const bool valid = idx < count(self);
    const ulonglong2 key = run(self)[idx];                                 // padding is ~0: harmless for the chunk bounds
    uint32_t pos; ulonglong2 pred; bool has_pred;
    rank_in_runs(L, key, valid, idx, self, R, kWG, run, count, pos, pred, has_pred);
    if (valid) reinterpret_cast<ulonglong2*>(block + 4)[pos] = key;
}

// grid (cap / 256, W).  blocks: W blocks of block_words uint32 each.  merged: kXchgHeaderWords words, then 5-word records
// (x, y, similarity, class position — or -1 - class position when std::unique drops the record —, template id).
__global__ void __launch_bounds__(kWG)
k_exchange_merge(XchgGroup G, int W, uint32_t cap, uint32_t block_words) {
    __shared__ RankLds L;
    __shared__ uint32_t s_or, s_total;
    const uint32_t* __restrict__ blocks = G.f[blockIdx.z].recv;           // grid (chunks, ranks, frames of the group)
    int32_t* __restrict__ merged = G.f[blockIdx.z].merged;
    const int self = blockIdx.y;
    const uint32_t b = blockIdx.x;
    if (threadIdx.x == 0) { s_or = 0; s_total = 0; }
    __syncthreads();
    {
        uint32_t f = 0, cnt = 0;
        for (int j = threadIdx.x; j < W; j += kWG) {
            const uint32_t* h = blocks + (size_t)j * block_words;
            f |= h[1] | (h[2] != cap ? kXchgRunOverflow : 0u);
            cnt += h[0];
        }
        if (f) atomicOr(&s_or, f);
        if (cnt) atomicAdd(&s_total, cnt);
    }
    __syncthreads();
    const uint32_t flags = s_or;
    if (b == 0 && self == 0) {
        if (threadIdx.x == 0) { merged[0] = flags ? 0 : (int32_t)s_total; merged[1] = (int32_t)flags; merged[2] = W; merged[3] = (int32_t)cap; }
        for (int j = threadIdx.x; j < W && j < kXchgHeaderWords - 8; j += kWG) merged[8 + j] = (int32_t)blocks[(size_t)j * block_words];
    }
    if (flags) return;
    auto run = [&](int j) { return reinterpret_cast<const ulonglong2*>(blocks + (size_t)j * block_words + 4); };
    auto count = [&](int j) { return blocks[(size_t)j * block_words]; };
    const uint32_t n_self = count(self);
    if (b * kWG >= n_self) return;
    const uint32_t idx = b * kWG + threadIdx.x;
    const bool valid = idx < n_self;
    const ulonglong2 key = valid ? run(self)[idx] : make_ulonglong2(~0ull, ~0ull);
    uint32_t pos; ulonglong2 pred; bool has_pred;
    rank_in_runs(L, key, valid, idx, self, W, cap, run, count, pos, pred, has_pred);
    if (!valid) return;
    const bool dup = has_pred && (pred.x >> 32) == (key.x >> 32) && pred.y == key.y;   // same x, y, similarity, class
    const int32_t cls = (int32_t)(uint32_t)(key.y >> 32);
    int32_t* o = merged + kXchgHeaderWords + (size_t)pos * 5;
    o[0] = (int32_t)(uint32_t)(key.y & 0xFFFFu) - 32768;
    o[1] = (int32_t)(uint32_t)((key.y >> 16) & 0xFFFFu) - 32768;
    o[2] = (int32_t)__float_as_uint(sim_from_key(key.x));
    o[3] = dup ? -1 - cls : cls;
    o[4] = (int32_t)(uint32_t)(key.x & 0xFFFFFFFFu);
}

void launch_exchange_pack_group(const XchgGroup& G, uint32_t cand_cap, uint32_t cap, hipStream_t s) {
    if (G.n <= 0) return;
    const dim3 grid((cap + kWG - 1) / kWG, G.n);
    hipLaunchKernelGGL(k_exchange_sort256, grid, dim3(kWG), 0, s, G, cand_cap, cap);
    hipLaunchKernelGGL(k_exchange_merge256, grid, dim3(kWG), 0, s, G, cand_cap, cap);
}

// The merged list of every frame of the group into its pinned host buffer: header + the records actually there (16-byte stores).  A
// kernel instead of one hipMemcpyAsync per frame: the copies were the only D2H transfers of the stream path, they sat on the calling
// thread (a call could take 0.3 ms) and moved the whole capacity (82 KB per frame at 4096 records) instead of what the frame produced.
__global__ void __launch_bounds__(kWG)
k_exchange_copyout(XchgGroup G) {
    const int32_t* __restrict__ src = G.f[blockIdx.y].merged;
    int32_t* __restrict__ dst = G.f[blockIdx.y].host;
    const uint32_t records = src[1] ? 0u : (uint32_t)src[0];              // flags set: header only
    const uint32_t words = kXchgHeaderWords + records * 5u;
    const uint32_t quads = (words + 3u) >> 2;                              // both buffers hold whole 16-byte units (allocated for the capacity)
    const uint4* s4 = reinterpret_cast<const uint4*>(src);
    uint4* d4 = reinterpret_cast<uint4*>(dst);
    for (uint32_t i = blockIdx.x * kWG + threadIdx.x; i < quads; i += gridDim.x * kWG) d4[i] = s4[i];
}

void launch_exchange_merge_group(const XchgGroup& G, int world, uint32_t cap, hipStream_t s, uint32_t rank_stride_words) {
    if (G.n <= 0) return;
    const uint32_t block_words = rank_stride_words ? rank_stride_words : 4 + cap * 4;   // distance between the blocks of consecutive ranks
    hipLaunchKernelGGL(k_exchange_merge, dim3((cap + kWG - 1) / kWG, world, G.n), dim3(kWG), 0, s, G, world, cap, block_words);
    if (G.f[0].host) hipLaunchKernelGGL(k_exchange_copyout, dim3(16, G.n), dim3(kWG), 0, s, G);
}

}  // namespace lm
