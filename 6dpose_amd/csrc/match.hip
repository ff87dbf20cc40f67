// Similarity kernels of Detector::matchClass on gfx950 (reference LL.cpp:1284-1428 similarity /
// similarityLocal, LL.cpp:1788-1941 matchClass).  Pure integer byte work: every response is a
// u8 in {0,1,4}, summed per template position.  Nothing here is a dense contraction, so no MFMA;
// the linear memories of one frame (1.2 MB coarse + 4.9 MB fine at VGA) live in L2 / Infinity
// Cache; the kernels are bound by the vector-L1 access rate (see "Gather discipline" below), not by
// HBM and not by arithmetic.
//
// Data layout (built by frontend.hip / detector.cpp, offsets in FrameGeom):
//   LM arena: per level, per modality: u8 [8 labels][T*T phases][(W/T)*(H/T)] + zero tail.
//   Bank:     TemplEntry per (pyramid, level).  Its features (both modalities) are sorted by alignment
//             class and padded to a multiple of kFeatBatch with entries that read zeros, so the inner
//             loops are branch-free.  feat_off[] = byte offset of the feature's linear-memory run
//             from the arena start (accessLinearMemory, LL.cpp:1248-1271, resolved on the host for
//             the current frame geometry with floor division, so adding a multiple-of-T window
//             offset stays exact).  At the top level, features outside the image (LL.cpp:1330) are
//             already redirected to the zero tail.  feat_xy[] = int16 x | int16 y << 16 is only
//             read on the slow path of the refinement (bounds test of LL.cpp:1394).
#include "lm_kernels.h"

namespace lm {

static __device__ __forceinline__ uint32_t ld_u32(const uint8_t* p) {
    uint32_t v;
    __builtin_memcpy(&v, p, 4);   // unaligned global_load_dword (gfx950 unaligned access mode)
    return v;
}

// score = (raw * 100.f) / (4 * num_features)  — LL.cpp:1842, 1918; IEEE single, no contraction
static __device__ __forceinline__ float score_of(int raw, int nfeat) {
    return __fdiv_rn(__fmul_rn((float)raw, 100.f), (float)(4 * nfeat));
}

// ---------------------------------------------------------------------------------------------
// Gather discipline (measured, profiles/r01_pmc_*.txt): the vector L1 (TCP) services one access per
// cycle and an access moves at most 64 aligned bytes.  Unaligned or narrow per-lane gathers explode
// into 3-4 accesses per quad of lanes and the kernels become TCP-access bound, so both kernels only
// issue ALIGNED 16-byte loads (1 KB per wave instruction in 16 accesses).  The byte misalignment of
// a feature's run is handled without per-feature shifting: the host sorts the features of every
// template entry by their alignment class (run offset mod 16), the kernel adds the aligned chunks of
// one class into packed-u8 accumulators (<= 63 features x 4 < 256, the reference's own no-overflow
// argument, README.md:67-69) and realigns ONCE per class run (v_alignbyte + one neighbour exchange)
// into the u16 position accumulators.
// ---------------------------------------------------------------------------------------------
static __device__ __forceinline__ uint4 ld_aligned16(const uint8_t* p) { return *reinterpret_cast<const uint4*>(p); }

// widen 4 packed bytes of `v` into two packed u16x2 accumulators: e gets bytes 0,2 ; o gets bytes 1,3
static __device__ __forceinline__ void add_bytes(uint32_t v, uint32_t& e, uint32_t& o) {
    e += v & 0x00FF00FFu;
    o += (v >> 8) & 0x00FF00FFu;
}

// ---------------------------------------------------------------------------------------------
// Coarse pass (LL.cpp:1284-1354 similarity + :1835-1852 scan): one workgroup per template pyramid.
// Lane i of wave w owns the 16 positions [16*(63w+i), +16) of the decimated top-level grid and loads
// the aligned 16-byte chunk that starts at or before them; the tail of its positions lives in lane
// i+1's chunk, hence 63 producing lanes per wave plus one feeder lane.
// ---------------------------------------------------------------------------------------------
constexpr int kChunksPerWave = 63;

__global__ void __launch_bounds__(1024)
k_coarse(const uint8_t* __restrict__ lm_arena, LevelGeom lv, int level, int levels,
         const TemplEntry* __restrict__ entries, const int32_t* __restrict__ feat_off,
         const int32_t* __restrict__ work_pyramids, float threshold, Candidate* __restrict__ cands, uint32_t cap,
         unsigned long long* __restrict__ counters, uint32_t* __restrict__ tcount, uint32_t* __restrict__ tlist, uint8_t* __restrict__ todo) {
    const int work = blockIdx.x;
    const int pyr = work_pyramids[work];
    const TemplEntry e = entries[(size_t)pyr * levels + level];
    const int nf = e.nf, nfp = e.nf_padded;
    const int32_t* fo = feat_off + e.feat_start;
    const int Wd = lv.Wd, Hd = lv.Hd, T = lv.T;
    const int npos = Wd * Hd;
    // LL.cpp:1299-1309
    const int wf = (e.width - 1) / T + 1, hf = (e.height - 1) / T + 1;
    const int tp = (Hd - hf) * Wd + (Wd - wf) + 1;
    const int offset = T / 2 + (T % 2 - 1);   // LL.cpp:1846
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwaves = blockDim.x >> 6;

    for (int chunk0 = wave * kChunksPerWave; chunk0 * 16 < npos; chunk0 += nwaves * kChunksPerWave) {
        const int j0 = (chunk0 + lane) * 16;                   // first position owned by this lane
        uint32_t even[4] = {0, 0, 0, 0}, odd[4] = {0, 0, 0, 0};
        if (chunk0 * 16 < tp && nfp > 0) {                     // wave-uniform
            const uint8_t* base = lm_arena + j0;
            uint32_t r8[4] = {0, 0, 0, 0};                      // packed-u8 sums of the current class run
            int cur = -1, cnt = 0;
            auto flush = [&](int cls) {                         // realign the run: bytes [cls, cls+16) of {own, next lane}
                uint32_t x[8];
#pragma unroll
                for (int k = 0; k < 4; ++k) { x[k] = r8[k]; x[4 + k] = (uint32_t)__shfl_down((int)r8[k], 1, 64); r8[k] = 0; }
                const int d = cls >> 2;
                const uint32_t sb = (uint32_t)(cls & 3);
                uint32_t o4[4];
                switch (d) {                                    // wave-uniform
                    case 0:
#pragma unroll
                        for (int k = 0; k < 4; ++k) o4[k] = __builtin_amdgcn_alignbyte(x[k + 1], x[k], sb);
                        break;
                    case 1:
#pragma unroll
                        for (int k = 0; k < 4; ++k) o4[k] = __builtin_amdgcn_alignbyte(x[k + 2], x[k + 1], sb);
                        break;
                    case 2:
#pragma unroll
                        for (int k = 0; k < 4; ++k) o4[k] = __builtin_amdgcn_alignbyte(x[k + 3], x[k + 2], sb);
                        break;
                    default:
#pragma unroll
                        for (int k = 0; k < 4; ++k) o4[k] = __builtin_amdgcn_alignbyte(x[k + 4], x[k + 3], sb);
                        break;
                }
#pragma unroll
                for (int k = 0; k < 4; ++k) add_bytes(o4[k], even[k], odd[k]);
            };
            int32_t o[kFeatBatch];
#pragma unroll
            for (int u = 0; u < kFeatBatch; ++u) o[u] = fo[u];                      // wave-uniform -> SMEM
            for (int f = 0; f < nfp; f += kFeatBatch) {
                const int fn = f + kFeatBatch < nfp ? f + kFeatBatch : f;             // prefetch next batch of offsets
                int32_t on[kFeatBatch];
#pragma unroll
                for (int u = 0; u < kFeatBatch; ++u) on[u] = fo[fn + u];
                uint4 v[kFeatBatch];
#pragma unroll
                for (int u = 0; u < kFeatBatch; ++u) v[u] = ld_aligned16(base + (o[u] & ~15));
#pragma unroll
                for (int u = 0; u < kFeatBatch; ++u) {
                    const int cls = o[u] & 15;
                    if (cls != cur || cnt == 63) {
                        if (cur >= 0) flush(cur);
                        cur = cls; cnt = 0;
                    }
                    r8[0] += v[u].x; r8[1] += v[u].y; r8[2] += v[u].z; r8[3] += v[u].w;   // bytes never carry (<= 252)
                    ++cnt;
                }
#pragma unroll
                for (int u = 0; u < kFeatBatch; ++u) o[u] = on[u];
            }
            if (cur >= 0) flush(cur);
        }
        // Threshold scan (LL.cpp:1835-1852).  One atomicAdd per WAVE reserves the slots of all its hits (a
        // single counter word saturates at ~90 atomics/us on this chip: per-candidate atomics made the pass
        // atomic-bound), then every lane writes its hits at the reserved base + its exclusive prefix.
        uint32_t hit_mask = 0;
        float sc[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const int j = j0 + k;
            const uint32_t pk = (k & 1) ? odd[k >> 2] : even[k >> 2];
            const int rawk = (int)((k & 2) ? (pk >> 16) : (pk & 0xFFFF));
            const int r = j < tp ? rawk : 0;                  // positions >= template_positions stay 0
            sc[k] = score_of(r, nf);
            if (lane < kChunksPerWave && j < npos && sc[k] > threshold) hit_mask |= 1u << k;   // LL.cpp:1844
        }
        int mine = __popc(hit_mask), incl = mine;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            int t = __shfl_up(incl, o, 64);
            if (lane >= o) incl += t;
        }
        const int total = __shfl(incl, 63, 64);
        if (total > 0) {                                       // wave-uniform
            unsigned long long wbase = 0;
            if (lane == 0) wbase = atomicAdd(&counters[0], (unsigned long long)total);
            wbase = ((unsigned long long)(uint32_t)__shfl((int)(wbase >> 32), 0, 64) << 32) | (uint32_t)__shfl((int)(uint32_t)wbase, 0, 64);
            unsigned long long slot = wbase + (unsigned long long)(incl - mine);
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                if (hit_mask & (1u << k)) {
                    if (slot < cap) {
                        const int j = j0 + k;
                        const int cy = j / Wd, cx = j - cy * Wd;
                        Candidate c;
                        c.x = cx * T + offset; c.y = cy * T + offset; c.score = sc[k]; c.work = work;
                        cands[slot] = c;
                        if (tcount) {                          // the template's own candidate list (k_local_region); todo = 1: not refined yet
                            todo[slot] = 1;
                            const uint32_t ti = atomicAdd(&tcount[work], 1u);
                            if (ti < (uint32_t)kRegionK) tlist[(size_t)work * kRegionK + ti] = (uint32_t)slot;
                        }
                    }
                    ++slot;
                }
            }
        }
    }
}

void launch_coarse(const uint8_t* lm_arena, const FrameGeom& g, const TemplEntry* entries, const int32_t* feat_off,
                   const int32_t* work_pyramids, int num_work, float threshold, Candidate* cands, uint32_t cap,
                   unsigned long long* counters, uint32_t* tcount, uint32_t* tlist, uint8_t* todo, hipStream_t s) {
    if (num_work <= 0) return;
    const int level = g.levels - 1;
    const LevelGeom lv = g.lv[level];
    int npos = lv.Wd * lv.Hd;
    int waves = (npos + kChunksPerWave * 16 - 1) / (kChunksPerWave * 16);
    if (waves > 16) waves = 16;
    if (waves < 1) waves = 1;
    hipLaunchKernelGGL(k_coarse, dim3(num_work), dim3(waves * 64), 0, s, lm_arena, lv, level, g.levels, entries, feat_off,
                       work_pyramids, threshold, cands, cap, counters, tcount, tlist, todo);
}

// ---------------------------------------------------------------------------------------------
// Local refinement (LL.cpp:1855-1938, similarityLocal :1366-1428): one wave64 per candidate, marching
// up the pyramid; the grid is persistent and strides over the candidate list whose length is read
// from device memory (no host round trip between the coarse and the local pass).
//
// Fast path (every feature's 16x16 window inside its plane — always, except for oversized templates):
// gathers from the strip-major copy of the linear memories ([strip of 16 columns][row][16 B]).  A
// window spans two strips, so 32 lanes x one aligned 16-byte load cover it: lane (row r, strip h).
// The two half-waves work on two features of the same alignment class at a time; class runs are
// accumulated in packed u8 and realigned once per run: half-waves combined (lane ^ 32), strip pairs
// exchanged (lane ^ 1), v_alignbyte by the run's byte phase, widened into the u16 window
// accumulators (lane (r,h) keeps window columns 8h..8h+7 of row r).
// Slow path: flat layout, per-feature bounds test, exactly the reference's reads (wrap-around included).
// First strict maximum (LL.cpp:1920) = wave max-reduction of the packed key (raw << 8 | 255 - index).
// ---------------------------------------------------------------------------------------------
static __device__ __forceinline__ uint32_t wave_max_u32(uint32_t v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        uint32_t t = (uint32_t)__shfl_xor((int)v, o, 64);
        v = t > v ? t : v;
    }
    return v;
}

__global__ void __launch_bounds__(256)
k_local(const uint8_t* __restrict__ lm_arena, const uint8_t* __restrict__ sm_arena, FrameGeom g,
        const TemplEntry* __restrict__ entries, const int32_t* __restrict__ feat_off,
        const FeatStrip* __restrict__ feat_strip, const uint32_t* __restrict__ feat_xy,
        const int32_t* __restrict__ work_pyramids, const Candidate* __restrict__ cands, uint32_t cand_cap,
        float threshold, Candidate* __restrict__ matches, Candidate* __restrict__ matches_dev, uint32_t cap,
        const unsigned long long* __restrict__ counters, unsigned long long* __restrict__ block_stats,
        unsigned long long* __restrict__ dedupe_table, uint32_t dedupe_cap_slots, const uint8_t* __restrict__ todo) {
    __shared__ unsigned long long s_stats[4][2];
    const int lane = threadIdx.x & 63;
    const uint32_t wave0 = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const uint32_t nwaves = gridDim.x * (blockDim.x >> 6);
    unsigned long long nc = counters[0];
    const uint32_t num_cands = nc < cand_cap ? (uint32_t)nc : cand_cap;
    unsigned long long evals = 0, bytes = 0;
    if (blockIdx.x == 0 && threadIdx.x == 0) block_stats[0] = nc;   // candidate count for the host (pinned memory)
    // the part of k_dedupe's hash table this frame will use, emptied here (k_dedupe runs next on the stream): no memset node
    if (dedupe_table) {
        const uint32_t tsize = dedupe_slots_for(num_cands, dedupe_cap_slots);
        for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < tsize; i += gridDim.x * blockDim.x) dedupe_table[i] = ~0ull;
    }

    for (uint32_t ci = wave0; ci < num_cands; ci += nwaves) {
        if (todo && !todo[ci]) continue;                     // refined by k_local_region already (wave-uniform)
        const Candidate cd = cands[ci];
        const int work = __builtin_amdgcn_readfirstlane(cd.work);
        const int pyr = __builtin_amdgcn_readfirstlane(work_pyramids[work]);
        int mx = __builtin_amdgcn_readfirstlane(cd.x), my = __builtin_amdgcn_readfirstlane(cd.y);
        float sim = cd.score;
        bool alive = true;

        for (int l = g.levels - 2; l >= 0 && alive; --l) {
            const LevelGeom lv = g.lv[l];
            const TemplEntry e = entries[(size_t)pyr * g.levels + l];
            const int T = lv.T, W = lv.W, H = lv.H, Wd = lv.Wd, Hd = lv.Hd;
            const int border = 8 * T, offset = T / 2 + (T % 2 - 1);
            const int max_x = W - e.width - border, max_y = H - e.height - border;
            int x = mx * 2 + 1, y = my * 2 + 1;               // LL.cpp:1871-1880
            x = x > border ? x : border;  y = y > border ? y : border;
            x = x < max_x ? x : max_x;    y = y < max_y ? y : max_y;
            const int gx = x / T - 8, gy = y / T - 8;          // C division (truncation), LL.cpp:1380-1381
            const int off_x = gx * T, off_y = gy * T;
            const int nf = e.nf, nfp = e.nf_padded;
            uint32_t key;                                        // lane-local best (raw << 8 | 255 - index)
            // Fast path iff every feature's 16x16 window lies inside its plane (no feature is discarded by
            // LL.cpp:1394, no row wrap, no spill into the next phase) — decided once per candidate from the
            // entry's feature bounding box; true unless the template is oversized for the frame.
            const bool all_in = e.min_x >= 0 && e.min_y >= 0 && gx >= 0 && gy >= 0 &&
                                ((e.max_x + off_x) / T + 16 <= Wd) && ((e.max_y + off_y) / T + 16 <= Hd);
            if (all_in) {
                const int half = lane >> 5, l5 = lane & 31, r = l5 >> 1, h = l5 & 1;
                const uint32_t HS = (uint32_t)Hd * 16u;
                const uint8_t* smp = sm_arena + (h * HS + r * 16);
                uint32_t w[4] = {0, 0, 0, 0};                   // u16x2: cols (0,2) (1,3) (4,6) (5,7) of this lane's 8 columns
                if (nfp > 0) {
                    const FeatStrip* fs = feat_strip + e.feat_start;
                    uint32_t r8[4] = {0, 0, 0, 0};
                    int cur = -1, cnt = 0;
                    auto flush = [&](int cls) {
                        uint32_t own[4], par[4];
#pragma unroll
                        for (int k = 0; k < 4; ++k) {
                            own[k] = r8[k] + (uint32_t)__shfl_xor((int)r8[k], 32, 64);   // both features of the pair
                            par[k] = (uint32_t)__shfl_xor((int)own[k], 1, 64);            // the other strip of this row
                            r8[k] = 0;
                        }
                        // bytes [c0 + 8h, +8) of the 32-byte row {strip S0, strip S0+1}; c0 = window start inside strip S0
                        const int c0 = (cls + gx) & 15;
                        uint32_t u6[6];
                        u6[0] = h ? par[2] : own[0]; u6[1] = h ? par[3] : own[1]; u6[2] = h ? own[0] : own[2];
                        u6[3] = h ? own[1] : own[3]; u6[4] = h ? own[2] : par[0]; u6[5] = h ? own[3] : par[1];
                        const uint32_t sb = (uint32_t)(c0 & 3);
                        uint32_t o0, o1;
                        switch (c0 >> 2) {                       // wave-uniform
                            case 0: o0 = __builtin_amdgcn_alignbyte(u6[1], u6[0], sb); o1 = __builtin_amdgcn_alignbyte(u6[2], u6[1], sb); break;
                            case 1: o0 = __builtin_amdgcn_alignbyte(u6[2], u6[1], sb); o1 = __builtin_amdgcn_alignbyte(u6[3], u6[2], sb); break;
                            case 2: o0 = __builtin_amdgcn_alignbyte(u6[3], u6[2], sb); o1 = __builtin_amdgcn_alignbyte(u6[4], u6[3], sb); break;
                            default: o0 = __builtin_amdgcn_alignbyte(u6[4], u6[3], sb); o1 = __builtin_amdgcn_alignbyte(u6[5], u6[4], sb); break;
                        }
                        add_bytes(o0, w[0], w[1]);
                        add_bytes(o1, w[2], w[3]);
                    };
                    FeatStrip c[kFeatBatch];
#pragma unroll
                    for (int u = 0; u < kFeatBatch; ++u) c[u] = fs[u];                    // wave-uniform -> SMEM
                    for (int f = 0; f < nfp; f += kFeatBatch) {
                        const int fn = f + kFeatBatch < nfp ? f + kFeatBatch : f;           // prefetch the next batch
                        FeatStrip cn[kFeatBatch];
#pragma unroll
                        for (int u = 0; u < kFeatBatch; ++u) cn[u] = fs[fn + u];
                        uint4 v[kFeatBatch / 2];
#pragma unroll
                        for (int k = 0; k < kFeatBatch / 2; ++k) {
                            // features 2k (lanes 0-31) and 2k+1 (lanes 32-63): same alignment class by construction
                            const uint32_t xa = (c[2 * k].cell & 0xFFFF) + gx, ya = (c[2 * k].cell >> 16) + gy;
                            const uint32_t xb = (c[2 * k + 1].cell & 0xFFFF) + gx, yb = (c[2 * k + 1].cell >> 16) + gy;
                            const uint32_t ba = c[2 * k].sbase + ((xa >> 4) * Hd + ya) * 16u;
                            const uint32_t bb = c[2 * k + 1].sbase + ((xb >> 4) * Hd + yb) * 16u;
                            v[k] = ld_aligned16(smp + (half ? bb : ba));
                        }
#pragma unroll
                        for (int k = 0; k < kFeatBatch / 2; ++k) {
                            const int cls = (int)(c[2 * k].cell & 15);
                            if (cls != cur || cnt == 31) {
                                if (cur >= 0) flush(cur);
                                cur = cls; cnt = 0;
                            }
                            r8[0] += v[k].x; r8[1] += v[k].y; r8[2] += v[k].z; r8[3] += v[k].w;   // <= 31 x 4 per byte and half
                            ++cnt;
                        }
#pragma unroll
                        for (int u = 0; u < kFeatBatch; ++u) c[u] = cn[u];
                    }
                    if (cur >= 0) flush(cur);
                }
                const uint32_t idx0 = (uint32_t)(r * 16 + 8 * h);
                const uint32_t raws[8] = {w[0] & 0xFFFF, w[1] & 0xFFFF, w[0] >> 16, w[1] >> 16,
                                          w[2] & 0xFFFF, w[3] & 0xFFFF, w[2] >> 16, w[3] >> 16};
                key = 0;
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    uint32_t kk = (raws[k] << 8) | (255u - (idx0 + k));
                    key = kk > key ? kk : key;
                }
            } else {
                const int row = lane >> 2, col = (lane & 3) * 4;
                const int32_t* fo = feat_off + e.feat_start;
                const uint32_t* fxy = feat_xy + e.feat_start;
                const uint8_t* base = lm_arena + ((int64_t)gy * Wd + gx + row * Wd + col);
                uint32_t even = 0, odd = 0;
                for (int f = 0; f < nfp; ++f) {
                    uint32_t xy = fxy[f];
                    int fx = (int16_t)(xy & 0xFFFF) + off_x, fy = (int16_t)(xy >> 16) + off_y;
                    if (fx < 0 || fy < 0 || fx >= W || fy >= H) continue;   // LL.cpp:1394 (padding: x = y = -32768)
                    uint32_t v = ld_u32(base + fo[f]);
                    add_bytes(v, even, odd);
                }
                const uint32_t idx0 = (uint32_t)(row * 16 + col);
                uint32_t k0 = ((even & 0xFFFF) << 8) | (255u - idx0);
                uint32_t k1 = ((odd & 0xFFFF) << 8) | (255u - (idx0 + 1));
                uint32_t k2 = ((even >> 16) << 8) | (255u - (idx0 + 2));
                uint32_t k3 = ((odd >> 16) << 8) | (255u - (idx0 + 3));
                uint32_t ka = k0 > k1 ? k0 : k1, kb = k2 > k3 ? k2 : k3;
                key = ka > kb ? ka : kb;
            }
            ++evals;
            bytes += 256ull * nf;   // algorithmic response bytes of this 16x16 evaluation
            const uint32_t k = wave_max_u32(key);
            const int raw = (int)(k >> 8);
            int br = -1, bc = -1;                               // LL.cpp:1910-1911
            float best = 0.f;
            if (raw > 0) {
                int idx = 255 - (int)(k & 0xFF);
                br = idx >> 4; bc = idx & 15;
                best = score_of(raw, nf);
            }
            sim = best;
            mx = (x / T - 8 + bc) * T + offset;                 // LL.cpp:1930-1931
            my = (y / T - 8 + br) * T + offset;
            if (sim < threshold) alive = false;                 // remove_if(MatchPredicate), LL.cpp:1935
        }
        // No atomics: the result of candidate ci goes to slot ci (work = -1 when it dropped below the
        // threshold on the way up, LL.cpp:1935); the host compacts.  matches[] is pinned host memory.
        if (lane == 0 && ci < cap) {
            Candidate m;
            m.x = mx; m.y = my; m.score = sim; m.work = alive ? work : -1;
            matches[ci] = m;
            matches_dev[ci] = m;                 // HBM copy for the on-device NMS / top-K (pipeline.cpp)
        }
    }
    // per-block statistics (16x16 evaluations, their algorithmic bytes): plain stores, summed on the host
    if (lane == 0) { s_stats[threadIdx.x >> 6][0] = evals; s_stats[threadIdx.x >> 6][1] = bytes; }
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned long long a = 0, b = 0;
        for (int w = 0; w < (int)(blockDim.x >> 6); ++w) { a += s_stats[w][0]; b += s_stats[w][1]; }
        block_stats[8 + 2 * blockIdx.x] = a;
        block_stats[8 + 2 * blockIdx.x + 1] = b;
    }
}

void launch_local(const uint8_t* lm_arena, const uint8_t* sm_arena, const FrameGeom& g, const TemplEntry* entries,
                  const int32_t* feat_off, const FeatStrip* feat_strip, const uint32_t* feat_xy,
                  const int32_t* work_pyramids, const Candidate* cands, uint32_t cand_cap,
                  float threshold, Candidate* matches, Candidate* matches_dev, uint32_t cap, const unsigned long long* counters,
                  unsigned long long* block_stats, unsigned long long* dedupe_table, uint32_t dedupe_cap_slots, const uint8_t* todo,
                  int grid_blocks, hipStream_t s) {
    if (grid_blocks <= 0) return;
    hipLaunchKernelGGL(k_local, dim3(grid_blocks), dim3(256), 0, s, lm_arena, sm_arena, g, entries, feat_off, feat_strip,
                       feat_xy, work_pyramids, cands, cand_cap, threshold, matches, matches_dev, cap, counters, block_stats,
                       dedupe_table, dedupe_cap_slots, todo);
}

// ---------------------------------------------------------------------------------------------
// Region refinement (two-level pyramids): the coarse candidates of ONE template sit a few cells apart, so their 16x16
// windows overlap — on the bench frame their union has 3x fewer cells than their sum (tests/analysis/candidate_clusters.py).
// One wave per template: its candidates (k_coarse's per-template list) are grouped into regions of <= 48 columns x 64 rows
// around a seed; the similarity of the whole region is accumulated ONCE — lane = row, per feature the 16-byte rows of the
// 3-4 strips the region touches, features of one alignment class summed in packed bytes and realigned once per class, like
// the coarse pass — into u16 sums in LDS, and every candidate then takes the first strict maximum of its own window from
// there.  The sums are the integers the per-candidate kernel computes, so the records are identical.  Candidates on the
// slow path, templates with more than kRegionK candidates and deeper pyramids are left to k_local (todo stays 1).
// STATUS: exact (the parity tests pass with LM_REGION=1) but OFF by default: 0.39 ms against k_local's 0.17 ms on the bench
// frame.  Two versions were measured.  (1) one pass per region, lane = row, one load per feature and strip: half the cache
// lines of k_local but 1.6x MORE load instructions, 0.38 ms.  (2) this one: every 64-lane load carries (region, row, strip)
// slots of all regions of the template, ~2.5 loads per feature and template instead of k_local's 3.7: 0.39 ms.  So it is not
// the vector L1 that bounds these kernels but their own overhead per template — wave 0 building the regions while three waves
// wait, per-lane slot descriptors and address arithmetic, a per-lane byte realignment at the end of each of the ~16 class
// runs, three barriers, and 183 VGPRs (2 waves per SIMD) — about 100 us per template and workgroup, where k_local's tight
// per-candidate loop needs 11 us per candidate.  Closing that gap means making the setup as cheap as k_local's (regions and
// slots prepared by a separate pass, uniform slot shapes so that addresses and realignment are wave-uniform again).
// ---------------------------------------------------------------------------------------------
constexpr int kSlotGroups = 4;        // 64-lane load groups per feature: up to 256 (region, row, strip) slots per template
constexpr int kMaxRegions = 8;

__global__ void __launch_bounds__(256)
k_local_region(const uint8_t* __restrict__ sm_arena, FrameGeom g, const TemplEntry* __restrict__ entries, const FeatStrip* __restrict__ feat_strip,
               const int32_t* __restrict__ work_pyramids, const Candidate* __restrict__ cands, uint32_t cand_cap, float threshold,
               Candidate* __restrict__ matches, Candidate* __restrict__ matches_dev, uint32_t cap, uint32_t* __restrict__ tcount,
               const uint32_t* __restrict__ tlist, uint8_t* __restrict__ todo, int num_work, unsigned long long* __restrict__ counters) {
    // One workgroup per template.  Its candidates are grouped into regions (<= 48 columns x 64 rows around a seed); every
    // (region, row, 16-column strip) is a SLOT = one lane of a load, so that a 64-lane load carries the rows of several regions at
    // once: a region of NC 16-column chunks takes NC + 1 adjacent lanes per row (the strips a misaligned feature plane needs), rows
    // never straddle a 64-lane group.  Per feature: one 16-byte load per slot; packed byte sums per alignment class; at the end of
    // a class run every lane realigns {its strip, the next lane's strip} by the class's byte phase in ITS region and adds the 16
    // bytes to its u16 column sums.  The four waves split the features (one LDS plane of sums each) and then the members.
    __shared__ uint16_t s_sum[4][kSlotGroups * 64 * 16];
    __shared__ int s_reg[kMaxRegions][8];                      // X0, Y0, Hr, NC, first slot, LDS offset of the sums
    __shared__ int s_nreg, s_groups;
    __shared__ unsigned char s_regof[64];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const LevelGeom lv = g.lv[0];
    const int T = lv.T, W = lv.W, H = lv.H, Wd = lv.Wd, Hd = lv.Hd;
    const int border = 8 * T, offset = T / 2 + (T % 2 - 1);
    unsigned long long evals = 0, bytes = 0;
    uint16_t* sum = s_sum[wv];

    for (int work = blockIdx.x; work < num_work; work += gridDim.x) {
        const uint32_t n = tcount[work];
        __syncthreads();                                      // every wave has read the count (and is done with the previous template's LDS)
        if (n == 0) continue;
        if (threadIdx.x == 0) tcount[work] = 0;               // the list is consumed: empty for the slot's next frame
        if (n > (uint32_t)kRegionK) continue;                 // more candidates than the list holds: all of them stay with k_local
        const int pyr = __builtin_amdgcn_readfirstlane(work_pyramids[work]);
        const TemplEntry e = entries[(size_t)pyr * g.levels];
        const int nf = e.nf, nfp = e.nf_padded;
        const int max_x = W - e.width - border, max_y = H - e.height - border;
        // this lane's candidate (every wave holds the same list)
        uint32_t ci = 0;
        int gx = 0, gy = 0, px = 0, py = 0;
        bool open = false;
        if (lane < (int)n) {
            ci = tlist[(size_t)work * kRegionK + lane];
            const Candidate cd = cands[ci];
            int x = cd.x * 2 + 1, y = cd.y * 2 + 1;             // LL.cpp:1871-1880
            x = x > border ? x : border;  y = y > border ? y : border;
            x = x < max_x ? x : max_x;    y = y < max_y ? y : max_y;
            px = x; py = y;
            gx = x / T - 8; gy = y / T - 8;
            open = ci < cap && e.min_x >= 0 && e.min_y >= 0 && gx >= 0 && gy >= 0 && ((e.max_x + gx * T) / T + 16 <= Wd) &&
                   ((e.max_y + gy * T) / T + 16 <= Hd);          // the fast-path test of k_local
        }
        // ---- regions and their slots (wave 0) ----
        if (wv == 0) {
            int nreg = 0, pos = 0, lds = 0, regof = 255;
            bool rem = open;
            while (nreg < kMaxRegions) {
                const unsigned long long pending = __ballot(rem);
                if (!pending) break;
                const int seed = __ffsll((long long)pending) - 1;
                const int sgx = __shfl(gx, seed, 64), sgy = __shfl(gy, seed, 64);
                const bool member = rem && gx >= sgx - 16 && gx <= sgx + 16 && gy >= sgy - 24 && gy <= sgy + 24;
                int X0 = member ? gx : INT_MAX, X1 = member ? gx + 16 : INT_MIN, Y0 = member ? gy : INT_MAX, Y1 = member ? gy + 16 : INT_MIN;
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) {
                    X0 = min(X0, __shfl_xor(X0, o, 64)); X1 = max(X1, __shfl_xor(X1, o, 64));
                    Y0 = min(Y0, __shfl_xor(Y0, o, 64)); Y1 = max(Y1, __shfl_xor(Y1, o, 64));
                }
                const int Wr = X1 - X0, Hr = Y1 - Y0;         // <= 48, <= 64 by construction of `member`
                const int NC = (Wr + 15) >> 4, w = NC + 1, rpg = 64 / w;
                const int off = pos & 63, rf = (64 - off) / w;
                int end;
                if (Hr <= rf) end = pos + Hr * w;
                else { const int rest = Hr - rf - 1; end = ((pos >> 6) + 1 + rest / rpg) * 64 + (rest % rpg) * w + w; }
                if (end > kSlotGroups * 64) break;            // no slots left: the remaining candidates stay with k_local
                if (lane == 0) { s_reg[nreg][0] = X0; s_reg[nreg][1] = Y0; s_reg[nreg][2] = Hr; s_reg[nreg][3] = NC; s_reg[nreg][4] = pos; s_reg[nreg][5] = lds; }
                if (member) { regof = nreg; rem = false; }
                pos = end; lds += Hr * 16 * NC; ++nreg;
            }
            s_regof[lane] = (unsigned char)regof;
            if (lane == 0) { s_nreg = nreg; s_groups = (pos + 63) >> 6; }
        }
        __syncthreads();
        const int nreg = s_nreg, G = s_groups;
        if (nreg == 0) continue;                               // uniform
        // ---- this lane's slot in each load group ----
        int dX0[kSlotGroups], dRow[kSlotGroups], dS[kSlotGroups], dLds[kSlotGroups];
        bool dValid[kSlotGroups], dOwn[kSlotGroups];           // a slot exists / it owns 16 region columns (not the extra right-hand strip)
#pragma unroll
        for (int q = 0; q < kSlotGroups; ++q) { dX0[q] = 0; dRow[q] = 0; dS[q] = 0; dLds[q] = 0; dValid[q] = false; dOwn[q] = false; }
        for (int r = 0; r < nreg; ++r) {
            const int X0 = s_reg[r][0], Y0 = s_reg[r][1], Hr = s_reg[r][2], NC = s_reg[r][3], pos = s_reg[r][4], lds = s_reg[r][5];
            const int w = NC + 1, rpg = 64 / w, off = pos & 63, g0 = pos >> 6, rf = (64 - off) / w;
#pragma unroll
            for (int q = 0; q < kSlotGroups; ++q) {
                int row = -1, sidx = 0;
                if (q == g0) {
                    const int local = lane - off;
                    if (local >= 0) { row = local / w; sidx = local - row * w; if (row >= rf) row = -1; }
                } else if (q > g0) {
                    const int j = lane / w;
                    sidx = lane - j * w;
                    if (j < rpg) row = rf + (q - g0 - 1) * rpg + j;
                }
                if (row >= 0 && row < Hr) {
                    dX0[q] = X0; dRow[q] = Y0 + row; dS[q] = sidx; dValid[q] = true; dOwn[q] = sidx < NC;
                    dLds[q] = lds + row * 16 * NC + 16 * sidx;
                }
            }
        }
        // ---- this wave's share of the features ----
        uint32_t accE[kSlotGroups][4], accO[kSlotGroups][4], r8[kSlotGroups][4];
#pragma unroll
        for (int q = 0; q < kSlotGroups; ++q)
#pragma unroll
            for (int k = 0; k < 4; ++k) { accE[q][k] = 0; accO[q][k] = 0; r8[q][k] = 0; }
        {
            const FeatStrip* fs = feat_strip + e.feat_start;
            int cur = -1, cnt = 0;
            auto flush = [&](int cls) {
#pragma unroll
                for (int q = 0; q < kSlotGroups; ++q) {
                    if (q < G) {                               // uniform
                        const int c0 = (cls + dX0[q]) & 15;   // byte phase of this class in this lane's region
                        const int dsel = c0 >> 2;
                        const uint32_t sb = (uint32_t)(c0 & 3);
                        uint32_t x[9];
#pragma unroll
                        for (int k = 0; k < 4; ++k) { x[k] = r8[q][k]; x[4 + k] = (uint32_t)__shfl_down((int)r8[q][k], 1, 64); r8[q][k] = 0; }
                        x[8] = 0;
                        uint32_t y[5];
#pragma unroll
                        for (int k = 0; k < 5; ++k) y[k] = dsel == 0 ? x[k] : dsel == 1 ? x[k + 1] : dsel == 2 ? x[k + 2] : x[k + 3];
#pragma unroll
                        for (int k = 0; k < 4; ++k) add_bytes(__builtin_amdgcn_alignbyte(y[k + 1], y[k], sb), accE[q][k], accO[q][k]);
                    }
                }
            };
            for (int f = 4 * wv; f < nfp; f += 16) {           // nf_padded is a multiple of kFeatBatch (8): batches of 4, dealt round-robin to the waves
                FeatStrip c[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) c[u] = fs[f + u];                           // wave-uniform -> SMEM
                uint4 v[4][kSlotGroups];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const uint32_t lx = c[u].cell & 0xFFFF, ly = c[u].cell >> 16;
#pragma unroll
                    for (int q = 0; q < kSlotGroups; ++q) {
                        v[u][q] = make_uint4(0, 0, 0, 0);
                        if (q < G && dValid[q]) {
                            const uint32_t pc = lx + (uint32_t)dX0[q];
                            v[u][q] = ld_aligned16(sm_arena + c[u].sbase + ((((pc >> 4) + (uint32_t)dS[q]) * (uint32_t)Hd) + ly + (uint32_t)dRow[q]) * 16u);
                        }
                    }
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int cls = (int)(c[u].cell & 15);
                    if (cls != cur || cnt == 63) {
                        if (cur >= 0) flush(cur);
                        cur = cls; cnt = 0;
                    }
#pragma unroll
                    for (int q = 0; q < kSlotGroups; ++q) { r8[q][0] += v[u][q].x; r8[q][1] += v[u][q].y; r8[q][2] += v[u][q].z; r8[q][3] += v[u][q].w; }
                    ++cnt;
                }
            }
            if (cur >= 0) flush(cur);
        }
        // this wave's partial sums to its LDS plane: 16 columns per owning slot
#pragma unroll
        for (int q = 0; q < kSlotGroups; ++q) {
            if (q < G && dOwn[q]) {
                uint16_t* o = sum + dLds[q];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    o[4 * k] = (uint16_t)(accE[q][k] & 0xFFFF); o[4 * k + 1] = (uint16_t)(accO[q][k] & 0xFFFF);
                    o[4 * k + 2] = (uint16_t)(accE[q][k] >> 16); o[4 * k + 3] = (uint16_t)(accO[q][k] >> 16);
                }
            }
        }
        __syncthreads();
        // ---- the members, dealt to the waves: first strict maximum of the member's own window (LL.cpp:1910-1931) ----
        const int myreg = s_regof[lane];
        unsigned long long mm = __ballot(myreg != 255);
        int turn = 0;
        while (mm) {
            const int m = __ffsll((long long)mm) - 1;
            mm &= mm - 1;
            if ((turn++ & 3) != wv) continue;
            const int r = __shfl(myreg, m, 64);
            const int X0 = s_reg[r][0], Y0 = s_reg[r][1], stride = 16 * s_reg[r][3], lds = s_reg[r][5];
            const int wx = __shfl(gx, m, 64) - X0, wy = __shfl(gy, m, 64) - Y0;
            uint32_t key = 0;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int idx = lane * 4 + k, rr = idx >> 4, cc = idx & 15;
                const int at = lds + (wy + rr) * stride + wx + cc;
                const uint32_t raw = (uint32_t)s_sum[0][at] + s_sum[1][at] + s_sum[2][at] + s_sum[3][at];
                const uint32_t kk = (raw << 8) | (255u - (uint32_t)idx);
                key = kk > key ? kk : key;
            }
            const uint32_t kbest = wave_max_u32(key);
            const int raw = (int)(kbest >> 8);
            int br = -1, bc = -1;
            float best = 0.f;
            if (raw > 0) {
                const int idx = 255 - (int)(kbest & 0xFF);
                br = idx >> 4; bc = idx & 15;
                best = score_of(raw, nf);
            }
            const int mpx = __shfl(px, m, 64), mpy = __shfl(py, m, 64);
            const uint32_t mci = (uint32_t)__shfl((int)ci, m, 64);
            if (lane == 0) {
                Candidate out;
                out.x = (mpx / T - 8 + bc) * T + offset;       // LL.cpp:1930-1931
                out.y = (mpy / T - 8 + br) * T + offset;
                out.score = best;
                out.work = best < threshold ? -1 : work;       // remove_if(MatchPredicate), LL.cpp:1935
                matches[mci] = out;
                matches_dev[mci] = out;
                todo[mci] = 0;
            }
            ++evals;
            bytes += 256ull * nf;
        }
    }
    if (lane == 0 && evals) { atomicAdd(&counters[4], evals); atomicAdd(&counters[5], bytes); }
}

void launch_local_region(const uint8_t* sm_arena, const FrameGeom& g, const TemplEntry* entries, const FeatStrip* feat_strip,
                         const int32_t* work_pyramids, const Candidate* cands, uint32_t cand_cap, float threshold, Candidate* matches,
                         Candidate* matches_dev, uint32_t cap, uint32_t* tcount, const uint32_t* tlist, uint8_t* todo, int num_work,
                         unsigned long long* counters, int grid_blocks, hipStream_t s) {
    if (grid_blocks <= 0 || num_work <= 0) return;
    hipLaunchKernelGGL(k_local_region, dim3(grid_blocks), dim3(256), 0, s, sm_arena, g, entries, feat_strip, work_pyramids, cands, cand_cap,
                       threshold, matches, matches_dev, cap, tcount, tlist, todo, num_work, counters);
}

}  // namespace lm
