// Similarity kernels of Detector::matchClass on gfx950 (reference LL.cpp:1284-1428 similarity /
// similarityLocal, LL.cpp:1788-1941 matchClass).  Pure integer byte work: every response is a
// u8 in {0,1,4}, summed per template position.  Nothing here is a dense contraction, so no MFMA;
// the linear memories of one frame (1.2 MB coarse + 4.9 MB fine at VGA) live in L2 / Infinity
// Cache and the kernels are bound by the L2->L1 gather path, so the design goal is to keep many
// independent gathers in flight per wave (kFeatBatch per batch) rather than arithmetic.
//
// Data layout (built by frontend.hip / detector.cpp, offsets in FrameGeom):
//   LM arena: per level, per modality: u8 [8 labels][T*T phases][(W/T)*(H/T)] + zero tail.
//   Bank:     TemplEntry per (pyramid, level).  Its features (colour then normal) are padded to a
//             multiple of kFeatBatch with entries that point at the level's zero tail, so the inner
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
// Coarse pass: one workgroup per template pyramid; each lane owns 16 consecutive positions of the
// decimated top-level grid and adds one (unaligned) 16-byte run of responses per feature, 8 features
// (8 x dwordx4) in flight, with the next batch of run offsets prefetched through the scalar cache
// while the gathers are outstanding.  Bytes are accumulated in packed u16x2 registers (even / odd
// bytes); 2 x 8191 x 4 < 65536 so neither modality split nor widening is needed (the reference's
// 8-bit and 16-bit paths give the same sums).
// ---------------------------------------------------------------------------------------------
static __device__ __forceinline__ uint4 ld_u128(const uint8_t* p) {
    uint4 v;
    __builtin_memcpy(&v, p, 16);   // unaligned global_load_dwordx4
    return v;
}

constexpr int kCoarsePos = 16;     // positions per lane

__global__ void __launch_bounds__(1024)
k_coarse(const uint8_t* __restrict__ lm_arena, LevelGeom lv, int level, int levels,
         const TemplEntry* __restrict__ entries, const int32_t* __restrict__ feat_off,
         const int32_t* __restrict__ work_pyramids, float threshold, Candidate* __restrict__ cands, uint32_t cap,
         unsigned long long* __restrict__ counters) {
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

    for (int j0 = threadIdx.x * kCoarsePos; j0 < npos; j0 += blockDim.x * kCoarsePos) {
        uint32_t even[4] = {0, 0, 0, 0}, odd[4] = {0, 0, 0, 0};
        if (j0 < tp && nfp > 0) {
            const uint8_t* base = lm_arena + j0;
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
                for (int u = 0; u < kFeatBatch; ++u) v[u] = ld_u128(base + o[u]);
#pragma unroll
                for (int u = 0; u < kFeatBatch; ++u) {
                    even[0] += v[u].x & 0x00FF00FFu; odd[0] += (v[u].x >> 8) & 0x00FF00FFu;
                    even[1] += v[u].y & 0x00FF00FFu; odd[1] += (v[u].y >> 8) & 0x00FF00FFu;
                    even[2] += v[u].z & 0x00FF00FFu; odd[2] += (v[u].z >> 8) & 0x00FF00FFu;
                    even[3] += v[u].w & 0x00FF00FFu; odd[3] += (v[u].w >> 8) & 0x00FF00FFu;
                }
#pragma unroll
                for (int u = 0; u < kFeatBatch; ++u) o[u] = on[u];
            }
        }
#pragma unroll
        for (int k = 0; k < kCoarsePos; ++k) {
            const int j = j0 + k;
            const uint32_t pk = (k & 1) ? odd[k >> 2] : even[k >> 2];
            const int rawk = (int)((k & 2) ? (pk >> 16) : (pk & 0xFFFF));
            if (j < npos) {
                int r = j < tp ? rawk : 0;                // positions >= template_positions stay 0
                float sc = score_of(r, nf);
                if (sc > threshold) {                     // LL.cpp:1844
                    unsigned long long slot = atomicAdd(&counters[0], 1ull);
                    if (slot < cap) {
                        int cy = j / Wd, cx = j - cy * Wd;
                        Candidate c;
                        c.x = cx * T + offset; c.y = cy * T + offset; c.score = sc; c.work = work;
                        cands[slot] = c;
                    }
                }
            }
        }
    }
}

void launch_coarse(const uint8_t* lm_arena, const FrameGeom& g, const TemplEntry* entries, const int32_t* feat_off,
                   const int32_t* work_pyramids, int num_work, float threshold, Candidate* cands, uint32_t cap,
                   unsigned long long* counters, hipStream_t s) {
    if (num_work <= 0) return;
    const int level = g.levels - 1;
    const LevelGeom lv = g.lv[level];
    int npos = lv.Wd * lv.Hd;
    int threads = ((npos + kCoarsePos - 1) / kCoarsePos + 63) / 64 * 64;
    if (threads > 1024) threads = 1024;
    if (threads < 64) threads = 64;
    hipLaunchKernelGGL(k_coarse, dim3(num_work), dim3(threads), 0, s, lm_arena, lv, level, g.levels, entries, feat_off,
                       work_pyramids, threshold, cands, cap, counters);
}

// ---------------------------------------------------------------------------------------------
// Local refinement: one wave64 per candidate, marching up the pyramid (LL.cpp:1855-1938).  Lane l
// owns row l/4, columns 4*(l%4)..+3 of the 16x16 patch; per feature it adds one dword of the linear
// memory at  run + (y/T-8)*Wd + (x/T-8) + row*Wd + col.  First strict maximum (LL.cpp:1920) is found
// with a packed key (raw<<8 | 255-index) and a wave max-reduction.  The grid is persistent: waves
// stride over the candidate list whose length is read from device memory (no host round trip
// between the coarse and the local pass).
//
// Fast path (every window inside its plane, i.e. always except for oversized templates): the gather
// reads the strip-major copy of the linear memories ([strip of 16 columns][row][16 B]).  A window row
// is then two aligned-dword pieces inside at most two strips whose 16 rows are contiguous, so one
// feature costs ~5 L2 lines instead of ~16 (PMC: the flat layout was L1-miss bound at 14 L2 requests
// per gather instruction).  Each lane loads the two aligned dwords around its 4 window bytes and
// funnel-shifts them (v_alignbyte) by the wave-uniform byte phase of the window start.
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
        float threshold, Candidate* __restrict__ matches, uint32_t cap, unsigned long long* __restrict__ counters) {
    const int lane = threadIdx.x & 63;
    const uint32_t wave0 = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const uint32_t nwaves = gridDim.x * (blockDim.x >> 6);
    unsigned long long nc = counters[0];
    const uint32_t num_cands = nc < cand_cap ? (uint32_t)nc : cand_cap;
    const int row = lane >> 2, col = (lane & 3) * 4;
    unsigned long long evals = 0, bytes = 0;

    for (uint32_t ci = wave0; ci < num_cands; ci += nwaves) {
        const Candidate cd = cands[ci];
        const int work = __builtin_amdgcn_readfirstlane(cd.work);
        const int pyr = __builtin_amdgcn_readfirstlane(work_pyramids[work]);
        int mx = __builtin_amdgcn_readfirstlane(cd.x), my = __builtin_amdgcn_readfirstlane(cd.y);
        float sim = cd.score;
        bool alive = true;

        for (int l = g.levels - 2; l >= 0 && alive; --l) {
            const LevelGeom lv = g.lv[l];
            const TemplEntry e = entries[(size_t)pyr * g.levels + l];
            const int T = lv.T, W = lv.W, H = lv.H, Wd = lv.Wd;
            const int border = 8 * T, offset = T / 2 + (T % 2 - 1);
            const int max_x = W - e.width - border, max_y = H - e.height - border;
            int x = mx * 2 + 1, y = my * 2 + 1;               // LL.cpp:1871-1880
            x = x > border ? x : border;  y = y > border ? y : border;
            x = x < max_x ? x : max_x;    y = y < max_y ? y : max_y;
            const int gx = x / T - 8, gy = y / T - 8;          // C division (truncation), LL.cpp:1380-1381
            const int off_x = gx * T, off_y = gy * T;
            const int nf = e.nf, nfp = e.nf_padded;
            const int32_t* fo = feat_off + e.feat_start;
            const uint8_t* base = lm_arena + ((int64_t)gy * Wd + gx + row * Wd + col);
            uint32_t even = 0, odd = 0;
            // Fast path iff every feature's 16x16 window lies inside its plane (no feature is discarded by
            // LL.cpp:1394, no row wrap, no spill into the next phase) — decided once per candidate from the
            // entry's feature bounding box; true unless the template is oversized for the frame.
            const bool all_in = (e.min_x + off_x >= 0) && (e.min_y + off_y >= 0) && gx >= 0 && gy >= 0 &&
                                ((e.max_x + off_x) / T + 16 <= Wd) && ((e.max_y + off_y) / T + 16 <= lv.Hd);
            if (all_in && nfp > 0) {
                const FeatStrip* fs = feat_strip + e.feat_start;
                const uint8_t* sm = sm_arena;
                const int Hd = lv.Hd, HS = Hd * 16;
                const int q = lane & 3, r16 = row * 16;
                FeatStrip c[kFeatBatch];
#pragma unroll
                for (int u = 0; u < kFeatBatch; ++u) c[u] = fs[u];                    // wave-uniform -> SMEM
                for (int f = 0; f < nfp; f += kFeatBatch) {
                    const int fn = f + kFeatBatch < nfp ? f + kFeatBatch : f;           // prefetch the next batch
                    FeatStrip cn[kFeatBatch];
#pragma unroll
                    for (int u = 0; u < kFeatBatch; ++u) cn[u] = fs[fn + u];
                    uint32_t va[kFeatBatch], vb[kFeatBatch];
                    int sh[kFeatBatch];
#pragma unroll
                    for (int u = 0; u < kFeatBatch; ++u) {
                        const int X0 = (int)(c[u].cell & 0xFFFF) + gx, Y0 = (int)(c[u].cell >> 16) + gy;   // window origin in the plane
                        const int A4 = X0 >> 2;                           // first aligned dword of the window row
                        const uint32_t sbase = c[u].sbase + (uint32_t)(((A4 >> 2) * Hd + Y0) * 16);
                        const int t = q + (A4 & 3), tb = t + 1;           // aligned dword index inside the 2-strip span
                        const uint8_t* pa = sm + sbase + ((t >> 2) * HS + (t & 3) * 4 + r16);
                        const uint8_t* pb = sm + sbase + ((tb >> 2) * HS + (tb & 3) * 4 + r16);
                        va[u] = *(const uint32_t*)pa;
                        vb[u] = *(const uint32_t*)pb;
                        sh[u] = X0 & 3;
                    }
#pragma unroll
                    for (int u = 0; u < kFeatBatch; ++u) {
                        const uint32_t v = __builtin_amdgcn_alignbyte(vb[u], va[u], (uint32_t)sh[u]);
                        even += v & 0x00FF00FFu;
                        odd += (v >> 8) & 0x00FF00FFu;
                    }
#pragma unroll
                    for (int u = 0; u < kFeatBatch; ++u) c[u] = cn[u];
                }
            } else if (!all_in) {
                const uint32_t* fxy = feat_xy + e.feat_start;
                for (int f = 0; f < nfp; ++f) {
                    uint32_t xy = fxy[f];
                    int fx = (int16_t)(xy & 0xFFFF) + off_x, fy = (int16_t)(xy >> 16) + off_y;
                    if (fx < 0 || fy < 0 || fx >= W || fy >= H) continue;   // LL.cpp:1394 (padding: x = y = -32768)
                    uint32_t v = ld_u32(base + fo[f]);
                    even += v & 0x00FF00FFu;
                    odd += (v >> 8) & 0x00FF00FFu;
                }
            }
            ++evals;
            bytes += 256ull * nf;   // algorithmic response bytes of this 16x16 evaluation
            const uint32_t idx0 = (uint32_t)(row * 16 + col);
            uint32_t k0 = ((even & 0xFFFF) << 8) | (255u - idx0);
            uint32_t k1 = ((odd & 0xFFFF) << 8) | (255u - (idx0 + 1));
            uint32_t k2 = ((even >> 16) << 8) | (255u - (idx0 + 2));
            uint32_t k3 = ((odd >> 16) << 8) | (255u - (idx0 + 3));
            uint32_t k = k0 > k1 ? k0 : k1;
            uint32_t kk = k2 > k3 ? k2 : k3;
            k = wave_max_u32(k > kk ? k : kk);
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
        if (lane == 0 && alive) {
            unsigned long long slot = atomicAdd(&counters[1], 1ull);
            if (slot < cap) {
                Candidate m;
                m.x = mx; m.y = my; m.score = sim; m.work = work;
                matches[slot] = m;
            }
        }
    }
    if (lane == 0 && evals) {
        atomicAdd(&counters[2], evals);
        atomicAdd(&counters[3], bytes);
    }
}

void launch_local(const uint8_t* lm_arena, const uint8_t* sm_arena, const FrameGeom& g, const TemplEntry* entries,
                  const int32_t* feat_off, const FeatStrip* feat_strip, const uint32_t* feat_xy,
                  const int32_t* work_pyramids, const Candidate* cands, uint32_t cand_cap,
                  float threshold, Candidate* matches, uint32_t cap, unsigned long long* counters, int grid_blocks,
                  hipStream_t s) {
    if (cand_cap == 0 || grid_blocks <= 0) return;
    hipLaunchKernelGGL(k_local, dim3(grid_blocks), dim3(256), 0, s, lm_arena, sm_arena, g, entries, feat_off, feat_strip,
                       feat_xy, work_pyramids, cands, cand_cap, threshold, matches, cap, counters);
}

}  // namespace lm
