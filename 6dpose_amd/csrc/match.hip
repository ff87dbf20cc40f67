// Similarity kernels of Detector::matchClass on gfx950 (reference LL.cpp:1284-1428 similarity /
// similarityLocal, LL.cpp:1788-1941 matchClass).  Pure integer byte work: every response is a
// u8 in {0,1,4}, summed per template position.  Nothing here is a dense contraction, so no MFMA;
// the linear memories of one frame (1.2 MB coarse + 4.9 MB fine at VGA) live in L2 / Infinity
// Cache; the byte kernels (k_coarse, k_local) are bound by the vector-L1 access rate (see "Gather discipline" below), the bit-plane refinement
// (k_local_bits, the default) by the number of 16-byte wave loads it issues — not by HBM and not by arithmetic.
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
#include <algorithm>

#include "knobs.h"
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

// The gathers of the three fast paths go through buffer loads: `buffer_load_dwordx4 v, v_off, s[rsrc], s_off offen` takes the
// arena as a resource in SGPRs, ONE 32-bit per-lane offset (constant for the whole item) and the feature's wave-uniform byte
// offset in an SGPR.  As global loads the compiler kept a 64-bit VGPR address per load in flight — 16 VGPRs for a batch of 8 and
// two VALU adds per load — which is what put k_local at 67 and k_coarse at 96 VGPRs (occupancy 7 / 5 waves per SIMD); the
// kernels' speed follows their occupancy (profiles/r02_local_experiments.txt).
using BufRsrc = __amdgpu_buffer_rsrc_t;
static __device__ __forceinline__ BufRsrc make_rsrc(const void* base) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, -1 /* no range check: 2^32 - 1 bytes */, 0x00020000 /* gfx9: raw dwords */);
}
static __device__ __forceinline__ uint4 ld_buf16(BufRsrc r, uint32_t lane_off, uint32_t uniform_off) {
    const auto v = __builtin_amdgcn_raw_buffer_load_b128(r, (int)lane_off, (int)uniform_off, 0);
    uint4 u;
    __builtin_memcpy(&u, &v, 16);
    return u;
}

// widen 4 packed bytes of `v` into two packed u16x2 accumulators: e gets bytes 0,2 ; o gets bytes 1,3
static __device__ __forceinline__ void add_bytes(uint32_t v, uint32_t& e, uint32_t& o) {
    e += v & 0x00FF00FFu;
    o += (v >> 8) & 0x00FF00FFu;
}

static __device__ __forceinline__ uint32_t wave_max_u32(uint32_t v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        uint32_t t = (uint32_t)__shfl_xor((int)v, o, 64);
        v = t > v ? t : v;
    }
    return v;
}

// ---------------------------------------------------------------------------------------------
// Coarse pass (LL.cpp:1284-1354 similarity + :1835-1852 scan): one workgroup per template pyramid.
// Lane i of wave w owns the 16 positions [16*(63w+i), +16) of the decimated top-level grid and loads
// the aligned 16-byte chunk that starts at or before them; the tail of its positions lives in lane
// i+1's chunk, hence 63 producing lanes per wave plus one feeder lane.
// ---------------------------------------------------------------------------------------------
constexpr int kChunksPerWave = 63;
constexpr int kMaxTilesPerTemplate = 48;       // tiles planned per template and frame (more candidates stay singles)
constexpr int kPlanHits = 64;                  // hits per template (raster order) the planner looks at: one per lane

// Everything the planner of k_coarse needs to know about the level below the top (two-level pyramids whose coarse cells are
// kTileStep fine cells apart; plan.enabled == 0 otherwise).
struct TilePlanGeom {
    int enabled;                               // the hits of a workgroup's templates are collected in LDS and their slots reserved with ONE atomic
    int tiles;                                 // ... and grouped into tiles (k_local's tile path); 0: every hit stays a single (bit-plane refinement: it needs no tiles)
    int W0, H0, T0, Wd0, Hd0;                  // level 0: image size, step, decimated grid
    uint32_t tile_cap;
    int dbg;                                   // LM_COARSE_DBG (timing experiments only): 1 = no grouping, 2 = no global atomic (wrong results)
};

static __device__ __forceinline__ uint32_t wave_sum_u32(uint32_t v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += (uint32_t)__shfl_xor((int)v, o, 64);
    return v;
}
static __device__ __forceinline__ unsigned long long bcast_u64(unsigned long long v, int src) {
    return ((unsigned long long)(uint32_t)__shfl((int)(v >> 32), src, 64) << 32) | (uint32_t)__shfl((int)(uint32_t)v, src, 64);
}
// exclusive prefix of `mine` over the lanes of the wave; total = the wave's sum
static __device__ __forceinline__ int wave_excl_scan(int mine, int lane, int& total) {
    int incl = mine;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        int t = __shfl_up(incl, o, 64);
        if (lane >= o) incl += t;
    }
    total = __shfl(incl, 63, 64);
    return incl - mine;
}

// A workgroup serves `group` templates at once, `wpt` waves each (blockDim = group * wpt * 64).  With tile planning ONE atomic
// per WORKGROUP reserves the candidate slots and the tile records of all its templates.  (Measured, 2000 templates at VGA: a
// workgroup per template with its own atomic on the frame's counter word takes 47 us, 17 us of them queueing at that word —
// the L2 serialises same-address atomics at ~120 per us — and a second atomic for the tiles another 20 us; without atomics the
// pass takes 31 us.)  The two counts share counters[0]: candidates in the low kCandBits bits, tiles above.
// Dynamic LDS when tiles are planned, per template of the group: [npos] f32 score of every hit | [nw] hit bitmap |
// kMaxTilesPerTemplate x 4 words of tile records | the first kPlanHits hit positions; then 4 words per template + 4 for the
// group's scan.
__global__ void __launch_bounds__(1024)
k_coarse(FrameBatch fb, LevelGeom lv, int level, int levels,
         const TemplEntry* __restrict__ entries, const int32_t* __restrict__ feat_off,
         const int32_t* __restrict__ work_pyramids, int num_work, int wpt, int cpw, float threshold, uint32_t cap, TilePlanGeom plan) {
    extern __shared__ uint32_t s_dyn[];
    const FrameSlot& F = fb.f[blockIdx.y];                        // the frame of the batch this workgroup serves
    const uint8_t* __restrict__ lm_arena = F.lm_arena;
    Candidate* __restrict__ cands = F.cands;
    unsigned long long* __restrict__ counters = F.counters;
    TileRec* __restrict__ tiles = F.tiles;
    uint8_t* __restrict__ todo = F.todo;
    const int lane = threadIdx.x & 63;
    const int wave_in_block = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));   // wave-uniform, and the compiler must know it: the
                                                                                         // template's entry and feature offsets then come through SMEM into SGPRs
    const int group = (int)(blockDim.x >> 6) / wpt;
    const int tslot = wave_in_block / wpt, wave = wave_in_block - tslot * wpt, nwaves = wpt;
    const int work_raw = blockIdx.x * group + tslot;
    const bool live = work_raw < num_work;
    const int work = live ? work_raw : num_work - 1;              // idle slots of the last workgroup shadow a real template and write nothing
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
    const int nw = (npos + 31) >> 5;
    const int per_template = npos + nw + 4 * kMaxTilesPerTemplate + kPlanHits;
    uint32_t* s_mine = s_dyn + (size_t)tslot * per_template;
    float* s_score = reinterpret_cast<float*>(s_mine);
    volatile uint32_t* s_hit = s_mine + npos;
    volatile uint32_t* s_tile = s_mine + npos + nw;
    volatile uint32_t* s_list = s_tile + 4 * kMaxTilesPerTemplate;
    volatile uint32_t* s_agg = s_dyn + (size_t)group * per_template;       // [group][4]: hits, tiles, first slot, first tile; then the atomic's result
    if (plan.enabled) {
        for (int i = wave * 64 + lane; i < nw; i += wpt * 64) s_hit[i] = 0;
        __syncthreads();
    }

    // A wave owns cpw <= kChunksPerWave consecutive 16-position chunks, one per lane; lane cpw reads the chunk after them (the
    // realignment of a run needs the next lane's bytes) and the lanes above it sit the loads out.  Only the chunks below the
    // template's position count tp carry sums: those are split EVENLY over as few waves as hold them — a template with tp <= 1008
    // keeps one wave (of ceil(tp / 16) + 1 lanes) and its second wave only scans zeros, 1200 positions are 38 + 37 lanes, not
    // 63 + 12 — because a 16-byte wave load costs the texture path ~24 cycles at 40 lanes, 31.5 at 64 and still 22.7 at 12
    // (profiles/r03_tcp_rotation_microbench.txt).
    {
        const int nch = ((tp < npos ? tp : npos) + 15) >> 4;
        const int wn = (nch + kChunksPerWave - 1) / kChunksPerWave;
        cpw = wn >= 1 && wn <= nwaves ? (nch + wn - 1) / wn : kChunksPerWave;
        if (cpw < 1) cpw = 1;
    }
    for (int chunk0 = wave * cpw; chunk0 * 16 < npos; chunk0 += nwaves * cpw) {
        const int j0 = (chunk0 + lane) * 16;                   // first position owned by this lane
        uint32_t even[4] = {0, 0, 0, 0}, odd[4] = {0, 0, 0, 0};
        if (chunk0 * 16 < tp && nfp > 0 && lane <= cpw) {      // wave-uniform but for the idle lanes
            const BufRsrc arena = make_rsrc(lm_arena);
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
                for (int u = 0; u < kFeatBatch; ++u) v[u] = ld_buf16(arena, (uint32_t)j0, (uint32_t)(o[u] & ~15));
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
        // Threshold scan (LL.cpp:1835-1852).
        auto raw_at = [&](int k) -> int {                        // the lane's k-th position sum (positions >= template_positions stay 0)
            const uint32_t pk = (k & 1) ? odd[k >> 2] : even[k >> 2];
            const int rawk = (int)((k & 2) ? (pk >> 16) : (pk & 0xFFFF));
            return j0 + k < tp ? rawk : 0;
        };
        uint32_t hit_mask = 0;
#pragma unroll
        for (int k = 0; k < 16; ++k)
            if (live && lane < cpw && j0 + k < npos && score_of(raw_at(k), nf) > threshold) hit_mask |= 1u << k;   // LL.cpp:1844
        if (plan.enabled) {
            // the hits of the whole template are collected in LDS; slots, tiles and records follow once all of them are known
            if (hit_mask) {
                atomicOr(const_cast<uint32_t*>(&s_hit[j0 >> 5]), hit_mask << (j0 & 31));
#pragma unroll
                for (int k = 0; k < 16; ++k) if (hit_mask & (1u << k)) s_score[j0 + k] = score_of(raw_at(k), nf);
            }
            continue;
        }
        // Without planning: one atomicAdd per WAVE reserves the slots of all its hits, then every lane writes its hits at the
        // reserved base + its exclusive prefix.
        int total;
        const int before = wave_excl_scan(__popc(hit_mask), lane, total);
        if (total > 0) {                                       // wave-uniform
            unsigned long long wbase = 0;
            if (lane == 0) wbase = atomicAdd(&counters[0], (unsigned long long)total);
            wbase = bcast_u64(wbase, 0);
            unsigned long long slot = wbase + (unsigned long long)before;
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                if (hit_mask & (1u << k)) {
                    if (slot < cap) {
                        const int j = j0 + k;
                        const int cy = j / Wd, cx = j - cy * Wd;
                        Candidate c;
                        c.x = cx * T + offset; c.y = cy * T + offset; c.score = score_of(raw_at(k), nf); c.work = work;
                        cands[slot] = c;
                    }
                    ++slot;
                }
            }
        }
    }
    if (!plan.enabled) return;

    // ---- planning (the first wave of every template): hits whose level-0 windows lie kTileStep cells apart are grouped into
    // tiles of kTileNx x kTileNy coarse cells (one accumulation of the shared window region in k_local instead of one per
    // candidate); the rest are singles (todo = 1).  See "Tile refinement" below.  The first kPlanHits hits (raster order) sit
    // one per lane and the grouping is ballots over those lanes — no serial pass over the grid.
    __syncthreads();
    const bool planner = wave == 0 && live;
    uint32_t nhit = 0;
    if (planner) {
        for (int w = lane; w < nw; w += 64) nhit += (uint32_t)__popc(s_hit[w]);
        nhit = wave_sum_u32(nhit);
    }
    int ntile = 0, nplan = 0, p = 0;
    uint32_t my_rel = 0;                                                    // this lane's slot relative to the template's first; top bit: a single
    if (planner && nhit > 0) {
        // hit h (raster order) -> lane h for h < kPlanHits (through LDS)
        uint32_t before = 0;
        for (int w0 = 0; w0 < nw && before < (uint32_t)kPlanHits; w0 += 64) {
            uint32_t w = (w0 + lane < nw) ? s_hit[w0 + lane] : 0u;
            int total;
            uint32_t h = before + (uint32_t)wave_excl_scan(__popc(w), lane, total);
            while (w && h < (uint32_t)kPlanHits) {
                const int b = __ffs((int)w) - 1;
                w &= w - 1;
                s_list[h++] = (uint32_t)((w0 + lane) * 32 + b);
            }
            before += (uint32_t)total;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        nplan = nhit < (uint32_t)kPlanHits ? (int)nhit : kPlanHits;
        const TemplEntry e0 = entries[(size_t)pyr * levels];
        const int T0 = plan.T0, border = 8 * T0;
        const int max_x = plan.W0 - e0.width - border, max_y = plan.H0 - e0.height - border;
        auto fine_x0 = [&](int c) { return (c * T + offset) * 2 + 1; };   // LL.cpp:1868-1869 for the coarse cell's candidate
        int cx = -100, cy = -100;
        bool elig = false;
        if (lane < nplan) p = (int)s_list[lane];
        if (lane < nplan && plan.tiles) {
            cy = p / Wd; cx = p - cy * Wd;
            // a tile may serve the hit iff LL.cpp:1871-1880 does not clamp it and it is on the fast path of the refinement (all
            // windows inside their planes) — from the entry's feature bounding box, exactly like k_local's `all_in`
            const int x = fine_x0(cx), y = fine_x0(cy);
            if (e0.min_x >= 0 && e0.min_y >= 0 && e0.nf_padded > 0 && x >= border && x <= max_x && y >= border && y <= max_y) {
                const int gx = x / T0 - 8, gy = y / T0 - 8;
                elig = gx >= 0 && gy >= 0 && (e0.max_x + gx * T0) / T0 + 16 <= plan.Wd0 && (e0.max_y + gy * T0) / T0 + 16 <= plan.Hd0;
            }
        }
        uint32_t next = 0;                                                  // slots handed out so far
        bool member = false;
        while (plan.tiles && ntile < kMaxTilesPerTemplate && !(plan.dbg & 1)) {
            const unsigned long long eb = __ballot(elig);
            if (!eb) break;
            const int seed = __ffsll((long long)eb) - 1;                   // first eligible hit in raster order: top row of its block
            const int r = __shfl(cy, seed, 64), c = __shfl(cx, seed, 64);
            // the block of kTileNx x kTileNy cells holding the seed and the most eligible hits (how far it reaches to the left)
            int best_n = 0, best_c0 = c;
#pragma unroll
            for (int dc = 0; dc < kTileNx; ++dc) {
                const int c0 = c - dc < 0 ? 0 : c - dc;
                const int n = __popcll(__ballot(elig && (cy == r || cy == r + 1) && cx >= c0 && cx < c0 + kTileNx));
                if (n > best_n) { best_n = n; best_c0 = c0; }
            }
            if (best_n < 2) {                                               // alone in its neighbourhood: a single
                if (lane == seed) elig = false;
                continue;
            }
            const bool in = elig && (cy == r || cy == r + 1) && cx >= best_c0 && cx < best_c0 + kTileNx;
            int c0 = best_c0;
#pragma unroll
            for (int sft = 0; sft < kTileNx - 1; ++sft)                     // tight on the left: a lone column of members needs two strips, not three
                if (!__ballot(in && cx == c0)) ++c0;
            const int bit = in ? (cx - c0) + kTileNx * (cy - r) : 0;
            uint32_t mask = in ? 1u << bit : 0u;                              // (members only: kTileMaskBits bits)
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) mask |= (uint32_t)__shfl_xor((int)mask, o, 64);
            if (in) { my_rel = next + (uint32_t)__popc(mask & ((1u << bit) - 1u)); member = true; elig = false; }
            if (lane == seed) {
                s_tile[4 * ntile + 0] = (uint32_t)work;
                s_tile[4 * ntile + 1] = (uint32_t)(uint16_t)(int16_t)(fine_x0(c0) / T0 - 8) | ((uint32_t)(uint16_t)(int16_t)(fine_x0(r) / T0 - 8) << 16);
                s_tile[4 * ntile + 2] = next;
                // window origins of the block's columns / second row relative to the first, in level-0 cells: 2 T_top / T_0 apart on
                // average, e.g. always 4 for T = {4, 8} and 3, 3, 3, 3, 4, ... for the reference's default T = {5, 8} (LL.cpp:1868-1881)
                uint32_t steps = 0;
#pragma unroll
                for (int i = 1; i < kTileNx; ++i) steps |= (uint32_t)(fine_x0(c0 + i) / T0 - fine_x0(c0 + i - 1) / T0) << (3 * (i - 1));
                steps |= (uint32_t)(fine_x0(r + 1) / T0 - fine_x0(r) / T0) << (3 * (kTileNx - 1));
                s_tile[4 * ntile + 3] = mask | (steps << kTileMaskBits);
            }
            next += (uint32_t)best_n;
            ++ntile;
        }
        const bool single = lane < nplan && !member;
        const unsigned long long sb = __ballot(single);
        if (single) my_rel = (next + (uint32_t)__popcll(sb & ((1ull << lane) - 1ull))) | 0x80000000u;
    }
    // the workgroup's templates line up: exclusive scan of their hit / tile counts, one atomic on the packed counter
    if (wave == 0 && lane == 0) { s_agg[4 * tslot] = nhit; s_agg[4 * tslot + 1] = (uint32_t)ntile; }
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned long long hs = 0, ts = 0;
        for (int t = 0; t < group; ++t) {
            const uint32_t h = s_agg[4 * t], n = s_agg[4 * t + 1];
            s_agg[4 * t + 2] = (uint32_t)hs; s_agg[4 * t + 3] = (uint32_t)ts;
            hs += h; ts += n;
        }
        unsigned long long got = 0;
        if (hs && !(plan.dbg & 2)) got = atomicAdd(&counters[0], hs | (ts << kCandBits));
        s_agg[4 * group] = (uint32_t)got; s_agg[4 * group + 1] = (uint32_t)(got >> 32);
    }
    __syncthreads();
    if (!planner || nhit == 0) return;
    const unsigned long long packed = (unsigned long long)s_agg[4 * group] | ((unsigned long long)s_agg[4 * group + 1] << 32);
    const unsigned long long base = (packed & kCandMask) + s_agg[4 * tslot + 2];
    const unsigned long long tbase = (packed >> kCandBits) + s_agg[4 * tslot + 3];
    auto put = [&](unsigned long long slot, int pos, int flag) {            // candidate record of coarse position pos
        if (slot >= cap) return;
        const int py = pos / Wd, px = pos - py * Wd;
        Candidate c;
        c.x = px * T + offset; c.y = py * T + offset; c.score = s_score[pos]; c.work = work;
        cands[slot] = c;
        todo[slot] = (uint8_t)flag;
    };
    if (lane < nplan) put(base + (my_rel & 0x7FFFFFFFu), p, (int)(my_rel >> 31));
    if (nhit > (uint32_t)kPlanHits) {                                       // hits beyond the planned ones: singles, slot = first + h
        uint32_t before = 0;
        for (int w0 = 0; w0 < nw; w0 += 64) {
            uint32_t w = (w0 + lane < nw) ? s_hit[w0 + lane] : 0u;
            int total;
            uint32_t h = before + (uint32_t)wave_excl_scan(__popc(w), lane, total);
            while (w) {
                const int b = __ffs((int)w) - 1;
                w &= w - 1;
                if (h >= (uint32_t)kPlanHits) put(base + h, (w0 + lane) * 32 + b, 1);
                ++h;
            }
            before += (uint32_t)total;
        }
    }
    if (lane < ntile && tiles && tbase + lane < plan.tile_cap) {
        TileRec t;
        t.work = (int32_t)s_tile[4 * lane]; t.gxy = s_tile[4 * lane + 1];
        t.slot_base = (uint32_t)(base + s_tile[4 * lane + 2]); t.mask = s_tile[4 * lane + 3];
        tiles[tbase + lane] = t;
    }
}

// Tiles are planned for two-level pyramids whose neighbouring coarse cells have level-0 windows at most kTileStep cells apart
// (T_0 <= T_top <= 2 T_0: T = {4, 8}, {2, 4}, and the reference's default {5, 8}, where the step alternates between 3 and 4 cells)
// and whose coarse grid fits the planner's LDS; anything else keeps the plain per-wave candidate emission.
bool tile_plan_possible(const FrameGeom& g) {
    if (g.levels != 2) return false;
    const LevelGeom& top = g.lv[1];
    const LevelGeom& l0 = g.lv[0];
    if (2 * top.T > kTileStep * l0.T || top.T < l0.T) return false;      // 2 <= step <= kTileStep: 5 columns span <= 16 cells, a second row <= 4
    const size_t npos = (size_t)top.Wd * top.Hd;
    return npos >= 1 && coarse_plan_lds_bytes(top.Wd, top.Hd) <= 60 * 1024;   // at least one template per workgroup
}

size_t coarse_plan_lds_bytes(int Wd, int Hd) {                 // per template of a workgroup
    const size_t npos = (size_t)Wd * Hd, nw = (npos + 31) / 32;
    return (npos + nw + 4 * (size_t)kMaxTilesPerTemplate + kPlanHits) * sizeof(uint32_t);   // scores | hit bitmap | tile records | hit list
}

void launch_coarse(const FrameBatch& fb, const FrameGeom& g, const TemplEntry* entries, const int32_t* feat_off,
                   const int32_t* work_pyramids, int num_work, float threshold, uint32_t cap, uint32_t tile_cap, hipStream_t s) {
    if (num_work <= 0 || fb.nb <= 0) return;
    const TileRec* tiles = fb.f[0].tiles;                      // tiles are planned for all frames of a batch or for none
    const uint8_t* todo = fb.f[0].todo;
    const int level = g.levels - 1;
    const LevelGeom lv = g.lv[level];
    int npos = lv.Wd * lv.Hd;
    const int nchunks = (npos + 15) / 16;
    int waves = (nchunks + kChunksPerWave - 1) / kChunksPerWave;
    if (waves > 16) waves = 16;
    if (waves < 1) waves = 1;
    const int cpw = std::min(kChunksPerWave, std::max(1, (nchunks + waves - 1) / waves));   // the map's chunks evenly over the waves
    TilePlanGeom plan{};
    size_t lds = 0;
    int group = 1;
    // Tiles are planned for k_local's tile path (tiles != null); with only `todo` given (the bit-plane refinement overwrites it) the hits are
    // still collected per workgroup — one reservation atomic instead of one per wave — but nothing is grouped: no tile half of counters[0].
    const bool plan_tiles = tiles && todo && tile_cap > 0 && tile_plan_possible(g);
    const bool group_only = !tiles && todo && npos >= 1 && coarse_plan_lds_bytes(lv.Wd, lv.Hd) <= 60 * 1024;
    if (plan_tiles || group_only) {
        plan.enabled = 1;
        plan.tiles = plan_tiles ? 1 : 0;
        plan.W0 = g.lv[0].W; plan.H0 = g.lv[0].H; plan.T0 = g.lv[0].T; plan.Wd0 = g.lv[0].Wd; plan.Hd0 = g.lv[0].Hd;
        plan.tile_cap = tile_cap;
        const int max_group = knobs().coarse_group;
#ifdef LM_DIAG
        plan.dbg = knobs().coarse_dbg;
#endif
        // Templates per workgroup: up to 4 (<= 16 waves, <= 64 KB of LDS).  More templates mean fewer atomics on the frame's counter,
        // but workgroups of 7-8 templates (14-16 waves) leave room for one workgroup per CU only: 16k templates at VGA 162 us with
        // 2-6 per workgroup, 300 us with 7-8; 2k templates 45 us with 2-8, 57 us with 1 (profiles/r02_coarse_group_sweep.txt).
        const size_t per = coarse_plan_lds_bytes(lv.Wd, lv.Hd);
        group = std::max(1, std::min(std::min(4, std::max(1, max_group)), std::min(16 / waves, (int)((64 * 1024 - 256) / per))));
        lds = per * group + (4 * (size_t)group + 4) * sizeof(uint32_t);
    }
    hipLaunchKernelGGL(k_coarse, dim3((num_work + group - 1) / group, fb.nb), dim3(group * waves * 64), lds, s, fb, lv, level, g.levels, entries,
                       feat_off, work_pyramids, num_work, waves, cpw, threshold, cap, plan);
}

// ---------------------------------------------------------------------------------------------
// Local refinement (LL.cpp:1855-1938, similarityLocal :1366-1428): one wave64 per candidate, marching
// up the pyramid; the grid is persistent and strides over the candidate list whose length is read
// from device memory (no host round trip between the coarse and the local pass).
//
// Fast path (every feature's 16x16 window inside its plane — always, except for oversized templates):
// gathers from the strip-major copy of the linear memories ([strip of 16 columns][row][16 B]).  A
// window spans two strips, so 32 lanes x one aligned 16-byte load cover it: lane (strip h, row r), rows fastest.
// The two half-waves work on two features of the same alignment class at a time; class runs are
// accumulated in packed u8 and realigned once per run: half-waves combined (lane ^ 32), strip pairs
// exchanged (lane ^ 16), v_alignbyte by the run's byte phase, widened into the u16 window
// accumulators (lane (r,h) keeps window columns 8h..8h+7 of row r).
// Slow path: flat layout, per-feature bounds test, exactly the reference's reads (wrap-around included).
// First strict maximum (LL.cpp:1920) = wave max-reduction of the packed key (raw << 8 | 255 - index).
// ---------------------------------------------------------------------------------------------

__global__ void __launch_bounds__(256)
k_local(FrameBatch fb, FrameGeom g, const TemplEntry* __restrict__ entries, const int32_t* __restrict__ feat_off,
        const uint32_t* __restrict__ feat_word, const uint32_t* __restrict__ run_mask, const uint32_t* __restrict__ feat_xy,
        const int32_t* __restrict__ work_pyramids, uint32_t cand_cap, float threshold, uint32_t cap,
        uint32_t dedupe_cap_slots, uint32_t tile_cap, int dbg) {
    // What bounds it (profiles/r02_pmc_tiles.txt, r02_local_experiments.txt): the vector L1.  Per CU 139k cycles of accesses +
    // 56k cycles stalled on pending misses of the 247k the kernel lasts (round 1: 345k + 37k of 363k) — every 128-byte line is
    // used by exactly one load instruction, so a third of the accesses miss.  Tried and measured WORSE: four waves sharing an
    // item through LDS (119 us: 104 VGPRs), 16 loads in flight per wave (123 us: 99 VGPRs) — occupancy matters more than the
    // length of a wave's chain of load batches —, tiles and singles on different workgroups (117 us: any alternation in the
    // workgroup index aliases with the round-robin over XCDs / shader engines) or on different waves of a workgroup (121 us).
    // Tiles alone take 72 us, singles alone 58 us, an empty launch of this grid 11 us (make DIAG=1, LM_LOCAL_DBG = 1 / 2 / 3).
    //
    // Work items of a BATCH of frames: the tiles of frame 0, 1, ... then the candidates of frame 0, 1, ...; s_tend / s_cend hold
    // the running ends, so a flat item index resolves to (frame, index) with at most nb comparisons (wave-uniform).
    __shared__ unsigned long long s_acc[kMaxBatch][2];            // per frame: 16x16 evaluations of this block, their algorithmic bytes
    __shared__ uint32_t s_tend[kMaxBatch + 1], s_cend[kMaxBatch + 1];
    const int lane = threadIdx.x & 63;
    const uint32_t wave0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)(blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)));   // wave-uniform, and the
    const uint32_t nwaves = gridDim.x * (blockDim.x >> 6);                                       // compiler must know it: item -> frame -> the frame's pointers stay in SGPRs
    const int nb = fb.nb;
    // Frame -> XCD affinity.  Workgroup b runs on XCD b % 8 (observed on gfx950, not a contract: only speed depends on it), each XCD has
    // its own 4 MB L2, and the strip planes of ONE frame are 4.9 MB at VGA.  With the items of a batch dealt to the waves in flat order
    // every L2 holds lines of all the batch's frames: L2 hit rate 94 % -> 76 %, fabric traffic per frame x4 (profiles/r03_pmc_batch_flat.txt).
    // So when the batch size divides 8, frame f is served by the workgroups of XCDs [f * 8 / nb, (f + 1) * 8 / nb) only.
    int f_lo = 0, f_hi = nb;
    uint32_t w_first = wave0, w_step = nwaves;
    if (nb > 1 && (8 % nb) == 0 && (gridDim.x & 7) == 0) {
        const int per = 8 / nb, xcd = (int)(blockIdx.x & 7);
        f_lo = xcd / per; f_hi = f_lo + 1;
        const uint32_t wpb = blockDim.x >> 6;
        w_first = ((blockIdx.x >> 3) * (uint32_t)per + (uint32_t)(xcd % per)) * wpb + (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
        w_step = (gridDim.x >> 3) * (uint32_t)per * wpb;
    }
    const bool tiled = fb.f[0].tiles != nullptr && !(dbg & 2);
    if (threadIdx.x == 0) {
        uint32_t te = 0, ce = 0;
        s_tend[0] = 0; s_cend[0] = 0;
        for (int f = 0; f < nb; ++f) {
            const unsigned long long packed = fb.f[f].counters[0];
            const unsigned long long nc = packed & kCandMask, nt = packed >> kCandBits;
            ce += nc < cand_cap ? (uint32_t)nc : cand_cap;
            te += !tiled ? 0u : (nt < tile_cap ? (uint32_t)nt : tile_cap);
            s_tend[f + 1] = te; s_cend[f + 1] = ce;
            s_acc[f][0] = 0; s_acc[f][1] = 0;
        }
    }
    __syncthreads();
    // the part of k_dedupe's hash table every frame will use, emptied here (k_dedupe runs next on the stream): no memset node
    for (int f = 0; f < nb; ++f) {
        unsigned long long* table = fb.f[f].dedupe_table;
        if (!table) continue;
        const uint32_t tsize = dedupe_slots_for(s_cend[f + 1] - s_cend[f], dedupe_cap_slots);
        for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < tsize; i += gridDim.x * blockDim.x) table[i] = ~0ull;
    }

    // ---- Tile refinement: the candidates of one template whose coarse cells are neighbours have level-0 windows kTileStep
    // cells apart, so most of every feature's window is shared.  A tile (planned by k_coarse) covers up to kTileNx x kTileNy
    // such candidates: R = 16 + kTileStep * (rows of candidates - 1) rows x S strips of the strip-major planes, lane = (strip q,
    // row r), ONE aligned 16-byte load per lane and feature.  Class runs are summed in packed bytes and realigned once per run
    // like everywhere else ({own strip, next strip's lane r} shifted by the run's byte phase) into u16 sums of tile columns
    // [16 q, 16 q + 16); every member then takes the first strict maximum of its own 16 x 16 window of those sums — the same
    // integers, the same tie-break (packed key) and the same float expression as the per-candidate path below.
    if (tiled) {
        const uint32_t ntiles = (uint32_t)__builtin_amdgcn_readfirstlane((int)s_tend[f_hi]);
        const LevelGeom lv = g.lv[0];
        const int T = lv.T, Hd = lv.Hd, offset = T / 2 + (T % 2 - 1);
        int fr = f_lo;
        for (uint32_t gi = (uint32_t)__builtin_amdgcn_readfirstlane((int)s_tend[f_lo]) + w_first; gi < ntiles; gi += w_step) {
            while (gi >= (uint32_t)__builtin_amdgcn_readfirstlane((int)s_tend[fr + 1])) ++fr;           // wave-uniform: item -> frame
            const FrameSlot& F = fb.f[fr];
            const uint8_t* __restrict__ sm_arena = F.sm_arena;
            Candidate* __restrict__ matches_dev = F.matches_dev;
            const TileRec t = F.tiles[gi - (uint32_t)__builtin_amdgcn_readfirstlane((int)s_tend[fr])];
            const int work = __builtin_amdgcn_readfirstlane(t.work);
            const uint32_t gxy = (uint32_t)__builtin_amdgcn_readfirstlane((int)t.gxy);
            const uint32_t mask_word = (uint32_t)__builtin_amdgcn_readfirstlane((int)t.mask);
            const uint32_t mask = mask_word & ((1u << kTileMaskBits) - 1u), steps = mask_word >> kTileMaskBits;   // members | window steps (3 bits each)
            const int dy1 = (int)((steps >> (3 * (kTileNx - 1))) & 7u);        // window origin of the second row of members
            const uint32_t slot_base = (uint32_t)__builtin_amdgcn_readfirstlane((int)t.slot_base);
            const int gx0 = (int16_t)(gxy & 0xFFFF), gy0 = (int16_t)(gxy >> 16);
            const int pyr = __builtin_amdgcn_readfirstlane(work_pyramids[work]);
            const TemplEntry e = entries[(size_t)pyr * g.levels];
            const int nf = e.nf, nfp = e.nf_padded;
            const uint32_t cols = (mask | (mask >> kTileNx)) & ((1u << kTileNx) - 1u);
            const int S = cols > 1u ? 3 : 2;                                   // a second column of candidates needs a third strip
            const int R = (mask >> kTileNx) ? 16 + dy1 : 16;
            int q = (lane >= R) + (lane >= 2 * R), r = lane - q * R;
            const bool active = lane < R * S;
            if (!active) { q = 0; r = 0; }                                     // idle lanes repeat lane 0's (valid) loads
            // Addressing (lm_kernels.h, feat_word): a feature's load = arena resource + lane VGPR + the feature's word (less its class
            // bits) as scalar offset.  The VGPR carries the lane's (strip, row) offset, the item's window origin K and the strip
            // carry of the run's class, rebuilt once per class run: ONE scalar instruction per feature.  (Every offset stays a
            // multiple of 16: folding the class bits into the VGPR as well made the loads return wrong data.)
            const BufRsrc strips = make_rsrc(sm_arena);
            const uint32_t lane_off = (uint32_t)(q * Hd + r) * 16u;
            const uint32_t K0 = (uint32_t)((gx0 >> 4) * Hd + gy0) * 16u, HS = (uint32_t)Hd * 16u;
            const int gxl = gx0 & 15;
            const int nb_lane = (lane + R < 64 ? lane + R : lane) << 2;       // the lane holding the next strip of this row
            uint32_t aE[4] = {0, 0, 0, 0}, aO[4] = {0, 0, 0, 0};              // u16x2 sums: aE[k] = cols 4k, 4k+2; aO[k] = cols 4k+1, 4k+3
            const uint32_t* fw = feat_word + e.feat_start;
            const uint32_t* rm = run_mask + (e.feat_start >> 3);
            uint32_t r8[4] = {0, 0, 0, 0};
            int cur = -1;
            auto flush = [&](int cls) {
                uint32_t x[8];
#pragma unroll
                for (int k = 0; k < 4; ++k) { x[k] = r8[k]; x[4 + k] = (uint32_t)__builtin_amdgcn_ds_bpermute(nb_lane, (int)r8[k]); r8[k] = 0; }
                const int c0 = (cls + gx0) & 15;                               // byte phase of the tile's first column in its strip
                const uint32_t sb = (uint32_t)(c0 & 3);
                uint32_t o4[4];
                switch (c0 >> 2) {                                             // wave-uniform
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
                for (int k = 0; k < 4; ++k) add_bytes(o4[k], aE[k], aO[k]);
            };
            if (nfp > 0) {
                uint32_t c[kFeatBatch], m = rm[0];
#pragma unroll
                for (int u = 0; u < kFeatBatch; ++u) c[u] = fw[u];                    // wave-uniform -> SMEM
                uint32_t vK = lane_off;                                                 // rebuilt at the first feature (always a run start)
                for (int f = 0; f < nfp; f += kFeatBatch) {
                    const int fn = f + kFeatBatch < nfp ? f + kFeatBatch : f;           // prefetch the next batch
                    uint32_t cn[kFeatBatch];
                    const uint32_t mn = rm[fn >> 3];
#pragma unroll
                    for (int u = 0; u < kFeatBatch; ++u) cn[u] = fw[fn + u];
                    uint4 v[kFeatBatch];
#pragma unroll
                    for (int u = 0; u < kFeatBatch; ++u) {
                        if (m & (1u << u)) {                                            // a class run starts: its lane offset
                            const int cls = (int)(c[u] & 15);
                            vK = lane_off + K0 + (cls + gxl >= 16 ? HS : 0u);
                        }
                        v[u] = ld_buf16(strips, vK, c[u] & ~15u);
                    }
#pragma unroll
                    for (int u = 0; u < kFeatBatch; ++u) {
                        if (m & (1u << u)) {
                            if (cur >= 0) flush(cur);
                            cur = (int)(c[u] & 15);
                        }
                        r8[0] += v[u].x; r8[1] += v[u].y; r8[2] += v[u].z; r8[3] += v[u].w;   // <= 62 x 4 per byte (host: runs of <= kRunMax)
                    }
#pragma unroll
                    for (int u = 0; u < kFeatBatch; ++u) c[u] = cn[u];
                    m = mn;
                }
                if (cur >= 0) flush(cur);
            }
            // members, in slot order
            uint32_t rest = mask;
            int member = 0;
            if (lane == 0) {
                const unsigned long long nmem = (unsigned long long)__popc(mask);
                atomicAdd(&s_acc[fr][0], nmem);
                atomicAdd(&s_acc[fr][1], nmem * 256ull * (unsigned long long)nf);          // algorithmic response bytes: 256 per feature and member
            }
            while (rest) {
                const int bit = __ffs((int)rest) - 1;
                rest &= rest - 1;
                const int j = bit / kTileNx, i = bit - j * kTileNx;
                int dx = 0;                                                     // window origin of the member's column (wave-uniform)
                for (int u = 0; u < i; ++u) dx += (int)((steps >> (3 * u)) & 7u);
                const int dy = j ? dy1 : 0;
                const int rr = r - dy;                                          // row of this lane inside the member's window
                uint32_t key = 0;
                if (active && q < 2 && rr >= 0 && rr < 16 && (dx & 3)) {        // a window that starts inside a 4-column group: column by column
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const uint32_t raws4[4] = {aE[k] & 0xFFFF, aO[k] & 0xFFFF, aE[k] >> 16, aO[k] >> 16};
#pragma unroll
                        for (int tcol = 0; tcol < 4; ++tcol) {
                            const int col = 16 * q + 4 * k + tcol - dx;
                            if (col >= 0 && col < 16) {
                                const uint32_t kk = (raws4[tcol] << 8) | (255u - (uint32_t)(rr * 16 + col));
                                key = kk > key ? kk : key;
                            }
                        }
                    }
                } else if (active && q < 2 && rr >= 0 && rr < 16) {
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const int col = 16 * q + 4 * k - dx;                    // window column of the lane's 4-column group k
                        if (col >= 0 && col < 16) {
                            const uint32_t idx = (uint32_t)(rr * 16 + col);
                            const uint32_t k0 = ((aE[k] & 0xFFFF) << 8) | (255u - idx), k1 = ((aO[k] & 0xFFFF) << 8) | (255u - (idx + 1));
                            const uint32_t k2 = ((aE[k] >> 16) << 8) | (255u - (idx + 2)), k3 = ((aO[k] >> 16) << 8) | (255u - (idx + 3));
                            const uint32_t ka = k0 > k1 ? k0 : k1, kb = k2 > k3 ? k2 : k3;
                            const uint32_t kk = ka > kb ? ka : kb;
                            key = kk > key ? kk : key;
                        }
                    }
                }
                const uint32_t k = wave_max_u32(key);
                const int raw = (int)(k >> 8);
                int br = -1, bc = -1;                                           // LL.cpp:1910-1911
                float best = 0.f;
                if (raw > 0) {
                    const int idx = 255 - (int)(k & 0xFF);
                    br = idx >> 4; bc = idx & 15;
                    best = score_of(raw, nf);
                }
                const uint32_t slot = slot_base + (uint32_t)member;
                ++member;
                if (lane == 0 && slot < cap) {
                    Candidate m;
                    m.x = (gx0 + dx + bc) * T + offset;                         // LL.cpp:1930-1931: x / T - 8 = the member's window origin
                    m.y = (gy0 + dy + br) * T + offset;
                    m.score = best;
                    m.work = best < threshold ? -1 : work;                      // LL.cpp:1935
                    matches_dev[slot] = m;
                }
            }
        }
    }

    const uint32_t num_cands = (uint32_t)__builtin_amdgcn_readfirstlane((int)s_cend[f_hi]);
    int fr = f_lo;
    for (uint32_t gi = (uint32_t)__builtin_amdgcn_readfirstlane((int)s_cend[f_lo]) + w_first; gi < num_cands && !(dbg & 1); gi += w_step) {
        while (gi >= (uint32_t)__builtin_amdgcn_readfirstlane((int)s_cend[fr + 1])) ++fr;               // wave-uniform: item -> frame
        const FrameSlot& F = fb.f[fr];
        const uint32_t ci = gi - (uint32_t)__builtin_amdgcn_readfirstlane((int)s_cend[fr]);
        if (tiled && !F.todo[ci]) continue;                  // a tile member: refined above (wave-uniform)
        const uint8_t* __restrict__ lm_arena = F.lm_arena;
        const uint8_t* __restrict__ sm_arena = F.sm_arena;
        Candidate* __restrict__ matches_dev = F.matches_dev;
        const Candidate cd = F.cands[ci];
        unsigned long long evals = 0, bytes = 0;
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
                // lane = (feature of the pair, strip h, row r) with the ROWS fastest: a quad of lanes reads 4 consecutive 16-byte rows of one strip
                // (one or two 64-byte lines) instead of 2 rows x 2 strips (profiles/r03_tcp_rotation_microbench.txt: 41 -> 36.6 cycles per wave load)
                const int half = lane >> 5, l5 = lane & 31, r = l5 & 15, h = l5 >> 4;
                const uint32_t HS = (uint32_t)Hd * 16u;
                const BufRsrc strips = make_rsrc(sm_arena);        // addressing as in the tile path: lane VGPR (rebuilt per class run) + feature word
                const uint32_t lane_off = h * HS + (uint32_t)r * 16u;
                const uint32_t K0 = (uint32_t)((gx >> 4) * Hd + gy) * 16u;
                const int gxl = gx & 15;
                uint32_t w[4] = {0, 0, 0, 0};                   // u16x2: cols (0,2) (1,3) (4,6) (5,7) of this lane's 8 columns
                if (nfp > 0) {
                    const uint32_t* fw = feat_word + e.feat_start;
                    const uint32_t* rm = run_mask + (e.feat_start >> 3);
                    uint32_t r8[4] = {0, 0, 0, 0};
                    int cur = -1;
                    auto flush = [&](int cls) {
                        uint32_t own[4], par[4];
#pragma unroll
                        for (int k = 0; k < 4; ++k) {
                            own[k] = r8[k] + (uint32_t)__shfl_xor((int)r8[k], 32, 64);   // both features of the pair
                            par[k] = (uint32_t)__shfl_xor((int)own[k], 16, 64);           // the other strip of this row
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
                    uint32_t c[kFeatBatch], m = rm[0];
#pragma unroll
                    for (int u = 0; u < kFeatBatch; ++u) c[u] = fw[u];                    // wave-uniform -> SMEM
                    uint32_t vK = lane_off;
                    for (int f = 0; f < nfp; f += kFeatBatch) {
                        const int fn = f + kFeatBatch < nfp ? f + kFeatBatch : f;           // prefetch the next batch
                        uint32_t cn[kFeatBatch];
                        const uint32_t mn = rm[fn >> 3];
#pragma unroll
                        for (int u = 0; u < kFeatBatch; ++u) cn[u] = fw[fn + u];
                        uint4 v[kFeatBatch / 2];
#pragma unroll
                        for (int k = 0; k < kFeatBatch / 2; ++k) {
                            // features 2k (lanes 0-31) and 2k+1 (lanes 32-63): one class run by construction (runs start at even indices)
                            if (m & (1u << (2 * k))) {
                                const int cls = (int)(c[2 * k] & 15);
                                vK = lane_off + K0 + (cls + gxl >= 16 ? HS : 0u);
                            }
                            v[k] = ld_buf16(strips, vK + ((half ? c[2 * k + 1] : c[2 * k]) & ~15u), 0u);   // (a lane offset that wraps below zero against the scalar offset reads zeros)
                        }
#pragma unroll
                        for (int k = 0; k < kFeatBatch / 2; ++k) {
                            if (m & (1u << (2 * k))) {
                                if (cur >= 0) flush(cur);
                                cur = (int)(c[2 * k] & 15);
                            }
                            r8[0] += v[k].x; r8[1] += v[k].y; r8[2] += v[k].z; r8[3] += v[k].w;   // <= 31 x 4 per byte and half (host: runs of <= kRunMax)
                        }
#pragma unroll
                        for (int u = 0; u < kFeatBatch; ++u) c[u] = cn[u];
                        m = mn;
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
        // threshold on the way up, LL.cpp:1935).
        if (lane == 0 && ci < cap) {
            Candidate m;
            m.x = mx; m.y = my; m.score = sim; m.work = alive ? work : -1;
            matches_dev[ci] = m;                 // in HBM: k_dedupe, the on-device NMS / top-K (pipeline.cpp) and the exchange read it there; the host
                                                 // fetches the raw records only when asked for them (lm_collect_frame, sort_unique = 0)
        }
        if (lane == 0) { atomicAdd(&s_acc[fr][0], evals); atomicAdd(&s_acc[fr][1], bytes); }
    }
    // statistics (16x16 evaluations, their algorithmic bytes) per frame: summed per block in LDS, then one pair of atomics per
    // block and frame it worked on into the frame's sharded counters (k_dedupe's last block publishes the totals)
    __syncthreads();
    if ((int)threadIdx.x < nb && s_acc[threadIdx.x][0] != 0) {
        unsigned long long* st = fb.f[threadIdx.x].counters + 8 + 2 * (blockIdx.x & (kStatShards - 1));
        atomicAdd(st, s_acc[threadIdx.x][0]);
        atomicAdd(st + 1, s_acc[threadIdx.x][1]);
    }
}

// ---- Bit planes: the default encoding of the response memories for both matching kernels --------------------------
// A response is 0, 1 or 4, so a position's sum is n1 + 4 n4 with n1 / n4 = the number of features whose response there is 1 / 4: two bits
// per cell carry what the byte planes do, and the sums can be kept BIT-SLICED (one dword = 32 positions of one counter bit), features
// entering through carry-save adders; integers are formed once per candidate / only for the hits.
//
// Levels below the top ("strip records", read by k_local_bits): the strip arena at half the offsets — per plane row and 16-column strip s an
// 8-byte record holding cells [16 s, 16 s + 32) of that row with TWO BITS PER CELL, bit 2c = "the response of cell 16 s + c is 1", bit
// 2c + 1 = "it is 4".  Strips are 32 cells wide at a stride of 16, so any 16-cell window of a row lies inside ONE record and comes out of it
// with ONE funnel shift, v_alignbit(hi, lo, 2 (column & 15)): a dword of 16 positions x {is-1, is-4}.  (Round 3 kept the two planes in
// separate dwords: 4 shifts + 2 ands + 2 or per lane and feature to cut the windows out, and every lane redid the address arithmetic of
// every feature — 30 VALU instructions per lane and feature at 0.92 of the issue slots)
// The top level ("pair stream", read by k_coarse_bits): the flat linear memories [label][phase][position] as one pair {is-1 dword, is-4
// dword} per 32 consecutive arena bytes, so the reference's reads that run from one plane into the next stay what they are.
// carry-save adder: two v_bitop3_b32 (gfx950's ternary logic op; truth tables with a = 0xF0, b = 0xCC, c = 0xAA: parity 0x96, majority 0xE8)
static __device__ __forceinline__ void csa(uint32_t& sum, uint32_t& carry, uint32_t a, uint32_t b, uint32_t c) {
    const uint32_t s = __builtin_amdgcn_bitop3_b32(a, b, c, 0x96);
    carry = __builtin_amdgcn_bitop3_b32(a, b, c, 0xE8);
    sum = s;
}
// eight 1-bit inputs into {ones, twos, fours}; returns the carry of weight 8 (7 carry-save adders = 14 v_bitop3)
static __device__ __forceinline__ uint32_t add8(const uint32_t (&x)[8], uint32_t& ones, uint32_t& twos, uint32_t& fours) {
    uint32_t ta, tb, fa, fb, e;
    csa(ones, ta, ones, x[0], x[1]);
    csa(ones, tb, ones, x[2], x[3]);
    csa(twos, fa, twos, ta, tb);
    csa(ones, ta, ones, x[4], x[5]);
    csa(ones, tb, ones, x[6], x[7]);
    csa(twos, fb, twos, ta, tb);
    csa(fours, e, fours, fa, fb);
    return e;
}
// Counter of kN bits per position: c[0..3] = ones, twos, fours, eights, c[4..] = 16s and up.  Two carries of weight 8 (16 features) enter
// through one more adder and ONE ripple of the sixteens (Harley-Seal: 2.5 operations per input instead of 3.25 with a ripple per 8).
template <int kN>
static __device__ __forceinline__ void add_eights2(uint32_t (&c)[kN], uint32_t e1, uint32_t e2) {
    uint32_t k16;
    csa(c[3], k16, c[3], e1, e2);
#pragma unroll
    for (int k = 4; k < kN; ++k) { const uint32_t t = c[k] & k16; c[k] ^= k16; k16 = t; }
}
template <int kN>
static __device__ __forceinline__ void add_eights1(uint32_t (&c)[kN], uint32_t e1) {
    uint32_t k16 = c[3] & e1;
    c[3] ^= e1;
#pragma unroll
    for (int k = 4; k < kN; ++k) { const uint32_t t = c[k] & k16; c[k] ^= k16; k16 = t; }
}

// 4 response bytes -> 8 bits, cell k at bits 2k (is 1 = bit 0 of the byte) and 2k + 1 (is 4 = bit 2 of the byte); the multiply gathers the
// four 2-bit fields into the top byte (partial products land on distinct bits: no carries)
static __device__ __forceinline__ uint32_t pack4(uint32_t d) {
    const uint32_t e = d & 0x05050505u;
    const uint32_t t = (e | (e >> 1)) & 0x03030303u;
    return (t * 0x01041040u) >> 24;
}
static __device__ __forceinline__ uint32_t pack16(const uint4& v) { return pack4(v.x) | (pack4(v.y) << 8) | (pack4(v.z) << 16) | (pack4(v.w) << 24); }

// Strip records from the strip bytes (the front end writes the records itself when nothing reads the bytes: frontend.hip, bits_rows_body;
// this kernel serves the banks and geometries that keep the byte planes — a fallback launch of k_local may follow — and the tests of both).
__global__ void __launch_bounds__(256)
k_pack_bits(BitsBatch B, uint32_t sm_off0, uint32_t records, int NS, int Hd) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;             // record = ((plane * NS + s) * Hd + row)
    if (i >= records) return;
    const uint32_t q = i / (uint32_t)Hd, s = q % (uint32_t)NS;
    const uint8_t* src = B.strips[blockIdx.y] + sm_off0 + (size_t)i * 16;
    const uint4 a = *reinterpret_cast<const uint4*>(src);
    uint4 b = make_uint4(0, 0, 0, 0);
    if (s + 1 < (uint32_t)NS) b = *reinterpret_cast<const uint4*>(src + (size_t)Hd * 16);      // the next strip of this row
    uint2 r;
    r.x = pack16(a);
    r.y = pack16(b);
    *reinterpret_cast<uint2*>(B.bits[blockIdx.y] + (sm_off0 >> 1) + (size_t)i * 8) = r;
}

// Refinement on strip records (LL.cpp:1855-1938, similarityLocal :1366-1428).  8 lanes per candidate — lane j owns window rows 2j and 2j + 1,
// ONE 16-byte load = their two records —, 8 candidates per wave, one feature per candidate and load instruction: a wave load serves 8
// (candidate, feature) pairs.  Per 16 features a lane fetches two feature words and forms, ONCE for its group, their record offsets (window
// origin, strip carry) and shift amounts; the group's lanes pick them up with two ds_bpermute per feature and add only their own row term.
// Per lane and feature: 1 add, 2 v_alignbit, 5 adder operations (two dwords x 2.5).  Counters of kN = 4 + kHi bits: kHi = 5 serves entries
// of up to 511 features, kHi = 10 up to 16383 (the reference allows 8191 per modality, LL.cpp:1291, 1816).
// What bounds it (round 5, profiles/r05_local_sharing/README.txt): the number of wave loads — one takes 20-23 CU cycles whatever the lanes or addresses —, and
// what keeps its L2 traffic down is that the 8 candidates of a wave are CONSECUTIVE candidates of one template (x-neighbours: ~5 distinct lines per wave load
// instead of 23).  Three schemes that share loads between neighbouring candidates by regrouping them (vertical runs, interleaved pairs, same-lane pairs) were
// built, bit-exact, and ran no faster: fewer wave loads, more L2 requests.  Hence: candidates in slot order, a grid of the resident workgroups.
// All pyramid levels below the top are walked here (LL.cpp:1855: level by level, dropping a candidate as soon as it falls below the
// threshold); a candidate whose windows leave their planes at some level (oversized template, features outside the frame) is marked in
// `todo` and left, from the top, to k_local's per-candidate path.
template <int kHi, int kWaves>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(kWaves, kWaves)))
k_local_bits(FrameBatch fb, BitsBatch B, FrameGeom g, const TemplEntry* __restrict__ entries, const uint32_t* __restrict__ feat_word,
             const int32_t* __restrict__ work_pyramids, uint32_t cand_cap, float threshold, uint32_t cap, uint32_t dedupe_cap_slots) {
    constexpr int kN = 4 + kHi;                                     // counter bits per plane
    constexpr int kS = kN + 3;                                      // bits of n1 + 4 n4
    __shared__ unsigned long long s_acc[kMaxBatch][2];
    __shared__ uint32_t s_cnt[kMaxBatch];
    const int lane = threadIdx.x & 63, grp = lane >> 3, j = lane & 7;
    const int nb = fb.nb;
    // Frame -> XCD affinity as in k_local (an XCD's L2 holds ONE frame's planes), for ANY batch size up to 8: frame f is served by the workgroups of
    // the XCDs x with x % nb == f — 8 / nb of them each when nb divides 8, else some frames get one XCD more than others.  (Round 3 fell back to
    // dealing the items of all frames to all workgroups unless nb divided 8: a 5-frame launch — the first and last launches of a short stream —
    // then cost 35 us per frame against 26 in an 8-frame launch.)
    int f_lo = 0, f_hi = nb;
    uint32_t w_first = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6), w_step = gridDim.x * (blockDim.x >> 6);
    if (nb > 1 && nb <= 8 && (gridDim.x & 7) == 0) {
        const int xcd = (int)(blockIdx.x & 7);
        f_lo = xcd % nb; f_hi = f_lo + 1;
        const uint32_t mine = (uint32_t)((7 - f_lo) / nb + 1);      // XCDs serving this frame: f_lo, f_lo + nb, ...
        const uint32_t wpb = blockDim.x >> 6;
        w_first = ((blockIdx.x >> 3) * mine + (uint32_t)(xcd / nb)) * wpb + (threadIdx.x >> 6);
        w_step = (gridDim.x >> 3) * mine * wpb;
    }
    if ((int)threadIdx.x < nb) {
        const unsigned long long nc = fb.f[threadIdx.x].counters[0] & kCandMask;
        s_cnt[threadIdx.x] = nc < cand_cap ? (uint32_t)nc : cand_cap;
        s_acc[threadIdx.x][0] = 0; s_acc[threadIdx.x][1] = 0;
    }
    __syncthreads();
    for (int f = 0; f < nb; ++f) {                                  // k_dedupe's hash table, emptied here like k_local does
        unsigned long long* table = fb.f[f].dedupe_table;
        if (!table) continue;
        const uint32_t tsize = dedupe_slots_for(s_cnt[f], dedupe_cap_slots);
        for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < tsize; i += gridDim.x * blockDim.x) table[i] = ~0ull;
    }
    // The pair stream k_coarse_bits has just read is zeroed for the slot's next frame: the front end ORs it together (frontend.hip, top_bits_body)
    if (B.top_clear_units)
        for (int f = 0; f < nb; ++f)
            for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < B.top_clear_units; i += gridDim.x * blockDim.x)
                reinterpret_cast<uint4*>(B.top_clear[f])[i] = make_uint4(0u, 0u, 0u, 0u);
    const int src0 = (lane & ~7) << 2;                              // ds_bpermute address of the group's first lane
    const uint32_t rowoff = 16u * (uint32_t)j;                      // records of rows 2j, 2j + 1 behind the window's first row
    for (int fr = f_lo; fr < f_hi; ++fr) {
        const FrameSlot& F = fb.f[fr];
        const BufRsrc bits = make_rsrc(B.bits[fr]);
        const uint32_t nc = s_cnt[fr], ngroups = (nc + 7u) >> 3;
        for (uint32_t gi = w_first; gi < ngroups; gi += w_step) {
            const uint32_t ci = gi * 8u + (uint32_t)grp;
            const bool valid = ci < nc;
            Candidate cd{0, 0, 0.f, 0};
            if (valid) cd = F.cands[ci];
            const int work = cd.work;
            const int pyr = work_pyramids[work];
            int mx = cd.x, my = cd.y;
            float sim = cd.score;
            bool alive = valid, leave = false;                      // leave: k_local's per-candidate path takes this candidate (from the top)
            uint32_t evals = 0, bytes = 0;
            for (int l = g.levels - 2; l >= 0; --l) {
                const LevelGeom& lv = g.lv[l];
                const int T = lv.T, W = lv.W, H = lv.H, Wd = lv.Wd, Hd = lv.Hd;
                const int border = 8 * T, offset = T / 2 + (T % 2 - 1);
                const uint32_t HS8 = (uint32_t)Hd * 8u;
                const uint32_t zero_off = (lv.sm_off[1] + 8u * (uint32_t)(T * T) * ((uint32_t)lv.NS * (uint32_t)Hd * 16u)) >> 1;   // the level's all-zero plane
                const TemplEntry e = entries[(size_t)pyr * g.levels + l];
                // LL.cpp:1871-1880 (the clamp) and 1380-1381 (window origin), exactly as k_local
                const int max_x = W - e.width - border, max_y = H - e.height - border;
                int x = mx * 2 + 1, y = my * 2 + 1;
                x = x > border ? x : border;  y = y > border ? y : border;
                x = x < max_x ? x : max_x;    y = y < max_y ? y : max_y;
                const int gx = x / T - 8, gy = y / T - 8;
                const int off_x = gx * T, off_y = gy * T;
                const bool all_in = e.min_x >= 0 && e.min_y >= 0 && gx >= 0 && gy >= 0 &&
                                    ((e.max_x + off_x) / T + 16 <= Wd) && ((e.max_y + off_y) / T + 16 <= Hd);
                const bool want = alive && !leave;
                if (want && !all_in) leave = true;
                const bool run = want && all_in;
                const int nf = e.nf, nfp = run ? (int)e.nf_padded : 0;
                int nmax = nfp;
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) { const int t = __shfl_xor(nmax, o, 64); nmax = t > nmax ? t : nmax; }
                nmax = __builtin_amdgcn_readfirstlane(nmax);
                const uint32_t Kbase = (uint32_t)(((gx >> 4) * Hd + gy) * 8);
                const uint32_t gxl = (uint32_t)(gx & 15);
                const uint32_t* fw = feat_word + e.feat_start;
                uint32_t cA[kN], cB[kN];                            // bit-sliced counters of rows 2j / 2j + 1: even bits count the 1s, odd bits the 4s
#pragma unroll
                for (int k = 0; k < kN; ++k) { cA[k] = 0; cB[k] = 0; }
                // lane j fetches the words of features f0 + 2j, f0 + 2j + 1 (entries start at multiples of 8 words, f0 is a multiple of 16)
                auto fetch = [&](int f0) -> uint2 {
                    uint2 w = make_uint2(0u, 0u);
                    if (f0 + 2 * j < nfp) w = *reinterpret_cast<const uint2*>(fw + f0 + 2 * j);
                    return w;
                };
                // a feature word (base0 | column class, lm_kernels.h) -> offset of its first record in the bit arena, 2 x the window's cell inside it
                auto prep = [&](uint32_t w, bool on, uint32_t& off, uint32_t& s2) {
                    s2 = ((w & 15u) + gxl) << 1;
                    const uint32_t o = ((w & ~15u) >> 1) + Kbase + ((s2 >> 5) ? HS8 : 0u);
                    off = on ? o : zero_off;                        // beyond this candidate's features (or no candidate): the zero plane
                };
                uint32_t offx = 0, offy = 0, sx = 0, sy = 0;
                auto batch = [&](int half, uint32_t& eA, uint32_t& eB) {           // features 8 half .. 8 half + 7 of the current 16
                    uint4 v[8];
                    uint32_t sh[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        const int src = src0 + 4 * (4 * half + (u >> 1));          // the lane that fetched feature 8 half + u
                        const uint32_t o = (uint32_t)__builtin_amdgcn_ds_bpermute(src, (int)((u & 1) ? offy : offx));
                        sh[u] = (uint32_t)__builtin_amdgcn_ds_bpermute(src, (int)((u & 1) ? sy : sx));
                        v[u] = ld_buf16(bits, o + rowoff, 0u);      // rows 2j, 2j + 1: 32 cells x 2 bits each
                    }
                    uint32_t xa[8], xb[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        xa[u] = __builtin_amdgcn_alignbit(v[u].y, v[u].x, sh[u]);      // 16 window cells of row 2j: bits 2c (is 1), 2c + 1 (is 4)
                        xb[u] = __builtin_amdgcn_alignbit(v[u].w, v[u].z, sh[u]);      // ... of row 2j + 1
                    }
                    eA = add8(xa, cA[0], cA[1], cA[2]);
                    eB = add8(xb, cB[0], cB[1], cB[2]);
                };
                uint2 wn = fetch(0);
                for (int f0 = 0; f0 < nmax; f0 += 16) {
                    const uint2 w = wn;
                    const bool on = f0 + 2 * j < nfp;
                    wn = fetch(f0 + 16);                            // the next 16 words while these are worked on
                    prep(w.x, on, offx, sx);
                    prep(w.y, on, offy, sy);
                    uint32_t e1A, e1B;
                    batch(0, e1A, e1B);
                    if (f0 + 8 < nmax) {                            // wave-uniform
                        uint32_t e2A, e2B;
                        batch(1, e2A, e2B);
                        add_eights2<kN>(cA, e1A, e2A);
                        add_eights2<kN>(cB, e1B, e2B);
                    } else {
                        add_eights1<kN>(cA, e1A);
                        add_eights1<kN>(cB, e1B);
                    }
                }
                // Once per candidate and level: the two rows of the lane side by side — bit 2c = row 2j, bit 2c + 1 = row 2j + 1 of window
                // column c —, S = n1 + 4 n4 bit-sliced, the lane's maximum by a descent from the top bit, its FIRST position in raster order.
                uint32_t S[kS], carry = 0;
                {
                    uint32_t n1[kN], n4[kN];
#pragma unroll
                    for (int k = 0; k < kN; ++k) {
                        n1[k] = (cA[k] & 0x55555555u) | ((cB[k] << 1) & 0xAAAAAAAAu);
                        n4[k] = ((cA[k] >> 1) & 0x55555555u) | (cB[k] & 0xAAAAAAAAu);
                    }
                    S[0] = n1[0]; S[1] = n1[1];
#pragma unroll
                    for (int k = 2; k < kS; ++k) {
                        const uint32_t a = k < kN ? n1[k] : 0u, b = k - 2 < kN ? n4[k - 2] : 0u;
                        csa(S[k], carry, a, b, carry);
                    }
                }
                uint32_t mask = 0xFFFFFFFFu, val = 0;
#pragma unroll
                for (int k = kS - 1; k >= 0; --k) {
                    const uint32_t t = mask & S[k];
                    if (t) { mask = t; val |= 1u << k; }
                }
                const uint32_t upper = mask & 0x55555555u;           // positions of row 2j attaining the maximum come first in raster order
                const uint32_t pick = upper ? upper : mask;
                const uint32_t bitp = (uint32_t)__ffs((int)pick) - 1u;
                const uint32_t pos = ((2u * (uint32_t)j + (bitp & 1u)) << 4) + (bitp >> 1);   // row * 16 + column
                uint32_t key = (val << 8) | (255u - pos);
#pragma unroll
                for (int o = 1; o < 8; o <<= 1) { const uint32_t t = (uint32_t)__shfl_xor((int)key, o, 64); key = t > key ? t : key; }
                const int raw = (int)(key >> 8);
                int br = -1, bc = -1;                               // LL.cpp:1910-1911
                float best = 0.f;
                if (raw > 0) {
                    const int idx = 255 - (int)(key & 0xFF);
                    br = idx >> 4; bc = idx & 15;
                    best = score_of(raw, nf);
                }
                if (run) {
                    ++evals;
                    bytes += 256u * (uint32_t)nf;                   // algorithmic response bytes of this 16x16 evaluation
                    sim = best;
                    mx = (x / T - 8 + bc) * T + offset;             // LL.cpp:1930-1931
                    my = (y / T - 8 + br) * T + offset;
                    if (sim < threshold) alive = false;             // LL.cpp:1935
                }
            }
            if (valid && j == 0) {
                F.todo[ci] = leave ? 1 : 0;
                if (!leave) {
                    if (ci < cap) {
                        Candidate m;
                        m.x = mx; m.y = my; m.score = sim;
                        m.work = alive ? work : -1;
                        F.matches_dev[ci] = m;
                    }
                    atomicAdd(&s_acc[fr][0], (unsigned long long)evals);
                    atomicAdd(&s_acc[fr][1], (unsigned long long)bytes);
                }
            }
        }
    }
    __syncthreads();
    if ((int)threadIdx.x < nb && s_acc[threadIdx.x][0] != 0) {
        unsigned long long* st = fb.f[threadIdx.x].counters + 8 + 2 * (blockIdx.x & (kStatShards - 1));
        atomicAdd(st, s_acc[threadIdx.x][0]);
        atomicAdd(st + 1, s_acc[threadIdx.x][1]);
    }
}

void launch_pack_bits(const BitsBatch& B, int nb, const LevelGeom& lv, hipStream_t s) {
    const uint32_t records = (uint32_t)(2 * 8 * lv.T * lv.T) * (uint32_t)lv.NS * (uint32_t)lv.Hd;
    hipLaunchKernelGGL(k_pack_bits, dim3((records + 255) / 256, nb), dim3(256), 0, s, B, lv.sm_off[0], records, lv.NS, lv.Hd);
}
void launch_local_bits(const FrameBatch& fb, const BitsBatch& B, const FrameGeom& g, const TemplEntry* entries, const uint32_t* feat_word,
                       const int32_t* work_pyramids, uint32_t cand_cap, float threshold, uint32_t cap, uint32_t dedupe_cap_slots, int grid_blocks,
                       int max_features, hipStream_t s) {
    if (max_features > kBitsSmallMax)
        hipLaunchKernelGGL((k_local_bits<10, 4>), dim3(grid_blocks), dim3(256), 0, s, fb, B, g, entries, feat_word, work_pyramids, cand_cap, threshold, cap, dedupe_cap_slots);
    else
        hipLaunchKernelGGL((k_local_bits<5, 4>), dim3(grid_blocks), dim3(256), 0, s, fb, B, g, entries, feat_word, work_pyramids, cand_cap, threshold, cap, dedupe_cap_slots);
}

// ---- Coarse pass on the pair stream (LL.cpp:1284-1354 similarity + :1835-1852 scan).  A wave per template, a lane = 32 consecutive
// positions of the template's map: a feature's load = two consecutive pairs per lane (16 bytes) at pair index (offset >> 5) + lane,
// funnel-shifted by offset & 31 — the offset is wave-uniform, so both are scalar —, i.e. ONE wave load per feature and 2048 positions
// (the byte kernel: one per 1008 positions and 16 bytes per position and lane).  Sums bit-sliced as in k_local_bits; the threshold test is
// a bit-sliced comparison with the smallest raw sum that passes, so integers are formed only for the hits.  The workgroup's 4 templates
// reserve their candidate slots with ONE atomic on the frame's counter (same-address atomics serialise in the L2: 2000 per frame cost
// the byte kernel 16 us).  No tiles: the bit-plane refinement needs none.
__global__ void __launch_bounds__(256)
k_pack_top(TopBits B, uint32_t byte0, uint32_t npairs) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= npairs) return;
    const uint8_t* src = B.lm[blockIdx.y] + byte0 + (size_t)i * 32;
    const uint4 a = *reinterpret_cast<const uint4*>(src), b = *reinterpret_cast<const uint4*>(src + 16);
    auto four = [](uint32_t d, int sh) -> uint32_t { return ((((d >> sh) & 0x01010101u) * 0x01020408u) >> 24) & 0xFu; };   // bit sh of 4 bytes -> 4 bits
    auto bits = [&](const uint4& v, int sh) -> uint32_t { return four(v.x, sh) | (four(v.y, sh) << 4) | (four(v.z, sh) << 8) | (four(v.w, sh) << 12); };
    uint2 r;
    r.x = bits(a, 0) | (bits(b, 0) << 16);                           // response 1 = bit 0, response 4 = bit 2 of the byte
    r.y = bits(a, 2) | (bits(b, 2) << 16);
    *reinterpret_cast<uint2*>(B.bits[blockIdx.y] + (size_t)i * 8) = r;
}

template <int kHi, int kWaves>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(kWaves, kWaves)))
k_coarse_bits(FrameBatch fb, TopBits B, LevelGeom lv, int level, int levels, const TemplEntry* __restrict__ entries, const int32_t* __restrict__ feat_off,
              const int32_t* __restrict__ work_pyramids, int num_work, float threshold, uint32_t cap, uint32_t byte0) {
    constexpr int kN = 4 + kHi, kS = kN + 3;
    __shared__ uint32_t s_tot[4];
    __shared__ unsigned long long s_base;
    const FrameSlot& F = fb.f[blockIdx.y];
    Candidate* __restrict__ cands = F.cands;
    unsigned long long* __restrict__ counters = F.counters;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int work_raw = (int)blockIdx.x * 4 + wave;                       // a wave per template
    const bool live = work_raw < num_work;
    const int work = live ? work_raw : num_work - 1;                       // idle waves of the last workgroup shadow a real template and emit nothing
    const int pyr = work_pyramids[work];
    const TemplEntry e = entries[(size_t)pyr * levels + level];
    const int nf = e.nf, nfp = e.nf_padded;
    const int32_t* fo = feat_off + e.feat_start;
    const int Wd = lv.Wd, Hd = lv.Hd, T = lv.T, npos = Wd * Hd;
    const int wf = (e.width - 1) / T + 1, hf = (e.height - 1) / T + 1;      // LL.cpp:1299-1309
    const int tp = (Hd - hf) * Wd + (Wd - wf) + 1;
    const int offset = T / 2 + (T % 2 - 1);                                 // LL.cpp:1846
    const int limit = tp < npos ? tp : npos;                                // positions that carry sums
    // the smallest raw sum whose score passes (score_of is monotone in raw): LL.cpp:1844 `> threshold`
    int rmin = (int)(threshold * (float)(4 * nf) / 100.f);
    if (!(rmin >= 0)) rmin = 0;
    if (rmin > 4 * nf + 1) rmin = 4 * nf + 1;
    while (rmin > 0 && score_of(rmin, nf) > threshold) --rmin;
    while (rmin <= 4 * nf && !(score_of(rmin, nf) > threshold)) ++rmin;
    const bool zero_passes = score_of(0, nf) > threshold;
    const BufRsrc bits = make_rsrc(B.bits[blockIdx.y]);
    for (int P0 = 0; P0 < npos; P0 += 2048) {                               // (the same number of passes for every wave of the workgroup)
        const int pos0 = P0 + 32 * lane;                                    // first position of this lane
        uint32_t c1[kN], c4[kN];
#pragma unroll
        for (int k = 0; k < kN; ++k) { c1[k] = 0; c4[k] = 0; }
        if (live && P0 < limit && nfp > 0 && pos0 < limit) {                // (lanes beyond the map sit the loads out)
            auto batch = [&](int f, uint32_t& e1, uint32_t& e4) {
                uint4 v[8];
                uint32_t sh[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const uint32_t ob = (uint32_t)fo[f + u] - byte0 + (uint32_t)P0;      // wave-uniform -> SMEM / SALU
                    sh[u] = ob & 31u;
                    v[u] = ld_buf16(bits, (uint32_t)lane * 8u, (ob >> 5) * 8u);          // pairs q + lane and q + lane + 1
                }
                uint32_t x1[8], x4[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    x1[u] = __builtin_amdgcn_alignbit(v[u].z, v[u].x, sh[u]);            // bits [s, s + 32) of {is-1 of pair q + lane + 1 : pair q + lane}
                    x4[u] = __builtin_amdgcn_alignbit(v[u].w, v[u].y, sh[u]);
                }
                e1 = add8(x1, c1[0], c1[1], c1[2]);
                e4 = add8(x4, c4[0], c4[1], c4[2]);
            };
            for (int f = 0; f < nfp; f += 16) {
                uint32_t a1, a4;
                batch(f, a1, a4);
                if (f + 8 < nfp) {
                    uint32_t b1, b4;
                    batch(f + 8, b1, b4);
                    add_eights2<kN>(c1, a1, b1);
                    add_eights2<kN>(c4, a4, b4);
                } else {
                    add_eights1<kN>(c1, a1);
                    add_eights1<kN>(c4, a4);
                }
            }
        }
        uint32_t S[kS], carry = 0;
        S[0] = c1[0]; S[1] = c1[1];
#pragma unroll
        for (int k = 2; k < kS; ++k) csa(S[k], carry, k < kN ? c1[k] : 0u, k - 2 < kN ? c4[k - 2] : 0u, carry);
        uint32_t gt = 0, eq = 0xFFFFFFFFu;                                  // S >= rmin, bit-sliced
#pragma unroll
        for (int k = kS - 1; k >= 0; --k) {
            if ((rmin >> k) & 1) eq &= S[k];                                // wave-uniform
            else { gt |= eq & S[k]; eq &= ~S[k]; }
        }
        uint32_t hit_mask = rmin < (1 << kS) ? (gt | eq) : 0u;
        // positions at or beyond the template's position count hold zero sums (LL.cpp:1299-1309): they are hits only if zero passes
        const int sums = limit - pos0;                                      // positions of this lane that carry sums
        const uint32_t summed = sums >= 32 ? 0xFFFFFFFFu : (sums > 0 ? (1u << sums) - 1u : 0u);
        const int inmap = npos - pos0;
        const uint32_t mapped = inmap >= 32 ? 0xFFFFFFFFu : (inmap > 0 ? (1u << inmap) - 1u : 0u);
        hit_mask = (hit_mask & summed) | (zero_passes ? (mapped & ~summed) : 0u);
        if (!live) hit_mask = 0;
        int total;
        const int before = wave_excl_scan(__popc(hit_mask), lane, total);
        // one atomic per workgroup and pass
        if (lane == 0) s_tot[wave] = (uint32_t)total;
        __syncthreads();
        if (threadIdx.x == 0) {
            const unsigned long long all = (unsigned long long)s_tot[0] + s_tot[1] + s_tot[2] + s_tot[3];
            s_base = all ? atomicAdd(&counters[0], all) : 0ull;
        }
        __syncthreads();
        unsigned long long slot = s_base + (unsigned long long)before;
        for (int w = 0; w < wave; ++w) slot += s_tot[w];
        __syncthreads();                                                    // s_tot / s_base are rewritten by the next pass
        if (total > 0) {                                                    // wave-uniform
            uint32_t m = hit_mask;
            while (m) {
                const int b = __ffs((int)m) - 1;
                m &= m - 1;
                if (slot < cap) {
                    int raw = 0;
                    if ((summed >> b) & 1u) {
#pragma unroll
                        for (int k = 0; k < kS; ++k) raw |= (int)((S[k] >> b) & 1u) << k;
                    }
                    const int jpos = pos0 + b;
                    const int cy = jpos / Wd, cx = jpos - cy * Wd;
                    Candidate c;
                    c.x = cx * T + offset; c.y = cy * T + offset; c.score = score_of(raw, nf); c.work = work;
                    cands[slot] = c;
                }
                ++slot;
            }
        }
    }
}

void launch_pack_top(const TopBits& B, int nb, uint32_t byte0, uint32_t npairs, hipStream_t s) {
    hipLaunchKernelGGL(k_pack_top, dim3((npairs + 255) / 256, nb), dim3(256), 0, s, B, byte0, npairs);
}
void launch_coarse_bits(const FrameBatch& fb, const TopBits& B, const FrameGeom& g, const TemplEntry* entries, const int32_t* feat_off,
                        const int32_t* work_pyramids, int num_work, float threshold, uint32_t cap, uint32_t byte0, int max_features, hipStream_t s) {
    if (num_work <= 0 || fb.nb <= 0) return;
    const int level = g.levels - 1;
#define LM_LAUNCH_COARSE_BITS(HI, WAVES) \
    hipLaunchKernelGGL((k_coarse_bits<HI, WAVES>), dim3((num_work + 3) / 4, fb.nb), dim3(256), 0, s, fb, B, g.lv[level], level, g.levels, entries, feat_off, \
                       work_pyramids, num_work, threshold, cap, byte0)
    if (max_features > kBitsSmallMax) LM_LAUNCH_COARSE_BITS(10, 5);
    else LM_LAUNCH_COARSE_BITS(5, 6);
#undef LM_LAUNCH_COARSE_BITS
}

void launch_local(const FrameBatch& fb, const FrameGeom& g, const TemplEntry* entries, const int32_t* feat_off, const uint32_t* feat_word,
                  const uint32_t* run_mask, const uint32_t* feat_xy, const int32_t* work_pyramids, uint32_t cand_cap, float threshold, uint32_t cap,
                  uint32_t dedupe_cap_slots, uint32_t tile_cap, int grid_blocks, hipStream_t s) {
    if (grid_blocks <= 0 || fb.nb <= 0) return;
    int dbg = 0;
#ifdef LM_DIAG
    dbg = knobs().local_dbg;                                    // timing experiments (wrong results): 1 = tiles only, 2 = singles only
#endif
    hipLaunchKernelGGL(k_local, dim3(grid_blocks), dim3(256), 0, s, fb, g, entries, feat_off, feat_word, run_mask, feat_xy, work_pyramids, cand_cap,
                       threshold, cap, dedupe_cap_slots, tile_cap, dbg);
}

}  // namespace lm
