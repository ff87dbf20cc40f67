// Template extraction on the device: Detector::addTemplate's feature selection for rendered views —
// ColorGradientPyramid::extractTemplate (LL.cpp:589-643), DepthNormalPyramid::extractTemplate (LL.cpp:888-966) and
// QuantizedPyramid::selectScatteredFeatures (LL.cpp:279-318) — after the quantisers of frontend.hip.  gfx950 only.
//
// Per view (pixel-parallel, on the detector's stream right after the view's front end):
//   k_train_mask   object mask = rendered depth > 0 (or the caller's object_mask != 0: single-image addTemplate), its nearest-neighbour pyramid (LL.cpp:576, 877), bounding box
//   k_train_prep   colour candidates: pixels of (mask - erode(mask)) with an orientation and magnitude > strong^2, as sort keys
//                  (score desc, row-major position asc = the order std::stable_sort leaves); normal labels inside erode^2(mask)
//   k_train_runs   per labelled pixel: distance to the end of its same-label run along the row
//   k_train_dt     chessboard distance to the nearest pixel that is not (same label, inside): min over rows of
//                  max(row offset, run distance there) — what cv::distanceTransform(DIST_C, 3) of the label's plane gives at that
//                  pixel (LL.cpp:899-907), searched outwards only as far as the current minimum; candidates with distance >=
//                  extract_threshold, label counts
// Per batch of views (one workgroup per view x level x modality, because the greedy selection is sequential):
//   k_train_select score = distance / label count (normals, LL.cpp:946-949), bitonic sort of the keys in LDS, then
//                  selectScatteredFeatures: candidates in order, 64 at a time — every lane tests its candidate against the
//                  features chosen so far, the survivors are then accepted first-come, each acceptance knocking out the later
//                  lanes it is too close to — with the reference's float distance schedule (start value, -1 per pass over the list).
#include "lm_kernels.h"

namespace lm {

namespace {

constexpr int kInf = 1 << 28;           // "no zero pixel": the value the two-pass chamfer of the host restatement leaves

__device__ __forceinline__ int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

}  // namespace

__global__ void __launch_bounds__(256)
k_train_mask(const uint16_t* __restrict__ depth, const uint8_t* __restrict__ user_mask, TrainGeom g, int32_t* __restrict__ bbox) {
    const int W = g.W[0], H = g.H[0];
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    int x0 = INT_MAX, y0 = INT_MAX, x1 = -1, y1 = -1;
    if (i < W * H) {
        const int x = i % W, y = i / W;
        const bool on = user_mask ? user_mask[i] != 0 : depth[i] > 0;   // addTemplate's object_mask, or the rendered view's depth > 0
        const uint8_t v = on ? 255 : 0;
        g.mask[0][i] = v;
        for (int l = 1; l < g.levels; ++l) {
            const int m = (1 << l) - 1;
            if (((x | y) & m) == 0 && (x >> l) < g.W[l] && (y >> l) < g.H[l]) g.mask[l][(size_t)(y >> l) * g.W[l] + (x >> l)] = v;
        }
        if (on) { x0 = x1 = x; y0 = y1 = y; }
    }
    for (int off = 32; off > 0; off >>= 1) {
        x0 = min(x0, __shfl_xor(x0, off, 64)); y0 = min(y0, __shfl_xor(y0, off, 64));
        x1 = max(x1, __shfl_xor(x1, off, 64)); y1 = max(y1, __shfl_xor(y1, off, 64));
    }
    if ((threadIdx.x & 63) == 0 && x1 >= 0) {                 // bbox = {max(-x), max(-y), max(x), max(y)}, initialised to a large negative value
        atomicMax(&bbox[0], -x0); atomicMax(&bbox[1], -y0); atomicMax(&bbox[2], x1); atomicMax(&bbox[3], y1);
    }
}

// counts: [0] colour candidates, [1] normal candidates, [2] pixels inside erode^2(mask), [3..10] candidates per label
__global__ void __launch_bounds__(256)
k_train_prep(TrainGeom g, float strong_sq, unsigned long long* __restrict__ keys_view, uint32_t cap, uint32_t* __restrict__ counts_view) {
    const int l = blockIdx.y;                                 // one launch for all levels: grid.x covers level 0, higher levels leave early
    unsigned long long* __restrict__ ckeys = keys_view + ((size_t)l * 2 + 0) * cap;
    uint32_t* __restrict__ counts = counts_view + (size_t)l * 16;
    const int W = g.W[l], H = g.H[l];
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const uint8_t* mask = g.mask[l];
    bool cand = false, inside = false;
    unsigned long long key = 0;
    if (i < W * H) {
        const int x = i % W, y = i / W;
        uint8_t lab = 0;
        if (mask[i]) {
            uint8_t e3 = 255, e5 = 255;                       // cv::erode(3x3, BORDER_REPLICATE) once / twice (LL.cpp:595, 894)
            for (int dy = -2; dy <= 2; ++dy) {
                const int yy = clampi(y + dy, 0, H - 1);
                for (int dx = -2; dx <= 2; ++dx) {
                    const uint8_t v = mask[(size_t)yy * W + clampi(x + dx, 0, W - 1)];
                    e5 = min(e5, v);
                    if (dy >= -1 && dy <= 1 && dx >= -1 && dx <= 1) e3 = min(e3, v);
                }
            }
            const uint8_t q = g.ang[l][i];
            const float m = g.mag[l][i];
            if (!e3 && q > 0 && m > strong_sq) {              // mask - erode(mask): the object's one-pixel rim (LL.cpp:596-624)
                cand = true;
                key = ((unsigned long long)(~__float_as_uint(m)) << 32) | ((uint32_t)i << 3) | (uint32_t)(__ffs((int)q) - 1);
            }
            if (e5) {
                inside = true;
                const uint8_t n = g.nrm[l][i];
                if (n != 0 && n != 255) lab = (uint8_t)__ffs((int)n);   // label + 1 (one-hot by construction of the quantiser)
            }
        }
        g.lab[l][i] = lab;
    }
    const unsigned long long mc = __ballot(cand), mi = __ballot(inside);
    const int lane = threadIdx.x & 63;
    if (mc) {
        uint32_t base = 0;
        if (lane == 0) base = atomicAdd(&counts[0], (uint32_t)__popcll(mc));
        base = __shfl(base, 0, 64);
        const uint32_t at = base + __popcll(mc & ((1ull << lane) - 1ull));
        if (cand && at < cap) ckeys[at] = key;
    }
    if (mi && lane == 0) atomicAdd(&counts[2], (uint32_t)__popcll(mi));
}

__global__ void __launch_bounds__(256)
k_train_runs(TrainGeom g) {
    const int l = blockIdx.y;
    const int W = g.W[l], H = g.H[l];
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= W * H) return;
    const uint8_t* lab = g.lab[l];
    const uint8_t mine = lab[i];
    if (!mine) return;
    const int x = i % W;
    const uint8_t* row = lab + (size_t)(i - x);
    int a = x - 1, b = x + 1;
    while (a >= 0 && row[a] == mine) --a;
    while (b < W && row[b] == mine) ++b;
    const int left = a >= 0 ? x - a : kInf, right = b < W ? b - x : kInf;      // beyond the image there is no zero pixel
    g.hrun[l][i] = min(left, right);
}

__global__ void __launch_bounds__(256)
k_train_dt(TrainGeom g, int extract_threshold0, unsigned long long* __restrict__ keys_view, uint32_t cap, uint32_t* __restrict__ counts_view) {
    const int l = blockIdx.y;
    const int extract_threshold = extract_threshold0 >> l;    // extract_threshold /= 2 per level (LL.cpp:861)
    unsigned long long* __restrict__ nkeys = keys_view + ((size_t)l * 2 + 1) * cap;
    uint32_t* __restrict__ counts = counts_view + (size_t)l * 16;
    const int W = g.W[l], H = g.H[l];
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    bool cand = false;
    unsigned long long rec = 0;
    int mine = 0;
    if (i < W * H) {
        const uint8_t* lab = g.lab[l];
        const int32_t* hrun = g.hrun[l];
        mine = lab[i];
        if (mine) {
            const int x = i % W, y = i / W;
            int D = hrun[i];
            for (int k = 1; k < D; ++k) {
                if (y - k < 0 && y + k >= H) break;
                if (y - k >= 0) {
                    const size_t o = (size_t)(y - k) * W + x;
                    const int gk = lab[o] == mine ? hrun[o] : 0;
                    D = min(D, max(k, gk));
                }
                if (y + k < H && k < D) {
                    const size_t o = (size_t)(y + k) * W + x;
                    const int gk = lab[o] == mine ? hrun[o] : 0;
                    D = min(D, max(k, gk));
                }
            }
            if (D >= extract_threshold) {                      // LL.cpp:930
                cand = true;
                rec = ((unsigned long long)(uint32_t)D << 32) | ((uint32_t)i << 3) | (uint32_t)(mine - 1);
            }
        }
    }
    const unsigned long long mc = __ballot(cand);
    const int lane = threadIdx.x & 63;
    if (mc) {
        uint32_t base = 0;
        if (lane == 0) base = atomicAdd(&counts[1], (uint32_t)__popcll(mc));
        base = __shfl(base, 0, 64);
        const uint32_t at = base + __popcll(mc & ((1ull << lane) - 1ull));
        if (cand && at < cap) nkeys[at] = rec;
        if (cand) atomicAdd(&counts[3 + mine - 1], 1u);
    }
}

// grid (levels * 2, views).  keys: [views][levels][2][cap]; counts: [views][levels][16]; out: [views][levels][2][4 + 3 * nf_cap]
// (status: 1 ok, 0 too few candidates, 2 the list did not fit `cap`; number of features; then x, y, label triples).
__global__ void __launch_bounds__(256)
k_train_select(const unsigned long long* __restrict__ keys_all, const uint32_t* __restrict__ counts_all, TrainGeom g, uint32_t cap,
               int num_features, int nf_cap, int32_t* __restrict__ out_all) {
    extern __shared__ unsigned long long s_keys[];            // [P2]
    __shared__ short s_fx[kTrainMaxFeatures], s_fy[kTrainMaxFeatures];
    const int l = blockIdx.x >> 1, mod = blockIdx.x & 1, view = blockIdx.y;
    const int W = g.W[l];
    const unsigned long long* keys = keys_all + (((size_t)view * g.levels + l) * 2 + mod) * cap;
    const uint32_t* counts = counts_all + ((size_t)view * g.levels + l) * 16;
    int32_t* out = out_all + (((size_t)view * g.levels + l) * 2 + mod) * (4 + 3 * (size_t)nf_cap);
    const int nf = num_features >> l;                         // num_features /= 2 per level (LL.cpp:560, 860)
    const uint32_t n = counts[mod];
    if (n > cap) { if (threadIdx.x == 0) { out[0] = 2; out[1] = 0; } return; }
    if (nf <= 0 || nf > kTrainMaxFeatures) { if (threadIdx.x == 0) { out[0] = 2; out[1] = 0; } return; }   // left to the host path
    if ((int)n < nf) { if (threadIdx.x == 0) { out[0] = 0; out[1] = 0; } return; }                           // LL.cpp:626, 943
    uint32_t P2 = 1;
    while (P2 < n) P2 <<= 1;
    for (uint32_t i = threadIdx.x; i < P2; i += blockDim.x) {
        unsigned long long k = ~0ull;
        if (i < n) {
            k = keys[i];
            if (mod == 1) {                                   // distance / candidates of that label (LL.cpp:946-949), as a descending key
                const uint32_t lab = (uint32_t)k & 7u;
                const float score = __fdiv_rn((float)(uint32_t)(k >> 32), (float)counts[3 + lab]);
                k = ((unsigned long long)(~__float_as_uint(score)) << 32) | (uint32_t)k;
            }
        }
        s_keys[i] = k;
    }
    __syncthreads();
    for (uint32_t k = 2; k <= P2; k <<= 1)
        for (uint32_t j = k >> 1; j > 0; j >>= 1) {
            for (uint32_t t = threadIdx.x; t < P2; t += blockDim.x) {
                const uint32_t p = t ^ j;
                if (p > t) {
                    const unsigned long long a = s_keys[t], b = s_keys[p];
                    if ((b < a) == ((t & k) == 0)) { s_keys[t] = b; s_keys[p] = a; }
                }
            }
            __syncthreads();
        }
    if (threadIdx.x >= 64) return;                            // the selection is sequential: one wave
    const int lane = threadIdx.x;
    float distance;
    if (mod == 0) distance = (float)(n / (uint32_t)nf + 1);   // LL.cpp:632
    else distance = __fadd_rn(__fdiv_rn(__fsqrt_rn((float)counts[2]), __fsqrt_rn((float)nf)), 1.5f);   // LL.cpp:957
    float dsq = __fmul_rn(distance, distance);
    int ns = 0;
    uint32_t base = 0;
    while (ns < nf) {
        const uint32_t i = base + lane;
        const bool have = i < n;
        int cx = 0, cy = 0;
        uint32_t low = 0;
        if (have) { low = (uint32_t)s_keys[i]; const int idx = (int)(low >> 3); cx = idx % W; cy = idx / W; }
        bool keep = have;
        for (int j = 0; j < ns && __any(keep); ++j) {         // against the features chosen so far (LL.cpp:291-296)
            const int dx = cx - s_fx[j], dy = cy - s_fy[j];
            if ((float)(dx * dx + dy * dy) < dsq) keep = false;
        }
        unsigned long long m = __ballot(keep);
        while (m && ns < nf) {                                // survivors in order: the first is accepted, later ones must clear it too
            const int f = __ffsll((long long)m) - 1;
            const int fx = __shfl(cx, f, 64), fy = __shfl(cy, f, 64);
            const uint32_t flow = (uint32_t)__shfl((int)low, f, 64);
            if (lane == 0) {
                s_fx[ns] = (short)fx; s_fy[ns] = (short)fy;
                out[4 + 3 * ns] = fx; out[4 + 3 * ns + 1] = fy; out[4 + 3 * ns + 2] = (int)(flow & 7u);
            }
            ++ns;
            if (lane == f) keep = false;
            else if (keep && lane > f) {
                const int dx = cx - fx, dy = cy - fy;
                if ((float)(dx * dx + dy * dy) < dsq) keep = false;
            }
            m = __ballot(keep);
        }
        base += 64;
        if (base >= n) {                                      // end of the list: start over with a smaller distance (LL.cpp:299-304)
            base = 0;
            distance = __fsub_rn(distance, 1.0f);
            dsq = __fmul_rn(distance, distance);
        }
    }
    if (lane == 0) { out[0] = 1; out[1] = ns; }
}

void launch_train_prep(const uint16_t* depth, const uint8_t* user_mask, const TrainGeom& g, float strong_sq, int extract_threshold, unsigned long long* keys_view,
                       uint32_t cap, uint32_t* counts_view, int32_t* bbox_view, hipStream_t s) {
    const int n0 = g.W[0] * g.H[0];
    hipLaunchKernelGGL(k_train_mask, dim3((n0 + 255) / 256), dim3(256), 0, s, depth, user_mask, g, bbox_view);
    const dim3 grid((n0 + 255) / 256, g.levels);              // level l uses the first W_l * H_l / 256 blocks of its row
    hipLaunchKernelGGL(k_train_prep, grid, dim3(256), 0, s, g, strong_sq, keys_view, cap, counts_view);
    hipLaunchKernelGGL(k_train_runs, grid, dim3(256), 0, s, g);
    hipLaunchKernelGGL(k_train_dt, grid, dim3(256), 0, s, g, extract_threshold < 0 ? 0 : extract_threshold, keys_view, cap, counts_view);
}

int launch_train_select(const unsigned long long* keys, const uint32_t* counts, const TrainGeom& g, uint32_t cap, int num_features, int nf_cap,
                        int views, int32_t* out, hipStream_t s) {
    static size_t configured = 0;
    const size_t lds = (size_t)cap * sizeof(unsigned long long);
    if (lds > configured) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(k_train_select), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return -1;
        configured = lds;
    }
    hipLaunchKernelGGL(k_train_select, dim3(g.levels * 2, views), dim3(256), lds, s, keys, counts, g, cap, num_features, nf_cap, out);
    return 0;
}

}  // namespace lm
