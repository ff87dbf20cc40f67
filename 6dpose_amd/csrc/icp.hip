// poseRefine::process on gfx950 (reference LL.cpp:27-155).  The cloud arithmetic is Open3D's
// (un-vendored): VoxelDownSample (LL.cpp:108-109), EstimateNormals(KNN 30) (LL.cpp:127),
// RegistrationICP + TransformationEstimationPointToPlane (LL.cpp:128-130), restated per SURVEY
// Appendix B with the deterministic rules of DESIGN.md §5 (shared with oracle/linemod_oracle.py).
//
// One stream of launches per batch of hypotheses, no host round trip in between:
//   k_icp_bbox     bounding box of modelDepth > 0                                   (LL.cpp:43-50)
//   k_icp_points   dilated mask, back-projection, raster-order compaction, centroids (LL.cpp:52-104)
//   k_icp_voxel    VoxelDownSample: stable radix sort of the point indices by voxel, segment means
//   k_icp_grid     bins the target cloud into <= 64 x 64 xy columns (cell >= 5 mm), sorted by (column, depth step)
//   k_icp_knn      eight lanes per target point: ring search over the columns, the k nearest selected by counting
//                  passes (ties by index), cumulants; whole waves / k_icp_knn_far for the points whose ring grows
//   k_icp_normals  covariance + Jacobi eigenvector, one thread per point
//   k_icp_eval     once per ICP evaluation (<= 31 + 1): exact nearest neighbours through the
//                  grid (search radius = distance to the previous correspondence), 29 double sums by a
//                  halving wave reduction, then 6x6 LU, Rz*Ry*Rx update and the convergence test.
// All arithmetic is double like Open3D's (f64 VALU; nothing here is a dense contraction, so no MFMA).
// The grid only prunes: candidate distances are the same expression the oracle evaluates and ties go
// to the lower original index, so correspondences equal a brute-force search.
#include <limits.h>
#include <stdlib.h>

#include <atomic>

#include "icp_kernels.h"
#include "knobs.h"
#include "lm_kernels.h"

namespace lm {

constexpr int kWG = 1024;          // workgroup size of the per-hypothesis kernels
constexpr int kSortLds = 16384;    // 64-bit keys sorted in LDS (128 KiB); longer lists use the global scratch
constexpr int kDilate = 4;         // LL.cpp:45 (9x9 dilation)
constexpr double kCellMin = 0.005; // search-grid cell edge (m), grown until the grid fits kIcpGrid / kIcpCells
constexpr int kIdxBits = 22;       // point-index bits of the grid sort key

static __device__ __forceinline__ double sqdist(double ax, double ay, double az, double bx, double by, double bz) {
    double dx = __dsub_rn(ax, bx), dy = __dsub_rn(ay, by), dz = __dsub_rn(az, bz);
    return __dadd_rn(__dadd_rn(__dmul_rn(dx, dx), __dmul_rn(dy, dy)), __dmul_rn(dz, dz));
}

static __device__ __forceinline__ double shfl_xor_d(double v, int m) { return __shfl_xor(v, m, 64); }

static __device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    return v;
}

// Exclusive prefix of a per-thread flag in thread order; `total` = number of flags set in the workgroup.
static __device__ __forceinline__ int block_scan_flag(bool flag, int* s_wave, int& total) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    const unsigned long long b = __ballot(flag);
    const int within = __popcll(b & ((1ull << lane) - 1ull));
    __syncthreads();
    if (lane == 0) s_wave[wave] = __popcll(b);
    __syncthreads();
    int base = 0, tot = 0;
    for (int w = 0; w < nw; ++w) {
        const int c = s_wave[w];
        if (w < wave) base += c;
        tot += c;
    }
    total = tot;
    return base + within;
}

// min / max of K doubles over the workgroup; result in s_out[0..K) (min) and s_out[K..2K) (max), visible after return.
template <int K>
static __device__ __forceinline__ void block_minmax(const double (&mn)[K], const double (&mx)[K], double* s_part /*[16][2K]*/,
                                                    double* s_out /*[2K]*/) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    double a[K], b[K];
#pragma unroll
    for (int k = 0; k < K; ++k) {
        a[k] = mn[k]; b[k] = mx[k];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            a[k] = fmin(a[k], shfl_xor_d(a[k], o));
            b[k] = fmax(b[k], shfl_xor_d(b[k], o));
        }
    }
    __syncthreads();
    if (lane == 0) {
#pragma unroll
        for (int k = 0; k < K; ++k) { s_part[wave * 2 * K + k] = a[k]; s_part[wave * 2 * K + K + k] = b[k]; }
    }
    __syncthreads();
    if (threadIdx.x < 2 * K) {
        double v = s_part[threadIdx.x];
        for (int w = 1; w < nw; ++w) {
            const double u = s_part[w * 2 * K + threadIdx.x];
            v = threadIdx.x < K ? fmin(v, u) : fmax(v, u);
        }
        s_out[threadIdx.x] = v;
    }
    __syncthreads();
}

// Ascending bitonic sort of npad (power of two) 64-bit keys by one workgroup.
template <typename P>
static __device__ __forceinline__ void bitonic_sort(P A, int npad) {
    for (int k = 2; k <= npad; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int t = threadIdx.x; t < (npad >> 1); t += blockDim.x) {
                const int i = ((t & ~(j - 1)) << 1) | (t & (j - 1));
                const int l = i | j;
                const unsigned long long a = A[i], b = A[l];
                const bool up = (i & k) == 0;
                if ((a > b) == up) { A[i] = b; A[l] = a; }
            }
            __syncthreads();
        }
    }
}

// The same for lists longer than the LDS buffer (npad > kSortLds keys in HBM): chunks of kSortLds keys are
// sorted / merged in LDS, only the compare-exchange steps whose partner lies in another chunk touch HBM —
// log2(npad / kSortLds) * (log2(npad / kSortLds) + 1) / 2 passes instead of ~log2(npad)^2 / 2.
static __device__ __forceinline__ void bitonic_sort_hybrid(unsigned long long* g, int npad, unsigned long long* s) {
    const int C = kSortLds;
    auto chunk_steps = [&](int c0, int k, int jmax) {            // steps j = jmax .. 1 of merge size k on the chunk at c0
        for (int i = threadIdx.x; i < C; i += blockDim.x) s[i] = g[c0 + i];
        __syncthreads();
        for (int j = jmax; j > 0; j >>= 1) {
            for (int t = threadIdx.x; t < (C >> 1); t += blockDim.x) {
                const int i = ((t & ~(j - 1)) << 1) | (t & (j - 1));
                const int l = i | j;
                const unsigned long long a = s[i], b = s[l];
                const bool up = ((c0 + i) & k) == 0;
                if ((a > b) == up) { s[i] = b; s[l] = a; }
            }
            __syncthreads();
        }
        for (int i = threadIdx.x; i < C; i += blockDim.x) g[c0 + i] = s[i];
        __syncthreads();
    };
    for (int c0 = 0; c0 < npad; c0 += C)
        for (int k = 2; k <= C; k <<= 1) {
            if (k == 2) {                                          // load once, run every k <= C in LDS, store once
                for (int i = threadIdx.x; i < C; i += blockDim.x) s[i] = g[c0 + i];
                __syncthreads();
            }
            for (int j = k >> 1; j > 0; j >>= 1) {
                for (int t = threadIdx.x; t < (C >> 1); t += blockDim.x) {
                    const int i = ((t & ~(j - 1)) << 1) | (t & (j - 1));
                    const int l = i | j;
                    const unsigned long long a = s[i], b = s[l];
                    const bool up = ((c0 + i) & k) == 0;
                    if ((a > b) == up) { s[i] = b; s[l] = a; }
                }
                __syncthreads();
            }
            if (k == C) {
                for (int i = threadIdx.x; i < C; i += blockDim.x) g[c0 + i] = s[i];
                __syncthreads();
            }
        }
    for (int k = 2 * C; k <= npad; k <<= 1) {
        for (int j = k >> 1; j >= C; j >>= 1) {                    // partner in another chunk: HBM pass
            for (int t = threadIdx.x; t < (npad >> 1); t += blockDim.x) {
                const int i = ((t & ~(j - 1)) << 1) | (t & (j - 1));
                const int l = i | j;
                const unsigned long long a = g[i], b = g[l];
                const bool up = (i & k) == 0;
                if ((a > b) == up) { g[i] = b; g[l] = a; }
            }
            __syncthreads();
        }
        for (int c0 = 0; c0 < npad; c0 += C) chunk_steps(c0, k, C >> 1);
    }
}

// Stable LSD radix sort by one workgroup of 1024 threads: n <= kSortLds 16-bit indices ordered by their 32-bit keys (the keys
// stay put in LDS, the indices move), 8 bits per pass.  Every wave owns a contiguous chunk of the list; inside a round of 64
// the lanes with the same digit find each other with 8 ballots (rank = lanes before me with my digit), the per-wave digit
// counts are prefix-summed digit-major / wave-minor, and the second walk scatters.  3 passes for the 24-bit voxel index of
// a 0.6 m cloud, 4 for the grid key — against ~105 barrier-separated compare-exchange steps of the bitonic network.
// Lists of up to 2 x kSortLds points keep their keys in HBM (u32, read through L2) and have the whole buffer for indices:
// keys | idx | idx | counts = 64 | 32 | 32 | 16 KiB for n <= 16384, idx | idx | counts = 64 | 64 | 16 KiB for n <= 32768.
constexpr int kRadixBig = 2 * kSortLds;
constexpr int kRadixLdsBytes = 2 * kRadixBig * 2 + 16 * 256 * 4;   // 144 KiB
struct RadixView { unsigned short* idx[2]; unsigned short* hist; unsigned int* key_lds; };   // hist: [16 waves][2^digit bits] 16-bit counts / positions
static __device__ __forceinline__ RadixView radix_view(unsigned char* raw, const bool big) {
    RadixView v;
    v.key_lds = reinterpret_cast<unsigned int*>(raw);
    v.idx[0] = reinterpret_cast<unsigned short*>(raw + (big ? 0 : kSortLds * 4));
    v.idx[1] = v.idx[0] + (big ? kRadixBig : kSortLds);
    v.hist = reinterpret_cast<unsigned short*>(raw + 2 * kRadixBig * 2);
    return v;
}

// One pass structure for 8- and 9-bit digits (DB): 9 bits when that saves a pass (a 25-bit voxel index: 3 passes instead of 4, a
// ninth ballot per round of 64 keys); the counts of a wave's chunk (<= 2048 keys) and the positions (< 32768) fit 16 bits, so
// 16 waves x 512 digits still are 16 KiB.
template <int DB, typename KeyF>
static __device__ __forceinline__ const unsigned short* wg_radix_passes(const RadixView& m, const int n, const int bits, int* s_wave /*[32]*/, KeyF&& key_of) {
    constexpr int ND = 1 << DB, E = ND / 64;                      // digits; (digit, wave) entries a thread owns in the prefix
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int C = ((n + 1023) >> 10) << 6;                         // chunk of a wave: a multiple of 64, 16 chunks cover n
    unsigned short* hist = m.hist;
    for (int i = tid; i < n; i += kWG) m.idx[0][i] = (unsigned short)i;
    int cur = 0;
    for (int shift = 0; shift < bits; shift += DB) {
        const unsigned short* in = m.idx[cur];
        unsigned short* out = m.idx[cur ^ 1];
        for (int d = lane; d < ND; d += 64) hist[wave * ND + d] = 0;
        __syncthreads();                                             // (the indices of the previous pass are written)
        for (int walk = 0; walk < 2; ++walk) {
            for (int r = 0; r < C; r += 64) {
                const int i = wave * C + r + lane;
                const bool active = i < n;
                const unsigned int id = active ? in[i] : 0;
                const unsigned int d = active ? (key_of(id) >> shift) & (unsigned int)(ND - 1) : 0;
                unsigned long long peers = __ballot(active);
#pragma unroll
                for (int b = 0; b < DB; ++b) {
                    const unsigned long long bb = __ballot(active && ((d >> b) & 1u));
                    peers &= ((d >> b) & 1u) ? bb : ~bb;
                }
                const int cnt = __popcll(peers), rank = __popcll(peers & ((1ull << lane) - 1ull));
                if (walk == 0) {
                    if (active && rank == 0) hist[wave * ND + d] = (unsigned short)(hist[wave * ND + d] + cnt);
                } else {
                    unsigned int base = 0;
                    if (active) base = hist[wave * ND + d];
                    if (active) out[base + rank] = (unsigned short)id;
                    if (active && rank == 0) hist[wave * ND + d] = (unsigned short)(base + cnt);
                }
            }
            if (walk == 0) {
                // exclusive prefix over (digit, wave), digit-major: thread t owns the E consecutive entries from t * E
                __syncthreads();
                const int d = (tid * E) >> 4, w0 = (tid * E) & 15;
                unsigned int c[E], sum = 0;
#pragma unroll
                for (int q = 0; q < E; ++q) { c[q] = hist[(w0 + q) * ND + d]; sum += c[q]; }
                unsigned int incl = sum;
#pragma unroll
                for (int o = 1; o < 64; o <<= 1) { const unsigned int t = __shfl_up(incl, o, 64); if (lane >= o) incl += t; }
                if (lane == 63) s_wave[wave] = (int)incl;
                __syncthreads();
                unsigned int base = incl - sum;
                for (int w = 0; w < wave; ++w) base += (unsigned int)s_wave[w];
#pragma unroll
                for (int q = 0; q < E; ++q) { hist[(w0 + q) * ND + d] = (unsigned short)base; base += c[q]; }
                __syncthreads();
            }
        }
        cur ^= 1;
    }
    __syncthreads();
    return m.idx[cur];
}
template <typename KeyF>
static __device__ __forceinline__ const unsigned short* wg_radix_sort(const RadixView& m, const int n, const int bits, int* s_wave /*[32]*/, KeyF&& key_of) {
    return (bits + 8) / 9 < (bits + 7) / 8 ? wg_radix_passes<9>(m, n, bits, s_wave, key_of) : wg_radix_passes<8>(m, n, bits, s_wave, key_of);
}

static __device__ __forceinline__ int next_pow2(int n) {
    int p = 1;
    while (p < n) p <<= 1;
    return p;
}

static __device__ __forceinline__ int bits_for(long long vmax) {   // bits needed for values 0..vmax
    return vmax <= 0 ? 1 : 64 - __clzll((unsigned long long)vmax);
}

static __device__ __forceinline__ int grid_coord(double v, double mn, double inv, int g) {
    const double f = floor((v - mn) * inv);
    return f >= 0.0 ? (f < (double)g ? (int)f : g - 1) : 0;      // NaN -> 0, never UB
}

// ---------------------------------------------------------------------------------------------
// k_icp_bbox: bounding rectangle of modelDepth > 0 (the 9x9 dilation only grows it by 4, LL.cpp:43-50)
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
k_icp_bbox(IcpBuffers B, int W, int H) {
    const int h = blockIdx.y;
    const int status = B.st[h].status, slot = B.in[h].model_slot;   // (both loads leave together)
    if (status != 0) return;                                       // slot without a detection (pipeline)
    // the box of a resident image is worked out once: a later run finds it in model_bbox (k_icp_points<false> of the first run put it there
    // once this kernel was through; the host clears the state word when the image changes)
    if (blockIdx.x == 0 && threadIdx.x < kIcpStrips) B.strip_pub[(size_t)h * kIcpStrips + threadIdx.x] = 0;   // (k_icp_points_fused: the strips' counts, not yet known)
    const int* known = B.model_bbox + (size_t)slot * 8;
    if (known[4] == 1) {
        if (blockIdx.x == 0 && threadIdx.x < 4) B.st[h].bbox[threadIdx.x] = threadIdx.x < 2 ? INT_MAX - known[threadIdx.x] : known[threadIdx.x] - 1;
        return;
    }
    const uint16_t* img = B.models + (size_t)slot * W * H;
    int x0 = INT_MAX, y0 = INT_MAX, x1 = -1, y1 = -1;
    const bool vec = (W & 7) == 0;
    for (int y = blockIdx.x; y < H; y += gridDim.x) {
        const uint16_t* row = img + (size_t)y * W;
        for (int x = threadIdx.x * 8; x < W; x += blockDim.x * 8) {
            uint16_t px[8];
            if (vec) {
                const uint4 v = *reinterpret_cast<const uint4*>(row + x);
                px[0] = v.x & 0xFFFF; px[1] = v.x >> 16; px[2] = v.y & 0xFFFF; px[3] = v.y >> 16;
                px[4] = v.z & 0xFFFF; px[5] = v.z >> 16; px[6] = v.w & 0xFFFF; px[7] = v.w >> 16;
            } else {
#pragma unroll
                for (int k = 0; k < 8; ++k) px[k] = x + k < W ? row[x + k] : 0;
            }
#pragma unroll
            for (int k = 0; k < 8; ++k)
                if (px[k]) {
                    x0 = min(x0, x + k); x1 = max(x1, x + k);
                    y0 = min(y0, y); y1 = max(y1, y);
                }
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        x0 = min(x0, __shfl_xor(x0, o, 64)); y0 = min(y0, __shfl_xor(y0, o, 64));
        x1 = max(x1, __shfl_xor(x1, o, 64)); y1 = max(y1, __shfl_xor(y1, o, 64));
    }
    if ((threadIdx.x & 63) == 0 && x1 >= 0) {
        int* bb = B.st[h].bbox;
        atomicMin(&bb[0], x0); atomicMin(&bb[1], y0); atomicMax(&bb[2], x1); atomicMax(&bb[3], y1);
    }
}

// k_icp_model_boxes: the same rectangle for resident model images AT UPLOAD (lm_icp_set_models, the pipeline's view upload): 32 workgroups
// per image, model_bbox[slot] = INT_MAX - x0, INT_MAX - y0, x1 + 1, y1 + 1 (so that a cleared record is the empty box and every word
// only grows: atomicMax), state 1.  A run whose slots all came that way does not launch k_icp_bbox at all: reading a 614 KB image per
// hypothesis and run was 12 us of every run for a fact that changes when the image does.
__global__ void __launch_bounds__(256)
k_icp_model_boxes(const uint16_t* __restrict__ models, int* __restrict__ model_bbox, int first_slot, int W, int H) {
    const int slot = first_slot + (int)blockIdx.y;
    const uint16_t* img = models + (size_t)slot * W * H;
    int x0 = INT_MAX, y0 = INT_MAX, x1 = -1, y1 = -1;
    const bool vec = (W & 7) == 0;
    for (int y = blockIdx.x; y < H; y += gridDim.x) {
        const uint16_t* row = img + (size_t)y * W;
        for (int x = threadIdx.x * 8; x < W; x += 256 * 8) {
            uint16_t px[8];
            if (vec) {
                const uint4 v = *reinterpret_cast<const uint4*>(row + x);
                px[0] = v.x & 0xFFFF; px[1] = v.x >> 16; px[2] = v.y & 0xFFFF; px[3] = v.y >> 16;
                px[4] = v.z & 0xFFFF; px[5] = v.z >> 16; px[6] = v.w & 0xFFFF; px[7] = v.w >> 16;
            } else {
#pragma unroll
                for (int k = 0; k < 8; ++k) px[k] = x + k < W ? row[x + k] : 0;
            }
#pragma unroll
            for (int k = 0; k < 8; ++k)
                if (px[k]) {
                    x0 = min(x0, x + k); x1 = max(x1, x + k);
                    y0 = min(y0, y); y1 = max(y1, y);
                }
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        x0 = min(x0, __shfl_xor(x0, o, 64)); y0 = min(y0, __shfl_xor(y0, o, 64));
        x1 = max(x1, __shfl_xor(x1, o, 64)); y1 = max(y1, __shfl_xor(y1, o, 64));
    }
    int* known = model_bbox + (size_t)slot * 8;
    if ((threadIdx.x & 63) == 0 && x1 >= 0) {
        atomicMax(&known[0], INT_MAX - x0); atomicMax(&known[1], INT_MAX - y0); atomicMax(&known[2], x1 + 1); atomicMax(&known[3], y1 + 1);
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) known[4] = 1;        // (read by the kernels of a later launch)
}

void launch_icp_model_boxes(const uint16_t* models, int* model_bbox, int first_slot, int count, int W, int H, hipStream_t s) {
    if (count <= 0) return;
    (void)hipMemsetAsync(model_bbox + (size_t)first_slot * 8, 0, (size_t)count * 8 * sizeof(int), s);
    hipLaunchKernelGGL(k_icp_model_boxes, dim3(32, count), dim3(256), 0, s, models, model_bbox, first_slot, W, H);
}

// ---------------------------------------------------------------------------------------------
// k_icp_points (LL.cpp:52-104): raster scan of the dilated bounding box; model point where
// modelDepth > 0, scene point where the dilated mask is set and sceneDepth (window shifted by
// detect - 4, clamped at 0) > 0; compaction keeps raster order; centroid difference = init_guess.
// The box is cut into kIcpStrips row strips, one workgroup each: pass 0 counts the points of every
// strip, pass 1 starts each strip at the sum of the counts before it and writes points + centroid sums
// (k_icp_grid adds the strips' sums in order -> init_guess).
// ---------------------------------------------------------------------------------------------
constexpr int kPtsWG = 256;

template <bool kWrite>
__global__ void __launch_bounds__(kPtsWG)
k_icp_points(IcpBuffers B, int W, int H, int flags) {
    __shared__ int s_wave[8];
    __shared__ double s_red[kPtsWG / 64][7];
    __shared__ double s_ext[kPtsWG / 64][12];
    __shared__ int s_tot[2];
    const int h = blockIdx.y, strip = blockIdx.x, tid = threadIdx.x;
    IcpState& S = B.st[h];
    if (S.status != 0) return;
    const IcpIn I = B.in[h];
    const int x0 = S.bbox[0], y0 = S.bbox[1], x1 = S.bbox[2], y1 = S.bbox[3];
    if (!kWrite && strip == 0 && tid == 0) {                       // k_icp_bbox is through: the box of this image is known from now on
        int* known = B.model_bbox + (size_t)B.in[h].model_slot * 8;
        if (known[4] == 0) { known[0] = INT_MAX - x0; known[1] = INT_MAX - y0; known[2] = x1 + 1; known[3] = y1 + 1; __threadfence(); known[4] = 1; }
    }
    if (x1 < 0) {                                                  // pass 1 never gets here: pass 0 set the status
        if (strip == 0 && tid == 0) { S.status = 2; S.n_model = 0; S.n_scene = 0; }
        return;
    }
    const int bx0 = max(x0 - kDilate, 0), by0 = max(y0 - kDilate, 0);
    const int bx1 = min(x1 + kDilate, W - 1), by1 = min(y1 + kDilate, H - 1);
    const int bw = bx1 - bx0 + 1, bh = by1 - by0 + 1;
    if (I.dx + bw >= W || I.dy + bh >= H) {                       // LL.cpp:52-55
        if (strip == 0 && tid == 0) { S.status = 1; S.n_model = 0; S.n_scene = 0; }
        return;
    }
    const uint16_t* model = B.models + (size_t)I.model_slot * W * H;
    const uint16_t* scene = B.scene;
    int* cnt = B.strip_cnt + ((size_t)h * kIcpStrips) * 2;
    const int r_lo = (int)((long long)bh * strip / kIcpStrips), r_hi = (int)((long long)bh * (strip + 1) / kIcpStrips);
    const int p_lo = r_lo * bw, p_hi = r_hi * bw;
    const bool keep_scene = (flags & 1) != 0;

    if (!kWrite) {
        int cm = 0, cs = 0;
        for (int p = p_lo + tid; p < p_hi; p += kPtsWG) {
            const int r = p / bw, c = p - r * bw;
            const int mr = r + by0, mc = c + bx0;
            const int sr = max(r + I.dy - kDilate, 0), sc = max(c + I.dx - kDilate, 0);
            const uint16_t md = model[(size_t)mr * W + mc];
            const uint16_t sd = scene[(size_t)sr * W + sc];
            cm += md > 0;
            if (sd > 0 && keep_scene) {
                bool in_mask = md > 0;
                if (!in_mask) {                                   // dilate(modelDepth > 0, 9x9) at (mr, mc)
                    const int ya = max(mr - kDilate, 0), yb = min(mr + kDilate, H - 1);
                    const int xa = max(mc - kDilate, 0), xb = min(mc + kDilate, W - 1);
                    for (int yy = ya; yy <= yb && !in_mask; ++yy)
                        for (int xx = xa; xx <= xb; ++xx)
                            if (model[(size_t)yy * W + xx]) { in_mask = true; break; }
                }
                cs += in_mask;
            }
        }
        if (tid < 2) s_tot[tid] = 0;
        __syncthreads();
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { cm += __shfl_xor(cm, o, 64); cs += __shfl_xor(cs, o, 64); }
        if ((tid & 63) == 0) { atomicAdd(&s_tot[0], cm); atomicAdd(&s_tot[1], cs); }
        __syncthreads();
        if (tid < 2) cnt[strip * 2 + tid] = s_tot[tid];
        return;
    }

    if (strip == 0 && tid < 2 * kIcpSortGroups) B.sort_look[(size_t)h * 2 * kIcpSortGroups + tid] = 0;   // (k_icp_voxel_wide: voxel counts of the groups, not yet known)
    const double anchor = model[(size_t)(H / 2) * W + W / 2] / 1000.0;   // LL.cpp:62
    double* mp = B.model_pts + (size_t)h * B.cap * 3;
    double* sp = B.scene_pts + (size_t)h * B.cap * 3;
    int nm = 0, nsn = 0, tot_m = 0, tot_s = 0;
    for (int k = 0; k < kIcpStrips; ++k) {
        const int a = cnt[k * 2], b2 = cnt[k * 2 + 1];
        if (k < strip) { nm += a; nsn += b2; }
        tot_m += a; tot_s += b2;
    }
    double acc[7] = {0, 0, 0, 0, 0, 0, 0};     // model xyz, scene-near-anchor xyz, its count
    double ext[12] = {1e300, 1e300, 1e300, -1e300, -1e300, -1e300, 1e300, 1e300, 1e300, -1e300, -1e300, -1e300};   // min, max of the strip's model points, of its scene points
    for (int base = p_lo; base < p_hi; base += kPtsWG) {
        const int p = base + tid;
        bool is_m = false, is_s = false;
        double mx = 0, my = 0, mz = 0, sx = 0, sy = 0, sz = 0;
        if (p < p_hi) {
            const int r = p / bw, c = p - r * bw;
            const int mr = r + by0, mc = c + bx0;
            const int sr = max(r + I.dy - kDilate, 0), sc = max(c + I.dx - kDilate, 0);
            const uint16_t md = model[(size_t)mr * W + mc];
            const uint16_t sd = scene[(size_t)sr * W + sc];
            if (md > 0) {
                is_m = true;
                mz = md / 1000.0;
                // (int - float) / float evaluated in float, then * double (LL.cpp:79-80)
                mx = (double)__fdiv_rn(__fsub_rn((float)mc, I.mK[2]), I.mK[0]) * mz;
                my = (double)__fdiv_rn(__fsub_rn((float)mr, I.mK[5]), I.mK[4]) * mz;
                acc[0] += mx; acc[1] += my; acc[2] += mz;
                ext[0] = fmin(ext[0], mx); ext[1] = fmin(ext[1], my); ext[2] = fmin(ext[2], mz);
                ext[3] = fmax(ext[3], mx); ext[4] = fmax(ext[4], my); ext[5] = fmax(ext[5], mz);
            }
            if (sd > 0) {
                bool in_mask = md > 0;
                if (!in_mask) {                                   // dilate(modelDepth > 0, 9x9) at (mr, mc)
                    const int ya = max(mr - kDilate, 0), yb = min(mr + kDilate, H - 1);
                    const int xa = max(mc - kDilate, 0), xb = min(mc + kDilate, W - 1);
                    for (int yy = ya; yy <= yb && !in_mask; ++yy)
                        for (int xx = xa; xx <= xb; ++xx)
                            if (model[(size_t)yy * W + xx]) { in_mask = true; break; }
                }
                if (in_mask) {
                    is_s = true;
                    sz = sd / 1000.0;
                    sx = (double)__fdiv_rn(__fsub_rn((float)sc, B.sK[2]), B.sK[0]) * sz;
                    sy = (double)__fdiv_rn(__fsub_rn((float)sr, B.sK[5]), B.sK[4]) * sz;
                    if (fabs(sz - anchor) < 0.4 && md > 0) { acc[3] += sx; acc[4] += sy; acc[5] += sz; acc[6] += 1.0; }
                    ext[6] = fmin(ext[6], sx); ext[7] = fmin(ext[7], sy); ext[8] = fmin(ext[8], sz);
                    ext[9] = fmax(ext[9], sx); ext[10] = fmax(ext[10], sy); ext[11] = fmax(ext[11], sz);
                }
            }
        }
        int tot;
        const int pm = nm + block_scan_flag(is_m, s_wave, tot);
        nm += tot;
        if (is_m) { mp[3 * (size_t)pm] = mx; mp[3 * (size_t)pm + 1] = my; mp[3 * (size_t)pm + 2] = mz; }
        if (keep_scene) {
            const int ps = nsn + block_scan_flag(is_s, s_wave, tot);
            nsn += tot;
            if (is_s) { sp[3 * (size_t)ps] = sx; sp[3 * (size_t)ps + 1] = sy; sp[3 * (size_t)ps + 2] = sz; }
        }
    }
#pragma unroll
    for (int k = 0; k < 7; ++k) {
        const double v = wave_sum(acc[k]);
        if ((tid & 63) == 0) s_red[tid >> 6][k] = v;
    }
#pragma unroll
    for (int k = 0; k < 12; ++k) {
        double v = ext[k];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v = (k % 6) < 3 ? fmin(v, shfl_xor_d(v, o)) : fmax(v, shfl_xor_d(v, o));
        if ((tid & 63) == 0) s_ext[tid >> 6][k] = v;
    }
    __syncthreads();
    if (tid < 7) {
        double v = 0;
        for (int w = 0; w < kPtsWG / 64; ++w) v += s_red[w][tid];
        B.strip_sum[((size_t)h * kIcpStrips + strip) * 8 + tid] = v;
    }
    if (tid >= 64 && tid < 76) {                                   // the strip's extents: what k_icp_voxel_keys takes the voxel origin from (min / max are exact in any order)
        const int k = tid - 64;
        double v = s_ext[0][k];
        for (int w = 1; w < kPtsWG / 64; ++w) v = (k % 6) < 3 ? fmin(v, s_ext[w][k]) : fmax(v, s_ext[w][k]);
        B.strip_mm[((size_t)h * kIcpStrips + strip) * 12 + k] = v;
    }
    if (strip == 0 && tid == 0) { S.n_model = tot_m; S.n_scene = keep_scene ? tot_s : 0; }
}

// ---------------------------------------------------------------------------------------------
// k_icp_points_fused: both passes of k_icp_points in one launch.  A strip classifies its pixels once (two bits per pixel and thread in
// registers: the 9x9 dilation test is the expensive part), publishes its two counts as one agent-scope word and waits for the strips
// before it (lower block indices, dispatched before it) — their sum is where its points start — then writes.  The last strip, which has
// seen every count, sets n_model / n_scene.  k_icp_bbox clears the words (B.strip_pub) for the next run.
// ---------------------------------------------------------------------------------------------
constexpr long long kStripTimeout = 1000ll * 100000;            // wall_clock64 ticks: 1 s (then: status kIcpStalled)

__global__ void __launch_bounds__(kPtsWG)
k_icp_points_fused(IcpBuffers B, int W, int H, int flags) {
    __shared__ int s_wave[8];
    __shared__ double s_red[kPtsWG / 64][7];
    __shared__ double s_ext[kPtsWG / 64][12];
    __shared__ int s_tot[2];
    __shared__ long long s_before[2];
    const int h = blockIdx.y, strip = blockIdx.x, tid = threadIdx.x;
    IcpState& S = B.st[h];
    if (S.status != 0) return;
    const IcpIn I = B.in[h];
    // the box of the model image: worked out when the image was uploaded (k_icp_model_boxes), else by k_icp_bbox of this run
    const int* known = B.model_bbox + (size_t)I.model_slot * 8;
    const bool boxed = known[4] == 1;
    const int x0 = boxed ? INT_MAX - known[0] : S.bbox[0], y0 = boxed ? INT_MAX - known[1] : S.bbox[1], x1 = boxed ? known[2] - 1 : S.bbox[2],
              y1 = boxed ? known[3] - 1 : S.bbox[3];
    if (strip == 0 && tid == 0 && !boxed) {                        // (k_icp_bbox is through: known from now on)
        int* kn = B.model_bbox + (size_t)I.model_slot * 8;
        kn[0] = INT_MAX - x0; kn[1] = INT_MAX - y0; kn[2] = x1 + 1; kn[3] = y1 + 1; __threadfence(); kn[4] = 1;
    }
    if (strip == 0 && tid < 2 * kIcpSortGroups) B.sort_look[(size_t)h * 2 * kIcpSortGroups + tid] = 0;   // (k_icp_voxel_wide: voxel counts of the groups, not yet known)
    if (x1 < 0) {
        if (strip == 0 && tid == 0) { S.status = kIcpEmptyModel; S.n_model = 0; S.n_scene = 0; }
        return;
    }
    const int bx0 = max(x0 - kDilate, 0), by0 = max(y0 - kDilate, 0);
    const int bx1 = min(x1 + kDilate, W - 1), by1 = min(y1 + kDilate, H - 1);
    const int bw = bx1 - bx0 + 1, bh = by1 - by0 + 1;
    if (I.dx + bw >= W || I.dy + bh >= H) {                       // LL.cpp:52-55
        if (strip == 0 && tid == 0) { S.status = kIcpOutOfFrame; S.n_model = 0; S.n_scene = 0; }
        return;
    }
    const uint16_t* model = B.models + (size_t)I.model_slot * W * H;
    const uint16_t* scene = B.scene;
    const int r_lo = (int)((long long)bh * strip / kIcpStrips), r_hi = (int)((long long)bh * (strip + 1) / kIcpStrips);
    const int p_lo = r_lo * bw, p_hi = r_hi * bw;
    const bool keep_scene = (flags & 1) != 0;
    // model point where modelDepth > 0; scene point where sceneDepth > 0 under the dilated mask (LL.cpp:43-50, 66-90)
    auto classify = [&](const int p, bool& is_m, bool& is_s) {
        const int r = p / bw, c = p - r * bw;
        const int mr = r + by0, mc = c + bx0;
        const int sr = max(r + I.dy - kDilate, 0), sc = max(c + I.dx - kDilate, 0);
        const uint16_t md = model[(size_t)mr * W + mc];
        const uint16_t sd = scene[(size_t)sr * W + sc];
        is_m = md > 0;
        is_s = false;
        if (sd > 0) {
            bool in_mask = md > 0;
            if (!in_mask) {                                       // dilate(modelDepth > 0, 9x9) at (mr, mc)
                const int ya = max(mr - kDilate, 0), yb = min(mr + kDilate, H - 1);
                const int xa = max(mc - kDilate, 0), xb = min(mc + kDilate, W - 1);
                for (int yy = ya; yy <= yb && !in_mask; ++yy)
                    for (int xx = xa; xx <= xb; ++xx)
                        if (model[(size_t)yy * W + xx]) { in_mask = true; break; }
            }
            is_s = in_mask;
        }
    };
    unsigned long long fm = 0, fs = 0;                             // the classes of this thread's first 64 pixels
    int cm = 0, cs = 0;
    {
        int it = 0;
        for (int p = p_lo + tid; p < p_hi; p += kPtsWG, ++it) {
            bool is_m, is_s;
            classify(p, is_m, is_s);
            cm += is_m ? 1 : 0; cs += (is_s && keep_scene) ? 1 : 0;
            if (it < 64) { fm |= (unsigned long long)is_m << it; fs |= (unsigned long long)is_s << it; }
        }
    }
    if (tid < 2) s_tot[tid] = 0;
    __syncthreads();
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { cm += __shfl_xor(cm, o, 64); cs += __shfl_xor(cs, o, 64); }
    if ((tid & 63) == 0) { atomicAdd(&s_tot[0], cm); atomicAdd(&s_tot[1], cs); }
    __syncthreads();
    unsigned long long* pub = B.strip_pub + (size_t)h * kIcpStrips;
    if (tid == 0) __hip_atomic_store(pub + strip, (1ull << 63) | ((unsigned long long)s_tot[1] << 32) | (unsigned long long)s_tot[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (tid < 64) {
        unsigned long long v = 1ull << 63;
        if (tid < strip) {
            const long long t0 = wall_clock64();
            do { v = __hip_atomic_load(pub + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); } while (!(v >> 63) && wall_clock64() - t0 < kStripTimeout);
        }
        const bool lost = __ballot(!(v >> 63)) != 0ull;
        long long bm = (long long)(v & 0xFFFFFFFFull), bs = (long long)((v >> 32) & 0x7FFFFFFFull);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { bm += __shfl_xor(bm, o, 64); bs += __shfl_xor(bs, o, 64); }
        if (tid == 0) { s_before[0] = lost ? -1 : bm; s_before[1] = bs; }
    }
    __syncthreads();
    if (s_before[0] < 0) {                                         // a strip before this one never came
        if (tid == 0) S.status = kIcpStalled;
        return;
    }
    int nm = (int)s_before[0], nsn = (int)s_before[1];

    const double anchor = model[(size_t)(H / 2) * W + W / 2] / 1000.0;   // LL.cpp:62
    double* mp = B.model_pts + (size_t)h * B.cap * 3;
    double* sp = B.scene_pts + (size_t)h * B.cap * 3;
    double acc[7] = {0, 0, 0, 0, 0, 0, 0};     // model xyz, scene-near-anchor xyz, its count
    double ext[12] = {1e300, 1e300, 1e300, -1e300, -1e300, -1e300, 1e300, 1e300, 1e300, -1e300, -1e300, -1e300};   // min, max of the strip's model points, of its scene points
    int it = 0;
    for (int base = p_lo; base < p_hi; base += kPtsWG, ++it) {
        const int p = base + tid;
        bool is_m = false, is_s = false;
        double mx = 0, my = 0, mz = 0, sx = 0, sy = 0, sz = 0;
        if (p < p_hi) {
            if (it < 64) { is_m = (fm >> it) & 1ull; is_s = (fs >> it) & 1ull; }
            else classify(p, is_m, is_s);
            const int r = p / bw, c = p - r * bw;
            const int mr = r + by0, mc = c + bx0;
            const int sr = max(r + I.dy - kDilate, 0), sc = max(c + I.dx - kDilate, 0);
            if (is_m) {
                const uint16_t md = model[(size_t)mr * W + mc];
                mz = md / 1000.0;
                // (int - float) / float evaluated in float, then * double (LL.cpp:79-80)
                mx = (double)__fdiv_rn(__fsub_rn((float)mc, I.mK[2]), I.mK[0]) * mz;
                my = (double)__fdiv_rn(__fsub_rn((float)mr, I.mK[5]), I.mK[4]) * mz;
                acc[0] += mx; acc[1] += my; acc[2] += mz;
                ext[0] = fmin(ext[0], mx); ext[1] = fmin(ext[1], my); ext[2] = fmin(ext[2], mz);
                ext[3] = fmax(ext[3], mx); ext[4] = fmax(ext[4], my); ext[5] = fmax(ext[5], mz);
            }
            if (is_s) {
                const uint16_t sd = scene[(size_t)sr * W + sc];
                sz = sd / 1000.0;
                sx = (double)__fdiv_rn(__fsub_rn((float)sc, B.sK[2]), B.sK[0]) * sz;
                sy = (double)__fdiv_rn(__fsub_rn((float)sr, B.sK[5]), B.sK[4]) * sz;
                if (fabs(sz - anchor) < 0.4 && is_m) { acc[3] += sx; acc[4] += sy; acc[5] += sz; acc[6] += 1.0; }
                ext[6] = fmin(ext[6], sx); ext[7] = fmin(ext[7], sy); ext[8] = fmin(ext[8], sz);
                ext[9] = fmax(ext[9], sx); ext[10] = fmax(ext[10], sy); ext[11] = fmax(ext[11], sz);
            }
        }
        int tot;
        const int pm = nm + block_scan_flag(is_m, s_wave, tot);
        nm += tot;
        if (is_m) { mp[3 * (size_t)pm] = mx; mp[3 * (size_t)pm + 1] = my; mp[3 * (size_t)pm + 2] = mz; }
        if (keep_scene) {
            const int ps = nsn + block_scan_flag(is_s, s_wave, tot);
            nsn += tot;
            if (is_s) { sp[3 * (size_t)ps] = sx; sp[3 * (size_t)ps + 1] = sy; sp[3 * (size_t)ps + 2] = sz; }
        }
    }
#pragma unroll
    for (int k = 0; k < 7; ++k) {
        const double v = wave_sum(acc[k]);
        if ((tid & 63) == 0) s_red[tid >> 6][k] = v;
    }
#pragma unroll
    for (int k = 0; k < 12; ++k) {
        double v = ext[k];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v = (k % 6) < 3 ? fmin(v, shfl_xor_d(v, o)) : fmax(v, shfl_xor_d(v, o));
        if ((tid & 63) == 0) s_ext[tid >> 6][k] = v;
    }
    __syncthreads();
    if (tid < 7) {
        double v = 0;
        for (int w = 0; w < kPtsWG / 64; ++w) v += s_red[w][tid];
        B.strip_sum[((size_t)h * kIcpStrips + strip) * 8 + tid] = v;
    }
    if (tid >= 64 && tid < 76) {
        const int k = tid - 64;
        double v = s_ext[0][k];
        for (int w = 1; w < kPtsWG / 64; ++w) v = (k % 6) < 3 ? fmin(v, s_ext[w][k]) : fmax(v, s_ext[w][k]);
        B.strip_mm[((size_t)h * kIcpStrips + strip) * 12 + k] = v;
    }
    if (strip == kIcpStrips - 1 && tid == 0) { S.n_model = nm; S.n_scene = keep_scene ? nsn : 0; }
}

// ---------------------------------------------------------------------------------------------
// k_icp_voxel: open3d PointCloud::VoxelDownSample — mean per voxel, output in ascending
// (ix,iy,iz) order, the points of a voxel summed in input order.  key = voxel index | point index
// with just enough bits per field; one workgroup sorts its cloud (LDS up to 16k points).
// blockIdx.y: 0 = model cloud -> src (and tgt in verbatim mode, LL.cpp:109), 1 = scene cloud -> tgt.
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(kWG)
k_icp_voxel(IcpBuffers B, int flags, double voxel) {
    __shared__ __attribute__((aligned(16))) unsigned char s_raw[kRadixLdsBytes];   // radix layout, or 128 KiB of 64-bit keys (bitonic)
    unsigned long long* s_keys = reinterpret_cast<unsigned long long*>(s_raw);
    __shared__ int s_wave[32];
    __shared__ double s_part[16 * 6];
    __shared__ double s_mm[6];
    const int h = blockIdx.x, which = blockIdx.y, tid = threadIdx.x;
    IcpState& S = B.st[h];
    const bool scene_mode = (flags & 1) != 0;
    if (S.vox_done[which] == kIcpSortGroups) return;               // k_icp_voxel_wide did this cloud
    if (S.status != 0) {
        if (tid == 0) { if (which == 0) { S.n_src = 0; if (!scene_mode) S.n_tgt = 0; } else S.n_tgt = 0; }
        return;
    }
    const int n = which ? S.n_scene : S.n_model;
    const double* pts = (which ? B.scene_pts : B.model_pts) + (size_t)h * B.cap * 3;
    double* out = (which ? B.tgt : B.src) + (size_t)h * B.cap * 3;
    if (n == 0) {
        if (tid == 0) { if (which == 0) { S.n_src = 0; if (!scene_mode) S.n_tgt = 0; } else S.n_tgt = 0; }
        return;
    }
    const long long v_t0 = (long long)__builtin_amdgcn_s_memtime();
    double mn[3] = {1e300, 1e300, 1e300}, mx[3] = {-1e300, -1e300, -1e300};
    for (int i = tid; i < n; i += kWG) {
#pragma unroll
        for (int k = 0; k < 3; ++k) { const double v = pts[3 * (size_t)i + k]; mn[k] = fmin(mn[k], v); mx[k] = fmax(mx[k], v); }
    }
    block_minmax<3>(mn, mx, s_part, s_mm);
    const double mnx = s_mm[0] - voxel * 0.5, mny = s_mm[1] - voxel * 0.5, mnz = s_mm[2] - voxel * 0.5;
    const double fx = floor(__ddiv_rn(s_mm[3] - mnx, voxel)), fy = floor(__ddiv_rn(s_mm[4] - mny, voxel)),
                 fz = floor(__ddiv_rn(s_mm[5] - mnz, voxel));
    const bool finite = fx >= 0 && fx < 4e18 && fy >= 0 && fy < 4e18 && fz >= 0 && fz < 4e18;
    const int bx = finite ? bits_for((long long)fx) : 64, by = finite ? bits_for((long long)fy) : 64,
              bz = finite ? bits_for((long long)fz) : 64, bi = bits_for(n - 1);
    if (bx + by + bz + bi > 64) {
        if (tid == 0) S.status = 3;
        return;
    }
    const long long v_t1 = (long long)__builtin_amdgcn_s_memtime();
    const int npad = next_pow2(n < 2 ? 2 : n);
    const bool in_lds = npad <= kSortLds;
    const bool radix = n <= kRadixBig && bx + by + bz <= 32;       // voxel index in 32 bits: stable radix sort of the point indices
    const bool big = n > kSortLds;                                 // ... whose keys then stay in HBM
    const RadixView s_rx = radix_view(s_raw, big);
    unsigned long long* gk = B.keys + ((size_t)h * 2 + which) * B.cap2;
    unsigned int* gk32 = reinterpret_cast<unsigned int*>(gk);
    for (int i = tid; i < (radix ? n : npad); i += kWG) {
        unsigned long long key = ~0ull;
        if (i < n) {
            const unsigned long long ix = (unsigned long long)(long long)floor(__ddiv_rn(pts[3 * (size_t)i] - mnx, voxel));
            const unsigned long long iy = (unsigned long long)(long long)floor(__ddiv_rn(pts[3 * (size_t)i + 1] - mny, voxel));
            const unsigned long long iz = (unsigned long long)(long long)floor(__ddiv_rn(pts[3 * (size_t)i + 2] - mnz, voxel));
            key = (((ix << by) | iy) << bz | iz);
            if (!radix) key = (key << bi) | (unsigned long long)i;
        }
        if (radix) { if (big) gk32[i] = (unsigned int)key; else s_rx.key_lds[i] = (unsigned int)key; }
        else if (in_lds) s_keys[i] = key; else gk[i] = key;
    }
    __syncthreads();
    const long long v_t2 = (long long)__builtin_amdgcn_s_memtime();
    const unsigned short* order = nullptr;
    if (radix && big) order = wg_radix_sort(s_rx, n, bx + by + bz, s_wave, [&](unsigned int id) { return gk32[id]; });
    else if (radix) order = wg_radix_sort(s_rx, n, bx + by + bz, s_wave, [&](unsigned int id) { return s_rx.key_lds[id]; });
    else if (in_lds) bitonic_sort(s_keys, npad);
    else bitonic_sort_hybrid(gk, npad, s_keys);
    // sorted position -> (voxel index << bi) | point index, whichever way the list was sorted
    auto key_at = [&](int i) -> unsigned long long {
        if (radix) { const unsigned int id = order[i]; return ((unsigned long long)(big ? gk32[id] : s_rx.key_lds[id]) << bi) | id; }
        return in_lds ? s_keys[i] : gk[i];
    };
    const long long v_t3 = (long long)__builtin_amdgcn_s_memtime();
    const unsigned long long imask = (1ull << bi) - 1ull;
    int nout = 0;
    if (radix) {
        // the points of 1024 sorted positions are gathered into LDS side by side (the index buffer the sort left free), then
        // the first thread of every voxel adds its points up in input order — out of LDS, not one dependent HBM gather each
        unsigned short* spare = order == s_rx.idx[0] ? s_rx.idx[1] : s_rx.idx[0];
        double* st = reinterpret_cast<double*>(spare);                       // [kWG][3]
        unsigned int* sv = reinterpret_cast<unsigned int*>(st + 3 * kWG);    // [kWG] voxel index per position
        auto vox_of = [&](int i) -> unsigned int { const unsigned int id = order[i]; return big ? gk32[id] : s_rx.key_lds[id]; };
        for (int base = 0; base < n; base += kWG) {
            const int i = base + tid;
            unsigned int v = 0;
            if (i < n) {
                const unsigned int id = order[i];
                v = big ? gk32[id] : s_rx.key_lds[id];
                st[3 * tid] = pts[3 * (size_t)id]; st[3 * tid + 1] = pts[3 * (size_t)id + 1]; st[3 * tid + 2] = pts[3 * (size_t)id + 2];
                sv[tid] = v;
            }
            __syncthreads();
            const bool head = i < n && (i == 0 || (tid > 0 ? sv[tid - 1] : vox_of(i - 1)) != v);
            int tot;
            const int pos = nout + block_scan_flag(head, s_wave, tot);
            nout += tot;
            if (head) {
                double sx = 0, sy = 0, sz = 0;
                int cnt = 0;
                for (int j = i; j < n; ++j) {
                    const int t = j - base;
                    if (t < kWG) {
                        if (sv[t] != v) break;
                        sx += st[3 * t]; sy += st[3 * t + 1]; sz += st[3 * t + 2];
                    } else {                                             // the voxel runs on into the next 1024 positions
                        if (vox_of(j) != v) break;
                        const size_t idx = order[j];
                        sx += pts[3 * idx]; sy += pts[3 * idx + 1]; sz += pts[3 * idx + 2];
                    }
                    ++cnt;
                }
                const double c = (double)cnt;
                out[3 * (size_t)pos] = sx / c; out[3 * (size_t)pos + 1] = sy / c; out[3 * (size_t)pos + 2] = sz / c;
            }
            __syncthreads();
        }
    } else
    for (int base = 0; base < n; base += kWG) {
        const int i = base + tid;
        bool head = false;
        unsigned long long vox = 0;
        if (i < n) {
            const unsigned long long k = key_at(i);
            vox = k >> bi;
            head = i == 0 || (key_at(i - 1) >> bi) != vox;
        }
        int tot;
        const int pos = nout + block_scan_flag(head, s_wave, tot);
        nout += tot;
        if (head) {
            double sx = 0, sy = 0, sz = 0;
            int cnt = 0, j = i;
            for (;;) {
                const unsigned long long k = key_at(j);
                if ((k >> bi) != vox) break;
                const size_t idx = (size_t)(k & imask);
                sx += pts[3 * idx]; sy += pts[3 * idx + 1]; sz += pts[3 * idx + 2];
                ++cnt; ++j;
                if (j >= n) break;
            }
            const double c = (double)cnt;
            out[3 * (size_t)pos] = sx / c; out[3 * (size_t)pos + 1] = sy / c; out[3 * (size_t)pos + 2] = sz / c;
        }
    }
    if (tid == 0) {
        if (which == 0) { S.n_src = nout; if (!scene_mode) S.n_tgt = nout; }
        else S.n_tgt = nout;
        if (which == 0) {                                          // diagnostics (model cloud): cycles for extent, keys, sort, voxel means
            const long long v_t4 = (long long)__builtin_amdgcn_s_memtime();
            S.vox_clk[0] = v_t1 - v_t0; S.vox_clk[1] = v_t2 - v_t1; S.vox_clk[2] = v_t3 - v_t2; S.vox_clk[3] = v_t4 - v_t3;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// k_icp_grid: the target cloud binned into xy columns (cell edge >= 5 mm, <= 64 x 64 columns) and, inside
// a column, ordered by quantised depth (>= 1 mm steps): points sorted by (x column, y column, z step,
// original index).  A search visits one run per column and finds the depth range it needs by bisection,
// so the table stays a few thousand entries however deep the cloud is (a scene cloud carries background
// far behind the object); an x slab of columns is one contiguous range of cells and of points (what a
// source slice stages in LDS).  cell_start[c] = first sorted position of column c.
// ---------------------------------------------------------------------------------------------
constexpr int kZBits = 20;         // quantised-depth bits of the grid sort key

static __device__ __forceinline__ int zq_of(double z, double minz, double inv_z, int zq_max) {
    const double f = floor((z - minz) * inv_z);
    return f >= 0.0 ? (f < (double)zq_max ? (int)f : zq_max) : 0;      // NaN -> 0
}

__global__ void __launch_bounds__(kWG)
k_icp_grid(IcpBuffers B, int flags) {
    __shared__ __attribute__((aligned(16))) unsigned char s_raw[kRadixLdsBytes];   // radix layout, or 128 KiB of 64-bit keys (bitonic)
    unsigned long long* s_keys = reinterpret_cast<unsigned long long*>(s_raw);
    __shared__ int s_wave[32];
    __shared__ double s_part[16 * 6];
    __shared__ double s_mm[6];
    const int h = blockIdx.x, tid = threadIdx.x;
    IcpState& S = B.st[h];
    if (S.grid_done == kIcpSortGroups) return;                     // k_icp_grid_wide did this cloud
    if (tid == 0 && S.status == 0) {                               // init_guess: centroid difference (LL.cpp:91-104), strips added in order
        double t[7] = {0, 0, 0, 0, 0, 0, 0};
        for (int k = 0; k < kIcpStrips; ++k)
            for (int q = 0; q < 7; ++q) t[q] += B.strip_sum[((size_t)h * kIcpStrips + k) * 8 + q];
        const double n = (double)S.n_model;
        S.init[0] = t[3] / t[6] - t[0] / n;      // NaN when no scene point is near the anchor, as the reference
        S.init[1] = t[4] / t[6] - t[1] / n;
        S.init[2] = t[5] / t[6] - t[2] / n;
    }
    int* cs = B.cell_start + (size_t)h * kIcpCells;
    const int nt = S.status == 0 ? S.n_tgt : 0;
    if (nt == 0 || nt >= (1 << kIdxBits)) {
        if (tid == 0) {
            if (nt > 0) { S.status = 3; S.n_tgt = 0; }
            S.gx = 1; S.gy = 1; S.zq_max = 0; S.gminx = 0; S.gminy = 0; S.gminz = 0; S.cell = kCellMin; S.inv_cell = 1.0 / kCellMin; S.inv_z = 1e4;
            cs[0] = 0; cs[1] = 0;
        }
        return;
    }
    const double* T = ((flags & 1) ? B.tgt : B.src) + (size_t)h * B.cap * 3;
    double mn[3] = {1e300, 1e300, 1e300}, mx[3] = {-1e300, -1e300, -1e300};
    for (int i = tid; i < nt; i += kWG) {
#pragma unroll
        for (int k = 0; k < 3; ++k) { const double v = T[3 * (size_t)i + k]; mn[k] = fmin(mn[k], v); mx[k] = fmax(mx[k], v); }
    }
    block_minmax<3>(mn, mx, s_part, s_mm);
    const double minx = s_mm[0], miny = s_mm[1], minz = s_mm[2];
    const double ext = fmax(s_mm[3] - minx, s_mm[4] - miny), extz = s_mm[5] - minz;
    double cell = ext / (double)kIcpGrid;
    if (!(cell > kCellMin)) cell = kCellMin;                       // also catches NaN
    double zres = extz / (double)((1 << kZBits) - 1);
    if (!(zres > 1e-3)) zres = 1e-3;                               // 1 mm steps: a metre of depth is 10 bits, so the sort key (column, step) is 3 radix passes, not 4
    const double inv = 1.0 / cell, inv_z = 1.0 / zres;
    if (!(ext < 1e30) || !(extz < 1e30)) {                         // non-finite coordinates
        if (tid == 0) { S.status = 3; S.n_tgt = 0; S.gx = 1; S.gy = 1; S.zq_max = 0; cs[0] = 0; cs[1] = 0; }
        return;
    }
    const int gx = grid_coord(s_mm[3], minx, inv, kIcpGrid) + 1, gy = grid_coord(s_mm[4], miny, inv, kIcpGrid) + 1;
    const int zq_max = zq_of(s_mm[5], minz, inv_z, (1 << kZBits) - 1);
    const int npad = next_pow2(nt < 2 ? 2 : nt);
    const bool radix = nt <= kRadixBig;                            // (column, depth step) is 32 bits: stable radix sort of the point indices
    const bool big = nt > kSortLds;                                // ... whose keys then stay in HBM
    const RadixView s_rx = radix_view(s_raw, big);
    const int cell_bits = bits_for((long long)gx * gy - 1), z_bits = bits_for(zq_max);
    unsigned long long* gk = B.keys + (size_t)h * 2 * B.cap2;
    unsigned int* gk32 = reinterpret_cast<unsigned int*>(gk);
    for (int i = tid; i < (radix ? nt : npad); i += kWG) {
        unsigned long long key = ~0ull;
        if (i < nt) {
            const int cx = grid_coord(T[3 * (size_t)i], minx, inv, gx), cy = grid_coord(T[3 * (size_t)i + 1], miny, inv, gy);
            const int zq = zq_of(T[3 * (size_t)i + 2], minz, inv_z, zq_max);
            if (radix) key = ((unsigned long long)(cx * gy + cy) << z_bits) | (unsigned long long)zq;
            else key = ((((unsigned long long)(cx * gy + cy) << kZBits) | (unsigned long long)zq) << kIdxBits) | (unsigned long long)i;
        }
        if (radix) { if (big) gk32[i] = (unsigned int)key; else s_rx.key_lds[i] = (unsigned int)key; }
        else gk[i] = key;
    }
    __syncthreads();
    const unsigned short* order = nullptr;
    if (radix && big) order = wg_radix_sort(s_rx, nt, cell_bits + z_bits, s_wave, [&](unsigned int id) { return gk32[id]; });
    else if (radix) order = wg_radix_sort(s_rx, nt, cell_bits + z_bits, s_wave, [&](unsigned int id) { return s_rx.key_lds[id]; });
    else bitonic_sort_hybrid(gk, npad, s_keys);
    // sorted position -> ((column << kZBits | depth step) << kIdxBits) | point index, whichever way the list was sorted
    auto key_at = [&](int i) -> unsigned long long {
        if (radix) {
            const unsigned int id = order[i], k32 = big ? gk32[id] : s_rx.key_lds[id];
            return ((((unsigned long long)(k32 >> z_bits) << kZBits) | (unsigned long long)(k32 & ((1u << z_bits) - 1u))) << kIdxBits) | id;
        }
        return gk[i];
    };
    double* Ts = B.tgt_sorted + (size_t)h * B.cap * 3;
    int* orig = B.tgt_orig + (size_t)h * B.cap;
    TgtRec* rec = B.tgt_rec + (size_t)h * B.cap;
    unsigned short* cs16 = B.cell_start16 + (size_t)h * kIcpCells16;
    for (int p = tid; p < nt; p += kWG) {
        const unsigned long long k = key_at(p);
        const size_t i = (size_t)(k & ((1ull << kIdxBits) - 1ull));
        Ts[3 * (size_t)p] = T[3 * i]; Ts[3 * (size_t)p + 1] = T[3 * i + 1]; Ts[3 * (size_t)p + 2] = T[3 * i + 2];
        orig[p] = (int)i;
        TgtRec r;
        r.x = T[3 * i]; r.y = T[3 * i + 1]; r.z = T[3 * i + 2]; r.orig = (int)i;
        r.zq = (int)((k >> kIdxBits) & ((1ull << kZBits) - 1ull));
        rec[p] = r;
    }
    const int ncell = gx * gy;
    for (int c = tid; c <= ncell; c += kWG) {                       // lower_bound of the column's smallest key
        const unsigned long long want = (unsigned long long)c << (kZBits + kIdxBits);
        int lo = 0, hi = nt;
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            const unsigned long long k = key_at(mid);
            if (k < want) lo = mid + 1; else hi = mid;
        }
        cs[c] = lo;
        cs16[c] = (unsigned short)lo;
    }
    if (tid == 0) {
        S.gx = gx; S.gy = gy; S.zq_max = zq_max; S.gminx = minx; S.gminy = miny; S.gminz = minz; S.cell = cell; S.inv_cell = inv;
        S.inv_z = inv_z;
    }
}

// ---------------------------------------------------------------------------------------------
// The two sorts above, spread over the chip (round 6).  One workgroup sorting a cloud of 12-17k points took 120-200 us with 15 of 16 CUs
// of the hypothesis' share idle.  k_icp_voxel_wide / k_icp_grid_wide give a cloud kIcpSortGroups workgroups, each a contiguous range of
// the LEADING coordinate of the sort key (voxel index ix, grid column cx: equal shares of its span):
//   a group walks the x coordinates of the whole cloud — every wave a contiguous piece: count, prefix over the waves, second walk — and
//   takes the points of its range in input order with the key relative to its range; the cloud's extent comes from the strips of
//   k_icp_points, not from a scan;
//   it orders them in LDS (group_sort: the keys are (column, depth) with a handful of points per column: count per column, prefix, rank
//   inside the column; the radix passes above only when a column is crowded);
//   the grid group then writes its part of the sorted cloud (its first position = the points below its range, counted on the way) and the
//   starts of its columns; the voxel group needs the number of voxels of the groups before it, which every group publishes as soon as its
//   order stands (one agent-scope word each; a group waits only for lower block indices, which were dispatched before it).
// The voxel means are those of k_icp_voxel bit for bit (same keys, stable order, the points of a voxel added up in input order); the grid
// spans the extent of the cloud the target was down-sampled FROM (a bound of the means' extent that costs no scan).  The one-workgroup
// kernels stay behind them for the clouds they leave: voxel index beyond 32 bits, a group of more than kGroupCap points, empty or rejected
// hypotheses — IcpState::vox_done / grid_done say which.
// ---------------------------------------------------------------------------------------------
constexpr int kGroupCap = 8192;     // points a group sorts (LDS: keys 32 KiB, point indices 32 KiB, two index lists 32 KiB, digit counts 16 KiB)
constexpr unsigned int kLookFail = 0xFFFFFFFFu;
constexpr long long kLookTimeout = 100ll * 100000;   // wall_clock64 ticks: 100 ms

// Extent of one of the two back-projected clouds of a hypothesis from its strips; all threads of wave 0 call it, result in s_mm[6] (after a barrier).
static __device__ __forceinline__ void strips_extent(const double* __restrict__ strip_mm /*[kIcpStrips][12]*/, const int which, double* s_mm) {
    const int lane = threadIdx.x & 63;
    double mn[3] = {1e300, 1e300, 1e300}, mx[3] = {-1e300, -1e300, -1e300};
    if (lane < kIcpStrips) {
        const double* p = strip_mm + (size_t)lane * 12 + which * 6;
#pragma unroll
        for (int k = 0; k < 3; ++k) { mn[k] = p[k]; mx[k] = p[3 + k]; }
    }
#pragma unroll
    for (int k = 0; k < 3; ++k)
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { mn[k] = fmin(mn[k], shfl_xor_d(mn[k], o)); mx[k] = fmax(mx[k], shfl_xor_d(mx[k], o)); }
    if (lane == 0) { s_mm[0] = mn[0]; s_mm[1] = mn[1]; s_mm[2] = mn[2]; s_mm[3] = mx[0]; s_mm[4] = mx[1]; s_mm[5] = mx[2]; }
}

// floor(a / v) as the one-workgroup kernels compute it (IEEE division, then floor), without the division where the answer cannot depend on it:
// a * (1 / v) is within 2^-52 of a / v relatively, so below 2^32 and further than 1e-6 from an integer both floor alike.
static __device__ __forceinline__ long long floor_quotient(const double a, const double v, const double inv_v) {
    const double q = a * inv_v, f = floor(q), d = q - f;
    if (!(d > 1e-6 && d < 1.0 - 1e-6)) return (long long)floor(__ddiv_rn(a, v));
    return (long long)f;
}

// A wave's piece of a cloud of n points in group_collect: positions [i0, i1), a multiple of 64 long, 16 pieces cover n.
static __device__ __forceinline__ void collect_piece(const int n, int& i0, int& i1) {
    const int piece = (((n + 15) >> 4) + 63) & ~63;
    i0 = (int)(threadIdx.x >> 6) * piece; i1 = min(n, i0 + piece);
}

// The points whose leading coordinate lead(x) lies in [lo, hi), in input order: key_of(lead, y, z) -> s_key[j], i -> s_gid[j].  Returns their
// number (nothing is written when it exceeds kGroupCap) and the number of points below lo.  Every wave walks the x coordinates of its piece
// twice (count, prefix over the waves, take), four rounds of 64 points at a time so that their loads are in flight together; x0 = the x
// coordinates of the first four rounds, which the caller loaded before it knew the cloud's extent; then the keys of the taken points, spread
// over all threads.  1024 threads; s_wave: 32 ints.
template <typename LoadF, typename LeadF, typename KeyF>
static __device__ __forceinline__ int group_collect(const int n, const long long lo, const long long hi, unsigned int* s_key, unsigned int* s_gid, int* s_wave,
                                                    int& below_out, const double (&x0)[4], LoadF&& load, LeadF&& lead, KeyF&& key_of, long long* tick = nullptr) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int i0, i1;
    collect_piece(n, i0, i1);
    int mine = 0, below = 0;
    long long v0[4];
    auto leads = [&](const int r0, const double (&x)[4], long long (&v)[4]) {
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = r0 + 64 * u + lane < i1 ? lead(x[u]) : hi;      // (hi: neither below nor inside)
    };
    auto fetch = [&](const int r0, double (&x)[4]) {
#pragma unroll
        for (int u = 0; u < 4; ++u) { const int i = r0 + 64 * u + lane; x[u] = i < i1 ? load(i, 0) : 0.0; }
    };
    auto count = [&](const long long (&v)[4]) {
#pragma unroll
        for (int u = 0; u < 4; ++u) { below += v[u] < lo ? 1 : 0; mine += (v[u] >= lo && v[u] < hi) ? 1 : 0; }
    };
    leads(i0, x0, v0);
    count(v0);
    for (int r0 = i0 + 256; r0 < i1; r0 += 256) {
        double x[4]; long long v[4];
        fetch(r0, x); leads(r0, x, v); count(v);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { mine += __shfl_xor(mine, o, 64); below += __shfl_xor(below, o, 64); }
    __syncthreads();
    if (lane == 0) { s_wave[wave] = mine; s_wave[16 + wave] = below; }
    __syncthreads();
    int base = 0, m = 0, nb = 0;
    for (int w = 0; w < 16; ++w) { const int c = s_wave[w]; if (w < wave) base += c; m += c; nb += s_wave[16 + w]; }
    below_out = nb;
    if (tick) tick[0] = (long long)__builtin_amdgcn_s_memtime();
    if (m > kGroupCap) return m;
    auto take = [&](const int r0, const long long (&v)[4]) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const bool in = v[u] >= lo && v[u] < hi;
            const unsigned long long b = __ballot(in);
            if (in) { const int j = base + __popcll(b & ((1ull << lane) - 1ull)); s_key[j] = (unsigned int)(v[u] - lo); s_gid[j] = (unsigned int)(r0 + 64 * u + lane); }
            base += __popcll(b);
        }
    };
    if (i0 < i1) {
        take(i0, v0);
        for (int r0 = i0 + 256; r0 < i1; r0 += 256) {
            double x[4]; long long v[4];
            fetch(r0, x); leads(r0, x, v); take(r0, v);
        }
    }
    __syncthreads();
    if (tick) tick[1] = (long long)__builtin_amdgcn_s_memtime();
    // the keys, one taken point per thread (a wave's 64 points of an image row hold a few points of every group: in the walk above seven
    // of eight lanes would compute keys nobody needs)
    for (int j = tid; j < m; j += kWG) {
        const int i = (int)s_gid[j];
        const double y = load(i, 1), z = load(i, 2);
        s_key[j] = key_of(lo + (long long)s_key[j], y, z);
    }
    __syncthreads();
    return m;
}

// Stable order of the m <= kGroupCap keys s_key[0..m) of `bits` bits: order[r] = index of the r-th.  The keys are (column, depth) with a
// handful of points per column, so: count the points of every column (key >> zb, at most 4096 columns: LDS atomics, which also hand every
// point a slot in its column), prefix the counts, drop the points into their columns, and rank every point among the ones that share its
// column by (key, input position) — four barriers, where a radix pass over the same keys has five and the key needs two or three.
// col_start (if not null): set to the columns' starts when that path was taken (start[c], c < 1 << (bits - zb)), else to null: columns with more
// than kColumnMax points, or more than 4096 columns, go through the radix passes.  s_idx: 2 x kGroupCap, s_cnt: 16 KiB.
constexpr int kColumnMax = 64;
static __device__ __forceinline__ const unsigned short* group_sort(const unsigned int* s_key, const int m, const int bits, int zb, unsigned short* s_idx, unsigned short* s_cnt,
                                                                   int* s_wave, const unsigned int** col_start) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (zb < bits - 12) zb = bits - 12;
    if (zb > bits) zb = bits;
    const int ncol = 1 << (bits - zb);
    unsigned int* cnt = reinterpret_cast<unsigned int*>(s_cnt);   // [4096]: counts, then starts
    unsigned short* slot = s_idx + kGroupCap;                      // a point's slot in its column (until the points are dropped), then the order
    unsigned short* tmp = s_idx;                                   // the points column by column, unordered inside
    for (int c = tid; c < ncol; c += kWG) cnt[c] = 0;
    __syncthreads();
    for (int j = tid; j < m; j += kWG) slot[j] = (unsigned short)atomicAdd(&cnt[s_key[j] >> zb], 1u);
    __syncthreads();
    {   // exclusive prefix of the counts in place: thread t owns columns [4t, 4t + 4); largest count on the way
        unsigned int c[4], sum = 0, big = 0;
#pragma unroll
        for (int q = 0; q < 4; ++q) { c[q] = 4 * tid + q < ncol ? cnt[4 * tid + q] : 0; sum += c[q]; big = max(big, c[q]); }
        unsigned int incl = sum;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const unsigned int t = __shfl_up(incl, o, 64); if (lane >= o) incl += t; }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) big = max(big, (unsigned int)__shfl_xor((int)big, o, 64));
        if (lane == 63) { s_wave[wave] = (int)incl; s_wave[16 + wave] = (int)big; }
        __syncthreads();
        unsigned int base = incl - sum, worst = 0;
        for (int w = 0; w < 16; ++w) { if (w < wave) base += (unsigned int)s_wave[w]; worst = max(worst, (unsigned int)s_wave[16 + w]); }
        if (worst > (unsigned int)kColumnMax) {                     // (every thread alike)
            __syncthreads();
            if (col_start) *col_start = nullptr;
            RadixView rx;
            rx.key_lds = const_cast<unsigned int*>(s_key); rx.idx[0] = s_idx; rx.idx[1] = s_idx + kGroupCap; rx.hist = s_cnt;
            return wg_radix_sort(rx, m, bits, s_wave, [&](unsigned int id) { return s_key[id]; });
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) { if (4 * tid + q < ncol) cnt[4 * tid + q] = base; base += c[q]; }
    }
    __syncthreads();
    for (int j = tid; j < m; j += kWG) tmp[cnt[s_key[j] >> zb] + slot[j]] = (unsigned short)j;
    __syncthreads();
    for (int j = tid; j < m; j += kWG) {
        const unsigned int k = s_key[j], col = k >> zb;
        const int a = (int)cnt[col], b = (int)col + 1 < ncol ? (int)cnt[col + 1] : m;
        int rank = 0;
        for (int q = a; q < b; ++q) {
            const int jj = tmp[q];
            const unsigned int kk = s_key[jj];
            rank += (kk < k || (kk == k && jj < j)) ? 1 : 0;
        }
        slot[a + rank] = (unsigned short)j;
    }
    __syncthreads();
    if (col_start) *col_start = cnt;
    return slot;
}

__global__ void __launch_bounds__(kWG)
k_icp_voxel_wide(IcpBuffers B, int flags, double voxel) {
    __shared__ __attribute__((aligned(16))) unsigned int s_key[kGroupCap];
    __shared__ __attribute__((aligned(16))) unsigned int s_gid[kGroupCap];
    __shared__ __attribute__((aligned(16))) unsigned short s_idx[2 * kGroupCap];
    __shared__ __attribute__((aligned(16))) unsigned short s_cnt[16 * 512];
    __shared__ __attribute__((aligned(16))) double s_st[3 * kWG];
    __shared__ unsigned int s_sv[kWG];
    __shared__ int s_wave[32];
    __shared__ double s_mm[6];
    __shared__ int s_nv, s_base;
    const int g = blockIdx.x, h = blockIdx.y, which = blockIdx.z, tid = threadIdx.x;
    const int cloud = h * 2 + which;
    IcpState& S = B.st[h];
    const bool scene_mode = (flags & 1) != 0;
    if (g == 0 && which == 0 && tid < kIcpStrips) B.strip_pub[(size_t)h * kIcpStrips + tid] = 0;   // (k_icp_points_fused is through: its strips' counts are the next run's to publish)
    const int n = S.status == 0 ? (which ? S.n_scene : S.n_model) : 0;
    if (n == 0) return;                                            // (k_icp_voxel sets the counts of these)
    long long clk[6];
    clk[0] = (long long)__builtin_amdgcn_s_memtime();
    unsigned int* look = B.sort_look + (size_t)cloud * kIcpSortGroups;
    const double* pts = (which ? B.scene_pts : B.model_pts) + (size_t)h * B.cap * 3;
    double x0[4];                                                  // (on their way while the extent is worked out)
    {
        int i0, i1;
        collect_piece(n, i0, i1);
#pragma unroll
        for (int u = 0; u < 4; ++u) { const int i = i0 + 64 * u + (tid & 63); x0[u] = i < i1 ? pts[3 * (size_t)i] : 0.0; }
    }
    if (tid < 64) {
        strips_extent(B.strip_mm + (size_t)h * kIcpStrips * 12, which, s_mm);
        if (tid == 0) {                                            // the arithmetic of k_icp_voxel (one lane: sixteen waves dividing alike is a thousand cycles of the SIMDs)
            const double mnx = s_mm[0] - voxel * 0.5, mny = s_mm[1] - voxel * 0.5, mnz = s_mm[2] - voxel * 0.5;
            const double fx = floor(__ddiv_rn(s_mm[3] - mnx, voxel)), fy = floor(__ddiv_rn(s_mm[4] - mny, voxel)),
                         fz = floor(__ddiv_rn(s_mm[5] - mnz, voxel));
            s_mm[0] = mnx; s_mm[1] = mny; s_mm[2] = mnz; s_mm[3] = fx; s_mm[4] = fy; s_mm[5] = fz;
        }
    }
    if (tid == 64) s_nv = 0;
    __syncthreads();
    const double mnx = s_mm[0], mny = s_mm[1], mnz = s_mm[2], fx = s_mm[3], fy = s_mm[4], fz = s_mm[5];
    const double inv_voxel = 1.0 / voxel;
    const bool finite = fx >= 0 && fx < 4e18 && fy >= 0 && fy < 4e18 && fz >= 0 && fz < 4e18;
    const int bx = finite ? bits_for((long long)fx) : 64, by = finite ? bits_for((long long)fy) : 64, bz = finite ? bits_for((long long)fz) : 64;
    if (bx + by + bz > 32) return;                                 // every group alike: the cloud is k_icp_voxel's
    const long long span = (long long)fx + 1;                      // ix = 0 .. fx
    const long long lo = span * g / kIcpSortGroups, hi = span * (g + 1) / kIcpSortGroups;
    const int bits = (hi > lo ? bits_for(hi - lo - 1) : 1) + by + bz;
    double* out = (which ? B.tgt : B.src) + (size_t)h * B.cap * 3;
    int below;
    const int m = group_collect(n, lo, hi, s_key, s_gid, s_wave, below, x0,
        [&](int i, int k) { return pts[3 * (size_t)i + k]; },
        [&](double x) { return floor_quotient(x - mnx, voxel, inv_voxel); },
        [&](long long ix, double y, double z) {
            const unsigned long long iy = (unsigned long long)floor_quotient(y - mny, voxel, inv_voxel);
            const unsigned long long iz = (unsigned long long)floor_quotient(z - mnz, voxel, inv_voxel);
            return (unsigned int)(((((unsigned long long)(ix - lo)) << by) | iy) << bz | iz);
        });
    clk[1] = (long long)__builtin_amdgcn_s_memtime();
    if (m > kGroupCap) {                                           // the groups behind must not wait for this one
        if (tid == 0) __hip_atomic_store(look + g, kLookFail, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return;
    }
    const unsigned short* order = m > 0 ? group_sort(s_key, m, bits, bz, s_idx, s_cnt, s_wave, nullptr) : s_idx;
    clk[2] = (long long)__builtin_amdgcn_s_memtime();
    // voxels of this group: published before the means are taken, so that the groups behind find it there
    {
        int heads = 0;
        for (int i = tid; i < m; i += kWG) heads += (i == 0 || s_key[order[i - 1]] != s_key[order[i]]) ? 1 : 0;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) heads += __shfl_xor(heads, o, 64);
        if ((tid & 63) == 0 && heads) atomicAdd(&s_nv, heads);
    }
    __syncthreads();
    const int nv = s_nv;
    if (tid == 0) __hip_atomic_store(look + g, (unsigned int)nv + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (tid < 64) {
        unsigned int v = 1;
        if (tid < g) {
            const long long t0 = wall_clock64();
            do { v = __hip_atomic_load(look + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); } while (v == 0 && wall_clock64() - t0 < kLookTimeout);
        }
        const bool lost = __ballot(v == 0 || v == kLookFail) != 0ull;
        int before = (int)v - 1;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) before += __shfl_xor(before, o, 64);
        if (tid == 0) s_base = lost ? -1 : before;
    }
    __syncthreads();
    const int base_out = s_base;
    clk[3] = (long long)__builtin_amdgcn_s_memtime();
    if (base_out < 0) {                                            // a group before this one gave up or never came: the cloud is k_icp_voxel's
        if (tid == 0) __hip_atomic_store(look + g, kLookFail, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return;
    }
    int nout = base_out;
    auto vox_of = [&](int i) -> unsigned int { return s_key[order[i]]; };
    for (int base = 0; base < m; base += kWG) {                     // as k_icp_voxel: 1024 sorted positions side by side in LDS, the first thread of a voxel adds them up in input order
        const int i = base + tid;
        unsigned int v = 0;
        if (i < m) {
            const unsigned int id = order[i];
            v = s_key[id];
            const size_t p = s_gid[id];
            s_st[3 * tid] = pts[3 * p]; s_st[3 * tid + 1] = pts[3 * p + 1]; s_st[3 * tid + 2] = pts[3 * p + 2];
            s_sv[tid] = v;
        }
        __syncthreads();
        const bool head = i < m && (i == 0 || (tid > 0 ? s_sv[tid - 1] : vox_of(i - 1)) != v);
        int tot;
        const int pos = nout + block_scan_flag(head, s_wave, tot);
        nout += tot;
        if (head) {
            double sx = 0, sy = 0, sz = 0;
            int cnt = 0;
            for (int j = i; j < m; ++j) {
                const int t = j - base;
                if (t < kWG) {
                    if (s_sv[t] != v) break;
                    sx += s_st[3 * t]; sy += s_st[3 * t + 1]; sz += s_st[3 * t + 2];
                } else {
                    if (vox_of(j) != v) break;
                    const size_t p = s_gid[order[j]];
                    sx += pts[3 * p]; sy += pts[3 * p + 1]; sz += pts[3 * p + 2];
                }
                ++cnt;
            }
            const double cd = (double)cnt;
            out[3 * (size_t)pos] = sx / cd; out[3 * (size_t)pos + 1] = sy / cd; out[3 * (size_t)pos + 2] = sz / cd;
        }
        __syncthreads();
    }
    if (tid == 0) {
        if (g == kIcpSortGroups - 1) {
            if (which == 0) { S.n_src = nout; if (!scene_mode) S.n_tgt = nout; }
            else S.n_tgt = nout;
        }
        atomicAdd(&S.vox_done[which], 1);
        if (which == 0) {
            clk[4] = (long long)__builtin_amdgcn_s_memtime();
            for (int k = 0; k < 4; ++k) atomicMax(reinterpret_cast<unsigned long long*>(&S.sort_clk[k]), (unsigned long long)(clk[k + 1] - clk[k]));
            atomicMax(reinterpret_cast<unsigned long long*>(&S.sort_clk[6]), (unsigned long long)m);
        }
    }
}

__global__ void __launch_bounds__(kWG)
k_icp_grid_wide(IcpBuffers B, int flags) {
    __shared__ __attribute__((aligned(16))) unsigned int s_key[kGroupCap];
    __shared__ __attribute__((aligned(16))) unsigned int s_gid[kGroupCap];
    __shared__ __attribute__((aligned(16))) unsigned short s_idx[2 * kGroupCap];
    __shared__ __attribute__((aligned(16))) unsigned short s_cnt[16 * 512];
    __shared__ int s_wave[32];
    __shared__ double s_mm[6], s_par[3];
    __shared__ int s_geo[3];
    const int g = blockIdx.x, h = blockIdx.y, tid = threadIdx.x, lane = tid & 63;
    IcpState& S = B.st[h];
    if (S.status != 0) return;
    long long clk[5];
    clk[0] = (long long)__builtin_amdgcn_s_memtime();
    const int which = (flags & 1) ? 1 : 0;
    if (g == 0 && tid >= 64 && tid < 128) {                        // init_guess: centroid difference (LL.cpp:91-104), strips added in order (as k_icp_grid)
        double v[7];
#pragma unroll
        for (int q = 0; q < 7; ++q) v[q] = lane < kIcpStrips ? B.strip_sum[((size_t)h * kIcpStrips + lane) * 8 + q] : 0.0;
        double t[7] = {0, 0, 0, 0, 0, 0, 0};
        for (int k = 0; k < kIcpStrips; ++k)
#pragma unroll
            for (int q = 0; q < 7; ++q) t[q] += __shfl(v[q], k, 64);
        if (lane == 0) {
            const double n = (double)S.n_model;
            S.init[0] = t[3] / t[6] - t[0] / n;
            S.init[1] = t[4] / t[6] - t[1] / n;
            S.init[2] = t[5] / t[6] - t[2] / n;
        }
    }
    const int nt = S.n_tgt;
    if (nt == 0 || nt >= (1 << kIdxBits)) return;                  // (k_icp_grid sets the state of these)
    const double* T = (which ? B.tgt : B.src) + (size_t)h * B.cap * 3;
    double x0[4];                                                  // (on their way while the extent is worked out)
    {
        int i0, i1;
        collect_piece(nt, i0, i1);
#pragma unroll
        for (int u = 0; u < 4; ++u) { const int i = i0 + 64 * u + lane; x0[u] = i < i1 ? T[3 * (size_t)i] : 0.0; }
    }
    if (tid < 64) {
        strips_extent(B.strip_mm + (size_t)h * kIcpStrips * 12, which, s_mm);
        if (tid == 0) {                                            // the arithmetic of k_icp_grid, on the extent of the cloud the target was down-sampled from (one lane)
            const double minx = s_mm[0], miny = s_mm[1], minz = s_mm[2];
            const double ext = fmax(s_mm[3] - minx, s_mm[4] - miny), extz = s_mm[5] - minz;
            double cell = ext / (double)kIcpGrid;
            if (!(cell > kCellMin)) cell = kCellMin;
            double zres = extz / (double)((1 << kZBits) - 1);
            if (!(zres > 1e-3)) zres = 1e-3;
            const double inv = 1.0 / cell, inv_z = 1.0 / zres;
            const bool ok = ext < 1e30 && extz < 1e30;
            s_par[0] = cell; s_par[1] = inv; s_par[2] = inv_z;
            s_geo[0] = ok ? grid_coord(s_mm[3], minx, inv, kIcpGrid) + 1 : 0;
            s_geo[1] = ok ? grid_coord(s_mm[4], miny, inv, kIcpGrid) + 1 : 0;
            s_geo[2] = ok ? zq_of(s_mm[5], minz, inv_z, (1 << kZBits) - 1) : 0;
        }
    }
    __syncthreads();
    const double minx = s_mm[0], miny = s_mm[1], minz = s_mm[2], cell = s_par[0], inv = s_par[1], inv_z = s_par[2];
    const int gx = s_geo[0], gy = s_geo[1], zq_max = s_geo[2];
    if (gx == 0) return;                                           // non-finite coordinates: k_icp_grid rejects the hypothesis
    long long tick[3];
    tick[0] = (long long)__builtin_amdgcn_s_memtime();
    const int z_bits = bits_for(zq_max);
    const int lo = gx * g / kIcpSortGroups, hi = gx * (g + 1) / kIcpSortGroups;      // x columns of this group
    const int bits = (hi > lo ? bits_for((long long)(hi - lo) * gy - 1) : 1) + z_bits;
    int p0;
    const int m = group_collect(nt, lo, hi, s_key, s_gid, s_wave, p0, x0,
        [&](int i, int k) { return T[3 * (size_t)i + k]; },
        [&](double x) { return (long long)grid_coord(x, minx, inv, gx); },
        [&](long long cx, double y, double z) {
            const int cy = grid_coord(y, miny, inv, gy), zq = zq_of(z, minz, inv_z, zq_max);
            return ((unsigned int)(((int)cx - lo) * gy + cy) << z_bits) | (unsigned int)zq;
        }, tick + 1);
    clk[1] = (long long)__builtin_amdgcn_s_memtime();
    if (m > kGroupCap) return;                                     // grid_done stays short: k_icp_grid does the cloud
    const unsigned int* col_start = nullptr;
    const unsigned short* order = m > 0 ? group_sort(s_key, m, bits, z_bits, s_idx, s_cnt, s_wave, &col_start) : s_idx;
    if (bits - z_bits > 12) col_start = nullptr;                   // (the columns of the sort were coarser than the grid's)
    clk[2] = (long long)__builtin_amdgcn_s_memtime();
    double* Ts = B.tgt_sorted + (size_t)h * B.cap * 3;
    int* orig = B.tgt_orig + (size_t)h * B.cap;
    TgtRec* rec = B.tgt_rec + (size_t)h * B.cap;
    int* cs = B.cell_start + (size_t)h * kIcpCells;
    unsigned short* cs16 = B.cell_start16 + (size_t)h * kIcpCells16;
    for (int j = tid; j < m; j += kWG) {
        const unsigned int id = order[j];
        const size_t i = s_gid[id], p = (size_t)p0 + j;
        TgtRec r;
        r.x = T[3 * i]; r.y = T[3 * i + 1]; r.z = T[3 * i + 2]; r.orig = (int)i;
        r.zq = (int)(s_key[id] & ((1u << z_bits) - 1u));
        Ts[3 * p] = r.x; Ts[3 * p + 1] = r.y; Ts[3 * p + 2] = r.z;
        orig[p] = (int)i;
        rec[p] = r;
    }
    clk[3] = (long long)__builtin_amdgcn_s_memtime();
    // starts of this group's columns (column c of the grid = column c - lo * gy of the group); the last group also writes the end marker
    const int c_lo = lo * gy, c_hi = hi * gy;
    for (int c = c_lo + tid; c < c_hi; c += kWG) {
        if (col_start) { cs[c] = p0 + (int)col_start[c - c_lo]; cs16[c] = (unsigned short)(p0 + (int)col_start[c - c_lo]); continue; }
        const unsigned int want = (unsigned int)(c - c_lo) << z_bits;
        int a = 0, b = m;
        while (a < b) {
            const int mid = (a + b) >> 1;
            if (s_key[order[mid]] < want) a = mid + 1; else b = mid;
        }
        cs[c] = p0 + a;
        cs16[c] = (unsigned short)(p0 + a);
    }
    if (g == kIcpSortGroups - 1 && tid == 0) { cs[gx * gy] = nt; cs16[gx * gy] = (unsigned short)nt; }
    if (tid == 0) {
        if (g == 0) {
            S.gx = gx; S.gy = gy; S.zq_max = zq_max; S.gminx = minx; S.gminy = miny; S.gminz = minz; S.cell = cell; S.inv_cell = inv;
            S.inv_z = inv_z;
        }
        atomicAdd(&S.grid_done, 1);
        clk[4] = (long long)__builtin_amdgcn_s_memtime();
        for (int k = 0; k < 4; ++k) atomicMax(reinterpret_cast<unsigned long long*>(&S.sort_clk[8 + k]), (unsigned long long)(clk[k + 1] - clk[k]));
        atomicMax(reinterpret_cast<unsigned long long*>(&S.sort_clk[4]), (unsigned long long)(tick[0] - clk[0]));
        atomicMax(reinterpret_cast<unsigned long long*>(&S.sort_clk[5]), (unsigned long long)(tick[1] - tick[0]));
        atomicMax(reinterpret_cast<unsigned long long*>(&S.sort_clk[7]), (unsigned long long)(tick[2] - tick[1]));
        atomicMax(reinterpret_cast<unsigned long long*>(&S.sort_clk[13]), (unsigned long long)m);
        atomicMax(reinterpret_cast<unsigned long long*>(&S.sort_clk[14]), (unsigned long long)(c_hi - c_lo));
        atomicMax(reinterpret_cast<unsigned long long*>(&S.sort_clk[15]), (unsigned long long)bits);
    }
}

// ---------------------------------------------------------------------------------------------
// k_icp_knn: open3d EstimateNormals(KDTreeSearchParamKNN(30)), neighbour search part.  The columns within ring R of a point's
// column hold every point closer than R*cell, so once >= k candidates are closer than that, the k nearest of them are the
// k nearest of the cloud.  A workgroup owns a contiguous range of sorted positions (= an x slab of the grid) and stages the
// columns its base rings reach in LDS (grown rings read HBM where they leave it).
// EIGHT LANES per point, eight points per wave.  (A wave per point — candidates across 64 lanes, ballots for the selection —
// was bound by instruction issue, ~4000 wave instructions per point; a lane per point needs ~400 but leaves the chip
// underfilled at the pipeline's sizes — 128k points are 2000 waves, two per SIMD, each a 300k-cycle serial walk — and stalls
// whole waves on the isolated points whose rings grow to thousands of candidates.  Eight lanes keep the instruction count
// of the lane-per-point version, give 16k waves, and share a grown ring eight ways.)
// Nothing is stored per candidate: every pass walks the ring again and recomputes the squared distances (the oracle's
// expression); the lanes of a point take the candidates four at a time, round robin, and add up their counts (DPP) —
//   pass 1   the candidates closer than the guarantee radius g and than three fractions of it (grows the ring when < k);
//   pass 2   counts below four thresholds interpolated inside the bracket pass 1 left (the distances of a surface patch are
//            close to uniform in d^2);
//   pass 3+  every lane keeps the 4 smallest DISTINCT distances of the bracket it sees, with multiplicities (equal distances
//            are common in a cloud that comes off a pixel grid); the lanes' lists are merged smallest first until the
//            running count reaches k, or the bracket moves past what was collected;
//   ties at the k-th distance go to the lower original index (one pass per tie taken, rare);
//   last     the cumulants of the selected points and the distance to the nearest other point.
// ---------------------------------------------------------------------------------------------
constexpr int kKnnWG = 512;
constexpr int kKnnLanes = 8;       // lanes per point
constexpr int kKnnSlabPts = 2048;  // target points staged per workgroup (32-byte records, 64 KiB: two workgroups per CU)
constexpr int kKnnRuns = 16;       // x columns of a ring kept as separate runs (R <= 7; wider rings take whole x columns)
constexpr int kKnnHard = 512;      // points per round of a workgroup, any of which may be handed to a whole wave
constexpr int kKnnFarMax = 256;    // points per hypothesis k_icp_knn_far takes (more than that stay with their wave)
constexpr int kKnnFarBlocks = 96;  // ... with this many workgroups per hypothesis
constexpr int kKnnFew = 4;         // distinct distances a lane sorts in registers in a collecting pass

// reductions over the 8 lanes of a point (xor 1, xor 2, mirror within the half row): every lane ends with the result
template <int CTRL> static __device__ __forceinline__ int dpp_mov(int v) { return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xF, 0xF, false); }
template <int CTRL> static __device__ __forceinline__ double dpp_mov(double v) {
    return __hiloint2double(dpp_mov<CTRL>(__double2hiint(v)), dpp_mov<CTRL>(__double2loint(v)));
}
constexpr unsigned long long kInfKey = 0x7FF0000000000000ull;   // +infinity as a distance key
template <int CTRL> static __device__ __forceinline__ unsigned long long dpp_mov64(unsigned long long v);
template <int CTRL> static __device__ __forceinline__ unsigned long long dpp_mov(unsigned long long v) { return dpp_mov64<CTRL>(v); }
template <int CTRL> static __device__ __forceinline__ unsigned long long dpp_mov64(unsigned long long v) {
    const unsigned int lo = (unsigned int)__builtin_amdgcn_update_dpp(0, (int)(unsigned int)v, CTRL, 0xF, 0xF, false);
    const unsigned int hi = (unsigned int)__builtin_amdgcn_update_dpp(0, (int)(unsigned int)(v >> 32), CTRL, 0xF, 0xF, false);
    return ((unsigned long long)hi << 32) | lo;
}
static __device__ __forceinline__ unsigned long long shfl_xor_u64(unsigned long long v, int m) {
    const unsigned int lo = (unsigned int)__shfl_xor((int)(unsigned int)v, m, 64), hi = (unsigned int)__shfl_xor((int)(unsigned int)(v >> 32), m, 64);
    return ((unsigned long long)hi << 32) | lo;
}

static __device__ __forceinline__ int sum8(int v) { v += dpp_mov<0xB1>(v); v += dpp_mov<0x4E>(v); v += dpp_mov<0x141>(v); return v; }
static __device__ __forceinline__ double sum8(double v) { v += dpp_mov<0xB1>(v); v += dpp_mov<0x4E>(v); v += dpp_mov<0x141>(v); return v; }
static __device__ __forceinline__ double min8(double v) {
    v = fmin(v, dpp_mov<0xB1>(v)); v = fmin(v, dpp_mov<0x4E>(v)); v = fmin(v, dpp_mov<0x141>(v)); return v;
}
static __device__ __forceinline__ int min8(int v) { v = min(v, dpp_mov<0xB1>(v)); v = min(v, dpp_mov<0x4E>(v)); v = min(v, dpp_mov<0x141>(v)); return v; }

// f(j, d): this lane's share of the candidates of the ring — nx runs [runs[i].x, runs[i].y) of sorted positions (the columns
// ya..yb of one x are contiguous; the runs of a point sit in LDS) — with their squared distances to p: four consecutive
// candidates per trip, the lanes of the point side by side.  A run inside the staged slab [p0, p0 + np) comes from LDS.
template <int L, bool kFromLds, typename F>
static __device__ __forceinline__ void knn_scan_run(const int a, const int b, const int sub, const TgtRec* s_rel /*s_tgt - p0*/,
                                                    const TgtRec* __restrict__ g_rec, const double px, const double py, const double pz, F&& f) {
    constexpr int U = kFromLds ? 4 : 8;                            // candidates per lane and trip (HBM: more bytes in flight)
    // candidate v of a trip: j0 + v * L + sub — the lanes of a point read consecutive records (LDS: all banks once per group of
    // eight; HBM: coalesced), U per lane
    for (int j0 = a + sub; j0 < b; j0 += U * L) {
        double d4[U], q4[U][3];
#pragma unroll
        for (int v = 0; v < U; ++v) {
            const int j = j0 + v * L < b ? j0 + v * L : b - 1;
            if (kFromLds) { const TgtRec& r = s_rel[j]; q4[v][0] = r.x; q4[v][1] = r.y; q4[v][2] = r.z; }
            else { const double2 xy = *reinterpret_cast<const double2*>(&g_rec[j].x); q4[v][0] = xy.x; q4[v][1] = xy.y; q4[v][2] = g_rec[j].z; }
        }
#pragma unroll
        for (int v = 0; v < U; ++v) d4[v] = sqdist(px, py, pz, q4[v][0], q4[v][1], q4[v][2]);
        // (the squared distance as the unsigned integer its bit pattern is: it is >= 0, so the order is the same, and a dependent integer
        // compare + select costs a lone wave a few cycles where the f64 pair costs ~30 — profiles/r06_latency_microbench.txt)
#pragma unroll
        for (int v = 0; v < U; ++v)
            if (j0 + v * L < b) f(j0 + v * L, (unsigned long long)__double_as_longlong(d4[v]), q4[v][0], q4[v][1], q4[v][2]);
    }
}
// (two loops, not one loop with a choice of pointer inside: a pointer that may be LDS or HBM makes every load a FLAT load,
// which costs an LDS read the latency of a trip to memory)
template <int L, typename F>
static __device__ __forceinline__ void knn_scan(const int2* runs, const int nx, const int sub, const TgtRec* s_tgt, const int p0, const int np,
                                                const TgtRec* __restrict__ T, const double px, const double py, const double pz, F&& f) {
    for (int xi = 0; xi < nx; ++xi) {
        const int2 ab = runs[xi];
        if (ab.x >= p0 && ab.y <= p0 + np) knn_scan_run<L, true>(ab.x, ab.y, sub, s_tgt - p0, T, px, py, pz, f);
        else knn_scan_run<L, false>(ab.x, ab.y, sub, s_tgt - p0, T, px, py, pz, f);
    }
}

// One point, L lanes (8: the lane group of the main loop; 64: a whole wave, for the points whose ring has to grow).
template <int L> struct Red;
static __device__ __forceinline__ unsigned long long min8(unsigned long long v) {
    unsigned long long o = dpp_mov64<0xB1>(v); v = o < v ? o : v;
    o = dpp_mov64<0x4E>(v); v = o < v ? o : v;
    o = dpp_mov64<0x141>(v); v = o < v ? o : v;
    return v;
}
template <> struct Red<8> {
    static __device__ __forceinline__ unsigned long long mn(unsigned long long v) { return min8(v); }
    static __device__ __forceinline__ int sum(int v) { return sum8(v); }
    static __device__ __forceinline__ double sum(double v) { return sum8(v); }
    static __device__ __forceinline__ int mn(int v) { return min8(v); }
    static __device__ __forceinline__ double mn(double v) { return min8(v); }
};
template <> struct Red<64> {
    static __device__ __forceinline__ unsigned long long mn(unsigned long long v) { v = min8(v); for (int o = 8; o < 64; o <<= 1) { const unsigned long long t = shfl_xor_u64(v, o); v = t < v ? t : v; } return v; }
    static __device__ __forceinline__ int sum(int v) { v = sum8(v); for (int o = 8; o < 64; o <<= 1) v += __shfl_xor(v, o, 64); return v; }
    static __device__ __forceinline__ double sum(double v) { v = sum8(v); for (int o = 8; o < 64; o <<= 1) v += shfl_xor_d(v, o); return v; }
    static __device__ __forceinline__ int mn(int v) { v = min8(v); for (int o = 8; o < 64; o <<= 1) v = min(v, __shfl_xor(v, o, 64)); return v; }
    static __device__ __forceinline__ double mn(double v) { v = min8(v); for (int o = 8; o < 64; o <<= 1) v = fmin(v, shfl_xor_d(v, o)); return v; }
};

// a whole workgroup of 512 threads per point (k_icp_knn_far): wave reduction, then the 8 waves meet in LDS.  Every thread of
// the workgroup must make the same calls (the point's state is the same in all of them, so the control flow is).
template <> struct Red<512> {
    template <typename V, typename Op> static __device__ __forceinline__ V all(V v, Op op) {
        __shared__ V s_v[8];
        v = op(v, dpp_mov<0xB1>(v)); v = op(v, dpp_mov<0x4E>(v)); v = op(v, dpp_mov<0x141>(v));
        for (int o = 8; o < 64; o <<= 1) v = op(v, shfl_any(v, o));
        __syncthreads();                                             // (the previous reduction has been read)
        if ((threadIdx.x & 63) == 0) s_v[threadIdx.x >> 6] = v;
        __syncthreads();
        V r = s_v[0];
#pragma unroll
        for (int w = 1; w < 8; ++w) r = op(r, s_v[w]);
        return r;
    }
    static __device__ __forceinline__ int shfl_any(int v, int o) { return __shfl_xor(v, o, 64); }
    static __device__ __forceinline__ double shfl_any(double v, int o) { return shfl_xor_d(v, o); }
    static __device__ __forceinline__ unsigned long long shfl_any(unsigned long long v, int o) { return shfl_xor_u64(v, o); }
    static __device__ __forceinline__ unsigned long long mn(unsigned long long v) { return all(v, [](unsigned long long a, unsigned long long b) { return a < b ? a : b; }); }
    static __device__ __forceinline__ int sum(int v) { return all(v, [](int a, int b) { return a + b; }); }
    static __device__ __forceinline__ double sum(double v) { return all(v, [](double a, double b) { return a + b; }); }
    static __device__ __forceinline__ int mn(int v) { return all(v, [](int a, int b) { return a < b ? a : b; }); }
    static __device__ __forceinline__ double mn(double v) { return all(v, [](double a, double b) { return fmin(a, b); }); }
};

template <int L>
static __device__ __forceinline__ bool knn_point(const int pos, const int sub, int2* runs, const int k, const int R0, const int Rmax, const TgtRec* s_tgt, const int p0,
                                                 const int np, const double* __restrict__ T, const TgtRec* __restrict__ rec, const int* __restrict__ orig,
                                                 const int* __restrict__ cs, double* __restrict__ cov, const int gx, const int gy, const double minx, const double miny,
                                                 const double inv, const double cell) {
    const double kInfD = __longlong_as_double(0x7FF0000000000000ll);
    auto key = [](const double d) { return (unsigned long long)__double_as_longlong(d); };
    auto val = [](const unsigned long long k) { return __longlong_as_double((long long)k); };
    const double px = T[3 * (size_t)pos], py = T[3 * (size_t)pos + 1], pz = T[3 * (size_t)pos + 2];
    const int cx = grid_coord(px, minx, inv, gx), cy = grid_coord(py, miny, inv, gy);
    int R = R0, nx, c0, c1, c2, c3, M;
        double g2;
    bool all;
    for (;;) {
        const int xa = max(cx - R, 0), xb = min(cx + R, gx - 1), ya = max(cy - R, 0), yb = min(cy + R, gy - 1);
        all = xa == 0 && ya == 0 && xb == gx - 1 && yb == gy - 1;
        nx = xb - xa + 1;
        if (nx > kKnnRuns) {                                   // a ring that wide (tiny cloud): whole x columns, which are one run
            nx = 1;
            if (sub == 0) runs[0] = make_int2(cs[xa * gy], cs[(xb + 1) * gy]);
        } else {
            for (int xi = sub; xi < nx; xi += L) runs[xi] = make_int2(cs[(xa + xi) * gy + ya], cs[(xa + xi) * gy + yb + 1]);
        }
        if (L > 64) __syncthreads();                               // (within a wave the LDS operations are in order)
        const double g = (double)R * cell * (1.0 - 1e-9);     // margin >> the rounding of grid_coord
        g2 = g * g;
        // pass 1: how many candidates are closer than g, g/sqrt(2), g/2, g/sqrt(8)
        const unsigned long long kg = key(g2), k1 = key(g2 * 0.5), k2 = key(g2 * 0.25), k3 = key(g2 * 0.125);
        c0 = 0; c1 = 0; c2 = 0; c3 = 0; M = 0;
        knn_scan<L>(runs, nx, sub, s_tgt, p0, np, rec, px, py, pz, [&](int, unsigned long long d, double, double, double) {
            ++M; c0 += d < kg ? 1 : 0; c1 += d < k1 ? 1 : 0; c2 += d < k2 ? 1 : 0; c3 += d < k3 ? 1 : 0;
        });
        c0 = Red<L>::sum(c0); c1 = Red<L>::sum(c1); c2 = Red<L>::sum(c2); c3 = Red<L>::sum(c3); M = Red<L>::sum(M);
        if (c0 >= k || all) break;
        if (R >= Rmax) return false;                            // the ring has to grow further: the caller hands the point to a whole wave
        // isolated points: grow geometrically, not ring by ring (a whole wave doubles)
        R = L == 64 ? 2 * R : R + (R > 1 ? R >> 1 : 1);
    }
    // bracket: `cl` distances are < lo, `ch` are < hi, cl < k <= ch.  Selected in the end: d < v, and of the candidates at
    // d == v none (ties 0), all (1) or those up to original index last_o (2).
    unsigned long long lo = 0ull, hi = c0 >= k ? key(g2) : key(kInfD);     // (keys: see knn_scan_run)
    int cl = 0, ch = c0 >= k ? c0 : M;
    if (c0 >= k) {
        const unsigned long long h1 = key(g2 * 0.5), h2 = key(g2 * 0.25), h3 = key(g2 * 0.125);
        if (c1 >= k) { hi = h1; ch = c1; } else if (c1 > cl) { lo = h1; cl = c1; }
        if (c2 >= k) { hi = h2; ch = c2; } else if (c2 > cl) { lo = h2; cl = c2; }
        if (c3 >= k) { hi = h3; ch = c3; } else if (c3 > cl) { lo = h3; cl = c3; }
    }
    unsigned long long v = hi;
    int ties = ch == k ? 0 : -1, last_o = -1;
    if (ties < 0 && hi < key(kInfD)) {
        // pass 2: four thresholds around where the k-th smallest should be if the count is linear in between
        const double dlo = val(lo), dhi = val(hi);
        const double w = dhi - dlo, f0 = ((double)(k - cl) + 0.5) / (double)(ch - cl + 1), df = 1.5 / (double)(ch - cl + 1);
        const double td[4] = {dlo + w * (f0 - 3.0 * df), dlo + w * (f0 - df), dlo + w * (f0 + df), dlo + w * (f0 + 3.0 * df)};
        unsigned long long t[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) t[q] = key(td[q] > dlo ? (td[q] < dhi ? td[q] : dhi) : dlo);
        int n4[4] = {0, 0, 0, 0};
        knn_scan<L>(runs, nx, sub, s_tgt, p0, np, rec, px, py, pz, [&](int, unsigned long long d, double, double, double) {
#pragma unroll
            for (int q = 0; q < 4; ++q) n4[q] += d < t[q] ? 1 : 0;
        });
#pragma unroll
        for (int q = 0; q < 4; ++q) {                          // ascending thresholds: the last one below k raises lo, the first one at or above k lowers hi
            const int n = Red<L>::sum(n4[q]);
            if (n < k) { if (n >= cl && t[q] > lo) { lo = t[q]; cl = n; } }
            else if (t[q] < hi) { hi = t[q]; ch = n; }
        }
        if (ch == k) { ties = 0; v = hi; }
    }
    while (ties < 0) {
        // the kKnnFew smallest distinct distances in [lo, hi) this lane sees, with their multiplicities
        const unsigned long long kInf = key(kInfD);
        unsigned long long sv[kKnnFew];
        int sc[kKnnFew];
#pragma unroll
        for (int q = 0; q < kKnnFew; ++q) { sv[q] = kInf; sc[q] = 0; }
        knn_scan<L>(runs, nx, sub, s_tgt, p0, np, rec, px, py, pz, [&](int, unsigned long long d, double, double, double) {
            if (d >= lo && d < hi) {
                unsigned long long x = d;
                int xc = 1;
#pragma unroll
                for (int q = 0; q < kKnnFew; ++q) {
                    if (x == sv[q]) { sc[q] += xc; xc = 0; x = kInf; }
                    else if (x < sv[q]) { const unsigned long long tt = sv[q]; const int tc = sc[q]; sv[q] = x; sc[q] = xc; x = tt; xc = tc; }
                }
            }
        });
        // merge the 8 lists smallest value first.  A lane whose list ran empty after being full may have dropped larger
        // values: nothing above the smallest such "last kept" value can be trusted (limit)
        const unsigned long long limit = Red<L>::mn(sc[kKnnFew - 1] > 0 ? sv[kKnnFew - 1] : kInf);
        int cum = cl, less = -1, eq = 0;
        for (int it = 0; it < kKnnFew * L && less < 0; ++it) {
            const unsigned long long head = Red<L>::mn(sv[0]);
            if (!(head < kInf) || head > limit) break;
            const int mult = Red<L>::sum(sv[0] == head ? sc[0] : 0);
            if (cum + mult >= k) { v = head; less = cum; eq = mult; break; }
            cum += mult;
            if (sv[0] == head) {                                // pop
#pragma unroll
                for (int q = 0; q + 1 < kKnnFew; ++q) { sv[q] = sv[q + 1]; sc[q] = sc[q + 1]; }
                sv[kKnnFew - 1] = kInf; sc[kKnnFew - 1] = 0;
            }
            if (head == limit) break;                           // everything up to the limit is counted; beyond it lists are incomplete
        }
        if (less < 0) {                                          // the k-th is above what was collected: go on from there
            // every distance <= the last merged value is counted in cum; restart just above it
            const unsigned long long top = limit < kInf ? limit : hi;       // (limit == inf: all lists complete, so cum == ch >= k cannot happen here)
            lo = top + 1ull; cl = cum;                            // (the next double above)
            continue;
        }
        if (less + eq == k) { ties = 1; break; }
        ties = 2;                                                // k - less of the eq candidates at v, by original index
        for (int n = less; n < k; ++n) {
            int bo = INT_MAX;
            const int lo_o = last_o;
        knn_scan<L>(runs, nx, sub, s_tgt, p0, np, rec, px, py, pz, [&](int j, unsigned long long d, double, double, double) {
                if (d == v) { const int o = orig[j]; if (o > lo_o && o < bo) bo = o; }
            });
            last_o = Red<L>::mn(bo);
        }
    }
    double sum[9];
#pragma unroll
    for (int q = 0; q < 9; ++q) sum[q] = 0.0;
    unsigned long long sep2 = key(1e300);
    int taken = 0;
    knn_scan<L>(runs, nx, sub, s_tgt, p0, np, rec, px, py, pz, [&](int j, unsigned long long d, double qx, double qy, double qz) {
        if (j != pos && d < sep2) sep2 = d;
        bool sel = d < v;
        if (ties && d == v) sel = ties == 1 || orig[j] <= last_o;
        if (sel) {
            sum[0] += qx; sum[1] += qy; sum[2] += qz;
            sum[3] += qx * qx; sum[4] += qx * qy; sum[5] += qx * qz; sum[6] += qy * qy; sum[7] += qy * qz; sum[8] += qz * qz;
            ++taken;
        }
    });
#pragma unroll
    for (int q = 0; q < 9; ++q) sum[q] = Red<L>::sum(sum[q]);
    taken = Red<L>::sum(taken);
    sep2 = Red<L>::mn(sep2);
    if (sub == 0) {
        double* c = cov + (size_t)pos * kIcpCovStride;
#pragma unroll
        for (int q = 0; q < 9; ++q) c[q] = sum[q];
        c[9] = (double)taken;
        c[10] = val(sep2);
    }
    return true;
}

__global__ void __launch_bounds__(kKnnWG, 4)       // two workgroups per CU (their LDS allows it): 128 VGPRs — at 165 a CU held one, and the ~26 workgroups of a cloud ran in two rounds
k_icp_knn(IcpBuffers B, int knn) {
    __shared__ TgtRec s_tgt[kKnnSlabPts];
    __shared__ int2 s_runs[kKnnWG / kKnnLanes][kKnnRuns];
    __shared__ int s_hard[kKnnHard];
    __shared__ int s_nhard;
    const int h = blockIdx.y;
    const IcpState& S = B.st[h];
    if (S.status != 0 || S.n_tgt == 0) return;
    const int nt = S.n_tgt;
    const double* T = B.tgt_sorted + (size_t)h * B.cap * 3;
    const int* orig = B.tgt_orig + (size_t)h * B.cap;
    const int* cs = B.cell_start + (size_t)h * kIcpCells;
    const TgtRec* rec = B.tgt_rec + (size_t)h * B.cap;
    double* cov = B.cov + (size_t)h * B.cap * kIcpCovStride;
    int* far_list = reinterpret_cast<int*>(B.keys + (size_t)h * 2 * B.cap2);   // the sort scratch is free by now
    const int gx = S.gx, gy = S.gy;
    const double minx = S.gminx, miny = S.gminy, inv = S.inv_cell, cell = S.cell;
    const int k = knn < nt ? knn : nt;
    int R0 = (int)ceil(0.009 / cell);
    if (R0 < 1) R0 = 1;
    // workgroups at work on this cloud: ~64 points each = one trip of the lane groups (a small cloud on all of the grid's workgroups would stage itself 64 times)
    const int want = (nt + 63) / 64, nb = want < 16 ? 16 : want > (int)gridDim.x ? (int)gridDim.x : want;
    if ((int)blockIdx.x >= nb) return;
    const int q0 = (int)((long long)nt * blockIdx.x / nb), q1 = (int)((long long)nt * (blockIdx.x + 1) / nb);
    if (q0 >= q1) return;
    // the slab: the columns the rings R0 + 1 of this workgroup's points reach (R0 if that is too much for the LDS)
    int xlo = max(grid_coord(T[3 * (size_t)q0], minx, inv, gx) - (R0 + 1), 0);
    int xhi = min(grid_coord(T[3 * (size_t)(q1 - 1)], minx, inv, gx) + (R0 + 1), gx - 1);
    if (cs[(xhi + 1) * gy] - cs[xlo * gy] > kKnnSlabPts) { xlo = min(xlo + 1, xhi); xhi = max(xhi - 1, xlo); }
    if (nt <= kKnnSlabPts) { xlo = 0; xhi = gx - 1; }             // a small cloud is staged whole: grown rings stay in LDS too
    const int p0 = cs[xlo * gy];
    int np = cs[(xhi + 1) * gy] - p0;
    if (np > kKnnSlabPts) np = 0;                                 // slab too large for LDS: every ring reads HBM
    const long long k_t0 = (long long)__builtin_amdgcn_s_memtime();
    {
        const uint4* src = reinterpret_cast<const uint4*>(rec + p0);
        uint4* dst = reinterpret_cast<uint4*>(s_tgt);
        for (int j = threadIdx.x; j < np * 2; j += kKnnWG) dst[j] = src[j];
    }
    __syncthreads();
    const long long k_t1 = (long long)__builtin_amdgcn_s_memtime();
    long long k_main = 0, k_hard = 0;
    const int sub = threadIdx.x & (kKnnLanes - 1), lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int base = q0; base < q1; base += kKnnHard) {
        const int end = base + kKnnHard < q1 ? base + kKnnHard : q1;
        if (threadIdx.x == 0) s_nhard = 0;
        __syncthreads();
        // every lane group takes a point per trip; groups past the end idle (their lanes stay together: the DPP exchanges only
        // ever pair lanes of one group)
        for (int pos = base + threadIdx.x / kKnnLanes; pos < end; pos += kKnnWG / kKnnLanes) {
            const bool done = knn_point<kKnnLanes>(pos, sub, s_runs[threadIdx.x / kKnnLanes], k, R0, R0 + 1, s_tgt, p0, np, T, rec, orig, cs, cov, gx, gy, minx, miny,
                                                          inv, cell);
            if (!done && sub == 0) s_hard[atomicAdd(&s_nhard, 1)] = pos;
        }
        __syncthreads();
        const long long k_t2 = (long long)__builtin_amdgcn_s_memtime();
        // the points whose base ring held fewer than k candidates inside the guarantee radius (isolated points, flying pixels:
        // 2-15 % of a scene cloud, rings of hundreds to thousands of candidates): a wave each
        // (packing them onto the 8-lane groups once more, rings growing to 8, was measured: 117k cycles for that pass against 71k, and its
        // code cost the main trip 17k cycles in spilled registers)
        const int nhard = s_nhard;
        for (int i = wave; i < nhard; i += kKnnWG / 64) {
            const int pos = s_hard[i];
            int2* runs = s_runs[wave * (64 / kKnnLanes)];
            if (knn_point<64>(pos, lane, runs, k, R0 + 1 + ((R0 + 1) >> 1), 8, s_tgt, p0, np, T, rec, orig, cs, cov, gx, gy, minx, miny, inv, cell)) continue;
            // more than 8 rings from its k-th neighbour (a depth outlier, or a blob of fewer than k points away from the rest):
            // such a point needs the whole cloud — left to k_icp_knn_far, a workgroup each
            int slot = 0;
            if (lane == 0) slot = atomicAdd(&B.st[h].n_far, 1);
            slot = __shfl(slot, 0, 64);
            if (slot < kKnnFarMax) { if (lane == 0) far_list[slot] = pos; }
            else (void)knn_point<64>(pos, lane, runs, k, gx > gy ? gx : gy, INT_MAX, s_tgt, p0, np, T, rec, orig, cs, cov, gx, gy, minx, miny, inv, cell);
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            const long long k_t3 = (long long)__builtin_amdgcn_s_memtime();
            k_main += k_t2 - (base == q0 ? k_t1 : k_t2); k_hard += k_t3 - k_t2;
            if (base == q0) k_main = k_t2 - k_t1;
            atomicAdd((unsigned long long*)&B.st[h].knn_clk[3], (unsigned long long)nhard);
        }
    }
    if (threadIdx.x == 0) {
        atomicMax((unsigned long long*)&B.st[h].knn_clk[0], (unsigned long long)(k_t1 - k_t0));
        atomicMax((unsigned long long*)&B.st[h].knn_clk[1], (unsigned long long)k_main);
        atomicMax((unsigned long long*)&B.st[h].knn_clk[2], (unsigned long long)k_hard);
    }
}

// (Round 6 built this search twice more with the candidates of a ring held in registers — one walk instead of 5-7, the selection passes over
// the slots; 16 lanes x 8-16 slots per point, then 8 lanes x 16 slots with whole waves x 8 slots for grown rings — exact on every test and
// slower both times: 198 and 130 us against 85 on the icp leg's clouds.  The kernel is bound by instruction issue, not by the walks: a pass
// is 4 sixty-four-bit compares + adds per slot whether the key comes out of a register or out of eight f64 operations, and the unrolled
// slots cost the idle ones in full.  profiles/r06_knn_notes.txt.)
// The points k_icp_knn could not finish within 8 rings: a workgroup each, the whole cloud as one coalesced run (one wave
// streaming 12k records four times over took ~1 M cycles, and a blob of such points sits in ONE workgroup of k_icp_knn).
__global__ void __launch_bounds__(512)
k_icp_knn_far(IcpBuffers B, int knn) {
    __shared__ int2 s_runs[kKnnRuns];
    const int h = blockIdx.y;
    const IcpState& S = B.st[h];
    if (S.status != 0 || S.n_tgt == 0) return;
    const int nfar = S.n_far < kKnnFarMax ? S.n_far : kKnnFarMax;
    if ((int)blockIdx.x >= nfar) return;
    const int nt = S.n_tgt;
    const int big = S.gx > S.gy ? S.gx : S.gy;
    // (kKnnFarBlocks workgroups per cloud walk the list: a workgroup per possible entry was 4096 workgroups to dispatch for 16 clouds — 11 us
    // when the lists are empty, as they are for clouds without depth outliers)
    for (int i = blockIdx.x; i < nfar; i += gridDim.x) {
        const int pos = reinterpret_cast<const int*>(B.keys + (size_t)h * 2 * B.cap2)[i];
        (void)knn_point<512>(pos, (int)threadIdx.x, s_runs, knn < nt ? knn : nt, big, INT_MAX, nullptr, 0, 0, B.tgt_sorted + (size_t)h * B.cap * 3,
                             B.tgt_rec + (size_t)h * B.cap, B.tgt_orig + (size_t)h * B.cap, B.cell_start + (size_t)h * kIcpCells,
                             B.cov + (size_t)h * B.cap * kIcpCovStride, S.gx, S.gy, S.gminx, S.gminy, S.inv_cell, S.cell);
        __syncthreads();                                             // (s_runs and the reduction scratch are the next point's)
    }
}

// ---- 3x3 symmetric eigen decomposition (cyclic Jacobi), eigenvector of the smallest eigenvalue ----
static __device__ __attribute__((noinline)) void smallest_eigvec(double a00, double a01, double a02, double a11, double a12, double a22, double n[3]) {
    double A[3][3] = {{a00, a01, a02}, {a01, a11, a12}, {a02, a12, a22}};
    double V[3][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}};
    for (int sweep = 0; sweep < 12; ++sweep) {
        double off = fabs(A[0][1]) + fabs(A[0][2]) + fabs(A[1][2]);
        // (the sweeps converge quadratically: 1e-4, 1e-8, 1e-16, 1e-32 of the diagonal ... waiting for an exact zero was four more sweeps)
        if (off <= 1e-30 * (fabs(A[0][0]) + fabs(A[1][1]) + fabs(A[2][2]))) break;
        for (int p = 0; p < 2; ++p)
            for (int q = p + 1; q < 3; ++q) {
                double apq = A[p][q];
                if (apq == 0.0) continue;
                double theta = (A[q][q] - A[p][p]) / (2.0 * apq);
                double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
                double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
                for (int k = 0; k < 3; ++k) {   // A <- A * G
                    double akp = A[k][p], akq = A[k][q];
                    A[k][p] = c * akp - s * akq;
                    A[k][q] = s * akp + c * akq;
                }
                for (int k = 0; k < 3; ++k) {   // A <- G^T * A
                    double apk = A[p][k], aqk = A[q][k];
                    A[p][k] = c * apk - s * aqk;
                    A[q][k] = s * apk + c * aqk;
                }
                for (int k = 0; k < 3; ++k) {
                    double vkp = V[k][p], vkq = V[k][q];
                    V[k][p] = c * vkp - s * vkq;
                    V[k][q] = s * vkp + c * vkq;
                }
            }
    }
    int m = 0;
    if (A[1][1] < A[m][m]) m = 1;
    if (A[2][2] < A[m][m]) m = 2;
    n[0] = V[0][m]; n[1] = V[1][m]; n[2] = V[2][m];
}

// The same eigenvector in closed form: the smallest root of the characteristic polynomial by the trigonometric formula (q + 2 p cos(phi + 2 pi / 3),
// absolute error ~1e-16 of the largest eigenvalue), then the cross product of the two best-conditioned rows of A - lambda I.  Its angle to the
// exact eigenvector is ~1e-15 / (relative gap to the next eigenvalue) — measured against LAPACK on 10^6 covariances of 30-point surface patches:
// 2e-14 at gaps of 0.05, 1e-11 at 1e-3 — where the Jacobi sweeps above are a chain of ~17 rotations of two square roots and three divisions each
// (~35k cycles of one lane; this is ~5k).  False (nothing written) when the matrix is a multiple of the identity or not finite.
static __device__ __forceinline__ bool smallest_eigvec_closed(double a00, double a01, double a02, double a11, double a12, double a22, double n[3]) {
    const double q = (a00 + a11 + a22) * (1.0 / 3.0);
    const double b00 = a00 - q, b11 = a11 - q, b22 = a22 - q;
    const double p1 = a01 * a01 + a02 * a02 + a12 * a12;
    const double p2 = (b00 * b00 + b11 * b11 + b22 * b22 + 2.0 * p1) * (1.0 / 6.0);
    if (!(p2 > 0.0) || !(p2 < 1e300)) return false;
    const double ip = rsqrt(p2), p = p2 * ip;
    const double c00 = b00 * ip, c01 = a01 * ip, c02 = a02 * ip, c11 = b11 * ip, c12 = a12 * ip, c22 = b22 * ip;
    double r = 0.5 * (c00 * (c11 * c22 - c12 * c12) - c01 * (c01 * c22 - c12 * c02) + c02 * (c01 * c12 - c11 * c02));
    r = r < -1.0 ? -1.0 : (r > 1.0 ? 1.0 : r);
    const double phi = acos(r) * (1.0 / 3.0);
    const double lam = q + 2.0 * p * cos(phi + 2.0943951023931954923);      // the smallest eigenvalue
    const double r0[3] = {a00 - lam, a01, a02}, r1[3] = {a01, a11 - lam, a12}, r2[3] = {a02, a12, a22 - lam};
    auto cross = [](const double (&u)[3], const double (&w)[3], double (&o)[3]) {
        o[0] = u[1] * w[2] - u[2] * w[1]; o[1] = u[2] * w[0] - u[0] * w[2]; o[2] = u[0] * w[1] - u[1] * w[0];
        return o[0] * o[0] + o[1] * o[1] + o[2] * o[2];
    };
    double x01[3], x02[3], x12[3];
    const double n01 = cross(r0, r1, x01), n02 = cross(r0, r2, x02), n12 = cross(r1, r2, x12);
    const bool f01 = n01 >= n02 && n01 >= n12, f02 = n02 >= n12;
    const double nn = f01 ? n01 : (f02 ? n02 : n12);
    if (!(nn > 0.0) || !(nn < 1e300)) return false;
    const double s = rsqrt(nn);
#pragma unroll
    for (int k = 0; k < 3; ++k) n[k] = (f01 ? x01[k] : (f02 ? x02[k] : x12[k])) * s;
    return true;
}

// k_icp_normals: open3d ComputeNormal from the cumulants: covariance = E[xx^T] - E[x]E[x]^T,
// normal = eigenvector of its smallest eigenvalue ((0,0,1) for fewer than 3 neighbours).
__global__ void __launch_bounds__(256)
k_icp_normals(IcpBuffers B) {
    const int h = blockIdx.y;
    const IcpState& S = B.st[h];
    const int nt = S.status == 0 ? S.n_tgt : 0;
    const double* cov = B.cov + (size_t)h * B.cap * kIcpCovStride;
    double* N = B.normals + (size_t)h * B.cap * 3;
    for (int pos = blockIdx.x * blockDim.x + threadIdx.x; pos < nt; pos += gridDim.x * blockDim.x) {
        const double* c = cov + (size_t)pos * kIcpCovStride;
        const double k = c[9];
        double nrm[3] = {0, 0, 1};
        if (k >= 3.0) {
            const double mx = c[0] / k, my = c[1] / k, mz = c[2] / k;
            const double a00 = c[3] / k - mx * mx, a01 = c[4] / k - mx * my, a02 = c[5] / k - mx * mz, a11 = c[6] / k - my * my, a12 = c[7] / k - my * mz,
                         a22 = c[8] / k - mz * mz;
            if (!smallest_eigvec_closed(a00, a01, a02, a11, a12, a22, nrm)) smallest_eigvec(a00, a01, a02, a11, a12, a22, nrm);
            if (nrm[0] == 0 && nrm[1] == 0 && nrm[2] == 0) nrm[2] = 1;
        }
        N[3 * (size_t)pos] = nrm[0]; N[3 * (size_t)pos + 1] = nrm[1]; N[3 * (size_t)pos + 2] = nrm[2];
    }
}

// ---- RegistrationICP ------------------------------------------------------------------------------
// One launch (k_icp_eval) per ICP evaluation, grid (G, hypotheses): workgroup g owns a slice of the
// source points.  Splitting a hypothesis over G workgroups is what fills the chip at the batch sizes of
// the pipeline (16 hypotheses x 16 slices = 256 workgroups = one per CU); the stream order of the
// launches is the only synchronisation, converged hypotheses return at once.
constexpr int kSearchWG = 256;      // workgroup of k_icp_eval
constexpr int kIcpFineFrom = 6;     // evaluations from this one on run on kIcpMaxSplit slices per hypothesis
constexpr double kFarMargin = 1.2;  // search radius (x max_dist) of a source point that has no correspondence (1.5: 49 columns per search instead of 36; profiles/r02_icp_experiments.txt)
constexpr int kClasses = 8;         // search-cost classes of the queue (by overlapped grid columns)
constexpr int kLoopQueue = 1024;    // source points per round whose correspondence needs a grid search
constexpr int kSlabPts = 1024;      // target points of a slice's x slab staged in LDS (32-byte records)
constexpr int kSlabCells = 4096;    // cells of that slab (16-bit starts)

// Gaussian elimination with partial pivoting, [A | b] (6 x 7), in registers: every loop is unrolled, a row exchange is a chain of
// conditional swaps (no dynamic indexing, so nothing goes to scratch), one reciprocal per pivot.  Returns false if singular /
// non-finite.  (On LDS arrays — the first version — the ~250 dependent LDS accesses of the elimination were most of the
// evaluation's 6 us prologue, which every slice of every hypothesis pays before it can transform a point.)
static __device__ __forceinline__ bool solve6(double (&M)[6][7], double (&x)[6]) {
    double inv[6];
#pragma unroll
    for (int c = 0; c < 6; ++c) {
        int piv = c;
        double best = fabs(M[c][c]);
#pragma unroll
        for (int r = c + 1; r < 6; ++r)
            if (fabs(M[r][c]) > best) { best = fabs(M[r][c]); piv = r; }
        if (!(best > 0.0)) return false;
#pragma unroll
        for (int r = c + 1; r < 6; ++r) {
            const bool sw = piv == r;
#pragma unroll
            for (int q = c; q < 7; ++q) { const double a = M[c][q], b = M[r][q]; M[c][q] = sw ? b : a; M[r][q] = sw ? a : b; }
        }
        inv[c] = 1.0 / M[c][c];
#pragma unroll
        for (int r = c + 1; r < 6; ++r) {
            const double f = M[r][c] * inv[c];
#pragma unroll
            for (int q = c + 1; q < 7; ++q) M[r][q] -= f * M[c][q];
        }
    }
#pragma unroll
    for (int r = 5; r >= 0; --r) {
        double s = M[r][6];
#pragma unroll
        for (int q = r + 1; q < 6; ++q) s -= M[r][q] * x[q];
        x[r] = s * inv[r];
    }
    bool ok = true;
#pragma unroll
    for (int r = 0; r < 6; ++r) ok = ok && isfinite(x[r]);
    return ok;
}

// Sum of 32 per-lane values over the wave with 32 shuffles instead of 6 x 32: every step halves the
// number of values a lane carries (lanes whose bit `off` is set keep the upper half).  Afterwards lane l
// holds the wave total of value (l >> 1).
template <int N, int OFF, int M>
static __device__ __forceinline__ void reduce_halve(double (&v)[M], int lane) {
    const bool hi = (lane & OFF) != 0;
#pragma unroll
    for (int k = 0; k < N; ++k) {
        const double send = hi ? v[k] : v[k + N];
        const double keep = hi ? v[k + N] : v[k];
        v[k] = keep + shfl_xor_d(send, OFF);
    }
}
static __device__ __forceinline__ double wave_reduce32(double (&v)[32], int lane) {
    reduce_halve<16, 32>(v, lane);
    reduce_halve<8, 16>(v, lane);
    reduce_halve<4, 8>(v, lane);
    reduce_halve<2, 4>(v, lane);
    reduce_halve<1, 2>(v, lane);
    return v[0] + shfl_xor_d(v[0], 1);
}
// The same for 16 values per lane: afterwards lane l holds the wave total of value (l >> 2).
static __device__ __forceinline__ double wave_reduce16(double (&v)[16], int lane) {
    reduce_halve<8, 32>(v, lane);
    reduce_halve<4, 16>(v, lane);
    reduce_halve<2, 8>(v, lane);
    reduce_halve<1, 4>(v, lane);
    const double a = v[0] + shfl_xor_d(v[0], 2);
    return a + shfl_xor_d(a, 1);
}

// One ICP evaluation of one source slice.  Prologue (evaluations >= 1, every workgroup of the hypothesis
// redundantly, so that no second launch or inter-workgroup barrier is needed): add the G partials of
// the previous evaluation in fixed order, Open3D's relative-change convergence test,
// TransformationEstimationPointToPlane::ComputeTransformation (6x6 LU with partial pivoting),
// transformation = update * transformation (workgroup 0 records it).  Then pcd.Transform on the slice and
// the correspondences (GetRegistrationResultAndCorrespondences):
//   A1  every source point first re-measures its previous correspondence j: with d = |p - t_j|^2 and
//       sep2(j) = squared distance from t_j to its nearest other target (from k_icp_knn), 4 d < sep2(j)
//       proves by the triangle inequality that t_j is still the unique nearest neighbour — no search.
//       A point without correspondence carries a lower bound on its nearest-target distance (what its
//       last search saw, minus its motion since); while that exceeds max_dist it needs no search either.
//       The other points are queued in LDS, ordered by the number of grid columns their search cube
//       overlaps, so that the searches a wave runs in lock-step cost about the same;
//   A2  queued points search the cells overlapping the cube of half-width sqrt(min(d_prev, r^2)):
//       exact lexicographic minimum of (d, original index); a point of class c (<= 2^c columns) has 2^c lanes, one column
//       each, and all classes are walked in one sweep of the workgroup's lanes;
// and the slice's 32 partial sums (21 JtJ upper + 6 Jtr + sum d^2 + count, padded) for the next prologue.
// Returns true when the hypothesis is finished (converged, or evaluation max_iter done).
static __device__ __forceinline__ bool icp_eval_body(const IcpBuffers& B, IcpState& S, const int h, const int it, const int Gprev, const int max_shift, TgtRec* s_tgt,
                                                     unsigned short* s_cs, int* s_q, unsigned char* s_cls, const double max_dist,
                                                     const int max_iter, const double rel_tol, double* fit_hist, double* rmse_hist) {
    __shared__ double s_part[kSearchWG / 64][32];
    __shared__ double s_sum[32];
    __shared__ double s_U[12];
    __shared__ int s_stop;
    __shared__ int s_cnt[kClasses], s_cur[kClasses];
    __shared__ double s_xmm[kSearchWG / 64][2];
    __shared__ double s_red8[8][32];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int G = gridDim.x, g = blockIdx.x;
    const int ns = S.n_src, nt = S.n_tgt;
    const long long t0 = (long long)__builtin_amdgcn_s_memtime();

    const double* Src = B.src + (size_t)h * B.cap * 3;
    const double* T = B.tgt_sorted + (size_t)h * B.cap * 3;
    const double* N = B.normals + (size_t)h * B.cap * 3;
    const double* cov = B.cov + (size_t)h * B.cap * kIcpCovStride;
    const int* orig = B.tgt_orig + (size_t)h * B.cap;
    const int* cs = B.cell_start + (size_t)h * kIcpCells;
    double* P = B.work + (size_t)h * B.cap * 3;
    int* prev = B.prev_nn + (size_t)h * B.cap;
    double* lb = B.nn_lb + (size_t)h * B.cap;
    const int gx = S.gx, gy = S.gy, zq_max = S.zq_max;
    const double minx = S.gminx, miny = S.gminy, minz = S.gminz, inv = S.inv_cell, inv_z = S.inv_z;
    const TgtRec* rec = B.tgt_rec + (size_t)h * B.cap;
    const double r2 = max_dist * max_dist;
    const double far = max_dist * kFarMargin, far2 = far * far, lb_need = max_dist * (1.0 + 1e-9);
    const int i_lo = (int)((long long)ns * g / G), i_hi = (int)((long long)ns * (g + 1) / G);

    // this thread's (first) point and what the transform needs of its correspondence, requested before the prologue waits for
    // the slices' partial sums and the solve: the loads ride out that wait instead of starting after it
    const int i_pf = i_lo + tid;
    const bool pf = it > 0 && i_pf < i_hi;
    double pfx = 0, pfy = 0, pfz = 0, pflb = 0, pftx = 0, pfty = 0, pftz = 0, pfsep = 0;
    int pfj = -1;
    if (pf) {
        pfx = P[3 * (size_t)i_pf]; pfy = P[3 * (size_t)i_pf + 1]; pfz = P[3 * (size_t)i_pf + 2];
        pfj = prev[i_pf]; pflb = lb[i_pf];
        if (pfj >= 0) { pftx = T[3 * (size_t)pfj]; pfty = T[3 * (size_t)pfj + 1]; pftz = T[3 * (size_t)pfj + 2]; pfsep = cov[(size_t)pfj * kIcpCovStride + 10]; }
    }

    // ---- prologue: finish evaluation it - 1 ----
    if (it > 0) {
        const double* part = B.partial + (((size_t)((it - 1) & 1) * B.count + h) * kIcpMaxSplit) * 32;
        {   // fixed association: 8 interleaved groups of <= 8 slices each, loads issued together
            const int k = tid & 31, grp = tid >> 5;
            double a8[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int gg = grp + 8 * u;
                a8[u] = 0.0;
                if (gg < Gprev) a8[u] = part[(size_t)gg * 32 + k];
            }
            double v = 0;
#pragma unroll
            for (int u = 0; u < 8; ++u) v += a8[u];
            s_red8[grp][k] = v;
        }
        __syncthreads();
        if (tid < 32) {
            double v = 0;
#pragma unroll
            for (int w = 0; w < 8; ++w) v += s_red8[w][tid];
            s_sum[tid] = v;
        }
        __syncthreads();
        if (tid == 0) {
            const int ncorr = (int)s_sum[28];
            const double fit = ncorr ? (double)ncorr / (double)ns : 0.0;
            const double rmse = ncorr ? sqrt(s_sum[27] / (double)ncorr) : 0.0;
            bool stop = false;
            if (it > 1 && fabs(fit_hist[it & 1] - fit) < rel_tol && fabs(rmse_hist[it & 1] - rmse) < rel_tol) stop = true;
            if (it - 1 == max_iter) stop = true;
            if (g == 0) { fit_hist[(it - 1) & 1] = fit; rmse_hist[(it - 1) & 1] = rmse; }
            if (g == 0) {
                S.fitness = fit; S.rmse = rmse; S.n_corr = ncorr;
                if (stop) S.stop = 1;
            }
            s_stop = stop ? 1 : 0;
            if (!stop) {
                double M[6][7], x[6];
                {
                    double up[21];
#pragma unroll
                    for (int q = 0; q < 21; ++q) up[q] = s_sum[q];
                    int k = 0;
#pragma unroll
                    for (int a = 0; a < 6; ++a)
#pragma unroll
                        for (int c = a; c < 6; ++c) { M[a][c] = up[k]; M[c][a] = up[k]; ++k; }
#pragma unroll
                    for (int a = 0; a < 6; ++a) M[a][6] = -s_sum[21 + a];
                }
                double U[12] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0};
                if (ncorr >= 6 && solve6(M, x)) {
                    double sx, cx, sy, cy, sz, cz;
                    sincos(x[0], &sx, &cx); sincos(x[1], &sy, &cy); sincos(x[2], &sz, &cz);
                    // Rz(x2) * Ry(x1) * Rx(x0)
                    U[0] = cz * cy; U[1] = cz * sy * sx - sz * cx; U[2] = cz * sy * cx + sz * sx; U[3] = x[3];
                    U[4] = sz * cy; U[5] = sz * sy * sx + cz * cx; U[6] = sz * sy * cx - cz * sx; U[7] = x[4];
                    U[8] = -sy;     U[9] = cy * sx;                U[10] = cy * cx;               U[11] = x[5];
                }
                for (int a = 0; a < 12; ++a) s_U[a] = U[a];
                if (g == 0) {                               // transformation = update * transformation
                    double Tn[12];
                    for (int r = 0; r < 3; ++r)
                        for (int c = 0; c < 4; ++c)
                            Tn[4 * r + c] = U[4 * r] * S.T[c] + U[4 * r + 1] * S.T[4 + c] + U[4 * r + 2] * S.T[8 + c] + (c == 3 ? U[4 * r + 3] : 0.0);
                    for (int a = 0; a < 12; ++a) S.T[a] = Tn[a];
                    S.iterations = it;
                }
            }
        }
        __syncthreads();
        if (s_stop) return true;
    } else if (g == 0 && tid == 0) {
        for (int a = 0; a < 16; ++a) S.T[a] = (a % 5 == 0) ? 1.0 : 0.0;
        S.T[3] = S.init[0]; S.T[7] = S.init[1]; S.T[11] = S.init[2];
        S.iterations = 0;
    }
    if (it > max_iter) return true;                         // the last round only finishes evaluation max_iter
    const long long t1 = (long long)__builtin_amdgcn_s_memtime();

    // pcd.Transform: the initial guess at evaluation 0, the update afterwards
    double xmn = 1e300, xmx = -1e300;
    if (it == 0) {
        const double t0 = S.init[0], t1 = S.init[1], t2 = S.init[2];
        for (int i = i_lo + tid; i < i_hi; i += kSearchWG) {
            const double x = Src[3 * (size_t)i], y = Src[3 * (size_t)i + 1], z = Src[3 * (size_t)i + 2];
            const double nx = 1.0 * x + 0.0 * y + 0.0 * z + t0;
            P[3 * (size_t)i] = nx;
            P[3 * (size_t)i + 1] = 0.0 * x + 1.0 * y + 0.0 * z + t1;
            P[3 * (size_t)i + 2] = 0.0 * x + 0.0 * y + 1.0 * z + t2;
            prev[i] = -1;
            lb[i] = 0.0;
            xmn = fmin(xmn, nx - far * 1.001); xmx = fmax(xmx, nx + far * 1.001);
        }
    } else {
        double U[12];
#pragma unroll
        for (int a = 0; a < 12; ++a) U[a] = s_U[a];
        auto move_point = [&](const int i, const double x, const double y, const double z, const int pj, const double lbi, const double tx,
                              const double ty, const double tz, const double sep) {
            const double nx = U[0] * x + U[1] * y + U[2] * z + U[3];
            const double ny = U[4] * x + U[5] * y + U[6] * z + U[7];
            const double nz = U[8] * x + U[9] * y + U[10] * z + U[11];
            P[3 * (size_t)i] = nx; P[3 * (size_t)i + 1] = ny; P[3 * (size_t)i + 2] = nz;
            // how far this point's search will reach (the same tests as the queue below): nothing when its previous
            // correspondence is certified or it is provably out of range, the distance to the previous correspondence, or
            // kFarMargin x max_dist for a point without one
            double reach = 0.0;
            if (pj < 0) {
                const double nlb = lbi - (sqrt(sqdist(nx, ny, nz, x, y, z)) * (1.0 + 1e-9) + 1e-12);
                lb[i] = nlb;
                if (!(nlb > lb_need)) reach = far;
            } else {
                const double d = sqdist(nx, ny, nz, tx, ty, tz);
                if (!(d < r2 && 4.0 * d * (1.0 + 1e-9) < sep)) reach = sqrt(d < r2 ? d : r2);
            }
            reach = reach * (1.0 + 1e-6) + 1e-9;
            xmn = fmin(xmn, nx - reach); xmx = fmax(xmx, nx + reach);
        };
        if (pf) move_point(i_pf, pfx, pfy, pfz, pfj, pflb, pftx, pfty, pftz, pfsep);
        for (int i = i_pf + kSearchWG; i < i_hi; i += kSearchWG) {
            const int pj = prev[i];
            double tx = 0, ty = 0, tz = 0, sep = 0;
            if (pj >= 0) { tx = T[3 * (size_t)pj]; ty = T[3 * (size_t)pj + 1]; tz = T[3 * (size_t)pj + 2]; sep = cov[(size_t)pj * kIcpCovStride + 10]; }
            move_point(i, P[3 * (size_t)i], P[3 * (size_t)i + 1], P[3 * (size_t)i + 2], pj, lb[i], tx, ty, tz, sep);
        }
    }
    // the x slab of the grid this slice's searches can reach: a contiguous range of cells [c0, c1] and of sorted target
    // points [p0, p1), staged in LDS when it fits (sized by the actual search radii: once most points keep their
    // correspondence the slab is a few columns, not the kFarMargin x max_dist margin on either side)
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { xmn = fmin(xmn, shfl_xor_d(xmn, o)); xmx = fmax(xmx, shfl_xor_d(xmx, o)); }
    if (lane == 0) { s_xmm[wave][0] = xmn; s_xmm[wave][1] = xmx; }
    __syncthreads();
    for (int w = 0; w < kSearchWG / 64; ++w) { xmn = fmin(xmn, s_xmm[w][0]); xmx = fmax(xmx, s_xmm[w][1]); }
    const int xlo = grid_coord(xmn, minx, inv, gx), xhi = grid_coord(xmx, minx, inv, gx);
    const int c0 = xlo * gy, c1 = (xhi + 1) * gy;
    const int c0a = c0 & ~7;                                  // 16-byte aligned start of the table copy
    const int p0 = cs[c0], p1 = cs[c1];
    const int np = p1 - p0;
    const bool kLds = np <= kSlabPts && c1 - c0a + 1 <= kSlabCells && nt < 65536;
    if (tid == 0) {                                          // diagnostics: slices whose slab did not fit LDS, largest slab seen
        if (!kLds) atomicAdd((unsigned long long*)&S.clk[6], 1ull);
        atomicMax((unsigned long long*)&S.clk[7], (unsigned long long)np);
    }
    if (kLds) {                                              // 16-byte copies of the prepared records / 16-bit cell table
        const uint4* src = reinterpret_cast<const uint4*>(B.tgt_rec + (size_t)h * B.cap + p0);
        uint4* dst = reinterpret_cast<uint4*>(s_tgt);
        for (int j = tid; j < np * 2; j += kSearchWG) dst[j] = src[j];
        const uint4* csrc = reinterpret_cast<const uint4*>(B.cell_start16 + (size_t)h * kIcpCells16 + c0a);
        uint4* cdst = reinterpret_cast<uint4*>(s_cs);
        for (int j = tid; j < (c1 - c0a + 8) / 8; j += kSearchWG) cdst[j] = csrc[j];
    }
    __syncthreads();
    const long long t2 = (long long)__builtin_amdgcn_s_memtime();
    long long t_a2 = 0;

    // target point j (sorted position): from the staged slab when it is inside (always, for the candidates of a search;
    // a previous correspondence may have been left behind by a large update)
    auto tgt_xyz = [&](int j, double& x, double& y, double& z) {
        if (kLds && (unsigned)(j - p0) < (unsigned)np) { const TgtRec& r = s_tgt[j - p0]; x = r.x; y = r.y; z = r.z; }
        else { x = T[3 * (size_t)j]; y = T[3 * (size_t)j + 1]; z = T[3 * (size_t)j + 2]; }
    };
    auto tgt_orig = [&](int j) { return (kLds && (unsigned)(j - p0) < (unsigned)np) ? s_tgt[j - p0].orig : orig[j]; };
    auto tgt_zq = [&](int j) { return (kLds && (unsigned)(j - p0) < (unsigned)np) ? s_tgt[j - p0].zq : rec[j].zq; };
    auto cell_at = [&](int c) { return kLds ? (int)s_cs[c - c0a] : cs[c]; };

    for (int base = i_lo; base < i_hi; base += kLoopQueue) {
        const int end = base + kLoopQueue < i_hi ? base + kLoopQueue : i_hi;
        if (tid < kClasses) s_cnt[tid] = 0;
        __syncthreads();
        for (int i0 = base; i0 < end; i0 += kSearchWG) {
            const int i = i0 + tid;
            const int pj = i < end ? prev[i] : -2;
            bool need = i < end;
            int cls = kClasses;
            if (need) {
                const double px = P[3 * (size_t)i], py = P[3 * (size_t)i + 1], pz = P[3 * (size_t)i + 2];
                double bd0 = far2;
                if (pj >= 0) {
                    double qx, qy, qz;
                    tgt_xyz(pj, qx, qy, qz);
                    const double d = sqdist(px, py, pz, qx, qy, qz);
                    need = !(d < r2 && 4.0 * d * (1.0 + 1e-9) < cov[(size_t)pj * kIcpCovStride + 10]);
                    bd0 = d < r2 ? d : r2;
                } else {
                    need = !(lb[i] > lb_need);          // nearest target provably beyond max_dist: still no correspondence
                }
                if (need) {
                    const double rad = sqrt(bd0) * (1.0 + 1e-9) + 1e-12;
                    const int nxc = grid_coord(px + rad, minx, inv, gx) - grid_coord(px - rad, minx, inv, gx) + 1;
                    const int nyc = grid_coord(py + rad, miny, inv, gy) - grid_coord(py - rad, miny, inv, gy) + 1;
                    const int ncol = nxc * nyc;
                    cls = ncol <= 1 ? 0 : ncol <= 2 ? 1 : ncol <= 4 ? 2 : ncol <= 8 ? 3 : ncol <= 16 ? 4 : ncol <= 32 ? 5 : 6;   // lanes = 2^cls, one column each
                }
            }
            if (i < end) s_cls[i - base] = (unsigned char)cls;
#pragma unroll
            for (int c = 0; c < kClasses; ++c) {
                const unsigned long long m = __ballot(cls == c);
                if (m && lane == 0) atomicAdd(&s_cnt[c], __popcll(m));
            }
        }
        __syncthreads();
        if (tid == 0) {
            int run = 0;
            for (int c = 0; c < kClasses; ++c) { s_cur[c] = run; run += s_cnt[c]; }
        }
        __syncthreads();
        for (int i0 = base; i0 < end; i0 += kSearchWG) {
            const int i = i0 + tid;
            const int cls = i < end ? (int)s_cls[i - base] : kClasses;
#pragma unroll
            for (int c = 0; c < kClasses; ++c) {
                const unsigned long long m = __ballot(cls == c);
                if (m) {                                      // one LDS atomic per wave and class reserves the slots
                    int qb = 0;
                    if (lane == 0) qb = atomicAdd(&s_cur[c], __popcll(m));
                    qb = __shfl(qb, 0, 64);
                    if (cls == c) s_q[qb + __popcll(m & ((1ull << lane) - 1ull))] = i;
                }
            }
        }
        __syncthreads();
        const long long ta = (long long)__builtin_amdgcn_s_memtime();
        // The queue is ordered by cost class; a point of class c gets 2^c lanes, one grid column each, so that the lanes of a
        // wave finish together — a search of kFarMargin x max_dist for a point without correspondence overlaps dozens of columns and
        // would otherwise hold 63 lanes up.  All classes in one sweep of the workgroup's lanes: the points are laid out over the lanes widest class first (so that a
        // point's 2^shift lanes are aligned and never straddle a wave), lane t finds its class in the table of lane offsets.
        // Walking the classes one after the other cost a latency-bound pass per non-empty class (five or six per evaluation).
        int lane_end[kClasses], q_start[kClasses], total_lanes = 0;
#pragma unroll
        for (int c = kClasses - 1; c >= 0; --c) {
            const int cnt = s_cnt[c];
            q_start[c] = s_cur[c] - cnt;                       // s_cur[c] = end of the class in the queue, after the scatter
            total_lanes += cnt << (c < max_shift ? c : max_shift);
            lane_end[c] = total_lanes;
        }
        for (int t0 = 0; t0 < total_lanes; t0 += kSearchWG) {
            const int t = t0 + tid;
            const bool active = t < total_lanes;
            int cq = 0, lane0 = lane_end[1], qs = q_start[0];
#pragma unroll
            for (int c = kClasses - 1; c >= 1; --c) {
                const int first = c == kClasses - 1 ? 0 : lane_end[c + 1];
                if (t >= first && t < lane_end[c]) { cq = c; lane0 = first; qs = q_start[c]; }
            }
            const int lpp_shift = cq < max_shift ? cq : max_shift, lpp = 1 << lpp_shift;
            const int sub = (t - lane0) & (lpp - 1);
            const int i = active ? s_q[qs + ((t - lane0) >> lpp_shift)] : i_lo;
            const double px = P[3 * (size_t)i], py = P[3 * (size_t)i + 1], pz = P[3 * (size_t)i + 2];
            const int pj = prev[i];
            // a point without correspondence searches kFarMargin x max_dist once: the distance it finds (or the search
            // radius) minus its later motion is the lower bound that keeps it out of the queue (A1)
            const double bound2 = pj >= 0 ? r2 : far2;
            double bd = bound2;
            int bo = INT_MAX, bp = -1;
            if (pj >= 0) {
                double qx, qy, qz;
                tgt_xyz(pj, qx, qy, qz);
                const double d = sqdist(px, py, pz, qx, qy, qz);
                if (d < bd) { bd = d; bo = tgt_orig(pj); bp = pj; }
            }
            if (active && nt > 0 && px == px && py == py && pz == pz) {
                // every target with d <= bd lies in the cube of half-width sqrt(bd) around p: the columns overlapping it,
                // cut to its depth range, suffice
                const double rad = sqrt(bd) * (1.0 + 1e-9) + 1e-12;
                const int xa = grid_coord(px - rad, minx, inv, gx), xb = grid_coord(px + rad, minx, inv, gx);
                const int ya = grid_coord(py - rad, miny, inv, gy), yb = grid_coord(py + rad, miny, inv, gy);
                const int zlo = zq_of(pz - rad, minz, inv_z, zq_max), zhi = zq_of(pz + rad, minz, inv_z, zq_max);
                const int nxc = xb - xa + 1, ncol = nxc * (yb - ya + 1);
                const float inv_nxc = 1.0f / (float)nxc;
                for (int r = sub; r < ncol; r += lpp) {            // one column per lane and trip
                    const int yy = (int)(((float)r + 0.5f) * inv_nxc);       // r / nxc, exact for these small integers
                    const int c = (xa + (r - yy * nxc)) * gy + ya + yy;
                    int a = cell_at(c);
                    const int b = cell_at(c + 1);
                    if (b - a > 8) {                              // long run: first point at depth step >= zlo by bisection (the run is depth-ordered)
                        int hi = b;
                        while (a < hi) { const int mid = (a + hi) >> 1; if (tgt_zq(mid) < zlo) a = mid + 1; else hi = mid; }
                    }
                    // four candidates per trip (independent LDS reads in flight); indices past the run are
                    // clamped to its last point, which only re-tests a candidate; past depth step zhi the run is done
                    for (int j0 = a; j0 < b; j0 += 4) {
                        double d4[4];
                        int j4[4];
                        bool more = true;
#pragma unroll
                        for (int v = 0; v < 4; ++v) {
                            j4[v] = j0 + v < b ? j0 + v : b - 1;
                            double qx, qy, qz;
                            tgt_xyz(j4[v], qx, qy, qz);
                            d4[v] = sqdist(px, py, pz, qx, qy, qz);
                            if (tgt_zq(j4[v]) > zhi) more = false;
                        }
#pragma unroll
                        for (int v = 0; v < 4; ++v) {
                            const int j = j4[v];
                            const double d = d4[v];
                            if (d < bd) { bd = d; bo = tgt_orig(j); bp = j; }
                            else if (d == bd && bp >= 0 && bp != j) { const int o = tgt_orig(j); if (o < bo) { bo = o; bp = j; } }
                        }
                        if (!more) break;
                    }
                }
            }
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) {          // combine the lanes that shared the point (every lane makes every exchange)
                const double od = shfl_xor_d(bd, off);
                const int oo = __shfl_xor(bo, off, 64), op = __shfl_xor(bp, off, 64);
                if (off < lpp && op >= 0 && (od < bd || (od == bd && oo < bo))) { bd = od; bo = oo; bp = op; }
            }
            if (active && sub == 0) {
                if (bp >= 0 && !(bd < r2)) bp = -1;          // seen, but not a correspondence (d^2 < max_dist^2 required)
                prev[i] = bp;
                if (bp < 0) lb[i] = sqrt(bd);                 // every target closer than sqrt(bound2) was visited
            }
        }
        __syncthreads();
        t_a2 += (long long)__builtin_amdgcn_s_memtime() - ta;
    }
    const long long t3 = (long long)__builtin_amdgcn_s_memtime();
    // --- JtJ / Jtr of TransformationEstimationPointToPlane over the correspondences of the slice ---
    double acc[32];
#pragma unroll
    for (int k = 0; k < 32; ++k) acc[k] = 0.0;
    for (int i = i_lo + tid; i < i_hi; i += kSearchWG) {
        const int bp = prev[i];
        if (bp < 0) continue;
        const double px = P[3 * (size_t)i], py = P[3 * (size_t)i + 1], pz = P[3 * (size_t)i + 2];
        double qx, qy, qz;
        tgt_xyz(bp, qx, qy, qz);
        const double bd = sqdist(px, py, pz, qx, qy, qz);
        const double nx = N[3 * (size_t)bp], ny = N[3 * (size_t)bp + 1], nz = N[3 * (size_t)bp + 2];
        const double r = (px - qx) * nx + (py - qy) * ny + (pz - qz) * nz;
        const double J[6] = {py * nz - pz * ny, pz * nx - px * nz, px * ny - py * nx, nx, ny, nz};
        int k = 0;
#pragma unroll
        for (int a = 0; a < 6; ++a)
#pragma unroll
            for (int b = a; b < 6; ++b) acc[k++] += J[a] * J[b];
#pragma unroll
        for (int a = 0; a < 6; ++a) acc[21 + a] += J[a] * r;
        acc[27] += bd;
        acc[28] += 1.0;
    }
    {
        const double v = wave_reduce32(acc, lane);
        if ((lane & 1) == 0) s_part[wave][lane >> 1] = v;
    }
    __syncthreads();
    if (tid < 32) {
        double v = 0;
        for (int w = 0; w < kSearchWG / 64; ++w) v += s_part[w][tid];
        double* dst = B.partial + ((((size_t)(it & 1) * B.count + h) * kIcpMaxSplit) + g) * 32 + tid;
        if (tid >= 29) {   // diagnostics in the padding: shader cycles of this slice's evaluation (total, search, prologue)
            const long long tn = (long long)__builtin_amdgcn_s_memtime();
            v = tid == 29 ? (double)(tn - t0) : tid == 30 ? (double)t_a2 : (double)(s_cnt[0] + s_cnt[1] + s_cnt[2] + s_cnt[3] + s_cnt[4] + s_cnt[5] + s_cnt[6] + s_cnt[7]);
        }
        *dst = v;
    }
    if (g == 0 && tid == 0) {      // shader-cycle split of workgroup 0 (diagnostics): prologue, staging+transform, queue, search, sums
        const long long t4 = (long long)__builtin_amdgcn_s_memtime();
        S.clk[0] += t1 - t0; S.clk[1] += t2 - t1; S.clk[2] += (t3 - t2) - t_a2; S.clk[3] += t_a2; S.clk[4] += t4 - t3; S.clk[5] += 1;
    }
    return false;
}

__global__ void __launch_bounds__(kSearchWG, 3)
k_icp_eval(IcpBuffers B, int it, int prev_slices, int max_shift, double max_dist, int max_iter, double rel_tol) {
    __shared__ TgtRec s_tgt[kSlabPts];
    __shared__ __attribute__((aligned(16))) unsigned short s_cs[kSlabCells + 8];
    __shared__ int s_q[kLoopQueue];
    __shared__ unsigned char s_cls[kLoopQueue];
    const int h = blockIdx.y;
    IcpState& S = B.st[h];
    if (S.status != 0 || S.stop != 0) return;
    (void)icp_eval_body(B, S, h, it, prev_slices, max_shift, s_tgt, s_cs, s_q, s_cls, max_dist, max_iter, rel_tol, S.fit_hist, S.rmse_hist);
}

// ---- k_icp_team: RegistrationICP as ONE launch — a team of workgroups per hypothesis, every evaluation inside the kernel -----------
// The sliced launches above pay, per evaluation, a kernel launch and five or six dependent global round trips (the other slices'
// partial sums, the points, their previous correspondences, the slab) for slices that hold ~32 points (profiles/r05_icp_account.txt:
// 25-42 us per evaluation, 32 evaluations back to back).  Here nothing of a hypothesis leaves its CUs between the set-up and the final
// state: every workgroup of the team keeps the whole target cloud (32-byte records, normals, certification radii) and the 16-bit
// cell table in LDS for all evaluations, a source point lives in the registers of the thread that owns it (position, previous
// correspondence, bound), the 29 sums are reduced inside the workgroup in a fixed order, and wave 0 — which owns no points —
// finishes the evaluation (convergence test, 6x6 solve spread over its lanes, update).  The only traffic between the workgroups
// of a team is the all-gather of their 29 partial sums per evaluation: agent-scope (sc1) stores of the sums, a drained flag per
// workgroup, one poll instruction for all flags, G x 32 sc1 loads — ~1-2 us (MI355X_MICROARCH.md, hand-off price list) —, added
// in workgroup order by every member, which then finishes the evaluation redundantly: no second exchange, identical updates.
// Why a team and not one workgroup per hypothesis: measured (profiles/r06_icp_solo_first.txt), the updates of the bench's
// hypotheses move the points by 5-20 mm per evaluation (sliding along the surface), so ~2100 points search in EVERY evaluation;
// one CU needs 75-160 us for that, sixteen need 5-10.  A hypothesis stops on its own convergence and its CUs go idle.
//
// Which points must search.  The certification of k_icp_eval (4 d^2 < sep^2: the previous correspondence is closer than half the
// distance to ITS nearest neighbour) holds for one point in ten at a voxel size of 2.5 mm.  A search here also leaves a bound
// B = min(distance to the second nearest target it saw, radius it covered): every target other than the correspondence is at
// least B away.  A rigid update moves a point by at most |R - I|_F * rho + |(R - I) c + t| (c, rho: centre and radius of the
// workgroup's source points), and A = the sum of those bounds over the evaluations is kept by wave 0; a point whose
// correspondence is at distance d1 now keeps it without a search while d1 < B - (A_now - A_at_search) — strict, so the
// correspondence is the unique nearest target and index ties cannot arise.  While the updates are small the search radius is
// d1 + a margin, so that B has room above d1; once a hypothesis settles only the points near a Voronoi boundary search.
// Searching points go through a queue in LDS (ordered by cost class, 2^class lanes per point, one grid column per lane: the
// sweep of k_icp_eval); the queue is worked off in windows when more points search than it holds.
// A hypothesis whose clouds do not fit (slice > 3520 points, or target records + normals + table + a minimal queue > the LDS) is
// left alone (stop stays 0): the host then runs the sliced launches for it.  So is one whose team waited kTeamTimeout for a
// member (the GPU is shared and the grid was not resident at once).
constexpr int kSoloWG = 768;                // 12 waves = 3 per SIMD: 168 VGPRs each (1024 threads: 128, and the points' state went to scratch; 512: no more lanes than a search needs)
constexpr int kSoloOwners = kSoloWG - 64;   // threads that own source points (waves 1-11)
constexpr int kSoloRaw = 160 * 1024 - 5120; // bytes of the carve-out (the rest: partial sums, update matrix, counters)
constexpr int kSoloMinQueue = 128;          // the hypothesis is taken only if at least this many queue entries fit
constexpr double kSoloMargin = 0.25;        // search radius beyond the previous correspondence, in units of max_dist
constexpr long long kTeamTimeout = 100ll * 100000;          // wall_clock64 ticks (100 MHz): 100 ms
// a searching point: position, best squared distance so far and its target (-1: none), and what to visit — the x rows of its cube as runs of
// targets [ra, rb) (the columns of a row are consecutive cells), cum = chunks of four targets before a row; out: bd, bp, x = second best
struct __attribute__((aligned(16))) SoloQ { double x, y, z, bd; int bp, chunks; unsigned short ra[8], rb[8], cum[8]; };

static __device__ __forceinline__ double readlane_d(double v, int l) {
    const long long b = __double_as_longlong(v);
    const unsigned int lo = (unsigned int)__builtin_amdgcn_readlane((int)b, l), hi = (unsigned int)__builtin_amdgcn_readlane((int)(b >> 32), l);
    return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}

static __device__ __forceinline__ double fast_rcp(double x) {      // v_rcp_f64 + two Newton steps (the IEEE division is ~40 dependent instructions)
    double r = __builtin_amdgcn_rcp(x);
    r = fma(r, fma(-x, r, 1.0), r);
    r = fma(r, fma(-x, r, 1.0), r);
    return r;
}

// A literal the compiler must materialise where it is used: as plain literals the f64 constants of the evaluation loop are hoisted out of
// it into registers the kernel does not have, spilled, and reloaded from SCRATCH inside wave 0's chain (the polynomial below: 20 loads from
// global memory per evaluation, ~3000 cycles; profiles/r06_finish_notes.txt).  A volatile asm is not moved.
static __device__ __forceinline__ double here(double c) { asm volatile("" : "+v"(c)); return c; }

// sin and cos of the small angles an ICP update consists of: Taylor polynomials below pi / 4 (truncation < 5e-17), libm beyond
static __device__ __forceinline__ void sincos_small(const double a, double* sn, double* cs) {
    if (fabs(a) < 0.78) {
        const double z = a * a;
        double s = here(-1.0 / 1307674368000.0);                    // x^15 / 15!
        s = fma(s, z, here(1.0 / 6227020800.0)); s = fma(s, z, here(-1.0 / 39916800.0)); s = fma(s, z, here(1.0 / 362880.0)); s = fma(s, z, here(-1.0 / 5040.0));
        s = fma(s, z, here(1.0 / 120.0)); s = fma(s, z, here(-1.0 / 6.0));
        *sn = fma(a * z, s, a);
        double c = here(1.0 / 20922789888000.0);                    // x^16 / 16!
        c = fma(c, z, here(-1.0 / 87178291200.0)); c = fma(c, z, here(1.0 / 479001600.0)); c = fma(c, z, here(-1.0 / 3628800.0)); c = fma(c, z, here(1.0 / 40320.0));
        c = fma(c, z, here(-1.0 / 720.0)); c = fma(c, z, here(1.0 / 24.0)); c = fma(c, z, -0.5);
        *cs = fma(c, z, 1.0);
    } else {
        sincos(a, sn, cs);
    }
}

// The 6x6 normal equations solved by ONE WAVE with the matrix spread over its lanes: lane 8 r + c holds [A | b](r, c) (r < 6, c < 7).
// The algorithm of solve6 (partial pivoting, first largest pivot, one reciprocal per pivot), but a pivot step is a handful of
// cross-lane reads instead of 35 dependent multiply-subtracts in one lane, and nothing of the matrix occupies registers of the
// other waves.  `v`: lane k < 29 holds total k (21 JtJ upper, 6 Jtr, ..).
static __device__ __forceinline__ bool solve6_lanes(const double v, const int lane, double (&x)[6]) {
    const int r = lane >> 3, c = lane & 7;
    const bool valid = r < 6 && c < 7;
    const int lo = r < c ? r : c, hi = r < c ? c : r;
    const int src = !valid ? 0 : c == 6 ? 21 + r : lo * 6 - ((lo * (lo - 1)) >> 1) + (hi - lo);
    double a = __shfl(v, src, 64);
    if (c == 6) a = -a;
    if (!valid) a = 0.0;
    bool ok = true;
    double inv[6];
#pragma unroll
    for (int p = 0; p < 6; ++p) {
        // the largest |a| of column p at or below the diagonal, the first one on equality: magnitudes compared as the integers their bit
        // patterns are (sign bit cleared) — a dependent f64 compare + select costs a lone wave ~30 cycles, the integer pair a few
        double colp[6];
        unsigned long long mag[6];
#pragma unroll
        for (int q = p; q < 6; ++q) { colp[q] = readlane_d(a, 8 * q + p); mag[q] = (unsigned long long)__double_as_longlong(colp[q]) & 0x7FFFFFFFFFFFFFFFull; }
        int piv = p;
        unsigned long long best = mag[p];
        double pp = colp[p];
#pragma unroll
        for (int q = p + 1; q < 6; ++q) {
            const bool gt = mag[q] > best && mag[q] <= 0x7FF0000000000000ull;   // (a NaN is never a pivot, as with fabs(..) > best)
            best = gt ? mag[q] : best; piv = gt ? q : piv; pp = gt ? colp[q] : pp;
        }
        if (best == 0ull || best > 0x7FF0000000000000ull) ok = false;
        const int from = r == p ? piv : r == piv ? p : r;          // rows p and piv change places
        a = __shfl(a, 8 * from + c, 64);
        inv[p] = fast_rcp(pp);
        const double f = __shfl(a, 8 * r + p, 64) * inv[p];
        const double prow = __shfl(a, 8 * p + c, 64);
        if (valid && r > p && c > p) a = fma(-f, prow, a);
    }
    // back substitution, column by column: once x[q] is known every right-hand side above it is updated at once (independent operations:
    // two dependent ones per unknown instead of up to six)
    double rhs[6], up[6][6];
#pragma unroll
    for (int q = 0; q < 6; ++q) {
        rhs[q] = readlane_d(a, 8 * q + 6);
#pragma unroll
        for (int u = q + 1; u < 6; ++u) up[q][u] = readlane_d(a, 8 * q + u);
    }
#pragma unroll
    for (int q = 5; q >= 0; --q) {
        x[q] = rhs[q] * inv[q];
#pragma unroll
        for (int t = 0; t < q; ++t) rhs[t] = fma(-up[t][q], x[q], rhs[t]);
    }
#pragma unroll
    for (int q = 0; q < 6; ++q) ok = ok && isfinite(x[q]);
    return ok;
}

// TransformationEstimationPointToPlane::ComputeTransformation of one wave: the 6x6 solve, Rz * Ry * Rx and the translation -> U (3 x 4, row-major,
// identity when there are fewer than six correspondences or the system is singular), written to `out` (LDS) by lane 0.  NOT inlined into
// k_icp_team: inside the kernel's evaluation loop its constants are hoisted into registers the kernel does not have and come back from
// scratch in the middle of wave 0's chain (~3000 cycles per evaluation; with 512 threads = 256 VGPRs the same code took 520).
static __device__ __attribute__((noinline)) void team_update(const double v, const int ncorr, double* out) {
    const int lane = threadIdx.x & 63;
    double U[12] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0};
    double x[6];
    if (ncorr >= 6 && solve6_lanes(v, lane, x)) {
        // the three sincos side by side in lanes 0-2
        double sn, cs;
        sincos_small(lane == 0 ? x[0] : lane == 1 ? x[1] : x[2], &sn, &cs);
        const double sx = readlane_d(sn, 0), cx = readlane_d(cs, 0), sy = readlane_d(sn, 1), cy = readlane_d(cs, 1), sz = readlane_d(sn, 2), cz = readlane_d(cs, 2);
        // Rz(x2) * Ry(x1) * Rx(x0)
        U[0] = cz * cy; U[1] = cz * sy * sx - sz * cx; U[2] = cz * sy * cx + sz * sx; U[3] = x[3];
        U[4] = sz * cy; U[5] = sz * sy * sx + cz * cx; U[6] = sz * sy * cx - cz * sx; U[7] = x[4];
        U[8] = -sy;     U[9] = cy * sx;                U[10] = cy * cx;               U[11] = x[5];
    }
    if (lane == 0) {
#pragma unroll
        for (int a = 0; a < 12; ++a) out[a] = U[a];
    }
}

// the products of one correspondence that the point-to-plane normal equations add up: HALF 0 = JtJ entries 0..15 (row-major upper
// triangle), HALF 1 = JtJ 16..20, Jtr (6), d^2, 1 — two passes of 16 accumulators instead of one of 32 keep the registers for the points
template <int HALF>
static __device__ __forceinline__ void solo_accumulate(double (&acc)[16], const double px, const double py, const double pz, const TgtRec& q, const double* n3) {
    const double nx = n3[0], ny = n3[1], nz = n3[2];
    const double J[6] = {py * nz - pz * ny, pz * nx - px * nz, px * ny - py * nx, nx, ny, nz};
    if (HALF == 0) {
        int k = 0;
#pragma unroll
        for (int a = 0; a < 6; ++a)
#pragma unroll
            for (int b = a; b < 6; ++b) { if (k < 16) acc[k] = fma(J[a], J[b], acc[k]); ++k; }
    } else {
        const double bd = sqdist(px, py, pz, q.x, q.y, q.z);
        const double r = (px - q.x) * nx + (py - q.y) * ny + (pz - q.z) * nz;
        int k = 0;
#pragma unroll
        for (int a = 0; a < 6; ++a)
#pragma unroll
            for (int b = a; b < 6; ++b) { if (k >= 16) acc[k - 16] = fma(J[a], J[b], acc[k - 16]); ++k; }
#pragma unroll
        for (int a = 0; a < 6; ++a) acc[5 + a] = fma(J[a], r, acc[5 + a]);
        acc[11] += bd;
        acc[12] += 1.0;
    }
}

// Workgroups of a hypothesis of w source points when `extra` workgroups beyond one each are dealt out in proportion to the points (total = all
// points still at work): never more than one per min_points points (knobs: icp_team_min_points) — below that another member costs the team more (it gathers one
// more row of sums every evaluation) than it saves (the searches of a member are already a fraction of its evaluation).
static __device__ __forceinline__ int team_members(const int w, const long long total, const int extra, const int min_points) {
    int m = 1 + (int)((long long)extra * w / total);
    const int cap = (w + min_points - 1) / min_points;
    m = m > cap ? cap : m;
    m = m > kIcpMaxSplit ? kIcpMaxSplit : m;
    return m < 1 ? 1 : m;
}

// a target cloud that does not fit the LDS whole (with its normals): the slab build takes the hypothesis
static __device__ __forceinline__ bool team_needs_slab(const int n_tgt, const int gx, const int gy) {
    return n_tgt > ((kSoloRaw - ((2 * (gx * gy + 1) + 15) & ~15) - kSoloMinQueue * (int)sizeof(SoloQ)) / 60 & ~3);
}

template <int KP, bool SLAB>     // source points per owner thread (in registers); a slab of the target cloud in LDS, not all of it
__global__ void __launch_bounds__(kSoloWG)
k_icp_team(IcpBuffers B, unsigned int run, int shift_floor, double max_dist, int max_iter, double rel_tol, int cut_index, int min_points) {
    __shared__ __attribute__((aligned(16))) unsigned char s_raw[kSoloRaw];
    __shared__ double s_part[kSoloWG / 64][32];
    __shared__ double s_U[12];
    __shared__ double s_T[12], s_hist[4], s_fin[2];             // wave 0's: transformation so far, fitness / rmse of the last two evaluations and of the last one
    __shared__ double s_mot[6];                                  // A (motion bound summed over the evaluations), centre of the workgroup's source points, their radius
    __shared__ long long s_clk[7], s_fclk[6];
    __shared__ int s_cnt[kClasses];
    __shared__ int s_stop, s_fin_i[2];
    __shared__ int s_slab[5], s_ext[2];                          // the x columns staged [lo, hi], first target and number of targets staged, overflow; the columns this evaluation needs
    __shared__ int s_it0, s_cut_at;                              // the evaluation index this launch starts the hypothesis at (> 0: resumed) and the one it is cut after (0: none) — here, not in registers
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int h = blockIdx.y, g = blockIdx.x, G = gridDim.x;
    int cut_at = 0;                                             // evaluation index after whose finish stage the hypotheses still at work leave the launch (0: none)
    if (gridDim.y == 1 && B.count <= 64) {
        // One-dimensional grid (up to 64 hypotheses): the workgroups are dealt out to the hypotheses still to do in proportion to their source
        // points — a cloud of 10000 points next to one of 1500 gets seven times the members; with equal teams the large one sets the length
        // of the launch while the CUs of the small ones idle.  Every wave of every workgroup computes the same table (lane = hypothesis).
        int w = 0;
        bool big = false;                                       // a cloud that needs the slab build
        if (lane < B.count) {
            const IcpState& T = B.st[lane];
            if (T.status == 0 && T.stop == 0) {
                w = T.n_src > 0 ? T.n_src : 1;
                big = team_needs_slab(T.n_tgt, T.gx, T.gy);
            }
        }
        const int ncand = __popcll(__ballot(w > 0));
        if (ncand == 0) return;
        // The build without a slab takes a batch only if it can take ALL of it: a batch of small and large clouds would otherwise run as two
        // launches one after the other, each with most of the chip idle (the pipeline's 16 detections: 1.73 ms against 1.53 in one launch)
        if (!SLAB && __ballot(big)) return;
        long long total = w;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) total += __shfl_xor(total, off, 64);
        const int extra = (int)gridDim.x - ncand;
        const int members = w > 0 ? team_members(w, total, extra, min_points) : 0;
        // cramped: the hypotheses could use half as many workgroups again as there are (one per min_points source points each)
        int want = w > 0 ? (w + min_points - 1) / min_points : 0;
        want = want > kIcpMaxSplit ? kIcpMaxSplit : want;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) want += __shfl_xor(want, off, 64);
        if (cut_index > 0 && 2 * want >= 3 * (int)gridDim.x) cut_at = cut_index;
        int incl = members;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) { const int t = __shfl_up(incl, off, 64); if (lane >= off) incl += t; }
        const int start = incl - members, b = (int)blockIdx.x;
        const unsigned long long mine = __ballot(b >= start && b < start + members);
        if (!mine) return;                                      // (a workgroup the rounding left over)
        h = __ffsll((long long)mine) - 1;
        g = b - __shfl(start, h, 64);
        G = __shfl(members, h, 64);
    }
    IcpState& S = B.st[h];
    if (S.status != 0 || S.stop != 0) return;
    const int ns = S.n_src, nt = S.n_tgt;
    const int gx = S.gx, gy = S.gy, zq_max = S.zq_max, ncell = gx * gy;
    const int it0 = S.resume_it;                                 // > 0: suspended by an earlier launch of the round after the finish stage of this evaluation index
    if (g == 0 && tid == 0) S.team_size = G;                       // (diagnostics)
    // (every member of the team takes the same decision: it depends on the hypothesis only)
    if ((ns + G - 1) / G > kSoloOwners * KP || nt > 65535 || ncell > kIcpCells - 1 || gx > 255 || gy > 255) {
        if (g == 0 && tid == 0) { S.team_note[0] = (ns + G - 1) / G > kSoloOwners * KP ? 1 : 2; S.team_note[1] = ns; S.team_note[2] = nt; S.team_note[3] = KP; }
        return;
    }
    // LDS: C target points (32-byte records, normals, certification radii), the whole 16-bit cell table (absolute sorted positions), the queue.
    // A cloud of up to C points is resident whole; of a larger one the workgroup holds a SLAB — the targets of the x columns its source points can
    // reach, a contiguous range [p0, p0 + np) of the sorted cloud — and stages it again when an update has moved its points out of it.
    const int cs_bytes = (2 * (ncell + 1) + 15) & ~15;
    const bool whole = nt <= ((kSoloRaw - cs_bytes - kSoloMinQueue * (int)sizeof(SoloQ)) / 60 & ~3);      // the cloud fits with its normals: resident whole
    if (!SLAB && !whole) return;                                 // (the build with a slab takes it)
    const bool slabbed = SLAB && !whole;
    // (a slab leaves the normals in global memory — one gather per correspondence and evaluation — for 1.7 times the targets)
    const int C = slabbed ? (kSoloRaw - cs_bytes - kSoloMinQueue * (int)sizeof(SoloQ)) / 36 & ~3 : (nt + 3) & ~3;
    const int off_sep = 32 * C, off_n = off_sep + 4 * C, off_cs = off_n + (slabbed ? 0 : 24 * C), off_q = off_cs + cs_bytes;
    const int Q = (kSoloRaw - off_q) / (int)sizeof(SoloQ);
    const int i_lo = (int)((long long)ns * g / G), i_hi = (int)((long long)ns * (g + 1) / G);      // this workgroup's source points (voxel order: an x slab)
    unsigned long long* xchg = B.xchg + ((size_t)h * kIcpMaxSplit) * 64;   // [parity: + count * kIcpMaxSplit * 64][member][64 granules {half of a sum, tag}]
    const long long t_begin = (long long)__builtin_amdgcn_s_memtime();

    TgtRec* s_tgt = reinterpret_cast<TgtRec*>(s_raw);
    double* s_nrm = reinterpret_cast<double*>(s_raw + off_n);
    float* s_sep = reinterpret_cast<float*>(s_raw + off_sep);
    unsigned short* s_cs = reinterpret_cast<unsigned short*>(s_raw + off_cs);
    SoloQ* s_q = reinterpret_cast<SoloQ*>(s_raw + off_q);
    const TgtRec* g_rec = B.tgt_rec + (size_t)h * B.cap;
    const double* g_nrm = B.normals + (size_t)h * B.cap * 3;
    const double* g_cov = B.cov + (size_t)h * B.cap * kIcpCovStride;
    // targets [q0, q0 + n) of the sorted cloud -> LDS slots 0..n (records, normals, certification radii rounded down: a stricter test only searches more)
    auto stage = [&](const int q0, const int n) {
        const uint4* src = reinterpret_cast<const uint4*>(g_rec + q0);
        uint4* dst = reinterpret_cast<uint4*>(s_tgt);
        for (int j = tid; j < n * 2; j += kSoloWG) dst[j] = src[j];
        if (!slabbed) for (int j = tid; j < n * 3; j += kSoloWG) s_nrm[j] = g_nrm[(size_t)q0 * 3 + j];
        for (int j = tid; j < n; j += kSoloWG) s_sep[j] = __double2float_rd(g_cov[(size_t)(q0 + j) * kIcpCovStride + 10]);
    };
    {
        const uint4* csrc = reinterpret_cast<const uint4*>(B.cell_start16 + (size_t)h * kIcpCells16);
        uint4* cdst = reinterpret_cast<uint4*>(s_cs);
        for (int j = tid; j < (ncell + 8) / 8; j += kSoloWG) cdst[j] = csrc[j];
        if (!slabbed) stage(0, nt);
        if (tid < kClasses) s_cnt[tid] = 0;
        if (tid == 0) { s_slab[0] = slabbed ? 1 : 0; s_slab[1] = slabbed ? 0 : gx - 1; s_slab[2] = 0; s_slab[3] = slabbed ? 0 : nt; s_slab[4] = 0; s_ext[0] = INT_MAX; s_ext[1] = -1; }
    }
    const double minx = S.gminx, miny = S.gminy, minz = S.gminz, inv = S.inv_cell, inv_z = S.inv_z;
    const double r2 = max_dist * max_dist;
    const unsigned long long kr2 = (unsigned long long)__double_as_longlong(r2);
    const double far = max_dist * kFarMargin, far2 = far * far, lb_need = max_dist * (1.0 + 1e-9);

    // the source side: point i = i_lo + o + k * kSoloOwners belongs to owner o = tid - 64
    const int o = tid - 64;
    double px[KP], py[KP], pz[KP];
    int prv[KP];
    float lbf[KP];                                          // A at the last search + the bound that search left (every target other than prv is farther), rounded down
    {
        const double* Src = B.src + (size_t)h * B.cap * 3;
        // (a hypothesis an earlier launch of the round suspended goes on from the transformation it had reached: launch_icp_team)
        double T0[12] = {1.0, 0.0, 0.0, S.init[0], 0.0, 1.0, 0.0, S.init[1], 0.0, 0.0, 1.0, S.init[2]};
        if (it0 > 0) {
#pragma unroll
            for (int a = 0; a < 12; ++a) T0[a] = S.T[a];
        }
        double mn[3] = {1e300, 1e300, 1e300}, mx[3] = {-1e300, -1e300, -1e300};
#pragma unroll
        for (int k = 0; k < KP; ++k) {
            const int i = i_lo + o + k * kSoloOwners;
            px[k] = 0; py[k] = 0; pz[k] = 0; prv[k] = -1; lbf[k] = 0.f;
            if (o >= 0 && i < i_hi) {                              // pcd.Transform(init_guess)
                const double x = Src[3 * (size_t)i], y = Src[3 * (size_t)i + 1], z = Src[3 * (size_t)i + 2];
                px[k] = T0[0] * x + T0[1] * y + T0[2] * z + T0[3];
                py[k] = T0[4] * x + T0[5] * y + T0[6] * z + T0[7];
                pz[k] = T0[8] * x + T0[9] * y + T0[10] * z + T0[11];
                mn[0] = fmin(mn[0], px[k]); mn[1] = fmin(mn[1], py[k]); mn[2] = fmin(mn[2], pz[k]);
                mx[0] = fmax(mx[0], px[k]); mx[1] = fmax(mx[1], py[k]); mx[2] = fmax(mx[2], pz[k]);
            }
        }
        // centre and radius of the source cloud (its bounding box: a rigid motion keeps |p - c|, c moves along in the finish stage)
#pragma unroll
        for (int q = 0; q < 3; ++q)
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) { mn[q] = fmin(mn[q], shfl_xor_d(mn[q], off)); mx[q] = fmax(mx[q], shfl_xor_d(mx[q], off)); }
        if (lane == 0) {
#pragma unroll
            for (int q = 0; q < 3; ++q) { s_part[wave][q] = mn[q]; s_part[wave][3 + q] = mx[q]; }
        }
    }
    if (tid < 12) s_T[tid] = it0 > 0 ? S.T[tid] : tid % 5 == 0 ? 1.0 : tid == 3 ? S.init[0] : tid == 7 ? S.init[1] : tid == 11 ? S.init[2] : 0.0;
    if (tid < 4) s_hist[tid] = it0 > 0 ? (tid < 2 ? S.fit_hist[tid] : S.rmse_hist[tid - 2]) : 0.0;
    if (tid < 2) { s_fin[tid] = 0.0; s_fin_i[tid] = tid == 1 ? it0 : 0; }
    if (tid < 7) s_clk[tid] = 0;
    __syncthreads();
    if (tid == 0) {
        double mn[3], mx[3], rho2 = 0.0;
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            mn[q] = s_part[1][q]; mx[q] = s_part[1][3 + q];
            for (int w = 2; w < kSoloWG / 64; ++w) { mn[q] = fmin(mn[q], s_part[w][q]); mx[q] = fmax(mx[q], s_part[w][3 + q]); }
            s_mot[1 + q] = 0.5 * (mn[q] + mx[q]);
            rho2 += 0.25 * (mx[q] - mn[q]) * (mx[q] - mn[q]);
        }
        s_mot[0] = 0.0; s_mot[5] = 0.0;
        s_mot[4] = sqrt(rho2) * (1.0 + 1e-9) + 1e-12;              // NaN for an empty / non-finite cloud: then no bound ever certifies
    }
    __syncthreads();

    if (tid == 0) { s_it0 = it0; s_cut_at = cut_at; }      // (visible after the barrier above the loop... the one below)
    __syncthreads();
    for (int it = s_it0;; ++it) {
        const long long ta = (long long)__builtin_amdgcn_s_memtime();
        // ---- finish evaluation it - 1: totals, Open3D's convergence test, ComputeTransformation, transformation = update * transformation ----
        if (it > s_it0) {
            if (wave == 0) {
                // this workgroup's sums -> all-gather over the team -> the totals, added in workgroup order by every member
                // (added as a tree: a chain of eleven dependent f64 additions is 440 cycles for a lone wave, four levels are 160; fixed order all the same)
                double v;
                {
                    double pw[kSoloWG / 64 - 1];
#pragma unroll
                    for (int w = 1; w < kSoloWG / 64; ++w) pw[w - 1] = s_part[w][lane & 31];
                    static_assert(kSoloWG / 64 == 12, "the tree below adds the sums of eleven owner waves");
                    v = (((pw[0] + pw[1]) + (pw[2] + pw[3])) + ((pw[4] + pw[5]) + (pw[6] + pw[7]))) + ((pw[8] + pw[9]) + pw[10]);
                    if (lane == 29) v = 0.0;                         // (sum 29: members that give the hypothesis up — none do at present)
                }
                bool timed_out = false;
                const long long tx0 = (long long)__builtin_amdgcn_s_memtime();
                if (G > 1) {
                    // Tagged granules (MI355X_MICROARCH.md, hand-off price list): a sum travels as two 8-byte {half, tag} words, each written by ONE
                    // sc1 store and valid by itself — no drained flag behind the data, no flag poll before the gather: the gather IS the poll.
                    // tag = (run, evaluation): unique within the life of the buffer, so nothing is zeroed between runs; two buffers by
                    // evaluation parity, because a member can be one evaluation ahead of the slowest reader of its granules, never two.
                    const unsigned int tag = (run << 6) | (unsigned int)it;
                    unsigned long long* mine = xchg + ((size_t)((it - 1) & 1) * B.count * kIcpMaxSplit + g) * 64;
                    const unsigned long long bits = (unsigned long long)__double_as_longlong(v);
                    if (lane < 30) {
                        __hip_atomic_store(mine + 2 * lane, ((unsigned long long)tag << 32) | (bits & 0xFFFFFFFFull), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        __hip_atomic_store(mine + 2 * lane + 1, ((unsigned long long)tag << 32) | (bits >> 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
                    const unsigned long long* all = xchg + ((size_t)((it - 1) & 1) * B.count * kIcpMaxSplit) * 64;
                    const long long t0w = wall_clock64();
                    // lanes 0-31 gather members 0-7 of a group of 16, lanes 32-63 members 8-15 (16 loads per lane in flight); the sums are added
                    // in member order inside a half, then first half + second half, group after group: the same order in every member of the team
                    const int kk = (lane & 31) < 30 ? (lane & 31) : 0, half = lane >> 5;
                    v = 0;
                    for (int m0 = 0; m0 < G; m0 += 16) {
                        unsigned long long lo[8], hi[8];
                        for (;;) {
                            bool miss = false;
#pragma unroll
                            for (int u = 0; u < 8; ++u) {
                                const int m = m0 + half * 8 + u < G ? m0 + half * 8 + u : G - 1;
                                lo[u] = __hip_atomic_load(all + (size_t)m * 64 + 2 * kk, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                                hi[u] = __hip_atomic_load(all + (size_t)m * 64 + 2 * kk + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            }
#pragma unroll
                            for (int u = 0; u < 8; ++u) miss = miss || (unsigned int)(lo[u] >> 32) != tag || (unsigned int)(hi[u] >> 32) != tag;
                            if (!__ballot(miss)) break;
                            __builtin_amdgcn_s_sleep(1);
                            if (wall_clock64() - t0w > kTeamTimeout) { timed_out = true; break; }
                        }
                        if (timed_out) break;
                        double pm[8];
#pragma unroll
                        for (int u = 0; u < 8; ++u) pm[u] = m0 + half * 8 + u < G ? __longlong_as_double((long long)((hi[u] << 32) | (lo[u] & 0xFFFFFFFFull))) : 0.0;
                        const double part = ((pm[0] + pm[1]) + (pm[2] + pm[3])) + ((pm[4] + pm[5]) + (pm[6] + pm[7]));   // (a tree, as above)
                        const double other = shfl_xor_d(part, 32);
                        v += half == 0 ? part + other : other + part;      // (members 0-7) + (members 8-15), in both halves of the wave
                    }
                }
                if (lane == 0) s_clk[6] += (long long)__builtin_amdgcn_s_memtime() - tx0;
                // ComputeTransformation first, Open3D's convergence test after it in program order: the two are independent chains of f64
                // operations (the division and square root of the test alone are ~1000 cycles for a lone wave) and overlap this way; the
                // update of an evaluation that turns out to be the last is computed in vain, once per hypothesis
                const long long f0 = (long long)__builtin_amdgcn_s_memtime();
                const int ncorr = (int)readlane_d(v, 28);
                team_update(v, ncorr, s_U);
                const long long f1 = (long long)__builtin_amdgcn_s_memtime(), f2 = f1;
                double U[12];
#pragma unroll
                for (int a = 0; a < 12; ++a) U[a] = s_U[a];
                const double fit = ncorr ? (double)ncorr / (double)ns : 0.0;
                const double rmse = ncorr ? sqrt(readlane_d(v, 27) / (double)ncorr) : 0.0;
                bool stop = false;
                const double fit2 = s_hist[it & 1], rmse2 = s_hist[2 + (it & 1)];       // of evaluation it - 2
                if (it > 1 && fabs(fit2 - fit) < rel_tol && fabs(rmse2 - rmse) < rel_tol) stop = true;
                if (it - 1 == max_iter) stop = true;
                const bool team_over = readlane_d(v, 29) > 0.0;
                // The launch is CUT here for a batch that is cramped (launch_icp_team): most hypotheses of such a batch converge within a couple of
                // evaluations and their workgroups would idle while the ones that go on keep the small team they were dealt.  Every hypothesis
                // still at work after the finish stage of evaluation index `cut_at` leaves the launch (transformation, the last two fitness /
                // rmse values and the count go to IcpState) and the next launch deals the chip out among those.  The rule depends on the sizes
                // of the clouds and on the evaluation index only — not on which team is how far at the time —, so a run repeats itself.
                const bool team_cut = it == s_cut_at;          // (0: never — the finish stage runs from index 1 on)
                const long long f3 = (long long)__builtin_amdgcn_s_memtime() + (stop ? 1 : 0);
                if (lane == 0) {
                    s_stop = timed_out || team_over ? 2 : stop ? 1 : team_cut ? 3 : 0;
                    s_ext[0] = INT_MAX; s_ext[1] = -1;
                    s_hist[(it - 1) & 1] = fit; s_hist[2 + ((it - 1) & 1)] = rmse;
                    s_fin[0] = fit; s_fin[1] = rmse; s_fin_i[0] = ncorr;
                    if (!stop) s_fin_i[1] = it;
                    // how far this update can move a point of the cloud: |R - I|_F * rho + |(R - I) c + t|; the centre moves along
                    const double cx0 = s_mot[1], cy0 = s_mot[2], cz0 = s_mot[3];
                    const double ncx = U[0] * cx0 + U[1] * cy0 + U[2] * cz0 + U[3], ncy = U[4] * cx0 + U[5] * cy0 + U[6] * cz0 + U[7],
                                 ncz = U[8] * cx0 + U[9] * cy0 + U[10] * cz0 + U[11];
                    double e2[9];
#pragma unroll
                    for (int a = 0; a < 3; ++a)
#pragma unroll
                        for (int b = 0; b < 3; ++b) { const double e = U[4 * a + b] - (a == b ? 1.0 : 0.0); e2[3 * a + b] = e * e; }
                    const double fro = (((e2[0] + e2[1]) + (e2[2] + e2[3])) + ((e2[4] + e2[5]) + (e2[6] + e2[7]))) + e2[8];
                    const double step = sqrt(fro) * s_mot[4] + sqrt(sqdist(ncx, ncy, ncz, cx0, cy0, cz0));
                    s_mot[0] += step * (1.0 + 1e-9) + 1e-12;
                    s_mot[5] = step;
                    s_mot[1] = ncx; s_mot[2] = ncy; s_mot[3] = ncz;
                }
                if (!stop && lane < 12) {                           // transformation = update * transformation: lane 4 r + c makes element (r, c)
                    const int rr = lane >> 2, cc = lane & 3;
                    const double tn = s_U[4 * rr] * s_T[cc] + s_U[4 * rr + 1] * s_T[4 + cc] + s_U[4 * rr + 2] * s_T[8 + cc] + (cc == 3 ? s_U[4 * rr + 3] : 0.0);
                    s_T[lane] = tn;
                }
                if (lane < kClasses) s_cnt[lane] = 0;
                if (lane == 0) { s_fclk[0] = f0 - ta; s_fclk[1] = f1 - f0; s_fclk[2] = f2 - f1; s_fclk[3] = f3 - f2; s_fclk[4] = (long long)__builtin_amdgcn_s_memtime() - f3; }
            }
            __syncthreads();
            if (s_stop) break;
        }
        if (it > max_iter) break;                                 // the last round only finishes evaluation max_iter
        const long long tb = (long long)__builtin_amdgcn_s_memtime();

        // ---- pcd.Transform(update), then which points must search ----
        int cls[KP], rk[KP];
        unsigned int rows[KP][8];                                 // the x rows of the cube a search visits, as runs of targets: first | end << 16 (sorted positions)
        int nchunks[KP];                                          // chunks of four targets in them
        double seed[KP];                                          // squared distance the search starts from
        bool any = false;
        const double A = s_mot[0];
        // the room a search leaves above the correspondence it finds pays only when the updates are smaller than it (otherwise the next
        // update voids the bound — then nothing is paid for bounds: no margin, and points without correspondence search max_dist, not beyond)
        const bool calm = s_mot[5] < 0.5 * kSoloMargin * max_dist;
        const double margin = calm ? kSoloMargin * max_dist : 0.0, none2 = calm ? far2 : r2;
        int p0 = s_slab[2];                                       // first target of the slab in LDS (0: the whole cloud)
        bool in_lds = !slabbed || s_slab[4] == 0;                    // (a member whose points need more targets than the LDS holds reads them from global memory: slow, exact)
        auto rec_at = [&](const int j) -> TgtRec { if (!SLAB) return s_tgt[j]; if (__builtin_expect(in_lds, 1)) return s_tgt[j - p0]; return g_rec[j]; };
        auto sep_at = [&](const int j) -> double { if (!SLAB) return (double)s_sep[j]; if (__builtin_expect(in_lds, 1)) return (double)s_sep[j - p0]; return g_cov[(size_t)j * kIcpCovStride + 10]; };
        {
            double U[12] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0};
            if (it > s_it0) {
#pragma unroll
                for (int a = 0; a < 12; ++a) U[a] = s_U[a];
            }
            int ext_lo = INT_MAX, ext_hi = -1;                    // the x columns this thread's points need in LDS
#pragma unroll
            for (int k = 0; k < KP; ++k) {
                cls[k] = kClasses; rk[k] = 0; nchunks[k] = 0; seed[k] = none2;
#pragma unroll
                for (int r = 0; r < 8; ++r) rows[k][r] = 0;
                if (o < 0 || i_lo + o + k * kSoloOwners >= i_hi) continue;
                if (it > s_it0) {
                    const double x = px[k], y = py[k], z = pz[k];
                    px[k] = U[0] * x + U[1] * y + U[2] * z + U[3];
                    py[k] = U[4] * x + U[5] * y + U[6] * z + U[7];
                    pz[k] = U[8] * x + U[9] * y + U[10] * z + U[11];
                }
                const double room = ((double)lbf[k] - A) * (1.0 - 1e-9) - 1e-12;      // every target other than prv is farther than this (if positive)
                bool need;
                if (prv[k] >= 0) {
                    const TgtRec q = rec_at(prv[k]);
                    const double d = sqdist(px[k], py[k], pz[k], q.x, q.y, q.z);
                    need = !(d < r2 && ((room > 0.0 && d * (1.0 + 4e-9) < room * room) || 4.0 * d * (1.0 + 1e-9) < sep_at(prv[k])));
                    if (d < r2) seed[k] = d; else { seed[k] = r2; prv[k] = -1; }       // out of range now: searches max_dist, without a start
                } else {
                    need = !(room > lb_need);                     // nearest target provably beyond max_dist: still no correspondence
                }
                if (!(px[k] == px[k] && py[k] == py[k] && pz[k] == pz[k] && nt > 0)) { need = false; prv[k] = -1; lbf[k] = 0.f; }   // nothing to search
                if (need) cls[k] = -1;                            // (its class: below)
                if (slabbed && (need || prv[k] >= 0)) {            // the columns its correspondence and its search can lie in
                    const double R = (prv[k] >= 0 ? max_dist + margin : calm ? far : max_dist) * (1.0 + 1e-9) + 1e-12;
                    const int lo = grid_coord(px[k] - R, minx, inv, gx), hi = grid_coord(px[k] + R, minx, inv, gx);
                    ext_lo = lo < ext_lo ? lo : ext_lo; ext_hi = hi > ext_hi ? hi : ext_hi;
                }
            }
            if (slabbed) {
                // the slab: staged again (as wide as the LDS holds, around what is needed) when the points have left the columns it covers
#pragma unroll
                for (int off = 32; off > 0; off >>= 1) { ext_lo = min(ext_lo, __shfl_xor(ext_lo, off, 64)); ext_hi = max(ext_hi, __shfl_xor(ext_hi, off, 64)); }
                if (lane == 0 && ext_hi >= ext_lo) { atomicMin(&s_ext[0], ext_lo); atomicMax(&s_ext[1], ext_hi); }
                __syncthreads();
                const int need_lo = s_ext[0], need_hi = s_ext[1], st_lo = s_slab[0], st_hi = s_slab[1];
                if (need_hi >= need_lo && (need_lo < st_lo || need_hi > st_hi)) {
                    __syncthreads();                              // (everyone has read the old slab)
                    if (tid == 0) {
                        auto count = [&](const int lo, const int hi) { return (int)s_cs[(hi + 1) * gy] - (int)s_cs[lo * gy]; };
                        int e = 0;                                // widest symmetric margin that fits
#pragma unroll
                        for (int bit = 32; bit > 0; bit >>= 1) {
                            const int t = e + bit, lo = need_lo - t > 0 ? need_lo - t : 0, hi = need_hi + t < gx - 1 ? need_hi + t : gx - 1;
                            if (count(lo, hi) <= C) e = t;
                        }
                        const int lo = need_lo - e > 0 ? need_lo - e : 0, hi = need_hi + e < gx - 1 ? need_hi + e : gx - 1;
                        const int n = count(lo, hi);
                        if (n <= C) { s_slab[0] = lo; s_slab[1] = hi; s_slab[2] = s_cs[lo * gy]; s_slab[3] = n; s_slab[4] = 0; }
                        else { s_slab[0] = 1; s_slab[1] = 0; s_slab[2] = 0; s_slab[3] = 0; s_slab[4] = 1; S.team_note[0] = 3; S.team_note[1] = g; S.team_note[2] = n; S.team_note[3] = C; }   // nothing staged: global mode until the points come together again
                    }
                    __syncthreads();
                    stage(s_slab[2], s_slab[3]);
                    __syncthreads();
                    p0 = s_slab[2];
                    in_lds = s_slab[4] == 0;
                }
            }
#pragma unroll
            for (int k = 0; k < KP; ++k) {
                if (cls[k] != -1) continue;
                {
                    // A better start than the previous correspondence, which an update of several millimetres leaves far behind (the search
                    // radius is the distance to the start): the targets of the point's own grid column next to its depth.  Any target will
                    // do as a start — the search that follows is exact within the distance to it.
                    const int hx = grid_coord(px[k], minx, inv, gx), hy = grid_coord(py[k], miny, inv, gy);
                    {
                        const int c = hx * gy + hy;
                        const int a = s_cs[c], b = s_cs[c + 1];
                        if (b > a) {
                            const int zq = zq_of(pz[k], minz, inv_z, zq_max);
                            int lo = a, hi = b;                  // first target of the column at depth step >= zq
                            while (lo < hi) { const int mid = (lo + hi) >> 1; if (rec_at(mid).zq < zq) lo = mid + 1; else hi = mid; }
                            const int j0 = lo - 2 > a ? lo - 2 : a;
#pragma unroll
                            for (int v = 0; v < 4; ++v) {
                                const int j = j0 + v < b ? j0 + v : b - 1;
                                const TgtRec q = rec_at(j);
                                const double d = sqdist(px[k], py[k], pz[k], q.x, q.y, q.z);
                                if (d < seed[k] && d < r2) { seed[k] = d; prv[k] = j; }
                            }
                        }
                    }
                    // every target within `reach` of the point lies in the columns overlapping the cube of that half-width, cut to its depth range
                    const double reach = prv[k] >= 0 ? sqrt(seed[k]) + margin : sqrt(seed[k]);
                    const double rad = reach * (1.0 + 1e-9) + 1e-12;
                    const int xa = grid_coord(px[k] - rad, minx, inv, gx), xb = grid_coord(px[k] + rad, minx, inv, gx);
                    const int ya = grid_coord(py[k] - rad, miny, inv, gy), yb = grid_coord(py[k] + rad, miny, inv, gy);
                    const int nxc = xb - xa + 1, nyc = yb - ya + 1;
                    // the rows as runs of targets (the columns (x, ya..yb) are consecutive cells), and their chunks of four: the class — lanes = 2^cls, a chunk
                    // each — is set by the chunks themselves (by the columns, with ~4 targets assumed in one, tilted surfaces left some lanes three trips
                    // and others none).  More than eight rows (a grid finer than the search radius allows): one run from the first row's start to the last row's end
                    int chunks = 0;
#pragma unroll
                    for (int r = 0; r < 8; ++r) {
                        int ra = 0, rb = 0;
                        if (r < nxc && nxc <= 8) { const int c0 = (xa + r) * gy + ya; ra = s_cs[c0]; rb = s_cs[c0 + nyc]; }
                        if (r == 0 && nxc > 8) { ra = s_cs[xa * gy + ya]; rb = s_cs[(xa + nxc - 1) * gy + ya + nyc]; }
                        rows[k][r] = (unsigned int)ra | (unsigned int)rb << 16;
                        chunks += (rb - ra + 3) >> 2;
                    }
                    nchunks[k] = chunks;
                    if (chunks == 0) {                            // no target in the cube at all: the start (if any) is the nearest, everything else is beyond reach
                        cls[k] = kClasses;
                        lbf[k] = calm ? __double2float_rd(reach * (1.0 - 1e-9) + A) : 0.f;
                    } else {
                        cls[k] = chunks <= 1 ? 0 : chunks <= 2 ? 1 : chunks <= 4 ? 2 : chunks <= 8 ? 3 : chunks <= 16 ? 4 : chunks <= 32 ? 5 : 6;
                        any = true;
                    }
                }
            }
        }
        // queue slots: rank inside the class over the workgroup — per wave one LDS atomic for all classes (lane c adds class c)
        if (__ballot(any)) {
            int wcnt[kClasses];
#pragma unroll
            for (int c = 0; c < kClasses; ++c) wcnt[c] = 0;
#pragma unroll
            for (int k = 0; k < KP; ++k)
#pragma unroll
                for (int c = 0; c < kClasses - 1; ++c) {
                    const unsigned long long m = __ballot(cls[k] == c);
                    if (cls[k] == c) rk[k] = wcnt[c] + __popcll(m & ((1ull << lane) - 1ull));
                    wcnt[c] += __popcll(m);
                }
            int mine = 0;
#pragma unroll
            for (int c = 0; c < kClasses - 1; ++c) mine = lane == c ? wcnt[c] : mine;
            int base = 0;
            if (lane < kClasses && mine) base = atomicAdd(&s_cnt[lane], mine);
#pragma unroll
            for (int c = 0; c < kClasses - 1; ++c) {
                const int bc = __builtin_amdgcn_readlane(base, c);
#pragma unroll
                for (int k = 0; k < KP; ++k) if (cls[k] == c) rk[k] += bc;
            }
        }
        __syncthreads();
        const long long tc = (long long)__builtin_amdgcn_s_memtime();

        // ---- the searches: queue in class order, widest first; all classes in one sweep of the lanes ----
        int cnt[kClasses], nq = 0;
        long long d_scatter = 0, d_sweep = 0, d_read = 0, d_sw1 = 0, d_sw2 = 0, d_sw3 = 0;
        int d_lanes = 0, d_shift = 0;
#pragma unroll
        for (int c = 0; c < kClasses; ++c) { cnt[c] = __builtin_amdgcn_readfirstlane(s_cnt[c]); nq += cnt[c]; }
        if (nq > 0) {
            // lanes per point (2^min(class, max_shift)): a sweep costs its set-up however few columns a lane walks, so ONE sweep of the
            // workgroup's lanes when the points allow it; many points (one workgroup per hypothesis): the schedule of k_icp_eval
            auto lanes_at = [&](const int s) {
                int lanes = 0;
#pragma unroll
                for (int c = 0; c < kClasses; ++c) lanes += cnt[c] << (c < s ? c : s);
                return lanes;
            };
            int max_shift = 0;
            if (nq > kSoloWG) {
                max_shift = shift_floor;
#pragma unroll
                for (int s = 4; s <= 6; ++s) if (s > max_shift && lanes_at(s) <= 2 * kSoloWG) max_shift = s;
            } else {
#pragma unroll
                for (int s = 1; s <= 6; ++s) if (lanes_at(s) <= kSoloWG) max_shift = s;
            }
            int lane_end[kClasses], q_start[kClasses], total_lanes = 0, run = 0;
#pragma unroll
            for (int c = kClasses - 1; c >= 0; --c) {
                q_start[c] = run; run += cnt[c];
                total_lanes += cnt[c] << (c < max_shift ? c : max_shift);
                lane_end[c] = total_lanes;
            }
            int pos[KP];
#pragma unroll
            for (int k = 0; k < KP; ++k) {
                int qs = 0;
#pragma unroll
                for (int c = 0; c < kClasses; ++c) qs = cls[k] == c ? q_start[c] : qs;
                pos[k] = cls[k] < kClasses ? qs + rk[k] : -1;
            }
            auto first_lane_of = [&](const int e) {               // queue position -> the first of its lanes
                int lo = total_lanes;
#pragma unroll
                for (int c = kClasses - 1; c >= 0; --c) {
                    const int first = c == kClasses - 1 ? 0 : lane_end[c + 1];
                    if (e >= q_start[c] && e < q_start[c] + cnt[c]) lo = first + ((e - q_start[c]) << (c < max_shift ? c : max_shift));
                }
                return lo;
            };
            d_lanes = total_lanes; d_shift = max_shift;
            for (int w0 = 0; w0 < nq; w0 += Q) {
                const long long q0 = (long long)__builtin_amdgcn_s_memtime();
                const int wend = w0 + Q < nq ? w0 + Q : nq;
                if (w0 > 0) __syncthreads();                      // the previous window's results have been read
#pragma unroll
                for (int k = 0; k < KP; ++k)
                    if (pos[k] >= w0 && pos[k] < wend) {
                        SoloQ e;
                        e.x = px[k]; e.y = py[k]; e.z = pz[k]; e.bd = seed[k]; e.bp = prv[k];
                        int chunks = 0;
#pragma unroll
                        for (int r = 0; r < 8; ++r) {
                            const int ra = (int)(rows[k][r] & 0xFFFFu), rb = (int)(rows[k][r] >> 16);
                            e.ra[r] = (unsigned short)ra; e.rb[r] = (unsigned short)rb; e.cum[r] = (unsigned short)chunks;
                            chunks += (rb - ra + 3) >> 2;
                        }
                        e.chunks = chunks;
                        s_q[pos[k] - w0] = e;
                    }
                __syncthreads();
                const long long q1 = (long long)__builtin_amdgcn_s_memtime();
                const int tA = first_lane_of(w0) & ~63, tB = first_lane_of(wend);
                for (int t0 = tA; t0 < tB; t0 += kSoloWG) {
                    const int t = t0 + tid;
                    const long long w_0 = (long long)__builtin_amdgcn_s_memtime();
                    int cq = 0, lane0 = lane_end[1], qs = q_start[0];
#pragma unroll
                    for (int c = kClasses - 1; c >= 1; --c) {
                        const int first = c == kClasses - 1 ? 0 : lane_end[c + 1];
                        if (t >= first && t < lane_end[c]) { cq = c; lane0 = first; qs = q_start[c]; }
                    }
                    const int lpp_shift = cq < max_shift ? cq : max_shift, lpp = 1 << lpp_shift;
                    const int sub = (t - lane0) & (lpp - 1);
                    const int e = qs + ((t - lane0) >> lpp_shift);
                    const bool active = t < total_lanes && e >= w0 && e < wend;
                    SoloQ& ent = s_q[active ? e - w0 : 0];
                    const double qx = ent.x, qy = ent.y, qz = ent.z;
                    // Squared distances are compared and selected as the unsigned integers their bit patterns are (they are >= 0, so the order is the
                    // same): a dependent f64 compare + select costs a lone wave ~30 cycles, the integer pair a few (profiles/r06_latency_microbench.txt).
                    // kb: best so far, k2: second best (over the targets other than bp).  Equal distances (the lower ORIGINAL index wins) are not
                    // resolved on this path: the chunk or the merge step that meets one is redone by the exact routine below (wave-uniform branch).
                    unsigned long long kb = (unsigned long long)__double_as_longlong(ent.bd), k2 = kInfKey;
                    int bp = ent.bp;
                    const long long w_1 = (long long)__builtin_amdgcn_s_memtime() + (bp == 123456789 ? 1 : 0);
                    auto exact_visit = [&](const int j, const unsigned long long k) {       // one candidate, ties by original index
                        if (j == bp) return;
                        const bool better = k < kb || (k == kb && bp >= 0 && rec_at(j).orig < rec_at(bp).orig);
                        const unsigned long long second = better ? (bp >= 0 ? kb : kInfKey) : k;
                        k2 = second < k2 ? second : k2;
                        if (better) { kb = k; bp = j; }
                    };
                    if (active) {
                        const int chunks = ent.chunks;
                        const uint4 w_ra = *reinterpret_cast<const uint4*>(ent.ra), w_rb = *reinterpret_cast<const uint4*>(ent.rb), w_cum = *reinterpret_cast<const uint4*>(ent.cum);
                        const unsigned int a8[4] = {w_ra.x, w_ra.y, w_ra.z, w_ra.w}, b8[4] = {w_rb.x, w_rb.y, w_rb.z, w_rb.w}, c8[4] = {w_cum.x, w_cum.y, w_cum.z, w_cum.w};
                        // the targets and squared distances (as keys) of chunk q; a chunk beyond the last one: nothing (keys = infinity)
                        auto load_chunk = [&](const int q, unsigned long long (&k4)[4], int (&j4)[4]) {
                            int ra = (int)(a8[0] & 0xFFFFu), rb = (int)(b8[0] & 0xFFFFu), cu = 0;
#pragma unroll
                            for (int r = 1; r < 8; ++r) {                  // the row of chunk q: the last one whose first chunk is <= q (empty rows share their successor's)
                                const int cr = (int)(r & 1 ? c8[r >> 1] >> 16 : c8[r >> 1] & 0xFFFFu);
                                const bool in = q >= cr;
                                ra = in ? (int)(r & 1 ? a8[r >> 1] >> 16 : a8[r >> 1] & 0xFFFFu) : ra;
                                rb = in ? (int)(r & 1 ? b8[r >> 1] >> 16 : b8[r >> 1] & 0xFFFFu) : rb;
                                cu = in ? cr : cu;
                            }
                            const int j0 = ra + 4 * (q - cu);
                            const bool live = q < chunks;
#pragma unroll
                            for (int v = 0; v < 4; ++v) {                   // (independent LDS reads and distance chains in flight; the tail of a run repeats its last target)
                                j4[v] = live ? (j0 + v < rb ? j0 + v : rb - 1) : p0;
                                const TgtRec rr = rec_at(j4[v]);
                                const unsigned long long k = (unsigned long long)__double_as_longlong(sqdist(qx, qy, qz, rr.x, rr.y, rr.z));
                                k4[v] = (live & (j4[v] != bp) & !(v > 0 && j4[v] == j4[v > 0 ? v - 1 : 0])) ? k : kInfKey;     // a revisit of the start, or of the slot before, is no candidate
                            }
                        };
                        // four candidates reduce among themselves as a tree (two independent pairs): winner (key, target), the smallest of the
                        // three others, and whether two equal distances met at a node (a tie: the exact routine decides)
                        auto reduce4 = [&](const unsigned long long (&c4)[4], const int (&j4)[4], unsigned long long& wk, int& wj, unsigned long long& others, bool& tie) {
                            const bool s01 = c4[1] < c4[0], s23 = c4[3] < c4[2];
                            const unsigned long long w01 = s01 ? c4[1] : c4[0], l01 = s01 ? c4[0] : c4[1], w23 = s23 ? c4[3] : c4[2], l23 = s23 ? c4[2] : c4[3];
                            const int jw01 = s01 ? j4[1] : j4[0], jw23 = s23 ? j4[3] : j4[2];
                            const bool sf = w23 < w01;
                            wk = sf ? w23 : w01; wj = sf ? jw23 : jw01;
                            const unsigned long long lk = sf ? w01 : w23;
                            others = l01 < l23 ? l01 : l23;
                            others = lk < others ? lk : others;
                            tie = ((c4[0] == c4[1]) & (w01 != kInfKey)) | ((c4[2] == c4[3]) & (w23 != kInfKey)) | ((w01 == w23) & (wk != kInfKey));
                        };
                        for (int q = sub; q < chunks; q += lpp) {          // one chunk of four targets per lane and trip
                            unsigned long long ka[4], wk, others;
                            int ja[4], wj;
                            bool t4;
                            load_chunk(q, ka, ja);
                            reduce4(ka, ja, wk, wj, others, t4);
                            const bool tie = t4 | ((wk == kb) & (wk != kInfKey));
                            if (__ballot(tie)) {                            // (never, in clouds off a sensor)
#pragma unroll
                                for (int v = 0; v < 4; ++v) if (ka[v] != kInfKey) exact_visit(ja[v], ka[v]);
                            } else {                                        // once against the running best; k2 collects every distance that does not end up best
                                const bool better = wk < kb;
                                const unsigned long long out = better ? (bp >= 0 ? kb : kInfKey) : wk;
                                others = out < others ? out : others;
                                k2 = others < k2 ? others : k2;
                                kb = better ? wk : kb; bp = better ? wj : bp;
                            }
                        }
                    }
                    const long long w_2 = (long long)__builtin_amdgcn_s_memtime() + (bp == 123456789 ? 1 : 0);
                    // combine the lanes that shared the point: an all-reduce over aligned groups of lpp lanes — inside a row of 16 lanes by DPP
                    // (after the steps over 1 and 2 lanes a quad is uniform, after the half-row mirror a half row, ...), beyond it by permutes;
                    // a wave whose widest group is narrower skips the rest (wave-uniform)
                    auto merge = [&](const unsigned long long ok, const unsigned long long ok2, const int op, const bool in_group) {
                        const bool tie = in_group & (ok == kb) & (op != bp) & (op >= 0) & (bp >= 0);
                        if (__ballot(tie)) {                                // equal distances to two targets: the lower original index (exact, slow, never in practice)
                            if (in_group) {
                                const bool take = op >= 0 && (ok < kb || (ok == kb && bp >= 0 && op != bp && rec_at(op).orig < rec_at(bp).orig));
                                const unsigned long long second = take ? (bp >= 0 && bp != op ? kb : kInfKey) : (op >= 0 && op != bp ? ok : kInfKey);
                                k2 = second < k2 ? second : k2;
                                k2 = ok2 < k2 ? ok2 : k2;
                                if (take) { kb = ok; bp = op; }
                            }
                        } else {
                            const bool take = in_group & (op >= 0) & (ok < kb);
                            const unsigned long long second = take ? (((bp >= 0) & (bp != op)) ? kb : kInfKey) : ((in_group & (op >= 0) & (op != bp)) ? ok : kInfKey);
                            const unsigned long long s2 = ((second < ok2) | !in_group) ? second : ok2;
                            k2 = s2 < k2 ? s2 : k2;
                            kb = take ? ok : kb; bp = take ? op : bp;
                        }
                    };
                    if (__ballot(lpp > 1)) merge(dpp_mov64<0xB1>(kb), dpp_mov64<0xB1>(k2), dpp_mov<0xB1>(bp), lpp > 1);
                    if (__ballot(lpp > 2)) merge(dpp_mov64<0x4E>(kb), dpp_mov64<0x4E>(k2), dpp_mov<0x4E>(bp), lpp > 2);
                    if (__ballot(lpp > 4)) merge(dpp_mov64<0x141>(kb), dpp_mov64<0x141>(k2), dpp_mov<0x141>(bp), lpp > 4);
                    if (__ballot(lpp > 8)) merge(dpp_mov64<0x140>(kb), dpp_mov64<0x140>(k2), dpp_mov<0x140>(bp), lpp > 8);
                    if (__ballot(lpp > 16)) merge(shfl_xor_u64(kb, 16), shfl_xor_u64(k2, 16), __shfl_xor(bp, 16, 64), lpp > 16);
                    if (__ballot(lpp > 32)) merge(shfl_xor_u64(kb, 32), shfl_xor_u64(k2, 32), __shfl_xor(bp, 32, 64), lpp > 32);
                    if (active && sub == 0) {
                        if (bp >= 0 && !(kb < kr2)) { k2 = kb < k2 ? kb : k2; bp = -1; }      // seen, but not a correspondence (d^2 < max_dist^2 required)
                        if (bp < 0) k2 = kb < k2 ? kb : k2;           // without correspondence the bound is on every target (kb: the nearest seen, or the radius covered)
                        ent.bd = __longlong_as_double((long long)kb); ent.bp = bp; ent.x = __longlong_as_double((long long)k2);        // (x: every lane of the point has read it)
                    }
                    d_sw1 += w_1 - w_0; d_sw2 += w_2 - w_1; d_sw3 += (long long)__builtin_amdgcn_s_memtime() - w_2;
                }
                __syncthreads();
                const long long q2 = (long long)__builtin_amdgcn_s_memtime();
#pragma unroll
                for (int k = 0; k < KP; ++k)
                    if (pos[k] >= w0 && pos[k] < wend) {
                        const SoloQ& e = s_q[pos[k] - w0];
                        const bool had_start = prv[k] >= 0;
                        prv[k] = e.bp;
                        lbf[k] = 0.f;
                        if (calm) {                                   // every target other than the correspondence is at least this far: the second nearest seen, or the radius covered
                            const double reach = had_start ? sqrt(seed[k]) + margin : sqrt(seed[k]);
                            lbf[k] = __double2float_rd(fmin(sqrt(e.x), reach) * (1.0 - 1e-9) + A);
                        }
                    }
                d_scatter += q1 - q0; d_sweep += q2 - q1; d_read += (long long)__builtin_amdgcn_s_memtime() - q2;
            }
        }
        const long long td = (long long)__builtin_amdgcn_s_memtime();

        // ---- JtJ / Jtr of TransformationEstimationPointToPlane over the correspondences, reduced in a fixed order ----
        bool has = false;
#pragma unroll
        for (int k = 0; k < KP; ++k) has = has || prv[k] >= 0;
        if (!__ballot(has)) {                                     // a wave without correspondences (most waves of a team member): its sums are zero
            if (lane < 32) s_part[wave][lane] = 0.0;
        } else if (KP <= 2) {
            // one or two points per thread: all 29 sums at once (32 accumulators) and ONE halving reduction — the two passes of 16 below exist for the
            // registers of the five-points build, and cost a second chain of six dependent exchanges
            double acc[32];
#pragma unroll
            for (int q = 0; q < 32; ++q) acc[q] = 0.0;
#pragma unroll
            for (int k = 0; k < KP; ++k) {
                if (prv[k] < 0) continue;
                const TgtRec q = rec_at(prv[k]);
                double n3[3];
                if (slabbed) { n3[0] = g_nrm[3 * (size_t)prv[k]]; n3[1] = g_nrm[3 * (size_t)prv[k] + 1]; n3[2] = g_nrm[3 * (size_t)prv[k] + 2]; }
                else { n3[0] = s_nrm[3 * prv[k]]; n3[1] = s_nrm[3 * prv[k] + 1]; n3[2] = s_nrm[3 * prv[k] + 2]; }
                double lo[16], hi[16];
#pragma unroll
                for (int u = 0; u < 16; ++u) { lo[u] = acc[u]; hi[u] = acc[16 + u]; }
                solo_accumulate<0>(lo, px[k], py[k], pz[k], q, n3);
                solo_accumulate<1>(hi, px[k], py[k], pz[k], q, n3);
#pragma unroll
                for (int u = 0; u < 16; ++u) { acc[u] = lo[u]; acc[16 + u] = hi[u]; }
            }
            const double v = wave_reduce32(acc, lane);              // lane l: the wave total of value l >> 1
            if ((lane & 1) == 0) s_part[wave][lane >> 1] = v;
        } else
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            double acc[16];
#pragma unroll
            for (int q = 0; q < 16; ++q) acc[q] = 0.0;
#pragma unroll
            for (int k = 0; k < KP; ++k) {
                if (prv[k] < 0) continue;
                const TgtRec q = rec_at(prv[k]);
                double n3[3];
                if (slabbed) { n3[0] = g_nrm[3 * (size_t)prv[k]]; n3[1] = g_nrm[3 * (size_t)prv[k] + 1]; n3[2] = g_nrm[3 * (size_t)prv[k] + 2]; }
                else { n3[0] = s_nrm[3 * prv[k]]; n3[1] = s_nrm[3 * prv[k] + 1]; n3[2] = s_nrm[3 * prv[k] + 2]; }
                if (half == 0) solo_accumulate<0>(acc, px[k], py[k], pz[k], q, n3);
                else solo_accumulate<1>(acc, px[k], py[k], pz[k], q, n3);
            }
            const double v = wave_reduce16(acc, lane);
            if ((lane & 3) == 0) s_part[wave][half * 16 + (lane >> 2)] = v;
        }
        __syncthreads();
        if (tid == 0) {
            const long long te = (long long)__builtin_amdgcn_s_memtime();
            s_clk[0] += tb - ta; s_clk[1] += tc - tb; s_clk[2] += td - tc; s_clk[3] += te - td; s_clk[4] += nq; s_clk[5] += 1;
            if (g == 0 && it < 32) {                             // per evaluation (read_debug kind 4, parity 0, row 32 + evaluation; B.partial is free while the team kernel runs)
                double* row = B.partial + ((size_t)h * kIcpMaxSplit + 32 + it) * 32;
#pragma unroll
                for (int c = 0; c < kClasses; ++c) row[c] = (double)cnt[c];
                row[8] = (double)d_lanes; row[9] = (double)d_shift; row[10] = (double)d_scatter; row[11] = (double)d_sweep; row[12] = (double)d_read;
                row[13] = s_mot[0]; row[14] = (double)(tc - tb); row[15] = (double)(te - td); row[16] = (double)(tb - ta);
                row[17] = (double)d_sw1; row[18] = (double)d_sw2; row[19] = (double)d_sw3; row[20] = (double)s_fclk[0]; row[21] = (double)s_fclk[1]; row[22] = (double)s_fclk[2]; row[23] = (double)s_fclk[3]; row[24] = (double)s_fclk[4];
            }
        }
    }
    if (tid == 0) {                                             // per member (read_debug kind 4, parity 1, row = member): cycles in exchange + finish, move + queue, search, sums; evaluations, searches, exchange alone
        double* row = B.partial + (((size_t)B.count + h) * kIcpMaxSplit + g) * 32;
        for (int a = 0; a < 7; ++a) row[a] = (double)s_clk[a];
        row[7] = (double)((long long)__builtin_amdgcn_s_memtime() - t_begin);
    }
    if (s_stop == 3) {                                          // suspended: the next launch of the round goes on from here (every member holds the same state)
        if (g == 0 && tid < 12) S.T[tid] = s_T[tid];
        if (g == 0 && tid == 0) {
            S.fit_hist[0] = s_hist[0]; S.fit_hist[1] = s_hist[1]; S.rmse_hist[0] = s_hist[2]; S.rmse_hist[1] = s_hist[3];
            S.resume_it = s_fin_i[1];                              // (the evaluation index of this finish stage)
            S.clk[0] += s_clk[0]; S.clk[1] += s_clk[1]; S.clk[2] += (long long)__builtin_amdgcn_s_memtime() - t_begin; S.clk[3] += s_clk[2]; S.clk[4] += s_clk[3];
            S.clk[5] += s_clk[5]; S.clk[6] += s_clk[4]; S.clk[7] += s_clk[6];
        }
        return;
    }
    if (tid == 0 && g == 0 && s_stop != 2) {                    // (a team that timed out leaves stop == 0: the host runs the sliced launches)
        for (int a = 0; a < 12; ++a) S.T[a] = s_T[a];
        S.T[12] = 0.0; S.T[13] = 0.0; S.T[14] = 0.0; S.T[15] = 1.0;
        S.fitness = s_fin[0]; S.rmse = s_fin[1]; S.n_corr = s_fin_i[0]; S.iterations = s_fin_i[1];
        S.fit_hist[0] = s_hist[0]; S.fit_hist[1] = s_hist[1]; S.rmse_hist[0] = s_hist[2]; S.rmse_hist[1] = s_hist[3];
        // diagnostics (shader cycles of wave 0 of member 0, which waits at the barriers for the other waves): exchange + finish, transform + queue, whole kernel, search, sums, evaluations, own searches, exchange alone
        S.clk[0] += s_clk[0]; S.clk[1] += s_clk[1]; S.clk[2] += (long long)__builtin_amdgcn_s_memtime() - t_begin; S.clk[3] += s_clk[2]; S.clk[4] += s_clk[3];
        S.clk[5] += s_clk[5]; S.clk[6] += s_clk[4]; S.clk[7] += s_clk[6];
        S.stop = 1;
    }
}

// Pipeline glue (pipeline.cpp): turns the detections kept by the on-device NMS into ICP hypotheses without a
// host round trip.  One thread per hypothesis slot: the view (rendered depth slot + camera matrix) of the
// matched template, detect = match position (linemod_and_levelup_test.py:354-367).
__global__ void k_icp_bind(const TopkSel* __restrict__ sel, const int32_t* __restrict__ nsel_status, const int32_t* __restrict__ class_base,
                           const float* __restrict__ view_K, const int32_t* __restrict__ view_valid, int num_views, IcpIn* __restrict__ in,
                           IcpState* __restrict__ st, int top_k) {
    const int h = blockIdx.x * blockDim.x + threadIdx.x;
    if (h >= top_k) return;
    IcpIn I;
    for (int k = 0; k < 9; ++k) I.mK[k] = 0.f;
    I.dx = 0; I.dy = 0; I.model_slot = 0; I.pad = 0;
    int status = kIcpNoDetection;                                // no detection for this slot
    if (nsel_status[1] == 0 && h < nsel_status[0]) {
        const TopkSel s = sel[h];
        const int base = class_base[s.class_index];
        const int v = base + s.template_id;
        status = kIcpNoView;                                     // the matched template has no rendered view
        if (base >= 0 && v >= 0 && v < num_views && view_valid[v]) {
            status = 0;
            for (int k = 0; k < 9; ++k) I.mK[k] = view_K[(size_t)v * 9 + k];
            I.dx = s.x; I.dy = s.y; I.model_slot = v;
        }
    }
    in[h] = I;
    IcpState& S = st[h];
    S = IcpState{};                                              // (a memset launch of its own cost 5 us + a gap)
    S.bbox[0] = INT_MAX; S.bbox[1] = INT_MAX; S.bbox[2] = -1; S.bbox[3] = -1;
    S.status = status;
}

void launch_icp_bind(const TopkSel* sel, const int32_t* nsel_status, const int32_t* class_base, const float* view_K,
                     const int32_t* view_valid, int num_views, IcpIn* in, IcpState* st, int top_k, hipStream_t s) {
    if (top_k <= 0) return;
    hipLaunchKernelGGL(k_icp_bind, dim3((top_k + 63) / 64), dim3(64), 0, s, sel, nsel_status, class_base, view_K, view_valid, num_views, in, st,
                       top_k);
}

// the sliced launches of evaluations [it_from, max_iter + 1]: evaluation `it` is finished (convergence test, solve, update) by the prologue
// of launch it + 1.  The first evaluations have every hypothesis at work (768 workgroups = three per CU); by the sixth most have
// converged and the ones that go on for all 30 are cut finer (their latency is what is left): 64 slices each
static int icp_slices(int count, int it) {
    const Knobs& kn = knobs();
    int splits = 768 / count;                                        // enough workgroups to cover the chip, at least ~128 source points each at typical sizes
    if (kn.icp_splits > 0) splits = kn.icp_splits;                     // tuning knob (profiles/)
    if (splits > kIcpMaxSplit) splits = kIcpMaxSplit;
    if (splits < 1) splits = 1;
    return it < kIcpFineFrom || kn.icp_splits > 0 ? splits : kIcpMaxSplit;
}

void launch_icp_evals(const IcpBuffers& B, int count, int it_from, int it_to, double max_dist, int max_iter, double rel_tol, hipStream_t s) {
    if (count <= 0) return;
    const Knobs& kn = knobs();
#ifdef LM_DIAG
    if (kn.icp_maxiter_diag >= 0) max_iter = kn.icp_maxiter_diag;            // diagnostics only (profiles/): stop after a few evaluations
#endif
    if (it_to > max_iter + 1) it_to = max_iter + 1;
    for (int it = it_from; it <= it_to; ++it) {
        // lanes per searching point: at most 8 while every point searches (the first evaluations: more lanes only multiply the
        // set-up), 16 afterwards (few searches left: their latency is what counts) — measured, profiles/r02_icp_experiments.txt
        hipLaunchKernelGGL(k_icp_eval, dim3(icp_slices(count, it), count), dim3(kSearchWG), 0, s, B, it, it > 0 ? icp_slices(count, it - 1) : 1,
                           it < kIcpFineFrom ? kn.icp_maxshift : kn.icp_maxshift_late, max_dist, max_iter, rel_tol);
    }
}

// RegistrationICP as one launch for all evaluations: a team of workgroups per hypothesis, as many as the chip holds at once (one workgroup per
// CU: the team's members wait for each other, so the whole grid must be resident) — hypotheses k_icp_team cannot hold keep stop == 0.
// Four builds, each taking the hypotheses the ones before left (stop == 0):
//   1  one source point per owner thread, the whole target cloud in LDS — clouds of a couple of thousand points per team member and
//      target clouds of <= ~2300 points: far inside its registers, every target access a plain LDS read;
//   2  one point per thread, a slab of the target cloud in LDS when it does not fit whole (a batch with such a cloud is taken whole);
//   4  two points per thread, slab — up to 1408 points per member;
//   8  five points per thread — batches so large that a team is one or two workgroups.
// A launch with nothing to take is not free (measured on the icp leg: the idle build 2 costs 3 us, the idle build 4 another 6-8), so up
// to 64 hypotheses — where the kernel deals the workgroups out by cloud size and a member holds more than 704 points only when the batch
// has more than ~170k source points — the first call launches builds 1 and 2 only; the caller tries `large` (4 and 8) on what is left
// before it goes to the sliced launches.
void launch_icp_team(const IcpBuffers& B, int count, int large, double max_dist, int max_iter, double rel_tol, hipStream_t s) {
    if (count <= 0) return;
    const Knobs& kn = knobs();
#ifdef LM_DIAG
    if (kn.icp_maxiter_diag >= 0) max_iter = kn.icp_maxiter_diag;
#endif
    static const int cus = [] {
        int dev = 0, n = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 64;
        return n;
    }();
    int team = kn.icp_team > 0 ? kn.icp_team : 16;
    if (team > cus / count) team = cus / count;
    if (team > kIcpMaxSplit) team = kIcpMaxSplit;
    if (team < 1) team = 1;
    static std::atomic<unsigned int> runs{0};                         // tags of the team's granules (see k_icp_team): unique per launch of the process, 0 = never published
    const bool dealt = count <= 64 && kn.icp_team == 0;              // (<= 64 hypotheses: the kernel deals the workgroups out itself, by cloud size)
    const dim3 grid = dealt ? dim3(cus) : dim3(team, count);
    int builds = kn.icp_builds > 0 ? kn.icp_builds : (dealt ? (large ? 4 | 8 : 1 | 2) : (large ? 0 : 1 | 2 | 4 | (team < 4 ? 8 : 0)));
    if (kn.icp_builds > 0 && large) builds = 0;
    // A SECOND LAUNCH for a cramped batch (dealt grids): most hypotheses of a batch converge within a couple of evaluations and their
    // workgroups then idle while the ones that go on for all 30 keep the team they were dealt at the start (the pipeline's 16 detections: 12
    // done after two evaluations, the launch as long as 31 evaluations of a hypothesis on 8 workgroups).  When the clouds of a batch could
    // use half as many workgroups again as the chip has, the first launch ends after evaluation kn.icp_cut_index for everyone still at work
    // (k_icp_team: the state goes to IcpState) and the second — the slab build alone, which holds whole clouds too — deals the chip out
    // among those and runs them to the end; it costs ~3 us when the batch was not cramped.  Same arithmetic, the sums of an evaluation
    // grouped by the new team size (rounding); the rule looks at cloud sizes and evaluation indices only, so a run repeats itself bit for bit.
    const int relaunch = dealt && !large && kn.icp_builds == 0 ? kn.icp_relaunch : 0;
    for (int ph = 0; ph <= relaunch; ++ph) {
        unsigned int run = (runs.fetch_add(1) + 1) & 0x3FFFFFFu;
        if (run == 0) run = (runs.fetch_add(1) + 1) & 0x3FFFFFFu;
        const int allow = ph < relaunch ? kn.icp_cut_index * (ph + 1) * (ph + 1) : 0, b = ph == 0 ? builds : 2;   // (cut indices 3, 12, 27 ...)
        if ((b & 3) == 3) {
            hipLaunchKernelGGL((k_icp_team<1, false>), grid, dim3(kSoloWG), 0, s, B, run, kn.icp_maxshift, max_dist, max_iter, rel_tol, allow, kn.icp_team_min_points);
            hipLaunchKernelGGL((k_icp_team<1, true>), grid, dim3(kSoloWG), 0, s, B, run, kn.icp_maxshift, max_dist, max_iter, rel_tol, allow, kn.icp_team_min_points);
        }
        else if (b & 1) hipLaunchKernelGGL((k_icp_team<1, false>), grid, dim3(kSoloWG), 0, s, B, run, kn.icp_maxshift, max_dist, max_iter, rel_tol, allow, kn.icp_team_min_points);
        else if (b & 2) hipLaunchKernelGGL((k_icp_team<1, true>), grid, dim3(kSoloWG), 0, s, B, run, kn.icp_maxshift, max_dist, max_iter, rel_tol, allow, kn.icp_team_min_points);
        if (b & 4) hipLaunchKernelGGL((k_icp_team<2, true>), grid, dim3(kSoloWG), 0, s, B, run, kn.icp_maxshift, max_dist, max_iter, rel_tol, allow, kn.icp_team_min_points);
        if (b & 8) hipLaunchKernelGGL((k_icp_team<5, true>), grid, dim3(kSoloWG), 0, s, B, run, kn.icp_maxshift, max_dist, max_iter, rel_tol, allow, kn.icp_team_min_points);
    }
}

void launch_icp_pipeline(const IcpBuffers& B, int count, int W, int H, int flags, double voxel, double max_dist, int max_iter,
                         double rel_tol, int knn, int solo_from, hipStream_t s) {
    if (count <= 0) return;
    const Knobs& kn = knobs();
#ifdef LM_DIAG
    if (kn.icp_maxiter_diag >= 0) max_iter = kn.icp_maxiter_diag;            // diagnostics only (profiles/): stop after a few evaluations
#endif
    const int scene_mode = flags & 1;
    if (!(flags & 0x100) || !kn.icp_wide_sort) hipLaunchKernelGGL(k_icp_bbox, dim3(32, count), dim3(256), 0, s, B, W, H);   // (0x100: every slot's box is in model_bbox)
    if (kn.icp_wide_sort) hipLaunchKernelGGL(k_icp_points_fused, dim3(kIcpStrips, count), dim3(kPtsWG), 0, s, B, W, H, flags);
    else {
        hipLaunchKernelGGL(k_icp_points<false>, dim3(kIcpStrips, count), dim3(kPtsWG), 0, s, B, W, H, flags);
        hipLaunchKernelGGL(k_icp_points<true>, dim3(kIcpStrips, count), dim3(kPtsWG), 0, s, B, W, H, flags);
    }
    // voxel down-sampling and the search grid by kIcpSortGroups workgroups per cloud; the one-workgroup kernels behind them take what those
    // left (IcpState::vox_done / grid_done) and cost ~2 us when there is nothing
    if (kn.icp_wide_sort) hipLaunchKernelGGL(k_icp_voxel_wide, dim3(kIcpSortGroups, count, scene_mode ? 2 : 1), dim3(kWG), 0, s, B, flags, voxel);
    hipLaunchKernelGGL(k_icp_voxel, dim3(count, scene_mode ? 2 : 1), dim3(kWG), 0, s, B, flags, voxel);
    if (kn.icp_wide_sort) hipLaunchKernelGGL(k_icp_grid_wide, dim3(kIcpSortGroups, count), dim3(kWG), 0, s, B, flags);
    hipLaunchKernelGGL(k_icp_grid, dim3(count), dim3(kWG), 0, s, B, flags);
    hipLaunchKernelGGL(k_icp_knn, dim3(kn.knn_blocks > 0 ? kn.knn_blocks : (count <= 32 ? 64 : 32), count), dim3(kKnnWG), 0, s, B, knn);
    hipLaunchKernelGGL(k_icp_knn_far, dim3(kKnnFarBlocks, count), dim3(512), 0, s, B, knn);
    hipLaunchKernelGGL(k_icp_normals, dim3(count <= 32 ? 64 : 16, count), dim3(256), 0, s, B);
    if (solo_from != 0) {                                            // sliced launches only
        launch_icp_evals(B, count, 0, max_iter + 1, max_dist, max_iter, rel_tol, s);
        return;
    }
    launch_icp_team(B, count, 0, max_dist, max_iter, rel_tol, s);
}

}  // namespace lm
