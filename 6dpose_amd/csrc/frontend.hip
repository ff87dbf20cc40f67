// Front end of Detector::match / addTemplate on gfx950: quantised gradient orientations
// (LL.cpp:350-505), quantised depth normals (LL.cpp:729-819), pyramid steps (LL.cpp:557-581,
// 857-880) and spread -> response -> linearise (LL.cpp:1026-1243) written straight into the
// linear-memory layout the similarity kernels read.
//
// Everything here is HBM/L2-bound byte work on <= a few MB per frame; kernels are one thread per
// output element with coalesced stores.  Parity-critical float steps use the *_rn intrinsics so
// that no FMA contraction or reassociation can occur (the reference is x86-64 SSE2 scalar float,
// OpenCV semantics per SURVEY Appendix A).
#include "lm_kernels.h"
#include "knobs.h"

namespace lm {

__constant__ uint8_t c_normal_lut[400];   // NORMAL_LUT[.][y][x] (z-independent), normal_lut.i

void upload_normal_lut(const uint8_t lut400[400]) {
    (void)hipMemcpyToSymbol(HIP_SYMBOL(c_normal_lut), lut400, 400);
}

// n / d by a multiplier the host prepares (lm_kernels.h, FeJob): exact while n * d < 2^32; m = 0 stands for d = 1
static __host__ __device__ __forceinline__ uint32_t div_magic(uint32_t d) { return d <= 1 ? 0u : (uint32_t)((0x100000000ull + d - 1) / d); }
// n / d with m = div_magic(d) = ceil(2^32 / d): umulhi(n, m) is floor(n / d) or one more (the excess n e / (d 2^32), e = m d - 2^32 < d, stays
// below 1 for every 32-bit n), so one comparison makes it exact for ANY n — an 8000 x 6000 frame puts n d past 2^32, where the bare product
// was wrong for the last blocks of a job.  The comparison itself cannot wrap: q d <= n + d, and every caller passes non-negative `int`
// values (block / pixel / phase indices and image sizes), so n + d < 2^31 + 2^31 = 2^32 — asserted in the debug build below.
static __device__ __forceinline__ uint32_t fast_div(uint32_t n, uint32_t m, uint32_t d) {
#ifdef LM_DIAG
    if ((n | d) >> 31) __builtin_trap();                   // n, d < 2^31: q * d <= n + d < 2^32
#endif
    if (!m) return n;
    const uint32_t q = __umulhi(n, m);
    return q - (q * d > n ? 1u : 0u);
}

static __device__ __forceinline__ int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

// Full adder on 32 bit positions at once (two v_bitop3_b32: parity 0x96, majority 0xE8).  The 3x3 majority vote of the colour chain and the
// 5x5 median of the normals count, per pixel, how many taps of a window satisfy a predicate: with one byte per tap (a bit per predicate) and
// four neighbouring pixels per word, a tree of these adders counts 8 predicates of 4 pixels at once.
static __device__ __forceinline__ void full_add(uint32_t a, uint32_t b, uint32_t c, uint32_t& sum, uint32_t& carry) {
    sum = __builtin_amdgcn_bitop3_b32(a, b, c, 0x96);
    carry = __builtin_amdgcn_bitop3_b32(a, b, c, 0xE8);
}

// ---- cv::phase(dx, dy, angle, true): OpenCV fastAtan2 polynomial (LL.cpp:423; Appendix A.3) -----
static __device__ __forceinline__ float fast_atan2_deg(float y, float x) {
    const float scale = (float)(180.0 / 3.14159265358979323846);
    const float p1 = __fmul_rn(0.9997878412794807f, scale);
    const float p3 = __fmul_rn(-0.3258083974640975f, scale);
    const float p5 = __fmul_rn(0.1555786518463281f, scale);
    const float p7 = __fmul_rn(-0.04432655554792128f, scale);
    const float eps = 2.220446049250313e-16f;   // (float)DBL_EPSILON
    float ax = fabsf(x), ay = fabsf(y), a, c, c2;
    if (ax >= ay) {
        c = __fdiv_rn(ay, __fadd_rn(ax, eps));
        c2 = __fmul_rn(c, c);
        a = __fmul_rn(__fadd_rn(__fmul_rn(__fadd_rn(__fmul_rn(__fadd_rn(__fmul_rn(p7, c2), p5), c2), p3), c2), p1), c);
    } else {
        c = __fdiv_rn(ax, __fadd_rn(ay, eps));
        c2 = __fmul_rn(c, c);
        a = __fsub_rn(90.f,
                      __fmul_rn(__fadd_rn(__fmul_rn(__fadd_rn(__fmul_rn(__fadd_rn(__fmul_rn(p7, c2), p5), c2), p3), c2), p1), c));
    }
    if (x < 0.f) a = __fsub_rn(180.f, a);
    if (y < 0.f) a = __fsub_rn(360.f, a);
    return a;
}

// ---- the colour chain in one launch (LL.cpp:367-504), one 32x16 output tile per workgroup, every intermediate in LDS:
//   cv::GaussianBlur(8UC3, 7x7, sigma 0, BORDER_REPLICATE) (LL.cpp:367; Appendix A.1): separable fixed point, weights
//     [8,28,56,72,56,28,8], out = (sum_ij + 32768) >> 16;
//   Sobel 3x3 (replicate) per channel, strongest channel (first that is >= the other two, LL.cpp:395-412), cv::phase,
//     q16 = interior ? (saturate_u8(rint(angle*16/360)) & 7) : 0 (LL.cpp:368-455);
//   hysteresisGradient's 3x3 majority vote (LL.cpp:457-504).
// Each stage keeps its own border rule by evaluating a stage at the CLAMPED position of the pixel the next stage asks
// for (blur and Sobel replicate the border: smoothed(clamp(p)), not a blur centred outside the image), which makes the
// fused kernel bit-identical to running the stages as separate whole-image passes (the oracle does exactly that).
constexpr int kCTX = 32, kCTY = 16;           // output tile (32 x 32 was measured in round 6: 12 % less halo work, but 29 KB of LDS per workgroup leave a CU 5 of them instead of 7: 52.7 us of kernels per frame against 50.5)
constexpr int kQRowWords = (kCTX + 2 + 3) / 4;   // dwords per row of the vote's / the median's byte planes (34 and 36 bytes -> 9)
static __device__ __forceinline__ void color_quant_body(const int bx, const int by, const uint8_t* __restrict__ rgb, float* __restrict__ mag,
                                                        uint8_t* __restrict__ onehot, int W, int H, float thr_sq) {
    // virtual coordinates: tile origin (x0, y0); halos: rgb 5, blurred 2, quantised 1.
    // The three channels of a pixel travel as ONE LDS word (R | G << 8 | B << 16) and R, B share the multiply-adds of the horizontal
    // pass (7 taps x 255 x 72 <= 65280 fits 16 bits): the stage was bound by its byte-wide LDS reads (21 + 21 + 24 per output pixel;
    // now 7 + 7 + 8).  The integer results are the same.
    __shared__ uint32_t s_rgb[kCTY + 10][kCTX + 10];
    __shared__ uint2 s_tmp[kCTY + 10][kCTX + 4];           // horizontal pass at columns clamp(x0-2 .. x0+TX+1), all halo rows: {R | B << 16, G}
    __shared__ uint32_t s_sm[kCTY + 4][kCTX + 4];           // smoothed at clamp(y0-2 ..), clamp(x0-2 ..), packed like s_rgb
    __shared__ uint32_t s_q4[(kCTY + 2) * kQRowWords];      // ONE-HOT of the 16-bin code & 7 at y0-1 .., x0-1 .. (bin 0 outside the interior), a byte per pixel, rows of kQRowWords dwords
    __shared__ __attribute__((aligned(16))) float s_mag[kCTY][kCTX];
    const int x0 = bx * kCTX, y0 = by * kCTY, tid = threadIdx.x;
    const uint32_t w7[7] = {8, 28, 56, 72, 56, 28, 8};
    for (int i = tid; i < (kCTY + 10) * (kCTX + 10); i += 256) {
        const int ty = i / (kCTX + 10), tx = i - ty * (kCTX + 10);
        const uint8_t* p = rgb + ((size_t)clampi(y0 - 5 + ty, 0, H - 1) * W + clampi(x0 - 5 + tx, 0, W - 1)) * 3;
        s_rgb[ty][tx] = (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16);
    }
    __syncthreads();
    // horizontal pass: rows = all halo rows (virtual y0-5+ty, already clamped by the load), columns cx = clamp(x0-2+tx)
    for (int i = tid; i < (kCTY + 10) * (kCTX + 4); i += 256) {
        const int ty = i / (kCTX + 4), tx = i - ty * (kCTX + 4);
        const int cx = clampi(x0 - 2 + tx, 0, W - 1) - (x0 - 5);           // tile column of the clamped centre
        uint32_t arb = 0, ag = 0;
#pragma unroll
        for (int k = 0; k < 7; ++k) {
            const uint32_t px = s_rgb[ty][cx + k - 3];
            arb += w7[k] * (px & 0x00FF00FFu);
            ag += w7[k] * ((px >> 8) & 0xFFu);
        }
        s_tmp[ty][tx] = make_uint2(arb, ag);
    }
    __syncthreads();
    // vertical pass at rows cy = clamp(y0-2+ty)
    for (int i = tid; i < (kCTY + 4) * (kCTX + 4); i += 256) {
        const int ty = i / (kCTX + 4), tx = i - ty * (kCTX + 4);
        const int cy = clampi(y0 - 2 + ty, 0, H - 1) - (y0 - 5);           // tile row of the clamped centre
        uint32_t ar = 0, ag = 0, ab = 0;
#pragma unroll
        for (int k = 0; k < 7; ++k) {
            const uint2 t = s_tmp[cy + k - 3][tx];
            ar += w7[k] * (t.x & 0xFFFFu); ab += w7[k] * (t.x >> 16); ag += w7[k] * t.y;
        }
        s_sm[ty][tx] = ((ar + 32768u) >> 16) | (((ag + 32768u) >> 16) << 8) | (((ab + 32768u) >> 16) << 16);
    }
    __syncthreads();
    // Sobel + strongest channel + quantisation at (y0-1+ty, x0-1+tx); s_sm index of virtual (vy, vx) = (vy - y0 + 2, vx - x0 + 2)
    for (int i = tid; i < (kCTY + 2) * (kCTX + 2); i += 256) {
        const int ty = i / (kCTX + 2), tx = i - ty * (kCTX + 2);
        const int y = y0 - 1 + ty, x = x0 - 1 + tx;
        uint8_t q = 0;
        if (x >= 0 && y >= 0 && x < W && y < H) {
            const int sy = ty + 1, sx = tx + 1;                             // this pixel in s_sm; its neighbours are the clamped ones by construction
            int dxs[3], dys[3], mags[3];
            {
                // the 3x3 neighbourhood as packed words; R and B ride one 32-bit lane each way (a + 2 b + c <= 1020 fits 16 bits)
                const uint32_t p00 = s_sm[sy - 1][sx - 1], p01 = s_sm[sy - 1][sx], p02 = s_sm[sy - 1][sx + 1];
                const uint32_t p10 = s_sm[sy][sx - 1], p12 = s_sm[sy][sx + 1];
                const uint32_t p20 = s_sm[sy + 1][sx - 1], p21 = s_sm[sy + 1][sx], p22 = s_sm[sy + 1][sx + 1];
                auto rb = [](uint32_t p) { return p & 0x00FF00FFu; };
                auto gg = [](uint32_t p) { return (p >> 8) & 0xFFu; };
                const uint32_t xp_rb = rb(p02) + 2u * rb(p12) + rb(p22), xn_rb = rb(p00) + 2u * rb(p10) + rb(p20);
                const uint32_t yp_rb = rb(p20) + 2u * rb(p21) + rb(p22), yn_rb = rb(p00) + 2u * rb(p01) + rb(p02);
                const int xp_g = (int)(gg(p02) + 2u * gg(p12) + gg(p22)), xn_g = (int)(gg(p00) + 2u * gg(p10) + gg(p20));
                const int yp_g = (int)(gg(p20) + 2u * gg(p21) + gg(p22)), yn_g = (int)(gg(p00) + 2u * gg(p01) + gg(p02));
                dxs[0] = (int)(xp_rb & 0xFFFFu) - (int)(xn_rb & 0xFFFFu); dys[0] = (int)(yp_rb & 0xFFFFu) - (int)(yn_rb & 0xFFFFu);
                dxs[1] = xp_g - xn_g;                                     dys[1] = yp_g - yn_g;
                dxs[2] = (int)(xp_rb >> 16) - (int)(xn_rb >> 16);         dys[2] = (int)(yp_rb >> 16) - (int)(yn_rb >> 16);
#pragma unroll
                for (int c = 0; c < 3; ++c) mags[c] = dxs[c] * dxs[c] + dys[c] * dys[c];
            }
            int bdx, bdy, bmag;
            if (mags[0] >= mags[1] && mags[0] >= mags[2]) { bdx = dxs[0]; bdy = dys[0]; bmag = mags[0]; }
            else if (mags[1] >= mags[0] && mags[1] >= mags[2]) { bdx = dxs[1]; bdy = dys[1]; bmag = mags[1]; }
            else { bdx = dxs[2]; bdy = dys[2]; bmag = mags[2]; }
            if (ty >= 1 && ty <= kCTY && tx >= 1 && tx <= kCTX) s_mag[ty - 1][tx - 1] = (float)bmag;
            if (x > 0 && y > 0 && x < W - 1 && y < H - 1) {
                const float ang = fast_atan2_deg((float)bdy, (float)bdx);
                const float v = rintf(__fmul_rn(ang, (float)(16.0 / 360.0)));   // cvRound: half to even
                const int iv = v < 0.f ? 0 : (v > 255.f ? 255 : (int)v);
                q = (uint8_t)(iv & 7);
            }
        }
        reinterpret_cast<uint8_t*>(s_q4)[ty * (kQRowWords * 4) + tx] = (uint8_t)(1u << q);
    }
    __syncthreads();
    // hysteresisGradient's vote (LL.cpp:457-504): the orientation that at least 5 of the 9 pixels of the 3x3 window carry (a strict maximum of
    // >= 5 votes of 9 is that orientation; fewer: nothing).  Four pixels per thread: the 9 taps are words of four one-hot bytes (two aligned
    // reads + v_alignbyte per row), a carry-save tree counts all 8 orientations of the 4 pixels at once, count >= 5 <=> b3 | b2 & (b1 | b0).
    // (Was: 9 byte reads + a packed histogram + an 8-way arg-max per pixel, 58 VALU; now ~12.)
    if (tid < kCTY * kCTX / 4) {
        const int ty = tid >> 3, c = tid & 7;
        const int y = y0 + ty, x = x0 + 4 * c;
        if (x < W && y < H) {
            uint32_t sr[3], kr[3];
#pragma unroll
            for (int dy = 0; dy < 3; ++dy) {
                const uint32_t lo = s_q4[(ty + dy) * kQRowWords + c], hi = s_q4[(ty + dy) * kQRowWords + c + 1];
                full_add(lo, __builtin_amdgcn_alignbyte(hi, lo, 1), __builtin_amdgcn_alignbyte(hi, lo, 2), sr[dy], kr[dy]);
            }
            uint32_t b0, k3, u, k4;
            full_add(sr[0], sr[1], sr[2], b0, k3);            // weight 1 done; weight 2: kr[0..2], k3
            full_add(kr[0], kr[1], kr[2], u, k4);
            const uint32_t b1 = u ^ k3, k5 = u & k3;          // weight 4: k4 k5
            const uint32_t b2 = k4 ^ k5, b3 = k4 & k5;
            const uint32_t win = b3 | (b2 & (b1 | b0));
            const size_t o = (size_t)y * W + x;
            const bool yin = y > 0 && y < H - 1;
            uint32_t keep = 0;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float m = s_mag[ty][4 * c + q];
                if (yin && x + q > 0 && x + q < W - 1 && m > thr_sq) keep |= 0xFFu << (8 * q);
                if (mag && x + q < W) mag[o + q] = m;         // (null for the frames of a match: only addTemplate reads the magnitudes, LL.cpp:589-643)
            }
            const uint32_t res = win & keep;
            if (x + 3 < W && ((reinterpret_cast<uintptr_t>(onehot) + o) & 3) == 0) *reinterpret_cast<uint32_t*>(onehot + o) = res;
            else {
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    if (x + q < W) onehot[o + q] = (uint8_t)(res >> (8 * q));
            }
        }
    }
}

__global__ void __launch_bounds__(256)
k_color_quant(const uint8_t* __restrict__ rgb, float* __restrict__ mag, uint8_t* __restrict__ onehot, int W, int H, float thr_sq) {
    color_quant_body(blockIdx.x, blockIdx.y, rgb, mag, onehot, W, H, thr_sq);
}

void launch_color_quant(const uint8_t* rgb, float* mag, uint8_t* onehot, int W, int H, float thr_sq, hipStream_t s) {
    hipLaunchKernelGGL(k_color_quant, dim3((W + kCTX - 1) / kCTX, (H + kCTY - 1) / kCTY), dim3(256), 0, s, rgb, mag, onehot, W, H, thr_sq);
}

// ---- cv::pyrDown 8UC3 (LL.cpp:566; Appendix A.6): 5x5 [1 4 6 4 1], REFLECT_101, (sum+128)>>8 ----
static __device__ __forceinline__ int reflect101(int p, int n) {
    if (n == 1) return 0;
    while (p < 0 || p >= n) p = p < 0 ? -p : 2 * n - 2 - p;
    return p;
}

static __device__ __forceinline__ void pyrdown_body(const int bx, const int by, const uint8_t* __restrict__ src, uint8_t* __restrict__ dst, int W, int H,
                                                    int Wo, int Ho) {
    int i = bx * 256 + threadIdx.x, y = by;   // i over Wo*3
    if (i >= Wo * 3) return;
    int x = i / 3, c = i - x * 3;
    const int w[5] = {1, 4, 6, 4, 1};
    int s = 0;
#pragma unroll
    for (int j = 0; j < 5; ++j) {
        const uint8_t* row = src + (size_t)reflect101(2 * y + j - 2, H) * W * 3;
        int rs = 0;
#pragma unroll
        for (int k = 0; k < 5; ++k) rs += w[k] * row[reflect101(2 * x + k - 2, W) * 3 + c];
        s += w[j] * rs;
    }
    dst[(size_t)y * Wo * 3 + i] = (uint8_t)((s + 128) >> 8);
}

__global__ void __launch_bounds__(256) k_pyrdown_rgb(const uint8_t* __restrict__ src, uint8_t* __restrict__ dst, int W, int H, int Wo, int Ho) {
    pyrdown_body(blockIdx.x, blockIdx.y, src, dst, W, H, Wo, Ho);
}

void launch_pyrdown_rgb(const uint8_t* src, uint8_t* dst, int W, int H, hipStream_t s) {
    int Wo = W / 2, Ho = H / 2;
    hipLaunchKernelGGL(k_pyrdown_rgb, dim3((Wo * 3 + 255) / 256, Ho), dim3(256), 0, s, src, dst, W, H, Wo, Ho);
}

// ---- quantizedNormals (LL.cpp:729-817; Appendix A.7) + cv::medianBlur(5) (LL.cpp:818), replicate border ----------
// One launch: the raw normals of a 32x16 tile (+2 halo, evaluated at the clamped positions medianBlur's replicate
// border asks for) live in LDS.  Values are 0 or one-hot -> 9 ranks; the median is the 13th smallest of 25, found with
// packed 5-bit counters in a 64-bit word.
static __device__ __forceinline__ uint8_t normal_at(const uint16_t* __restrict__ depth, int x, int y, int W, int H, int dist_thr, int diff_thr) {
    const int r = 5;
    uint8_t res = 0;
    if (x >= r && y >= r && x < W - r - 1 && y < H - r - 1) {
        const uint16_t* p = depth + (size_t)y * W + x;
        long long d = p[0];
        if (d < dist_thr) {
            const int oi[8] = {-r, 0, r, -r, r, -r, 0, r};
            const int oj[8] = {-r, -r, -r, 0, 0, r, r, r};
            float nx, ny, nz;
            if (diff_thr <= 150) {
                // the reference's `long` arithmetic (LL.cpp:701-819) in 32 bits: |delta| < diff_thr on every tap that counts, so
                // |b| <= 40 diff_thr, |ddx|, |ddy| <= 12000 diff_thr, 1150 |ddx| <= 1.38e7 diff_thr < 2^31 and det * d <= 22500 * 65535
                // < 2^31 — the same integers, a third of the instructions (64-bit multiplies are emulated)
                int A0 = 0, A1 = 0, A3 = 0, b0 = 0, b1 = 0;
                const int di = (int)d;
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const int delta = (int)p[oj[k] * W + oi[k]] - di;
                    const int ad = delta < 0 ? -delta : delta;
                    if (ad < diff_thr) {
                        A0 += oi[k] * oi[k]; A1 += oi[k] * oj[k]; A3 += oj[k] * oj[k];
                        b0 += oi[k] * delta; b1 += oj[k] * delta;
                    }
                }
                const int det = A0 * A3 - A1 * A1;
                const int ddx = A3 * b0 - A1 * b1;
                const int ddy = -A1 * b0 + A0 * b1;
                nx = (float)(1150 * ddx); ny = (float)(1150 * ddy); nz = (float)(-det * di);
            } else {
                long long A0 = 0, A1 = 0, A3 = 0, b0 = 0, b1 = 0;
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    long long delta = (long long)p[oj[k] * W + oi[k]] - d;
                    long long ad = delta < 0 ? -delta : delta;
                    long long f = ad < diff_thr ? 1 : 0;
                    long long fi = f * oi[k], fj = f * oj[k];
                    A0 += fi * oi[k]; A1 += fi * oj[k]; A3 += fj * oj[k];
                    b0 += fi * delta; b1 += fj * delta;
                }
                long long det = A0 * A3 - A1 * A1;
                long long ddx = A3 * b0 - A1 * b1;
                long long ddy = -A1 * b0 + A0 * b1;
                nx = (float)(1150 * ddx); ny = (float)(1150 * ddy); nz = (float)(-det * d);
            }
            float ss = __fadd_rn(__fadd_rn(__fmul_rn(nx, nx), __fmul_rn(ny, ny)), __fmul_rn(nz, nz));
            float sq = __fsqrt_rn(ss);
            if (sq > 0.f) {
                float inv = __fdiv_rn(1.0f, sq);
                nx = __fmul_rn(nx, inv); ny = __fmul_rn(ny, inv); nz = __fmul_rn(nz, inv);
                int v1 = (int)__fadd_rn(__fmul_rn(nx, 10.f), 10.f);
                int v2 = (int)__fadd_rn(__fmul_rn(ny, 10.f), 10.f);
                // v3 = (int)(nz*20+20) only selects the (z-independent) table plane; an index of 20 in
                // x or y continues into the next row exactly as the reference's flat memory read does.
                int flat = (v2 * 20 + v1) % 400;
                if (flat < 0) flat += 400;
                res = c_normal_lut[flat];
            }
        }
    }
    return res;
}

static __device__ __forceinline__ void normals_median_body(const int bx, const int by, const uint16_t* __restrict__ depth, uint8_t* __restrict__ raw,
                                                           uint8_t* __restrict__ med, int W, int H, int dist_thr, int diff_thr) {
    __shared__ uint32_t s_m[(kCTY + 4) * kQRowWords];        // a byte per pixel of the tile + 2: bit k = (rank of the raw normal <= k), rank 0 = none, k + 1 = 1 << k
    const int x0 = bx * kCTX, y0 = by * kCTY, tid = threadIdx.x;
    for (int i = tid; i < (kCTY + 4) * (kCTX + 4); i += 256) {
        const int ty = i / (kCTX + 4), tx = i - ty * (kCTX + 4);
        const int y = clampi(y0 - 2 + ty, 0, H - 1), x = clampi(x0 - 2 + tx, 0, W - 1);
        const uint8_t v = normal_at(depth, x, y, W, H, dist_thr, diff_thr);
        reinterpret_cast<uint8_t*>(s_m)[ty * (kQRowWords * 4) + tx] = (uint8_t)(0xFFu << (32 - __clz((int)v)));   // (__clz(0) = 32)
        if (ty >= 2 && ty < kCTY + 2 && tx >= 2 && tx < kCTX + 2 && y0 - 2 + ty < H && x0 - 2 + tx < W) raw[(size_t)y * W + x] = v;
    }
    __syncthreads();
    // cv::medianBlur(5) of values that are 0 or one-hot = the 13th smallest of 25 ranks: median rank <= k <=> at least 13 taps have rank <= k.
    // Four pixels per thread; a tap word holds the 8 predicates "rank <= k" of 4 neighbouring pixels, a carry-save tree (20 full + 2 half
    // adders) counts them over the 25 taps, count >= 13 <=> b4 | b3 & b2 & (b1 | b0).  The predicates of a pixel are monotone in k, so its byte
    // of the result is 0xFF << rank and the value 1 << (rank - 1) (0 for rank 0) is its lowest set bit (of the byte | 0x100) >> 1.
    // (Was: 25 byte reads into packed 5-bit counters of a 64-bit word + a 9-step scan per pixel, 251 VALU; now ~26.)
    if (tid < kCTY * kCTX / 4) {
        const int ty = tid >> 3, c = tid & 7;
        const int y = y0 + ty, x = x0 + 4 * c;
        if (x < W && y < H) {
            // a row at a time (five taps -> a 3-bit count: 2 full + 1 half adder), added into the bit-sliced total by a ripple adder: a few more
            // operations than one tree over the 25 taps (62 against 44) but a dozen live registers instead of 30 — the tree cost k_fe_stage a wave of occupancy
            uint32_t b0 = 0, b1 = 0, b2 = 0, b3 = 0, b4 = 0;
#pragma unroll
            for (int dy = 0; dy < 5; ++dy) {
                const uint32_t lo = s_m[(ty + dy) * kQRowWords + c], hi = s_m[(ty + dy) * kQRowWords + c + 1];
                uint32_t sa, ka, r0, kb;
                full_add(lo, __builtin_amdgcn_alignbyte(hi, lo, 1), __builtin_amdgcn_alignbyte(hi, lo, 2), sa, ka);
                full_add(sa, __builtin_amdgcn_alignbyte(hi, lo, 3), hi, r0, kb);
                const uint32_t r1 = ka ^ kb, r2 = ka & kb;
                if (dy == 0) { b0 = r0; b1 = r1; b2 = r2; }
                else {
                    const uint32_t k0 = b0 & r0; b0 ^= r0;
                    uint32_t k1, k2;
                    full_add(b1, r1, k0, b1, k1);
                    full_add(b2, r2, k1, b2, k2);
                    const uint32_t k3 = b3 & k2; b3 ^= k2;
                    b4 ^= k3;
                }
            }
            const uint32_t ge = b4 | (b3 & b2 & (b1 | b0));
            uint32_t out = 0;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const uint32_t B = ((ge >> (8 * q)) & 0xFFu) | 0x100u;
                out |= ((B & (0u - B)) >> 1) << (8 * q);
            }
            const size_t o = (size_t)y * W + x;
            if (x + 3 < W && ((reinterpret_cast<uintptr_t>(med) + o) & 3) == 0) *reinterpret_cast<uint32_t*>(med + o) = out;
            else {
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    if (x + q < W) med[o + q] = (uint8_t)(out >> (8 * q));
            }
        }
    }
}

__global__ void __launch_bounds__(256)
k_normals_median(const uint16_t* __restrict__ depth, uint8_t* __restrict__ raw, uint8_t* __restrict__ med, int W, int H, int dist_thr,
                 int diff_thr) {
    normals_median_body(blockIdx.x, blockIdx.y, depth, raw, med, W, H, dist_thr, diff_thr);
}

void launch_normals_fused(const uint16_t* depth, uint8_t* raw, uint8_t* med, int W, int H, int dist_thr, int diff_thr, hipStream_t s) {
    hipLaunchKernelGGL(k_normals_median, dim3((W + kCTX - 1) / kCTX, (H + kCTY - 1) / kCTY), dim3(256), 0, s, depth, raw, med, W, H, dist_thr,
                       diff_thr);
}

// cv::resize(INTER_NEAREST) to (cols/2, rows/2) (LL.cpp:576, 867, 877) = pixel (2y, 2x)
static __device__ __forceinline__ void nn_down2_body(const int bx, const int by, const uint8_t* __restrict__ src, uint8_t* __restrict__ dst, int W, int Wo) {
    int x = bx * 256 + threadIdx.x, y = by;
    if (x >= Wo) return;
    dst[(size_t)y * Wo + x] = src[(size_t)(2 * y) * W + 2 * x];
}
__global__ void __launch_bounds__(256) k_nn_down2(const uint8_t* __restrict__ src, uint8_t* __restrict__ dst, int W, int Wo) {
    nn_down2_body(blockIdx.x, blockIdx.y, src, dst, W, Wo);
}

void launch_nn_down2(const uint8_t* src, uint8_t* dst, int W, int H, hipStream_t s) {
    int Wo = W / 2, Ho = H / 2;
    hipLaunchKernelGGL(k_nn_down2, dim3((Wo + 255) / 256, Ho), dim3(256), 0, s, src, dst, W, Wo);
}

// ---- spread (LL.cpp:1094-1109) fused with computeResponseMaps (LL.cpp:1134-1203, active SIMILARITY_LUT
// :1121 in closed form) and linearize (LL.cpp:1215-1243), both modalities of a level in ONE launch ------
// One thread per linear-memory position (phase, decimated index) and modality (blockIdx.z): OR the T x T
// quantised pixels below / right of it (zero beyond the edges; mask applied as quantize() does), derive the
// 8 responses (4 / 1 / 0), store to LM[label][phase][idx] — consecutive lanes write consecutive bytes of
// each label's plane.  Levels below the top also get the "strip" copy the refinement kernel gathers from:
// every plane cut into 16-column strips stored strip-major ([strip][row][16 B]), so that a 16x16 window
// touches 2 strips x 256 contiguous bytes instead of 16 rows x 1 cache line.  (The frame is a few hundred
// KB: the T*T byte reads per thread hit L1/L2; what counts here is one launch instead of four.)
// OR of the T x T quantised pixels at and right / below (x, y) (zero beyond the edges; the mask applied as quantize() does): spread, LL.cpp:1094-1109.
// kT > 0: T known at compile time — the rows' loads are all issued before the first is waited for (with a run-time T the compiler
// emitted load, wait, OR per row: T dependent L2 round trips per thread were most of the linear-memory launch).
template <int kT>
static __device__ __forceinline__ uint32_t spread_or_t(const LmJob& J, const int x, const int y, const int W, const int H, const int Trt) {
    const int T = kT > 0 ? kT : Trt;
    uint32_t v = 0;
    const int ye = y + T < H ? y + T : H, xe = x + T < W ? x + T : W;
    if (!J.mask && x + T <= W && (kT == 0 || y + T <= H)) {
        // the T pixels of a row as (unaligned) dwords
        uint32_t acc = 0;
        if (kT > 0) {
            uint32_t w[kT > 0 ? kT * ((kT + 3) / 4) : 1];
#pragma unroll
            for (int r = 0; r < kT; ++r) {
                const uint8_t* row = J.quant + (size_t)(y + r) * W + x;
#pragma unroll
                for (int c = 0; c < (kT + 3) / 4; ++c) {
                    if (4 * c + 4 <= kT) __builtin_memcpy(&w[r * ((kT + 3) / 4) + c], row + 4 * c, 4);
                    else {                                               // the last 1-3 pixels of the row
                        uint32_t t = 0;
#pragma unroll
                        for (int b = 4 * c; b < kT; ++b) t |= row[b];
                        w[r * ((kT + 3) / 4) + c] = t;
                    }
                }
            }
#pragma unroll
            for (int i = 0; i < kT * ((kT + 3) / 4); ++i) acc |= w[i];
        } else {
            for (int r = y; r < ye; ++r) {
                const uint8_t* row = J.quant + (size_t)r * W + x;
                int c = 0;
                for (; c + 4 <= T; c += 4) { uint32_t w; __builtin_memcpy(&w, row + c, 4); acc |= w; }
                for (; c < T; ++c) acc |= row[c];
            }
        }
        v = (acc | (acc >> 8) | (acc >> 16) | (acc >> 24)) & 0xFFu;
    } else {
        for (int r = y; r < ye; ++r) {
            const uint8_t* row = J.quant + (size_t)r * W;
            if (J.mask) {
                const uint8_t* mrow = J.mask + (size_t)r * W;
                for (int c = x; c < xe; ++c) v |= mrow[c] ? (uint32_t)row[c] : 0u;
            } else {
                for (int c = x; c < xe; ++c) v |= row[c];
            }
        }
    }
    return v;
}
static __device__ __forceinline__ uint32_t spread_or(const LmJob& J, const int x, const int y, const int W, const int H, const int T) {
    switch (T) {                                                         // wave-uniform
        case 2: return spread_or_t<2>(J, x, y, W, H, T);
        case 4: return spread_or_t<4>(J, x, y, W, H, T);
        case 5: return spread_or_t<5>(J, x, y, W, H, T);
        case 8: return spread_or_t<8>(J, x, y, W, H, T);
        default: return spread_or_t<0>(J, x, y, W, H, T);
    }
}

static __device__ __forceinline__ void build_lm_body(const int bx, const int by, const LmJob& J, int W, int H, int T, int Wd, int Hd, int NS, uint32_t m_wd,
                                                     uint32_t m_t) {
    int idx = bx * 256 + threadIdx.x;                    // decimated raster index
    int phase = by;                                      // r_start*T + c_start
    int npos = Wd * Hd;
    if (idx >= npos) return;
    int ry = (int)fast_div((uint32_t)idx, m_wd, (uint32_t)Wd), rx = idx - ry * Wd;
    int rs = (int)fast_div((uint32_t)phase, m_t, (uint32_t)T), cs = phase - rs * T;
    int y = ry * T + rs, x = rx * T + cs;
    const uint32_t v = spread_or(J, x, y, W, H, T);
    uint32_t adj = ((v << 1) | (v >> 7) | (v >> 1) | (v << 7)) & 0xFFu;
    size_t plane = (size_t)T * T * npos;
    uint8_t* o = J.lm + (size_t)phase * npos + idx;
    const size_t splane1 = (size_t)NS * Hd * 16;             // one (label, phase) plane in strip form
    uint8_t* so = J.strips ? J.strips + (size_t)phase * splane1 + ((size_t)(rx >> 4) * Hd + ry) * 16 + (rx & 15) : nullptr;
#pragma unroll
    for (int ori = 0; ori < 8; ++ori) {
        uint8_t r = ((v >> ori) & 1u) ? 4 : (((adj >> ori) & 1u) ? 1 : 0);
        o[plane * ori] = r;
        if (so) so[splane1 * T * T * ori] = r;
    }
}

// The same with dword stores (Wd % 4 == 0: the four positions of a lane quad share a row and a strip): every lane still ORs its own
// T x T pixels, then the quad exchanges the four OR bytes and lane k of the quad stores labels 2k and 2k + 1 for all four positions —
// one dword each to the flat plane and to the strip plane, 4 stores per lane instead of 16 byte stores (the launch is bound by the
// number of store instructions: 44 MB per 4-frame batch took 48 us = 0.9 TB/s).
static __device__ __forceinline__ void build_lm_body4(const int bx, const int by, const LmJob& J, int W, int H, int T, int Wd, int Hd, int NS, uint32_t m_wd,
                                                      uint32_t m_t) {
    const int idx = bx * 256 + (int)threadIdx.x;         // decimated raster index
    const int phase = by;
    const int npos = Wd * Hd;
    if (idx >= npos) return;                             // whole quads: npos is a multiple of 4
    const int ry = (int)fast_div((uint32_t)idx, m_wd, (uint32_t)Wd), rx = idx - ry * Wd;
    const int rs = (int)fast_div((uint32_t)phase, m_t, (uint32_t)T), cs = phase - rs * T;
    const int y = ry * T + rs, x = rx * T + cs;
    const uint32_t v = spread_or(J, x, y, W, H, T);
    const int lane = (int)threadIdx.x & 63, k4 = lane & 3, q0 = lane & ~3;
    uint32_t vq[4], adj[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        vq[k] = (uint32_t)__shfl((int)v, q0 + k, 64);
        adj[k] = ((vq[k] << 1) | (vq[k] >> 7) | (vq[k] >> 1) | (vq[k] << 7)) & 0xFFu;
    }
    const int idx0 = idx - k4, rx0 = rx - k4;            // first position of the quad
    const size_t plane = (size_t)T * T * npos;
    const size_t splane1 = (size_t)NS * Hd * 16;
    uint8_t* o = J.lm + (size_t)phase * npos + idx0;
    uint8_t* so = J.strips ? J.strips + (size_t)phase * splane1 + ((size_t)(rx0 >> 4) * Hd + ry) * 16 + (rx0 & 15) : nullptr;
#pragma unroll
    for (int li = 0; li < 2; ++li) {
        const int ori = 2 * k4 + li;
        uint32_t packed = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const uint32_t r = ((vq[k] >> ori) & 1u) ? 4u : (((adj[k] >> ori) & 1u) ? 1u : 0u);
            packed |= r << (8 * k);
        }
        *reinterpret_cast<uint32_t*>(o + plane * ori) = packed;
        if (so) *reinterpret_cast<uint32_t*>(so + splane1 * T * T * ori) = packed;
    }
}

__global__ void __launch_bounds__(256) k_build_lm(LmJob j0, LmJob j1, int W, int H, int T, int Wd, int Hd, int NS, uint32_t m_wd, uint32_t m_t) {
    if ((Wd & 3) == 0) build_lm_body4(blockIdx.x, blockIdx.y, blockIdx.z ? j1 : j0, W, H, T, Wd, Hd, NS, m_wd, m_t);
    else build_lm_body(blockIdx.x, blockIdx.y, blockIdx.z ? j1 : j0, W, H, T, Wd, Hd, NS, m_wd, m_t);
}

void launch_build_lm(const uint8_t* const quant[2], const uint8_t* const mask[2], uint8_t* const lm[2], uint8_t* const strips[2],
                     int W, int H, int T, hipStream_t s) {
    int Wd = W / T, Hd = H / T, NS = (Wd + 15) / 16;
    LmJob j0{quant[0], mask[0], lm[0], strips[0]}, j1{quant[1], mask[1], lm[1], strips[1]};
    hipLaunchKernelGGL(k_build_lm, dim3((Wd * Hd + 255) / 256, T * T, 2), dim3(256), 0, s, j0, j1, W, H, T, Wd, Hd, NS, div_magic((uint32_t)Wd), div_magic((uint32_t)T));
}

// ---- the bit planes written directly -------------------------------------------------------------------------------
// When nothing reads the byte planes of a level (bit-plane kernels for both passes, every window inside its plane) the linear-memory
// stage does not write them at all: 8 response bytes per position and encoding (flat + strip-major: 9.8 MB per VGA frame at level 0)
// become 2 bits per position and label (2.5 MB), and k_pack_bits / k_pack_top disappear from the batch.
//
// Strip records of a level below the top (match.hip: per plane row and 16-column strip 64 bits, cell c of [16 s, 16 s + 32) at bits 2c =
// "response is 1" and 2c + 1 = "response is 4").  A workgroup takes R rows of one (modality, phase): every thread ORs the T x T
// pixels of a few cells (spread) and leaves {neighbour bits & ~own bits | own bits << 8} per cell in LDS — response 4 iff the label's own
// bit is set, 1 iff only a neighbouring label's is (LL.cpp:1121) —; then one thread per (label, row, strip) gathers the label's two bits of
// its 32 cells into a record.  Rows fastest in the thread order: the records of a strip's R rows are R x 8 contiguous bytes.
constexpr int kBitsCells = 1152;              // 16-bit cell words of a workgroup's rows in LDS: R x (16 NS + 16) <= this (VGA level 0: 6 x 176)
static __host__ __device__ inline int fe_bits_rows(int Wd) { const int wp = ((Wd + 15) / 16) * 16 + 16; const int r = kBitsCells / wp; return r < 8 ? r : 8; }

static __device__ __forceinline__ void bits_rows_body(const int bx, const int by, const LmJob& J, int W, int H, int T, int Wd, int Hd, int NS, uint32_t m_wd,
                                                      uint32_t m_t, uint16_t* __restrict__ s_cells /* kBitsCells */) {
    const int Wp = NS * 16 + 16, R = fe_bits_rows(Wd);
    const int phase = by, ry0 = bx * R;
    const int rows = Hd - ry0 < R ? Hd - ry0 : R;
    const int rs = (int)fast_div((uint32_t)phase, m_t, (uint32_t)T), cs = phase - rs * T;
    // every thread's cells (<= kBitsCells / 256, rounded up) at once: their pixel loads are all in flight together
    constexpr int kPer = (kBitsCells + 255) / 256;
    const float inv_wp = 1.0f / (float)Wp;
    uint32_t w[kPer];
#pragma unroll
    for (int k = 0; k < kPer; ++k) {
        const int i = (int)threadIdx.x + 256 * k;
        const int r = (int)(((float)i + 0.5f) * inv_wp), rx = i - r * Wp;      // exact: i < 2048, Wp >= 32
        w[k] = 0;
        if (i < rows * Wp && rx < Wd) {
            const uint32_t v = spread_or(J, rx * T + cs, (ry0 + r) * T + rs, W, H, T);
            const uint32_t adj = ((v << 1) | (v >> 7) | (v >> 1) | (v << 7)) & 0xFFu;
            w[k] = (adj & ~v) | (v << 8);
        }
    }
#pragma unroll
    for (int k = 0; k < kPer; ++k) {
        const int i = (int)threadIdx.x + 256 * k;
        if (i < rows * Wp) s_cells[i] = (uint16_t)w[k];
    }
    __syncthreads();
    const size_t splane1 = (size_t)NS * Hd * 8;                           // one (label, phase) plane of records
    uint8_t* out = J.lm + (size_t)phase * splane1;
    const float inv_rows = 1.0f / (float)rows, inv_ns = 1.0f / (float)NS;
    for (int i = (int)threadIdx.x; i < 8 * rows * NS; i += 256) {
        const int q = (int)(((float)i + 0.5f) * inv_rows), r = i - q * rows;   // (small integers: the float quotients are exact)
        const int label = (int)(((float)q + 0.5f) * inv_ns), st = q - label * NS;
        const uint32_t* cw = reinterpret_cast<const uint32_t*>(s_cells + r * Wp + 16 * st);     // 32 cells = 16 dwords of two cells each
        uint32_t lo = 0, hi = 0;
#pragma unroll 2
        for (int k = 0; k < 8; ++k) {
            // bits label and 8 + label of two cells -> {is 1, is 4, is 1, is 4}
            lo |= (((((cw[k] >> label) & 0x01010101u) * 0x01020408u) >> 24) & 0xFu) << (4 * k);
            hi |= (((((cw[8 + k] >> label) & 0x01010101u) * 0x01020408u) >> 24) & 0xFu) << (4 * k);
        }
        *reinterpret_cast<uint2*>(out + (size_t)label * T * T * splane1 + ((size_t)st * Hd + ry0 + r) * 8) = make_uint2(lo, hi);
    }
}

// The pair stream of the top level (match.hip: {is-1 dword, is-4 dword} per 32 consecutive bytes of the flat arena).  One thread per
// (phase, position) as in the byte stage; a wave's 64 consecutive positions of a label's plane are 64 consecutive bits of the stream at an
// arbitrary bit offset (planes are not multiples of 32 positions): the ballots are shifted into place and OR-ed into the three dwords
// they touch.  The stream was zeroed by a job of the batch's first launch (fe_job_zero).
static __device__ __forceinline__ void top_bits_body(const int bx, const int by, const LmJob& J, int W, int H, int T, int Wd, int Hd, uint32_t m_wd, uint32_t m_t) {
    const int idx = bx * 256 + (int)threadIdx.x;
    const int phase = by, npos = Wd * Hd;
    const bool in = idx < npos;
    uint32_t v = 0;
    if (in) {
        const int ry = (int)fast_div((uint32_t)idx, m_wd, (uint32_t)Wd), rx = idx - ry * Wd;
        const int rs = (int)fast_div((uint32_t)phase, m_t, (uint32_t)T), cs = phase - rs * T;
        v = spread_or(J, rx * T + cs, ry * T + rs, W, H, T);
    }
    const uint32_t one = (((v << 1) | (v >> 7) | (v >> 1) | (v << 7)) & 0xFFu) & ~v;
    const int lane = (int)threadIdx.x & 63;
    const uint32_t idx0 = (uint32_t)(idx - lane);                         // the wave's first position (wave-uniform)
    uint32_t* stream = reinterpret_cast<uint32_t*>(J.lm);                 // pair p: stream[2 p] = is 1, stream[2 p + 1] = is 4
    const uint32_t top_bit0 = (uint32_t)reinterpret_cast<uintptr_t>(J.strips);
    const int plane = lane >= 3, k = lane - 3 * plane;                    // lanes 0..5 carry the three dwords of the two planes
#pragma unroll
    for (int ori = 0; ori < 8; ++ori) {
        const unsigned long long m1 = __ballot(in && ((one >> ori) & 1u)), m4 = __ballot(in && ((v >> ori) & 1u));
        if ((m1 | m4) == 0ull) continue;                                  // wave-uniform
        const uint32_t fo = top_bit0 + (uint32_t)(ori * T * T + phase) * (uint32_t)npos + idx0;   // flat position of the wave's first bit, from the stream's start
        const uint32_t sh = fo & 31u, p0 = fo >> 5;
        const unsigned long long m = plane ? m4 : m1;
        const uint32_t lo = (uint32_t)m, hi = (uint32_t)(m >> 32);
        uint32_t val;
        if (k == 0) val = lo << sh;
        else if (k == 1) val = sh ? (lo >> (32u - sh)) | (hi << sh) : hi;
        else val = sh ? hi >> (32u - sh) : 0u;
        if (lane < 6 && val) atomicOr(stream + 2 * (size_t)(p0 + (uint32_t)k) + (uint32_t)plane, val);
    }
}

// The same when the label planes start on 64-position boundaries of the stream (T * T * positions a multiple of 64: T = 8, or T = 4 with an
// even number of... any grid whose T^2 W_d H_d is): one thread per bit of a label's T * T * positions planes taken as ONE run — phase and
// position follow from the bit index —, a wave = 64 consecutive bits of that run in EVERY label's block, so its ballots are whole dwords:
// lane `label` stores the two pairs {is 1, is 4} x 2 with one 16-byte store.  No atomics, nothing to zero beforehand.
static __device__ __forceinline__ void top_bits_aligned_body(const int bx, const LmJob& J, int W, int H, int T, int Wd, int Hd, uint32_t m_wd, uint32_t m_t, uint32_t m_np) {
    const int npos = Wd * Hd, run = T * T * npos;
    const int b = bx * 256 + (int)threadIdx.x;
    const bool in = b < run;
    uint32_t v = 0;
    if (in) {
        const int phase = (int)fast_div((uint32_t)b, m_np, (uint32_t)npos), idx = b - phase * npos;
        const int ry = (int)fast_div((uint32_t)idx, m_wd, (uint32_t)Wd), rx = idx - ry * Wd;
        const int rs = (int)fast_div((uint32_t)phase, m_t, (uint32_t)T), cs = phase - rs * T;
        v = spread_or(J, rx * T + cs, ry * T + rs, W, H, T);
    }
    const uint32_t one = (((v << 1) | (v >> 7) | (v >> 1) | (v << 7)) & 0xFFu) & ~v;
    const int lane = (int)threadIdx.x & 63;
    const uint32_t b0 = (uint32_t)(b - lane);                             // the wave's first bit of the run (a multiple of 64)
    uint4 mine = make_uint4(0u, 0u, 0u, 0u);
#pragma unroll
    for (int ori = 0; ori < 8; ++ori) {
        const unsigned long long m1 = __ballot(in && ((one >> ori) & 1u)), m4 = __ballot(in && ((v >> ori) & 1u));
        if (lane == ori) mine = make_uint4((uint32_t)m1, (uint32_t)m4, (uint32_t)(m1 >> 32), (uint32_t)(m4 >> 32));
    }
    if (lane < 8 && b0 < (uint32_t)run) {
        const uint32_t pair = ((uint32_t)reinterpret_cast<uintptr_t>(J.strips) + (uint32_t)lane * (uint32_t)run + b0) >> 5;                  // even: the label's block and b0 are multiples of 64 positions
        *reinterpret_cast<uint4*>(J.lm + (size_t)pair * 8) = mine;
    }
}

// The pair stream from pixel TILES (T = 4 or 8, W_d a multiple of 8, the stream's first bit on a byte boundary): a workgroup takes one
// row of cells of one modality.  The 2T - 1 pixel rows its windows touch go to LDS as dwords of four pixels (masked, zero beyond the
// image); the T x T OR of EVERY pixel position of the cell row is separable — along x on the dwords (v_alignbyte), along y on the result —
// so the T x T positions of a cell (= the T x T phases of the linear memory, LL.cpp:1026-1243) share their loads and their ORs instead of
// every (phase, cell) thread loading its own T x T pixels (16 unaligned dword loads for T = 8).  A unit = (phase, 8 neighbouring cells):
// the 8 spread bytes (8 labels x 8 cells) are transposed as an 8 x 8 bit matrix, once for "own bit set" (response 4) and once for
// "only a neighbouring label's" (response 1), which gives for each label the byte of its 8 cells: 16 byte stores, no ballots, no atomics,
// nothing to clear.  (top_bits_aligned_body: 29.5 us per 8-frame batch at VGA — three times the cost per pixel of the level below.)
constexpr int kTileWords = 6144;                 // LDS pool of k_fe_bits (24 KB; 6 workgroups per CU): the staged pixel rows / spread rows / records of the tile writers
static_assert(kTileWords * 2 >= kBitsCells, "the pool holds bits_rows_body's cell words");
static __host__ __device__ inline bool fe_top_tile_fits(int W, int T) { return (T == 4 || T == 8) && (W / T) % 8 == 0 && (3 * T - 1) * (W / 4 + T / 4) <= kTileWords; }
static __host__ __device__ inline int fe_rows_block_rows(int NS, int T);
static __host__ __device__ inline bool fe_rows_tile_fits(int W, int T) { return (T == 4 || T == 5 || T == 8) && W % 16 == 0 && fe_rows_block_rows(((W / T) + 15) / 16, T) > 0; }   // (T = 5: the reference's default step, Detector() = {5, 8})

static __device__ __forceinline__ unsigned long long transpose8x8(unsigned long long x) {      // byte i bit j <-> byte j bit i
    unsigned long long t;
    t = (x ^ (x >> 7)) & 0x00AA00AA00AA00AAull; x = x ^ t ^ (t << 7);
    t = (x ^ (x >> 14)) & 0x0000CCCC0000CCCCull; x = x ^ t ^ (t << 14);
    t = (x ^ (x >> 28)) & 0x00000000F0F0F0F0ull; x = x ^ t ^ (t << 28);
    return x;
}

// stages of the top level's tile writer: the 2T - 1 pixel rows of cell row ry as dwords (masked, zero beyond the image, RW dwords per row),
// OR-ed along x in place, then along y into s_sp (row rs = the T x T OR of every window that starts in pixel row ry T + rs)
template <int kT>
static __device__ __forceinline__ void tile_spread(const int ry, const LmJob& J, int W, int H, int RW, uint32_t* __restrict__ s_px, uint32_t* __restrict__ s_sp) {
    constexpr int kRows = 2 * kT - 1;
    const int tid = (int)threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t* q4 = reinterpret_cast<const uint32_t*>(J.quant);
    const uint32_t* m4 = reinterpret_cast<const uint32_t*>(J.mask);
    for (int r = wave; r < kRows; r += 4) {
        const int y = ry * kT + r;
        for (int cw = lane; cw < RW; cw += 64) {
            uint32_t v = 0;
            if (y < H && 4 * cw < W) {
                v = q4[((size_t)y * W >> 2) + cw];
                if (m4) {
                    const uint32_t m = m4[((size_t)y * W >> 2) + cw];
                    const uint32_t nz = ((((m & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | m) & 0x80808080u) >> 7;      // 1 per non-zero mask byte
                    v &= nz * 0xFFu;
                }
            }
            s_px[r * RW + cw] = v;
        }
    }
    __syncthreads();
    // OR of kT consecutive pixels, a wave per row: every lane reads its dwords before any lane of the wave overwrites them
    for (int r = wave; r < kRows; r += 4) {
        for (int c0 = 0; c0 < RW; c0 += 64) {
            const int cw = c0 + lane;
            uint32_t h = 0;
            if (cw < RW) {
                const uint32_t d0 = s_px[r * RW + cw], d1 = cw + 1 < RW ? s_px[r * RW + cw + 1] : 0u;
                h = d0 | __builtin_amdgcn_alignbyte(d1, d0, 1) | __builtin_amdgcn_alignbyte(d1, d0, 2) | __builtin_amdgcn_alignbyte(d1, d0, 3);
                if (kT == 8) {
                    const uint32_t d2 = cw + 2 < RW ? s_px[r * RW + cw + 2] : 0u;
                    h |= d1 | __builtin_amdgcn_alignbyte(d2, d1, 1) | __builtin_amdgcn_alignbyte(d2, d1, 2) | __builtin_amdgcn_alignbyte(d2, d1, 3);
                }
            }
            __builtin_amdgcn_wave_barrier();
            if (cw < RW) s_px[r * RW + cw] = h;
            __builtin_amdgcn_wave_barrier();
        }
    }
    __syncthreads();
    for (int i = tid; i < kT * RW; i += 256) {            // ... and of kT consecutive rows
        const int rs = i / RW, cw = i - rs * RW;
        uint32_t v = 0;
#pragma unroll
        for (int k = 0; k < kT; ++k) v |= s_px[(rs + k) * RW + cw];
        s_sp[i] = v;
    }
    __syncthreads();
}
// response 1 of a label = a neighbouring label's bit without its own (LL.cpp:1121), on four packed spread bytes
static __device__ __forceinline__ uint32_t only_neighbours(uint32_t v) {
    return (((v << 1) & 0xFEFEFEFEu) | ((v >> 7) & 0x01010101u) | ((v >> 1) & 0x7F7F7F7Fu) | ((v << 7) & 0x80808080u)) & ~v;
}

template <int kT>
static __device__ __forceinline__ void top_bits_tile_body(const int ry, const LmJob& J, int W, int H, int Wd, int Hd, uint32_t* __restrict__ s_tile) {
    const int RW = W / 4 + kT / 4, tid = (int)threadIdx.x;
    uint32_t* const s_sp = s_tile + (2 * kT - 1) * RW;
    tile_spread<kT>(ry, J, W, H, RW, s_tile, s_sp);
    const int G = Wd >> 3, npos = Wd * Hd, units = kT * kT * G;
    const uint32_t run = (uint32_t)(kT * kT) * (uint32_t)npos, top_bit0 = (uint32_t)reinterpret_cast<uintptr_t>(J.strips);
    const uint8_t* sp = reinterpret_cast<const uint8_t*>(s_sp);
    for (int u = tid; u < units; u += 256) {
        const int phase = u / G, g = u - phase * G;
        const int rs = phase / kT, cs = phase - rs * kT;
        unsigned long long v = 0;
#pragma unroll
        for (int j = 0; j < 8; ++j) v |= (unsigned long long)sp[rs * (RW * 4) + (8 * g + j) * kT + cs] << (8 * j);
        const unsigned long long adj = ((v << 1) & 0xFEFEFEFEFEFEFEFEull) | ((v >> 7) & 0x0101010101010101ull) |
                                       ((v >> 1) & 0x7F7F7F7F7F7F7F7Full) | ((v << 7) & 0x8080808080808080ull);
        const unsigned long long t4 = transpose8x8(v), t1 = transpose8x8(adj & ~v);
        const uint32_t b = top_bit0 + (uint32_t)phase * (uint32_t)npos + (uint32_t)(ry * Wd + 8 * g);    // bit position (label 0) from the stream's start: a multiple of 8
#pragma unroll
        for (int l = 0; l < 8; ++l) {
            const uint32_t bl = b + (uint32_t)l * run;
            uint8_t* o = J.lm + (size_t)(bl >> 5) * 8 + ((bl >> 3) & 3u);
            o[0] = (uint8_t)(t1 >> (8 * l));                   // the pair's "is 1" dword ...
            o[4] = (uint8_t)(t4 >> (8 * l));                   // ... and its "is 4" dword
        }
    }
}

// The strip records of a level below the top from pixel tiles (T = 4, 5 or 8 — 5 is the reference's default step, LL.cpp:1663-1692 —, rows of a multiple of 16 pixels).  A workgroup takes a block of
// R = 16 (or 8) rows of cells of one modality and one pixel-row phase rs; records are [label][phase][strip][row], so R rows of a
// (label, phase, strip) are R x 8 contiguous bytes.
//   A. a wave per row of cells, a lane per four dwords of the row: over the T pixel rows of the windows, the OR of T consecutive pixels starting
//      at each pixel (one 16-byte load + the one or two dwords behind it, masked, zero beyond the image) -> s_sp[row][dword], byte b = the
//      T x T OR of the window at pixel 4 dword + b.  Nothing is staged: every pixel row is read once per phase rs, from L1 / L2.
//   B. per column phase cs: a unit = (16 neighbouring cells, row): per cell the byte "only a neighbouring label's bit" and the byte "own bit" —
//      rows 2c and 2c + 1 of a 32 x 8 bit matrix whose transpose is, per label, exactly the dword {is 1, is 4} x 16 cells of half a record
//      (four 8 x 8 transposes; the interleave is free).  The halves go to LDS as records (half h = low dword of record h, high dword of
//      record h - 1) and leave as 8-byte stores, rows fastest: a wave writes four whole 128-byte lines.
// (bits_rows_body: every (phase, cell) thread loads its own T x T pixels and every record gathers 32 cells with multiplies: 37.7 us per
// 8-frame batch at VGA.  The same tile arithmetic with each unit storing its 16 dwords straight to HBM was as slow: 5.4 M scattered dword stores.)
static __host__ __device__ inline int fe_rows_block_rows(int NS, int T) {       // rows of cells per workgroup: the spread rows + the records of one cs fit the pool
    for (int R = 16; R >= 8; R >>= 1)
        if (R * 4 * T * (NS + 1) + 8 * NS * R * 2 <= kTileWords) return R;
    return 0;
}
template <int kT>
static __device__ __forceinline__ void bits_rows_block_body(const int blk, const int part, const int csn, const LmJob& J, int W, int H, int Wd, int Hd, int NS, uint32_t* __restrict__ s_tile) {
    const int rs = part / (kT / csn), cs0 = (part - rs * (kT / csn)) * csn;        // a workgroup takes csn of the kT column phases of its row phase
    const int RW = (NS + 1) * 4 * kT, R = fe_rows_block_rows(NS, kT);       // dwords per spread row: the pixels of 16 NS + 16 cells
    const int tid = (int)threadIdx.x, lane = tid & 63, wave = tid >> 6, ry0 = blk * R;
    uint32_t* const s_sp = s_tile;                                         // R x RW
    uint2* const s_rec = reinterpret_cast<uint2*>(s_tile + R * RW);        // [label][strip][row]
    const uint32_t* q4 = reinterpret_cast<const uint32_t*>(J.quant);
    const uint32_t* m4 = reinterpret_cast<const uint32_t*>(J.mask);
    const int W4 = W >> 2;
    auto keep = [](uint32_t m) -> uint32_t { return (((((m & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | m) & 0x80808080u) >> 7) * 0xFFu; };   // 0xFF per non-zero mask byte
    auto px = [&](int y, int cw) -> uint32_t {                             // four pixels of row y, masked; zero beyond the image
        uint32_t v = 0;
        if (cw < W4) {
            v = q4[(size_t)y * W4 + cw];
            if (m4) v &= keep(m4[(size_t)y * W4 + cw]);
        }
        return v;
    };
    for (int rr = wave; rr < R; rr += 4) {
        const int ry = ry0 + rr;
        if (ry >= Hd) break;                                               // (wave-uniform)
        for (int qd = lane; 4 * qd < RW; qd += 64) {
            uint32_t acc[4] = {0u, 0u, 0u, 0u};
#pragma unroll 4
            for (int k = 0; k < kT; ++k) {
                const int y = ry * kT + rs + k;
                if (y >= H) continue;
                uint32_t d[6] = {0u, 0u, 0u, 0u, 0u, 0u};
                if (4 * qd < W4) {                                         // (rows are multiples of 16 pixels: a quad is inside or outside)
                    const uint4 v = *reinterpret_cast<const uint4*>(q4 + (size_t)y * W4 + 4 * qd);
                    d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
                    if (m4) {
                        const uint4 m = *reinterpret_cast<const uint4*>(m4 + (size_t)y * W4 + 4 * qd);
                        d[0] &= keep(m.x); d[1] &= keep(m.y); d[2] &= keep(m.z); d[3] &= keep(m.w);
                    }
                }
                d[4] = px(y, 4 * qd + 4);
                if (kT > 5) d[5] = px(y, 4 * qd + 5);
#pragma unroll
                for (int q = 0; q < 4; ++q) {                              // byte b of acc[q]: the OR of the kT pixels from pixel 4 q + b on (kT = 4 .. 8)
                    acc[q] |= d[q] | __builtin_amdgcn_alignbyte(d[q + 1], d[q], 1) | __builtin_amdgcn_alignbyte(d[q + 1], d[q], 2) | __builtin_amdgcn_alignbyte(d[q + 1], d[q], 3);
                    if (kT > 4) acc[q] |= d[q + 1];
                    if (kT > 5) acc[q] |= __builtin_amdgcn_alignbyte(d[q + 2], d[q + 1], 1);
                    if (kT > 6) acc[q] |= __builtin_amdgcn_alignbyte(d[q + 2], d[q + 1], 2);
                    if (kT > 7) acc[q] |= __builtin_amdgcn_alignbyte(d[q + 2], d[q + 1], 3);
                }
            }
            *reinterpret_cast<uint4*>(s_sp + rr * RW + 4 * qd) = make_uint4(acc[0], acc[1], acc[2], acc[3]);
        }
    }
    __syncthreads();
    const uint8_t* sp = reinterpret_cast<const uint8_t*>(s_sp);
    const size_t splane1 = (size_t)NS * Hd * 8;                            // one (label, phase) plane of records
    const int rows = Hd - ry0 < R ? Hd - ry0 : R;
    for (int cs = cs0; cs < cs0 + csn; ++cs) {
        for (int u = tid; u < (NS + 1) * R; u += 256) {
            const int h = u / R, rr = u - h * R;                           // rows fastest
            if (rr >= rows) continue;
            const uint8_t* cell = sp + rr * (RW * 4) + (16 * h) * kT + cs; // the spread byte of cell 16 h + c sits kT bytes further per cell
            unsigned long long t[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const uint32_t v = (uint32_t)cell[(4 * k) * kT] | ((uint32_t)cell[(4 * k + 1) * kT] << 8) | ((uint32_t)cell[(4 * k + 2) * kT] << 16) | ((uint32_t)cell[(4 * k + 3) * kT] << 24);
                const uint32_t o = only_neighbours(v);
                // rows of the bit matrix: o0 v0 o1 v1 | o2 v2 o3 v3
                const uint32_t lo = __builtin_amdgcn_perm(v, o, 0x05010400u), hi = __builtin_amdgcn_perm(v, o, 0x07030602u);
                t[k] = transpose8x8(((unsigned long long)hi << 32) | lo);  // byte l = label l: bits 2c, 2c + 1 of cells 4 k .. 4 k + 3
            }
#pragma unroll
            for (int half = 0; half < 2; ++half) {                         // label l: bytes l of t[0..3]
                const uint32_t a0 = (uint32_t)(t[0] >> (32 * half)), a1 = (uint32_t)(t[1] >> (32 * half)), a2 = (uint32_t)(t[2] >> (32 * half)), a3 = (uint32_t)(t[3] >> (32 * half));
                const uint32_t p0 = __builtin_amdgcn_perm(a1, a0, 0x05010400u), p1 = __builtin_amdgcn_perm(a1, a0, 0x07030602u);
                const uint32_t q0 = __builtin_amdgcn_perm(a3, a2, 0x05010400u), q1 = __builtin_amdgcn_perm(a3, a2, 0x07030602u);
                const uint32_t w[4] = {__builtin_amdgcn_perm(q0, p0, 0x05040100u), __builtin_amdgcn_perm(q0, p0, 0x07060302u),
                                       __builtin_amdgcn_perm(q1, p1, 0x05040100u), __builtin_amdgcn_perm(q1, p1, 0x07060302u)};
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int l = 4 * half + q;
                    if (h < NS) s_rec[(l * NS + h) * R + rr].x = w[q];
                    if (h > 0) s_rec[(l * NS + h - 1) * R + rr].y = w[q];
                }
            }
        }
        __syncthreads();
        const int phase = rs * kT + cs;
        for (int i = tid; i < 8 * NS * R; i += 256) {
            const int ls = i / R, rr = i - ls * R;                         // ls = label * NS + strip
            if (rr >= rows) continue;
            const int l = ls / NS, st = ls - l * NS;
            *reinterpret_cast<uint2*>(J.lm + ((size_t)l * kT * kT + phase) * splane1 + ((size_t)st * Hd + ry0 + rr) * 8) = s_rec[i];
        }
        __syncthreads();
    }
}

// ---- several independent front-end jobs in ONE launch ---------------------------------------------------------------------
// The seven kernels of a frame are small (5-17 us) and dependent kernels on a queue start ~7 us apart, so the front end is
// mostly launch latency.  The jobs that do not depend on each other share a launch: stage k = {colour chain of level k,
// normals + median (k = 0) or nearest-neighbour normals of level k, pyrDown to level k + 1}; the last launch builds the linear
// memories of every level.  A job is a range of the flat block index; the bodies are the kernels above, unchanged.
// The workgroups are persistent: a tile takes a workgroup ~1.5 us, and the dispatcher hands an XCD a new workgroup only every
// ~30 ns — with one workgroup per tile (10.8k for the first stage of four VGA frames) a CU held 1.3 workgroups on average and the
// stage took as long as the dispatcher needed (42 us; profiles/r03_pmc.txt: 5 waves per CU).  A few workgroups per CU walk the
// tiles instead (flat index + k * grid).
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(kFeWaves, kFeWaves)))   // kFeWaves persistent workgroups per CU (knobs: fe_wgs_per_cu) = as many waves per SIMD
k_fe_stage(FeStage st, int total) {
    for (int blk = (int)blockIdx.x; blk < total; blk += (int)gridDim.x) {
        int j = 0;
        while (j + 1 < st.njobs && blk >= st.job[j + 1].first) ++j;
        const FeJob& J = st.job[j];
        const int local = blk - J.first;
        const int bz = (int)fast_div((uint32_t)local, J.m_gxgy, (uint32_t)(J.gx * J.gy)), rem = local - bz * J.gx * J.gy;
        const int by = (int)fast_div((uint32_t)rem, J.m_gx, (uint32_t)J.gx), bx = rem - by * J.gx;
        switch (J.kind) {
            case kFeColour: color_quant_body(bx, by, (const uint8_t*)J.in, (float*)J.out0, (uint8_t*)J.out1, J.W, J.H, J.f); break;
            case kFeNormals: normals_median_body(bx, by, (const uint16_t*)J.in, (uint8_t*)J.out0, (uint8_t*)J.out1, J.W, J.H, J.a, J.b); break;
            case kFePyrDown: pyrdown_body(bx, by, (const uint8_t*)J.in, (uint8_t*)J.out0, J.W, J.H, J.a, J.b); break;
            case kFeNnDown: nn_down2_body(bx, by, (const uint8_t*)J.in, (uint8_t*)J.out0, J.W, J.a); break;
            case kFeBuildLm:
                if ((J.Wd & 3) == 0) build_lm_body4(bx, by, J.lm[bz], J.W, J.H, J.a, J.Wd, J.Hd, (J.Wd + 15) >> 4, J.m_wd, J.m_t);
                else build_lm_body(bx, by, J.lm[bz], J.W, J.H, J.a, J.Wd, J.Hd, (J.Wd + 15) >> 4, J.m_wd, J.m_t);
                break;
            default: break;
        }
        if (blk + (int)gridDim.x < total) __syncthreads();      // the next tile reuses the bodies' LDS
    }
}

// The bit-plane jobs of a batch (strip records of every level below the top, pair stream of the top level: bits_rows_body, top_bits_body)
// in a launch of their own, the last of the front end: inside k_fe_stage they would cost the colour chain a wave of occupancy (95 VGPRs
// against 79).  Same job table, same persistent walk.
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(6, 6)))
k_fe_bits(FeStage st, int total) {
    __shared__ __attribute__((aligned(16))) uint32_t s_tile[kTileWords];                // one pool for the bodies (a block runs one of them): the cell words of bits_rows_body, the pixel tile of top_bits_tile_body
    for (int blk = (int)blockIdx.x; blk < total; blk += (int)gridDim.x) {
        int j = 0;
        while (j + 1 < st.njobs && blk >= st.job[j + 1].first) ++j;
        const FeJob& J = st.job[j];
        const int local = blk - J.first;
        const int bz = (int)fast_div((uint32_t)local, J.m_gxgy, (uint32_t)(J.gx * J.gy)), rem = local - bz * J.gx * J.gy;
        const int by = (int)fast_div((uint32_t)rem, J.m_gx, (uint32_t)J.gx), bx = rem - by * J.gx;
        if (J.kind == kFeBitsRows) bits_rows_body(bx, by, J.lm[bz], J.W, J.H, J.a, J.Wd, J.Hd, (J.Wd + 15) >> 4, J.m_wd, J.m_t, reinterpret_cast<uint16_t*>(s_tile));
        else if (J.kind == kFeTopBits) top_bits_body(bx, by, J.lm[bz], J.W, J.H, J.a, J.Wd, J.Hd, J.m_wd, J.m_t);
        else if (J.kind == kFeTopBitsAligned) top_bits_aligned_body(bx, J.lm[bz], J.W, J.H, J.a, J.Wd, J.Hd, J.m_wd, J.m_t, J.m_np);
        else if (J.kind == kFeBitsRowsTile) {
            if (J.a == 8) bits_rows_block_body<8>(bx, by, J.b, J.lm[bz], J.W, J.H, J.Wd, J.Hd, (J.Wd + 15) >> 4, s_tile);
            else if (J.a == 5) bits_rows_block_body<5>(bx, by, J.b, J.lm[bz], J.W, J.H, J.Wd, J.Hd, (J.Wd + 15) >> 4, s_tile);
            else bits_rows_block_body<4>(bx, by, J.b, J.lm[bz], J.W, J.H, J.Wd, J.Hd, (J.Wd + 15) >> 4, s_tile);
        }
        else if (J.kind == kFeTopBitsTile) { if (J.a == 8) top_bits_tile_body<8>(bx, J.lm[bz], J.W, J.H, J.Wd, J.Hd, s_tile); else top_bits_tile_body<4>(bx, J.lm[bz], J.W, J.H, J.Wd, J.Hd, s_tile); }
        if (blk + (int)gridDim.x < total) __syncthreads();      // the next block of rows reuses the cell words in LDS
    }
}

void fe_job_colour(FeJob& j, const uint8_t* rgb, float* mag, uint8_t* onehot, int W, int H, float thr_sq) {
    j = FeJob{}; j.kind = kFeColour; j.gx = (W + kCTX - 1) / kCTX; j.gy = (H + kCTY - 1) / kCTY; j.gz = 1;
    j.in = rgb; j.out0 = mag; j.out1 = onehot; j.W = W; j.H = H; j.f = thr_sq;
}
void fe_job_normals(FeJob& j, const uint16_t* depth, uint8_t* raw, uint8_t* med, int W, int H, int dist_thr, int diff_thr) {
    j = FeJob{}; j.kind = kFeNormals; j.gx = (W + kCTX - 1) / kCTX; j.gy = (H + kCTY - 1) / kCTY; j.gz = 1;
    j.in = depth; j.out0 = raw; j.out1 = med; j.W = W; j.H = H; j.a = dist_thr; j.b = diff_thr;
}
void fe_job_pyrdown(FeJob& j, const uint8_t* src, uint8_t* dst, int W, int H) {
    j = FeJob{}; j.kind = kFePyrDown; j.a = W / 2; j.b = H / 2; j.gx = (j.a * 3 + 255) / 256; j.gy = j.b; j.gz = 1;
    j.in = src; j.out0 = dst; j.W = W; j.H = H;
}
void fe_job_nn_down2(FeJob& j, const uint8_t* src, uint8_t* dst, int W, int H) {
    j = FeJob{}; j.kind = kFeNnDown; j.a = W / 2; j.gx = (j.a + 255) / 256; j.gy = H / 2; j.gz = 1;
    j.in = src; j.out0 = dst; j.W = W; j.H = H;
}
void fe_job_build_lm(FeJob& j, const uint8_t* const quant[2], const uint8_t* const mask[2], uint8_t* const lm[2], uint8_t* const strips[2],
                     int W, int H, int T) {
    j = FeJob{}; j.kind = kFeBuildLm; j.a = T; j.gx = ((W / T) * (H / T) + 255) / 256; j.gy = T * T; j.gz = 2; j.W = W; j.H = H;
    j.Wd = W / T; j.Hd = H / T; j.m_wd = div_magic((uint32_t)j.Wd); j.m_t = div_magic((uint32_t)T);
    j.lm[0] = LmJob{quant[0], mask[0], lm[0], strips[0]}; j.lm[1] = LmJob{quant[1], mask[1], lm[1], strips[1]};
}
// strip records of a level below the top, written directly (bits[m]: the level's record block of modality m inside the bit arena)
bool fe_bits_rows_possible(int W, int T) { return fe_bits_rows(W / T) >= 1; }
void fe_job_bits_rows(FeJob& j, const uint8_t* const quant[2], const uint8_t* const mask[2], uint8_t* const bits[2], int W, int H, int T, bool tiles) {
    j = FeJob{}; j.kind = kFeBitsRows; j.a = T; j.W = W; j.H = H;
    j.Wd = W / T; j.Hd = H / T; j.m_wd = div_magic((uint32_t)j.Wd); j.m_t = div_magic((uint32_t)T);
    const int R = fe_bits_rows(j.Wd);
    j.gx = (j.Hd + R - 1) / R; j.gy = T * T; j.gz = 2;
    if (tiles && fe_rows_tile_fits(W, T)) {                // a workgroup per block of rows of cells and pixel-row phase
        const int Rb = fe_rows_block_rows((j.Wd + 15) / 16, T);
        const int k = knobs().fe_rows_cs;
        j.b = k >= 1 && k <= T && T % k == 0 ? k : (T % 2 == 0 ? 2 : 1);   // column phases per workgroup, a divisor of T (LM_FE_ROWS_CS; VGA level 0 at T = 4, per 8-frame batch: all four 23.7 us, two 21.1, one 25.9)
        j.kind = kFeBitsRowsTile; j.gx = (j.Hd + Rb - 1) / Rb; j.gy = T * (T / j.b);
    }
    j.lm[0] = LmJob{quant[0], mask[0], bits[0], nullptr}; j.lm[1] = LmJob{quant[1], mask[1], bits[1], nullptr};
}
// pair stream of the top level, written directly; bit0[m] = flat arena offset of modality m's block less the stream's first byte
int fe_top_bits_kind(int W, int H, int T, const uint32_t bit0[2], int mode) {
    if (mode != 1 && mode != 2 && fe_top_tile_fits(W, T) && bit0[0] % 8 == 0 && bit0[1] % 8 == 0) return kFeTopBitsTile;
    if (mode != 1 && ((long)T * T * (W / T) * (H / T)) % 64 == 0) return kFeTopBitsAligned;
    return kFeTopBits;
}
void fe_job_top_bits(FeJob& j, const uint8_t* const quant[2], const uint8_t* const mask[2], uint8_t* stream, const uint32_t bit0[2], int W, int H, int T, int mode) {
    j = FeJob{}; j.kind = kFeTopBits; j.a = T; j.gx = ((W / T) * (H / T) + 255) / 256; j.gy = T * T; j.gz = 2; j.W = W; j.H = H;
    j.Wd = W / T; j.Hd = H / T; j.m_wd = div_magic((uint32_t)j.Wd); j.m_t = div_magic((uint32_t)T);
    j.lm[0] = LmJob{quant[0], mask[0], stream, reinterpret_cast<uint8_t*>((uintptr_t)bit0[0])}; j.lm[1] = LmJob{quant[1], mask[1], stream, reinterpret_cast<uint8_t*>((uintptr_t)bit0[1])};
    const int kind = fe_top_bits_kind(W, H, T, bit0, mode);
    if (kind == kFeTopBitsTile) {                           // a workgroup per row of cells: whole bytes, no atomics, nothing to clear
        j.kind = kFeTopBitsTile; j.gx = j.Hd; j.gy = 1;
    } else if (kind == kFeTopBitsAligned) {                 // whole dwords per wave: no atomics, nothing to clear
        const int npos = j.Wd * j.Hd;
        j.kind = kFeTopBitsAligned; j.gx = (T * T * npos + 255) / 256; j.gy = 1; j.m_np = div_magic((uint32_t)npos);
    }
}
static int fe_prepare(FeStage& st) {                             // drops empty jobs, lays the jobs' blocks out on one flat index; returns the number of blocks
    int total = 0, n = 0;
    for (int i = 0; i < st.njobs; ++i) {
        const int blocks = st.job[i].gx * st.job[i].gy * st.job[i].gz;
        if (blocks <= 0) continue;                         // an empty job (degenerate level) is dropped
        st.job[n] = st.job[i]; st.job[n].first = total; total += blocks;
        st.job[n].m_gx = div_magic((uint32_t)st.job[n].gx); st.job[n].m_gxgy = div_magic((uint32_t)(st.job[n].gx * st.job[n].gy));
        ++n;
    }
    st.njobs = n;
    return total;
}
static int fe_cus() {                                                // CUs of the CURRENT device (a process may drive several)
    static int cus[64] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 256;
    if (!cus[dev] && (hipDeviceGetAttribute(&cus[dev], hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus[dev] <= 0)) cus[dev] = 256;
    return cus[dev];
}
void launch_fe_stage(FeStage& st, hipStream_t s) {
    const int total = fe_prepare(st);
    if (total <= 0) return;
    const int per_cu = knobs().fe_wgs_per_cu;
    const int grid = per_cu > 0 ? std::min(total, fe_cus() * per_cu) : total;
    hipLaunchKernelGGL(k_fe_stage, dim3(grid), dim3(256), 0, s, st, total);
}
void launch_fe_bits(FeStage& st, hipStream_t s) {
    if (knobs().fe_bits_split) {                          // LM_FE_BITS_SPLIT=1 (measurements): the strip-record jobs and the pair-stream jobs as two launches
        for (int kind : {kFeBitsRows, kFeTopBits}) {
            FeStage part{};
            for (int i = 0; i < st.njobs; ++i) if ((st.job[i].kind == kFeBitsRows || st.job[i].kind == kFeBitsRowsTile) == (kind == kFeBitsRows)) part.job[part.njobs++] = st.job[i];
            const int total = fe_prepare(part);
            if (total > 0) hipLaunchKernelGGL(k_fe_bits, dim3(std::min(total, fe_cus() * 8)), dim3(256), 0, s, part, total);
        }
        return;
    }
    const int total = fe_prepare(st);
    if (total <= 0) return;
    hipLaunchKernelGGL(k_fe_bits, dim3(std::min(total, fe_cus() * 8)), dim3(256), 0, s, st, total);
}

}  // namespace lm
