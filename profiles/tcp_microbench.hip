// Micro-benchmark of the access pattern of k_local (DESIGN §8 "where the next factor would come from"): how many useful
// window bytes per second the vector L1 delivers for different layouts of a 16x16 response window.  Stand-alone:
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/tcp_microbench profiles/tcp_microbench.hip && /tmp/tcp_microbench
// Every "feature" is a window at a random position of a random plane of an L2-resident arena (5-10 MB); a wave sums the
// bytes of `feats` windows, 8 independent loads in flight per lane like the real kernel.  Patterns:
//   0  strips (current): 16-byte-wide strips stored [strip][row][16 B]; a window = 16 rows x 2 strips, 32 lanes x 16 B, half useful
//   1  double-width overlapping strips: [strip][row][32 B], strip k = columns 16k..16k+31; a window = 16 rows x one unaligned
//      16-byte load at byte offset c0 -> 16 lanes per feature, every byte useful
//   2  double-width strips, aligned 2 x 16 B per row (32 lanes per feature, half useful): isolates the effect of the stride
//   3  flat rows (no strips): 16 rows Wd bytes apart, unaligned 16-byte loads, 16 lanes per feature
//   4  nibble strips: responses are 0 / 1 / 4, two per byte -> [strip][row][8 B]; a window = 16 rows x 2 strips, 32 lanes x 8 B;
//      half the arena (fits one XCD's L2) and half the bytes per row; 'useful' is counted as 256 positions per window like the others
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

struct Feat { uint32_t base; uint32_t c0; };   // byte offset of the window's first row (strip S0 for pattern 0), column offset in the strip

template <int P>
__global__ void __launch_bounds__(256) k_bench(const uint8_t* __restrict__ arena, const Feat* __restrict__ feats, int feats_per_wave, uint32_t strip_stride,
                                               uint32_t row_stride, unsigned long long* __restrict__ out) {
    const int lane = threadIdx.x & 63;
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const Feat* f = feats + (size_t)wave * feats_per_wave;
    uint32_t acc0 = 0, acc1 = 0, acc2 = 0, acc3 = 0;
    if (P == 0 || P == 2) {                        // 32 lanes per feature: row r, half h
        const int half = lane >> 5, l5 = lane & 31, r = l5 >> 1, h = l5 & 1;
        for (int i = 0; i < feats_per_wave; i += 16) {
            uint4 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const Feat ft = f[i + 2 * u + half];
                const uint32_t off = P == 0 ? ft.base + h * strip_stride + r * 16 : ft.base + r * 32 + h * 16;
                v[u] = *reinterpret_cast<const uint4*>(arena + off);
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) { acc0 += v[u].x; acc1 += v[u].y; acc2 += v[u].z; acc3 += v[u].w; }
        }
    } else if (P == 4) {                           // 32 lanes per feature, 8-byte rows
        const int half = lane >> 5, l5 = lane & 31, r = l5 >> 1, h = l5 & 1;
        for (int i = 0; i < feats_per_wave; i += 16) {
            uint2 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const Feat ft = f[i + 2 * u + half];
                v[u] = *reinterpret_cast<const uint2*>(arena + ft.base + h * strip_stride + r * 8);
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {           // the unpacking the real kernel would need: low / high nibbles to bytes
                acc0 += v[u].x & 0x0F0F0F0Fu; acc1 += (v[u].x >> 4) & 0x0F0F0F0Fu; acc2 += v[u].y & 0x0F0F0F0Fu; acc3 += (v[u].y >> 4) & 0x0F0F0F0Fu;
            }
        }
    } else if (P == 1 || P == 3) {                 // 16 lanes per feature: row r, one unaligned 16-byte load
        const int q = lane >> 4, r = lane & 15;
        for (int i = 0; i < feats_per_wave; i += 32) {
            uint4 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const Feat ft = f[i + 4 * u + q];
                const uint32_t off = ft.base + r * row_stride + ft.c0;
                const uint8_t* p = arena + off;
                // one global_load_dwordx4 from a byte address (compute queues run with unaligned access enabled); written as
                // inline assembly because the compiler would split an align-1 access into byte loads
                asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(v[u]) : "v"(p) : "memory");
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
            for (int u = 0; u < 8; ++u) { acc0 += v[u].x; acc1 += v[u].y; acc2 += v[u].z; acc3 += v[u].w; }
        }
    }
    unsigned long long s = (unsigned long long)acc0 + acc1 + acc2 + acc3;
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor((int)s, o, 64);
    if (lane == 0) out[wave] = s;
}

int main() {
    const int Hd = 120, Wd = 160, T = 4;
    const int planes = 2 * 8 * T * T;                               // (label, phase) planes of both modalities at level 0
    const int NS = (Wd + 15) / 16;
    const size_t strip_plane = (size_t)NS * Hd * 16;               // pattern 0: one plane in strip form
    const size_t dstrip_plane = (size_t)NS * Hd * 32;              // patterns 1, 2, 4
    const size_t flat_plane = (size_t)Wd * Hd;
    const size_t arena_bytes = planes * dstrip_plane + 4096;
    const int waves = 768 * 4, feats_per_wave = 600 * 4;            // ~ candidates per wave x features per candidate of the bench frame
    uint8_t* d_arena; Feat* d_f; unsigned long long* d_out;
    CK(hipMalloc(&d_arena, arena_bytes)); CK(hipMemset(d_arena, 1, arena_bytes));
    CK(hipMalloc(&d_f, sizeof(Feat) * (size_t)waves * feats_per_wave)); CK(hipMalloc(&d_out, 8 * waves));
    std::vector<Feat> hf((size_t)waves * feats_per_wave);
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int P = 0; P < 5; ++P) {
        srand(1);
        for (auto& ft : hf) {
            const int plane = rand() % planes, gx = rand() % (Wd - 32), gy = rand() % (Hd - 16);
            const int S0 = gx >> 4, c0 = gx & 15;
            if (P == 0) ft.base = (uint32_t)(plane * strip_plane + ((size_t)S0 * Hd + gy) * 16), ft.c0 = c0;
            else if (P == 4) ft.base = (uint32_t)(plane * (strip_plane / 2) + ((size_t)S0 * Hd + gy) * 8), ft.c0 = c0;
            else if (P == 1 || P == 2) ft.base = (uint32_t)(plane * dstrip_plane + ((size_t)S0 * Hd + gy) * 32), ft.c0 = c0;
            else ft.base = (uint32_t)(plane * flat_plane + (size_t)gy * Wd + (gx & ~15)), ft.c0 = c0;
        }
        CK(hipMemcpy(d_f, hf.data(), sizeof(Feat) * hf.size(), hipMemcpyHostToDevice));
        const uint32_t strip_stride = P == 4 ? Hd * 8 : Hd * 16, row_stride = P == 3 ? Wd : 32;
        float best = 1e9f;
        for (int rep = 0; rep < 5; ++rep) {
            CK(hipEventRecord(e0));
            if (P == 0) hipLaunchKernelGGL(k_bench<0>, dim3(768), dim3(256), 0, 0, d_arena, d_f, feats_per_wave, strip_stride, row_stride, d_out);
            if (P == 1) hipLaunchKernelGGL(k_bench<1>, dim3(768), dim3(256), 0, 0, d_arena, d_f, feats_per_wave, strip_stride, row_stride, d_out);
            if (P == 2) hipLaunchKernelGGL(k_bench<2>, dim3(768), dim3(256), 0, 0, d_arena, d_f, feats_per_wave, strip_stride, row_stride, d_out);
            if (P == 4) hipLaunchKernelGGL(k_bench<4>, dim3(768), dim3(256), 0, 0, d_arena, d_f, feats_per_wave, strip_stride, row_stride, d_out);
            if (P == 3) hipLaunchKernelGGL(k_bench<3>, dim3(768), dim3(256), 0, 0, d_arena, d_f, feats_per_wave, strip_stride, row_stride, d_out);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            if (ms < best) best = ms;
        }
        CK(hipGetLastError());
        const double useful = (double)waves * feats_per_wave * 256.0;
        printf("pattern %d: %.3f ms for %.2f GB of window bytes -> %.1f GB/s useful (k_local today: ~7100)\n", P, best, useful / 1e9, useful / best / 1e6);
    }
    return 0;
}
