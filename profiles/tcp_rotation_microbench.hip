// Round 3 micro-benchmark: does aligning the lane quads of k_local's wave loads to 64-byte lines pay?  The strip layout stores 16-byte rows
// consecutively, a 64-byte line = 4 consecutive rows; a window starts at an arbitrary row, so 3 of 4 times every quad of lanes (4 rows)
// straddles two lines.  "Rotated" patterns give lanes 4g..4g+3 the g-th ALIGNED group of rows of the window and put the two partial
// groups (head and tail of the window) into the last quad: lines touched = accesses.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/tcp_rot profiles/tcp_rotation_microbench.hip && /tmp/tcp_rot
// Patterns (all: random plane + position in an L2-resident 4.9 MB arena, 8 independent loads in flight per lane):
//   0  tile as k_local reads it today: lane = (strip q of 3, row r of 20), 60 lanes, rows consecutive from the window's first row
//   1  tile rotated: lane (q, j) reads aligned slot ((j + 4 - rho) mod 20) + rho, rho = first row & 3
//   2  single as today: two features per load, lane = (half, row r of 16, strip h of 2) with h fastest
//   3  single, rows fastest: lane = (half, strip h, row r), unrotated
//   4  single, rows fastest, rotated
//   5  tile with only 2 of the 3 strips loaded (40 active lanes, the others masked off): is the cost per active lane?
//   6  tile, 3 strips, but 8-byte loads (half the bytes per lane)
//   7  tile, all 64 lanes re-reading ONE small region (L1 hits only): the issue rate of the load path itself
//   8-11  tile with only the first 40 / 32 / 20 / 16 lanes alive (the others leave the kernel at once): cost per active lane or per instruction?
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <int P>
__global__ void __launch_bounds__(256) k_bench(const uint8_t* __restrict__ arena, const uint32_t* __restrict__ feats, int feats_per_wave, uint32_t strip_stride,
                                               unsigned long long* __restrict__ out) {
    const int lane = threadIdx.x & 63;
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const uint32_t* f = feats + (size_t)wave * feats_per_wave;      // per feature: byte offset of (strip S0, row y0)
    uint32_t a0 = 0, a1 = 0, a2 = 0, a3 = 0;
    if (P >= 8) { const int alive = P == 8 ? 40 : P == 9 ? 32 : P == 10 ? 20 : 16; if (lane >= alive) return; }
    if (P <= 1 || P >= 5) {
        const int q = lane / 20 > 2 ? 0 : lane / 20, j = lane >= 60 ? 0 : lane % 20;
        for (int i = 0; i < feats_per_wave; i += 8) {
            uint4 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const uint32_t b = f[i + u];
                uint32_t off;
                if (P == 0 || P >= 5) off = (P == 7 ? (b & 0xFFF0u) : b) + q * strip_stride + j * 16;   // (P >= 8: the same addresses, fewer lanes)
                else { const uint32_t rho = (b >> 4) & 3; off = (b & ~63u) + q * strip_stride + ((((uint32_t)j + 4 - rho) % 20) + rho) * 16; }
                if (P == 5) { v[u] = make_uint4(0, 0, 0, 0); if (lane < 40) v[u] = *reinterpret_cast<const uint4*>(arena + off); }
                else if (P == 6) { const uint2 t = *reinterpret_cast<const uint2*>(arena + off); v[u] = make_uint4(t.x, t.y, 0, 0); }
                else v[u] = *reinterpret_cast<const uint4*>(arena + off);
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) { a0 += v[u].x; a1 += v[u].y; a2 += v[u].z; a3 += v[u].w; }
        }
    } else {
        const int half = lane >> 5, l5 = lane & 31;
        const int r = P == 2 ? l5 >> 1 : l5 & 15, h = P == 2 ? l5 & 1 : l5 >> 4;
        for (int i = 0; i < feats_per_wave; i += 16) {
            uint4 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const uint32_t b = f[i + 2 * u + half];
                uint32_t off;
                if (P != 4) off = b + h * strip_stride + r * 16;
                else { const uint32_t rho = (b >> 4) & 3; off = (b & ~63u) + h * strip_stride + ((((uint32_t)r + 4 - rho) & 15) + rho) * 16; }
                v[u] = *reinterpret_cast<const uint4*>(arena + off);
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) { a0 += v[u].x; a1 += v[u].y; a2 += v[u].z; a3 += v[u].w; }
        }
    }
    unsigned long long s = (unsigned long long)a0 + a1 + a2 + a3;
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor((int)s, o, 64);
    if (lane == 0) out[wave] = s;
}

int main() {
    const int Hd = 120, Wd = 160, T = 4, planes = 2 * 8 * T * T, NS = (Wd + 15) / 16;
    const size_t strip_plane = (size_t)NS * Hd * 16, arena_bytes = planes * strip_plane + 65536;
    const int waves = 768 * 4, feats_per_wave = 2400;
    uint8_t* d_arena; uint32_t* d_f; unsigned long long* d_out;
    CK(hipMalloc(&d_arena, arena_bytes)); CK(hipMemset(d_arena, 1, arena_bytes));
    CK(hipMalloc(&d_f, 4 * (size_t)waves * feats_per_wave)); CK(hipMalloc(&d_out, 8 * waves));
    std::vector<uint32_t> hf((size_t)waves * feats_per_wave);
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int P = 0; P < 12; ++P) {
        srand(1);
        const int rows = (P <= 1 || P >= 5) ? 24 : 20, strips = (P <= 1 || P >= 5) ? 3 : 2;
        for (auto& b : hf) {
            const int plane = rand() % planes, S0 = rand() % (NS - strips + 1), gy = rand() % (Hd - rows);
            b = (uint32_t)(plane * strip_plane + ((size_t)S0 * Hd + gy) * 16);
        }
        CK(hipMemcpy(d_f, hf.data(), 4 * hf.size(), hipMemcpyHostToDevice));
        float best = 1e9f;
        for (int rep = 0; rep < 5; ++rep) {
            CK(hipEventRecord(e0));
            if (P == 0) hipLaunchKernelGGL(k_bench<0>, dim3(768), dim3(256), 0, 0, d_arena, d_f, feats_per_wave, (uint32_t)Hd * 16, d_out);
            if (P == 1) hipLaunchKernelGGL(k_bench<1>, dim3(768), dim3(256), 0, 0, d_arena, d_f, feats_per_wave, (uint32_t)Hd * 16, d_out);
            if (P == 2) hipLaunchKernelGGL(k_bench<2>, dim3(768), dim3(256), 0, 0, d_arena, d_f, feats_per_wave, (uint32_t)Hd * 16, d_out);
            if (P == 3) hipLaunchKernelGGL(k_bench<3>, dim3(768), dim3(256), 0, 0, d_arena, d_f, feats_per_wave, (uint32_t)Hd * 16, d_out);
            if (P == 4) hipLaunchKernelGGL(k_bench<4>, dim3(768), dim3(256), 0, 0, d_arena, d_f, feats_per_wave, (uint32_t)Hd * 16, d_out);
            if (P == 5) hipLaunchKernelGGL(k_bench<5>, dim3(768), dim3(256), 0, 0, d_arena, d_f, feats_per_wave, (uint32_t)Hd * 16, d_out);
            if (P == 6) hipLaunchKernelGGL(k_bench<6>, dim3(768), dim3(256), 0, 0, d_arena, d_f, feats_per_wave, (uint32_t)Hd * 16, d_out);
            if (P == 7) hipLaunchKernelGGL(k_bench<7>, dim3(768), dim3(256), 0, 0, d_arena, d_f, feats_per_wave, (uint32_t)Hd * 16, d_out);
            if (P == 8) hipLaunchKernelGGL(k_bench<8>, dim3(768), dim3(256), 0, 0, d_arena, d_f, feats_per_wave, (uint32_t)Hd * 16, d_out);
            if (P == 9) hipLaunchKernelGGL(k_bench<9>, dim3(768), dim3(256), 0, 0, d_arena, d_f, feats_per_wave, (uint32_t)Hd * 16, d_out);
            if (P == 10) hipLaunchKernelGGL(k_bench<10>, dim3(768), dim3(256), 0, 0, d_arena, d_f, feats_per_wave, (uint32_t)Hd * 16, d_out);
            if (P == 11) hipLaunchKernelGGL(k_bench<11>, dim3(768), dim3(256), 0, 0, d_arena, d_f, feats_per_wave, (uint32_t)Hd * 16, d_out);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            if (ms < best) best = ms;
        }
        CK(hipGetLastError());
        const double loads = (double)waves * feats_per_wave / ((P <= 1 || P >= 5) ? 1 : 2);   // wave-level load instructions
        printf("pattern %d: %.3f ms, %.1f M wave loads -> %.2f ns per wave load (x 256 CUs = %.1f CU cycles at 2.4 GHz)\n", P, best, loads / 1e6, best * 1e6 / loads,
               best * 1e6 / loads * 256 * 2.4);
    }
    return 0;
}
