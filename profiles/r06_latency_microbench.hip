// Round 6: what a lone wave pays per DEPENDENT instruction on gfx950 (s_memtime ticks and ns, one wave on an idle CU).
// hipcc --offload-arch=gfx950 -O3 -o /tmp/lat profiles/r06_latency_microbench.hip && /tmp/lat
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <chrono>

#define N 2048
__global__ void k(double* out, long long* clk, int* idx, int mode) {
    __shared__ double s[1024];
    for (int i = threadIdx.x; i < 1024; i += blockDim.x) s[i] = (double)((i * 37) & 1023);
    __syncthreads();
    double a = out[threadIdx.x], b = 1.0000001, c = 0.5;
    int j = idx[threadIdx.x] & 1023;
    long long t0 = __builtin_amdgcn_s_memtime();
    if (mode == 0) { for (int i = 0; i < N; ++i) a = fma(a, b, c); }
    else if (mode == 1) { for (int i = 0; i < N; ++i) a = a * b + c; }
    else if (mode == 2) { for (int i = 0; i < N; ++i) { a = a < c ? a * b : a + b; } }
    else if (mode == 3) { for (int i = 0; i < N; ++i) { j = (int)s[j] & 1023; } a = j; }
    else if (mode == 4) { for (int i = 0; i < N; ++i) { a = __shfl_xor(a, 16, 64) + b; } }
    else if (mode == 5) { for (int i = 0; i < N; ++i) { int v = __builtin_amdgcn_update_dpp(0, j, 0xB1, 0xF, 0xF, false); j = v + 1; } a = j; }
    else if (mode == 6) { for (int i = 0; i < N; ++i) { a = sqrt(a) + b; } }
    else if (mode == 7) { for (int i = 0; i < N; ++i) { unsigned long long m = __ballot(a < c); a += (double)__popcll(m); } }
    else if (mode == 8) { int x = j; for (int i = 0; i < N; ++i) { x = x * 3 + 1; } a = x; }
    else if (mode == 9) { float f = (float)a; for (int i = 0; i < N; ++i) { f = fmaf(f, 1.0001f, 0.5f); } a = f; }
    long long t1 = __builtin_amdgcn_s_memtime();
    out[threadIdx.x] = a;
    if (threadIdx.x == 0) clk[0] = t1 - t0;
}
int main() {
    double* out; long long* clk; int* idx;
    hipMalloc(&out, 1024 * 8); hipMalloc(&clk, 8); hipMalloc(&idx, 1024 * 4);
    hipMemset(out, 0, 1024 * 8); hipMemset(idx, 0, 1024 * 4);
    const char* names[] = {"fma f64", "mul+add f64 (2 ops)", "cmp f64 + select of (mul | add)", "LDS read (f64) -> index", "ds_bpermute f64 (2) + add", "DPP mov + add i32", "sqrt f64 + add",
                           "ballot + popc + cvt + add", "mul+add i32", "fma f32"};
    for (int waves = 1; waves <= 4; waves *= 4)
    for (int mode = 0; mode < 10; ++mode) {
        long long c = 0;
        hipLaunchKernelGGL(k, dim3(1), dim3(64 * waves), 0, 0, out, clk, idx, mode);
        hipDeviceSynchronize();
        auto t0 = std::chrono::steady_clock::now();
        hipLaunchKernelGGL(k, dim3(1), dim3(64 * waves), 0, 0, out, clk, idx, mode);
        hipDeviceSynchronize();
        auto t1 = std::chrono::steady_clock::now();
        hipMemcpy(&c, clk, 8, hipMemcpyDeviceToHost);
        printf("%d wave(s)  %-34s %7.1f ticks per iteration (%lld ticks; launch+run wall %.1f us)\n", waves, names[mode], (double)c / N, c,
               std::chrono::duration<double, std::micro>(t1 - t0).count());
    }
    return 0;
}
