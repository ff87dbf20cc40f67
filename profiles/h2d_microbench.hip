// H2D of VGA RGB-D frames (921600 + 614400 bytes each) from pinned memory, back to back, as the ingest ring does: ONE copy per frame on one stream against the frame
// cut in 2 / 3 / 4 pieces on as many streams (one SDMA engine each?).  hipcc --offload-arch=gfx950 -O2 -o bin_tmp/h2d profiles/h2d_microbench.hip && bin_tmp/h2d
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
int main() {
    const size_t N = 921600 + 614400;
    const int ring = 16, frames = 400;
    char* h[ring]; char* d[ring];
    for (int i = 0; i < ring; ++i) { CK(hipHostMalloc((void**)&h[i], N, hipHostMallocDefault)); memset(h[i], i, N); CK(hipMalloc((void**)&d[i], N)); }
    hipStream_t s[4];
    for (auto& q : s) CK(hipStreamCreateWithFlags(&q, hipStreamNonBlocking));
    for (int rep = 0; rep < 2; ++rep)
        for (int parts = 1; parts <= 4; ++parts) {
            for (auto& q : s) CK(hipStreamSynchronize(q));
            const auto t0 = std::chrono::steady_clock::now();
            for (int f = 0; f < frames; ++f) {
                const int r = f % ring;
                for (int p = 0; p < parts; ++p) {
                    const size_t lo = (N * p / parts) & ~size_t(255), hi = p + 1 == parts ? N : (N * (p + 1) / parts) & ~size_t(255);
                    CK(hipMemcpyAsync(d[r] + lo, h[r] + lo, hi - lo, hipMemcpyHostToDevice, s[p]));
                }
            }
            const auto t1 = std::chrono::steady_clock::now();
            for (auto& q : s) CK(hipStreamSynchronize(q));
            const auto t2 = std::chrono::steady_clock::now();
            const double enq = std::chrono::duration<double>(t1 - t0).count() / frames * 1e6, tot = std::chrono::duration<double>(t2 - t0).count() / frames * 1e6;
            printf("streams %d: %.1f us per frame (%.1f GB/s), host enqueue %.1f us per frame\n", parts, tot, N / tot / 1e3, enq);
        }
    return 0;
}
