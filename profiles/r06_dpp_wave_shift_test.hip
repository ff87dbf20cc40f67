#include <hip/hip_runtime.h>
__global__ void k(int* out) {
    int v = threadIdx.x * 3 + 1;
    int a = __builtin_amdgcn_update_dpp(0, v, 0x130, 0xF, 0xF, false);   // wave_shl:1
    int b = __builtin_amdgcn_update_dpp(0, v, 0x138, 0xF, 0xF, false);   // wave_shr:1
    out[threadIdx.x] = a; out[64 + threadIdx.x] = b;
}
int main() {
    int* d; hipMalloc(&d, 128 * 4); k<<<1, 64>>>(d); int h[128]; hipMemcpy(h, d, 512, hipMemcpyDeviceToHost);
    for (int i : {0, 1, 15, 16, 17, 31, 32, 62, 63}) printf("lane %d: shl %d shr %d (own %d)\n", i, h[i], h[64 + i], i * 3 + 1);
    return 0;
}
