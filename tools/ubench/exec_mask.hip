// Does a wave64 vector instruction cost less when only part of the wave is enabled?  v_fma_f32 throughput per
// SIMD at 8 waves per SIMD with the first `active` lanes of every wave enabled (contiguous) or every k-th lane.
//   hipcc -O2 --offload-arch=gfx950 -o tools/ubench/exec_mask tools/ubench/exec_mask.hip && tools/ubench/exec_mask
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

__global__ __launch_bounds__(256) void k(float *out, int iters, unsigned long long mask) {
    float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    float m = 1.0001f, c = 0.5f;
    if ((mask >> (threadIdx.x & 63)) & 1ull) {
        for (int i = 0; i < iters; i++) {
#pragma unroll
            for (int r = 0; r < 4; r++) {
                asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                             "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m), "v"(c));
            }
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
}

int main() {
    int cus = 256; hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
    const int iters = 4096, per_iter = 32, wps = 8;
    float *out; hipMalloc(&out, sizeof(float) * cus * wps * 256);
    struct { const char *name; unsigned long long mask; } cases[] = {
        {"all 64 lanes", ~0ull}, {"lanes 0-31", 0xFFFFFFFFull}, {"lanes 32-63", 0xFFFFFFFF00000000ull}, {"lanes 0-15", 0xFFFFull},
        {"lane 0", 1ull}, {"every 2nd lane", 0x5555555555555555ull}, {"every 4th lane", 0x1111111111111111ull}, {"lanes 16-31", 0xFFFF0000ull}};
    for (auto &cs : cases) {
        k<<<cus * wps, 256>>>(out, 16, cs.mask);
        hipDeviceSynchronize();
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0);
        k<<<cus * wps, 256>>>(out, iters, cs.mask);
        hipEventRecord(e1);
        hipDeviceSynchronize();
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("%-16s wall %.3f ms  ns per wave-instruction per SIMD %.3f\n", cs.name, ms, ms * 1e6 / ((double)iters * per_iter) / wps);
    }
    hipFree(out);
    return 0;
}
