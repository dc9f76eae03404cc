// Vector-ALU issue rate on gfx950: cycles per wave64 instruction per SIMD as a function of the
// number of waves resident on the SIMD, for v_fma_f32, v_pk_fma_f32, v_min/v_max, v_cndmask mixes.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef float v2f __attribute__((ext_vector_type(2)));

template <int KIND>
__global__ __launch_bounds__(256) void k(float *out, unsigned long long *cyc, int iters) {
    float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    v2f p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7};
    float m = 1.0001f, c = 0.5f;
    v2f pm = {m, m}, pc = {c, c};
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; i++) {
        if (KIND == 0) {
#pragma unroll
            for (int r = 0; r < 4; r++) {
                asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                             "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m), "v"(c));
            }
        } else if (KIND == 1) {
#pragma unroll
            for (int r = 0; r < 8; r++) {
                asm volatile("v_pk_fma_f32 %0, %0, %4, %5\n v_pk_fma_f32 %1, %1, %4, %5\n v_pk_fma_f32 %2, %2, %4, %5\n v_pk_fma_f32 %3, %3, %4, %5\n"
                             : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(pm), "v"(pc));
            }
        } else if (KIND == 2) {
#pragma unroll
            for (int r = 0; r < 4; r++) {
                asm volatile("v_max_f32 %0, %0, %8\n v_min_f32 %1, %1, %9\n v_max_f32 %2, %2, %8\n v_min_f32 %3, %3, %9\n"
                             "v_max_f32 %4, %4, %8\n v_min_f32 %5, %5, %9\n v_max_f32 %6, %6, %8\n v_min_f32 %7, %7, %9\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m), "v"(c));
            }
        } else if (KIND == 3) {   // compare + select pairs
#pragma unroll
            for (int r = 0; r < 4; r++) {
                asm volatile("v_cmp_lt_f32 vcc, %0, %8\n v_cndmask_b32 %1, %1, %9, vcc\n v_cmp_lt_f32 vcc, %2, %8\n v_cndmask_b32 %3, %3, %9, vcc\n"
                             "v_cmp_lt_f32 vcc, %4, %8\n v_cndmask_b32 %5, %5, %9, vcc\n v_cmp_lt_f32 vcc, %6, %8\n v_cndmask_b32 %7, %7, %9, vcc\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m), "v"(c) : "vcc");
            }
        } else if (KIND == 4) {   // integer: 32-bit multiply-add and shifts
            uint32_t u0 = __float_as_uint(a0), u1 = __float_as_uint(a1), u2 = __float_as_uint(a2), u3 = __float_as_uint(a3);
#pragma unroll
            for (int r = 0; r < 8; r++) {
                asm volatile("v_mul_lo_u32 %0, %0, %4\n v_mul_lo_u32 %1, %1, %4\n v_mul_lo_u32 %2, %2, %4\n v_mul_lo_u32 %3, %3, %4\n"
                             : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3) : "v"(__float_as_uint(m)));
            }
            a0 = __uint_as_float(u0); a1 = __uint_as_float(u1); a2 = __uint_as_float(u2); a3 = __uint_as_float(u3);
        } else if (KIND == 5) {   // v_rcp_f32
#pragma unroll
            for (int r = 0; r < 4; r++) {
                asm volatile("v_rcp_f32 %0, %0\n v_rcp_f32 %1, %1\n v_rcp_f32 %2, %2\n v_rcp_f32 %3, %3\n"
                             "v_rcp_f32 %4, %4\n v_rcp_f32 %5, %5\n v_rcp_f32 %6, %6\n v_rcp_f32 %7, %7\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
            }
        } else if (KIND == 6) {   // VALU interleaved with SALU
#pragma unroll
            for (int r = 0; r < 4; r++) {
                asm volatile("v_fma_f32 %0, %0, %8, %9\n s_nop 0\n v_fma_f32 %1, %1, %8, %9\n s_nop 0\n v_fma_f32 %2, %2, %8, %9\n s_nop 0\n v_fma_f32 %3, %3, %8, %9\n s_nop 0\n"
                             "v_fma_f32 %4, %4, %8, %9\n s_nop 0\n v_fma_f32 %5, %5, %8, %9\n s_nop 0\n v_fma_f32 %6, %6, %8, %9\n s_nop 0\n v_fma_f32 %7, %7, %8, %9\n s_nop 0\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m), "v"(c));
            }
        } else if (KIND == 7) {   // 64-bit multiply (v_mad_u64_u32)
            unsigned long long q0 = threadIdx.x, q1 = q0 + 1, q2 = q0 + 2, q3 = q0 + 3;
            uint32_t mm = __float_as_uint(m);
#pragma unroll
            for (int r = 0; r < 8; r++) {
                asm volatile("v_mad_u64_u32 %0, vcc, %4, %4, %0\n v_mad_u64_u32 %1, vcc, %4, %4, %1\n v_mad_u64_u32 %2, vcc, %4, %4, %2\n v_mad_u64_u32 %3, vcc, %4, %4, %3\n"
                             : "+v"(q0), "+v"(q1), "+v"(q2), "+v"(q3) : "v"(mm) : "vcc");
            }
            a0 = (float)(q0 + q1 + q2 + q3);
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float sum = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + p0.x + p0.y + p1.x + p1.y + p2.x + p2.y + p3.x + p3.y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = sum;
    if ((threadIdx.x & 63) == 0) cyc[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = t1 - t0;
}

template <int KIND>
void run(const char *name, int cus) {
    const int iters = 4096, per_iter = 32;
    float *out; unsigned long long *cyc;
    hipMalloc(&out, sizeof(float) * cus * 8 * 256);
    hipMalloc(&cyc, 8 * cus * 8 * 4);
    for (int wps = 1; wps <= 8; wps *= 2) {
        // blocks of 256 threads = 4 waves = one wave per SIMD; wps blocks per CU
        int blocks = cus * wps;
        k<KIND><<<blocks, 256>>>(out, cyc, 16);
        hipDeviceSynchronize();
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0);
        k<KIND><<<blocks, 256>>>(out, cyc, iters);
        hipEventRecord(e1);
        hipDeviceSynchronize();
        float ms; hipEventElapsedTime(&ms, e0, e1);
        unsigned long long h[64];
        hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
        double avg = 0; for (int i = 0; i < 64; i++) avg += (double)h[i]; avg /= 64;
        double n = (double)iters * per_iter;
        printf("%-10s waves/SIMD %d  wall %.3f ms  memtime ticks/instr/wave %.2f  -> ticks per instr per SIMD %.2f  wall ns per instr per SIMD %.3f\n",
               name, wps, ms, avg / n, avg / n / wps, ms * 1e6 / n / wps);
    }
    hipFree(out); hipFree(cyc);
}

int main() {
    int cus = 256; hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
    int clk = 0; hipDeviceGetAttribute(&clk, hipDeviceAttributeClockRate, 0);
    printf("CUs %d, clock %d kHz\n", cus, clk);
    run<0>("fma", cus); run<1>("pk_fma", cus); run<2>("minmax", cus); run<3>("cmp+sel", cus);
    run<4>("mul_lo", cus); run<5>("rcp", cus); run<6>("fma+snop", cus); run<7>("mad64", cus);
    return 0;
}
