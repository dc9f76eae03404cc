// How fast can a CU fetch tree nodes?  Every enabled lane of a wave loads the 32 bytes of one node (two 16-byte
// loads, as the traversal's node step does) from its own pseudo-random place in an array of `bytes` bytes; the
// next address comes from an integer recurrence, not from the data, so the loads of a wave queue up behind each
// other and the rate is what the vector memory path delivers, not a latency.  8 waves per SIMD, as the spheres kernel.
// Cases: a 31-KB array (the 484 nodes of the random-spheres scene: L1-resident) and a 32-MB one (a million spheres:
// beyond the 4-MB L2 of an XCD), with 64 / 41 / 27 lanes of a wave enabled.
//   hipcc -O2 --offload-arch=gfx950 -o tools/ubench/node_fetch tools/ubench/node_fetch.hip && tools/ubench/node_fetch
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

// share: 1 = every lane its own node; 4 = four neighbouring lanes take the four nodes of one 128-byte line;
// 64 = the whole wave reads one node
__global__ __launch_bounds__(256) void k(const uint4 *nodes, uint32_t n_nodes_mask, int iters, int lanes, int share, uint32_t *out) {
    uint32_t x = (blockIdx.x * 256u + threadIdx.x) * 2654435761u + 12345u;
    uint32_t acc = 0;
    if ((int)(threadIdx.x & 63) < lanes) {
        for (int i = 0; i < iters; i++) {
#pragma unroll
            for (int r = 0; r < 4; r++) {
                x = x * 1664525u + 1013904223u;
                uint32_t idx = (x >> 8) & n_nodes_mask;
                if (share == 4) idx = ((uint32_t)__shfl((int)idx, (int)(threadIdx.x & 60u), 64) & ~3u) | (threadIdx.x & 3u);
                if (share == 64) idx = (uint32_t)__shfl((int)idx, 0, 64);
                const uint4 *n = nodes + 2 * idx;
                uint4 a = n[0], b = n[1];
                acc += a.x ^ a.w ^ b.y ^ b.z;
            }
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}

int main() {
    int cus = 256, khz = 2400000;
    hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
    hipDeviceGetAttribute(&khz, hipDeviceAttributeClockRate, 0);
    const int wps = 8, iters = 2048, per_iter = 4;
    const size_t big = 32u << 20;
    uint4 *nodes; uint32_t *out;
    if (hipMalloc(&nodes, big) != hipSuccess || hipMalloc(&out, sizeof(uint32_t) * cus * wps * 256) != hipSuccess) return 1;
    hipMemset(nodes, 1, big);
    struct { const char *name; uint32_t mask; } arrays[] = {{"32 KB (1024 nodes)", 1023u}, {"32 MB (1M nodes)", (1u << 20) - 1}};
    const int lane_cases[] = {64, 41, 27};
    const int share_cases[] = {1, 4, 64};
    printf("%d CUs at %.0f MHz, %d waves per SIMD\n", cus, khz / 1e3, wps);
    for (auto &ar : arrays)
        for (int share : share_cases)
        for (int lanes : lane_cases) {
            if (share != 1 && lanes != 64) continue;
            k<<<cus * wps, 256>>>(nodes, ar.mask, 8, lanes, share, out);
            hipDeviceSynchronize();
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            hipEventRecord(e0);
            k<<<cus * wps, 256>>>(nodes, ar.mask, iters, lanes, share, out);
            hipEventRecord(e1);
            hipDeviceSynchronize();
            float ms; hipEventElapsedTime(&ms, e0, e1);
            double wave_fetches_per_cu = (double)iters * per_iter * wps * 4;      // node fetches of whole waves per CU
            double ns = ms * 1e6 / wave_fetches_per_cu;
            printf("%-20s %2d lanes, %2d lanes to a line or node: %.2f ns per wave node fetch per CU (%.1f cycles), %.1f B/cycle/CU, chip %.2f TB/s\n", ar.name, lanes, share,
                   ns, ns * khz / 1e6, 32.0 * lanes / (ns * khz / 1e6), 32.0 * lanes * cus / ns / 1e3);
            hipEventDestroy(e0); hipEventDestroy(e1);
        }
    hipFree(nodes); hipFree(out);
    return 0;
}
