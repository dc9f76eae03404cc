// What would a node fetch cost in other forms?  Same set-up as node_fetch.hip (every enabled lane of a wave fetches
// one node from its own pseudo-random place, addresses from an integer recurrence, 8 waves per SIMD, 256-thread
// workgroups), for:
//   g32   32-byte node from global memory, two 16-byte loads per lane (what the node step does today)
//   g16   16-byte node, one 16-byte load per lane (a node form with 8-bit planes)
//   g8    8-byte load, g4  4-byte load (does a narrower load cost the address unit less?)
//   lds32 32-byte node from a 14-KB table in LDS, two ds_read_b128 per lane (the top of the tree kept in LDS)
//   mixNN 32-byte node: NN % of the lanes (chosen per fetch) read the LDS table, the others global memory
//   aXX   g32 written in assembly with a cache-policy bit set on the loads (nt, sc0, sc1, sc0 sc1): does any policy
//         make a miss cost less than a whole 128-byte line on the way from L2?
//   hipcc -O2 --offload-arch=gfx950 -o tools/ubench/node_fetch2 tools/ubench/node_fetch2.hip && tools/ubench/node_fetch2
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

#define TOP_NODES 448u

template <int MODE, int MIX>
__global__ __launch_bounds__(256) void k(const uint4 *nodes, uint32_t n_nodes_mask, int iters, int lanes, uint32_t *out) {
    __shared__ uint4 top[TOP_NODES * 2];
    for (uint32_t i = threadIdx.x; i < TOP_NODES * 2; i += 256) top[i] = nodes[i];
    __syncthreads();
    uint32_t x = (blockIdx.x * 256u + threadIdx.x) * 2654435761u + 12345u;
    uint32_t acc = 0;
    if ((int)(threadIdx.x & 63) < lanes) {
        for (int i = 0; i < iters; i++) {
#pragma unroll
            for (int r = 0; r < 4; r++) {
                x = x * 1664525u + 1013904223u;
                uint32_t idx = (x >> 8) & n_nodes_mask;
                if (MODE == 0) {
                    const uint4 *n = nodes + 2 * idx;
                    uint4 a = n[0], b = n[1];
                    acc += a.x ^ a.w ^ b.y ^ b.z;
                } else if (MODE == 1) {
                    uint4 a = nodes[idx];
                    acc += a.x ^ a.w ^ a.y ^ a.z;
                } else if (MODE == 2) {
                    uint2 a = ((const uint2 *)nodes)[idx];
                    acc += a.x ^ a.y;
                } else if (MODE == 3) {
                    acc += ((const uint32_t *)nodes)[idx];
                } else if (MODE == 4) {
                    const uint32_t j = ((x >> 8) % TOP_NODES) * 2u;
                    uint4 a = top[j], b = top[j + 1];
                    acc += a.x ^ a.w ^ b.y ^ b.z;
                } else {
                    // the choice of path differs from lane to lane and from fetch to fetch
                    const bool in_top = ((x >> 20) % 100u) < (uint32_t)MIX;
                    uint4 a, b;
                    if (in_top) {
                        const uint32_t j = ((x >> 8) % TOP_NODES) * 2u;
                        a = top[j]; b = top[j + 1];
                    } else {
                        const uint4 *n = nodes + 2 * idx;
                        a = n[0]; b = n[1];
                    }
                    acc += a.x ^ a.w ^ b.y ^ b.z;
                }
            }
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}

// POL: 0 plain, 1 nt, 2 sc0, 3 sc1, 4 sc0 sc1, 5 sc0 sc1 nt
template <int POL>
__device__ __forceinline__ void load_node(const uint4 *n, uint4 &a, uint4 &b) {
#define LD2(bits) asm volatile("global_load_dwordx4 %0, %2, off " bits "\n\tglobal_load_dwordx4 %1, %2, off offset:16 " bits "\n\ts_waitcnt vmcnt(0)" : "=&v"(a), "=&v"(b) : "v"(n) : "memory")
    if (POL == 0) LD2("");
    else if (POL == 1) LD2("nt");
    else if (POL == 2) LD2("sc0");
    else if (POL == 3) LD2("sc1");
    else if (POL == 4) LD2("sc0 sc1");
    else LD2("sc0 sc1 nt");
#undef LD2
}
template <int POL>
__global__ __launch_bounds__(256) void ka(const uint4 *nodes, uint32_t n_nodes_mask, int iters, int lanes, uint32_t *out) {
    uint32_t x = (blockIdx.x * 256u + threadIdx.x) * 2654435761u + 12345u;
    uint32_t acc = 0;
    if ((int)(threadIdx.x & 63) < lanes) {
        for (int i = 0; i < iters; i++) {
#pragma unroll
            for (int r = 0; r < 4; r++) {
                x = x * 1664525u + 1013904223u;
                uint32_t idx = (x >> 8) & n_nodes_mask;
                uint4 a, b;
                load_node<POL>(nodes + 2 * idx, a, b);
                acc += a.x ^ a.w ^ b.y ^ b.z;
            }
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}

typedef void (*kern_t)(const uint4 *, uint32_t, int, int, uint32_t *);

int main() {
    int cus = 256, khz = 2400000;
    hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
    hipDeviceGetAttribute(&khz, hipDeviceAttributeClockRate, 0);
    const int wps = 8, iters = 1024, per_iter = 4;
    const size_t big = 32u << 20;
    uint4 *nodes; uint32_t *out;
    if (hipMalloc(&nodes, big) != hipSuccess || hipMalloc(&out, sizeof(uint32_t) * cus * wps * 256) != hipSuccess) return 1;
    hipMemset(nodes, 1, big);
    struct { const char *name; kern_t fn; int bytes; } modes[] = {
        {"g32", k<0, 0>, 32}, {"g16", k<1, 0>, 16}, {"g8", k<2, 0>, 8}, {"g4", k<3, 0>, 4}, {"lds32", k<4, 0>, 32},
        {"mix15", k<5, 15>, 32}, {"mix30", k<5, 30>, 32}, {"mix50", k<5, 50>, 32},
        {"a32", ka<0>, 32}, {"a32nt", ka<1>, 32}, {"a32sc0", ka<2>, 32}, {"a32sc1", ka<3>, 32}, {"a32s01", ka<4>, 32}, {"a32all", ka<5>, 32}};
    // array sizes in elements of the mode's own width are derived from a byte size
    struct { const char *name; size_t bytes; } arrays[] = {{"32 KB", 32u << 10}, {"2 MB", 2u << 20}, {"32 MB", 32u << 20}};
    const int lane_cases[] = {64, 29};
    printf("%d CUs at %.0f MHz, %d waves per SIMD\n", cus, khz / 1e3, wps);
    for (auto &ar : arrays)
        for (auto &m : modes)
            for (int lanes : lane_cases) {
                const uint32_t mask = (uint32_t)(ar.bytes / (size_t)m.bytes) - 1u;
                m.fn<<<cus * wps, 256>>>(nodes, mask, 8, lanes, out);
                hipDeviceSynchronize();
                hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
                hipEventRecord(e0);
                m.fn<<<cus * wps, 256>>>(nodes, mask, iters, lanes, out);
                hipEventRecord(e1);
                hipDeviceSynchronize();
                float ms; hipEventElapsedTime(&ms, e0, e1);
                double wave_fetches_per_cu = (double)iters * per_iter * wps * 4;
                double ns = ms * 1e6 / wave_fetches_per_cu;
                printf("%-6s %-6s %2d lanes: %7.2f ns per wave fetch per CU (%6.1f cycles), %.2f cycles per lane\n", ar.name, m.name, lanes, ns,
                       ns * khz / 1e6, ns * khz / 1e6 / lanes);
                hipEventDestroy(e0); hipEventDestroy(e1);
            }
    hipFree(nodes); hipFree(out);
    return 0;
}
