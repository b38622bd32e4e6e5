// Microbenchmark: how does the fp32 global atomic rate depend on the address pattern inside one
// wave instruction?  G lane-atomics/s for contiguous groups of size 1,2,4,6,8,16,64 and same-address.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__device__ inline uint32_t hash(uint32_t x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }

// lanes are split into groups of GROUP consecutive lanes; each group picks a random base (aligned
// to GROUP floats if ALIGNED) and lane j of the group adds to base + j (SAME: to base).
template <int GROUP, bool SAME>
__global__ void __launch_bounds__(256) k(float *g, uint32_t n, int iters)
{
    uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t grp = tid / GROUP, j = tid % GROUP;
    uint32_t s = hash(grp * 9781u + 12345u);
    for (int it = 0; it < iters; ++it) {
        s = hash(s + it);
        uint32_t base = (s % (n / GROUP - 1)) * GROUP;
        atomicAdd(g + base + (SAME ? 0 : j), 1.0f);
    }
}
template <int GROUP, bool SAME>
void run(float *g, uint32_t n)
{
    int blocks = 4096, iters = 512;
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL((k<GROUP, SAME>), dim3(blocks), dim3(256), 0, 0, g, n, 4);
    hipDeviceSynchronize();
    hipEventRecord(a);
    hipLaunchKernelGGL((k<GROUP, SAME>), dim3(blocks), dim3(256), 0, 0, g, n, iters);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    printf("group %2d %-10s %8.3f ms  %7.1f G lane-atomics/s\n", GROUP, SAME ? "same-addr" : "contiguous", ms, (double) blocks * 256 * iters / ms / 1e6);
}
int main()
{
    uint32_t n = 64u << 20; float *g; hipMalloc(&g, (size_t) n * 4); hipMemset(g, 0, (size_t) n * 4);
    run<1, false>(g, n); run<2, false>(g, n); run<4, false>(g, n); run<8, false>(g, n);
    run<16, false>(g, n); run<32, false>(g, n); run<64, false>(g, n);
    run<2, true>(g, n); run<8, true>(g, n); run<64, true>(g, n);
    return 0;
}
