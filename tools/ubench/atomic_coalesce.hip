// Microbenchmark: does the fp32 atomic coalescer need ADJACENT lanes for same-line merging?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__device__ inline uint32_t hash(uint32_t x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }

// MODE 0: lanes (i, i+32) form a contiguous pair            -> same line, non-adjacent lanes
// MODE 1: wave covers 64 contiguous floats, lane->slot = bit-reversed lane   (scrambled order)
// MODE 2: groups of 8 lanes cover 8 floats spread as 2 floats in each of 4 random lines (sigma_t splat shape, x-pairs adjacent lanes)
// MODE 3: like 2 but the lanes of a pair are 4 apart (lane j and j+4)
// MODE 4: groups of 8 lanes: 4 floats in each of 2 random lines (4x2x2 brick shape)
// MODE 5: groups of 8 lanes: all 8 floats in 1 random line
template <int MODE>
__global__ void __launch_bounds__(256) k(float *g, uint32_t n, int iters)
{
    uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t lane = threadIdx.x & 63, wave = tid >> 6;
    uint32_t lines = n / 16;
    for (int it = 0; it < iters; ++it) {
        uint32_t idx;
        if (MODE == 0) { uint32_t pr = lane & 31; uint32_t s = hash(wave * 64 + pr + it * 7919u); idx = (s % (n / 2 - 1)) * 2 + (lane >> 5); }
        else if (MODE == 1) { uint32_t s = hash(wave + it * 7919u); uint32_t r = __brev(lane) >> 26; idx = (s % (lines / 4 - 1)) * 64 + r; }
        else {
            uint32_t grp = lane >> 3, j = lane & 7;
            uint32_t sub, off;
            if (MODE == 2) { sub = j >> 1; off = j & 1; }
            else if (MODE == 3) { sub = j & 3; off = j >> 2; }
            else if (MODE == 4) { sub = j >> 2; off = j & 3; }
            else { sub = 0; off = j; }
            uint32_t s = hash((wave * 8 + grp) * 131u + sub * 17u + it * 7919u);
            idx = (s % (lines - 1)) * 16 + (s >> 28 & 7) + off;
        }
        atomicAdd(g + idx, 1.0f);
    }
}
template <int MODE>
void run(const char *name, float *g, uint32_t n)
{
    int blocks = 4096, iters = 512;
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL((k<MODE>), dim3(blocks), dim3(256), 0, 0, g, n, 4);
    hipDeviceSynchronize();
    hipEventRecord(a);
    hipLaunchKernelGGL((k<MODE>), dim3(blocks), dim3(256), 0, 0, g, n, iters);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    printf("%-58s %8.3f ms %7.1f G lane-atomics/s\n", name, ms, (double) blocks * 256 * iters / ms / 1e6);
}
int main()
{
    uint32_t n = 64u << 20; float *g; hipMalloc(&g, (size_t) n * 4); hipMemset(g, 0, (size_t) n * 4);
    run<0>("pairs on lanes (i, i+32)", g, n);
    run<1>("64 contiguous floats, bit-reversed lane order", g, n);
    run<2>("8-lane splat: 4 lines x 2 floats, pair lanes adjacent", g, n);
    run<3>("8-lane splat: 4 lines x 2 floats, pair lanes 4 apart", g, n);
    run<4>("8-lane splat: 2 lines x 4 floats", g, n);
    run<5>("8-lane splat: 1 line x 8 floats", g, n);
    return 0;
}
