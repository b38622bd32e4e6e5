// Correctness of ds_add_u64 under heavy same-address contention (tiny tiles): every thread adds known
// values to a handful of LDS addresses; the totals must be exact.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__global__ void __launch_bounds__(256) k(int iters, int naddr, unsigned long long *out)
{
    extern __shared__ unsigned long long tile[];
    for (int j = threadIdx.x; j < 9537; j += 256) tile[j] = 0ull;
    __syncthreads();
    uint32_t s = threadIdx.x * 2654435761u + blockIdx.x * 40503u + 12345u;
    for (int it = 0; it < iters; ++it) {
        s = s * 1664525u + 1013904223u;
        uint32_t a = (s >> 8) % naddr;
        long long v = (long long) ((s >> 3) & 1023) - 512;          // signed values
        atomicAdd(&tile[a * 7 % 9537], (unsigned long long) v);
        atomicAdd(&tile[(a * 7 + 1) % 9537], (unsigned long long) (2 * v));
    }
    __syncthreads();
    for (int j = threadIdx.x; j < 9537; j += 256) if (tile[j]) atomicAdd(out + j, tile[j]);
}
int main()
{
    for (int naddr : {1, 4, 27, 64, 1000}) {
        unsigned long long *d; hipMalloc(&d, 9537 * 8); hipMemset(d, 0, 9537 * 8);
        const int iters = 4096, blocks = 512;
        hipFuncSetAttribute((const void *) k, hipFuncAttributeMaxDynamicSharedMemorySize, 9537 * 8);
        hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 9537 * 8, 0, iters, naddr, d);
        static unsigned long long h[9537]; hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
        static long long ref[9537]; for (auto &r : ref) r = 0;
        for (int b = 0; b < blocks; ++b) for (int t = 0; t < 256; ++t) {
            uint32_t s = t * 2654435761u + b * 40503u + 12345u;
            for (int it = 0; it < iters; ++it) { s = s * 1664525u + 1013904223u; uint32_t a = (s >> 8) % naddr; long long v = (long long) ((s >> 3) & 1023) - 512;
                ref[a * 7 % 9537] += v; ref[(a * 7 + 1) % 9537] += 2 * v; }
        }
        int bad = 0; for (int j = 0; j < 9537; ++j) if ((long long) h[j] != ref[j]) ++bad;
        printf("naddr %5d: mismatching entries %d\n", naddr, bad);
        hipFree(d);
    }
    return 0;
}
