// LDS atomic throughput on gfx950: ds_add_f32 vs ds_add_u32 vs ds_add_u64, random addresses in a
// 9537-entry tile (the deferred-splat reduction's access pattern).  hipcc --offload-arch=gfx950 -O3
// -munsafe-fp-atomics lds_atomic_rate.hip -o lds_atomic_rate && ./lds_atomic_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

constexpr int kTile = 9537;

template <int MODE>
__global__ void __launch_bounds__(256) k(int iters, float *sink)
{
    __shared__ unsigned long long tile64[MODE == 2 ? kTile : 1];
    __shared__ float tilef[MODE == 0 ? kTile : 1];
    __shared__ uint32_t tileu[MODE == 1 || MODE == 3 ? kTile : 1];
    for (int j = threadIdx.x; j < kTile; j += 256) {
        if (MODE == 0) tilef[j] = 0; else if (MODE == 2) tile64[j] = 0; else tileu[j] = 0;
    }
    __syncthreads();
    uint32_t s = threadIdx.x * 2654435761u + blockIdx.x * 40503u + 12345u;
    for (int it = 0; it < iters; ++it) {
        s = s * 1664525u + 1013904223u;
        uint32_t a = (s >> 8) % (kTile - 600);
        const int offs[8] = { 0, 1, 33, 34, 561, 562, 594, 595 };
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            if (MODE == 0) atomicAdd(&tilef[a + offs[c]], 1.0f + c);
            else if (MODE == 1) atomicAdd(&tileu[a + offs[c]], 1u + c);
            else if (MODE == 2) atomicAdd(&tile64[a + offs[c]], 1ull + c);
            else { uint32_t old = tileu[a + offs[c]]; tileu[a + offs[c]] = old + 1u + c; }   // racy RMW (rate reference)
        }
    }
    __syncthreads();
    float acc = 0;
    for (int j = threadIdx.x; j < kTile; j += 256) acc += MODE == 0 ? tilef[j] : MODE == 2 ? (float) tile64[j] : (float) tileu[j];
    if (acc == -1.0f) sink[0] = acc;
}

template <int MODE>
void run(const char *name)
{
    float *sink; hipMalloc(&sink, 4);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    const int iters = 2048, blocks = 1024;
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, 16, sink);
    hipEventRecord(a);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, iters, sink);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    double n = (double) blocks * 256 * iters * 8;
    printf("%-28s %8.3f ms  %8.1f G lane-atomics/s\n", name, ms, n / ms / 1e6);
    hipFree(sink);
}

int main()
{
    run<0>("ds_add_f32 (no return)");
    run<1>("ds_add_u32 (no return)");
    run<2>("ds_add_u64 (no return)");
    run<3>("plain ds_read + ds_write");
    return 0;
}
