// LDS atomic throughput on gfx950 against the number of DISTINCT addresses the 64 lanes of an instruction hit (the nerf tile adjoint's
// window adds: neighbouring pixels splat around the same few voxels).  ds_add_f32 / ds_add_u32 / ds_add_u64, no return.
// hipcc --offload-arch=gfx950 -O3 -munsafe-fp-atomics lds_atomic_conflict_rate.hip -o lds_atomic_conflict_rate && ./lds_atomic_conflict_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

constexpr int kTile = 4432;

template <int MODE>
__global__ void __launch_bounds__(512) k(int iters, int naddr, float *sink)
{
    __shared__ unsigned long long tile64[MODE == 2 ? kTile : 1];
    __shared__ float tilef[MODE == 0 ? kTile : 1];
    __shared__ uint32_t tileu[MODE == 1 ? kTile : 1];
    for (int j = threadIdx.x; j < kTile; j += 512) { if (MODE == 0) tilef[j] = 0; else if (MODE == 2) tile64[j] = 0; else tileu[j] = 0; }
    __syncthreads();
    uint32_t s = threadIdx.x * 2654435761u + blockIdx.x * 40503u + 12345u;
    for (int it = 0; it < iters; ++it) {
        s = s * 1664525u + 1013904223u;
        // the wave's lanes draw from naddr addresses (17 apart: different banks) around a base that moves with the iteration
        const uint32_t a = ((s >> 8) % (uint32_t) naddr) * 17u + (uint32_t) (it & 63);
        const int offs[8] = { 0, 1, 17, 18, 277, 278, 294, 295 };
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            const uint32_t i = (a + offs[c]) % (kTile - 1);
            if (MODE == 0) atomicAdd(&tilef[i], 1.0f + c);
            else if (MODE == 1) atomicAdd(&tileu[i], 1u + c);
            else atomicAdd(&tile64[i], 1ull + c);
        }
    }
    __syncthreads();
    float acc = 0;
    for (int j = threadIdx.x; j < kTile; j += 512) acc += MODE == 0 ? tilef[j] : MODE == 2 ? (float) tile64[j] : (float) tileu[j];
    if (acc == -1.0f) sink[0] = acc;
}

template <int MODE>
void run(const char *name)
{
    float *sink; hipMalloc(&sink, 4);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    const int iters = 1024, blocks = 1024;
    for (int naddr : { 1, 2, 4, 8, 16, 32, 64, 200 }) {
        hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(512), 0, 0, 16, naddr, sink);
        hipEventRecord(a);
        hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(512), 0, 0, iters, naddr, sink);
        hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        double n = (double) blocks * 512 * iters * 8;
        printf("%-12s %4d addresses per instruction: %8.3f ms  %8.1f G lane-atomics/s\n", name, naddr, ms, n / ms / 1e6);
    }
    hipFree(sink);
}

int main()
{
    run<0>("ds_add_f32");
    run<1>("ds_add_u32");
    run<2>("ds_add_u64");
    return 0;
}
