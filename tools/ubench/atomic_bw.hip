// Microbenchmark: fp32 global atomic-add (no return) throughput on gfx950 for the
// access patterns of the gradient scatter.  hipcc --offload-arch=gfx950 -O3 -munsafe-fp-atomics
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

__device__ inline uint32_t hash(uint32_t x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }

// mode 0: each lane 8 atomics at random voxel's 2x2x2 corners (grid res^3)
// mode 1: same but lanes of a wave share the same base voxel +- small jitter (coherent)
// mode 2: plain random single atomics
// mode 3: random loads (8 corners) instead of atomics, result reduced
template <int MODE>
__global__ void __launch_bounds__(256) k(float *g, int res, int iters, float *sink)
{
    uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t s = hash(tid * 9781u + 12345u);
    float acc = 0.f;
    int sy = res, sz = res * res;
    for (int it = 0; it < iters; ++it) {
        s = hash(s + it);
        uint32_t base;
        if (MODE == 1) {
            uint32_t ws = hash((tid >> 6) * 7919u + it);
            base = ws;
            s = hash(s);
        } else base = s;
        int x = base % (res - 1), y = (base / res) % (res - 1), z = (base / (res * res)) % (res - 1);
        if (MODE == 1) { x = min(res - 2, x + (int)(s & 3)); y = min(res - 2, y + (int)((s >> 2) & 1)); }
        int i0 = z * sz + y * sy + x;
        if (MODE == 2) { atomicAdd(g + i0, 1.0f); continue; }
        if (MODE == 3) {
            acc += g[i0] + g[i0 + 1] + g[i0 + sy] + g[i0 + sy + 1] + g[i0 + sz] + g[i0 + sz + 1] + g[i0 + sz + sy] + g[i0 + sz + sy + 1];
            continue;
        }
        atomicAdd(g + i0, 1.0f); atomicAdd(g + i0 + 1, 1.0f);
        atomicAdd(g + i0 + sy, 1.0f); atomicAdd(g + i0 + sy + 1, 1.0f);
        atomicAdd(g + i0 + sz, 1.0f); atomicAdd(g + i0 + sz + 1, 1.0f);
        atomicAdd(g + i0 + sz + sy, 1.0f); atomicAdd(g + i0 + sz + sy + 1, 1.0f);
    }
    if (MODE == 3 && acc == 12345.678f) sink[0] = acc;
}

template <int MODE>
void run(const char *name, float *g, int res, int blocks, int iters, float *sink, int per_iter)
{
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, g, res, 2, sink);
    hipDeviceSynchronize();
    hipEventRecord(a);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, g, res, iters, sink);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    double n = (double) blocks * 256 * iters * per_iter;
    printf("%-34s res %3d  %8.3f ms  %7.1f G ops/s\n", name, res, ms, n / ms / 1e6);
}

int main()
{
    for (int res : {64, 256, 512}) {
        float *g, *sink; size_t n = (size_t) res * res * res;
        hipMalloc(&g, n * 4); hipMemset(g, 0, n * 4); hipMalloc(&sink, 4);
        int blocks = 256 * 16, iters = 64;
        run<0>("atomic 8-corner random", g, res, blocks, iters, sink, 8);
        run<1>("atomic 8-corner wave-coherent", g, res, blocks, iters, sink, 8);
        run<2>("atomic single random", g, res, blocks, iters * 8, sink, 1);
        run<3>("load 8-corner random", g, res, blocks, iters, sink, 8);
        hipFree(g); hipFree(sink);
    }
    return 0;
}
