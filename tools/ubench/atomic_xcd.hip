// Microbenchmark: XCD-private accumulation with workgroup-scope (L2-resident) fp32 atomics
// vs agent-scope atomics.  Checks that nothing is lost.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <numeric>

__device__ inline uint32_t hash(uint32_t x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }

__device__ inline uint32_t xcc_id() { return __builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | ((4 - 1) << 11)) & 0xf; }  // HW_REG_XCC_ID = 20, bits [3:0]

// SCOPE 0: agent (atomicAdd), 1: workgroup scope into the XCD-private copy, 2: wavefront scope
template <int SCOPE, bool COHERENT>
__global__ void __launch_bounds__(256) k(float *g, size_t copy_stride, int res, int iters, uint32_t *xcc_hist)
{
    uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t xcc = xcc_id();
    if (threadIdx.x == 0) atomicAdd(xcc_hist + (xcc & 15), 1u);
    float *base_ptr = SCOPE == 0 ? g : g + (size_t) xcc * copy_stride;
    uint32_t s = hash(tid * 9781u + 12345u);
    int sy = res, sz = res * res;
    for (int it = 0; it < iters; ++it) {
        s = hash(s + it);
        uint32_t b = s;
        if (COHERENT) { b = hash((tid >> 6) * 7919u + it); s = hash(s); }
        int x = b % (res - 1), y = (b / res) % (res - 1), z = (b / (res * res)) % (res - 1);
        if (COHERENT) { x = min(res - 2, x + (int)(s & 3)); y = min(res - 2, y + (int)((s >> 2) & 1)); }
        int i0 = z * sz + y * sy + x;
        const int offs[8] = { 0, 1, sy, sy + 1, sz, sz + 1, sz + sy, sz + sy + 1 };
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            float *p = base_ptr + i0 + offs[c];
            if (SCOPE == 0) atomicAdd(p, 1.0f);
            else if (SCOPE == 1) __hip_atomic_fetch_add(p, 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            else __hip_atomic_fetch_add(p, 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
        }
    }
}

__global__ void reduce_copies(const float *g, size_t copy_stride, size_t n, int copies, double *total)
{
    double acc = 0;
    for (size_t i = blockIdx.x * (size_t) blockDim.x + threadIdx.x; i < n; i += (size_t) gridDim.x * blockDim.x)
        for (int c = 0; c < copies; ++c) acc += g[c * copy_stride + i];
    atomicAdd(total, acc);
}

template <int SCOPE, bool COHERENT>
void run(const char *name, int res, int blocks, int iters)
{
    size_t n = (size_t) res * res * res;
    int copies = SCOPE == 0 ? 1 : 8;
    float *g; double *total; uint32_t *hist;
    hipMalloc(&g, n * 4 * copies); hipMemset(g, 0, n * 4 * copies);
    hipMalloc(&total, 8); hipMemset(total, 0, 8); hipMalloc(&hist, 64); hipMemset(hist, 0, 64);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipDeviceSynchronize();
    hipEventRecord(a);
    hipLaunchKernelGGL((k<SCOPE, COHERENT>), dim3(blocks), dim3(256), 0, 0, g, n, res, iters, hist);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    hipLaunchKernelGGL(reduce_copies, dim3(1024), dim3(256), 0, 0, g, n, n, copies, total);
    double t; hipMemcpy(&t, total, 8, hipMemcpyDeviceToHost);
    uint32_t h[16]; hipMemcpy(h, hist, 64, hipMemcpyDeviceToHost);
    double expect = (double) blocks * 256 * iters * 8;
    printf("%-40s res %3d %8.3f ms %7.1f G/s  sum %.0f expect %.0f %s  xcc hist:", name, res, ms, expect / ms / 1e6, t, expect, t == expect ? "OK" : "LOST");
    for (int i = 0; i < 9; ++i) printf(" %u", h[i]);
    printf("\n");
    hipFree(g); hipFree(total); hipFree(hist);
}

int main()
{
    for (int res : {64, 256}) {
        run<0, false>("agent scope random", res, 4096, 64);
        run<1, false>("workgroup scope XCD-private random", res, 4096, 64);
        run<2, false>("wavefront scope XCD-private random", res, 4096, 64);
        run<0, true>("agent scope coherent", res, 4096, 64);
        run<1, true>("workgroup scope XCD-private coherent", res, 4096, 64);
    }
    return 0;
}
