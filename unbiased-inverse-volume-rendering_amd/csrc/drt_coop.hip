// drt_coop.hip -- kernels of the one-ray-per-lane tracer with wave-cooperative tracking loops (CoopTracer,
// drt_coop_tracer.h): VolpathSimpleIntegrator.sample in both AD modes (volpathsimple.py:38-655).
#include "drt_coop_kernel.h"

namespace drt {

namespace {
// blocks by descending cost: counting sort on floor(log2(cost + 1)), one workgroup
__global__ void __launch_bounds__(1024) block_order_kernel(const uint32_t *cost, uint32_t n, uint32_t *order)
{
    __shared__ uint32_t hist[33], cur[33];
    if (threadIdx.x < 33) hist[threadIdx.x] = 0;
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < n; i += blockDim.x) atomicAdd(&hist[32 - __clz(cost[i])], 1u);   // bucket 0..32
    __syncthreads();
    if (threadIdx.x == 0) { uint32_t run = 0; for (int k = 32; k >= 0; --k) { cur[k] = run; run += hist[k]; } }
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < n; i += blockDim.x) order[atomicAdd(&cur[32 - __clz(cost[i])], 1u)] = i;
}

// large launches: keep the XCD-contiguous block map (L2 locality is worth more than the order) but move
// the LIGHT blocks - the cheapest buckets, up to ~1/4 of the launch - to the end of the dispatch order, so
// that the last workgroups to start are short ones.  Stable partition of the XCD-mapped sequence.
__global__ void __launch_bounds__(1024) block_light_last_kernel(const uint32_t *cost, uint32_t n, uint32_t *order, uint32_t xcd_run)
{
    __shared__ uint32_t hist[33], scan[1024], thr, n_heavy;
    if (threadIdx.x < 33) hist[threadIdx.x] = 0;
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < n; i += blockDim.x) atomicAdd(&hist[32 - __clz(cost[i])], 1u);
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t run = 0, t = 0;
        for (int k = 0; k <= 32; ++k) { if (run + hist[k] > n / 4) break; run += hist[k]; t = (uint32_t) k + 1; }
        thr = t;                                               // buckets < thr are light
        n_heavy = n - run;
    }
    __syncthreads();
    const uint32_t T = thr, per = (n + blockDim.x - 1) / blockDim.x, j0 = threadIdx.x * per;
    auto logical = [&](uint32_t j) {
        const uint32_t span = 8u * xcd_run, full = xcd_run ? (n / span) * span : 0u;
        if (j < full) { const uint32_t grp = j / span, r = j % span; return grp * span + (r % 8u) * xcd_run + r / 8u; }
        return j;
    };
    uint32_t heavy = 0;
    for (uint32_t j = j0; j < j0 + per && j < n; ++j) heavy += (32u - (uint32_t) __clz(cost[logical(j)])) >= T ? 1u : 0u;
    scan[threadIdx.x] = heavy;
    __syncthreads();
    for (uint32_t off = 1; off < blockDim.x; off <<= 1) {
        const uint32_t v = threadIdx.x >= off ? scan[threadIdx.x - off] : 0u;
        __syncthreads();
        scan[threadIdx.x] += v;
        __syncthreads();
    }
    uint32_t h = scan[threadIdx.x] - heavy;                      // heavy blocks before my range
    uint32_t l = n_heavy + (j0 < n ? j0 : n) - h;                // light ones go behind all heavy ones
    for (uint32_t j = j0; j < j0 + per && j < n; ++j) {
        const uint32_t b = logical(j);
        if ((32u - (uint32_t) __clz(cost[b])) >= T) order[h++] = b; else order[l++] = b;
    }
}
}  // namespace

namespace {
// Ray -> lane schedule of the adjoint pass: every group of kPermGroup consecutive rays (4 workgroups) is sorted by the
// number of bounce-loop iterations the primal pass counted (longest first), so that the 64 rays
// of a wave leave the adjoint's bounce loop together.  One workgroup per group; rays beyond n_rays sort last.
// It also completes the primal pass's block costs (the adjoint's dispatch order, block_light_last_kernel): a wave of the
// adjoint runs its bounce loop as long as its longest ray, so every wave adds 4 units per lane and iteration of the
// longest of its 64 rays - counted HERE from the per-ray iteration counts and not by the primal tracer, whose waves no
// longer hold the rays they started with (CoopTracer::wg_handoff).
__global__ void __launch_bounds__(kPermGroup) ray_perm_kernel(const uint8_t *iters, uint64_t n_rays, uint16_t *perm, uint32_t *block_cost)
{
    __shared__ uint32_t hist[64];
    const uint64_t i = (uint64_t) blockIdx.x * kPermGroup + threadIdx.x;
    const uint32_t it = i < n_rays ? iters[i] : 0u;
    if (block_cost) {
        uint32_t mx = it;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) mx = max(mx, (uint32_t) __shfl_down((int) mx, off, 64));
        if ((threadIdx.x & 63) == 0 && mx) atomicAdd(block_cost + i / 256, 256u * mx);
    }
    const uint32_t key = i < n_rays ? (it < 63u ? 63u - it : 0u) : 63u;      // bucket 0 = longest
    if (threadIdx.x < 64) hist[threadIdx.x] = 0;
    __syncthreads();
    atomicAdd(&hist[key], 1u);
    __syncthreads();
    if (threadIdx.x < 64) {                                      // exclusive scan of the 64 buckets by one wave
        const uint32_t c = hist[threadIdx.x];
        uint32_t incl = c;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) { const uint32_t t = __shfl_up(incl, off, 64); if ((int) threadIdx.x >= off) incl += t; }
        hist[threadIdx.x] = incl - c;
    }
    __syncthreads();
    const uint32_t pos = atomicAdd(&hist[key], 1u);              // ties in arrival order: a schedule, not a result
    perm[(uint64_t) blockIdx.x * kPermGroup + pos] = (uint16_t) threadIdx.x;
}
}  // namespace

hipError_t launch_ray_perm(const uint8_t *iters, uint64_t n_rays, uint16_t *perm, uint32_t *block_cost, hipStream_t stream)
{
    if (n_rays == 0) return hipSuccess;
    hipLaunchKernelGGL(ray_perm_kernel, dim3((unsigned) ((n_rays + kPermGroup - 1) / kPermGroup)), dim3(kPermGroup), 0, stream, iters, n_rays, perm, block_cost);
    return hipGetLastError();
}

hipError_t launch_block_order(const uint32_t *cost, uint32_t n_blocks, uint32_t *order, bool heavy_first, hipStream_t stream)
{
    if (n_blocks == 0) return hipSuccess;
    if (heavy_first) hipLaunchKernelGGL(block_order_kernel, dim3(1), dim3(1024), 0, stream, cost, n_blocks, order);
    else hipLaunchKernelGGL(block_light_last_kernel, dim3(1), dim3(1024), 0, stream, cost, n_blocks, order, (uint32_t) DRT_XCD_RUN);
    return hipGetLastError();
}

hipError_t launch_trace_coop(const Params &P, bool adjoint, bool count, hipStream_t stream, coop_between_fn between, void *between_ctx,
                             bool *called)
{
    if (P.mgrid) return launch_trace_coop_super(P, adjoint, count, stream);      // drt_coop_super.hip
    return launch_trace_coop_t<false>(P, adjoint, count, stream, between, between_ctx, called);
}

}  // namespace drt
