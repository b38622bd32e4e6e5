// drt_coop_kernel.h -- the kernel of the one-ray-per-lane tracer (CoopTracer, drt_coop_tracer.h) and its launcher,
// templated on SUPER (majorant supergrid): instantiated by drt_coop.hip (global majorant: wave-cooperative tracking
// rounds) and drt_coop_super.hip (supergrid: every tracking step on the walk's own lane) - two translation units so
// that the two sets of kernels compile side by side.
#pragma once
#include "drt_coop_tracer.h"

namespace drt {

namespace {

using namespace coop;

// TAIL: the second launch of a specialised kernel - its workgroups start from 256 paths of the tail pool each
// (CoopTracer::wg_handoff) instead of from rays, and finish them
template <bool ADJ, bool COUNT, bool ENV, bool DEFER, bool SPEC = false, bool SUPER = false, bool TAIL = false>
__global__ void __launch_bounds__(256, ADJ ? DRT_COOP_WAVES : DRT_COOP_WAVES_PRIMAL) trace_coop_kernel(const Params P)
{
    if constexpr (TAIL) { if (blockIdx.x * 256u >= *P.tail_count) return; }   // (workgroup-uniform) nothing for this workgroup
    uint32_t b = blockIdx.x;                                    // XCD-aware block -> ray-chunk map (see trace_kernel)
    if (P.block_order) b = P.block_order[blockIdx.x];            // heavy blocks first (adjoint: this job's primal costs; primal: the previous launch's)
    else
#if DRT_XCD_RUN > 0
    {
        const uint32_t span = 8u * DRT_XCD_RUN;
        const uint32_t full = (gridDim.x / span) * span;
        if (b < full) {
            uint32_t grp = b / span, r = b % span;
            b = grp * span + (r % 8u) * DRT_XCD_RUN + r / 8u;
        }
    }
#endif
    const uint64_t i_block = P.ray_first + (uint64_t) b * blockDim.x;
    uint64_t i = i_block + threadIdx.x;
    if constexpr (ADJ && !SUPER) {                              // rays of similar length share a wave (ray_perm_kernel)
        if (P.ray_perm) i = (i_block & ~(uint64_t) (kPermGroup - 1)) + P.ray_perm[i_block + threadIdx.x];
    }
    CoopTracer<COUNT, ENV, DEFER, SPEC, false, SUPER> tr(P);
    __shared__ uint32_t slot_lds[4 * 64];
    tr.slots = slot_lds + (threadIdx.x >> 6) * 64;
    tr.i_block = TAIL ? 0 : i_block;
    tr.tail_load = TAIL;
    if constexpr (SPEC) {                                       // sparse waves hand their last (adjoint: recursive, primal: main) paths to wave 0 (wg_handoff)
        __shared__ uint32_t wgc_lds[kWgcWords];
        static_assert(DRT_COOP_WAVES >= 1, "");
        if (threadIdx.x < 4) wgc_lds[threadIdx.x] = 0xffffffffu; // nothing published yet (made visible by the barrier below)
        if (!dbg(P.debug_flags, 33554432u) && !(!ADJ && dbg(P.debug_flags, 67108864u)) && P.max_depth < 32768) tr.wgc = wgc_lds;   // (the path-cache cursor travels in 16 bits)
    }
    __shared__ uint64_t jump_lds[2 * (kJumpMax + 1)];
    if (threadIdx.x <= kJumpMax) { jump_lds[2 * threadIdx.x] = kJump.A[threadIdx.x]; jump_lds[2 * threadIdx.x + 1] = kJump.G[threadIdx.x]; }
    tr.jump = jump_lds;
    __syncthreads();
    if constexpr (ADJ && DEFER) {
        __shared__ uint32_t rec_state[4 * 8];                   // per wave: cur[4], end[4]
        tr.rec = rec_state + (threadIdx.x >> 6) * 8;
        if ((threadIdx.x & 63) < 8) tr.rec[threadIdx.x & 63] = 0;
        coop_stage_sync();
    } else if constexpr (ADJ) {
        __shared__ uint32_t coop_rec[4 * 64 * kCoopDwords];
        tr.rec = coop_rec + (threadIdx.x >> 6) * (64 * kCoopDwords);
    }
    __shared__ uint32_t occ_lds[kOccWords];
    if (P.occ && !dbg(P.debug_flags, 16u)) {
        for (int w = threadIdx.x; w < P.occ_words; w += blockDim.x) occ_lds[w] = P.occ[w];
        __syncthreads();
        tr.occ = occ_lds;
    }
    if constexpr (SUPER) {                                      // non-empty supergrid cells -> LDS (dda_collision skips the others)
        __shared__ uint32_t mocc_lds[kOccWords];
        if (P.mocc && P.mocc_words <= kOccWords && !dbg(P.debug_flags, 8388608u)) {
            for (int w = threadIdx.x; w < P.mocc_words; w += blockDim.x) mocc_lds[w] = P.mocc[w];
            __syncthreads();
            tr.mocc = mocc_lds;
        }
    }
    const bool job = !TAIL && i < P.n_rays;
    Pcg32 S; S.state = 0; S.inc = 1;
    Ray ray; ray.o = v3(0, 0, 0); ray.d = v3(0, 0, 1); ray.maxt = kLargest;
    float dL[3] = { 0, 0, 0 }, Lin[3] = { 0, 0, 0 };
    if (job) {
        uint64_t g64 = P.chunk ? P.ray_offset + (i / P.chunk) * P.stride + (i % P.chunk) : P.ray_offset + i;
        uint32_t gi = (uint32_t) g64;
        tr.ray_index = gi;
        S.seed(P.seed, gi);
        if (P.sensor_flow) {
            float ux = S.next_1d(), uy = S.next_1d();
            sensor_ray(P, gi / P.spp, ux, uy, ray.o, ray.d);
        } else {
            ray.o = v3(P.rays_o[3 * i], P.rays_o[3 * i + 1], P.rays_o[3 * i + 2]);
            ray.d = v3(P.rays_d[3 * i], P.rays_d[3 * i + 1], P.rays_d[3 * i + 2]);
        }
        tr.count(C_RAYS);
        if (P.path_cache_mode) {
            // one word per ray ties the entries to THIS ray: explicit rays are hashed (the buffers may have
            // been refilled between the two passes), sensor rays are determined by the job signature
            uint32_t hsh = 0x9e3779b9u ^ gi;
            if (!P.sensor_flow) {
                const uint32_t w[6] = { __float_as_uint(ray.o.x), __float_as_uint(ray.o.y), __float_as_uint(ray.o.z),
                                        __float_as_uint(ray.d.x), __float_as_uint(ray.d.y), __float_as_uint(ray.d.z) };
#pragma unroll
                for (int k = 0; k < 6; ++k) hsh = (hsh ^ w[k]) * 0x01000193u + (hsh >> 15);
            }
            if (P.path_cache_mode == 1) { P.ray_hash[i] = hsh; tr.pc = P.path_cache + (size_t) i * P.path_cache_cap * 2; }
            else if (P.ray_hash[i] == hsh) tr.pc = P.path_cache + (size_t) i * P.path_cache_cap * 2;
        }
        if (ADJ) {
            dL[0] = P.dL[3 * i]; dL[1] = P.dL[3 * i + 1]; dL[2] = P.dL[3 * i + 2];
            Lin[0] = P.L_in[3 * i]; Lin[1] = P.L_in[3 * i + 1]; Lin[2] = P.L_in[3 * i + 2];
        }
    }
    float L[3];
    if constexpr (TAIL && ADJ) {                                // recursive DRT paths of the pool, their gradient splats included
        Tail tl;
        tl.p = v3(0, 0, 0); tl.sig = 0.0f;
#pragma unroll
        for (int q = 0; q < 3; ++q) { tl.alb[q] = 0.0f; tl.wadj[q] = 0.0f; tl.nee[q] = 0.0f; }
        PathState ps;
        ps.depth = 0; ps.last_pdf = kInvFourPi; ps.escaped = false; ps.active = false; ps.tail = &tl;
        ps.si.valid = false; ps.si.t = kInf; ps.si.p = v3(0, 0, 0); ps.si.n = v3(0, 0, 0);
        tr.template sample<false, true>(false, S, ray, nullptr, nullptr, &ps, L);
    } else if (ADJ) tr.template sample<true, false>(job, S, ray, dL, Lin, nullptr, L);
    else {
        tr.template sample<false, false>(job, S, ray, nullptr, nullptr, nullptr, L);
        if (job && !(SPEC && tr.wgc)) { P.L_out[3 * i] = L[0]; P.L_out[3 * i + 1] = L[1]; P.L_out[3 * i + 2] = L[2]; }   // (hand-off: written by wg_handoff)
    }
    if constexpr (ADJ && DEFER) close_records(P, tr.rec);
    if constexpr (!ADJ) {
        if (P.ray_iters && job && !(SPEC && tr.wgc)) P.ray_iters[i] = (uint8_t) (tr.iters < 255u ? tr.iters : 255u);   // sort key of ray_perm_kernel
        if (P.block_cost && !TAIL) {
            uint32_t v = tr.work;
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
            if ((threadIdx.x & 63) == 0 && v) atomicAdd(P.block_cost + b, v);
        }
    }
    if (COUNT) {
#pragma unroll
        for (int s = 0; s < C_COUNT; ++s) {
            uint32_t v = tr.cnt[s];
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
            if ((threadIdx.x & 63) == 0 && v) atomicAdd(P.counters + s, (unsigned long long) v);
        }
    }
}


// launch of the instantiation that fits the job
template <bool SUPER>
hipError_t launch_trace_coop_t(const Params &P, bool adjoint, bool count, hipStream_t stream, coop_between_fn between = nullptr,
                               void *between_ctx = nullptr, bool *called = nullptr)
{
    if (P.n_rays <= P.ray_first) return hipSuccess;
    dim3 block(256), grid((unsigned)((P.n_rays - P.ray_first + 255) / 256));
    const bool env = P.env_pix != nullptr, defer = adjoint && P.rec_buf[0] != nullptr;
#define DRT_COOP_LAUNCH(A, C, E, D) hipLaunchKernelGGL((trace_coop_kernel<A, C, E, D, false, SUPER>), grid, block, 0, stream, P)
    // the registered `volpathsimple-drt` configuration (either emitter): specialised kernels
    const bool spec = P.use_nee && P.use_drt && P.use_drt_subsampling && !count && !dbg(P.debug_flags, 2097152u);
#if DRT_PHASE_PROFILE
    // experiment build: the counting launches run the specialised kernels too (their counters then hold phase cycles)
    if (P.use_nee && P.use_drt && P.use_drt_subsampling && count && !env) {
        if (!adjoint) { hipLaunchKernelGGL((trace_coop_kernel<false, true, false, false, true, SUPER>), grid, block, 0, stream, P); return hipGetLastError(); }
        if (defer) { hipLaunchKernelGGL((trace_coop_kernel<true, true, false, true, true, SUPER>), grid, block, 0, stream, P); return hipGetLastError(); }
    }
#endif
    // ... each followed by its tail launch (the workgroups' last few paths, pooled: wg_handoff) when the caller gave a pool
    // (adjoint only: a tail launch exposes the longest path of the job - ~0.5 ms - which the primal pass has nothing to
    // hide behind: primal 2.58 -> 2.85 ms with it, adjoint tracer 6.19 -> 5.99 ms)
    // (capacity invariant of the pool's reservation, CoopTracer::wg_handoff: one push of <= DRT_TAIL_PUSH entries per workgroup)
    const bool tail = !SUPER && adjoint && P.tail_pool && P.tail_count && P.tail_cap >= 256u && !dbg(P.debug_flags, 33554432u) &&
                      (uint64_t) P.tail_cap >= (uint64_t) DRT_TAIL_PUSH * grid.x;
    Params T = P;
    if (spec && (!adjoint || defer)) {
        if (tail) {
            hipError_t e = hipMemsetAsync(P.tail_count, 0, sizeof(uint32_t), stream);
            if (e != hipSuccess) return e;
            T.tail_mode = 1; T.block_order = nullptr; T.ray_perm = nullptr;
        } else T.tail_pool = nullptr;
    }
    Params M = P;
    if (!tail) M.tail_pool = nullptr;
    M.tail_mode = 0;
    const dim3 tgrid(tail ? P.tail_cap / 256u : 1u);
    // (pool capacity = 1/8 of the launch's rays; the tail kernel's surplus workgroups return at once)
    if (spec && !adjoint) {
        if (env) hipLaunchKernelGGL((trace_coop_kernel<false, false, true, false, true, SUPER>), grid, block, 0, stream, M);
        else hipLaunchKernelGGL((trace_coop_kernel<false, false, false, false, true, SUPER>), grid, block, 0, stream, M);
        return hipGetLastError();
    }
    if (spec && defer) {
        if (env) hipLaunchKernelGGL((trace_coop_kernel<true, false, true, true, true, SUPER>), grid, block, 0, stream, M);
        else hipLaunchKernelGGL((trace_coop_kernel<true, false, false, true, true, SUPER>), grid, block, 0, stream, M);
        if constexpr (!SUPER) if (tail) {
            if (between) {                                     // (e.g. the early histogram pass of the record streams)
                hipError_t e = between(between_ctx);
                if (e != hipSuccess) return e;
                if (called) *called = true;
            }
            if (env) hipLaunchKernelGGL((trace_coop_kernel<true, false, true, true, true, false, true>), tgrid, block, 0, stream, T);
            else hipLaunchKernelGGL((trace_coop_kernel<true, false, false, true, true, false, true>), tgrid, block, 0, stream, T);
        }
        return hipGetLastError();
    }
    if (!adjoint) {
        if (count) { if (env) DRT_COOP_LAUNCH(false, true, true, false); else DRT_COOP_LAUNCH(false, true, false, false); }
        else       { if (env) DRT_COOP_LAUNCH(false, false, true, false); else DRT_COOP_LAUNCH(false, false, false, false); }
    } else if (defer) {
        if (count) { if (env) DRT_COOP_LAUNCH(true, true, true, true); else DRT_COOP_LAUNCH(true, true, false, true); }
        else       { if (env) DRT_COOP_LAUNCH(true, false, true, true); else DRT_COOP_LAUNCH(true, false, false, true); }
    } else {
        if (count) { if (env) DRT_COOP_LAUNCH(true, true, true, false); else DRT_COOP_LAUNCH(true, true, false, false); }
        else       { if (env) DRT_COOP_LAUNCH(true, false, true, false); else DRT_COOP_LAUNCH(true, false, false, false); }
    }
#undef DRT_COOP_LAUNCH
    return hipGetLastError();
}

}  // namespace

}  // namespace drt
