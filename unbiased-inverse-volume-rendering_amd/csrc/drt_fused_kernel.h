// drt_fused_kernel.h -- the kernel of the fused nerf + volpathsimple pass (see drt_fused.hip) and its launcher, templated on
// the emitter (ENV: environment map) and the majorants (SUPER: majorant supergrid): four translation units
// (drt_fused.hip, drt_fused_env.hip, drt_fused_super.hip, drt_fused_env_super.hip) so that the four sets of kernels
// compile side by side.  The reference's nerf scenes run with an environment map AND majorant_resolution_factor 8
// (python/scene_config.py:36,102-141).
#pragma once
#include "drt_coop_tracer.h"

namespace drt {

namespace {

using namespace coop;

template <bool ADJ, bool COUNT, bool SPEC, bool ENV, bool SUPER>
__global__ void __launch_bounds__(256, DRT_COOP_WAVES) fused_kernel(const Params P)
{
    uint32_t b = blockIdx.x;                                    // XCD-aware block -> ray-chunk map (see trace_kernel)
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
    const uint64_t i = P.ray_first + (uint64_t) b * blockDim.x + threadIdx.x;
    CoopTracer<COUNT, ENV, ADJ, SPEC, true, SUPER> tr(P);            // DEFER == ADJ: the fused adjoint always uses record streams
    __shared__ uint32_t slot_lds[4 * 64];
    tr.slots = slot_lds + (threadIdx.x >> 6) * 64;
    __shared__ uint64_t jump_lds[2 * (kJumpMax + 1)];
    if (threadIdx.x <= kJumpMax) { jump_lds[2 * threadIdx.x] = kJump.A[threadIdx.x]; jump_lds[2 * threadIdx.x + 1] = kJump.G[threadIdx.x]; }
    tr.jump = jump_lds;
    __syncthreads();
    if constexpr (ADJ) {
        __shared__ uint32_t rec_state[4 * 8];                   // per wave: cur[2], -, -, end[2]
        tr.rec = rec_state + (threadIdx.x >> 6) * 8;
        if ((threadIdx.x & 63) < 8) tr.rec[threadIdx.x & 63] = 0;
        coop_stage_sync();
    }
    __shared__ uint32_t occ_lds[kOccWords];
    if (P.occ) {
        for (int w = threadIdx.x; w < P.occ_words; w += blockDim.x) occ_lds[w] = P.occ[w];
        __syncthreads();
        tr.occ = occ_lds;
    }
    if constexpr (SUPER) {                                      // non-empty supergrid cells -> LDS (as trace_coop_kernel)
        __shared__ uint32_t mocc_lds[kOccWords];
        if (P.mocc && P.mocc_words <= kOccWords) {
            for (int w = threadIdx.x; w < P.mocc_words; w += blockDim.x) mocc_lds[w] = P.mocc[w];
            __syncthreads();
            tr.mocc = mocc_lds;
        }
    }
    const bool job = i < P.n_rays;
    Pcg32 S; S.state = 0; S.inc = 1;
    Ray ray; ray.o = v3(0, 0, 0); ray.d = v3(0, 0, 1); ray.maxt = kLargest;
    float dLd[3] = { 0, 0, 0 }, Lind[3] = { 0, 0, 0 };
    uint32_t n_q = 0;
    if (job) {
        const uint64_t g64 = P.chunk ? P.ray_offset + (i / P.chunk) * P.stride + (i % P.chunk) : P.ray_offset + i;
        const uint32_t gi = (uint32_t) g64;
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

        // ---- 1. NeRFIntegrator.sample (nerf.py:47-148) on its own copy of the stream: the same draws the
        //         stand-alone integrator makes (the volpathsimple half below continues from S, as it would alone)
        {
            Pcg32 N = S;
            V3 o = ray.o; const V3 d = ray.d;
            float result[3] = { 0.0f, 0.0f, 0.0f }, dL[3] = { 0.0f, 0.0f, 0.0f };
            if constexpr (ADJ) {
                result[0] = P.L_in2[3 * i]; result[1] = P.L_in2[3 * i + 1]; result[2] = P.L_in2[3 * i + 2];
                dL[0] = P.dL2[3 * i]; dL[1] = P.dL2[3 * i + 1]; dL[2] = P.dL2[3 * i + 2];
            }
            float throughput = 1.0f, weights_sum = 0.0f;
            Hit si = box_hit(P, o, d);                                           // nerf.py:67-79
            bool active = si.valid; const bool escaped = !active;
            if (active) {
                o = offset_p(si, d);
                si = box_hit(P, o, d);
                active = si.valid;
            }
            if (active) {
                const int N_q = P.nerf_queries;
                const float step = P.nerf_jitter ? (si.t - 0.0f) / (float) N_q : (si.t - 0.0f) / (float) (N_q - 1);   // :6-10,82
                float t_a = 0.0f;
                const float jit = N.next_1d();                                   // :88
                for (int j = 0; j < N_q; ++j) {                                  // :94-129
                    const float t_b = P.nerf_jitter ? step * ((float) (j + 1) + jit) : step * (float) (j + 1);
                    const float dt = t_b - t_a;
                    const V3 p = ray_at(o, d, t_b);                              // query_medium :151-165
                    float raw, em[3] = { 0.0f, 0.0f, 0.0f };
                    // the primal needs the emission only where the query has weight, i.e. where sigma_t != 0: the
                    // empty-space mask answers most queries of a sparse volume without a fetch; the adjoint's
                    // sigma_t gradient needs the emission everywhere
                    if (!ADJ && tr.occ && occ_empty(P, p, tr.occ)) raw = 0.0f;
                    else eval4(P, p, raw, em);
                    const float sigma = P.nerf_relu ? fmaxf(0.0f, raw) : raw;
                    n_q++;
                    const bool last = !(j + 1 < N_q);
                    const float a = last ? 1.0f : drt_expf(-sigma * dt);         // :104-106
                    const float weight = (1.0f - a) * throughput;
                    const float safe_a = a + 1e-10f;
                    if (!ADJ && weight == 0.0f) { em[0] = em[1] = em[2] = 0.0f; }   // (as the stand-alone kernel: no lookup, no term)
#pragma unroll
                    for (int k = 0; k < 3; ++k) result[k] = ADJ ? result[k] - weight * em[k] : result[k] + weight * em[k];
                    if constexpr (ADJ) {                                         // :122-129
                        float gs = 0.0f, ge[3];
                        const float da = last ? 0.0f : -dt * a;
#pragma unroll
                        for (int k = 0; k < 3; ++k) {
                            gs += dL[k] * (em[k] * (-da * throughput) + (result[k] / safe_a) * da);
                            ge[k] = dL[k] * weight;
                        }
                        if (P.nerf_relu && !(raw > 0.0f)) gs = 0.0f;
                        splat_scatter<true>(P, p, gs, ge, tr.rec);
                    }
                    t_a = t_b;
                    if (!last) { throughput *= safe_a; weights_sum += weight; }  // :117-120
                }
            }
            bool active_e = escaped || active;                                   // :131-146
            if (P.hide_emitters_nerf) active_e = active_e && (weights_sum > 0.0f);
            if (active_e) {
                float Le[3];
                emitter_eval<ENV>(P, d, Le);                                   // (the scene's emitter: constant or environment map)
#pragma unroll
                for (int k = 0; k < 3; ++k) result[k] += (1.0f - weights_sum) * Le[k];
            }
            if constexpr (!ADJ) { P.L_out2[3 * i] = result[0]; P.L_out2[3 * i + 1] = result[1]; P.L_out2[3 * i + 2] = result[2]; }
        }

        // ---- 2. set-up of the volpathsimple half (as trace_coop_kernel)
        if (P.path_cache_mode) {
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
            dLd[0] = P.dL[3 * i]; dLd[1] = P.dL[3 * i + 1]; dLd[2] = P.dL[3 * i + 2];
            Lind[0] = P.L_in[3 * i]; Lind[1] = P.L_in[3 * i + 1]; Lind[2] = P.L_in[3 * i + 2];
        }
    }
    float L[3];
    if (ADJ) tr.template sample<true, false>(job, S, ray, dLd, Lind, nullptr, L);
    else {
        tr.template sample<false, false>(job, S, ray, nullptr, nullptr, nullptr, L);
        if (job) { P.L_out[3 * i] = L[0]; P.L_out[3 * i + 1] = L[1]; P.L_out[3 * i + 2] = L[2]; }
    }
    if constexpr (ADJ) close_records(P, tr.rec);
    if (COUNT) {
        // the nerf half counts as the stand-alone kernel does: one sigma_t + one colour lookup per query, and in the
        // adjoint one sigma_t + one colour splat per query
        tr.cnt[C_DT] += n_q; tr.cnt[C_ALB] += n_q;
        if (ADJ) { tr.cnt[C_SC] += n_q; tr.cnt[C_SC_ALB] += n_q; }
#pragma unroll
        for (int s = 0; s < C_COUNT; ++s) {
            uint32_t v = tr.cnt[s];
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
            if ((threadIdx.x & 63) == 0 && v) atomicAdd(P.counters + s, (unsigned long long) v);
        }
    }
}


template <bool ENV, bool SUPER>
hipError_t launch_fused_t(const Params &P, bool adjoint, bool count, hipStream_t stream)
{
    if (P.n_rays <= P.ray_first) return hipSuccess;
    dim3 block(256), grid((unsigned) ((P.n_rays - P.ray_first + 255) / 256));
    const bool spec = P.use_nee && P.use_drt && P.use_drt_subsampling && !count && !dbg(P.debug_flags, 2097152u);
#define DRT_FUSED_LAUNCH(A, C, S) hipLaunchKernelGGL((fused_kernel<A, C, S, ENV, SUPER>), grid, block, 0, stream, P)
    if (!adjoint) {
        if (count) DRT_FUSED_LAUNCH(false, true, false); else if (spec) DRT_FUSED_LAUNCH(false, false, true); else DRT_FUSED_LAUNCH(false, false, false);
    } else {
        if (count) DRT_FUSED_LAUNCH(true, true, false); else if (spec) DRT_FUSED_LAUNCH(true, false, true); else DRT_FUSED_LAUNCH(true, false, false);
    }
#undef DRT_FUSED_LAUNCH
    return hipGetLastError();
}

}  // namespace

}  // namespace drt
