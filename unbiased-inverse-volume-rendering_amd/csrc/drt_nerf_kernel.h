// drt_nerf_kernel.h -- NeRFIntegrator.sample as one ray per lane (the primal pass; the adjoint of explicit ray batches and of the record /
// atomic gradient paths), shared by drt_kernels.hip and drt_own.hip (the same kernel with the colour grids on their own lattice,
// DRT_COLOUR_OWN).  Internal linkage: every including unit gets its own instantiations.
#pragma once
#include "drt_device.h"
#include "drt_launch.h"

namespace drt {
namespace {

// ---------------------------------------------------------------------------
// NeRFIntegrator.sample (python/integrators/nerf.py:47-148): emission-absorption ray marching,
// queries_per_ray jittered queries per ray, PRB-style backward.  One ray per lane; the loop is
// regular (no divergence besides rays that miss the box).
// ---------------------------------------------------------------------------
template <bool ADJ, bool COUNT, bool DEFER>
__global__ void __launch_bounds__(256) nerf_kernel(const Params P)
{
    uint64_t i = P.ray_first + (uint64_t) blockIdx.x * blockDim.x + threadIdx.x;
    __shared__ uint32_t occ_lds[kOccWords];
    const uint32_t *occ = nullptr;
    if (P.occ) {
        for (int w = threadIdx.x; w < P.occ_words; w += blockDim.x) occ_lds[w] = P.occ[w];
        __syncthreads();
        occ = occ_lds;
    }
    uint32_t *rec = nullptr;
    if constexpr (ADJ && DEFER) {
        __shared__ uint32_t rec_state[4 * 8];                   // per wave: cur[4], end[4]
        rec = rec_state + (threadIdx.x >> 6) * 8;
        if ((threadIdx.x & 63) < 8) rec[threadIdx.x & 63] = 0;
        coop_stage_sync();
    } else if constexpr (ADJ) {
        __shared__ uint32_t coop_rec[4 * 64 * kCoopDwords];
        rec = coop_rec + (threadIdx.x >> 6) * (64 * kCoopDwords);
    }
    uint32_t n_q = 0, n_rays = 0;
    if (i < P.n_rays) {
        uint64_t g64 = P.chunk ? P.ray_offset + (i / P.chunk) * P.stride + (i % P.chunk) : P.ray_offset + i;
        uint32_t gi = (uint32_t) g64;
        Pcg32 S; S.seed(P.seed, gi);
        V3 o, d;
        if (P.sensor_flow) {
            float ux = S.next_1d(), uy = S.next_1d();
            sensor_ray(P, gi / P.spp, ux, uy, o, d);
        } else {
            o = v3(P.rays_o[3 * i], P.rays_o[3 * i + 1], P.rays_o[3 * i + 2]);
            d = v3(P.rays_d[3 * i], P.rays_d[3 * i + 1], P.rays_d[3 * i + 2]);
        }
        n_rays = 1;
        float result[3] = { 0.0f, 0.0f, 0.0f }, dL[3] = { 0.0f, 0.0f, 0.0f };
        if constexpr (ADJ) {
            result[0] = P.L_in[3 * i]; result[1] = P.L_in[3 * i + 1]; result[2] = P.L_in[3 * i + 2];
            dL[0] = P.dL[3 * i]; dL[1] = P.dL[3 * i + 1]; dL[2] = P.dL[3 * i + 2];
        }
        float throughput = 1.0f, weights_sum = 0.0f;
        Hit si = box_hit(P, o, d);                                           // nerf.py:67-79
        bool active = si.valid, escaped = !active;
        if (active) {
            o = offset_p(si, d);
            si = box_hit(P, o, d);
            active = si.valid;
        }
        if (active) {
            const int N = P.nerf_queries;
            float step = P.nerf_jitter ? (si.t - 0.0f) / (float) N : (si.t - 0.0f) / (float)(N - 1);   // :6-10,82
            float t_a = 0.0f;
            float jit = S.next_1d();                                         // :88
            for (int j = 0; j < N; ++j) {                                    // :94-129
                float t_b = P.nerf_jitter ? step * ((float)(j + 1) + jit) : step * (float)(j + 1);
                float dt = t_b - t_a;
                V3 p = ray_at(o, d, t_b);                                    // query_medium :151-165
                if constexpr (!ADJ) {
                    // a query in empty space (every voxel its lookup can touch is exactly 0: the occupancy mask) changes nothing in the primal:
                    // sigma = 0, a = exp(-0) = 1 (or the last query's 1), weight = 0 x throughput = 0, 1 + 1e-10 == 1 in fp32 - most queries of a sparse
                    // volume end here, without the exponential and the bookkeeping of exact zeros
                    if (occ && occ_empty(P, p, occ)) { n_q++; t_a = t_b; continue; }
                }
                float raw = eval_sigma_t(P, p, occ);
                float sigma = P.nerf_relu ? fmaxf(0.0f, raw) : raw;
                n_q++;
                bool last = !(j + 1 < N);
                float a = last ? 1.0f : drt_expf(-sigma * dt);               // :104-106
                float weight = (1.0f - a) * throughput;
                float safe_a = a + 1e-10f;
                // the primal only needs the emission where the query has weight (adding weight * em with
                // weight == 0 changes nothing); the adjoint's sigma_t gradient needs it everywhere
                float em[3] = { 0.0f, 0.0f, 0.0f };
                if (ADJ || weight != 0.0f) eval_rgb(P, P.emission, p, em);
#pragma unroll
                for (int k = 0; k < 3; ++k) result[k] = ADJ ? result[k] - weight * em[k] : result[k] + weight * em[k];
                if constexpr (ADJ) {                                         // :122-129
                    float gs = 0.0f, ge[3];
                    float da = last ? 0.0f : -dt * a;
#pragma unroll
                    for (int k = 0; k < 3; ++k) {
                        gs += dL[k] * (em[k] * (-da * throughput) + (result[k] / safe_a) * da);
                        ge[k] = dL[k] * weight;
                    }
                    if (P.nerf_relu && !(raw > 0.0f)) gs = 0.0f;
                    splat_scatter<DEFER>(P, p, gs, ge, rec);   // colour planes = emission gradients here
                }
                t_a = t_b;
                if (!last) { throughput *= safe_a; weights_sum += weight; }  // :117-120
            }
        }
        bool active_e = escaped || active;                                   // :131-146
        if (P.hide_emitters) active_e = active_e && (weights_sum > 0.0f);
        if (active_e) {
            float Le[3];
            if (P.env_pix) emitter_eval<true>(P, d, Le); else emitter_eval<false>(P, d, Le);
#pragma unroll
            for (int k = 0; k < 3; ++k) result[k] += (1.0f - weights_sum) * Le[k];
        }
        if constexpr (!ADJ) { P.L_out[3 * i] = result[0]; P.L_out[3 * i + 1] = result[1]; P.L_out[3 * i + 2] = result[2]; }
    }
    if constexpr (ADJ && DEFER) close_records(P, rec);
    if (COUNT) {
        uint32_t vals[C_COUNT] = { P.nerf_fused_half ? 0u : n_rays, n_q, 0, 0, n_q, 0, 0, ADJ ? n_q : 0u, ADJ ? n_q : 0u };
#pragma unroll
        for (int s = 0; s < C_COUNT; ++s) {
            uint32_t v = vals[s];
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
            if ((threadIdx.x & 63) == 0 && v) atomicAdd(P.counters + s, (unsigned long long) v);
        }
    }
}

hipError_t launch_nerf_t(const Params &P, bool adjoint, bool count, hipStream_t stream)
{
    if (P.n_rays <= P.ray_first) return hipSuccess;
    dim3 block(256), grid((unsigned)((P.n_rays - P.ray_first + 255) / 256));
    const bool defer = adjoint && P.rec_buf[0] != nullptr;
    if (adjoint && defer) {
        if (count) hipLaunchKernelGGL((nerf_kernel<true, true, true>), grid, block, 0, stream, P);
        else       hipLaunchKernelGGL((nerf_kernel<true, false, true>), grid, block, 0, stream, P);
    } else if (adjoint) {
        if (count) hipLaunchKernelGGL((nerf_kernel<true, true, false>), grid, block, 0, stream, P);
        else       hipLaunchKernelGGL((nerf_kernel<true, false, false>), grid, block, 0, stream, P);
    } else {
        if (count) hipLaunchKernelGGL((nerf_kernel<false, true, false>), grid, block, 0, stream, P);
        else       hipLaunchKernelGGL((nerf_kernel<false, false, false>), grid, block, 0, stream, P);
    }
    return hipGetLastError();
}

}  // namespace
}  // namespace drt
