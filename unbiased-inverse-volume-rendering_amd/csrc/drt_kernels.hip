// drt_kernels.hip -- gfx950 kernels around the tracer: the nerf integrator, box film, majorant / supergrid / empty-space
// reductions, brick layouts, batch ray generation, gradient untiling, the primitive-evaluation kernel of the parity
// tests - and, ONLY in the library flavour with test hooks (-DDRT_TEST_HOOKS), the plain one-ray-per-lane tracer `Tracer`
// (v1 of VolpathSimpleIntegrator.sample, python/integrators/volpathsimple.py:38-290): the variant tests keep it in
// lock-step with the production tracer (CoopTracer, drt_coop_tracer.h), the production library does not contain it.
#include "drt_device.h"
#include "drt_launch.h"
#include "drt_coop_tracer.h"
#include "drt_nerf_kernel.h"

#ifndef DRT_TRACE_WAVES
#define DRT_TRACE_WAVES 4      // waves per SIMD the tracing kernels are compiled for (VGPR <= 128)
#endif
#ifndef DRT_XCD_RUN
#define DRT_XCD_RUN 256        // workgroups per XCD-contiguous run (0 = identity block map); swept 16..4096 on MI355X
#endif
namespace drt {

struct Ray { V3 o, d; float maxt; };
struct Mei { bool valid; float t; V3 p; float sigma_t; };
struct PathState { int depth; Hit si; float last_pdf; bool escaped; bool active; };

#ifdef DRT_TEST_HOOKS
template <bool COUNT, bool ENV, bool DEFER = false>
struct Tracer {
    const Params &P;
    float maj, inv_maj;
    uint32_t ray_index;
    uint32_t *rec;          // wave-private LDS: staging area of the cooperative scatter, or (DEFER) the record-stream state
    const uint32_t *occ;    // empty-space bitmask (LDS copy) or nullptr
    const float *mg;        // majorant supergrid as the DDA reads it (global memory, L2-resident; an LDS copy
                            // was measured slower: it costs a wave per SIMD of occupancy)
    uint4 *pc;              // this ray's path-cache entries (drt_device.h: Params::path_cache) or nullptr
    const uint32_t *mocc;   // non-empty supergrid cells (LDS copy) or nullptr
    uint32_t cnt[C_COUNT];

    __device__ __forceinline__ Tracer(const Params &p) : P(p)
    {
        maj = p.majorant[0]; inv_maj = p.majorant[1];
        ray_index = 0; rec = nullptr; mg = p.mgrid; occ = nullptr; pc = nullptr; mocc = nullptr;
#pragma unroll
        for (int i = 0; i < C_COUNT; ++i) cnt[i] = 0;
    }
    __device__ __forceinline__ void count(int slot) { if (COUNT) cnt[slot]++; }

    // Medium::sample_interaction free-flight distance with a global majorant
    // (call sites volpathsimple.py:348,469)
    __device__ __forceinline__ float sample_distance(float u) const
    {
        if (maj == 0.0f) return kInf;
        return -drt_logf(1.0f - u) * inv_maj;
    }

    // Medium::sample_interaction with a majorant supergrid (dda_collision, drt_device.h); without a supergrid:
    // the global majorant (bit-identical to sample_distance).
    __device__ __forceinline__ float sample_collision(V3 o, V3 d, float tmax, float u, float &m_out, float &im_out) const
    {
        if (!P.mgrid) { m_out = maj; im_out = inv_maj; return sample_distance(u); }
        return dda_collision(P, mg, mocc, o, d, tmax, u, m_out, im_out);
    }

    // estimate_transmittance: ratio tracking (volpathsimple.py:436-504)
    template <bool ADJ>
    __device__ float estimate_transmittance(V3 o, V3 d, float tmax, Pcg32 &S, const float *adj, uint32_t *steps_out = nullptr)
    {
        float T = 1.0f;
        uint32_t steps = 0;
        for (;;) {
            float lm, lim;
            float dt = sample_collision(o, d, tmax, S.next_1d(), lm, lim);
            if (!(dt <= tmax)) break;                                   // :480-481
            V3 p = ray_at(o, d, dt);
            float sig = eval_sigma_t(P, p, occ);
            float tr = (lm - sig) * lim;                                // :473-476
            count(C_RT);
            if constexpr (ADJ) if (tr > 0.0f) {                         // :487-492
                float a = (adj[0] + adj[1]) + adj[2];
                splat_sigma_t<DEFER>(P, p, -(a * lim) / tr, rec);
                count(C_RT_ADJ);
            }
            T *= tr;                                                    // :495
            ++steps;
            o = p; tmax -= dt;                                          // :497-499
            if (T == 0.0f) break;                                       // :502
        }
        if (steps_out) *steps_out = steps;
        return T;
    }

    // sample_emitter (volpathsimple.py:406-433): emitter_val * transmittance in out[], ds.pdf returned
    template <bool ADJ>
    __device__ float sample_emitter(V3 p, Pcg32 &S, const float *adj, float out[3], int cmode = 0, uint4 *ce = nullptr)
    {
        float ux = S.next_1d(), uy = S.next_1d();                       // :418
        float val[3];
        V3 wd = emitter_sample_dir<ENV>(P, ux, uy);                          // Scene::sample_emitter_direction
        float pdf = emitter_sample_value<ENV>(P, wd, val);                   // ds.pdf, radiance / pdf
        float T = 0.0f;
        if (cmode == 2) {                                               // path cache: the primal pass walked this
            const uint4 e = *ce;
            T = __uint_as_float(e.x); S.state = ((uint64_t) e.z << 32) | e.y;
            if (COUNT) cnt[C_RT] += e.w;
        } else {
            uint32_t steps = 0;
            if (pdf != 0.0f) {                                          // sampling_worked :421-423
                Hit si = box_hit(P, p, wd);                             // :427-428
                if (si.valid) T = estimate_transmittance<ADJ>(p, wd, si.t, S, adj, &steps);
            }
            if (cmode == 1) *ce = make_uint4(__float_as_uint(T), (uint32_t) S.state, (uint32_t) (S.state >> 32), steps);
        }
#pragma unroll
        for (int k = 0; k < 3; ++k) out[k] = val[k] * T;
        return pdf;
    }

    // sample_emitter_for_nee (volpathsimple.py:380-403)
    template <bool ADJ>
    __device__ void sample_emitter_for_nee(V3 p, Pcg32 &S, const float beta[3], const float *dL,
                                           float contrib[3], int cmode = 0, uint4 *ce = nullptr)
    {
        Pcg32 clone = S;                                                // :383
        float emitted[3];
        float ds_pdf = sample_emitter<false>(p, S, nullptr, emitted, cmode, ce);   // :385
        float w = mis_weight(ds_pdf, kInvFourPi);                       // :391
#pragma unroll
        for (int k = 0; k < 3; ++k) contrib[k] = ((beta[k] * kInvFourPi) * w) * emitted[k];
        if constexpr (ADJ) {                                            // :393-401
            float adj[3] = { dL[0] * contrib[0], dL[1] * contrib[1], dL[2] * contrib[2] };
            float unused[3];
            (void) sample_emitter<true>(p, clone, adj, unused);
        }
    }

    // sample_real_interaction: delta tracking (volpathsimple.py:323-377)
    template <bool ATTACHED>
    __device__ Mei sample_real_interaction(const Ray &ray, Pcg32 &S, int cmode = 0, uint4 *ce = nullptr)
    {
        Mei mei; mei.valid = false; mei.t = kInf; mei.p = v3(0, 0, 0); mei.sigma_t = 0.0f;
        V3 ro = ray.o; float rmaxt = ray.maxt, running_t = 0.0f;
        uint32_t steps = 0;
        if (cmode == 2) {                                               // path cache: the primal pass walked this
            const uint4 e = *ce;
            mei.t = __uint_as_float(e.x); mei.valid = mei.t < kInf;
            S.state = ((uint64_t) e.z << 32) | e.y;
            if (COUNT) cnt[C_DT] += e.w;
        } else
        for (;;) {
            float lm, lim;
            float dt = sample_collision(ro, ray.d, rmaxt, S.next_1d(), lm, lim);   // :348
            if (!(dt <= rmaxt)) break;                                  // :358
            V3 p = ray_at(ro, ray.d, dt);
            float sig = eval_sigma_t(P, p, occ);
            count(C_DT); ++steps;
            float r = sig * lim;                                        // :354
            float u = S.next_1d();                                      // :359
            if (!(u >= r)) { mei.valid = true; mei.t = running_t + dt; break; }   // :351
            ro = p; rmaxt -= dt; running_t += dt;                       // :364-367
        }
        if (cmode == 1) *ce = make_uint4(__float_as_uint(mei.valid ? mei.t : kInf), (uint32_t) S.state, (uint32_t) (S.state >> 32), steps);
        if (mei.valid) {
            mei.p = ray_at(ray.o, ray.d, mei.t);                        // :371
            if (ATTACHED) { mei.sigma_t = eval_sigma_t(P, mei.p, occ); count(C_DT); }   // :373-375
        }
        return mei;
    }

    // Medium::sample_interaction_drt (call site volpathsimple.py:549-551): ratio
    // tracking over [0,maxt]; tentative collision i has weight T_i/majorant; one
    // is kept by weighted reservoir sampling; W = sum of weights.
    __device__ bool sample_interaction_drt(const Ray &ray, Pcg32 &A, float &t_out, float &W_out)
    {
        float t = 0.0f, T = 1.0f, wsum = 0.0f, tsel = kInf;
        bool valid = false;
        for (;;) {
            float lm, lim;
            if (P.mgrid) t += sample_collision(ray_at(ray.o, ray.d, t), ray.d, ray.maxt - t, A.next_1d(), lm, lim);
            else { t += sample_distance(A.next_1d()); lm = maj; lim = inv_maj; }
            if (!(t <= ray.maxt)) break;
            float sig = eval_sigma_t(P, ray_at(ray.o, ray.d, t), occ);
            count(C_DRT);
            float w = T * lim;
            wsum += w;
            float u = A.next_1d();
            if (w > 0.0f && u * wsum <= w) { tsel = t; valid = true; }
            T *= (lm - sig) * lim;
            if (T == 0.0f) break;
        }
        t_out = tsel; W_out = wsum;
        return valid;
    }

    // sample_recursive (volpathsimple.py:610-655)
    __device__ void sample_recursive(Pcg32 &A, V3 p, int depth, float Li[3])
    {
        Li[0] = Li[1] = Li[2] = 0.0f;
        if (P.use_nee) {                                                // :621-624
            const float one[3] = { 1.0f, 1.0f, 1.0f };
            float nee[3];
            sample_emitter_for_nee<false>(p, A, one, nullptr, nee);
#pragma unroll
            for (int k = 0; k < 3; ++k) Li[k] += nee[k];
        }
        (void) A.next_1d();                                             // :632
        float ux = A.next_1d(), uy = A.next_1d();
        Ray rr; rr.o = p; rr.d = square_to_uniform_sphere(ux, uy);
        Hit sn = box_hit(P, p, rr.d);                                   // :637
        rr.maxt = sn.valid ? sn.t : kLargest;                           // :639-640
        PathState ps;
        ps.depth = depth + 1; ps.si = sn; ps.last_pdf = kInvFourPi; ps.escaped = false;
        ps.active = (ps.depth < P.max_depth) && sn.valid;               // :647 (+ DESIGN.md deviation)
        float Lr[3];
        sample<false, true>(A, rr, nullptr, nullptr, &ps, Lr);          // :651
#pragma unroll
        for (int k = 0; k < 3; ++k) Li[k] += Lr[k];
    }

    // backpropagate_scattering_drt, final / quadratic branch (volpathsimple.py:543-581)
    __device__ void drt_backprop(Pcg32 &A, const Ray &ray, float si_t, int depth, const float adj[3])
    {
        Ray sub = ray;
        sub.maxt = isfinite(si_t) ? si_t : kLargest;                    // :544-545
        float tp, W;
        if (!sample_interaction_drt(sub, A, tp, W)) return;             // :550,558
        V3 p = ray_at(sub.o, sub.d, tp);
        float sig = eval_sigma_t(P, p, occ);                                 // :553-554
        count(C_DRT);
        float Li[3];
        sample_recursive(A, p, depth, Li);                              // :565-568
        float w = P.use_drt_mis ? 1.0f / (1.0f + sig * sig) : 1.0f;     // :571-575
        float alb[3];
        eval_albedo(P, p, alb);                                         // :578
        count(C_ALB);
        float ww = w * W;
        float gs = 0.0f, ga[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            float a = (ww * adj[k]) * Li[k];
            gs += a * alb[k];
            ga[k] = a * sig;
        }
        splat_scatter<DEFER>(P, p, gs, ga, rec); count(C_SC); count(C_SC_ALB);   // :577-581
    }

    // backpropagate_transmittance (volpathsimple.py:584-607)
    __device__ void backprop_transmittance(Pcg32 &A, const Ray &ray, float interval,
                                           const float dL[3], const float result[3])
    {
        float adjw = (dL[0] * result[0] + dL[1] * result[1]) + dL[2] * result[2];
        float g = -(adjw * (interval / 4.0f));
        for (int j = 0; j < 4; ++j) {
            float t = A.next_1d() * interval;                           // :595
            splat_sigma_t<DEFER>(P, ray_at(ray.o, ray.d, t), g, rec);
            count(C_TR);
        }
    }

    // VolpathSimpleIntegrator.sample (volpathsimple.py:38-290)
    template <bool ADJ, bool RECURSIVE>
    __device__ void sample(Pcg32 &S, Ray ray, const float *dL, const float *state_in,
                           const PathState *ps, float out[3])
    {
        float result[3] = { 0.0f, 0.0f, 0.0f };
        float beta[3] = { 1.0f, 1.0f, 1.0f };
        if (ADJ) { result[0] = state_in[0]; result[1] = state_in[1]; result[2] = state_in[2]; }

        bool active, escaped; int depth; Hit si;
        if (RECURSIVE) {                                                // :61-67
            active = ps->active; depth = ps->depth; si = ps->si; escaped = ps->escaped;
        } else {
            active = true; depth = 0; escaped = false;
            (void) S.next_1d();                                         // :71
            si = box_hit(P, ray.o, ray.d);                              // reach_medium :292-319
            if (!si.valid) { escaped = true; active = false; }
            else {
                ray.o = offset_p(si, ray.d);
                Hit sn = box_hit(P, ray.o, ray.d);
                if (!sn.valid) active = false;
                else { ray.maxt = sn.t; si = sn; }
            }
        }
        bool has_scattered = RECURSIVE ? (active && !escaped) : false;  // :84-89
        float last_pdf = RECURSIVE ? ps->last_pdf : 1.0f;

        // DRTReservoir(n=1) + DRTPathState (:94-96, :710-765)
        int r_depth = -1; float r_si_t = kInf; Ray r_ray = ray;
        float r_wsum[3] = { 0, 0, 0 }, r_cw[3] = { 0, 0, 0 };

        Pcg32 A; A.state = 0; A.inc = 1;
        if (active) (void) S.next_1d();                                 // :99
        if constexpr (ADJ) A.seed(P.alt_seed, ray_index);               // :100-107

        int it = 0;                                                     // bounce-loop iterations that reached their walk
        while (active) {                                                // :114
            float q = fminf(fmaxf(beta[0], fmaxf(beta[1], beta[2])), 0.99f);   // :117-121
            bool perform_rr = depth > P.rr_depth;
            float u_rr = S.next_1d();
            active = (beta[0] != 0.0f || beta[1] != 0.0f || beta[2] != 0.0f)
                     && (!perform_rr || (u_rr < q));
            if (perform_rr) { float iq = 1.0f / q; beta[0] *= iq; beta[1] *= iq; beta[2] *= iq; }
            if (!active) break;

            const int cmode = (!RECURSIVE && pc && it < (int) P.path_cache_cap) ? (int) P.path_cache_mode : 0;
            uint4 *ce = cmode ? pc + 2 * it : nullptr;
            ++it;
            Mei mei = sample_real_interaction<ADJ>(ray, S, cmode, ce);  // :126
            bool did_escape = !mei.valid, did_scatter = mei.valid;      // :130-134
            has_scattered |= did_scatter;

            float albedo[3] = { 1.0f, 1.0f, 1.0f };                     // :141
            if (did_scatter) { eval_albedo(P, mei.p, albedo); count(C_ALB); }

            if constexpr (ADJ) {
                if (P.use_drt) {                                        // :143-150
                    if (P.use_drt_subsampling) {                        // :521-539, :745-753
                        float u = A.next_1d();
                        float m = 0.0f;
#pragma unroll
                        for (int k = 0; k < 3; ++k) { r_wsum[k] += beta[k]; m += beta[k] / r_wsum[k]; }
                        m = m / 3.0f;
                        if (u <= m) {
                            r_cw[0] = beta[0]; r_cw[1] = beta[1]; r_cw[2] = beta[2];
                            r_depth = depth; r_si_t = si.t; r_ray = ray;
                        }
                    } else {
                        float adj[3] = { dL[0] * beta[0], dL[1] * beta[1], dL[2] * beta[2] };
                        drt_backprop(A, ray, si.t, depth, adj);
                    }
                }
                if ((!P.use_drt || P.use_drt_mis) && did_scatter) {     // :152-172
                    float w = 1.0f;
                    if (P.use_drt && P.use_drt_mis) {
                        float s2 = mei.sigma_t * mei.sigma_t;
                        w = s2 / (1.0f + s2);
                    }
                    float inv_pdf = 1.0f / mei.sigma_t;
                    float gs = 0.0f, ga[3];
#pragma unroll
                    for (int k = 0; k < 3; ++k) {
                        float Li = result[k] / fmaxf(1e-8f, albedo[k]); // :167
                        float a = ((w * dL[k]) * Li) * inv_pdf;
                        gs += a * albedo[k];
                        ga[k] = a * mei.sigma_t;
                    }
                    splat_scatter<DEFER>(P, mei.p, gs, ga, rec); count(C_SC); count(C_SC_ALB);
                }
                backprop_transmittance(A, ray, did_escape ? si.t : mei.t, dL, result);   // :181-189
            }

            beta[0] *= albedo[0]; beta[1] *= albedo[1]; beta[2] *= albedo[2];           // :193
            if (did_scatter) depth += 1;                                // :199
            active = did_scatter && (depth < P.max_depth);              // :200

            if (P.use_nee && did_scatter && active) {                   // :206-215
                float nee[3];
                sample_emitter_for_nee<ADJ>(mei.p, S, beta, dL, nee, cmode, ce ? ce + 1 : nullptr);
#pragma unroll
                for (int k = 0; k < 3; ++k) result[k] = ADJ ? result[k] - nee[k] : result[k] + nee[k];
            }

            if (did_scatter) {                                          // :221-230
                (void) S.next_1d();
                float ux = S.next_1d(), uy = S.next_1d();
                ray.o = mei.p; ray.d = square_to_uniform_sphere(ux, uy); ray.maxt = kLargest;
                last_pdf = kInvFourPi;
            }
            si = box_hit(P, ray.o, ray.d);                              // :233-235
            ray.maxt = isfinite(si.t) ? si.t : kLargest;
            if (did_scatter && !si.valid) active = false;               // :240-241
            if (did_escape) {                                           // :244-245
                if (si.valid) ray.o = offset_p(si, ray.d);
                escaped = true;
            }
        }

        if constexpr (ADJ) {
            if (P.use_drt && P.use_drt_subsampling && r_depth >= 0) {   // :249-259, :756-760
                float d = ((r_cw[0] + r_cw[1]) + r_cw[2]) / 3.0f;
                float ws = ((r_wsum[0] + r_wsum[1]) + r_wsum[2]) / 3.0f;
                float adj[3];
#pragma unroll
                for (int k = 0; k < 3; ++k) adj[k] = (d != 0.0f ? (ws * r_cw[k]) / d : 0.0f) * dL[k];
                drt_backprop(A, r_ray, r_si_t, r_depth, adj);
            }
        } else {                                                        // :263-287
            if (escaped && !(depth <= 0 && P.hide_emitters)) {
                float w = 1.0f, Le[3];
                if (P.use_nee) {
                    float epdf = 0.0f;                                  // :273-277
                    if (has_scattered) epdf = emitter_pdf<ENV>(P, ray.d);
                    w = mis_weight(last_pdf, epdf);
                }
                emitter_eval<ENV>(P, ray.d, Le);                        // :284
#pragma unroll
                for (int k = 0; k < 3; ++k) result[k] += (beta[k] * w) * Le[k];
            }
        }
        out[0] = result[0]; out[1] = result[1]; out[2] = result[2];
    }
};

template <bool ADJ, bool COUNT, bool ENV, bool DEFER>
__global__ void __launch_bounds__(256, DRT_TRACE_WAVES) trace_kernel(const Params P)
{
    // XCD-aware block -> ray-chunk map.  Workgroup b runs on XCD b % 8 (observed dispatch
    // order, used for speed only); consecutive ray chunks are spatially coherent, so XCD k is
    // given the k-th contiguous run of DRT_XCD_RUN workgroups of every group of 8 runs: its
    // private 4 MiB L2 then serves one image region instead of every 8th 8-pixel strip.
    uint32_t b = blockIdx.x;
#if DRT_XCD_RUN > 0
    {
        const uint32_t span = 8u * DRT_XCD_RUN;
        const uint32_t full = (gridDim.x / span) * span;     // tail blocks keep the identity map
        if (b < full) {
            uint32_t grp = b / span, r = b % span;
            b = grp * span + (r % 8u) * DRT_XCD_RUN + r / 8u;
        }
    }
#endif
    const uint64_t i_block = P.ray_first + (uint64_t) b * blockDim.x;
    uint64_t i = i_block + threadIdx.x;
    if constexpr (ADJ) {                                        // rays of similar length share a wave (ray_perm_kernel, drt_coop.hip)
        if (P.ray_perm) i = (i_block & ~(uint64_t) (kPermGroup - 1)) + P.ray_perm[i_block + threadIdx.x];
    }
    Tracer<COUNT, ENV, DEFER> tr(P);
    if constexpr (ADJ && DEFER) {
        __shared__ uint32_t rec_state[4 * 8];                   // per wave: cur[4], end[4]
        tr.rec = rec_state + (threadIdx.x >> 6) * 8;
        if ((threadIdx.x & 63) < 8) tr.rec[threadIdx.x & 63] = 0;
        coop_stage_sync();
    } else if constexpr (ADJ) {
        __shared__ uint32_t coop_rec[4 * 64 * kCoopDwords];
        tr.rec = coop_rec + (threadIdx.x >> 6) * (64 * kCoopDwords);
    }
    // empty-space bitmask -> LDS (4 KiB): most lookups of a sparse volume never leave the CU
    __shared__ uint32_t occ_lds[kOccWords];
    if (P.occ && !dbg(P.debug_flags, 16u)) {
        for (int w = threadIdx.x; w < P.occ_words; w += blockDim.x) occ_lds[w] = P.occ[w];
        __syncthreads();
        tr.occ = occ_lds;
    }
    __shared__ uint32_t mocc_lds[kOccWords];
    if (P.mgrid && P.mocc && P.mocc_words <= kOccWords && !dbg(P.debug_flags, 8388608u)) {
        for (int w = threadIdx.x; w < P.mocc_words; w += blockDim.x) mocc_lds[w] = P.mocc[w];
        __syncthreads();
        tr.mocc = mocc_lds;
    }
    if (i < P.n_rays) {
        uint64_t g64 = P.chunk ? P.ray_offset + (i / P.chunk) * P.stride + (i % P.chunk) : P.ray_offset + i;
        uint32_t gi = (uint32_t) g64;
        tr.ray_index = gi;
        Pcg32 S; S.seed(P.seed, gi);
        Ray ray;
        if (P.sensor_flow) {
            float ux = S.next_1d(), uy = S.next_1d();
            sensor_ray(P, gi / P.spp, ux, uy, ray.o, ray.d);
        } else {
            ray.o = v3(P.rays_o[3 * i], P.rays_o[3 * i + 1], P.rays_o[3 * i + 2]);
            ray.d = v3(P.rays_d[3 * i], P.rays_d[3 * i + 1], P.rays_d[3 * i + 2]);
        }
        ray.maxt = kLargest;
        tr.count(C_RAYS);
        if (P.path_cache_mode) {                                        // see trace_coop_kernel
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
        float L[3];
        if (ADJ) {
            float dL[3] = { P.dL[3 * i], P.dL[3 * i + 1], P.dL[3 * i + 2] };
            float Lin[3] = { P.L_in[3 * i], P.L_in[3 * i + 1], P.L_in[3 * i + 2] };
            tr.template sample<true, false>(S, ray, dL, Lin, nullptr, L);
        } else {
            tr.template sample<false, false>(S, ray, nullptr, nullptr, nullptr, L);
            P.L_out[3 * i] = L[0]; P.L_out[3 * i + 1] = L[1]; P.L_out[3 * i + 2] = L[2];
        }
    }
    if constexpr (ADJ && DEFER) close_records(P, tr.rec);
    if (COUNT) {
        // wave reduction, then one device atomic per wave and slot
#pragma unroll
        for (int s = 0; s < C_COUNT; ++s) {
            uint32_t v = tr.cnt[s];
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
            if ((threadIdx.x & 63) == 0 && v) atomicAdd(P.counters + s, (unsigned long long) v);
        }
    }
}

#endif  // DRT_TEST_HOOKS (Tracer, trace_kernel)

// majorant = scale * max(sigma_t grid) (Medium::get_majorant with a global
// majorant; carries no gradient, refreshed on parameter update - optimize.py:195-199)
__global__ void __launch_bounds__(256) majorant_reduce_kernel(const float *sigma_t, size_t n, uint32_t *max_bits)
{
    float m = 0.0f;
    for (size_t i = (size_t) blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t) gridDim.x * blockDim.x)
        m = fmaxf(m, sigma_t[i]);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_down(m, off, 64));
    // non-negative floats order like their bit patterns
    if ((threadIdx.x & 63) == 0) atomicMax(max_bits, __float_as_uint(m));
}

__global__ void majorant_finalize_kernel(const uint32_t *max_bits, float scale, float *majorant)
{
    float m = __uint_as_float(*max_bits) * scale;
    majorant[0] = m;
    majorant[1] = (m != 0.0f) ? 1.0f / m : 0.0f;
}

// Majorant supergrid: cell (I,J,K) = scale * max over the voxels a trilinear lookup inside the
// cell can touch, padded by one voxel: [floor(I*res/G) - 1, ceil((I+1)*res/G)] per axis, clamped; rounded up to bf16.
// One wavefront per cell.
// max_bits (optional): the cells cover every voxel, so the largest un-rounded cell maximum IS the grid's maximum - the global
// majorant comes out of this pass and the separate reduction over the grid (majorant_reduce_kernel) is not launched.
__global__ void __launch_bounds__(256) majorant_grid_kernel(const float *sigma_t, int rx, int ry, int rz,
                                                            int gx, int gy, int gz, float scale, float *out, uint32_t *mask,
                                                            uint32_t *max_bits)
{
    uint32_t cell = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, lane = threadIdx.x & 63;
    if (cell >= (uint32_t) gx * gy * gz) return;
    int I = cell % gx, J = (cell / gx) % gy, K = cell / (gx * gy);
    auto lo = [](int c, int R, int G) { int v = (int)(((long long) c * R) / G) - 1; return v < 0 ? 0 : v; };
    auto hi = [](int c, int R, int G) { int v = (int)((((long long)(c + 1)) * R + G - 1) / G); return v > R - 1 ? R - 1 : v; };
    int x0 = lo(I, rx, gx), x1 = hi(I, rx, gx), y0 = lo(J, ry, gy), y1 = hi(J, ry, gy), z0 = lo(K, rz, gz), z1 = hi(K, rz, gz);
    int nx = x1 - x0 + 1, ny = y1 - y0 + 1, nz = z1 - z0 + 1;
    float m = 0.0f;
    for (int i = lane; i < nx * ny * nz; i += 64) {
        int x = x0 + i % nx, y = y0 + (i / nx) % ny, z = z0 + i / (nx * ny);
        m = fmaxf(m, sigma_t[((size_t) z * ry + y) * rx + x]);
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_down(m, off, 64));
    if (lane == 0) {
        // rounded UP to the next bf16-representable value (a majorant only has to bound; at most 0.8 % looser): the
        // supergrid tracer keeps the cells as 16-bit values in LDS (drt_super.hip) - as oracle/drt_oracle.c scene_init
        if (max_bits && __float_as_uint(m) > *(volatile uint32_t *) max_bits) atomicMax(max_bits, __float_as_uint(m));   // (few cells get past the read)
        uint32_t b = __float_as_uint(m * scale);
        if (b & 0xffffu) b = (b | 0xffffu) + 1u;
        const float mm = __uint_as_float(b);
        out[cell] = mm;
        if (mask && mm > 0.0f) atomicOr(mask + (cell >> 5), 1u << (cell & 31u));   // (mask zeroed by the launcher)
    }
}

// Empty-space bitmask (Params::occ): one thread per cell ORs the voxels [c*S, (c+1)*S] per axis
// (S = 2^shift; the +1 is the far corner of a lookup whose base corner is the cell's last voxel).
__global__ void __launch_bounds__(256) occupancy_kernel(const float *sigma_t, int rx, int ry, int rz, int shift,
                                                        int ox, int oy, int oz, uint32_t *occ)
{
    uint32_t c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= (uint32_t) ox * oy * oz) return;
    int I = c % ox, J = (c / ox) % oy, K = c / (ox * oy);
    int S = 1 << shift;
    int x1 = min(rx - 1, (I + 1) * S), y1 = min(ry - 1, (J + 1) * S), z1 = min(rz - 1, (K + 1) * S);
    bool any = false;
    for (int z = K * S; z <= z1 && !any; ++z)
        for (int y = J * S; y <= y1 && !any; ++y)
            for (int x = I * S; x <= x1; ++x)
                if (sigma_t[((size_t) z * ry + y) * rx + x] != 0.0f) { any = true; break; }
    if (any) atomicOr(occ + (c >> 5), 1u << (c & 31));
}

// Caller's (Z,Y,X,1) sigma_t -> apron-brick copy (see eval_sigma_t); one thread per stored float.
__global__ void __launch_bounds__(256) brick_sigma_kernel(const float *src, float *dst, int rx, int ry, int rz,
                                                          int nbx, int nby)
{
    size_t t = (size_t) blockIdx.x * blockDim.x + threadIdx.x;
    size_t total = (size_t) nbx * nby * rz * 32;
    if (t >= total) return;
    uint32_t o = (uint32_t)(t & 31), brick = (uint32_t)(t >> 5);
    uint32_t bx = brick % (uint32_t) nbx, r = brick / (uint32_t) nbx;
    uint32_t by = r % (uint32_t) nby, z = r / (uint32_t) nby;
    int x = min((int)(3 * bx + (o & 3)), rx - 1);
    int y = min((int)(3 * by + ((o >> 2) & 3)), ry - 1);
    int zz = min((int)(z + (o >> 4)), rz - 1);
    dst[t] = src[((size_t) zz * ry + y) * rx + x];
}

// Gradient scratch (apron layout, make_grad_indices) -> caller's (Z,Y,X,1) / (Z,Y,X,3) buffers:
// sum the up to 8 slots of every voxel, += into the caller's grids (the ABI accumulates) and reset
// the scratch for the next launch.  Each slot belongs to exactly one voxel.
//
// Streaming form: a thread owns the LINE column (bx, y_begin..y_end, Z), i.e. the voxels
// X = 3bx..3bx+2 of row Z, and marches y along consecutive line-rows with 16-byte loads.  Line
// (z, y, bx) holds contributions to voxel rows (y, z) [quad 0], (y+1, z) [quad 1], (y, z+1)
// [quad 2] and (y+1, z+1) [quad 3]: the thread reads quads 0/1 of its own line and quads 2/3 of
// line (Z-1, y, bx) (the neighbour wave's own line: L1 hit), carries the dy = 1 quads in registers
// to the next step, and takes the seam slot (slot 3 of line bx-1 is voxel 3bx) from the previous
// lane.  Every 64-byte line crosses the L1 four times (one-dword-per-voxel gathers crossed it 16
// times and were bound by that, 2 TB/s) and HBM about once.  The summation order per voxel is
// fixed (dz, dy, own slot before seam slot): the result does not depend on the decomposition.
constexpr int kUntileLanes = 32;                    // lines along x per workgroup row
constexpr int kUntileRows = 8;                      // z rows per workgroup

struct Quad { float v[4]; };

// read a 16-byte quad and reset it; the consumer of slot 3 is the next line's thread: when that is
// another wave (`keep3`), slot 3 is left for it to reset
__device__ __forceinline__ Quad take_quad(float *p, bool keep3, bool wr)
{
    const float4 q = *reinterpret_cast<const float4 *>(p);
    if (wr && (q.x != 0.0f || q.y != 0.0f || q.z != 0.0f || q.w != 0.0f)) {
        if (!keep3) *reinterpret_cast<float4 *>(p) = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        else { p[0] = 0.0f; p[1] = 0.0f; p[2] = 0.0f; }
    }
    Quad r; r.v[0] = q.x; r.v[1] = q.y; r.v[2] = q.z; r.v[3] = q.w;
    return r;
}

__device__ __forceinline__ float take_slot(float *p, bool wr)
{
    float g = *p;
    if (wr && g != 0.0f) *p = 0.0f;
    return g;
}

// slot 3 of the previous line: from the previous lane, or (first lane of a row segment) from memory
__device__ __forceinline__ float seam_value(const Quad &q, float *line, int quad, bool first, bool have_prev, bool active, bool wr)
{
    float s = __shfl_up(q.v[3], 1);
    if (first) s = (have_prev && active) ? take_slot(line - 16 + 4 * quad + 3, wr) : 0.0f;
    return s;
}

template <int NPL>
__device__ __forceinline__ void untile_march(const Params &P, int plane0, int bx, int Z, int y_begin, int y_end, bool valid)
{
    const bool first = (threadIdx.x & (kUntileLanes - 1)) == 0;
    const bool last_lane = (threadIdx.x & (kUntileLanes - 1)) == kUntileLanes - 1;
    const bool have_prev = bx > 0;
    const bool keep3 = last_lane && bx + 1 < P.gt_nbx;      // slot 3 is read by another wave's first lane
    const bool has_z = Z > 0;
    const bool wr = !dbg(P.debug_flags, 64u);
    const size_t row = (size_t) P.gt_nbx << 4;              // floats per line-row
    const int X = 3 * bx, nvx = min(3, P.rx - X);           // voxels of this line that exist
    Quad c1[NPL], c3[NPL];                                  // dy = 1 quads of the previous step
    float s1[NPL], s3[NPL];                                 // ... and their seam values
    const Quad zero = {{0.0f, 0.0f, 0.0f, 0.0f}};
#pragma unroll
    for (int k = 0; k < NPL; ++k) { c1[k] = zero; c3[k] = zero; s1[k] = 0.0f; s3[k] = 0.0f; }
    for (int y = max(y_begin - 1, 0); y < y_end; ++y) {
        const bool out = valid && y >= y_begin;             // dy = 0 quads feed voxel row Y = y (ours)
        const bool keep = valid && y + 1 < y_end;           // dy = 1 quads feed voxel row Y = y + 1 (ours?)
        const size_t offA = ((size_t) Z * P.ry + y) * row + ((size_t) bx << 4);     // line (Z, y, bx)
        const size_t offB = offA - (size_t) P.ry * row;                             // line (Z - 1, y, bx)
        float acc[NPL][3];
#pragma unroll
        for (int k = 0; k < NPL; ++k) {
            float *plane = P.gt + (size_t) (plane0 + k) * P.gt_plane;
            float *A = plane + offA, *B = plane + offB;
            Quad q0 = zero, q1 = zero, q2 = zero, q3 = zero;
            if (out) q0 = take_quad(A, keep3, wr);
            if (keep) q1 = take_quad(A + 4, keep3, wr);
            if (out && has_z) q2 = take_quad(B + 8, keep3, wr);
            if (keep && has_z) q3 = take_quad(B + 12, keep3, wr);
            const float e0 = seam_value(q0, A, 0, first, have_prev, out, wr);
            const float e1 = seam_value(q1, A, 1, first, have_prev, keep, wr);
            const float e2 = seam_value(q2, B, 2, first, have_prev, out && has_z, wr);
            const float e3 = seam_value(q3, B, 3, first, have_prev, keep && has_z, wr);
            // fixed order: (dz0,dy0) (dz0,dy1) (dz1,dy0) (dz1,dy1), own slot then seam slot
            float a0 = 0.0f;
            a0 += q0.v[0]; a0 += e0; a0 += c1[k].v[0]; a0 += s1[k]; a0 += q2.v[0]; a0 += e2; a0 += c3[k].v[0]; a0 += s3[k];
            acc[k][0] = a0;
#pragma unroll
            for (int j = 1; j < 3; ++j) {
                float a = 0.0f;
                a += q0.v[j]; a += c1[k].v[j]; a += q2.v[j]; a += c3[k].v[j];
                acc[k][j] = a;
            }
            c1[k] = q1; c3[k] = q3; s1[k] = e1; s3[k] = e3;
        }
        if (out) {
            const size_t v = ((size_t) Z * P.ry + y) * P.rx + X;
            for (int j = 0; j < nvx; ++j) {
                // (atomic adds: every voxel is flushed by ONE thread of this kernel, but the fused nerf + volpathsimple pass runs the nerf half's
                //  window flush - atomics into the same grids - on another stream beside it, drt_capi.cpp: drt_fused_render_backward; only
                //  the non-zero entries, i.e. the support of this job's gradient, pay for it)
                if (NPL == 1) {
                    if (acc[0][j] != 0.0f) atomicAdd(P.g_sigma + v + j, acc[0][j]);
                } else {
#pragma unroll
                    for (int k = 0; k < NPL; ++k)
                        if (acc[k][j] != 0.0f) atomicAdd(P.g_albedo + 3 * (v + j) + k, acc[k][j]);
                }
            }
        }
    }
}

__global__ void __launch_bounds__(kUntileLanes * kUntileRows) untile_gradients_kernel(const Params P, int n_chunks, int chunk)
{
    const int bx = blockIdx.x * kUntileLanes + (threadIdx.x & (kUntileLanes - 1));
    const int Z = blockIdx.y * kUntileRows + (threadIdx.x / kUntileLanes);
    const int group = blockIdx.z / n_chunks;        // 0: sigma_t plane, 1: the three colour planes
    const int y_begin = (blockIdx.z - group * n_chunks) * chunk;
    const int y_end = min(y_begin + chunk, P.ry);
    const bool valid = bx < P.gt_nbx && Z < P.rz;   // invalid lanes stay for the shuffles
    const int bxc = valid ? bx : 0, Zc = valid ? Z : 0;
    if (group == 0) untile_march<1>(P, 0, bxc, Zc, y_begin, y_end, valid);
    else untile_march<3>(P, 1, bxc, Zc, y_begin, y_end, valid);
}

// sample_batch_pixels + sample_batch_rays (python/batched.py:397-467): one thread per ray r of the
// batch; b = r / spp picks (sensor, pixel) from sampler 0's lane b, the sub-pixel offset comes from
// the ray sampler's lane r.  `sensors`: n_sensors x 16 floats {origin3, left3, up3, dir3, tan_x,
// tan_y, width, height}.
// `batch_first`: first batch entry of this rank's share (sharded batches): the samplers' lanes are
// the GLOBAL batch entry / ray index, the outputs are local (entry b - batch_first, ray r - batch_first * spp).
__global__ void __launch_bounds__(256) batch_raygen_kernel(const float *sensors, int n_sensors, uint32_t batch_first,
                                                           uint32_t batch_size,
                                                           uint32_t spp, uint32_t seed_pixels, uint32_t seed_rays,
                                                           float *rays_o, float *rays_d, uint32_t *sensor_idx,
                                                           uint32_t *pixels)
{
    const uint32_t rl = blockIdx.x * blockDim.x + threadIdx.x;
    if (rl >= batch_size * spp) return;
    const uint32_t r = rl + batch_first * spp;
    uint32_t b = r / spp;
    Pcg32 S0; S0.seed(seed_pixels, b);
    float us = S0.next_1d(), ux = S0.next_1d(), uy = S0.next_1d();
    uint32_t si = (uint32_t)((float) n_sensors * us);
    if (si >= (uint32_t) n_sensors) si = (uint32_t) n_sensors - 1;
    const float *sn = sensors + 16 * si;
    Params P;
#pragma unroll
    for (int k = 0; k < 3; ++k) { P.cam_o[k] = sn[k]; P.cam_left[k] = sn[3 + k]; P.cam_up[k] = sn[6 + k]; P.cam_dir[k] = sn[9 + k]; }
    P.tan_x = sn[12]; P.tan_y = sn[13]; P.width = (int) sn[14]; P.height = (int) sn[15];
    uint32_t px = (uint32_t)((float) P.width * ux), py = (uint32_t)((float) P.height * uy);
    if (r == b * spp) {
        const uint32_t bl = b - batch_first;
        if (sensor_idx) sensor_idx[bl] = si;
        if (pixels) { pixels[2 * bl] = px; pixels[2 * bl + 1] = py; }
    }
    Pcg32 S1; S1.seed(seed_rays, r);
    float ox = S1.next_1d(), oy = S1.next_1d();
    V3 o, d;
    sensor_ray(P, py * (uint32_t) P.width + px, ox, oy, o, d);
    rays_o[3 * (size_t) rl] = o.x; rays_o[3 * (size_t) rl + 1] = o.y; rays_o[3 * (size_t) rl + 2] = o.z;
    rays_d[3 * (size_t) rl] = d.x; rays_d[3 * (size_t) rl + 1] = d.y; rays_d[3 * (size_t) rl + 2] = d.z;
}

// box film: image[p] = mean_spp L (batched.py:176-197)
__global__ void __launch_bounds__(256) film_develop_kernel(const float *L, uint64_t n_pixels, uint32_t spp, float *image)
{
    uint64_t t = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x;   // one thread per (pixel, channel)
    if (t >= n_pixels * 3) return;
    uint64_t p = t / 3; uint32_t c = (uint32_t)(t - p * 3);
    const float *src = L + 3 * p * spp + c;
    float s = 0.0f;
    for (uint32_t j = 0; j < spp; ++j) s += src[3 * (uint64_t) j];
    image[t] = s * (1.0f / (float) spp);
}

// the same for many samples per pixel (the optimisation loop develops 1024 spp): one wave per pixel - lane l sums the
// samples l, l + 64, ... (coalesced 768-byte rows), then a fixed-order wave reduction; 0.41 -> 0.1 ms for 32768 x 1024
__global__ void __launch_bounds__(256) film_develop_wave_kernel(const float *L, uint64_t n_pixels, uint32_t spp, float *image)
{
    const uint64_t p = (uint64_t) blockIdx.x * 4 + (threadIdx.x >> 6);
    if (p >= n_pixels) return;
    const uint32_t lane = threadIdx.x & 63u;
    const float *src = L + 3 * p * spp;
    float s0 = 0.0f, s1 = 0.0f, s2 = 0.0f;
    for (uint32_t j = lane; j < spp; j += 64u) { s0 += src[3 * (uint64_t) j]; s1 += src[3 * (uint64_t) j + 1]; s2 += src[3 * (uint64_t) j + 2]; }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) { s0 += __shfl_down(s0, off, 64); s1 += __shfl_down(s1, off, 64); s2 += __shfl_down(s2, off, 64); }
    if (lane == 0) {
        const float inv = 1.0f / (float) spp;
        image[3 * p] = s0 * inv; image[3 * p + 1] = s1 * inv; image[3 * p + 2] = s2 * inv;
    }
}

// dL[i] = grad_image[i / spp] / spp (batched.py:298-306)
__global__ void __launch_bounds__(256) film_backward_kernel(const float *grad_image, uint64_t n_pixels, uint32_t spp, float *dL)
{
    uint64_t t = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x;   // one thread per (sample, channel)
    if (t >= n_pixels * spp * 3) return;
    uint64_t i = t / 3; uint32_t c = (uint32_t)(t - i * 3);
    dL[t] = grad_image[3 * (i / spp) + c] * (1.0f / (float) spp);
}

// Primitive evaluation for the parity tests (tests/test_gpu_primitives.py): one
// thread per item, 6 floats in, 6 floats out.
__global__ void __launch_bounds__(256) debug_eval_kernel(const Params P, int op, const float *in, uint64_t n, float *out)
{
    uint64_t i = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float *a = in + 6 * i;
    float *o = out + 6 * i;
    for (int k = 0; k < 6; ++k) o[k] = 0.0f;
    switch (op) {
        case 0: o[0] = drt_logf(a[0]); break;
        case 1: drt_sincos_2pi(a[0], o[0], o[1]); break;
        case 2: { V3 d = square_to_uniform_sphere(a[0], a[1]); o[0] = d.x; o[1] = d.y; o[2] = d.z; } break;
        case 3: o[0] = eval_sigma_t(P, v3(a[0], a[1], a[2]), P.occ); break;
        case 4: eval_albedo(P, v3(a[0], a[1], a[2]), o); break;
        case 5: {
            Hit h = box_hit(P, v3(a[0], a[1], a[2]), v3(a[3], a[4], a[5]));
            o[0] = h.valid ? 1.0f : 0.0f; o[1] = h.t; o[2] = h.n.x; o[3] = h.n.y; o[4] = h.n.z;
        } break;
        case 6: {
            Pcg32 S; S.seed(__float_as_uint(a[0]), __float_as_uint(a[1]));
            for (int k = 0; k < 6; ++k) o[k] = S.next_1d();
        } break;
        case 7: {
            V3 ro, rd; sensor_ray(P, __float_as_uint(a[0]), a[1], a[2], ro, rd);
            o[0] = ro.x; o[1] = ro.y; o[2] = ro.z; o[3] = rd.x; o[4] = rd.y; o[5] = rd.z;
        } break;
        case 10: o[0] = drt_expf(a[0]); break;
        case 11: o[0] = drt_atan2f(a[0], a[1]); break;
        case 12: if (P.env_pix) {                       // envmap: eval(d) rgb, pdf_direction(d)
            V3 d = v3(a[0], a[1], a[2]); float Le[3];
            envmap_eval(P, d, Le); o[0] = Le[0]; o[1] = Le[1]; o[2] = Le[2]; o[3] = envmap_pdf(P, d);
        } break;
        case 13: if (P.env_pix) {                       // envmap: sample_direction(u1, u2) -> d, pdf
            V3 d = envmap_sample_dir(P, a[0], a[1]);
            o[0] = d.x; o[1] = d.y; o[2] = d.z; o[3] = envmap_pdf(P, d);
        } break;
        case 14: {                                      // E2: sample_interaction_drt from o along d to the box exit
            coop::Ray r; r.o = v3(a[0], a[1], a[2]); r.d = v3(a[3], a[4], a[5]);
            Hit h = box_hit(P, r.o, r.d);
            r.maxt = h.valid ? h.t : 0.0f;
            Pcg32 A; A.seed(0x5eedu, (uint32_t) i);
            float t = kInf, W = 0.0f;
            bool ok;
            if (P.mgrid) { coop::CoopTracer<false, false, false, false, false, true> tr(P); tr.occ = P.occ; tr.mocc = P.mocc; ok = tr.sample_interaction_drt(r, A, t, W); }
            else { coop::CoopTracer<false, false, false> tr(P); tr.occ = P.occ; ok = tr.sample_interaction_drt(r, A, t, W); }
            o[0] = ok ? 1.0f : 0.0f; o[1] = t; o[2] = W; o[3] = r.maxt;
        } break;
        case 9: if (P.mgrid) { o[0] = P.mgrid[__float_as_uint(a[0])]; } break;
        case 8: o[0] = mis_weight(a[0], a[1]); o[1] = a[0] / a[1]; o[2] = sqrtf(a[0]); o[3] = fmaf(a[0], a[1], a[2]); break;
        default: break;
    }
}

hipError_t launch_debug_eval(const Params &P, int op, const float *in, uint64_t n, float *out, hipStream_t stream)
{
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(debug_eval_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, P, op, in, n, out);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------
// launch wrappers (host)
// ---------------------------------------------------------------------------
#ifdef DRT_TEST_HOOKS
hipError_t launch_trace(const Params &P, bool adjoint, bool count, hipStream_t stream)
{
    if (P.n_rays <= P.ray_first) return hipSuccess;
    dim3 block(256), grid((unsigned)((P.n_rays - P.ray_first + 255) / 256));
    const int variant = (adjoint ? 4 : 0) | (count ? 2 : 0) | (P.env_pix ? 1 : 0);
    const bool defer = adjoint && P.rec_buf[0] != nullptr;
    switch (variant) {
        case 0: hipLaunchKernelGGL((trace_kernel<false, false, false, false>), grid, block, 0, stream, P); break;
        case 1: hipLaunchKernelGGL((trace_kernel<false, false, true, false>), grid, block, 0, stream, P); break;
        case 2: hipLaunchKernelGGL((trace_kernel<false, true, false, false>), grid, block, 0, stream, P); break;
        case 3: hipLaunchKernelGGL((trace_kernel<false, true, true, false>), grid, block, 0, stream, P); break;
        case 4: if (defer) hipLaunchKernelGGL((trace_kernel<true, false, false, true>), grid, block, 0, stream, P);
                else hipLaunchKernelGGL((trace_kernel<true, false, false, false>), grid, block, 0, stream, P);
                break;
        case 5: if (defer) hipLaunchKernelGGL((trace_kernel<true, false, true, true>), grid, block, 0, stream, P);
                else hipLaunchKernelGGL((trace_kernel<true, false, true, false>), grid, block, 0, stream, P);
                break;
        case 6: if (defer) hipLaunchKernelGGL((trace_kernel<true, true, false, true>), grid, block, 0, stream, P);
                else hipLaunchKernelGGL((trace_kernel<true, true, false, false>), grid, block, 0, stream, P);
                break;
        default: if (defer) hipLaunchKernelGGL((trace_kernel<true, true, true, true>), grid, block, 0, stream, P);
                 else hipLaunchKernelGGL((trace_kernel<true, true, true, false>), grid, block, 0, stream, P);
                 break;
    }
    return hipGetLastError();
}
#endif  // DRT_TEST_HOOKS

// bit c of `dil`: supergrid cell c or one of its 26 neighbours has a non-zero majorant (Params::mocc_dil: the per-pixel emptiness proof of
// build_unit_empty tests ONE bit per sample of the pixel's centre ray).  One thread per word of 32 cells.
__global__ void __launch_bounds__(256) dilate_mask_kernel(const uint32_t *mask, int gx, int gy, int gz, uint32_t *dil)
{
    const uint32_t w = blockIdx.x * blockDim.x + threadIdx.x, cells = (uint32_t) gx * gy * gz;
    if (w >= (cells + 31u) / 32u) return;
    uint32_t out = 0;
    for (uint32_t b = 0; b < 32u && w * 32u + b < cells; ++b) {
        const uint32_t c = w * 32u + b;
        const int x = (int) (c % (uint32_t) gx), y = (int) ((c / (uint32_t) gx) % (uint32_t) gy), z = (int) (c / ((uint32_t) gx * (uint32_t) gy));
        bool any = false;
        for (int dz = -1; dz <= 1 && !any; ++dz)
            for (int dy = -1; dy <= 1 && !any; ++dy)
                for (int dx = -1; dx <= 1; ++dx) {
                    const int xx = x + dx, yy = y + dy, zz = z + dz;
                    if (xx < 0 || yy < 0 || zz < 0 || xx >= gx || yy >= gy || zz >= gz) continue;
                    const uint32_t cc = (uint32_t) ((zz * gy + yy) * gx + xx);
                    if ((mask[cc >> 5] >> (cc & 31u)) & 1u) { any = true; break; }
                }
        out |= any ? 1u << b : 0u;
    }
    dil[w] = out;
}

hipError_t launch_majorant_grid(const float *sigma_t, int rx, int ry, int rz, int gx, int gy, int gz, float scale,
                                float *out, uint32_t *mask, hipStream_t stream, uint32_t *max_bits, float *majorant, uint32_t *mask_dil)
{
    uint32_t cells = (uint32_t) gx * gy * gz;
    if (max_bits) {
        hipError_t e = hipMemsetAsync(max_bits, 0, sizeof(uint32_t), stream);
        if (e != hipSuccess) return e;
    }
    if (mask) {
        hipError_t e = hipMemsetAsync(mask, 0, (size_t) ((cells + 31) / 32) * sizeof(uint32_t), stream);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(majorant_grid_kernel, dim3((cells * 64 + 255) / 256), dim3(256), 0, stream, sigma_t, rx, ry, rz, gx, gy, gz, scale, out, mask, max_bits);
    if (max_bits && majorant) hipLaunchKernelGGL(majorant_finalize_kernel, dim3(1), dim3(1), 0, stream, max_bits, scale, majorant);
    if (mask && mask_dil) hipLaunchKernelGGL(dilate_mask_kernel, dim3(((cells + 31) / 32 + 255) / 256), dim3(256), 0, stream, mask, gx, gy, gz, mask_dil);
    return hipGetLastError();
}

hipError_t launch_occupancy(const float *sigma_t, int rx, int ry, int rz, int shift, int ox, int oy, int oz,
                            uint32_t *occ, int words, hipStream_t stream)
{
    hipError_t e = hipMemsetAsync(occ, 0, (size_t) words * 4, stream);
    if (e != hipSuccess) return e;
    uint32_t cells = (uint32_t) ox * oy * oz;
    hipLaunchKernelGGL(occupancy_kernel, dim3((cells + 255) / 256), dim3(256), 0, stream, sigma_t, rx, ry, rz, shift, ox, oy, oz, occ);
    return hipGetLastError();
}

hipError_t launch_brick_sigma(const float *src, float *dst, int rx, int ry, int rz, int nbx, int nby,
                              hipStream_t stream)
{
    size_t total = (size_t) nbx * nby * rz * 32;
    hipLaunchKernelGGL(brick_sigma_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, src, dst, rx, ry, rz, nbx, nby);
    return hipGetLastError();
}

hipError_t launch_nerf(const Params &P, bool adjoint, bool count, hipStream_t stream) { return launch_nerf_t(P, adjoint, count, stream); }

hipError_t launch_untile(const Params &P, hipStream_t stream)
{
    constexpr int chunk = 8;                                     // y rows per thread march
    const int nc = (P.ry + chunk - 1) / chunk;
    hipLaunchKernelGGL(untile_gradients_kernel,
                       dim3((P.gt_nbx + kUntileLanes - 1) / kUntileLanes, (P.rz + kUntileRows - 1) / kUntileRows, nc * 2),
                       dim3(kUntileLanes * kUntileRows), 0, stream, P, nc, chunk);
    return hipGetLastError();
}

hipError_t launch_majorant(const float *sigma_t, size_t n, float scale, uint32_t *scratch_bits,
                           float *majorant, hipStream_t stream)
{
    hipError_t e = hipMemsetAsync(scratch_bits, 0, sizeof(uint32_t), stream);
    if (e != hipSuccess) return e;
    unsigned blocks = (unsigned) ((n + 255) / 256);
    if (blocks > 2048) blocks = 2048;
    if (blocks == 0) blocks = 1;
    hipLaunchKernelGGL(majorant_reduce_kernel, dim3(blocks), dim3(256), 0, stream, sigma_t, n, scratch_bits);
    hipLaunchKernelGGL(majorant_finalize_kernel, dim3(1), dim3(1), 0, stream, scratch_bits, scale, majorant);
    return hipGetLastError();
}

hipError_t launch_batch_raygen(const float *sensors, int n_sensors, uint32_t batch_first, uint32_t batch_size, uint32_t spp,
                               uint32_t seed_pixels, uint32_t seed_rays, float *rays_o, float *rays_d, uint32_t *sensor_idx,
                               uint32_t *pixels, hipStream_t stream)
{
    uint64_t n = (uint64_t) batch_size * spp;
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(batch_raygen_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, sensors, n_sensors,
                       batch_first, batch_size, spp, seed_pixels, seed_rays, rays_o, rays_d, sensor_idx, pixels);
    return hipGetLastError();
}

hipError_t launch_film_develop(const float *L, uint64_t n_pixels, uint32_t spp, float *image, hipStream_t stream)
{
    uint64_t n = n_pixels * 3;
    if (n == 0) return hipSuccess;
    if (spp >= 128) hipLaunchKernelGGL(film_develop_wave_kernel, dim3((unsigned)((n_pixels + 3) / 4)), dim3(256), 0, stream, L, n_pixels, spp, image);
    else hipLaunchKernelGGL(film_develop_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, L, n_pixels, spp, image);
    return hipGetLastError();
}

// One Adam step in one pass over (p, g, m, v) (N2; mi.ad.Adam as optimize.py:329,352 uses it):
//   m = b1 m + (1 - b1) g;  v = b2 v + (1 - b2) g^2;  p -= lr_t m / (sqrt(v) + eps)      (lr_t carries the bias corrections)
// instead of seven elementwise passes: 0.55 -> 0.2 ms per step for the 256^3 x (1 + 3) parameters.
// lo / hi: the parameter's valid range applied to the updated value in the same pass (enforce_valid_params, python/optimize.py:169-179;
// torch.clamp's semantics: a NaN stays a NaN); -inf / +inf: no clamp
__global__ void __launch_bounds__(256) adam_step_kernel(float *p, const float *g, float *m, float *v, uint64_t n,
                                                        float b1, float a1, float b2, float a2, float eps, float lr_t, float lo, float hi)
{
    const uint64_t stride = (uint64_t) gridDim.x * 256;
    const uint64_t n4 = n / 4;
    float4 *p4 = reinterpret_cast<float4 *>(p), *m4 = reinterpret_cast<float4 *>(m), *v4 = reinterpret_cast<float4 *>(v);
    const float4 *g4 = reinterpret_cast<const float4 *>(g);
    auto upd = [&](float &pp, float gg, float &mm, float &vv) {
        mm = mm * b1 + a1 * gg;
        vv = vv * b2 + a2 * (gg * gg);
        pp = pp - lr_t * (mm / (sqrtf(vv) + eps));
        pp = pp < lo ? lo : (pp > hi ? hi : pp);
    };
    for (uint64_t i = (uint64_t) blockIdx.x * 256 + threadIdx.x; i < n4; i += stride) {
        float4 P = p4[i], M = m4[i], V = v4[i]; const float4 G = g4[i];
        upd(P.x, G.x, M.x, V.x); upd(P.y, G.y, M.y, V.y); upd(P.z, G.z, M.z, V.z); upd(P.w, G.w, M.w, V.w);
        p4[i] = P; m4[i] = M; v4[i] = V;
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) { const uint64_t i = n4 * 4 + threadIdx.x; upd(p[i], g[i], m[i], v[i]); }
}

hipError_t launch_adam_step(float *p, const float *g, float *m, float *v, uint64_t n, double b1, double b2, double eps, double lr_t,
                            hipStream_t stream, float lo, float hi)
{
    if (n == 0) return hipSuccess;
    const uint64_t n4 = n / 4;
    unsigned blocks = (unsigned) ((n4 + 255) / 256 < 16384 ? (n4 + 255) / 256 : 16384);
    if (blocks == 0) blocks = 1;
    // (1 - beta in double, THEN to float: 1.0f - 0.999f is off by 1.3e-5 relative)
    hipLaunchKernelGGL(adam_step_kernel, dim3(blocks), dim3(256), 0, stream, p, g, m, v, n, (float) b1, (float) (1.0 - b1), (float) b2,
                       (float) (1.0 - b2), (float) eps, (float) lr_t, lo, hi);
    return hipGetLastError();
}

// Which blocks of a gradient buffer hold anything but zeros (distributed.py: the compacted all-reduce).
// One float4 per lane: a wave-wide load covers 256 floats = 256 / BLOCK blocks; NaN / inf count as non-zero.
template <int BLOCK>
__global__ __launch_bounds__(256) void block_mask_kernel(const float4 *buf, uint64_t n_vec, uint8_t *mask)
{
    constexpr int kLanes = BLOCK / 4;                       // lanes per block
    const uint64_t stride = (uint64_t) gridDim.x * 256;
    for (uint64_t v = (uint64_t) blockIdx.x * 256 + threadIdx.x; v < (n_vec + 63) / 64 * 64; v += stride) {
        bool nz = false;
        if (v < n_vec) {
            const float4 f = buf[v];
            nz = !(f.x == 0.f && f.y == 0.f && f.z == 0.f && f.w == 0.f);
        }
        const uint64_t b = __ballot(nz);
        const unsigned lane = threadIdx.x & 63u;
        if (lane % kLanes == 0 && v < n_vec) {
            const uint64_t field = (b >> lane) & (kLanes == 64 ? ~0ull : ((1ull << (kLanes & 63)) - 1ull));
            mask[v / kLanes] = field ? 1 : 0;
        }
    }
}

hipError_t launch_block_mask(const float *buf, uint64_t n_blocks, uint32_t block_floats, uint8_t *mask, hipStream_t stream)
{
    if (n_blocks == 0) return hipSuccess;
    const uint64_t n_vec = n_blocks * (block_floats / 4);
    unsigned blocks = (unsigned) ((n_vec + 255) / 256 < 8192 ? (n_vec + 255) / 256 : 8192);
    const float4 *b4 = reinterpret_cast<const float4 *>(buf);
    switch (block_floats) {
    case 64:  hipLaunchKernelGGL(block_mask_kernel<64>, dim3(blocks), dim3(256), 0, stream, b4, n_vec, mask); break;
    case 128: hipLaunchKernelGGL(block_mask_kernel<128>, dim3(blocks), dim3(256), 0, stream, b4, n_vec, mask); break;
    case 256: hipLaunchKernelGGL(block_mask_kernel<256>, dim3(blocks), dim3(256), 0, stream, b4, n_vec, mask); break;
    default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

// Which blocks of the flat gradient buffer CAN be non-zero, from sigma_t alone (distributed.gradient_support: the packing set of the
// one-collective gradient all-reduce, known before the adjoint pass).  Pass 1: one bit per voxel, sigma_t != 0 (x fastest, rows padded
// to whole words).  Pass 2: one thread per block; a block inside the per-voxel plane [sparse_off, sparse_off + V * ch) covers a run of
// voxels in linear order - it is in the set iff some voxel of the run has a non-zero sigma_t voxel in its 3 x 3 x 3 neighbourhood
// (a scattering vertex has sigma_t(x) > 0 and a trilinear footprint); every other block (the dense sigma_t plane, padding) is in the set.
__global__ void __launch_bounds__(256) support_bits_kernel(const float *sigma_t, int rx, int ry, int rz, int row_words, uint32_t *bits)
{
    // one wave per 64 voxels of a row (coalesced), two words per ballot
    const size_t wave = ((size_t) blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int lane = threadIdx.x & 63;
    const int chunks = (rx + 63) / 64;
    const size_t total = (size_t) chunks * ry * rz;
    if (wave >= total) return;
    const size_t row = wave / chunks; const int x = (int) (wave % chunks) * 64 + lane;
    const bool nz = x < rx && sigma_t[row * rx + x] != 0.0f;
    const uint64_t m = __ballot(nz);
    const int w0 = (int) (wave % chunks) * 2;
    if (lane == 0) bits[row * row_words + w0] = (uint32_t) m;
    if (lane == 1 && w0 + 1 < row_words) bits[row * row_words + w0 + 1] = (uint32_t) (m >> 32);
}

__device__ __forceinline__ bool support_row_any(const uint32_t *bits, int row_words, int rx, size_t row, int xa, int xb)
{
    xa = xa < 0 ? 0 : xa; xb = xb > rx - 1 ? rx - 1 : xb;                 // voxels [xa, xb] of the row
    if (xa > xb) return false;
    const uint32_t *r = bits + row * row_words;
    for (int w = xa >> 5; w <= (xb >> 5); ++w) {
        uint32_t m = r[w];
        const int lo = w * 32;
        if (xa > lo) m &= ~0u << (xa - lo);
        if (xb < lo + 31) m &= ~0u >> (lo + 31 - xb);
        if (m) return true;
    }
    return false;
}

__global__ void __launch_bounds__(256) support_mask_kernel(const uint32_t *bits, int rx, int ry, int rz, int row_words, uint64_t sparse_off,
                                                           uint32_t ch, uint64_t n_blocks, uint32_t block_floats, uint8_t *mask)
{
    const uint64_t b = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= n_blocks) return;
    const uint64_t f0 = b * block_floats, f1 = f0 + block_floats - 1;                        // the block's floats [f0, f1]
    const uint64_t V = (uint64_t) rx * ry * rz;
    if (f0 < sparse_off || f1 >= sparse_off + V * ch) { mask[b] = 1; return; }              // (not wholly inside the per-voxel plane)
    const uint64_t v0 = (f0 - sparse_off) / ch, v1 = (f1 - sparse_off) / ch;                 // voxels [v0, v1], linear (x fastest)
    bool any = false;
    uint64_t v = v0;
    while (v <= v1 && !any) {                                                                // row by row (a block spans one or two rows)
        const uint64_t row = v / rx; const int xa = (int) (v % rx);
        const uint64_t row_end = row * rx + rx - 1;
        const int xb = (int) ((v1 < row_end ? v1 : row_end) - row * rx);
        const int y = (int) (row % ry), z = (int) (row / ry);
        for (int dz = -1; dz <= 1 && !any; ++dz)
            for (int dy = -1; dy <= 1 && !any; ++dy) {
                const int yy = y + dy, zz = z + dz;
                if (yy < 0 || yy >= ry || zz < 0 || zz >= rz) continue;
                any = support_row_any(bits, row_words, rx, (size_t) zz * ry + yy, xa - 1, xb + 1);
            }
        v = row_end + 1;
    }
    mask[b] = any ? 1 : 0;
}

hipError_t launch_support_mask(const float *sigma_t, int rx, int ry, int rz, uint64_t sparse_off, uint32_t ch, uint64_t n_blocks,
                               uint32_t block_floats, uint32_t *bits, uint8_t *mask, hipStream_t stream)
{
    if (n_blocks == 0) return hipSuccess;
    const int row_words = (rx + 31) / 32;
    const size_t words = (size_t) row_words * ry * rz;
    (void) words;
    const size_t waves = (size_t) ((rx + 63) / 64) * ry * rz;
    hipLaunchKernelGGL(support_bits_kernel, dim3((unsigned) ((waves * 64 + 255) / 256)), dim3(256), 0, stream, sigma_t, rx, ry, rz, row_words, bits);
    hipLaunchKernelGGL(support_mask_kernel, dim3((unsigned) ((n_blocks + 255) / 256)), dim3(256), 0, stream, bits, rx, ry, rz, row_words,
                       sparse_off, ch, n_blocks, block_floats, mask);
    return hipGetLastError();
}


// ---- packing of the one-collective gradient all-reduce (distributed._allreduce_flat, round 5) ------------------------------------
// The packing set is a byte per 256-byte block (gradient_support / the agreed history set).  block_positions: pos[b] = rank of block b
// among the set's blocks (-1: not in the set) - counts per group of 1024 blocks, then every workgroup sums the counts of the groups
// before its own and ranks its own 1024 bytes with wave ballots.  grad_pack: ONE pass over the flat buffer - blocks of the set are copied
// to their place in the packed buffer, the others are tested and the number of non-zero ones is added to *check (the float that rides
// at the end of the packed buffer: a non-zero sum over the ranks says the set was too small).  grad_unpack: the reverse copy.
// (torch formulation before: cumsum + searchsorted + index_select + cat + a mask pass + three mask ops, index_copy_.)
constexpr int kPosGroup = 1024;

__global__ void __launch_bounds__(256) block_count_kernel(const uint8_t *mask, uint64_t n_blocks, uint32_t *group_count)
{
    __shared__ uint32_t part[4];
    const uint64_t first = (uint64_t) blockIdx.x * kPosGroup;
    uint32_t c = 0;
    for (int k = threadIdx.x; k < kPosGroup; k += 256) c += (first + k < n_blocks && mask[first + k]) ? 1u : 0u;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) c += __shfl_down(c, off, 64);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) group_count[blockIdx.x] = part[0] + part[1] + part[2] + part[3];
}

__global__ void __launch_bounds__(256) block_pos_kernel(const uint8_t *mask, uint64_t n_blocks, const uint32_t *group_count, uint32_t n_groups,
                                                        int32_t *pos, int32_t *count_out)
{
    __shared__ uint32_t part[4], wbase[4];
    uint32_t c = 0;
    const uint32_t upto = count_out && blockIdx.x == 0 ? n_groups : blockIdx.x;            // (workgroup 0 also writes the total)
    for (uint32_t g = threadIdx.x; g < upto; g += 256) c += group_count[g];
    uint32_t mine = 0;                                                                     // groups before this one
    for (uint32_t g = threadIdx.x; g < blockIdx.x; g += 256) mine += group_count[g];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) { c += __shfl_down(c, off, 64); mine += __shfl_down(mine, off, 64); }
    if ((threadIdx.x & 63) == 0) { part[threadIdx.x >> 6] = c; wbase[threadIdx.x >> 6] = mine; }
    __syncthreads();
    if (count_out && blockIdx.x == 0 && threadIdx.x == 0) *count_out = (int32_t) (part[0] + part[1] + part[2] + part[3]);
    uint32_t base = wbase[0] + wbase[1] + wbase[2] + wbase[3];
    __syncthreads();
    const uint64_t first = (uint64_t) blockIdx.x * kPosGroup;
    const unsigned lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    for (int round = 0; round < kPosGroup / 256; ++round) {
        const uint64_t b = first + (uint64_t) round * 256 + threadIdx.x;
        const bool in = b < n_blocks && mask[b] != 0;
        const uint64_t m = __ballot(in);
        if (lane == 0) part[wave] = (uint32_t) __popcll(m);
        __syncthreads();
        uint32_t before = 0;
        for (unsigned w = 0; w < wave; ++w) before += part[w];
        const uint32_t total = part[0] + part[1] + part[2] + part[3];
        if (b < n_blocks) pos[b] = in ? (int32_t) (base + before + (uint32_t) __popcll(m & ((1ull << lane) - 1ull))) : -1;
        base += total;
        __syncthreads();
    }
}

template <int BLOCK, bool UNPACK>
__global__ void __launch_bounds__(256) grad_pack_kernel(float4 *flat, const int32_t *pos, uint64_t n_vec, float4 *packed, float *check)
{
    constexpr int kLanes = BLOCK / 4;                       // lanes (float4s) per block
    const uint64_t stride = (uint64_t) gridDim.x * 256;
    const unsigned lane = threadIdx.x & 63u;
    uint32_t outside = 0;
    for (uint64_t v = (uint64_t) blockIdx.x * 256 + threadIdx.x; v < (n_vec + 63) / 64 * 64; v += stride) {
        bool nz = false;
        if (v < n_vec) {
            const int32_t p = pos[v / kLanes];
            if constexpr (UNPACK) {
                if (p >= 0) flat[v] = packed[(uint64_t) p * kLanes + v % kLanes];
            } else {
                const float4 f = flat[v];
                if (p >= 0) packed[(uint64_t) p * kLanes + v % kLanes] = f;
                else nz = !(f.x == 0.f && f.y == 0.f && f.z == 0.f && f.w == 0.f);
            }
        }
        if constexpr (!UNPACK) {
            const uint64_t b = __ballot(nz);
            if (lane % kLanes == 0 && ((b >> lane) & (kLanes == 64 ? ~0ull : ((1ull << (kLanes & 63)) - 1ull)))) ++outside;
        }
    }
    if constexpr (!UNPACK) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) outside += __shfl_down(outside, off, 64);
        if (lane == 0 && outside) atomicAdd(check, (float) outside);
    }
}

hipError_t launch_block_positions(const uint8_t *mask, uint64_t n_blocks, int32_t *pos, int32_t *count, uint32_t *group_count, hipStream_t stream)
{
    if (n_blocks == 0) return count ? hipMemsetAsync(count, 0, sizeof(int32_t), stream) : hipSuccess;
    const uint64_t n_groups = (n_blocks + kPosGroup - 1) / kPosGroup;
    if (n_groups > 0x7fffffffull) return hipErrorInvalidValue;
    hipLaunchKernelGGL(block_count_kernel, dim3((unsigned) n_groups), dim3(256), 0, stream, mask, n_blocks, group_count);
    hipLaunchKernelGGL(block_pos_kernel, dim3((unsigned) n_groups), dim3(256), 0, stream, mask, n_blocks, group_count, (uint32_t) n_groups, pos, count);
    return hipGetLastError();
}

hipError_t launch_grad_pack(float *flat, const int32_t *pos, uint64_t n_blocks, uint32_t block_floats, float *packed, float *check, bool unpack,
                            hipStream_t stream)
{
    if (n_blocks == 0) return hipSuccess;
    const uint64_t n_vec = n_blocks * (block_floats / 4);
    unsigned blocks = (unsigned) ((n_vec + 255) / 256 < 16384 ? (n_vec + 255) / 256 : 16384);
    float4 *f4 = reinterpret_cast<float4 *>(flat), *p4 = reinterpret_cast<float4 *>(packed);
#define DRT_PACK(B) do { if (unpack) hipLaunchKernelGGL((grad_pack_kernel<B, true>), dim3(blocks), dim3(256), 0, stream, f4, pos, n_vec, p4, check); \
                         else hipLaunchKernelGGL((grad_pack_kernel<B, false>), dim3(blocks), dim3(256), 0, stream, f4, pos, n_vec, p4, check); } while (0)
    switch (block_floats) {
    case 64:  DRT_PACK(64); break;
    case 128: DRT_PACK(128); break;
    case 256: DRT_PACK(256); break;
    default: return hipErrorInvalidValue;
    }
#undef DRT_PACK
    return hipGetLastError();
}

hipError_t launch_film_backward(const float *grad_image, uint64_t n_pixels, uint32_t spp, float *dL, hipStream_t stream)
{
    uint64_t n = n_pixels * spp * 3;
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(film_backward_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, grad_image, n_pixels, spp, dL);
    return hipGetLastError();
}

uint32_t host_alt_seed(uint32_t seed, bool sensor_flow)
{
    // volpathsimple.py:99-107: alt_seed = tea32(bits(lane-0 draw), 1)[0]; lane 0's
    // draw is the 2nd of its stream (4th in the sensor flow) whatever the geometry.
    Pcg32 S; S.seed(seed, 0);
    int skip = sensor_flow ? 3 : 1;
    for (int k = 0; k < skip; ++k) (void) S.next_1d();
    float u = S.next_1d();
    uint32_t bits; __builtin_memcpy(&bits, &u, 4);
    uint32_t v0, v1;
    tea32(bits, 1u, v0, v1);
    return v0;
}

}  // namespace drt
