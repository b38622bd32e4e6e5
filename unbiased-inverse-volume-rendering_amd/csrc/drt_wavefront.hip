// drt_wavefront.hip -- wave-synchronous state-machine kernel for VolpathSimpleIntegrator.sample
// (python/integrators/volpathsimple.py:38-655), both AD modes.
//
// Why: the one-ray-per-lane kernel (drt_kernels.hip) nests three divergent while-loops inside the
// bounce loop; a wavefront pays the MAX trip count of every loop over its 64 lanes and measured
// ~17 % VALU lane utilisation.  Here every lane is a small state machine and the wavefront runs ONE
// flat loop whose body is
//   (B) transitions : the irregular per-event code (scatter / escape handling, emitter direction
//                     + box exit, end of path, ...), executed only when enough lanes wait for it
//                     (ballot / popcount threshold) or nobody can step;
//   (C) step        : ONE tracking step for every lane that is inside a walk - delta tracking,
//                     ratio tracking (+ adjoint replay), the DRT sampler, the 4 transmittance
//                     resampling splats - all through the same draw -> distance -> trilinear
//                     lookup / splat code, followed by a few instructions of per-phase epilogue.
// Lanes that finish a ray pull the next one from their XCD's queue (regeneration), so a wavefront
// is never held hostage by its longest path.  The detached recursive path of the DRT estimator
// (:610-655) re-uses the same states in REC mode with the alt sampler.
// Gradient splats of the step are issued cooperatively by the whole wavefront: 8 lanes per splat,
// 8 splats per instruction, one 64-byte request each (apron scratch layout, drt_device.h).
//
// Arithmetic, random-number consumption and event counts are identical to the scalar restatement
// (oracle/drt_oracle.c): the primal radiance is bit-exact and the counters are equal; only the
// order of gradient accumulation differs.  Not supported here: quadratic DRT
// (use_drt && !use_drt_subsampling) - the host keeps trace_kernel for it.
#include "drt_device.h"
#include "drt_launch.h"

#ifndef DRT_WF_WAVES
#define DRT_WF_WAVES 4            // waves per SIMD the kernel is compiled for
#endif
#ifndef DRT_WF_HEAVY_MIN
#define DRT_WF_HEAVY_MIN 32       // run the transition blocks when at least this many lanes wait (swept 4..64)
#endif
#ifndef DRT_WF_CHUNK
#define DRT_WF_CHUNK 256           // queue positions a wave reserves per refill (divides DRT_QUEUE_RUN)
#endif
#ifndef DRT_WF_REGEN_MIN
#define DRT_WF_REGEN_MIN 8         // idle lanes needed before the ray prologue runs
#endif
#ifndef DRT_QUEUE_RUN
#define DRT_QUEUE_RUN 65536       // consecutive rays per XCD-owned run
#endif

namespace drt {

namespace {

enum Phase : int {
    // step phases (one tracking step per loop iteration)
    PH_DT = 0, PH_RT, PH_RTA, PH_DRT, PH_TR,
    // transition phases
    PH_HEAD, PH_SCAT, PH_ESC, PH_POST, PH_NEE, PH_RT_END, PH_RTA_END, PH_PHASE, PH_END, PH_DRT_END,
    PH_IDLE, PH_DEAD
};

__device__ __forceinline__ uint32_t xcc_id()
{
    return __builtin_amdgcn_s_getreg(20 | (0 << 6) | ((4 - 1) << 11)) & 7u;   // HW_REG_XCC_ID[3:0]
}

// Medium::sample_interaction free-flight distance (global majorant or supergrid DDA); identical to
// Tracer::sample_collision in drt_kernels.hip.
__device__ __forceinline__ float collide(const Params &P, const uint32_t *mocc, float maj, float inv_maj, V3 o, V3 d, float tmax, float u,
                                         float &m_out, float &im_out)
{
    if (!P.mgrid) {
        m_out = maj; im_out = inv_maj;
        if (maj == 0.0f) return kInf;
        return -drt_logf(1.0f - u) * inv_maj;
    }
    return dda_collision(P, P.mgrid, mocc, o, d, tmax, u, m_out, im_out);
}

// Whole-wave cooperative sigma_t splat: lanes with `pending` own one splat each (8 scratch indices
// + 8 values); all lanes of the wave (converged call) then retire them 8 per instruction, lane L
// adding corner L % 8 of source round * 8 + L / 8.  With the apron scratch layout the 8 corners of a
// splat share one 64-byte line: one atomic request per splat.
__device__ __forceinline__ void wave_splat_sigma(const Params &P, bool pending, V3 p, float g, uint32_t *rec)
{
    pending = pending && g != 0.0f;   // exact zeros change nothing
    const uint64_t mask = __ballot(pending);
    if (!mask) return;
    const uint32_t lane = __lane_id();
    if (pending) {
        float w[8]; int idx[8];
        make_grad_indices(P, p, idx, w);
        const float gs = g * P.scale;
        uint32_t *mine = rec + (uint32_t) __popcll(mask & ((1ull << lane) - 1ull)) * 16u;
#pragma unroll
        for (int c = 0; c < 8; ++c) { mine[c] = (uint32_t) idx[c]; mine[8 + c] = __float_as_uint(w[c] * gs); }
    }
    coop_stage_sync();
    if (!dbg(P.debug_flags, 1u)) {
        const uint32_t k = (uint32_t) __popcll(mask), c = lane & 7u;
        for (uint32_t s = lane >> 3; s < k; s += 8u) {
            const uint32_t *src = rec + s * 16u;
            atomicAdd(P.gt + src[c], __uint_as_float(src[8 + c]));
        }
    }
    coop_stage_sync();
}

}  // namespace

template <bool ADJ, bool COUNT, bool ENV>
__global__ void __launch_bounds__(256, DRT_WF_WAVES) trace_wavefront_kernel(const Params P)
{
    __shared__ uint32_t occ_lds[kOccWords];
    const uint32_t *occ = nullptr;
    if (P.occ && !dbg(P.debug_flags, 16u)) {
        for (int w = threadIdx.x; w < P.occ_words; w += blockDim.x) occ_lds[w] = P.occ[w];
        __syncthreads();
        occ = occ_lds;
    }
    const uint32_t *mocc = nullptr;
    __shared__ uint32_t mocc_lds[kOccWords];
    if (P.mgrid && P.mocc && P.mocc_words <= kOccWords && !dbg(P.debug_flags, 8388608u)) {
        for (int w = threadIdx.x; w < P.mocc_words; w += blockDim.x) mocc_lds[w] = P.mocc[w];
        __syncthreads();
        mocc = mocc_lds;
    }
    uint32_t *rec = nullptr;
    if constexpr (ADJ) {
        __shared__ uint32_t coop_rec[4 * 64 * kCoopDwords];
        rec = coop_rec + (threadIdx.x >> 6) * (64 * kCoopDwords);
    }
    const float maj = P.majorant[0], inv_maj = P.majorant[1];
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t xcc = xcc_id();
    const uint64_t n_runs = (P.n_rays + DRT_QUEUE_RUN - 1) / DRT_QUEUE_RUN;
    // queue x serves the runs x, x + 8, ...; a wave starts on the queue of the XCD it runs on (L2 locality) and, when
    // that one is drained, moves on to the next ones: every ray is traced whatever the placement of the workgroups
    // (HIP promises no block -> XCD map; another process on the device can keep whole XCDs busy)
    uint32_t qsel = 0;                         // wave-uniform: queues drained so far by this wave
    uint32_t qx = xcc;                         // current queue
    uint64_t my_len = (n_runs > qx ? (n_runs - qx + 7) / 8 : 0) * DRT_QUEUE_RUN;
    unsigned long long *queue = P.queues + qx;
    uint64_t pool_next = 0, pool_end = 0;      // wave-uniform: this wave's reserved queue positions
    uint32_t cnt[C_COUNT];
#pragma unroll
    for (int i = 0; i < C_COUNT; ++i) cnt[i] = 0;
#define DRT_COUNT(slot) do { if (COUNT) cnt[slot]++; } while (0)

    // ---- per-lane state ------------------------------------------------------------------------------
    int ph = PH_IDLE;
    bool rec_mode = false;          // detached recursive path of the DRT estimator (:610-655)
    bool rec_first = false;         // its first phase sample still has the :647 / :99 prologue to run
    bool escaped = false, has_scattered = false, scat_once = false, did_scatter = false;
    int depth = 0;
    uint64_t li = 0; uint32_t gi = 0;
    V3 ro = v3(0, 0, 0), rd = v3(0, 0, 1); float si_t = kInf;      // current segment: origin, direction, exit t
    float beta[3] = { 1, 1, 1 }, result[3] = { 0, 0, 0 }, dL[3] = { 0, 0, 0 };
    Pcg32 S; S.state = 0; S.inc = 1;
    Pcg32 A; A.state = 0; A.inc = 1;
    uint64_t Cst = 0;               // sampler clone for the NEE adjoint replay (:383)
    // walk registers (meaning depends on the phase)
    V3 wo = v3(0, 0, 0);            // DT: running origin | RT/RTA: running origin
    float wmax = 0.0f;              // DT: remaining maxt | RT/RTA: remaining tmax | DRT: maxt
    float wt = 0.0f;                // DT: running_t      | RT/RTA: transmittance  | DRT: t
    V3 nd = v3(0, 0, 1); float nt0 = 0.0f;   // NEE direction and exit distance
    float dT = 1.0f, dws = 0.0f, dsel = kInf;   // DRT: T, wsum, selected t
    float mei_t = 0.0f, mei_sig = 0.0f; V3 mp = v3(0, 0, 0);
    float albedo[3] = { 1, 1, 1 };
    float adjsum = 0.0f;            // RTA: sum_c dL_c * contrib_c (:399,491)
    float tr_int = 0.0f, tr_g = 0.0f; int tr_j = 0;   // TR: interval, gradient value, sample index
    // DRTReservoir (:730-765); after the main path: DRT vertex x' (in r_o), its sigma_t, Li' coefficients
    int r_depth = -1; float r_si_t = kInf; V3 r_o = ro, r_d = rd;
    float r_wsum[3] = { 0, 0, 0 }, r_cw[3] = { 0, 0, 0 };
    float xp_sig = 0.0f, xp_coef[3] = { 0, 0, 0 };
    // path cache (primal pass only, Params::path_cache): bounce-loop iteration of this ray, steps of the current walk
    int pc_it = 0; uint32_t pc_steps = 0; bool pc_on = false;

    for (;;) {
        // ================= (A) regeneration ===========================================================
        // Ray indices come from a wave-local pool refilled DRT_WF_CHUNK at a time with ONE returning
        // atomic on the XCD's queue head (a round trip per finished ray would serialise the wave);
        // idle lanes wait until DRT_WF_REGEN_MIN of them can run the ray prologue together.
        {
            const uint64_t wmask = __ballot(ph == PH_IDLE);
            if (wmask && (__popcll(wmask) >= DRT_WF_REGEN_MIN || !__ballot(ph < PH_IDLE))) {
                while (pool_next >= pool_end && qsel < 8) {                      // refill (wave-uniform)
                    const int leader = __ffsll((long long) wmask) - 1;
                    unsigned long long base = 0;
                    if ((int) lane == leader) base = atomicAdd(queue, (unsigned long long) DRT_WF_CHUNK);
                    base = ((unsigned long long)(unsigned int) __shfl((int)(base >> 32), leader, 64) << 32)
                         | (unsigned int) __shfl((int) base, leader, 64);
                    if (base < my_len) { pool_next = base; pool_end = base + DRT_WF_CHUNK; }
                    else {                                                       // this queue is drained: next one
                        ++qsel;
                        qx = (xcc + qsel) & 7u;
                        my_len = (n_runs > qx ? (n_runs - qx + 7) / 8 : 0) * DRT_QUEUE_RUN;
                        queue = P.queues + qx;
                    }
                }
                const bool drained = qsel >= 8;
                const uint64_t q = pool_next + (uint64_t) __popcll(wmask & ((1ull << lane) - 1ull));
                const bool take = (ph == PH_IDLE) && !drained && q < pool_end;
                if (drained && ph == PH_IDLE) ph = PH_DEAD;                      // all eight queues are empty
                pool_next += (uint64_t) __popcll(wmask);
                if (pool_next > pool_end) pool_next = pool_end;
                if (take) {
                    const uint64_t i = ((q / DRT_QUEUE_RUN) * 8 + qx) * DRT_QUEUE_RUN + (q % DRT_QUEUE_RUN);
                    if (q >= my_len) { /* tail of the reserved chunk beyond the queue: stay idle and draw again */ }
                    else if (i < P.n_rays) {
                        // ---- sample() prologue (:51-108) + reach_medium (:292-319) ----
                        li = i;
                        const uint64_t g64 = P.chunk ? P.ray_offset + (i / P.chunk) * P.stride + (i % P.chunk) : P.ray_offset + i;
                        gi = (uint32_t) g64;
                        S.seed(P.seed, gi);
                        if (P.sensor_flow) {
                            float ux = S.next_1d(), uy = S.next_1d();
                            sensor_ray(P, gi / P.spp, ux, uy, ro, rd);
                        } else {
                            ro = v3(P.rays_o[3 * i], P.rays_o[3 * i + 1], P.rays_o[3 * i + 2]);
                            rd = v3(P.rays_d[3 * i], P.rays_d[3 * i + 1], P.rays_d[3 * i + 2]);
                        }
                        DRT_COUNT(C_RAYS);
                        if constexpr (!ADJ) {
                            pc_on = P.path_cache_mode == 1; pc_it = 0;
                            if (pc_on) {                                        // see trace_coop_kernel
                                uint32_t hsh = 0x9e3779b9u ^ gi;
                                if (!P.sensor_flow) {
                                    const uint32_t w[6] = { __float_as_uint(ro.x), __float_as_uint(ro.y), __float_as_uint(ro.z),
                                                            __float_as_uint(rd.x), __float_as_uint(rd.y), __float_as_uint(rd.z) };
#pragma unroll
                                    for (int k = 0; k < 6; ++k) hsh = (hsh ^ w[k]) * 0x01000193u + (hsh >> 15);
                                }
                                P.ray_hash[i] = hsh;
                            }
                        }
                        beta[0] = beta[1] = beta[2] = 1.0f;
                        result[0] = result[1] = result[2] = 0.0f;
                        if constexpr (ADJ) {
                            dL[0] = P.dL[3 * i]; dL[1] = P.dL[3 * i + 1]; dL[2] = P.dL[3 * i + 2];
                            result[0] = P.L_in[3 * i]; result[1] = P.L_in[3 * i + 1]; result[2] = P.L_in[3 * i + 2];
                        }
                        depth = 0; escaped = false; has_scattered = false; scat_once = false;
                        rec_mode = false; rec_first = false;
                        (void) S.next_1d();                                     // :71
                        bool active = true;
                        Hit si = box_hit(P, ro, rd);
                        if (!si.valid) { escaped = true; active = false; }
                        else {
                            ro = offset_p(si, rd);
                            Hit sn = box_hit(P, ro, rd);
                            if (!sn.valid) active = false; else si_t = sn.t;
                        }
                        r_depth = -1;
                        r_wsum[0] = r_wsum[1] = r_wsum[2] = 0.0f;
                        r_cw[0] = r_cw[1] = r_cw[2] = 0.0f;
                        if (active) (void) S.next_1d();                         // :99
                        if constexpr (ADJ) A.seed(P.alt_seed, gi);              // :100-107
                        ph = active ? PH_HEAD : PH_END;
                    }
                    // (i >= n_rays inside the last run: stay idle and draw again)
                }
            }
        }
        if (!__ballot(ph != PH_DEAD)) break;

        // ================= (B) transitions (batched) ==================================================
        {
            const uint64_t heavy = __ballot(ph >= PH_HEAD && ph < PH_IDLE);
            const uint64_t steps = __ballot(ph < PH_HEAD);
            if (heavy && (__popcll(heavy) >= DRT_WF_HEAVY_MIN || !steps)) {
                const bool adj_lane = ADJ && !rec_mode;

                // ---- end of a path (:249-287) -----------------------------------------------------
                if (ph == PH_END) {
                    if (!ADJ || rec_mode) {                                     // envmap block, primal only
                        if (escaped && !(depth <= 0 && P.hide_emitters)) {
                            float w = 1.0f, Le[3];
                            // (radiance and density of the direction from the same taps of the map: emitter_eval_pdf)
                            const float e_pdf = emitter_eval_pdf<ENV>(P, rd, Le);
                            if (P.use_nee) w = mis_weight(scat_once ? kInvFourPi : 1.0f, has_scattered ? e_pdf : 0.0f);
#pragma unroll
                            for (int k = 0; k < 3; ++k) result[k] += (beta[k] * w) * Le[k];
                        }
                    }
                    if constexpr (!ADJ) {
                        P.L_out[3 * li] = result[0]; P.L_out[3 * li + 1] = result[1]; P.L_out[3 * li + 2] = result[2];
                        if (P.ray_iters) P.ray_iters[li] = (uint8_t) (pc_it < 255 ? pc_it : 255);   // sort key of the adjoint's ray schedule
                        ph = PH_IDLE;
                    } else {
                        if (rec_mode) {
                            // result = Li': gradient splat at x' (:577-581)
                            float alb[3];
                            eval_albedo(P, r_o, alb);                           // :578 (x' is kept in r_o)
                            DRT_COUNT(C_ALB);
                            float gs = 0.0f, ga[3];
#pragma unroll
                            for (int k = 0; k < 3; ++k) {
                                float a = xp_coef[k] * result[k];
                                gs += a * alb[k];
                                ga[k] = a * xp_sig;
                            }
                            splat_sigma_t(P, r_o, gs, rec); DRT_COUNT(C_SC);
                            splat_albedo(P, r_o, ga, rec);  DRT_COUNT(C_SC_ALB);
                            ph = PH_IDLE;
                        } else if (P.use_drt && r_depth >= 0) {                 // :249-259, DRTReservoir.get :756-760
                            float d = ((r_cw[0] + r_cw[1]) + r_cw[2]) / 3.0f;
                            float ws = ((r_wsum[0] + r_wsum[1]) + r_wsum[2]) / 3.0f;
#pragma unroll
                            for (int k = 0; k < 3; ++k) xp_coef[k] = (d != 0.0f ? (ws * r_cw[k]) / d : 0.0f) * dL[k];   // adjoint
                            // sample_interaction_drt along the selected segment (:543-551)
                            wmax = isfinite(r_si_t) ? r_si_t : kLargest;
                            wt = 0.0f; dT = 1.0f; dws = 0.0f; dsel = kInf;
                            ph = PH_DRT;
                        } else {
                            ph = PH_IDLE;
                        }
                    }
                }

                // ---- DRT vertex selected: enter the detached recursive path (:553-575, :610-655) -----
                if constexpr (ADJ) {
                    if (ph == PH_DRT_END) {
                        if (!(dsel < kInf)) ph = PH_IDLE;                       // no tentative collision (:558)
                        else {
                            mp = ray_at(r_o, r_d, dsel);
                            r_o = mp;           // x' lives in r_o from here on (mp is reused by the recursive path)
                            xp_sig = eval_sigma_t(P, mp, occ);                  // :553-554
                            DRT_COUNT(C_DRT);
                            float w = P.use_drt_mis ? 1.0f / (1.0f + xp_sig * xp_sig) : 1.0f;
                            float ww = w * dws;
                            xp_coef[0] = ww * xp_coef[0]; xp_coef[1] = ww * xp_coef[1]; xp_coef[2] = ww * xp_coef[2];
                            S = A;                                              // the recursion samples with alt_sampler
                            rec_mode = true; rec_first = true;
                            result[0] = result[1] = result[2] = 0.0f;
                            beta[0] = beta[1] = beta[2] = 1.0f;
                            depth = r_depth + 1;
                            escaped = false; scat_once = true; has_scattered = false;
                            ph = P.use_nee ? PH_NEE : PH_PHASE;                 // :621-624 NEE at x' whatever the depth
                        }
                    }
                }

                // ---- real collision found (:130-189) ----------------------------------------------------
                if constexpr (!ADJ) {                                           // path cache: what this iteration's walk returned
                    if ((ph == PH_SCAT || ph == PH_ESC) && pc_on && pc_it < (int) P.path_cache_cap)
                        P.path_cache[((size_t) li * P.path_cache_cap + pc_it) * 2] =
                            make_uint4(__float_as_uint(ph == PH_SCAT ? mei_t : kInf), (uint32_t) S.state, (uint32_t) (S.state >> 32), pc_steps);
                }
                if (ph == PH_SCAT) {
                    mp = ray_at(ro, rd, mei_t);                                 // :371
                    if (adj_lane) { mei_sig = eval_sigma_t(P, mp, occ); DRT_COUNT(C_DT); }   // :373-375
                    has_scattered = true;
                    did_scatter = true;
                    eval_albedo(P, mp, albedo);                                 // :141
                    DRT_COUNT(C_ALB);
                    ph = PH_POST;
                    if constexpr (ADJ) {
                        if (adj_lane) {
                            if (P.use_drt) {                                    // DRTReservoir.update :745-753
                                float u = A.next_1d();
                                float m = 0.0f;
#pragma unroll
                                for (int k = 0; k < 3; ++k) { r_wsum[k] += beta[k]; m += beta[k] / r_wsum[k]; }
                                m = m / 3.0f;
                                if (u <= m) {
                                    r_cw[0] = beta[0]; r_cw[1] = beta[1]; r_cw[2] = beta[2];
                                    r_depth = depth; r_si_t = si_t; r_o = ro; r_d = rd;
                                }
                            }
                            if (!P.use_drt || P.use_drt_mis) {                  // :152-172
                                float w = 1.0f;
                                if (P.use_drt && P.use_drt_mis) { float s2 = mei_sig * mei_sig; w = s2 / (1.0f + s2); }
                                float inv_pdf = 1.0f / mei_sig;
                                float gs = 0.0f, ga[3];
#pragma unroll
                                for (int k = 0; k < 3; ++k) {
                                    float Li = result[k] / fmaxf(1e-8f, albedo[k]);
                                    float a = ((w * dL[k]) * Li) * inv_pdf;
                                    gs += a * albedo[k];
                                    ga[k] = a * mei_sig;
                                }
                                splat_sigma_t(P, mp, gs, rec); DRT_COUNT(C_SC);
                                splat_albedo(P, mp, ga, rec);  DRT_COUNT(C_SC_ALB);
                            }
                            tr_int = mei_t;                                     // :181-189, :584-607
                            tr_g = -(((dL[0] * result[0] + dL[1] * result[1]) + dL[2] * result[2]) * (tr_int / 4.0f));
                            tr_j = 0;
                            ph = PH_TR;
                        }
                    }
                }

                // ---- escaped the medium (:130, :143-150, :181-189) -----------------------------------------
                if (ph == PH_ESC) {
                    did_scatter = false;
                    albedo[0] = albedo[1] = albedo[2] = 1.0f;
                    ph = PH_POST;
                    if constexpr (ADJ) {
                        if (adj_lane) {
                            if (P.use_drt) {
                                float u = A.next_1d();
                                float m = 0.0f;
#pragma unroll
                                for (int k = 0; k < 3; ++k) { r_wsum[k] += beta[k]; m += beta[k] / r_wsum[k]; }
                                m = m / 3.0f;
                                if (u <= m) {
                                    r_cw[0] = beta[0]; r_cw[1] = beta[1]; r_cw[2] = beta[2];
                                    r_depth = depth; r_si_t = si_t; r_o = ro; r_d = rd;
                                }
                            }
                            tr_int = si_t;
                            tr_g = -(((dL[0] * result[0] + dL[1] * result[1]) + dL[2] * result[2]) * (tr_int / 4.0f));
                            tr_j = 0;
                            ph = PH_TR;
                        }
                    }
                }

                // ---- after the scatter / escape bookkeeping (:193-215, :244-245) -------------------------------
                if (ph == PH_POST) {
                    if (did_scatter) {
                        beta[0] *= albedo[0]; beta[1] *= albedo[1]; beta[2] *= albedo[2];   // :193
                        depth += 1;                                             // :199
                        if (depth < P.max_depth) ph = P.use_nee ? PH_NEE : PH_PHASE;   // :200, :206-207
                        else ph = PH_END;          // killed inside the medium; its phase draws are unobservable
                    } else {
                        escaped = true;                                         // :245
                        ph = PH_END;
                    }
                }

                // ---- NEE walk finished (:388-403) -------------------------------------------------------------
                if constexpr (!ADJ) {
                    if (ph == PH_RT_END && pc_on && pc_it < (int) P.path_cache_cap)
                        P.path_cache[((size_t) li * P.path_cache_cap + pc_it) * 2 + 1] =
                            make_uint4(__float_as_uint(wt), (uint32_t) S.state, (uint32_t) (S.state >> 32), pc_steps);
                }
                if (ph == PH_RT_END) {
                    float val[3], contrib[3];
                    const float ds_pdf = emitter_sample_value<ENV>(P, nd, val);      // recomputed from the direction
                    const float w = mis_weight(ds_pdf, kInvFourPi);             // :391
#pragma unroll
                    for (int k = 0; k < 3; ++k) {
                        contrib[k] = ((beta[k] * kInvFourPi) * w) * (val[k] * wt);
                        result[k] = adj_lane ? result[k] - contrib[k] : result[k] + contrib[k];   // :211-214
                    }
                    ph = PH_PHASE;
                    if constexpr (ADJ) {
                        if (adj_lane) {                                         // replay with the clone (:393-401)
                            adjsum = (dL[0] * contrib[0] + dL[1] * contrib[1]) + dL[2] * contrib[2];
                            uint64_t tmp = S.state; S.state = Cst; Cst = tmp;
                            (void) S.next_1d(); (void) S.next_1d();             // same direction again (:418)
                            if (nt0 < kInf) { wo = mp; wmax = nt0; wt = 1.0f; ph = PH_RTA; }
                            else ph = PH_RTA_END;
                        }
                    }
                }
                if constexpr (ADJ) {
                    if (ph == PH_RTA_END) { S.state = Cst; ph = PH_PHASE; }     // back to the primary stream
                }

                // ---- emitter direction + boundary exit for NEE (:406-433) ------------------------------------------
                if (ph == PH_NEE) {
                    if (adj_lane) Cst = S.state;                                // :383
                    float ux = S.next_1d(), uy = S.next_1d();                   // :418
                    nd = emitter_sample_dir<ENV>(P, ux, uy);
                    Hit h = box_hit(P, mp, nd);                                 // :427-428
                    if constexpr (ENV) { if (envmap_pdf(P, nd) == 0.0f) h.valid = false; }   // sampling_worked :421-423
                    pc_steps = 0;
                    if (h.valid) { nt0 = h.t; wo = mp; wmax = h.t; wt = 1.0f; ph = PH_RT; }
                    else { nt0 = kInf; wt = 0.0f; ph = PH_RT_END; }
                }

                // ---- phase sampling + new segment (:221-246) -------------------------------------------------------
                if (ph == PH_PHASE) {
                    ++pc_it;                                                    // next bounce-loop iteration (path cache index)
                    (void) S.next_1d();
                    float ux = S.next_1d(), uy = S.next_1d();
                    ro = mp; rd = square_to_uniform_sphere(ux, uy);
                    scat_once = true;
                    Hit h = box_hit(P, ro, rd);                                 // :233-235
                    si_t = h.valid ? h.t : kLargest;
                    bool active = h.valid;                                      // :240-241 accidental escape
                    if (rec_first) {                                            // sample_recursive -> sample() (:641-651)
                        rec_first = false;
                        active = active && (depth < P.max_depth);               // :647 (+ DESIGN.md deviation)
                        has_scattered = active;                                 // :84-85
                        if (active) (void) S.next_1d();                         // :99 of the recursive sample()
                    }
                    ph = active ? PH_HEAD : PH_END;
                }

                // ---- loop head: Russian roulette, start delta tracking (:116-127) -------------------------------------
                if (ph == PH_HEAD) {
                    float q = fminf(fmaxf(beta[0], fmaxf(beta[1], beta[2])), 0.99f);
                    bool perform_rr = depth > P.rr_depth;
                    float u_rr = S.next_1d();
                    bool active = (beta[0] != 0.0f || beta[1] != 0.0f || beta[2] != 0.0f) && (!perform_rr || (u_rr < q));
                    if (perform_rr) { float iq = 1.0f / q; beta[0] *= iq; beta[1] *= iq; beta[2] *= iq; }
                    if (active) { wo = ro; wmax = si_t; wt = 0.0f; ph = PH_DT; pc_steps = 0; }
                    else ph = PH_END;
                }
            }
        }

        // ================= (C) one tracking step ==========================================================
        {
            const bool stepping = ph < PH_HEAD;
            if (__ballot(stepping)) {
                bool splat = false; V3 sp = v3(0, 0, 0); float sg = 0.0f;
                if (stepping) {
                    const bool useA = ADJ && !rec_mode && (ph == PH_DRT || ph == PH_TR);
                    Pcg32 R; R.state = useA ? A.state : S.state; R.inc = useA ? A.inc : S.inc;
                    const float u = R.next_1d();
                    if (ph == PH_TR) {                                          // :594-607
                        sp = ray_at(ro, rd, u * tr_int); sg = tr_g; splat = true;
                        DRT_COUNT(C_TR);
                        if (++tr_j == 4) ph = PH_POST;
                    }
                    // every other step phase: free-flight distance -> ONE shared trilinear lookup -> epilogue
                    const bool walking = ph < PH_TR && !splat;
                    const bool drt = ph == PH_DRT;
                    float lm = maj, lim = inv_maj, dt = kInf, sig = 0.0f;
                    V3 p = v3(0, 0, 0);
                    bool inside = false;
                    if (walking) {
                        const V3 o = drt ? ray_at(r_o, r_d, wt) : wo;
                        const V3 d = drt ? r_d : (ph == PH_DT ? rd : nd);
                        const float tmax = drt ? wmax - wt : wmax;
                        if (drt && !P.mgrid) dt = (maj == 0.0f) ? kInf : -drt_logf(1.0f - u) * inv_maj;
                        else dt = collide(P, mocc, maj, inv_maj, o, d, tmax, u, lm, lim);
                        if (drt) { wt += dt; inside = wt <= wmax; p = ray_at(r_o, r_d, wt); }
                        else { inside = dt <= tmax; p = ray_at(o, d, dt); }
                        if (inside) sig = eval_sigma_t(P, p, occ);
                    }
                    if (walking && !inside) {                                   // left the segment
                        ph = drt ? PH_DRT_END : (ph == PH_DT) ? PH_ESC : (ph == PH_RT ? PH_RT_END : PH_RTA_END);
                    } else if (walking) {
                        if (drt) {                                              // Medium::sample_interaction_drt (:549-551)
                            DRT_COUNT(C_DRT);
                            float w = dT * lim;
                            dws += w;
                            float u2 = R.next_1d();
                            if (w > 0.0f && u2 * dws <= w) dsel = wt;
                            dT *= (lm - sig) * lim;
                            if (dT == 0.0f) ph = PH_DRT_END;
                        } else if (ph == PH_DT) {                               // :348-367
                            DRT_COUNT(C_DT); ++pc_steps;
                            float r = sig * lim;
                            float u2 = R.next_1d();
                            if (!(u2 >= r)) { mei_t = wt + dt; ph = PH_SCAT; }
                            else { wo = p; wmax -= dt; wt += dt; }
                        } else {                                                // ratio tracking :465-502
                            DRT_COUNT(C_RT); ++pc_steps;
                            float tr = (lm - sig) * lim;
                            if (ph == PH_RTA && tr > 0.0f) {                    // :487-492
                                sp = p; sg = -(adjsum * lim) / tr; splat = true;
                                DRT_COUNT(C_RT_ADJ);
                            }
                            wt *= tr; wo = p; wmax -= dt;
                            if (wt == 0.0f) ph = (ph == PH_RT) ? PH_RT_END : PH_RTA_END;
                        }
                    }
                    if (useA) A.state = R.state; else S.state = R.state;
                }
                if constexpr (ADJ) wave_splat_sigma(P, splat, sp, sg, rec);
            }
        }
    }

    if (COUNT) {
#pragma unroll
        for (int s = 0; s < C_COUNT; ++s) {
            uint32_t v = cnt[s];
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
            if (lane == 0 && v) atomicAdd(P.counters + s, (unsigned long long) v);
        }
    }
#undef DRT_COUNT
}

hipError_t launch_trace_wavefront(const Params &P, bool adjoint, bool count, int n_cus, hipStream_t stream)
{
    if (P.n_rays == 0) return hipSuccess;
    unsigned blocks = (unsigned) n_cus * DRT_WF_WAVES;      // WAVES per SIMD x 4 SIMDs = that many 256-thread groups per CU
    uint64_t need = (P.n_rays + 255) / 256;
    if (need < blocks) blocks = (unsigned) need;
    dim3 block(256), grid(blocks);
    const int variant = (adjoint ? 4 : 0) | (count ? 2 : 0) | (P.env_pix ? 1 : 0);
    switch (variant) {
        case 0: hipLaunchKernelGGL((trace_wavefront_kernel<false, false, false>), grid, block, 0, stream, P); break;
        case 1: hipLaunchKernelGGL((trace_wavefront_kernel<false, false, true>), grid, block, 0, stream, P); break;
        case 2: hipLaunchKernelGGL((trace_wavefront_kernel<false, true, false>), grid, block, 0, stream, P); break;
        case 3: hipLaunchKernelGGL((trace_wavefront_kernel<false, true, true>), grid, block, 0, stream, P); break;
        case 4: hipLaunchKernelGGL((trace_wavefront_kernel<true, false, false>), grid, block, 0, stream, P); break;
        case 5: hipLaunchKernelGGL((trace_wavefront_kernel<true, false, true>), grid, block, 0, stream, P); break;
        case 6: hipLaunchKernelGGL((trace_wavefront_kernel<true, true, false>), grid, block, 0, stream, P); break;
        default: hipLaunchKernelGGL((trace_wavefront_kernel<true, true, true>), grid, block, 0, stream, P); break;
    }
    return hipGetLastError();
}

}  // namespace drt
