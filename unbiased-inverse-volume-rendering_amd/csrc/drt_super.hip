// drt_super.hip -- the tracer for scenes with a majorant supergrid (majorant_resolution_factor > 0, the reference's
// default: python/scene_config.py:36, optimize.py:182-199): VolpathSimpleIntegrator.sample
// (python/integrators/volpathsimple.py:38-655), both AD modes, as a lane-level state machine whose micro-step is ONE
// SUPERGRID CELL of a free flight.
//
// Why (measured): with a supergrid a tracking step is a 3-D DDA through 1..60 cells (mean 12.6,
// bimodal) followed by one grid lookup, and a walk is ~25 cells but only 0.7..2 lookups: 80 % of the work is cell
// stepping.  The one-ray-per-lane tracer ran it at 11.8 % VALU lane utilisation (every lane waits for the longest flight
// of the few lanes that have one), the older state machine (drt_wavefront.hip, one whole flight per step) at 18 %.
// Here every lane is a state machine with three kinds of work,
//   (D) one cell of its current flight            - the hot loop, ~30 instructions, majorants read from LDS,
//   (F) a flight boundary                          - the collision a flight ended in (grid lookup, acceptance / ratio /
//                                                    reservoir epilogue of its walk) and the set-up of the next flight,
//   (B) a path transition                          - scatter / escape / emitter sampling / end of path ...,
// and the wavefront runs ONE flat loop in which (F) and (B) are executed only when enough lanes wait for them (ballot /
// popcount thresholds) or nothing else can run, while (D) runs for whoever is in flight.  Lanes that finish a ray pull
// the next one from their XCD's queue, so a wavefront is never held hostage by its longest path.
//
// One workgroup of DRT_SUPER_THREADS threads per CU: the whole majorant supergrid (32^3 floats = 128 KiB for a 256^3
// grid at factor 8) sits in that workgroup's LDS - gfx950 has 160 KiB per CU and lets one workgroup take all of it -
// so the DDA never touches global memory.  Larger supergrids keep their non-empty-cell bitmask in LDS and load the
// majorants of non-empty cells from L2 (template flag MGL = false).
//
// Arithmetic, random-number consumption and event counts are those of the scalar restatement (oracle/drt_oracle.c):
// radiance is bit-exact per ray, counters are equal; gradients differ by summation order only.  The adjoint emits its
// splats as records (drt_deferred.hip) and reads the path cache its primal pass wrote.  Not handled here (the host
// keeps the one-ray-per-lane kernels for them): quadratic DRT (use_drt && !use_drt_subsampling), the atomic gradient
// path, supergrids with more than 1023 cells along an axis.
#include "drt_device.h"
#include "drt_launch.h"

#ifndef DRT_SUPER_THREADS
#define DRT_SUPER_THREADS 768      // primal pass: threads per workgroup = per CU (12 waves, 3 per SIMD: 168 registers; with
                                   // 1024 threads = 128 registers it spills 60 B per lane: 5.2 vs 4.0 ms)
#endif
#ifndef DRT_SUPER_THREADS_ADJ
#define DRT_SUPER_THREADS_ADJ 768  // adjoint pass: its state wants 178 registers; 768 threads (168 registers, 32 B of scratch):
                                   // 8.3 ms, 512 threads (no scratch): 9.6 ms, 1024 threads (176 B of scratch): 14.1 ms
#endif
#ifndef DRT_SUPER_K
#define DRT_SUPER_K 2              // cells per lane and loop iteration
#endif
#ifndef DRT_SUPER_FMIN
#define DRT_SUPER_FMIN 20          // lanes waiting at a flight boundary before (F) runs
#endif
#ifndef DRT_SUPER_BMIN
#define DRT_SUPER_BMIN 24          // lanes waiting at a path transition before (B) runs
#endif
#ifndef DRT_SUPER_REGEN_MIN
#define DRT_SUPER_REGEN_MIN 8      // idle lanes before the ray prologue runs
#endif
#ifndef DRT_SUPER_CHUNK
#define DRT_SUPER_CHUNK 128        // queue positions a wave reserves per refill (divides DRT_SUPER_RUN)
#endif
#ifndef DRT_SUPER_PROFILE
#define DRT_SUPER_PROFILE 0
#endif
#ifndef DRT_SUPER_RUN
#define DRT_SUPER_RUN 16384        // consecutive rays per XCD-owned run
#endif

namespace drt {

namespace {

enum Phase : int {
    // walk phases: the lane is inside a tracking walk (its flight state `fl` says where)
    PH_DT = 0, PH_RT, PH_RTA, PH_DRT,
    // transition phases
    PH_HEAD, PH_SCAT, PH_ESC, PH_TR, PH_POST, PH_NEE, PH_RT_END, PH_RTA_END, PH_PHASE, PH_END, PH_DRT_END,
    PH_IDLE, PH_DEAD
};
enum Flight : int { FL_NEW = 0, FL_FLY = 1, FL_END = 2 };   // first flight of a walk to set up | in flight | flight ended

__device__ __forceinline__ uint32_t xcc_id()
{
    return __builtin_amdgcn_s_getreg(20 | (0 << 6) | ((4 - 1) << 11)) & 7u;   // HW_REG_XCC_ID[3:0]
}

}  // namespace

template <bool ADJ, bool COUNT, bool ENV, bool MGL>
__global__ void __launch_bounds__(ADJ ? DRT_SUPER_THREADS_ADJ : DRT_SUPER_THREADS) trace_super_kernel(const Params P)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
    // LDS: [majorant grid (MGL) | non-empty-cell bitmask][empty-space bitmask of the voxel grid][record state per wave]
    const int n_cells = P.gx * P.gy * P.gz;
    const int mg_words = MGL ? n_cells : P.mocc_words;
    uint32_t *mg_lds = lds;
    uint32_t *occ_lds = lds + ((mg_words + 3) & ~3);
    uint32_t *rec_lds = occ_lds + kOccWords;
    {
        const uint32_t *src = MGL ? (const uint32_t *) P.mgrid : P.mocc;
        for (int w = threadIdx.x; w < mg_words; w += blockDim.x) mg_lds[w] = src[w];
    }
    const uint32_t *occ = nullptr;
    if (P.occ) {
        for (int w = threadIdx.x; w < P.occ_words; w += blockDim.x) occ_lds[w] = P.occ[w];
        occ = occ_lds;
    }
    uint32_t *rec = rec_lds + (threadIdx.x >> 6) * 8;           // record-stream state of this wave (emit_record)
    if ((threadIdx.x & 63) < 8) rec[threadIdx.x & 63] = 0;
    __syncthreads();

    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t xcc = xcc_id();
    const uint64_t n_runs = (P.n_rays - P.ray_first + DRT_SUPER_RUN - 1) / DRT_SUPER_RUN;
    // queue x serves the runs x, x + 8, ...; a wave starts on the queue of the XCD it runs on (L2 locality) and moves on
    // to the next ones when that one is drained: every ray is traced whatever the placement of the workgroups
    uint32_t qsel = 0, qx = xcc;
    uint64_t my_len = (n_runs > qx ? (n_runs - qx + 7) / 8 : 0) * DRT_SUPER_RUN;
    unsigned long long *queue = P.queues + qx;
    uint64_t pool_next = 0, pool_end = 0;      // wave-uniform: this wave's reserved queue positions
    uint32_t cnt[C_COUNT];
#pragma unroll
    for (int i = 0; i < C_COUNT; ++i) cnt[i] = 0;
#if DRT_SUPER_PROFILE
    // experiment build (tools/mk_variant.sh NAME -DDRT_SUPER_PROFILE=1): the counting kernels' slots hold, summed over waves,
    // [1] loop iterations, [2] (B) runs, [3] lanes waiting when (B) ran, [4] (F) runs, [5] lanes served by (F), [6] (D) runs,
    // [7] lanes in flight summed over the cell steps, [8] passes of (B)
#define DRT_COUNT(slot) do { } while (0)
#define DRT_PROF(slot, v) do { if (COUNT && lane == 0) cnt[slot] += (uint32_t) (v); } while (0)
#else
#define DRT_COUNT(slot) do { if (COUNT) cnt[slot]++; } while (0)
#define DRT_PROF(slot, v) do { } while (0)
#endif

    // uniform supergrid constants
    const int gx = P.gx, gy = P.gy, gz = P.gz;
    const float fgx = (float) gx, fgy = (float) gy, fgz = (float) gz;
    const int lin_y = gx, lin_z = gx * gy;

    // ---- per-lane state ------------------------------------------------------------------------------
    // (registers are what limits this kernel's occupancy: state that is dead in some phase carries another phase's values)
    int ph = PH_IDLE, fl = FL_NEW;
    bool rec_mode = false;          // detached recursive path of the DRT estimator (:610-655)
    bool rec_first = false;         // its first phase sample still has the :647 / :99 prologue to run
    bool escaped = false, has_scattered = false, scat_once = false;
    int depth = 0;
    uint32_t li = 0;                // ray index inside the job (L_out / dL / path-cache slot)
    // current segment: origin, direction, exit t.  After a scatter event `ro` is the scatter point and, during the NEE
    // walks, `rd` the emitter direction (the next segment's direction is drawn afterwards); during the DRT walk
    // (sample_interaction_drt) they hold the reservoir's segment, in the recursive path x' and its directions.
    V3 ro = v3(0, 0, 0), rd = v3(0, 0, 1); float si_t = kInf;
    float beta[3] = { 1, 1, 1 }, result[3] = { 0, 0, 0 }, dL[3] = { 0, 0, 0 };
    Pcg32 S; S.state = 0; S.inc = 1;
    Pcg32 A; A.state = 0; A.inc = 1;
    uint64_t Cst = 0;               // sampler clone for the NEE adjoint replay (:383)
    // walk registers (meaning depends on the phase)
    V3 wo = v3(0, 0, 0);            // DT / RT / RTA: running origin | DRT: {T, wsum, selected t}
    float wmax = 0.0f;              // DT / RT / RTA: remaining tmax | DRT: maxt
    float wt = 0.0f;                // DT: running_t, then mei.t | RT / RTA: transmittance | DRT: t
    float nt0 = 0.0f;               // NEE: exit distance along the emitter direction
    float adjsum = 0.0f;            // RTA: sum_c dL_c * contrib_c (:399,491)
    // DRTReservoir (:730-765).  After the main path: r_o = the DRT vertex x', r_si_t = sigma_t there, r_cw = the
    // coefficients of Li' in the gradient splat (w * W * adjoint)
    int r_depth = -1; float r_si_t = kInf; V3 r_o = ro, r_d = rd;
    float r_wsum[3] = { 0, 0, 0 }, r_cw[3] = { 0, 0, 0 };
    // path cache: bounce-loop iteration of this ray, steps of the current walk, cache usable for this ray
    int pc_it = 0; uint32_t pc_steps = 0; bool pc_on = false;
    // flight (DDA) registers: next crossing time and crossing-time increment per axis, linear cell index and linear
    // strides (0: parallel), steps left to the grid border (10 bits per axis), position / optical depth so far /
    // target optical depth / end of the segment.  After the flight: f_t = distance (inf: left the segment), f_acc = the
    // majorant it was sampled with.
    float tnx = kInf, tny = kInf, tnz = kInf, tdx = kInf, tdy = kInf, tdz = kInf;
    int cell = 0, sx = 0, sy = 0, sz = 0;
    uint32_t rem = 0;
    float f_t = 0.0f, f_acc = 0.0f, f_tau = 0.0f, f_tmax = 0.0f;

    for (;;) {
        DRT_PROF(1, 1);
        // ================= (A) regeneration ===========================================================
        // Ray indices come from a wave-local pool refilled DRT_SUPER_CHUNK at a time with ONE returning atomic on the
        // XCD's queue head; idle lanes wait until DRT_SUPER_REGEN_MIN of them can run the ray prologue together.
        {
            const uint64_t wmask = __ballot(ph == PH_IDLE);
            if (wmask && (__popcll(wmask) >= DRT_SUPER_REGEN_MIN || !__ballot(ph < PH_IDLE))) {
                while (pool_next >= pool_end && qsel < 8) {                      // refill (wave-uniform)
                    const int leader = __ffsll((long long) wmask) - 1;
                    unsigned long long base = 0;
                    if ((int) lane == leader) base = atomicAdd(queue, (unsigned long long) DRT_SUPER_CHUNK);
                    base = ((unsigned long long)(unsigned int) __shfl((int)(base >> 32), leader, 64) << 32)
                         | (unsigned int) __shfl((int) base, leader, 64);
                    if (base < my_len) { pool_next = base; pool_end = base + DRT_SUPER_CHUNK; }
                    else {                                                       // this queue is drained: next one
                        ++qsel;
                        qx = (xcc + qsel) & 7u;
                        my_len = (n_runs > qx ? (n_runs - qx + 7) / 8 : 0) * DRT_SUPER_RUN;
                        queue = P.queues + qx;
                    }
                }
                const bool drained = qsel >= 8;
                const uint64_t q = pool_next + (uint64_t) __popcll(wmask & ((1ull << lane) - 1ull));
                const bool take = (ph == PH_IDLE) && !drained && q < pool_end;
                if (drained && ph == PH_IDLE) ph = PH_DEAD;                      // all eight queues are empty
                pool_next += (uint64_t) __popcll(wmask);
                if (pool_next > pool_end) pool_next = pool_end;
                if (take && q < my_len) {
                    const uint64_t i = P.ray_first + ((q / DRT_SUPER_RUN) * 8 + qx) * DRT_SUPER_RUN + (q % DRT_SUPER_RUN);
                    if (i < P.n_rays) {
                        // ---- sample() prologue (:51-108) + reach_medium (:292-319) ----
                        li = (uint32_t) i;
                        const uint64_t g64 = P.chunk ? P.ray_offset + (i / P.chunk) * P.stride + (i % P.chunk) : P.ray_offset + i;
                        const uint32_t gi = (uint32_t) g64;
                        S.seed(P.seed, gi);
                        if (P.sensor_flow) {
                            float ux = S.next_1d(), uy = S.next_1d();
                            sensor_ray(P, gi / P.spp, ux, uy, ro, rd);
                        } else {
                            ro = v3(P.rays_o[3 * i], P.rays_o[3 * i + 1], P.rays_o[3 * i + 2]);
                            rd = v3(P.rays_d[3 * i], P.rays_d[3 * i + 1], P.rays_d[3 * i + 2]);
                        }
                        DRT_COUNT(C_RAYS);
                        pc_on = false; pc_it = 0;
                        if (P.path_cache_mode) {
                            // one word per ray ties the cache entries to THIS ray: explicit rays are hashed (the buffers
                            // may have been refilled between the two passes), sensor rays follow from the job signature
                            uint32_t hsh = 0x9e3779b9u ^ gi;
                            if (!P.sensor_flow) {
                                const uint32_t w[6] = { __float_as_uint(ro.x), __float_as_uint(ro.y), __float_as_uint(ro.z),
                                                        __float_as_uint(rd.x), __float_as_uint(rd.y), __float_as_uint(rd.z) };
#pragma unroll
                                for (int k = 0; k < 6; ++k) hsh = (hsh ^ w[k]) * 0x01000193u + (hsh >> 15);
                            }
                            if (!ADJ && P.path_cache_mode == 1) { P.ray_hash[i] = hsh; pc_on = true; }
                            if (ADJ && P.path_cache_mode == 2) pc_on = P.ray_hash[i] == hsh;
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

        // ================= (B) path transitions (batched) ==============================================
        {
            const uint64_t heavy = __ballot(ph >= PH_HEAD && ph < PH_IDLE);
            const uint64_t flying = __ballot(ph < PH_HEAD && fl == FL_FLY);
            if (heavy && (__popcll(heavy) >= DRT_SUPER_BMIN || !flying)) {
                // A pass takes every waiting lane to its next walk (or to the end of its ray); lanes whose walk comes
                // out of the path cache (adjoint pass) go round once more.
                DRT_PROF(2, 1); DRT_PROF(3, __popcll(heavy));
                do {
                DRT_PROF(8, 1);
                // ---- end of a path (:249-287) -----------------------------------------------------
                if (ph == PH_END) {
                    if (!ADJ || rec_mode) {                                     // envmap block, primal only
                        if (escaped && !(depth <= 0 && P.hide_emitters)) {
                            float w = 1.0f, Le[3];
                            if (P.use_nee) w = mis_weight(scat_once ? kInvFourPi : 1.0f, has_scattered ? emitter_pdf<ENV>(P, rd) : 0.0f);
                            emitter_eval<ENV>(P, rd, Le);
#pragma unroll
                            for (int k = 0; k < 3; ++k) result[k] += (beta[k] * w) * Le[k];
                        }
                    }
                    if constexpr (!ADJ) {
                        const size_t o3 = 3 * (size_t) li;
                        P.L_out[o3] = result[0]; P.L_out[o3 + 1] = result[1]; P.L_out[o3 + 2] = result[2];
                        if (P.ray_iters) P.ray_iters[li] = (uint8_t) (pc_it < 255 ? pc_it : 255);
                        ph = PH_IDLE;
                    } else {
                        if (rec_mode) {
                            // result = Li': gradient splat at x' (:577-581)
                            float alb[3];
                            eval_albedo(P, r_o, alb);                           // :578
                            DRT_COUNT(C_ALB);
                            float gs = 0.0f, ga[3];
#pragma unroll
                            for (int k = 0; k < 3; ++k) {
                                float a = r_cw[k] * result[k];
                                gs += a * alb[k];
                                ga[k] = a * r_si_t;
                            }
                            splat_scatter<true>(P, r_o, gs, ga, rec); DRT_COUNT(C_SC); DRT_COUNT(C_SC_ALB);
                            ph = PH_IDLE;
                        } else if (P.use_drt && r_depth >= 0) {                 // :249-259, DRTReservoir.get :756-760
                            const float d = ((r_cw[0] + r_cw[1]) + r_cw[2]) / 3.0f;
                            const float ws = ((r_wsum[0] + r_wsum[1]) + r_wsum[2]) / 3.0f;
#pragma unroll
                            for (int k = 0; k < 3; ++k) r_cw[k] = (d != 0.0f ? (ws * r_cw[k]) / d : 0.0f) * dL[k];   // adjoint
                            // sample_interaction_drt along the selected segment (:543-551)
                            wmax = isfinite(r_si_t) ? r_si_t : kLargest;
                            ro = r_o; rd = r_d;
                            wt = 0.0f; wo = v3(1.0f, 0.0f, kInf);               // T, wsum, selected t
                            ph = PH_DRT; fl = FL_NEW;
                        } else {
                            ph = PH_IDLE;
                        }
                    }
                }

                // ---- DRT vertex selected: enter the detached recursive path (:553-575, :610-655) -----
                if constexpr (ADJ) {
                    if (ph == PH_DRT_END) {
                        if (!(wo.z < kInf)) ph = PH_IDLE;                       // no tentative collision (:558)
                        else {
                            const V3 xp = ray_at(ro, rd, wo.z);
                            r_o = xp; ro = xp;
                            const float sig = eval_sigma_t(P, xp, occ);         // :553-554
                            r_si_t = sig;
                            DRT_COUNT(C_DRT);
                            const float w = P.use_drt_mis ? 1.0f / (1.0f + sig * sig) : 1.0f;
                            const float ww = w * wo.y;
                            r_cw[0] = ww * r_cw[0]; r_cw[1] = ww * r_cw[1]; r_cw[2] = ww * r_cw[2];
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

                // ---- NEE walk finished (:388-403) -------------------------------------------------------------
                if constexpr (!ADJ) {
                    if (ph == PH_RT_END && pc_on && pc_it < (int) P.path_cache_cap)
                        P.path_cache[((size_t) li * P.path_cache_cap + pc_it) * 2 + 1] =
                            make_uint4(__float_as_uint(wt), (uint32_t) S.state, (uint32_t) (S.state >> 32), pc_steps);
                }
                if (ph == PH_RT_END) {
                    float val[3], contrib[3];
                    const float ds_pdf = emitter_sample_value<ENV>(P, rd, val);      // recomputed from the direction
                    const float w = mis_weight(ds_pdf, kInvFourPi);             // :391
#pragma unroll
                    for (int k = 0; k < 3; ++k) {
                        contrib[k] = ((beta[k] * kInvFourPi) * w) * (val[k] * wt);
                        result[k] = (ADJ && !rec_mode) ? result[k] - contrib[k] : result[k] + contrib[k];   // :211-214
                    }
                    ph = PH_PHASE;
                    if constexpr (ADJ) {
                        if (!rec_mode) {                                        // replay with the clone (:393-401)
                            adjsum = (dL[0] * contrib[0] + dL[1] * contrib[1]) + dL[2] * contrib[2];
                            uint64_t tmp = S.state; S.state = Cst; Cst = tmp;
                            (void) S.next_1d(); (void) S.next_1d();             // same direction again (:418)
                            if (nt0 < kInf) { wo = ro; wmax = nt0; wt = 1.0f; ph = PH_RTA; fl = FL_NEW; }
                            else ph = PH_RTA_END;
                        }
                    }
                }
                if constexpr (ADJ) {
                    if (ph == PH_RTA_END) { S.state = Cst; ph = PH_PHASE; }     // back to the primary stream
                }

                // ---- phase sampling + new segment (:221-246) -------------------------------------------------------
                if (ph == PH_PHASE) {
                    ++pc_it;                                                    // next bounce-loop iteration (path cache index)
                    (void) S.next_1d();
                    float ux = S.next_1d(), uy = S.next_1d();
                    rd = square_to_uniform_sphere(ux, uy);                      // (ro is the scatter point already)
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
                    if (!active) ph = PH_END;
                    else if (ADJ && !rec_mode && pc_on && pc_it < (int) P.path_cache_cap) {
                        // the adjoint takes this iteration's delta-tracking walk from the primal pass of the same job
                        const uint4 e = P.path_cache[((size_t) li * P.path_cache_cap + pc_it) * 2];
                        wt = __uint_as_float(e.x);                              // mei.t
                        S.state = ((uint64_t) e.z << 32) | e.y;
                        if (COUNT && !DRT_SUPER_PROFILE) cnt[C_DT] += e.w;
                        ph = wt < kInf ? PH_SCAT : PH_ESC;
                    } else { wo = ro; wmax = si_t; wt = 0.0f; ph = PH_DT; fl = FL_NEW; pc_steps = 0; }
                }

                // ---- the walk found a real collision (wt = mei.t) or left the medium (:130-215, :244-245) -----------
                if constexpr (!ADJ) {                                           // path cache: what this iteration's walk returned
                    if ((ph == PH_SCAT || ph == PH_ESC) && pc_on && pc_it < (int) P.path_cache_cap)
                        P.path_cache[((size_t) li * P.path_cache_cap + pc_it) * 2] =
                            make_uint4(__float_as_uint(ph == PH_SCAT ? wt : kInf), (uint32_t) S.state, (uint32_t) (S.state >> 32), pc_steps);
                }
                if (ph == PH_SCAT || ph == PH_ESC) {
                    const bool scat = ph == PH_SCAT;
                    const bool adj_lane = ADJ && !rec_mode;
                    float albedo[3] = { 1.0f, 1.0f, 1.0f }, mei_sig = 0.0f;
                    V3 mp = ro;
                    if (scat) {
                        mp = ray_at(ro, rd, wt);                                // :371
                        if (adj_lane) { mei_sig = eval_sigma_t(P, mp, occ); DRT_COUNT(C_DT); }   // :373-375
                        has_scattered = true;
                        eval_albedo(P, mp, albedo);                             // :141
                        DRT_COUNT(C_ALB);
                    }
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
                            if (scat && (!P.use_drt || P.use_drt_mis)) {        // :152-172
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
                                splat_scatter<true>(P, mp, gs, ga, rec); DRT_COUNT(C_SC); DRT_COUNT(C_SC_ALB);
                            }
                            // backpropagate_transmittance: 4 resampled points on the segment (:181-189, :584-607)
                            const float tr_int = scat ? wt : si_t;
                            const float tr_g = -(((dL[0] * result[0] + dL[1] * result[1]) + dL[2] * result[2]) * (tr_int / 4.0f));
                            V3 pts[4];
#pragma unroll
                            for (int j = 0; j < 4; ++j) {
                                const float u = A.next_1d();                    // :595
                                pts[j] = ray_at(ro, rd, u * tr_int);
                                DRT_COUNT(C_TR);
                            }
                            if (tr_g != 0.0f) emit_records0<4>(P, pts, tr_g * P.scale, rec);
                        }
                    }
                    if (scat) {
                        beta[0] *= albedo[0]; beta[1] *= albedo[1]; beta[2] *= albedo[2];   // :193
                        depth += 1;                                             // :199
                        ro = mp;
                        if (depth < P.max_depth) ph = P.use_nee ? PH_NEE : PH_PHASE;   // :200, :206-207
                        else ph = PH_END;          // killed inside the medium; its phase draws are unobservable
                    } else {
                        escaped = true;                                         // :245
                        ph = PH_END;
                    }
                }

                // ---- emitter direction + boundary exit for NEE (:406-433) ------------------------------------------
                if (ph == PH_NEE) {
                    if (ADJ && !rec_mode) Cst = S.state;                        // :383
                    float ux = S.next_1d(), uy = S.next_1d();                   // :418
                    rd = emitter_sample_dir<ENV>(P, ux, uy);
                    Hit h = box_hit(P, ro, rd);                                 // :427-428
                    if constexpr (ENV) { if (envmap_pdf(P, rd) == 0.0f) h.valid = false; }   // sampling_worked :421-423
                    pc_steps = 0;
                    nt0 = h.valid ? h.t : kInf;
                    if (ADJ && !rec_mode && pc_on && pc_it < (int) P.path_cache_cap) {
                        // the value walk of the main path comes out of the path cache: transmittance, stream, steps
                        const uint4 e = P.path_cache[((size_t) li * P.path_cache_cap + pc_it) * 2 + 1];
                        wt = __uint_as_float(e.x);
                        S.state = ((uint64_t) e.z << 32) | e.y;
                        if (COUNT && !DRT_SUPER_PROFILE) cnt[C_RT] += e.w;
                        ph = PH_RT_END;
                    } else if (h.valid) { wo = ro; wmax = h.t; wt = 1.0f; ph = PH_RT; fl = FL_NEW; }
                    else { wt = 0.0f; ph = PH_RT_END; }
                }
                } while (__ballot(ph >= PH_HEAD && ph < PH_IDLE));
            }
        }

        // ================= (F) flight boundaries (batched) ===============================================
        // the collision the previous flight of a walk ended in (lookup + the walk's epilogue), then the next flight's
        // set-up: draw -> target optical depth, DDA state at the new origin
        {
            const bool walking = ph < PH_HEAD;
            const bool need_f = walking && fl != FL_FLY;
            const uint64_t mF = __ballot(need_f);
            const uint64_t mD = __ballot(walking && fl == FL_FLY);
            if (mF && (__popcll(mF) >= DRT_SUPER_FMIN || !mD)) {
                DRT_PROF(4, 1); DRT_PROF(5, __popcll(mF));
                if (need_f) {
                    const bool drt = ph == PH_DRT;
                    const bool useA = ADJ && !rec_mode && drt;
                    Pcg32 R; R.state = useA ? A.state : S.state; R.inc = useA ? A.inc : S.inc;
                    bool cont = true;
                    if (fl == FL_END) {
                        const float dt = f_t, lm = f_acc;
                        const float lim = lm > 0.0f ? 1.0f / lm : 0.0f;
                        bool inside; V3 p;
                        if (drt) { wt += dt; inside = wt <= wmax; p = ray_at(ro, rd, wt); }
                        else { inside = dt <= f_tmax; p = ray_at(wo, rd, dt); }
                        const float sig = inside ? eval_sigma_t(P, p, occ) : 0.0f;
                        if (!inside) {                                          // left the segment
                            ph = drt ? PH_DRT_END : (ph == PH_DT) ? PH_ESC : (ph == PH_RT ? PH_RT_END : PH_RTA_END);
                            cont = false;
                        } else if (drt) {                                       // Medium::sample_interaction_drt (:549-551); wo = {T, wsum, selected t}
                            DRT_COUNT(C_DRT);
                            const float w = wo.x * lim;
                            wo.y += w;
                            const float u2 = R.next_1d();
                            if (w > 0.0f && u2 * wo.y <= w) wo.z = wt;
                            wo.x *= (lm - sig) * lim;
                            if (wo.x == 0.0f) { ph = PH_DRT_END; cont = false; }
                        } else if (ph == PH_DT) {                               // :348-367
                            DRT_COUNT(C_DT); ++pc_steps;
                            const float r = sig * lim;
                            const float u2 = R.next_1d();
                            if (!(u2 >= r)) { wt = wt + dt; ph = PH_SCAT; cont = false; }   // mei.t
                            else { wo = p; wmax -= dt; wt += dt; }
                        } else {                                                // ratio tracking :465-502
                            DRT_COUNT(C_RT); ++pc_steps;
                            const float tr = (lm - sig) * lim;
                            if constexpr (ADJ) {
                                if (ph == PH_RTA && tr > 0.0f) {                // :487-492
                                    splat_sigma_t<true>(P, p, -(adjsum * lim) / tr, rec);
                                    DRT_COUNT(C_RT_ADJ);
                                }
                            }
                            wt *= tr; wo = p; wmax -= dt;
                            if (wt == 0.0f) { ph = (ph == PH_RT) ? PH_RT_END : PH_RTA_END; cont = false; }
                        }
                    } else {
                        // first flight of a walk: the direction's share of the DDA (Medium::sample_interaction [M3-ext];
                        // oracle: sample_collision) - crossing-time increments 1 / |dg| and the linear cell strides
                        const float dgx = (rd.x * P.inv_ext[0]) * fgx, dgy = (rd.y * P.inv_ext[1]) * fgy, dgz = (rd.z * P.inv_ext[2]) * fgz;
                        if (dgx >= 1e-20f) { tdx = 1.0f / dgx; sx = 1; } else if (dgx <= -1e-20f) { tdx = 1.0f / -dgx; sx = -1; } else { tdx = kInf; sx = 0; }
                        if (dgy >= 1e-20f) { tdy = 1.0f / dgy; sy = lin_y; } else if (dgy <= -1e-20f) { tdy = 1.0f / -dgy; sy = -lin_y; } else { tdy = kInf; sy = 0; }
                        if (dgz >= 1e-20f) { tdz = 1.0f / dgz; sz = lin_z; } else if (dgz <= -1e-20f) { tdz = 1.0f / -dgz; sz = -lin_z; } else { tdz = kInf; sz = 0; }
                    }
                    if (cont) {
                        const float u = R.next_1d();
                        f_tau = -drt_logf(1.0f - u);
                        const V3 o = drt ? ray_at(ro, rd, wt) : wo;
                        f_tmax = drt ? wmax - wt : wmax;
                        const float gxf = ((o.x - P.bmin[0]) * P.inv_ext[0]) * fgx;
                        const float gyf = ((o.y - P.bmin[1]) * P.inv_ext[1]) * fgy;
                        const float gzf = ((o.z - P.bmin[2]) * P.inv_ext[2]) * fgz;
                        const float flx = fminf(fmaxf(floorf(gxf), 0.0f), (float) (gx - 1));
                        const float fly = fminf(fmaxf(floorf(gyf), 0.0f), (float) (gy - 1));
                        const float flz = fminf(fmaxf(floorf(gzf), 0.0f), (float) (gz - 1));
                        const int cx = (int) flx, cy = (int) fly, cz = (int) flz;
                        tnx = sx > 0 ? ((flx + 1.0f) - gxf) * tdx : sx < 0 ? (gxf - flx) * tdx : kInf;
                        tny = sy > 0 ? ((fly + 1.0f) - gyf) * tdy : sy < 0 ? (gyf - fly) * tdy : kInf;
                        tnz = sz > 0 ? ((flz + 1.0f) - gzf) * tdz : sz < 0 ? (gzf - flz) * tdz : kInf;
                        const uint32_t rx_ = (uint32_t) (sx > 0 ? gx - 1 - cx : cx), ry_ = (uint32_t) (sy > 0 ? gy - 1 - cy : cy),
                                       rz_ = (uint32_t) (sz > 0 ? gz - 1 - cz : cz);
                        rem = rx_ | (ry_ << 10) | (rz_ << 20);
                        cell = (cz * gy + cy) * gx + cx;
                        f_t = 0.0f; f_acc = 0.0f;
                        fl = FL_FLY;
                    }
                    if (useA) A.state = R.state; else S.state = R.state;
                }
            }
        }

        // ================= (D) supergrid cells ============================================================
        {
            bool fly = ph < PH_HEAD && fl == FL_FLY;
            if (__ballot(fly)) {
                DRT_PROF(6, 1);
#pragma unroll
                for (int k = 0; k < DRT_SUPER_K; ++k) {
#if DRT_SUPER_PROFILE
                    DRT_PROF(7, __popcll(__ballot(fly)));
#endif
                    if (fly) {
                        const float tmin = fminf(fminf(tnx, tny), tnz);         // (crossing times are finite or +inf, never NaN)
                        const float texit = fminf(tmin, f_tmax);
                        float mc;
                        if constexpr (MGL) mc = __uint_as_float(mg_lds[cell]);
                        else mc = ((mg_lds[cell >> 5] >> (cell & 31)) & 1u) ? P.mgrid[cell] : 0.0f;
                        bool hit = false;
                        if (mc > 0.0f) {
                            const float dtau = mc * (texit - f_t);
                            if (f_acc + dtau >= f_tau) hit = true;
                            else f_acc += dtau;
                        }
                        if (hit) {                                              // the tentative collision lies in this cell
                            f_t = fmaf(f_tau - f_acc, 1.0f / mc, f_t); f_acc = mc;
                            fly = false;
                        } else {
                            f_t = texit;
                            const bool isx = tnx == tmin, isy = !isx && tny == tmin;   // first axis with the earliest crossing
                            const uint32_t sh = isx ? 0u : isy ? 10u : 20u;
                            if (!(texit < f_tmax) || ((rem >> sh) & 1023u) == 0u) {    // end of the segment / of the grid
                                f_t = kInf; f_acc = 0.0f;
                                fly = false;
                            } else {
                                rem -= 1u << sh;
                                cell += isx ? sx : isy ? sy : sz;
                                if (isx) tnx += tdx; else if (isy) tny += tdy; else tnz += tdz;
                            }
                        }
                        if (!fly) fl = FL_END;
                    }
                }
            }
        }
    }

    if constexpr (ADJ) close_records(P, rec);
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
#undef DRT_PROF
}

// LDS bytes of a launch; 0: this supergrid cannot be served (the host keeps the one-ray-per-lane kernels)
static size_t super_lds_bytes(const Params &P, bool &mgl)
{
    const size_t cells = (size_t) P.gx * P.gy * P.gz;
    const size_t fixed = ((size_t) kOccWords + (1024 / 64) * 8) * 4;
    const size_t limit = 160u * 1024u;
    mgl = ((cells + 3) & ~(size_t) 3) * 4 + fixed <= limit;
    const size_t words = mgl ? cells : (size_t) P.mocc_words;
    const size_t need = ((words + 3) & ~(size_t) 3) * 4 + fixed;
    return need <= limit ? need : 0;
}

bool super_supported(const Params &P)
{
    bool mgl;
    return P.mgrid && P.mocc && P.gx <= 1023 && P.gy <= 1023 && P.gz <= 1023 && super_lds_bytes(P, mgl) != 0;
}

hipError_t launch_trace_super(const Params &P, bool adjoint, bool count, int n_cus, hipStream_t stream)
{
    if (P.n_rays <= P.ray_first) return hipSuccess;
    bool mgl = false;
    const size_t lds = super_lds_bytes(P, mgl);
    if (!lds) return hipErrorInvalidValue;
    // one workgroup per CU when the majorants live in LDS; with the bitmask only, as many as fit
    const unsigned threads = adjoint ? DRT_SUPER_THREADS_ADJ : DRT_SUPER_THREADS;
    unsigned per_cu = 1;
    if (!mgl) { per_cu = (unsigned) ((160u * 1024u) / lds); if (per_cu > 1024 / threads) per_cu = 1024 / threads; if (per_cu < 1) per_cu = 1; }
    unsigned blocks = (unsigned) n_cus * per_cu;
    const uint64_t need = (P.n_rays - P.ray_first + 63) / 64;                  // no more waves than 64-ray groups
    const uint64_t waves_per_block = threads / 64;
    if ((need + waves_per_block - 1) / waves_per_block < blocks) blocks = (unsigned) ((need + waves_per_block - 1) / waves_per_block);
    dim3 block(threads), grid(blocks);
    const bool env = P.env_pix != nullptr;
    hipError_t e = hipSuccess;
#define DRT_SUPER_LAUNCH(A, C, E, M)                                                                              \
    do {                                                                                                          \
        auto kern = trace_super_kernel<A, C, E, M>;                                                               \
        static size_t lds_set = 0;                                                                                \
        if (lds > lds_set) {                                                                                      \
            e = hipFuncSetAttribute((const void *) kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds); \
            if (e != hipSuccess) return e;                                                                        \
            lds_set = lds;                                                                                        \
        }                                                                                                         \
        hipLaunchKernelGGL(kern, grid, block, lds, stream, P);                                                    \
    } while (0)
    const int variant = (adjoint ? 8 : 0) | (count ? 4 : 0) | (env ? 2 : 0) | (mgl ? 1 : 0);
    switch (variant) {
        case 0: DRT_SUPER_LAUNCH(false, false, false, false); break;
        case 1: DRT_SUPER_LAUNCH(false, false, false, true); break;
        case 2: DRT_SUPER_LAUNCH(false, false, true, false); break;
        case 3: DRT_SUPER_LAUNCH(false, false, true, true); break;
        case 4: DRT_SUPER_LAUNCH(false, true, false, false); break;
        case 5: DRT_SUPER_LAUNCH(false, true, false, true); break;
        case 6: DRT_SUPER_LAUNCH(false, true, true, false); break;
        case 7: DRT_SUPER_LAUNCH(false, true, true, true); break;
        case 8: DRT_SUPER_LAUNCH(true, false, false, false); break;
        case 9: DRT_SUPER_LAUNCH(true, false, false, true); break;
        case 10: DRT_SUPER_LAUNCH(true, false, true, false); break;
        case 11: DRT_SUPER_LAUNCH(true, false, true, true); break;
        case 12: DRT_SUPER_LAUNCH(true, true, false, false); break;
        case 13: DRT_SUPER_LAUNCH(true, true, false, true); break;
        case 14: DRT_SUPER_LAUNCH(true, true, true, false); break;
        default: DRT_SUPER_LAUNCH(true, true, true, true); break;
    }
#undef DRT_SUPER_LAUNCH
    return hipGetLastError();
}

}  // namespace drt
