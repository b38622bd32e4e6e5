// drt_coop_tracer.h -- wave-cooperative tracking loops for the one-ray-per-lane tracer (CoopTracer).
// Included by drt_coop.hip / drt_coop_super.hip (the volpathsimple kernels of global-majorant scenes and the supergrid fallback).
//
// Measured on the headline workload (wave-level vs lane-level iteration counts): 85 % of
// the per-lane adjoint kernel's VALU instructions are the delta- and ratio-tracking step loops, executed with
// 10 and 6 active lanes out of 64 - every lane waits for the longest walk of its wave.  The steps of ONE
// walk are almost independent, though: with a global majorant the free-flight distances depend only on the
// ray's PCG32 stream (an LCG: jump-ahead by k is one multiply-add with tabulated constants), so the
// tentative collisions k, k+1, ... of a walk can be generated and looked up by different lanes at once.
//
// Here every tracking loop is entered by ALL 64 lanes of the wave (the bounce loop is wave-uniform, lanes
// without a live ray carry job = false) and executed in rounds: the J pending walks share the wave,
// m = min(DRT_COOP_MAXM, 2^floor(log2(64 / J))) candidate steps each; lane L serves walk L / m, step L % m.
// What the sequential algorithm computes in order - the moving origin o_{k+1} = fma(d, dt_k, o_k), the
// remaining length, the running transmittance product, "first accepted / first outside" - every lane
// recomputes for its own step by replaying the earlier steps of the round in the SAME order with the SAME
// operations (their distances / transmittance factors gathered from the lanes that produced them), so every
// walk returns bit for bit what the per-lane loop (and the oracle) returns; only the grid lookups, the
// logarithms and the random numbers, i.e. the expensive part, run in parallel.  Speculative steps behind a
// walk's end are discarded (and not counted).  Supergrid scenes (majorant_resolution_factor > 0: the
// distance depends on the position) take every step on the walk's own lane (template flag SUPER).
//
// Headline workload, adjoint tracer: 44.7 M wave-level loop iterations -> 16.4 M rounds, VALU lane
// utilisation 15 % -> 64 %, 8.4 G -> 6.1 G wave instructions (still VALU-issue-bound: the replay and the
// round set-up are pure overhead), 13.9 -> 11.0 ms.
//
// Same algorithm and line references as Tracer in drt_kernels.hip (volpathsimple.py:38-655).
#pragma once
#include "drt_device.h"
#include "drt_launch.h"

#ifndef DRT_COOP_WAVES
#define DRT_COOP_WAVES 4
#endif
#ifndef DRT_COOP_WAVES_PRIMAL
#define DRT_COOP_WAVES_PRIMAL 5    // the primal kernels need fewer registers: 5 waves per SIMD (2.76 -> 2.59 ms; the adjoint is slower at 5: 7.15 vs 6.17 ms)
#endif
#ifndef DRT_XCD_RUN
#define DRT_XCD_RUN 256
#endif
#ifndef DRT_WGC_DONATE
#define DRT_WGC_DONATE 16          // live recursive paths at or below which a wave hands them to wave 0 (wg_handoff; swept 8 / 16 / 21 / 32 / 48: 6.36 / 6.17 / 6.17 / 6.18 / 6.44 ms)
#endif
#ifndef DRT_TAIL_PUSH
#define DRT_TAIL_PUSH 24           // live recursive paths at or below which a workgroup's last wave sends them to the tail pool (4 / 8 / 16 / 24 / 32 / 64: 6.13 / 5.96 / 5.85 / 5.75 / 5.76 / 5.73 ms)
#endif
#ifndef DRT_COOP_MAXM
#define DRT_COOP_MAXM 8        // candidate steps per walk and round = chain length (swept 4 / 8 / 16: 11.7 / 11.0 / 11.5 ms)
#endif

namespace drt {

namespace coop {

struct Ray { V3 o, d; float maxt; };
struct Mei { bool valid; float t; V3 p; float sigma_t; };
// what the end of a recursive DRT path needs for its gradient splat (backpropagate_scattering_drt, volpathsimple.py:565-581):
// the reservoir vertex, sigma_t / albedo there, w * W * adjoint, and the NEE term of Li - it travels with the path when
// the path moves to another wave (CoopTracer::wg_handoff)
struct Tail { V3 p; float sig, alb[3], wadj[3], nee[3]; };
struct PathState { int depth; Hit si; float last_pdf; bool escaped; bool active; const Tail *tail; };
// workgroup hand-off of sparse recursive paths: waves 1..3 give their last <= kWgcDonate live paths to wave 0 through LDS
constexpr int kWgcDonate = DRT_WGC_DONATE, kWgcFields = 32;
constexpr int kWgcWords = 4 + 3 * kWgcFields * kWgcDonate;     // flags[3] (+1 pad) + three donor regions

// PCG32 jump-ahead: s_{n+k} = A_k s_n + G_k inc,  A_k = a^k,  G_k = 1 + a + ... + a^(k-1)  (mod 2^64)
constexpr int kJumpMax = 2 * DRT_COOP_MAXM + 2;
struct JumpTable { uint64_t A[kJumpMax + 1], G[kJumpMax + 1]; };
constexpr JumpTable make_jump_table()
{
    JumpTable t{};
    uint64_t a = 1, g = 0;
    for (int k = 0; k <= kJumpMax; ++k) { t.A[k] = a; t.G[k] = g; g = g * 0x5851f42d4c957f2dull + 1ull; a = a * 0x5851f42d4c957f2dull; }
    return t;
}
static __device__ const JumpTable kJump = make_jump_table();

constexpr uint64_t kPcgMul = 0x5851f42d4c957f2dull;

// walks pending in a wave from which a tracking round is taken on the walks' own lanes (33: exactly the m = 1 rounds)
#ifndef DRT_PHASE_PROFILE
#define DRT_PHASE_PROFILE 0
#endif
#ifndef DRT_COOP_SOLO_MIN
#define DRT_COOP_SOLO_MIN 33
#endif

// output function of the draw whose pre-advance state is `old` (Pcg32::next_u32 / next_1d)
__device__ __forceinline__ float pcg_float(uint64_t old)
{
    uint32_t xs = (uint32_t)(((old >> 18) ^ old) >> 27);
    uint32_t rot = (uint32_t)(old >> 59);
    uint32_t bits = (((xs >> rot) | (xs << ((0u - rot) & 31u))) >> 9) | 0x3f800000u;
    return __uint_as_float(bits) - 1.0f;
}

__device__ __forceinline__ uint64_t shfl64(uint64_t v, int src)
{
    uint32_t lo = (uint32_t) __shfl((int) (uint32_t) v, src), hi = (uint32_t) __shfl((int) (uint32_t)(v >> 32), src);
    return ((uint64_t) hi << 32) | lo;
}

// SPEC: the registered `volpathsimple-drt` estimator (use_nee, use_drt, use_drt_subsampling all on; use_drt_mis stays a
// runtime flag) as compile-time constants: the in-loop quadratic DRT branch - a second inlined copy of the whole
// recursive path - and the flag tests disappear from the production kernels (smaller code, fewer live registers
// across the bounce loop).  Every other configuration runs the generic instantiation.
// G4: sigma_t and albedo at scatter points come from ONE lookup into the interleaved four-channel apron-brick copy
// (Params::grid4, eval4) instead of two lookups into two layouts - same voxel values, same interpolation arithmetic.
// SUPER: majorant supergrid (majorant_resolution_factor > 0).  The free-flight distance then depends on the position
// (Medium::sample_interaction walks the supergrid cells, dda_collision in drt_device.h), the steps of a walk are no
// longer independent, and every tracking step is taken on the walk's own lane - the same loops, path cache, record
// streams and estimator specialisation otherwise, so that ONE tracer serves every configuration.
template <bool COUNT, bool ENV, bool DEFER, bool SPEC = false, bool G4 = false, bool SUPER = false>
struct CoopTracer {
    const Params &P;
    float maj, inv_maj;
    const uint32_t *mocc;   // SUPER: non-empty supergrid cells (LDS copy) or nullptr
    uint64_t i_block;       // first ray of this workgroup (hand-off of main paths: home ray = i_block + home)
    bool tail_load;         // tail-mode launch: this lane still has to fetch its path from the tail pool
    uint32_t *wgc;          // LDS area of the workgroup hand-off (wg_handoff; kWgcWords words, flags preset to ~0) or nullptr
    uint32_t ray_index;
    uint32_t *rec;          // wave-private LDS: cooperative-scatter staging area or (DEFER) record-stream state
    uint32_t *slots;        // wave-private LDS, 64 words: walk slot -> owner lane
    const uint64_t *jump;   // LDS copy of the jump-ahead table: A_k at [2k], G_k at [2k + 1]
    const uint32_t *occ;
    uint4 *pc;              // this ray's path-cache entries (2 x uint4 per bounce-loop iteration) or nullptr
    uint32_t work;          // tracking steps of this ray's main path (primal pass: feeds block_cost)
    uint32_t iters;         // bounce-loop iterations of this ray's main path (primal pass: feeds ray_perm)
    uint32_t cnt[C_COUNT];

    __device__ __forceinline__ CoopTracer(const Params &p) : P(p)
    {
        maj = p.majorant[0]; inv_maj = p.majorant[1];
        ray_index = 0; rec = nullptr; slots = nullptr; occ = nullptr; jump = nullptr; pc = nullptr; work = 0; iters = 0;
        mocc = nullptr; wgc = nullptr; i_block = 0; tail_load = false;
#if DRT_PHASE_PROFILE
        ph_t = __builtin_readcyclecounter();
#endif
#pragma unroll
        for (int i = 0; i < C_COUNT; ++i) cnt[i] = 0;
    }
#if DRT_PHASE_PROFILE
    // experiment build (tools/phase_profile.py): the counter slots hold shader cycles per phase of the tracer instead of
    // event counts (the clock is wave-uniform, every lane adds the same value: the flushed sum / 64 = cycles per wave)
    uint64_t ph_t;
    __device__ __forceinline__ void count(int) {}
    __device__ __forceinline__ void stamp(int slot)
    {
        if (COUNT) { const uint64_t t = __builtin_readcyclecounter(); cnt[slot] += (uint32_t) ((t - ph_t) >> 4); ph_t = t; }
    }
    __device__ __forceinline__ void phase(int slot) { if (DRT_PHASE_PROFILE == 1) stamp(slot); }
    // modes 2 / 3: cycles of the coop_rt / coop_dt rounds by the number of pending walks (everything else: n_tr)
    __device__ __forceinline__ void round_begin(int which) { if (DRT_PHASE_PROFILE == which) stamp(C_TR); }
    __device__ __forceinline__ void round_end(int which, int J)
    {
        if (DRT_PHASE_PROFILE == which)
            stamp(J >= 33 ? C_DT : J >= 17 ? C_RT : J >= 9 ? C_DRT : J >= 5 ? C_ALB : J >= 3 ? C_RT_ADJ : J == 2 ? C_SC : C_SC_ALB);
    }
#else
    __device__ __forceinline__ void round_begin(int) {}
    __device__ __forceinline__ void round_end(int, int) {}
    __device__ __forceinline__ void count(int slot) { if (COUNT) cnt[slot]++; }
    __device__ __forceinline__ void phase(int) {}
#endif
    __device__ __forceinline__ bool use_nee() const { return SPEC ? true : P.use_nee != 0; }
    __device__ __forceinline__ bool use_drt() const { return SPEC ? true : P.use_drt != 0; }
    __device__ __forceinline__ bool use_sub() const { return SPEC ? true : P.use_drt_subsampling != 0; }

    __device__ __forceinline__ uint64_t pcg_jump(uint64_t state, uint64_t inc, int k) const
    {
        typedef __attribute__((address_space(3))) const uint64_t lds_u64;
        lds_u64 *t = (lds_u64 *) jump;
        return t[2 * k] * state + t[2 * k + 1] * inc;
    }

    __device__ __forceinline__ float sample_distance(float u) const
    {
        if (maj == 0.0f) return kInf;
        return -drt_logf(1.0f - u) * inv_maj;
    }

    // own-lane steps for crowded rounds only in the specialised kernels: in the generic ones (two inlined copies of the
    // recursive path) the extra code spills (scratch 324 -> 944 B per lane, adjoint 8.0 -> 17.9 ms)
    static constexpr bool kSolo = SPEC;

    // one free flight from o along d: distance to the tentative collision and the majorant it was sampled with
    // (global majorant: bit-identical to sample_distance; supergrid: Tracer::sample_collision)
    __device__ __forceinline__ float flight(V3 o, V3 d, float tmax, float u, float &lm, float &lim) const
    {
        if constexpr (SUPER) return dda_collision(P, P.mgrid, mocc, o, d, tmax, u, lm, lim);
        else { lm = maj; lim = inv_maj; return sample_distance(u); }
    }

    // round geometry shared by the cooperative loops: J pending walks -> m steps each, owner lane of my slot
    __device__ __forceinline__ void round_setup(bool job, uint64_t pending, int &m, int &lg, int &js, int &c, bool &serve, int &owner, int &my_rank)
    {
        const uint32_t lane = __lane_id();
        const int J = __popcll(pending);
        lg = J > 1 ? __clz(J - 1) - 26 : 6;                 // floor(log2(64 / J)) = 6 - ceil(log2(J)), no division
        if (lg > (31 - __clz(DRT_COOP_MAXM))) lg = 31 - __clz(DRT_COOP_MAXM);
        m = 1 << lg;
        my_rank = (int) __popcll(pending & ((1ull << lane) - 1ull));
        lds_u32 *sl = (lds_u32 *) slots;
        if (job) sl[my_rank] = lane;
        coop_stage_sync();
        js = (int) (lane >> lg); c = (int) (lane & (uint32_t)(m - 1));
        serve = js < J;
        owner = serve ? (int) sl[js] : (int) lane;
        coop_stage_sync();
    }

    // ---------------------------------------------------------------------------------------------
    // estimate_transmittance: ratio tracking (volpathsimple.py:436-504), all lanes of the wave call it
    // ---------------------------------------------------------------------------------------------
    template <bool ADJ>
    __device__ __forceinline__ float coop_rt(bool job, V3 o, V3 d, float tmax, Pcg32 &S, float a_sum, uint32_t *steps_out = nullptr)
    {
        float T = 1.0f;
        uint32_t steps = 0;
        uint64_t pending = __ballot(job);
        while (pending) {
            const int Jprof = __popcll(pending); (void) Jprof;
            round_begin(2);
            if (SUPER || (kSolo && __popcll(pending) >= DRT_COOP_SOLO_MIN)) {
                // more than half of the lanes carry a walk: a round would give every walk ONE lane (m = 1) - take that
                // step on the walk's own lane, without the slot table, the gathers and the jump-ahead (same arithmetic).
                // With a supergrid every step is taken this way.
                if (job) {
                    const uint64_t sc = S.state;
                    float lm, lim;
                    const float dt = flight(o, d, tmax, pcg_float(sc), lm, lim);
                    const bool inside = dt <= tmax;                             // :480-481
                    const V3 p = v3(fmaf(d.x, dt, o.x), fmaf(d.y, dt, o.y), fmaf(d.z, dt, o.z));
                    const float sig = inside ? eval_sigma_t(P, p, occ) : 0.0f;
                    const float tr = (lm - sig) * lim;                          // :473-476
                    const float Tout = inside ? T * tr : T;
                    if (inside) {
                        count(C_RT);
                        if constexpr (ADJ) if (tr > 0.0f) {                     // :487-492
                            splat_sigma_t<DEFER>(P, p, -(a_sum * lim) / tr, rec);
                            count(C_RT_ADJ);
                        }
                        ++steps;
                    }
                    S.state = sc * kPcgMul + S.inc;
                    T = Tout; o = p; tmax = tmax - dt;
                    if (!inside || Tout == 0.0f) job = false;                   // :495, :502
                }
                pending = __ballot(job);
                round_end(2, Jprof);
                continue;
            }
            if constexpr (!SUPER) {
                int m, lg, js, c, owner, my_rank; bool serve;
                round_setup(job, pending, m, lg, js, c, serve, owner, my_rank);
                // the walk this lane serves
                float cx = __shfl(o.x, owner), cy = __shfl(o.y, owner), cz = __shfl(o.z, owner);
                const float dx = __shfl(d.x, owner), dy = __shfl(d.y, owner), dz = __shfl(d.z, owner);
                float ct = __shfl(tmax, owner);
                float Tin = __shfl(T, owner);
                const float asum = ADJ ? __shfl(a_sum, owner) : 0.0f;
                const uint64_t st = shfl64(S.state, owner), inc = shfl64(S.inc, owner);
                // step c of this round: its own draw and free-flight distance
                const uint64_t sc = pcg_jump(st, inc, c);
                const float dt = sample_distance(pcg_float(sc));
                // chain 1 (sequential semantics): origin and remaining length before step c; reached = every
                // earlier step of the round found its tentative collision inside the segment
                // (every lane replays the earlier steps of its walk itself, in order, with their distances
                // gathered from the lanes that drew them: independent ds_bpermutes instead of a dependent chain)
                bool reached = serve;
                const int gb = (int) __lane_id() - c;                               // first lane of my group
                for (int k = 0; k + 1 < m; ++k) {
                    const float dk = __shfl(dt, gb + k);
                    if (k < c) {
                        reached = reached && dk <= ct;
                        cx = fmaf(dx, dk, cx); cy = fmaf(dy, dk, cy); cz = fmaf(dz, dk, cz); ct = ct - dk;
                    }
                }
                const bool inside = reached && dt <= ct;                            // :480-481
                const V3 p = v3(fmaf(dx, dt, cx), fmaf(dy, dt, cy), fmaf(dz, dt, cz));
                const float sig = inside ? eval_sigma_t(P, p, occ) : 0.0f;
                const float tr = (maj - sig) * inv_maj;                             // :473-476
                // chain 2: running product before step c; a step is executed iff the product is still non-zero
                bool live = reached;                                                // step c is started (its draw is consumed)
                for (int k = 0; k + 1 < m; ++k) {
                    const float trk = __shfl(tr, gb + k);
                    if (k < c) { Tin = Tin * trk; live = live && Tin != 0.0f; }       // :495, :502
                }
                const bool exec = live && inside;
                const float Tout = exec ? Tin * tr : Tin;
                if (exec) {
                    count(C_RT);
                    if constexpr (ADJ) if (tr > 0.0f) {                             // :487-492
                        splat_sigma_t<DEFER>(P, p, -(asum * inv_maj) / tr, rec);
                        count(C_RT_ADJ);
                    }
                }
                // the last started step of every walk reports: draws consumed = its index + 1
                const bool ends = live && (!inside || Tout == 0.0f);
                const int nxt = __shfl_down(live ? 1 : 0, 1);
                const bool last = live && (ends || c == m - 1 || nxt == 0);
                const uint64_t last_mask = __ballot(last);
                // (shuffles only in wave-uniform control flow: a lane that is switched off cannot be read)
                const int base = my_rank << lg;
                const int src = job ? base + __ffsll((long long) ((last_mask >> base) & ((1ull << m) - 1ull))) - 1 : (int) __lane_id();
                const float rT = __shfl(Tout, src), rx = __shfl(p.x, src), ry = __shfl(p.y, src), rz = __shfl(p.z, src);
                const float rt = __shfl(ct - dt, src);
                const int rend = __shfl(ends ? 1 : 0, src);
                const int rout = __shfl((live && !inside) ? 1 : 0, src);
                const uint64_t rs = shfl64(sc * kPcgMul + inc, src);               // stream after the last started step's draw
                if (job) {
                    T = rT; o = v3(rx, ry, rz); tmax = rt;
                    S.state = rs;
                    steps += (uint32_t) (src - base + 1 - rout);                    // executed steps of my walk in this round
                    if (rend) job = false;
                }
                pending = __ballot(job);
                round_end(2, Jprof);
            }
        }
        if (steps_out) *steps_out = steps;
        return T;
    }

    // ---------------------------------------------------------------------------------------------
    // sample_real_interaction: delta tracking (volpathsimple.py:323-377), all lanes of the wave call it
    // ---------------------------------------------------------------------------------------------
    __device__ __forceinline__ Mei coop_dt(bool job, const Ray &ray, Pcg32 &S, uint32_t &steps)
    {
        Mei mei; mei.valid = false; mei.t = kInf; mei.p = v3(0, 0, 0); mei.sigma_t = 0.0f;
        steps = 0;
        V3 ro = ray.o; float rmaxt = ray.maxt, running_t = 0.0f;
        uint64_t pending = __ballot(job);
        while (pending) {
            const int Jprof = __popcll(pending); (void) Jprof;
            round_begin(3);
            if (SUPER || (kSolo && __popcll(pending) >= DRT_COOP_SOLO_MIN)) {               // m = 1 rounds on the walks' own lanes (see coop_rt)
                if (job) {
                    const uint64_t s0 = S.state, s1 = s0 * kPcgMul + S.inc;
                    float lm, lim;
                    const float dt = flight(ro, ray.d, rmaxt, pcg_float(s0), lm, lim);   // :348
                    const float u2 = pcg_float(s1);                             // :359
                    const bool inside = dt <= rmaxt;                            // :358
                    const V3 p = v3(fmaf(ray.d.x, dt, ro.x), fmaf(ray.d.y, dt, ro.y), fmaf(ray.d.z, dt, ro.z));
                    const float sig = inside ? eval_sigma_t(P, p, occ) : 0.0f;
                    const float r = sig * lim;                                  // :354
                    const bool accepted = inside && !(u2 >= r);                 // :351
                    if (inside) count(C_DT);
                    S.state = inside ? s1 * kPcgMul + S.inc : s1;
                    const float rtm = running_t + dt;
                    if (!inside || accepted) {
                        job = false;
                        if (accepted) { mei.valid = true; mei.t = rtm; steps += 1u; }
                    } else { ro = p; rmaxt = rmaxt - dt; running_t = rtm; steps += 1u; }
                }
                pending = __ballot(job);
                round_end(3, Jprof);
                continue;
            }
            if constexpr (!SUPER) {
                int m, lg, js, c, owner, my_rank; bool serve;
                round_setup(job, pending, m, lg, js, c, serve, owner, my_rank);
                float cx = __shfl(ro.x, owner), cy = __shfl(ro.y, owner), cz = __shfl(ro.z, owner);
                const float dx = __shfl(ray.d.x, owner), dy = __shfl(ray.d.y, owner), dz = __shfl(ray.d.z, owner);
                float ct = __shfl(rmaxt, owner), crun = __shfl(running_t, owner);
                const uint64_t st = shfl64(S.state, owner), inc = shfl64(S.inc, owner);
                // step c: draws 2c (distance) and 2c + 1 (acceptance)
                const uint64_t s0 = pcg_jump(st, inc, 2 * c), s1 = s0 * kPcgMul + inc;
                const float dt = sample_distance(pcg_float(s0));                    // :348
                const float u2 = pcg_float(s1);                                     // :359
                bool reached = serve;
                const int gb = (int) __lane_id() - c;                               // first lane of my group
                for (int k = 0; k + 1 < m; ++k) {
                    const float dk = __shfl(dt, gb + k);
                    if (k < c) {                                                    // :364-367
                        reached = reached && dk <= ct;
                        cx = fmaf(dx, dk, cx); cy = fmaf(dy, dk, cy); cz = fmaf(dz, dk, cz); ct = ct - dk; crun = crun + dk;
                    }
                }
                const bool inside = reached && dt <= ct;                            // :358
                const V3 p = v3(fmaf(dx, dt, cx), fmaf(dy, dt, cy), fmaf(dz, dt, cz));
                const float sig = inside ? eval_sigma_t(P, p, occ) : 0.0f;
                const float r = sig * inv_maj;                                      // :354
                const bool accepted = inside && !(u2 >= r);                         // :351
                const bool term = reached && (!inside || accepted);
                const uint64_t term_mask = __ballot(term);
                const int gbase = js << lg;                                         // first lane of the group I serve
                const uint64_t gmask = ((1ull << m) - 1ull);
                const uint64_t gterm = serve ? ((term_mask >> gbase) & gmask) : 0ull;
                const int e = gterm ? __ffsll((long long) gterm) - 1 : m - 1;      // last started step of my walk in this round
                if (inside && c <= e) count(C_DT);
                // owner side
                const int base = my_rank << lg;
                const uint64_t oterm = job ? ((term_mask >> base) & gmask) : 0ull;
                const int oe = oterm ? __ffsll((long long) oterm) - 1 : m - 1;
                const int src = job ? base + oe : (int) __lane_id();
                const int racc = __shfl(accepted ? 1 : 0, src);
                const float rtm = __shfl(crun + dt, src);                           // running_t + dt: mei.t or the new running_t
                const float rx = __shfl(p.x, src), ry = __shfl(p.y, src), rz = __shfl(p.z, src), rt = __shfl(ct - dt, src);
                // stream after the last started step: one draw if it fell outside the segment, two otherwise
                const uint64_t rs = shfl64((reached && !inside) ? s1 : s1 * kPcgMul + inc, src);
                if (job) {
                    S.state = rs;
                    if (oterm) {
                        job = false;
                        if (racc) { mei.valid = true; mei.t = rtm; }
                        steps += (uint32_t) (racc ? oe + 1 : oe);
                    } else { ro = v3(rx, ry, rz); rmaxt = rt; running_t = rtm; steps += (uint32_t) m; }
                }
                pending = __ballot(job);
                round_end(3, Jprof);
            }
        }
        return mei;                                                             // mei.p / attached sigma_t: the caller (:371-375)
    }

    // sample_emitter (volpathsimple.py:406-433): emitter_val * transmittance in out[], ds.pdf returned.
    // cache (value walk only): mode 1 stores {T, sampler state behind the walk, steps} into *ce, mode 2 takes
    // them from it instead of walking.
    template <bool ADJ>
    __device__ __forceinline__ float sample_emitter(bool job, V3 p, Pcg32 &S, const float *adj, float out[3], int cmode = 0, uint4 *ce = nullptr)
    {
        float val[3] = { 0.0f, 0.0f, 0.0f }, pdf = 0.0f, tmax = 0.0f;
        V3 wd = v3(0, 0, 1);
        bool walk = false;
        if (job) {
            float ux = S.next_1d(), uy = S.next_1d();                           // :418
            wd = emitter_sample_dir<ENV>(P, ux, uy);
            pdf = emitter_sample_value<ENV>(P, wd, val);
            if (pdf != 0.0f && cmode != 2) {                                    // sampling_worked :421-423
                Hit si = box_hit(P, p, wd);                                     // :427-428
                walk = si.valid; tmax = si.t;
            }
        }
        const float a_sum = (ADJ && job) ? (adj[0] + adj[1]) + adj[2] : 0.0f;
        uint32_t steps = 0;
        float T = coop_rt<ADJ>(walk, p, wd, tmax, S, a_sum, &steps);
        if (!walk) T = 0.0f;
        if (job && cmode == 1) *ce = make_uint4(__float_as_uint(T), (uint32_t) S.state, (uint32_t) (S.state >> 32), steps);
        if (job && cmode == 2) {
            const uint4 e = *ce;
            T = __uint_as_float(e.x); S.state = ((uint64_t) e.z << 32) | e.y;
            if (COUNT) cnt[C_RT] += e.w;
        }
#pragma unroll
        for (int k = 0; k < 3; ++k) out[k] = val[k] * T;
        return pdf;
    }

    // sample_emitter_for_nee (volpathsimple.py:380-403)
    template <bool ADJ>
    __device__ __forceinline__ void sample_emitter_for_nee(bool job, V3 p, Pcg32 &S, const float beta[3], const float *dL, float contrib[3],
                                           int cmode = 0, uint4 *ce = nullptr)
    {
        Pcg32 clone = S;                                                        // :383
        float emitted[3];
        float ds_pdf = sample_emitter<false>(job, p, S, nullptr, emitted, cmode, ce);   // :385
        float w = mis_weight(ds_pdf, kInvFourPi);                               // :391
#pragma unroll
        for (int k = 0; k < 3; ++k) contrib[k] = job ? ((beta[k] * kInvFourPi) * w) * emitted[k] : 0.0f;
        if constexpr (ADJ) {                                                    // :393-401
            float adj[3] = { 0.0f, 0.0f, 0.0f };
            if (job) { adj[0] = dL[0] * contrib[0]; adj[1] = dL[1] * contrib[1]; adj[2] = dL[2] * contrib[2]; }
            float unused[3];
            (void) sample_emitter<true>(job, p, clone, adj, unused);
        }
    }

    // Medium::sample_interaction_drt (call site volpathsimple.py:549-551); one lane per walk (54 % lane
    // utilisation on the headline workload: the reservoir vertex of every ray is walked at the same time)
    __device__ __forceinline__ bool sample_interaction_drt(const Ray &ray, Pcg32 &A, float &t_out, float &W_out)
    {
        float t = 0.0f, T = 1.0f, wsum = 0.0f, tsel = kInf;
        bool valid = false;
        for (;;) {
            float lm, lim;
            if constexpr (SUPER) t += flight(ray_at(ray.o, ray.d, t), ray.d, ray.maxt - t, A.next_1d(), lm, lim);
            else t += flight(ray.o, ray.d, ray.maxt, A.next_1d(), lm, lim);
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
    // `tail`: hand-off mode - the path's end (Li, gradient splat) happens inside sample(), on whichever lane it ends
    __device__ __forceinline__ void sample_recursive(bool job, Pcg32 &A, V3 p, int depth, float Li[3], Tail *tail = nullptr)
    {
        Li[0] = Li[1] = Li[2] = 0.0f;
        if (use_nee()) {                                                        // :621-624 (wave-uniform condition)
            const float one[3] = { 1.0f, 1.0f, 1.0f };
            float nee[3];
            phase(C_SC);
            sample_emitter_for_nee<false>(job, p, A, one, nullptr, nee);
            phase(C_RT_ADJ);
#pragma unroll
            for (int k = 0; k < 3; ++k) Li[k] += nee[k];
        }
        Ray rr; rr.o = p; rr.d = v3(0, 0, 1); rr.maxt = kLargest;
        PathState ps;
        ps.tail = tail;
        if (tail) { tail->nee[0] = Li[0]; tail->nee[1] = Li[1]; tail->nee[2] = Li[2]; }
        ps.depth = depth + 1; ps.last_pdf = kInvFourPi; ps.escaped = false; ps.active = false;
        ps.si.valid = false; ps.si.t = kInf; ps.si.p = v3(0, 0, 0); ps.si.n = v3(0, 0, 0);
        if (job) {
            (void) A.next_1d();                                                 // :632
            float ux = A.next_1d(), uy = A.next_1d();
            rr.d = square_to_uniform_sphere(ux, uy);
            Hit sn = box_hit(P, p, rr.d);                                       // :637
            rr.maxt = sn.valid ? sn.t : kLargest;                               // :639-640
            ps.si = sn;
            ps.active = (ps.depth < P.max_depth) && sn.valid;                   // :647 (+ DESIGN.md deviation)
        }
        float Lr[3];
        sample<false, true>(job, A, rr, nullptr, nullptr, &ps, Lr);             // :651
#pragma unroll
        for (int k = 0; k < 3; ++k) Li[k] += Lr[k];
    }

    // backpropagate_scattering_drt, final / quadratic branch (volpathsimple.py:543-581)
    __device__ __forceinline__ void drt_backprop(bool job, Pcg32 &A, const Ray &ray, float si_t, int depth, const float adj[3])
    {
        Ray sub = ray;
        sub.maxt = isfinite(si_t) ? si_t : kLargest;                            // :544-545
        float tp = kInf, W = 0.0f;
        bool found = false;
        phase(C_TR);
        if (job) found = sample_interaction_drt(sub, A, tp, W);                 // :550,558
        phase(C_DRT);
        V3 p = v3(0, 0, 0);
        float sig = 0.0f;
        float alb[3] = { 0.0f, 0.0f, 0.0f };
        if (found) {
            p = ray_at(sub.o, sub.d, tp);
            if constexpr (G4) eval4(P, p, sig, alb);                            // :553-554 and :578 in one lookup
            else sig = eval_sigma_t(P, p, occ);                                 // :553-554
            count(C_DRT);
        }
        float Li[3];
        if constexpr (SPEC) {
            if (wgc) {                                                          // (workgroup-uniform: set by the kernel)
                // hand-off mode: everything the final splat needs is computed now and travels with the recursive path
                Tail tl;
                tl.p = p; tl.sig = sig;
                if (found) { if constexpr (!G4) eval_albedo(P, p, alb); count(C_ALB); }
                const float w = P.use_drt_mis ? 1.0f / (1.0f + sig * sig) : 1.0f;   // :571-575
                const float ww = w * W;
#pragma unroll
                for (int k = 0; k < 3; ++k) { tl.alb[k] = alb[k]; tl.wadj[k] = ww * adj[k]; tl.nee[k] = 0.0f; }
                sample_recursive(found, A, p, depth, Li, &tl);                  // :565-581 (splat included)
                return;
            }
        }
        sample_recursive(found, A, p, depth, Li);                               // :565-568
        if (found) {
            float w = P.use_drt_mis ? 1.0f / (1.0f + sig * sig) : 1.0f;         // :571-575
            if constexpr (!G4) eval_albedo(P, p, alb);                          // :578
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
    }

    // backpropagate_transmittance (volpathsimple.py:584-607)
    __device__ __forceinline__ void backprop_transmittance(Pcg32 &A, const Ray &ray, float interval, const float dL[3], const float result[3])
    {
        float adjw = (dL[0] * result[0] + dL[1] * result[1]) + dL[2] * result[2];
        float g = -(adjw * (interval / 4.0f));
        for (int j = 0; j < 4; ++j) {
            float t = A.next_1d() * interval;                                   // :595
            splat_sigma_t<DEFER>(P, ray_at(ray.o, ray.d, t), g, rec);
            count(C_TR);
        }
    }

    // emitter seen by a path that left the medium (volpathsimple.py:263-287), primal mode
    __device__ __forceinline__ void add_escaped_emission(bool escaped, int depth, bool has_scattered, float last_pdf, V3 d,
                                                         const float beta[3], float result[3])
    {
        if (escaped && !(depth <= 0 && P.hide_emitters)) {
            float w = 1.0f, Le[3];
            const float e_pdf = emitter_eval_pdf<ENV>(P, d, Le);                // :284 and :273: the same taps of the map
            if (use_nee()) w = mis_weight(last_pdf, has_scattered ? e_pdf : 0.0f);   // :273-277
#pragma unroll
            for (int k = 0; k < 3; ++k) result[k] += (beta[k] * w) * Le[k];
        }
    }

    // Workgroup hand-off of the recursive DRT paths, called by every wave after each bounce-loop iteration.
    // Measured: iterations with <= 8 live lanes are half of the recursive paths' time - the
    // tracking rounds of a sparse wave serve few walks.  So: a path that ended is finished HERE (its radiance, the
    // gradient splat of its reservoir vertex: `tl`) and leaves its lane; waves 1..3, once they have at most kWgcDonate
    // live paths, write them - complete state + tail - to their LDS region, publish the count and are done (they run
    // to the end of the kernel and free their slot); wave 0 picks the donated paths up into its free lanes as they
    // appear and runs them to the end.  No workgroup barrier: donors never wait; wave 0 only waits when it has nothing
    // to do while another wave is still dense.  A path computes the same numbers on any lane.
    // Returns false when this wave has no path left and expects none.
    // MAIN: the main path of the primal pass instead (same protocol): a path that ended writes its radiance and its
    // ray-schedule key for its home ray `i_block + home`; `it` (its path-cache cursor) travels along.
    // Tail pool (adjoint pass): when wave 0 has taken every donation and is itself down to DRT_TAIL_PUSH live paths, it
    // writes them to a global pool and ends as well; a second launch of the same kernel in tail mode
    // (Params::tail_mode) starts with 256 of those paths per workgroup - the longest recursive paths of the job, densely
    // packed again.  (Dropping them, as a timing experiment, took 0.8 ms off the adjoint tracer: what they cost at
    // <= 8 lanes per wave.  The tail launch gives back 0.35 of it: its own duration is the job's longest path.  The
    // primal pass has nothing to hide that behind and keeps its paths.)
    template <bool MAIN>
    __device__ __forceinline__ bool wg_handoff(bool &job, bool &active, uint32_t &taken, Ray &ray, float beta[3], float result[3], Pcg32 &S,
                               int &depth, bool &escaped, bool &has_scattered, float &last_pdf, Tail &tl, int &home, int &it)
    {
        lds_u32 *flags = (lds_u32 *) wgc, *pool = flags + 4;
        const int wave = (int) (threadIdx.x >> 6), lane = (int) __lane_id();
        constexpr int kUsed = MAIN ? 20 : kWgcFields;                           // words of an entry in use
// the complete state of a path as kWgcFields words (MAIN: word 19 = home ray, relative to i_block) and back
#define DRT_PACK_PATH(w)                                                                                                   \
        const uint32_t w[kWgcFields] = {                                                                                   \
            __float_as_uint(ray.o.x), __float_as_uint(ray.o.y), __float_as_uint(ray.o.z),                                  \
            __float_as_uint(ray.d.x), __float_as_uint(ray.d.y), __float_as_uint(ray.d.z), __float_as_uint(ray.maxt),       \
            __float_as_uint(beta[0]), __float_as_uint(beta[1]), __float_as_uint(beta[2]),                                  \
            __float_as_uint(result[0]), __float_as_uint(result[1]), __float_as_uint(result[2]),                            \
            (uint32_t) S.state, (uint32_t) (S.state >> 32), (uint32_t) S.inc, (uint32_t) (S.inc >> 32),                    \
            (uint32_t) depth, (escaped ? 1u : 0u) | (has_scattered ? 2u : 0u) | (MAIN ? ((uint32_t) it << 16) : 0u),       \
            MAIN ? (uint32_t) home : __float_as_uint(tl.p.x), __float_as_uint(tl.p.y), __float_as_uint(tl.p.z),            \
            __float_as_uint(tl.sig), __float_as_uint(tl.alb[0]), __float_as_uint(tl.alb[1]), __float_as_uint(tl.alb[2]),   \
            __float_as_uint(tl.wadj[0]), __float_as_uint(tl.wadj[1]), __float_as_uint(tl.wadj[2]),                         \
            __float_as_uint(tl.nee[0]), __float_as_uint(tl.nee[1]), __float_as_uint(tl.nee[2]) }
#define DRT_UNPACK_PATH(v)                                                                                                 \
        do {                                                                                                               \
            ray.o = v3(__uint_as_float(v[0]), __uint_as_float(v[1]), __uint_as_float(v[2]));                               \
            ray.d = v3(__uint_as_float(v[3]), __uint_as_float(v[4]), __uint_as_float(v[5]));                               \
            ray.maxt = __uint_as_float(v[6]);                                                                              \
            _Pragma("unroll") for (int k_ = 0; k_ < 3; ++k_) {                                                             \
                beta[k_] = __uint_as_float(v[7 + k_]); result[k_] = __uint_as_float(v[10 + k_]);                           \
                tl.alb[k_] = __uint_as_float(v[23 + k_]); tl.wadj[k_] = __uint_as_float(v[26 + k_]);                       \
                tl.nee[k_] = __uint_as_float(v[29 + k_]);                                                                  \
            }                                                                                                              \
            S.state = ((uint64_t) v[14] << 32) | v[13];                                                                    \
            S.inc = ((uint64_t) v[16] << 32) | v[15];                                                                      \
            depth = (int) v[17];                                                                                           \
            escaped = (v[18] & 1u) != 0u; has_scattered = (v[18] & 2u) != 0u;                                              \
            if constexpr (MAIN) {                                                                                          \
                home = (int) v[19]; it = (int) (v[18] >> 16);                                                              \
                last_pdf = has_scattered ? kInvFourPi : 1.0f;                                                              \
                pc = P.path_cache_mode == 1 ? P.path_cache + (size_t) (i_block + (uint64_t) (uint32_t) home) * P.path_cache_cap * 2 : nullptr; \
            }                                                                                                              \
            tl.p = v3(__uint_as_float(v[19]), __uint_as_float(v[20]), __uint_as_float(v[21]));                             \
            tl.sig = __uint_as_float(v[22]);                                                                               \
            job = active = true;                                                                                           \
        } while (0)

        if (!MAIN && tail_load) {                                               // tail-mode launch: fetch this lane's path from the pool
            tail_load = false;
            const uint32_t idx = blockIdx.x * 256u + threadIdx.x;
            if (idx < min(*P.tail_count, P.tail_cap)) {
                const uint4 *src = P.tail_pool + (size_t) idx * 8;
                uint32_t v[kWgcFields];
#pragma unroll
                for (int q = 0; q < 8; ++q) { const uint4 u = (4 * q < kUsed) ? src[q] : make_uint4(0, 0, 0, 0); v[4 * q] = u.x; v[4 * q + 1] = u.y; v[4 * q + 2] = u.z; v[4 * q + 3] = u.w; }
                DRT_UNPACK_PATH(v);
            }
        }
        if (MAIN && job && !active) {                                           // the path ended: radiance, schedule key
            add_escaped_emission(escaped, depth, has_scattered, last_pdf, ray.d, beta, result);
            const uint64_t i = i_block + (uint64_t) (uint32_t) home;
            P.L_out[3 * i] = result[0]; P.L_out[3 * i + 1] = result[1]; P.L_out[3 * i + 2] = result[2];
            if (P.ray_iters) P.ray_iters[i] = (uint8_t) (it < 255 ? it : 255);
            job = false;
        }
        if (!MAIN && job && !active) {                                          // the path ended: Li, gradient splat (:565-581)
            add_escaped_emission(escaped, depth, has_scattered, last_pdf, ray.d, beta, result);
            float gs = 0.0f, ga[3];
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const float Li = tl.nee[k] + result[k];
                const float a = tl.wadj[k] * Li;
                gs += a * tl.alb[k];
                ga[k] = a * tl.sig;
            }
            splat_scatter<DEFER>(P, tl.p, gs, ga, rec); count(C_SC); count(C_SC_ALB);
            job = false;
        }
        uint64_t am = __ballot(active);
        int n = __popcll(am);
        if (wave != 0) {
            if (n > kWgcDonate) return true;                                    // dense: carry on
            if (active) {
                lds_u32 *q = pool + (wave - 1) * (kWgcFields * kWgcDonate) + __popcll(am & ((1ull << lane) - 1ull));
                DRT_PACK_PATH(w);
#pragma unroll
                for (int f = 0; f < kUsed; ++f) q[f * kWgcDonate] = w[f];
            }
            coop_stage_sync();                                                  // the states are in LDS ...
            if (lane == 0) flags[wave - 1] = (uint32_t) n;                      // ... before the count is published
            job = active = false;
            return false;
        }
        for (;;) {                                                              // wave 0: collect what the others have published
#pragma unroll
            for (int w = 0; w < 3; ++w) {
                if (taken & (1u << w)) continue;
                const uint32_t f = (uint32_t) __builtin_amdgcn_readfirstlane((int) flags[w]);
                if (f == 0xffffffffu || (int) f > 64 - n) continue;
                const int slot = __popcll(~am & ((1ull << lane) - 1ull));       // my rank among the free lanes
                if (!active && slot < (int) f) {
                    lds_u32 *q = pool + w * (kWgcFields * kWgcDonate) + slot;
                    uint32_t v[kWgcFields];
#pragma unroll
                    for (int k = 0; k < kWgcFields; ++k) v[k] = k < kUsed ? q[k * kWgcDonate] : 0u;
                    DRT_UNPACK_PATH(v);
                }
                taken |= 1u << w;
                am = __ballot(active);
                n = __popcll(am);
            }
            if (!MAIN && taken == 7u && n > 0 && n <= DRT_TAIL_PUSH && P.tail_pool && !P.tail_mode) {
                // the workgroup's last few paths: to the tail pool, if it has room
                uint32_t base = 0xffffffffu;
                if (lane == 0) {
                    // every workgroup reserves at most ONCE (its wave 0 ends with the push) and at most DRT_TAIL_PUSH entries, and
                    // the launcher only binds a pool with tail_cap >= DRT_TAIL_PUSH * gridDim.x (launch_trace_coop_t): the
                    // reservation cannot overflow, so a plain add is linearizable here (a compare-and-swap loop was measured:
                    // 10x slower adjoint - 32 k workgroups retrying on one address).  The overflow branch only guards against
                    // a caller that breaks the invariant: the paths then stay where they are.
                    base = atomicAdd(P.tail_count, (uint32_t) n);
                    if (base + (uint32_t) n > P.tail_cap) { atomicSub(P.tail_count, (uint32_t) n); base = 0xffffffffu; }
                }
                base = (uint32_t) __builtin_amdgcn_readfirstlane((int) base);
                if (base != 0xffffffffu) {
                    if (active) {
                        DRT_PACK_PATH(w);
                        uint32_t ww[kWgcFields];
#pragma unroll
                        for (int f = 0; f < kWgcFields; ++f) ww[f] = w[f];
                        if constexpr (MAIN) ww[19] = (uint32_t) (i_block + (uint64_t) (uint32_t) home);   // the tail launch has i_block = 0
                        uint4 *dst = P.tail_pool + (size_t) (base + (uint32_t) __popcll(am & ((1ull << lane) - 1ull))) * 8;
#pragma unroll
                        for (int q = 0; 4 * q < kUsed; ++q) dst[q] = make_uint4(ww[4 * q], ww[4 * q + 1], ww[4 * q + 2], ww[4 * q + 3]);
                    }
                    job = active = false;
                    return false;
                }
            }
            if (n > 0) return true;
            if (taken == 7u) return false;
            __builtin_amdgcn_s_sleep(16);                                       // nothing to do until another wave publishes
        }
#undef DRT_PACK_PATH
#undef DRT_UNPACK_PATH
    }

    // VolpathSimpleIntegrator.sample (volpathsimple.py:38-290); `job`: this lane carries a ray
    template <bool ADJ, bool RECURSIVE>
    __device__ __forceinline__ void sample(bool job, Pcg32 &S, Ray ray, const float *dL, const float *state_in, const PathState *ps, float out[3])
    {
        float result[3] = { 0.0f, 0.0f, 0.0f };
        float beta[3] = { 1.0f, 1.0f, 1.0f };
        if (ADJ && job) { result[0] = state_in[0]; result[1] = state_in[1]; result[2] = state_in[2]; }

        bool active = false, escaped = false; int depth = 0;
        Hit si; si.valid = false; si.t = kInf; si.p = v3(0, 0, 0); si.n = v3(0, 0, 0);
        if (RECURSIVE) {                                                        // :61-67
            active = job && ps->active; depth = ps->depth; si = ps->si; escaped = ps->escaped;
        } else if (job) {
            active = true;
            (void) S.next_1d();                                                 // :71
            si = box_hit(P, ray.o, ray.d);                                      // reach_medium :292-319
            if (!si.valid) { escaped = true; active = false; }
            else {
                ray.o = offset_p(si, ray.d);
                Hit sn = box_hit(P, ray.o, ray.d);
                if (!sn.valid) active = false;
                else { ray.maxt = sn.t; si = sn; }
            }
        }
        bool has_scattered = RECURSIVE ? (active && !escaped) : false;          // :84-89
        float last_pdf = RECURSIVE ? ps->last_pdf : 1.0f;

        // DRTReservoir(n=1) + DRTPathState (:94-96, :710-765)
        int r_depth = -1; float r_si_t = kInf; Ray r_ray = ray;
        float r_wsum[3] = { 0, 0, 0 }, r_cw[3] = { 0, 0, 0 };

        Pcg32 A; A.state = 0; A.inc = 1;
        if (active) (void) S.next_1d();                                         // :99
        if constexpr (ADJ) { if (job) A.seed(P.alt_seed, ray_index); }          // :100-107

        int it = 0;                                                             // bounce-loop iterations this ray has run
        // recursive paths of the specialised adjoint kernels: sparse waves hand their last paths to wave 0 (wg_handoff)
        // ... and so do the main paths of the specialised primal kernels (their radiance is then written by wg_handoff)
        constexpr bool kWgc = !ADJ && SPEC;
        bool wgc_on = false;
        Tail tl;
        uint32_t taken = 0;
        int home = (int) threadIdx.x;
        if constexpr (kWgc && RECURSIVE) { if (wgc && ps->tail) { wgc_on = true; tl = *ps->tail; } }
        if constexpr (kWgc && !RECURSIVE) {
            wgc_on = wgc != nullptr;
            tl.p = v3(0, 0, 0); tl.sig = 0.0f;
#pragma unroll
            for (int k = 0; k < 3; ++k) { tl.alb[k] = 0.0f; tl.wadj[k] = 0.0f; tl.nee[k] = 0.0f; }
        }
        for (;;) {                                                              // :114, wave-uniform
            const uint64_t any_active = __ballot(active);
            if (!wgc_on && !any_active) break;
            const int Jit = __popcll(any_active); (void) Jit;
            round_begin(RECURSIVE ? 4 : 5);          // phase-profile modes 4 / 5: bounce-loop iterations by live lanes
            if (any_active) {
            bool run = active;
            if (run) {
                float q = fminf(fmaxf(beta[0], fmaxf(beta[1], beta[2])), 0.99f);    // :117-121
                bool perform_rr = depth > P.rr_depth;
                float u_rr = S.next_1d();
                active = (beta[0] != 0.0f || beta[1] != 0.0f || beta[2] != 0.0f) && (!perform_rr || (u_rr < q));
                if (perform_rr) { float iq = 1.0f / q; beta[0] *= iq; beta[1] *= iq; beta[2] *= iq; }
                run = active;
            }

            // path cache (main path only): the adjoint takes this iteration's walk from the primal pass
            const int cmode = (!RECURSIVE && pc && run && it < (int) P.path_cache_cap) ? (int) P.path_cache_mode : 0;
            uint4 *ce = cmode ? pc + 2 * it : nullptr;
            uint32_t dt_steps = 0;
            phase(RECURSIVE ? C_SC : C_TR);
            Mei mei = coop_dt(run && cmode != 2, ray, S, dt_steps);             // :126
            phase(RECURSIVE ? C_ALB : C_DT);
            work += dt_steps;                 // (+ 256 per bounce-loop iteration of the wave's longest path: ray_perm_kernel)
            if (cmode == 2) {
                const uint4 e = ce[0];
                mei.t = __uint_as_float(e.x); mei.valid = mei.t < kInf;
                S.state = ((uint64_t) e.z << 32) | e.y;
                if (COUNT) cnt[C_DT] += e.w;
            } else if (cmode == 1) {
                ce[0] = make_uint4(__float_as_uint(mei.valid ? mei.t : kInf), (uint32_t) S.state, (uint32_t) (S.state >> 32), dt_steps);
            }
            float albedo[3] = { 1.0f, 1.0f, 1.0f };                             // :141
            if (run && mei.valid) {
                mei.p = ray_at(ray.o, ray.d, mei.t);                            // :371
                if constexpr (G4) { float s4; eval4(P, mei.p, s4, albedo); if (ADJ) { mei.sigma_t = s4; count(C_DT); } }
                else if (ADJ) { mei.sigma_t = eval_sigma_t(P, mei.p, occ); count(C_DT); }   // :373-375
            }
            const bool did_escape = run && !mei.valid, did_scatter = run && mei.valid;   // :130-134
            has_scattered |= did_scatter;

            if (did_scatter) { if constexpr (!G4) eval_albedo(P, mei.p, albedo); count(C_ALB); }

            if constexpr (ADJ) {
                if (use_drt()) {                                                // :143-150
                    if (use_sub()) {                                            // :521-539, :745-753
                        if (run) {
                            float u = A.next_1d();
                            float mm = 0.0f;
#pragma unroll
                            for (int k = 0; k < 3; ++k) { r_wsum[k] += beta[k]; mm += beta[k] / r_wsum[k]; }
                            mm = mm / 3.0f;
                            if (u <= mm) {
                                r_cw[0] = beta[0]; r_cw[1] = beta[1]; r_cw[2] = beta[2];
                                r_depth = depth; r_si_t = si.t; r_ray = ray;
                            }
                        }
                    } else if constexpr (!SPEC) {
                        float adj[3] = { 0.0f, 0.0f, 0.0f };
                        if (run) { adj[0] = dL[0] * beta[0]; adj[1] = dL[1] * beta[1]; adj[2] = dL[2] * beta[2]; }
                        drt_backprop(run, A, ray, si.t, depth, adj);
                    }
                }
                if ((!use_drt() || P.use_drt_mis) && did_scatter) {             // :152-172
                    float w = 1.0f;
                    if (use_drt() && P.use_drt_mis) {
                        float s2 = mei.sigma_t * mei.sigma_t;
                        w = s2 / (1.0f + s2);
                    }
                    float inv_pdf = 1.0f / mei.sigma_t;
                    float gs = 0.0f, ga[3];
#pragma unroll
                    for (int k = 0; k < 3; ++k) {
                        float Li = result[k] / fmaxf(1e-8f, albedo[k]);         // :167
                        float a = ((w * dL[k]) * Li) * inv_pdf;
                        gs += a * albedo[k];
                        ga[k] = a * mei.sigma_t;
                    }
                    splat_scatter<DEFER>(P, mei.p, gs, ga, rec); count(C_SC); count(C_SC_ALB);
                }
                if (run) backprop_transmittance(A, ray, did_escape ? si.t : mei.t, dL, result);   // :181-189
            }

            bool nee_job = false;
            if (run) {
                beta[0] *= albedo[0]; beta[1] *= albedo[1]; beta[2] *= albedo[2];   // :193
                if (did_scatter) depth += 1;                                    // :199
                active = did_scatter && (depth < P.max_depth);                  // :200
                nee_job = use_nee() && did_scatter && active;                   // :206-215
            }
            if (use_nee()) {
                float nee[3];
                phase(RECURSIVE ? C_SC : C_TR);
                sample_emitter_for_nee<ADJ>(nee_job, mei.p, S, beta, dL, nee, nee_job ? cmode : 0, ce ? ce + 1 : nullptr);
                phase(RECURSIVE ? C_RT_ADJ : C_RT);
                if (nee_job) {
#pragma unroll
                    for (int k = 0; k < 3; ++k) result[k] = ADJ ? result[k] - nee[k] : result[k] + nee[k];
                }
            }

            if (run) {
                if (did_scatter) {                                              // :221-230
                    (void) S.next_1d();
                    float ux = S.next_1d(), uy = S.next_1d();
                    ray.o = mei.p; ray.d = square_to_uniform_sphere(ux, uy); ray.maxt = kLargest;
                    last_pdf = kInvFourPi;
                }
                si = box_hit(P, ray.o, ray.d);                                  // :233-235
                ray.maxt = isfinite(si.t) ? si.t : kLargest;
                if (did_scatter && !si.valid) active = false;                   // :240-241
                if (did_escape) {                                               // :244-245
                    if (si.valid) ray.o = offset_p(si, ray.d);
                    escaped = true;
                }
                ++it;
            }
            }
            if (any_active) round_end(RECURSIVE ? 4 : 5, Jit);
            if constexpr (kWgc) {
                if (wgc_on && !wg_handoff<!RECURSIVE>(job, active, taken, ray, beta, result, S, depth, escaped, has_scattered, last_pdf, tl, home, it)) break;
            }
        }

        if constexpr (ADJ) {
            if (use_drt() && use_sub()) {                                       // :249-259, :756-760
                const bool rjob = job && r_depth >= 0;
                float adj[3] = { 0.0f, 0.0f, 0.0f };
                if (rjob) {
                    float dd = ((r_cw[0] + r_cw[1]) + r_cw[2]) / 3.0f;
                    float ws = ((r_wsum[0] + r_wsum[1]) + r_wsum[2]) / 3.0f;
#pragma unroll
                    for (int k = 0; k < 3; ++k) adj[k] = (dd != 0.0f ? (ws * r_cw[k]) / dd : 0.0f) * dL[k];
                }
                drt_backprop(rjob, A, r_ray, r_si_t, r_depth, adj);
            }
        } else {                                                                // :263-287
            if (job && !wgc_on) add_escaped_emission(escaped, depth, has_scattered, last_pdf, ray.d, beta, result);   // (hand-off mode: done in wg_handoff)
        }
        phase(RECURSIVE ? C_SC : C_TR);
        if (!RECURSIVE) iters = (uint32_t) it;                                  // (hand-off mode: written per ray by wg_handoff)
        out[0] = result[0]; out[1] = result[1]; out[2] = result[2];
    }
};

}  // namespace coop

}  // namespace drt
