// drt_super.hip -- the tracer for scenes with a majorant supergrid (majorant_resolution_factor > 0, the reference's
// default: python/scene_config.py:36, optimize.py:182-199): VolpathSimpleIntegrator.sample
// (python/integrators/volpathsimple.py:38-655), both AD modes.
//
// Why (measured): with a supergrid a tracking step is a free flight through 1..60 supergrid cells
// (3-D DDA; mean 12.6, bimodal) followed by one grid lookup, and a walk is ~25 cells but only 0.7..2 lookups: most of
// the work is cell stepping, in flights whose lengths have nothing to do with each other.  The one-ray-per-lane tracer
// ran it at 11.8 % VALU lane utilisation (every lane waits for the longest flight of the few lanes that have one), a
// state machine with one whole flight per step (drt_wavefront.hip) at 18 %, a state machine with one CELL per step in
// which every wave mixes cell steps, flight boundaries and path transitions at 23-30 % (lanes that wait for a batched
// block still occupy the wave's issue slots).
//
// Here the kinds of work are sorted INSIDE A COMPUTE UNIT, through LDS.  One workgroup per CU:
//   * the majorant supergrid lives in its LDS as bf16 (rounded UP when the grid is built - a majorant only has to bound -:
//     32^3 cells = 64 KiB for a 256^3 grid at factor 8) - the DDA never touches global memory;
//   * every lane holds one ray as a small state machine (registers) and owns one FLIGHT SLOT in LDS (12 words of DDA
//     state).  A wave does the irregular per-ray work in batches - "heavy runs": the collision a flight ended in (grid
//     lookup + the walk's acceptance / ratio / reservoir epilogue), path transitions (scatter / escape / emitter
//     sampling / end of path / next ray), the set-up of the next flight, which is then POSTED: slot written, bit set in
//     the wave's ready mask - and only when enough of its lanes have something to do;
//   * whenever it has too few such lanes, a wave WALKS instead: it pulls posted flights out of the ready masks - whoever
//     posted them - into its free walker lanes (whole masks with one 64-bit exchange, handed out by rank), steps them
//     through the supergrid DRT_SUPER_K cells at a time (~26 branch-free vector instructions per cell, majorants from
//     LDS), refilling lanes as flights end; a flight that ends leaves {entry distance, optical depth, majorant} of its last
//     cell in its slot and a bit in its owner's done mask.  When enough of the wave's own lanes are ready again, the
//     flights still under way are written back to their slots and posted again: nothing of a flight outlives the
//     walking block in registers, so walking costs the heavy code no registers, cell steps run on densely refilled
//     lanes whoever owns the flights, and lanes that wait for a heavy run cost no issue slots.
// Lanes that finish a ray pull the next one from their XCD's queue.
//
// Larger supergrids (the majorants do not fit next to the slots) keep their non-empty-cell bitmask in LDS and load the
// majorants of non-empty cells from L2 (template flag MGL = false).
//
// Arithmetic, random-number consumption and event counts are those of the scalar restatement (oracle/drt_oracle.c):
// radiance is bit-exact per ray, counters are equal; gradients differ by summation order only.  The adjoint emits its
// splats as records (drt_deferred.hip) and reads the path cache its primal pass wrote.  Not handled here (the host
// keeps the one-ray-per-lane kernels for them): quadratic DRT (use_drt && !use_drt_subsampling), the atomic gradient
// path, supergrids with more than 511 cells along an axis.
#include <atomic>
#include "drt_device.h"
#include "drt_launch.h"

#ifndef DRT_SUPER_THREADS
#define DRT_SUPER_THREADS 768      // threads per workgroup = per CU of the primal kernels: 12 waves, 3 per SIMD (168 registers)
#endif
#ifndef DRT_SUPER_THREADS_ADJ
#define DRT_SUPER_THREADS_ADJ 768  // ... of the adjoint kernels
#endif
#ifndef DRT_SUPER_K
#define DRT_SUPER_K 8              // cells per walker lane between two looks at the masks
#endif
#ifndef DRT_SUPER_K_ADJ
#define DRT_SUPER_K_ADJ 8          // ... in the adjoint kernels
#endif
#ifndef DRT_SUPER_LOOSE_ADJ
#define DRT_SUPER_LOOSE_ADJ 0      // unpredicated cell steps in the adjoint kernels too (measured: +4 % at 8 cells per look)
#endif
#ifndef DRT_SUPER_REFILL_MIN
#define DRT_SUPER_REFILL_MIN 16    // free walker lanes before more flights are pulled
#endif
#ifndef DRT_SUPER_REGEN_MIN
#define DRT_SUPER_REGEN_MIN 16     // finished lanes of a path wave before it takes new rays
#endif
#ifndef DRT_SUPER_HMIN
#define DRT_SUPER_HMIN 32          // lanes of a wave with something to do before it makes a heavy run (it walks otherwise); with the ray order: 16 / 24 / 32 / 40 / 52 -> 634 / 657 / 660 / 654 / 568 Msamples/s
#endif
#ifndef DRT_SUPER_BMIN
#define DRT_SUPER_BMIN 12          // lanes of a wave waiting for a path transition before the transition blocks run
#endif
#ifndef DRT_SUPER_MAXPOLL
#define DRT_SUPER_MAXPOLL 8        // ... or after this many polls with nothing to walk and at least one lane ready
#endif
#ifndef DRT_SUPER_EARLY_OUT
#define DRT_SUPER_EARLY_OUT 1      // flights that cannot collide (target optical depth > largest majorant x segment length) are not walked
#endif
#ifndef DRT_SUPER_CHUNK
#define DRT_SUPER_CHUNK 256        // queue positions a wave reserves per refill (64 / 128 / 256 / 512 / 1024: 616 / 652 / 670 / 659 / 471 Msamples/s; reservations that shrink towards the end of the queue: 530)
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
    PH_HEAD, PH_SCAT, PH_ESC, PH_NEE, PH_RT_END, PH_RTA_END, PH_PHASE, PH_END, PH_DRT_END,
    PH_IDLE, PH_DEAD
};
enum Flight : int { FL_NEW = 0, FL_NEXT = 1, FL_WAIT = 2 };   // first flight of a walk to set up | next flight to set up | posted
// flight slot: q0 = {tn.x, tn.y, tn.z, cell}, q1 = {td.x, td.y, td.z, steps left (9 bits per axis) + direction signs},
// q2 = {tau, tmax, t, acc}; a finished flight leaves its cell's majorant (0: left the segment) in q0.x
constexpr int kSlotWords = 12;

// LDS of one wave is accessed through volatile LDS pointers: other waves write it
typedef __attribute__((address_space(3))) volatile uint32_t lds_vu32;
typedef __attribute__((address_space(3))) volatile unsigned long long lds_vu64;
__device__ __forceinline__ void lds_fence() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }

__device__ __forceinline__ uint32_t xcc_id()
{
    return __builtin_amdgcn_s_getreg(20 | (0 << 6) | ((4 - 1) << 11)) & 7u;   // HW_REG_XCC_ID[3:0]
}

}  // namespace

template <bool ADJ, bool COUNT, bool ENV, bool MGL>
__global__ void __launch_bounds__(ADJ ? DRT_SUPER_THREADS_ADJ : DRT_SUPER_THREADS) trace_super_kernel(const Params P)
{
    constexpr int NWV = (ADJ ? DRT_SUPER_THREADS_ADJ : DRT_SUPER_THREADS) / 64;                              // waves = 64-lane groups of flight slots
    constexpr int NS = NWV * 64;
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
    // LDS: [flight slots: 3 x uint4 each][majorants as bf16 (MGL) | non-empty-cell bitmask][record state per wave]
    //      [ready masks: 2 words per wave][done masks][pull lists: 64 words per wave]
    const int n_cells = P.gx * P.gy * P.gz;
    const int mg_words = MGL ? (n_cells + 1) / 2 : P.mocc_words;
    uint4 *slot_lds = (uint4 *) lds;
    uint32_t *mg_lds = lds + kSlotWords * NS;
    uint32_t *rec_lds = mg_lds + ((mg_words + 3) & ~3);
    unsigned long long *ready_lds = (unsigned long long *) (rec_lds + NWV * 8);   // (8-byte aligned: every part is a multiple of 4 words)
    unsigned long long *done_lds = ready_lds + NWV;
    uint32_t *list_lds = (uint32_t *) (done_lds + NWV);
    uint32_t *mmax_lds = list_lds + NWV * 64;                                // bits of the largest majorant of the supergrid
    if (threadIdx.x == 0) *mmax_lds = MGL ? 0u : 0x7f800000u;                // (majorants in global memory: not scanned, no bound)
    __syncthreads();
    if constexpr (MGL) {                                                     // (the grid's values are bf16-representable: exact)
        uint32_t top = 0u;                                                   // (non-negative floats order like their bit patterns)
        for (int w = threadIdx.x; w < mg_words; w += blockDim.x) {
            const uint32_t a = __float_as_uint(P.mgrid[2 * w]), b = 2 * w + 1 < n_cells ? __float_as_uint(P.mgrid[2 * w + 1]) : 0u;
            mg_lds[w] = (a >> 16) | (b & 0xffff0000u);
            top = max(top, max(a, b));
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) top = max(top, (uint32_t) __shfl_down((int) top, off, 64));
        if ((threadIdx.x & 63u) == 0u && top) atomicMax(mmax_lds, top);
    } else {
        for (int w = threadIdx.x; w < mg_words; w += blockDim.x) mg_lds[w] = P.mocc[w];
    }
    for (int w = threadIdx.x; w < NWV * 12; w += blockDim.x) rec_lds[w] = 0;  // record state; ready + done masks
    __syncthreads();
    // A flight whose target optical depth exceeds (largest majorant) x (length of its segment) cannot end in a collision
    // whatever cells it crosses: it is not walked (flight set-up below).  The walk's sums are bounded rigorously: with
    // e = 2^-24, acc_N <= mmax * sum_k (texit_k - t_k) * (1 + e)^(N + 2) <= mmax * tmax * (1 + e)^(N + 2), N < 3 * 512 cells,
    // against the factor 1.001 below.  Optically thin media - every optimisation starts from one (scene_config.py:117,167,
    // 221: sigma_t = 0.04) - lose most of their cell steps this way; results, draws and counters are unchanged.
    const float mmax = __uint_as_float(__builtin_amdgcn_readfirstlane((int) *mmax_lds));
    const uint32_t *occ = nullptr;   // (tentative collisions lie in non-empty supergrid cells: the voxel bitmask would rarely say "empty")

    const uint32_t lane = threadIdx.x & 63u;
    const int wave = __builtin_amdgcn_readfirstlane((int) (threadIdx.x >> 6));   // (wave-uniform: tell the compiler, or every mask operation below becomes vector code)
    lds_vu64 *ready = (lds_vu64 *) ready_lds, *done = (lds_vu64 *) done_lds;
    lds_vu32 *list = (lds_vu32 *) list_lds + wave * 64;
    const uint16_t *mg16 = (const uint16_t *) mg_lds;
    uint32_t cnt[C_COUNT];
#pragma unroll
    for (int i = 0; i < C_COUNT; ++i) cnt[i] = 0;
#if DRT_SUPER_PROFILE
    // experiment build (tools/mk_variant.sh NAME -DDRT_SUPER_PROFILE=1): the counting kernels' slots hold, summed over waves,
    // [1] polls, [2] heavy runs, [3] lanes with something to do when the wave ran, [4] lanes served by the flight epilogue,
    // [5] lanes that posted a flight, [8] passes of the transition block, [6] pulls, [7] flights pulled, [0] cell steps taken
#define DRT_COUNT(slot) do { } while (0)
#define DRT_PROF(slot, v) do { if (DRT_SUPER_PROFILE == 1 && COUNT) { const uint32_t v_ = (uint32_t) (v); if (lane == 0) cnt[slot] += v_; } } while (0)
    // DRT_SUPER_PROFILE=4: the walker's and the transition blocks' occupancy - [0] lane cell steps, [1] wave cell steps, [2] batches,
    // [3] lanes flying at batch start, [4] flights finished, [5] lanes in a transition phase at pass start, [6] transition passes,
    // [7] regeneration blocks, [8] rays started
#define DRT_PROF4(slot, v) do { if (DRT_SUPER_PROFILE == 4 && COUNT) { const uint32_t v_ = (uint32_t) (v); if (lane == 0) cnt[slot] += v_; } } while (0)
    // DRT_SUPER_PROFILE=2: shader clock (units of 64 cycles) per wave spent - [1] polling / sleeping, [3] flight epilogue,
    // [4] regeneration, [5] transitions, [8] flight set-up, [6] pulling flights, [7] cell steps + write-back; [2] heavy runs
    uint64_t pt_last = __builtin_readcyclecounter();
#if DRT_SUPER_PROFILE == 3
    const unsigned long long pt_start = wall_clock64(); unsigned long long pt_drained = 0;
#endif
#define DRT_STAMP(slot) do { if (DRT_SUPER_PROFILE == 2 && COUNT) { const uint64_t t_ = __builtin_readcyclecounter(); if (lane == 0) cnt[slot] += (uint32_t) ((t_ - pt_last) >> 6); pt_last = t_; } } while (0)
#else
#define DRT_STAMP(slot) do { } while (0)
#define DRT_COUNT(slot) do { if (COUNT) cnt[slot]++; } while (0)
#define DRT_PROF(slot, v) do { } while (0)
#define DRT_PROF4(slot, v) do { } while (0)
#endif

    // uniform supergrid constants
    const int gx = P.gx, gy = P.gy, gz = P.gz;
    const float fgx = (float) gx, fgy = (float) gy, fgz = (float) gz;
    const int lin_y = gx, lin_z = gx * gy;

    const int pw = wave;
    const uint32_t my_slot = threadIdx.x;             // = 64 * wave + lane
    uint32_t *rec = rec_lds + pw * 8;                                           // record-stream state of this wave (emit_record)
    const uint32_t xcc = xcc_id();
    // (with a ray order the queue positions cover whole units: the last unit may reach past the launch's last ray)
    const uint64_t span = P.order ? (uint64_t) P.order_units * P.order_unit : P.n_rays - P.ray_first;
    const uint64_t n_runs = (span + DRT_SUPER_RUN - 1) / DRT_SUPER_RUN;
    // queue x serves the runs x, x + 8, ...; a wave starts on the queue of the XCD it runs on (L2 locality) and moves on
    // to the next ones when that one is drained: every ray is traced whatever the placement of the workgroups
    uint32_t qsel = 0, qx = xcc;
    uint64_t my_len = (n_runs > qx ? (n_runs - qx + 7) / 8 : 0) * DRT_SUPER_RUN;
    unsigned long long *queue = P.queues + qx;
    uint64_t pool_next = 0, pool_end = 0;      // wave-uniform: this wave's reserved queue positions
    int rr = wave;                             // wave whose ready mask is looked at first when pulling flights

    // ---- per-lane state ------------------------------------------------------------------------------
    // (registers are what limits this kernel's occupancy: state that is dead in some phase carries another phase's values)
    int ph = PH_IDLE, fl = FL_NEW;
    // (the compiler keeps these as lane masks in scalar registers: packing them into a vector register costs more)
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
    int polls = 0;                  // (wave-uniform) polls since this wave last ran

    for (;;) {
        // ---- what can this wave do now? ---------------------------------------------------------------------
        const bool walking = ph < PH_HEAD;
        const unsigned long long dword = done[pw];
        const bool back = walking && fl == FL_WAIT && ((dword >> lane) & 1ull) != 0ull;   // my flight's result is in my slot
        // (finished lanes take new rays DRT_SUPER_REGEN_MIN at a time - the ray prologue is long - or when nothing else is left)
        const uint64_t m_idle = __ballot(ph == PH_IDLE);
        const bool regen_ok = __popcll(m_idle) >= DRT_SUPER_REGEN_MIN || !__ballot(ph < PH_IDLE);
        // (path transitions are a dozen short blocks, each for the few lanes in its phase: they run when DRT_SUPER_BMIN
        //  lanes wait for one, or when no flight of this wave is under way any more)
        const uint64_t m_wait = __ballot(walking && fl == FL_WAIT && !back);
        const uint64_t m_tr = __ballot(ph >= PH_HEAD && ph < PH_IDLE);
        const bool trans_ok = __popcll(m_tr) >= DRT_SUPER_BMIN || !m_wait;
        const bool can = back || (walking && fl != FL_WAIT) || (trans_ok && ph >= PH_HEAD && ph < PH_IDLE) || (regen_ok && ph == PH_IDLE);
        const uint64_t m_can = __ballot(can);
        if (!m_can && !m_wait) break;                                            // every lane is dead
        if (__popcll(m_can) < DRT_SUPER_HMIN) {
            // ================= walk: posted flights -> supergrid cells -> results ==============================
            // The wave turns into a walker until enough of its own lanes have something to do: flights are pulled
            // out of the ready masks (whoever posted them) into its free lanes, stepped DRT_SUPER_K cells at a time,
            // and leave a result + a bit in their owner's done mask when they end; on the way out the flights that
            // are still under way are written back to their slots and posted again.  Nothing of it outlives this block.
            bool fly = false, walked = false;
            uint32_t slot = my_slot;
            float tnx = kInf, tny = kInf, tnz = kInf, tdx = kInf, tdy = kInf, tdz = kInf, t = 0.0f, acc = 0.0f, tau = 0.0f, tmax = 0.0f;
            int cell = 0, sx = 0, sy = 0, sz = 0;
            uint32_t rem = 0;
            for (;;) {
                const uint64_t flym = __ballot(fly);
                const int nfree = 64 - __popcll(flym);
                if (nfree >= DRT_SUPER_REFILL_MIN) {
                    const unsigned long long peek = lane < (uint32_t) NWV ? ready[lane] : 0ull;
                    const uint64_t havem = __ballot(peek != 0ull);
                    if (havem) {
                        // Choose whole masks, starting at wave `rr`, while they fit into the free lanes (the peeked values are
                        // hints), take them with ONE exchange (lane w takes wave w's), hand the flights out by rank; what does
                        // not fit after all (bits that arrived between the look and the exchange) is put back.
                        const int np = __popcll(peek);
                        uint64_t choose = 0;
                        int room = nfree;
                        for (int i = 0; i < NWV; ++i) {
                            const int w = rr + i < NWV ? rr + i : rr + i - NWV;
                            if (!((havem >> w) & 1ull)) continue;
                            const int n = __builtin_amdgcn_readlane(np, w);
                            if (choose && n > room) continue;
                            choose |= 1ull << w; room -= n;
                            if (room <= 0) break;
                        }
                        rr = rr + 1 < NWV ? rr + 1 : 0;
                        unsigned long long bits = 0ull;
                        if ((choose >> lane) & 1ull) bits = atomicExch(ready_lds + lane, 0ull);
                        int got = 0;
                        for (uint64_t c = choose; c; c &= c - 1ull) {
                            const int w = __ffsll((long long) c) - 1;
                            const uint64_t b = ((uint64_t) (uint32_t) __builtin_amdgcn_readlane((int) (uint32_t) (bits >> 32), w) << 32) |
                                               (uint32_t) __builtin_amdgcn_readlane((int) (uint32_t) bits, w);
                            if (!b) continue;
                            const bool mine = ((b >> lane) & 1ull) != 0ull;
                            const int rank = __popcll(b & ((1ull << lane) - 1ull));
                            const bool keep = mine && got + rank < nfree;
                            const uint64_t kept = __ballot(keep), putback = b & ~kept;
                            if (putback && lane == 0) atomicOr(ready_lds + w, (unsigned long long) putback);
                            if (keep) list[got + rank] = (uint32_t) (w * 64) + lane;
                            got += __popcll(kept);
                        }
                        if (got) {
                            DRT_PROF(6, 1); DRT_PROF(7, got);
                            lds_fence();
                            const int frank = __popcll(~flym & ((1ull << lane) - 1ull));   // my rank among the free lanes
                            if (!fly && frank < got) {
                                slot = list[frank];
                                const uint4 *sp = slot_lds + 3 * slot;
                                const uint4 q0 = sp[0], q1 = sp[1], q2 = sp[2];
                                tnx = __uint_as_float(q0.x); tny = __uint_as_float(q0.y); tnz = __uint_as_float(q0.z); cell = (int) q0.w;
                                tdx = __uint_as_float(q1.x); tdy = __uint_as_float(q1.y); tdz = __uint_as_float(q1.z); rem = q1.w;
                                tau = __uint_as_float(q2.x); tmax = __uint_as_float(q2.y); t = __uint_as_float(q2.z); acc = __uint_as_float(q2.w);
                                sx = (rem & (1u << 27)) ? -1 : 1; sy = (rem & (1u << 28)) ? -lin_y : lin_y; sz = (rem & (1u << 29)) ? -lin_z : lin_z;
                                fly = true;
                            }
                        }
                    }
                }
                DRT_STAMP(6);
                if (!__ballot(fly)) break;                                       // nothing to walk (any more)
                walked = true;
                // kLoose (primal kernels: measured -3 %, the adjoint kernels +4 %): the steps are not predicated on `fly` - a
                // lane whose flight ended, or that has none, keeps stepping its private registers (garbage that is never written
                // back; its cell index is clamped for the LDS read), and what a flight leaves behind is captured in the step it
                // ends in: fewer mask operations per step.
                constexpr bool kLoose = !ADJ || DRT_SUPER_LOOSE_ADJ;
                bool fin = false; float res_mc = 0.0f, res_t = 0.0f, res_acc = 0.0f;
#pragma unroll
                for (int k = 0; k < (ADJ ? DRT_SUPER_K_ADJ : DRT_SUPER_K); ++k) {
#if DRT_SUPER_PROFILE == 1 || DRT_SUPER_PROFILE == 4
                    { const int nf = __popcll(__ballot(fly)); DRT_PROF(0, nf); DRT_PROF4(0, nf); DRT_PROF4(1, 1); if (k == 0) { DRT_PROF4(2, 1); DRT_PROF4(3, nf); } }
#endif
                    // one supergrid cell (oracle: the loop of sample_collision).  Crossing times are finite or +inf, never NaN.
                    const float tmin = fminf(fminf(tnx, tny), tnz);
                    const float texit = fminf(tmin, tmax);
                    const uint32_t ci = kLoose ? min((uint32_t) cell, (uint32_t) (n_cells - 1)) : (uint32_t) cell;
                    float mc;
                    if constexpr (MGL) mc = __uint_as_float((uint32_t) mg16[ci] << 16);
                    else mc = ((!kLoose || fly) && ((mg_lds[ci >> 5] >> (ci & 31u)) & 1u)) ? P.mgrid[ci] : 0.0f;
                    const float nacc = acc + mc * (texit - t);                  // (an empty cell adds an exact zero)
                    const bool hit = mc > 0.0f && nacc >= tau;                  // the tentative collision lies in this cell
                    const bool isx = tnx == tmin, isy = !isx && tny == tmin;     // first axis with the earliest crossing
                    const uint32_t sh = isx ? 0u : isy ? 9u : 18u;
                    const bool end = !(texit < tmax) || ((rem >> sh) & 511u) == 0u;   // end of the segment / of the grid
                    const float tnn = tmin + (isx ? tdx : isy ? tdy : tdz);
                    if constexpr (kLoose) {
                        const bool ends = fly && (hit || end);
                        res_mc = ends ? (hit ? mc : 0.0f) : res_mc; res_t = ends ? t : res_t; res_acc = ends ? acc : res_acc;
                        fin = fin || ends; fly = fly && !ends;
                        acc = nacc; t = texit;
                        rem -= 1u << sh;
                        cell += isx ? sx : isy ? sy : sz;
                        tnx = isx ? tnn : tnx; tny = isy ? tnn : tny; tnz = (isx || isy) ? tnz : tnn;
                    } else {
                        if (fly && (hit || end)) { fin = true; res_mc = hit ? mc : 0.0f; fly = false; }
                        const bool go = fly;
                        acc = go ? nacc : acc;
                        t = go ? texit : t;
                        rem = go ? rem - (1u << sh) : rem;
                        cell += go ? (isx ? sx : isy ? sy : sz) : 0;
                        tnx = (go && isx) ? tnn : tnx; tny = (go && isy) ? tnn : tny; tnz = (go && !isx && !isy) ? tnn : tnz;
                    }
                }
                DRT_PROF4(4, __popcll(__ballot(fin)));
                if (__ballot(fin)) {
                    // result: where the last cell was entered, the optical depth up to there, its majorant (0: left the segment)
                    if (fin) {
                        uint4 *sp = slot_lds + 3 * slot;
                        sp[0].x = __float_as_uint(res_mc);
                        sp[2].z = __float_as_uint(kLoose ? res_t : t); sp[2].w = __float_as_uint(kLoose ? res_acc : acc);
                    }
                    lds_fence();
                    if (fin) atomicOr(done_lds + (slot >> 6), 1ull << (slot & 63u));
                }
                DRT_STAMP(7);
                // enough of my own lanes ready by now?  (results of my flights, walked by me or by others)
                const unsigned long long dw = done[pw];
                const bool bk = ph < PH_HEAD && fl == FL_WAIT && ((dw >> lane) & 1ull) != 0ull;
                if (__popcll(m_can | __ballot(bk)) >= DRT_SUPER_HMIN) break;
            }
            if (__ballot(fly)) {                                                 // flights still under way: back to their slots
                if (fly) {
                    uint4 *sp = slot_lds + 3 * slot;
                    sp[0] = make_uint4(__float_as_uint(tnx), __float_as_uint(tny), __float_as_uint(tnz), (uint32_t) cell);
                    sp[1].w = rem;
                    sp[2].z = __float_as_uint(t); sp[2].w = __float_as_uint(acc);
                }
                lds_fence();
                if (fly) atomicOr(ready_lds + (slot >> 6), 1ull << (slot & 63u));
            }
            DRT_STAMP(7);
            if (walked) { polls = 0; continue; }
            // nothing to walk
            if (!m_can || (m_wait && polls < DRT_SUPER_MAXPOLL)) {              // results are on their way: wait for them
                ++polls;
                DRT_PROF(1, 1);
                __builtin_amdgcn_s_sleep(2);
                DRT_STAMP(1);
                continue;
            }
        }
        polls = 0;
        DRT_STAMP(1);
        if (DRT_SUPER_PROFILE == 2 && COUNT && lane == 0) cnt[2] += 1;
        DRT_PROF(2, 1); DRT_PROF(3, __popcll(m_can));

        // ================= (Fe) the collision a flight ended in =========================================
        {
            const uint64_t m_back = __ballot(back);
            if (m_back) {
                DRT_PROF(4, __popcll(m_back));
                if (lane == 0) atomicAnd(done_lds + pw, (unsigned long long) ~m_back);   // results consumed: clear their bits
                if (back) {
                    const bool drt = ph == PH_DRT;
                    const bool useA = ADJ && !rec_mode && drt;
                    Pcg32 R; R.state = useA ? A.state : S.state; R.inc = useA ? A.inc : S.inc;
                    // Medium::sample_interaction [M3-ext] (oracle: sample_collision): the walker left {entry distance of the
                    // last cell, optical depth up to there, that cell's majorant (0: the flight left the segment)}
                    const uint4 q2 = slot_lds[3 * my_slot + 2];
                    const float lm = __uint_as_float(slot_lds[3 * my_slot].x), tau = __uint_as_float(q2.x);
                    const float c_t = __uint_as_float(q2.z), c_acc = __uint_as_float(q2.w);
                    const float lim = lm > 0.0f ? 1.0f / lm : 0.0f;
                    const float dt = lm > 0.0f ? fmaf(tau - c_acc, lim, c_t) : kInf;
                    bool inside; V3 p;
                    if (drt) { const float tm = wmax - wt; (void) tm; wt += dt; inside = wt <= wmax; p = ray_at(ro, rd, wt); }
                    else { inside = dt <= wmax; p = ray_at(wo, rd, dt); }
                    const float sig = inside ? eval_sigma_t(P, p, occ) : 0.0f;
                    fl = FL_NEXT;
                    if (!inside) {                                              // left the segment
                        ph = drt ? PH_DRT_END : (ph == PH_DT) ? PH_ESC : (ph == PH_RT ? PH_RT_END : PH_RTA_END);
                    } else if (drt) {                                           // Medium::sample_interaction_drt (:549-551); wo = {T, wsum, selected t}
                        DRT_COUNT(C_DRT);
                        const float w = wo.x * lim;
                        wo.y += w;
                        const float u2 = R.next_1d();
                        if (w > 0.0f && u2 * wo.y <= w) wo.z = wt;
                        wo.x *= (lm - sig) * lim;
                        if (wo.x == 0.0f) ph = PH_DRT_END;
                    } else if (ph == PH_DT) {                                   // :348-367
                        DRT_COUNT(C_DT); ++pc_steps;
                        const float r = sig * lim;
                        const float u2 = R.next_1d();
                        if (!(u2 >= r)) { wt = wt + dt; ph = PH_SCAT; }          // mei.t
                        else { wo = p; wmax -= dt; wt += dt; }
                    } else {                                                    // ratio tracking :465-502
                        DRT_COUNT(C_RT); ++pc_steps;
                        const float tr = (lm - sig) * lim;
                        if constexpr (ADJ) {
                            if (ph == PH_RTA && tr > 0.0f) {                    // :487-492
                                splat_sigma_t<true>(P, p, -(adjsum * lim) / tr, rec);
                                DRT_COUNT(C_RT_ADJ);
                            }
                        }
                        wt *= tr; wo = p; wmax -= dt;
                        if (wt == 0.0f) ph = (ph == PH_RT) ? PH_RT_END : PH_RTA_END;
                    }
                    if (useA) A.state = R.state; else S.state = R.state;
                }
            }
        }

        DRT_STAMP(3);
        // ================= (A) regeneration ===========================================================
        // Ray indices come from a wave-local pool refilled DRT_SUPER_CHUNK at a time with ONE returning atomic on the
        // XCD's queue head.
        {
            const uint64_t wmask = m_idle;
            if (wmask && regen_ok) {
                DRT_PROF4(7, 1); DRT_PROF4(8, __popcll(wmask));
                while (pool_next >= pool_end && qsel < 8) {                      // refill (wave-uniform)
                    const int leader = __ffsll((long long) wmask) - 1;
                    unsigned long long base = 0;
                    constexpr unsigned long long chunk = DRT_SUPER_CHUNK;
                    if ((int) lane == leader) base = atomicAdd(queue, (unsigned long long) DRT_SUPER_CHUNK);
                    base = ((unsigned long long)(unsigned int) __shfl((int)(base >> 32), leader, 64) << 32)
                         | (unsigned int) __shfl((int) base, leader, 64);
                    if (base < my_len) { pool_next = base; pool_end = base + chunk; }
                    else {                                                       // this queue is drained: next one
                        ++qsel;
                        qx = (xcc + qsel) & 7u;
                        my_len = (n_runs > qx ? (n_runs - qx + 7) / 8 : 0) * DRT_SUPER_RUN;
                        queue = P.queues + qx;
                    }
                }
                const bool drained = qsel >= 8;
#if DRT_SUPER_PROFILE == 3
                if (drained && !pt_drained) pt_drained = wall_clock64();
#endif
                const uint64_t q = pool_next + (uint64_t) __popcll(wmask & ((1ull << lane) - 1ull));
                const bool take = (ph == PH_IDLE) && !drained && q < pool_end;
                if (drained && ph == PH_IDLE) ph = PH_DEAD;                      // all eight queues are empty
                pool_next += (uint64_t) __popcll(wmask);
                if (pool_next > pool_end) pool_next = pool_end;
                if (take && q < my_len) {
                    uint64_t i = ((q / DRT_SUPER_RUN) * 8 + qx) * DRT_SUPER_RUN + (q % DRT_SUPER_RUN);
                    if (P.order) {                                              // position -> unit of the order -> ray
                        const uint32_t g = (uint32_t) i, u = P.order_unit == 1u ? g : g / P.order_unit;
                        i = i < span ? (uint64_t) P.order[u] * P.order_unit + (g - u * P.order_unit) : P.n_rays;
                    }
                    i += P.ray_first;
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

        DRT_STAMP(4);
        // ================= (B) path transitions ==========================================================
        if (trans_ok && __ballot(ph >= PH_HEAD && ph < PH_IDLE)) {
            // A pass takes every waiting lane to its next walk (or to the end of its ray); lanes whose walk comes
            // out of the path cache (adjoint pass) go round once more.
            do {
                DRT_PROF(8, 1); DRT_PROF4(6, 1); DRT_PROF4(5, __popcll(__ballot(ph >= PH_HEAD && ph < PH_IDLE)));
                uint4 pce1 = make_uint4(0u, 0u, 0u, 0u); bool pce1_ok = false;   // this iteration's NEE entry of the path cache, read at the loop head
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
                        const uint4 *pce = P.path_cache + ((size_t) li * P.path_cache_cap + pc_it) * 2;
                        const uint4 e = pce[0];
                        pce1 = pce[1]; pce1_ok = true;                          // (adjacent: one round trip for both)
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
                        uint4 e = pce1;
                        if (!pce1_ok) e = P.path_cache[((size_t) li * P.path_cache_cap + pc_it) * 2 + 1];
                        wt = __uint_as_float(e.x);
                        S.state = ((uint64_t) e.z << 32) | e.y;
                        if (COUNT && !DRT_SUPER_PROFILE) cnt[C_RT] += e.w;
                        ph = PH_RT_END;
                    } else if (h.valid) { wo = ro; wmax = h.t; wt = 1.0f; ph = PH_RT; fl = FL_NEW; }
                    else { wt = 0.0f; ph = PH_RT_END; }
                }
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
            } while (__ballot(ph >= PH_HEAD && ph < PH_IDLE));
        }

        DRT_STAMP(5);
        // ================= (Fs) the next flight: set up and post =========================================
        {
            const bool setup = ph < PH_HEAD && fl != FL_WAIT;
            const uint64_t m_set = __ballot(setup);
            if (m_set) {
                DRT_PROF(5, __popcll(m_set));
                bool posted = false;
                if (setup) {
                    const bool drt = ph == PH_DRT;
                    const bool useA = ADJ && !rec_mode && drt;
                    Pcg32 R; R.state = useA ? A.state : S.state; R.inc = useA ? A.inc : S.inc;
                    // the direction's share of the DDA (oracle: sample_collision): crossing-time increments 1 / |dg|, direction
                    // signs.  It is the same for every flight of a walk: kept in the slot, recomputed for a walk's first flight.
                    float tdx, tdy, tdz; int sgx, sgy, sgz;
                    if (fl == FL_NEW) {
                        const float dgx = (rd.x * P.inv_ext[0]) * fgx, dgy = (rd.y * P.inv_ext[1]) * fgy, dgz = (rd.z * P.inv_ext[2]) * fgz;
                        if (dgx >= 1e-20f) { tdx = 1.0f / dgx; sgx = 1; } else if (dgx <= -1e-20f) { tdx = 1.0f / -dgx; sgx = -1; } else { tdx = kInf; sgx = 0; }
                        if (dgy >= 1e-20f) { tdy = 1.0f / dgy; sgy = 1; } else if (dgy <= -1e-20f) { tdy = 1.0f / -dgy; sgy = -1; } else { tdy = kInf; sgy = 0; }
                        if (dgz >= 1e-20f) { tdz = 1.0f / dgz; sgz = 1; } else if (dgz <= -1e-20f) { tdz = 1.0f / -dgz; sgz = -1; } else { tdz = kInf; sgz = 0; }
                    } else {
                        const uint4 q1 = slot_lds[3 * my_slot + 1];             // (the walk's previous flight left them there)
                        tdx = __uint_as_float(q1.x); tdy = __uint_as_float(q1.y); tdz = __uint_as_float(q1.z);
                        sgx = tdx == kInf ? 0 : (q1.w & (1u << 27)) ? -1 : 1;
                        sgy = tdy == kInf ? 0 : (q1.w & (1u << 28)) ? -1 : 1;
                        sgz = tdz == kInf ? 0 : (q1.w & (1u << 29)) ? -1 : 1;
                    }
                    const float u = R.next_1d();
                    const float tau = -drt_logf(1.0f - u);
                    const V3 o = drt ? ray_at(ro, rd, wt) : wo;
                    const float tmax = drt ? wmax - wt : wmax;
                    const float gxf = ((o.x - P.bmin[0]) * P.inv_ext[0]) * fgx;
                    const float gyf = ((o.y - P.bmin[1]) * P.inv_ext[1]) * fgy;
                    const float gzf = ((o.z - P.bmin[2]) * P.inv_ext[2]) * fgz;
                    const float flx = fminf(fmaxf(floorf(gxf), 0.0f), (float) (gx - 1));
                    const float fly = fminf(fmaxf(floorf(gyf), 0.0f), (float) (gy - 1));
                    const float flz = fminf(fmaxf(floorf(gzf), 0.0f), (float) (gz - 1));
                    const int cx = (int) flx, cy = (int) fly, cz = (int) flz;
                    const float tnx = sgx > 0 ? ((flx + 1.0f) - gxf) * tdx : sgx < 0 ? (gxf - flx) * tdx : kInf;
                    const float tny = sgy > 0 ? ((fly + 1.0f) - gyf) * tdy : sgy < 0 ? (gyf - fly) * tdy : kInf;
                    const float tnz = sgz > 0 ? ((flz + 1.0f) - gzf) * tdz : sgz < 0 ? (gzf - flz) * tdz : kInf;
                    const uint32_t rx_ = (uint32_t) (sgx > 0 ? gx - 1 - cx : cx), ry_ = (uint32_t) (sgy > 0 ? gy - 1 - cy : cy),
                                   rz_ = (uint32_t) (sgz > 0 ? gz - 1 - cz : cz);
                    const uint32_t rem = rx_ | (ry_ << 9) | (rz_ << 18) | (sgx < 0 ? 1u << 27 : 0u) | (sgy < 0 ? 1u << 28 : 0u) | (sgz < 0 ? 1u << 29 : 0u);
                    if (DRT_SUPER_EARLY_OUT && tau > (mmax * tmax) * 1.001f) {
                        // no cell of this segment can bring the optical depth to tau: the flight leaves the segment, as
                        // the epilogue above finds it after a walk (majorant 0 in the slot: dt = inf, not inside)
                        if (drt) wt += kInf;
                        fl = FL_NEXT;
                        ph = drt ? PH_DRT_END : (ph == PH_DT) ? PH_ESC : (ph == PH_RT ? PH_RT_END : PH_RTA_END);
                    } else {
                        uint4 *sp = slot_lds + 3 * my_slot;
                        sp[0] = make_uint4(__float_as_uint(tnx), __float_as_uint(tny), __float_as_uint(tnz), (uint32_t) ((cz * gy + cy) * gx + cx));
                        sp[1] = make_uint4(__float_as_uint(tdx), __float_as_uint(tdy), __float_as_uint(tdz), rem);
                        sp[2] = make_uint4(__float_as_uint(tau), __float_as_uint(tmax), 0u, 0u);
                        fl = FL_WAIT;
                        posted = true;
                    }
                    if (useA) A.state = R.state; else S.state = R.state;
                }
                const uint64_t m_post = __ballot(posted);
                lds_fence();                                                     // the slots are written ...
                if (lane == 0 && m_post) atomicOr(ready_lds + pw, (unsigned long long) m_post);    // ... before the flights are posted
            }
        }
        DRT_STAMP(8);
    }

    if constexpr (ADJ) close_records(P, rec);
#if DRT_SUPER_PROFILE == 3
    // experiment build: when do the waves run dry, when do they end (100 MHz clock; slots: [0] 2^62 - first start, [1] 2^62 -
    // first end, [2] last end, [3] sum of (end - start), [4] waves, [5] sum of (queues dry - start), [6] 2^62 - first dry)
    if (COUNT && lane == 0) {
        const unsigned long long te = wall_clock64(), q62 = 1ull << 62;
        if (!pt_drained) pt_drained = te;
        atomicMax(P.counters + 0, q62 - pt_start); atomicMax(P.counters + 1, q62 - te); atomicMax(P.counters + 2, te);
        atomicAdd(P.counters + 3, te - pt_start); atomicAdd(P.counters + 4, 1ull);
        atomicAdd(P.counters + 5, pt_drained - pt_start); atomicMax(P.counters + 6, q62 - pt_drained);
    }
    if (false) {
#else
    if (COUNT) {
#endif
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
#undef DRT_PROF4
#undef DRT_STAMP
}

// LDS bytes of a launch; 0: this supergrid cannot be served (the host keeps the one-ray-per-lane kernels)
static size_t super_lds_bytes(const Params &P, bool adjoint, bool &mgl)
{
    const size_t cells = (size_t) P.gx * P.gy * P.gz;
    const size_t nwv = (adjoint ? DRT_SUPER_THREADS_ADJ : DRT_SUPER_THREADS) / 64;
    // (the kernel's layout: flight slots, record state, ready + done masks, pull lists)
    const size_t fixed = ((size_t) kSlotWords * 64 * nwv + nwv * 8 + 4 * nwv + 64 * nwv + 4) * 4;   // (+ the largest majorant)
    const size_t limit = 160u * 1024u;
    mgl = ((((cells + 1) / 2) + 3) & ~(size_t) 3) * 4 + fixed <= limit;
    const size_t words = mgl ? (cells + 1) / 2 : (size_t) P.mocc_words;
    const size_t need = ((words + 3) & ~(size_t) 3) * 4 + fixed;
    return need <= limit ? need : 0;
}

bool super_supported(const Params &P)
{
    bool mgl;
    return P.mgrid && P.mocc && P.gx <= 511 && P.gy <= 511 && P.gz <= 511 && super_lds_bytes(P, false, mgl) != 0 && super_lds_bytes(P, true, mgl) != 0;
}

hipError_t launch_trace_super(const Params &P, bool adjoint, bool count, int n_cus, hipStream_t stream)
{
    if (P.n_rays <= P.ray_first) return hipSuccess;
    bool mgl = false;
    const size_t lds = super_lds_bytes(P, adjoint, mgl);
    if (!lds) return hipErrorInvalidValue;
    // one workgroup per CU when the majorants live in LDS; with the bitmask only, as many as fit
    const unsigned threads = adjoint ? DRT_SUPER_THREADS_ADJ : DRT_SUPER_THREADS;
    unsigned blocks = (unsigned) n_cus;                                         // one workgroup per CU
    const uint64_t need = (P.n_rays - P.ray_first + 63) / 64;                  // no more path waves than 64-ray groups
    const uint64_t waves_per_block = threads / 64;
    if ((need + waves_per_block - 1) / waves_per_block < blocks) blocks = (unsigned) ((need + waves_per_block - 1) / waves_per_block);
    dim3 block(threads), grid(blocks);
    const bool env = P.env_pix != nullptr;
    hipError_t e = hipSuccess;
#define DRT_SUPER_LAUNCH(A, C, E, M)                                                                              \
    do {                                                                                                          \
        auto kern = trace_super_kernel<A, C, E, M>;                                                               \
        /* (the attribute belongs to the function ON A DEVICE: remembered per device; the kernels of one handle are \
            launched from one host thread, handles on different devices keep different entries) */                 \
        static std::atomic<size_t> lds_set[64];                                                                        \
        int dev_ = 0;                                                                                             \
        if (hipGetDevice(&dev_) != hipSuccess || dev_ < 0 || dev_ >= 64) dev_ = 63;                               \
        if (lds > lds_set[dev_] || dev_ == 63) {                                                                  \
            e = hipFuncSetAttribute((const void *) kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds); \
            if (e != hipSuccess) return e;                                                                        \
            lds_set[dev_] = lds;                                                                                  \
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
