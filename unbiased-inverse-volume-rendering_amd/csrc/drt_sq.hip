// drt_sq.hip -- the tracer for scenes with a majorant supergrid (majorant_resolution_factor > 0, the reference's default:
// python/scene_config.py:36, optimize.py:182-199), round 4: VolpathSimpleIntegrator.sample
// (python/integrators/volpathsimple.py:38-655), both AD modes, as WORK QUEUES INSIDE A COMPUTE UNIT.
//
// Why (measured): the round-3 tracer (drt_super.hip) keeps every ray in the registers of the lane that
// owns it.  Flights travel to whichever wave walks them, but everything else of a ray - the collision a flight ended in,
// the next flight's set-up, the path transitions, the ray prologue - can only run on the owner lane, so those blocks run
// when "enough" of a wave's 64 lanes happen to be ready: 23-28 of 64 in the collision / set-up blocks, fewer in the
// transition blocks, and every block of a heavy run is issued if a single lane needs it.  78 % of that kernel's vector
// instructions are such heavy runs; cell stepping is 22 %.
//
// Here a ray lives in a RECORD and belongs to no lane: 112 bytes in LDS (its flight + what the collision / flight set-up
// code needs: direction, moving origin, the active generator, flags) and 48 (adjoint: 144) bytes in global memory
// (Params::sq_cold, per workgroup, L2-resident: what only the path transitions touch - throughput, radiance, reservoir, dL).
// As many records as fit LDS next to the majorants (up to DRT_SQ_MAX_RAYS; 768 with a 32^3 supergrid): measured, the
// number of rays a compute unit holds is what the speed of this tracer follows (256 / 512 / 768 records: 5.45 / 3.35 / 2.89 ms
// primal).  Ring buffers of ray ids - flights to walk, collisions to evaluate, path transitions (two rings: rays that come
// from a delta-tracking / DRT walk, rays that come from a transmittance walk; only the adjoint kernels use both), free
// records - say what is to be done; a wave takes up to 64 ids of ONE kind, loads those rays, runs that kind's code with all
// its lanes, stores them and pushes their ids to the queues of what they need next (one reserving LDS atomic for all kinds).
// A wave that finds no full batch walks flights (as in drt_super.hip: DRT_SQ_K cells per look, lanes refilled from the
// flight queue); the lanes that set a flight up step its first DRT_SQ_INLINE_K cells themselves.  A transition batch runs a
// second pass only for >= DRT_SQ_T_PASS rays (the others are re-queued: a pass for a few lanes costs as much as one for 64),
// and the adjoint kernels run the "NEE walk finished" block twice per pass so that a main path whose walks come out of the
// path cache does a whole bounce in one pass.  Supergrids whose bf16 majorants do not fit LDS (64^3 cells: a 512^3 grid at
// the reference's factor 8) run the MG instantiations: one bit per cell in LDS, the majorants of non-empty cells from L2.
//
// Arithmetic, random-number consumption and event counts are those of the scalar restatement (oracle/drt_oracle.c):
// radiance is bit-exact per ray, counters are equal; gradients differ by summation order only.  A ray computes the same
// numbers whichever lanes run its pieces.  Not handled here (the host keeps drt_super.hip / the one-ray-per-lane kernels):
// supergrids of more than 511 cells per axis or whose cell bitmask does not fit LDS either, the atomic gradient path.  Quadratic DRT
// (the paper's comparison estimator) runs in the QUAD instantiations of the adjoint kernels: the main path is suspended at every
// vertex for the DRT walk + recursive path the subsampled estimator runs once at the end of a path.  Design history, profiles and what was measured and not kept: DESIGN.md section 6.2, profiles/r04_sq_experiments.txt.
#include <atomic>
#include "drt_device.h"
#include "drt_launch.h"

#ifndef DRT_SQ_THREADS
#define DRT_SQ_THREADS 768         // threads per workgroup = per CU: 12 waves
#endif
#ifndef DRT_SQ_MAX_RAYS
#define DRT_SQ_MAX_RAYS 896        // most ray records per workgroup (a launch takes what fits LDS, a multiple of 64: Params::sq_rays); measured on
                                   // supergrids that leave room for more than the headline's 768 (config 2, 16^3 cells: 640 / 768 / 896 / 1024 records:
                                   // 590 / 621 / 629-632 / 620-626 Msamples/s; config 4, majorants in L2: 896 / 1024: 631-634 / 628-631): beyond ~900
                                   // the records' global halves outgrow the L2
#endif
#ifndef DRT_SQ_RING
#define DRT_SQ_RING 1024           // entries per ring buffer of ids (a power of two >= DRT_SQ_MAX_RAYS)
#endif
#ifndef DRT_SQ_MIN_RAYS
#define DRT_SQ_MIN_RAYS 256        // fewer records than this: the host keeps drt_super.hip
#endif
#ifndef DRT_SQ_K
#define DRT_SQ_K 8                 // cells per walker lane between two looks at the queues
#endif
#ifndef DRT_SQ_REFILL_MIN
#define DRT_SQ_REFILL_MIN 16       // free walker lanes before more flights are taken
#endif
#ifndef DRT_SQ_BATCH
#define DRT_SQ_BATCH 56            // entries of a heavy queue that make a batch worth taking at once (48 / 56 / 64, alternating runs on one box:
                                   // headline 876 / 874 / 868 Msamples/s, config 2 628 / 633 / 626, envmap + factor 8 - / 711 / 702, config 3 the same)
#endif
#ifndef DRT_SQ_REGEN_MIN
#define DRT_SQ_REGEN_MIN 48        // free records before new rays are started (the prologue is long)
#endif
#ifndef DRT_SQ_LEAVE_MAX
#define DRT_SQ_LEAVE_MAX 40        // a walker with at most this many flights under way leaves for a full heavy batch
#endif
#ifndef DRT_SQ_MAXPOLL
#define DRT_SQ_MAXPOLL 3           // polls with nothing full to do before a partial batch is taken
#endif
#ifndef DRT_SQ_TAIL_FAST
#define DRT_SQ_TAIL_FAST 1         // a workgroup whose ray queues are drained takes partial batches at once (no polls: its last paths are latency)
#endif
#ifndef DRT_SQ_T_PASS
#define DRT_SQ_T_PASS 24           // a transition batch goes round again while at least this many of its rays are not at their next walk yet (flights
                                   // that cannot collide end their walk in the set-up; adjoint: walks out of the path cache); fewer go back to the
                                   // transition queue and meet a full batch: a pass for a few lanes costs the wave as much as one for 64
                                   // (measured 1 / 8 / 16 / 24 / 32 / 48: headline 792 / 809 / 815 / 823 / 815 / 804 Msamples/s, profiles/r04_sq_experiments.txt)
#endif
#ifndef DRT_SQ_RT2
#define DRT_SQ_RT2 1               // adjoint kernels: the "NEE walk finished" block a second time behind the emitter direction block
#endif
#ifndef DRT_SQ_PUSH_ALL
#define DRT_SQ_PUSH_ALL 1          // a batch's rays go to their queues with ONE reserving LDS atomic (0: one sq_push per kind)
#endif
#ifndef DRT_SQ_EARLY_OUT
#define DRT_SQ_EARLY_OUT 1         // flights that cannot collide (target optical depth > largest majorant x segment length) are not walked
#endif
#ifndef DRT_SQ_CHUNK
#define DRT_SQ_CHUNK 256           // queue positions a workgroup reserves per refill of its ray pool (launches with a ray order)
#endif
#ifndef DRT_SQ_CHUNK_MAX
#define DRT_SQ_CHUNK_MAX 4096      // ... launches in index order: span / (32 x workgroups), between DRT_SQ_CHUNK and this.  Round 6: the refill is ONE
                                   // returning atomic on one of eight queue heads, and returning atomics on one address serialise in L2 (~10 M/s):
                                   // the optimisation loop's 33.5 M-ray primal launch - 131 000 refills of 256, 16 000 per head - took 1.66 ms whatever
                                   // its rays did (thin medium, every ray over at once: profiles/r06_config3_levels.txt); with 4096 positions per
                                   // refill the heads see 1 000 each.  Launches with a ray order keep 256: their units are sorted thick-first, and
                                   // larger reservations concentrate the expensive rays on few workgroups (measured in round 3: 512 / 1024: -2 % / -30 %)
#endif
#ifndef DRT_SQ_RUN
#define DRT_SQ_RUN 16384           // consecutive rays per XCD-owned run
#endif
#ifndef DRT_SQ_INLINE_K
#define DRT_SQ_INLINE_K 4          // cells a flight is stepped by the lanes that set it up, before it is posted for the walkers
#endif
#ifndef DRT_SQ_REGEN_FINISH
#define DRT_SQ_REGEN_FINISH 2      // primal kernels: 1 = rays that are over before they begin (box misses, a first flight that cannot collide) are finished in the
                                   // regeneration block; 2 = ... and when most of a batch's records are free again they take the next rays in the same block (rounds)
#endif
#ifndef DRT_SQ_REGEN_AGAIN
#define DRT_SQ_REGEN_AGAIN 48      // DRT_SQ_REGEN_FINISH 2: free records of the batch that make another round worth it (a thin medium: nearly every ray is over at once)
#endif
#ifndef DRT_SQ_TAIL_PUSH
#define DRT_SQ_TAIL_PUSH 64        // adjoint launches with a tail pool (Params::tail_pool): a workgroup whose ray queues are drained and that holds at most
                                   // this many live records writes them to the pool and ends; a second launch (tail_mode) finishes them beside the
                                   // partition passes of the gradient reduction (measured: a drained workgroup's last paths are 0.45 ms of the launch)
#endif
#ifndef DRT_SQ_TAIL_BLOCKS
#define DRT_SQ_TAIL_BLOCKS 64      // workgroups of the tail launch (the partition passes of the reduction run on the other compute units)
#endif
constexpr int kSqTailQuads = 20;   // uint4 per pool entry: 7 (the LDS record) + 1 {queue kind} + 3 (global part a) + up to 9 (part b)
#ifndef DRT_SQ_PROFILE
#define DRT_SQ_PROFILE 0
#endif

#if DRT_SQ_PROFILE == 6
// experiment build: when does a path END, how old is it then and how long was it (bounce-loop iterations of the main + recursive
// path)?  32 buckets of 0.25 ms of the workgroup's clock x {paths, sum of ages (1.28 us), sum of iterations, paths older than half
// of the launch so far, largest age}; read with drt_sq_debug_read (tools/finish_age_profile.py)
__device__ unsigned long long g_sq_dbg[160];
extern "C" int drt_sq_debug_read(unsigned long long *out, int n, int reset)
{
    if (n > 160) n = 160;
    hipError_t e = hipMemcpyFromSymbol(out, HIP_SYMBOL(g_sq_dbg), (size_t) n * sizeof(unsigned long long));
    if (e == hipSuccess && reset) { static unsigned long long z[160]; e = hipMemcpyToSymbol(HIP_SYMBOL(g_sq_dbg), z, sizeof(z)); }
    return (int) e;
}
#endif

namespace drt {

namespace {

enum SqPhase : int {
    // walk phases: the ray is inside a tracking walk
    SP_DT = 0, SP_RT, SP_RTA, SP_DRT,
    // transition phases
    SP_HEAD, SP_SCAT, SP_ESC, SP_NEE, SP_RT_END, SP_RTA_END, SP_PHASE, SP_END, SP_DRT_END,
    SP_IDLE, SP_NONE,
    SP_QSCAT2                      // quadratic DRT (QUAD kernels): the main path resumes behind the DRT detour of a vertex - a transition phase like
                                   // SP_HEAD .. SP_DRT_END (sq_is_trans); numbered behind the others so that their constants are those of the other kernels
};
enum SqFlight : int { SF_NEW = 0, SF_NEXT = 1, SF_WAIT = 2 };   // first flight of a walk to set up | next flight to set up | posted
#ifndef DRT_SQ_SPLIT
#define DRT_SQ_SPLIT 1             // two queues of transitions: 1 in the adjoint kernels, 2 in all, 0 in none (measured, headline primal / adjoint
                                   // ms: none 2.67 / 5.27, all 2.75 / 5.21; config 2: 1.98 / 3.89, 1.97 / 3.80; profiles/r04_sq_experiments.txt)
#endif
// queues: flights to walk | collisions to evaluate | path transitions, by the block a ray enters them with - TA: a real collision /
// an escape / the DRT vertex / the emitter direction / the end of a path (the rays that come from a delta-tracking or DRT walk), TB:
// the end of a transmittance walk / phase sampling / the loop head (the rays that come from a ratio-tracking walk) - | free records.
// (One queue gave batches whose rays needed different blocks: each block ran with 14-27 of 64 lanes, profiles/r04_sq_experiments.txt.)
enum SqKind : int { SQ_WALK = 0, SQ_COLL, SQ_TA, SQ_TB, SQ_REGEN, SQ_KINDS };
constexpr uint32_t kSqEmpty = 0xffffu;
template <bool QUAD>
__device__ __forceinline__ bool sq_is_trans(int ph) { return (ph >= SP_HEAD && ph < SP_IDLE) || (QUAD && ph == SP_QSCAT2); }
template <bool SPLIT>
__device__ __forceinline__ int sq_trans_kind(int ph)
{
    if (!SPLIT) return SQ_TB;
    return (ph == SP_RT_END || ph == SP_RTA_END || ph == SP_PHASE || ph == SP_HEAD) ? SQ_TB : SQ_TA;
}

typedef __attribute__((address_space(3))) volatile uint32_t sq_vu32;
typedef __attribute__((address_space(3))) volatile uint16_t sq_vu16;
typedef __attribute__((address_space(3))) volatile unsigned long long sq_vu64;
__device__ __forceinline__ void sq_fence() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }

__device__ __forceinline__ uint32_t sq_xcc_id()
{
    return __builtin_amdgcn_s_getreg(20 | (0 << 6) | ((4 - 1) << 11)) & 7u;   // HW_REG_XCC_ID[3:0]
}

// Ring buffers of ray ids: ctl[kind] = {tail (pushes reserved) : head (pops reserved)}, both counting up.  A ray is in at
// most one queue, so DRT_SQ_RING >= (records) entries per ring never overflow; an entry is kSqEmpty until its id has been written
// (a pop may be reserved between a push's reservation and its write: the popping lane waits for the id).

// up to max_n entries, none if fewer than min_n are there; returns the count, the first position in `h` (wave-uniform)
__device__ __forceinline__ uint32_t sq_pop(unsigned long long *ctl, int kind, uint32_t max_n, uint32_t min_n, uint32_t lane, uint32_t &h)
{
    uint32_t hh = 0, n = 0;
    if (lane == 0) {
        for (;;) {
            const unsigned long long c = ((sq_vu64 *) ctl)[kind];
            const uint32_t head = (uint32_t) c, avail = (uint32_t) (c >> 32) - head;
            const uint32_t take = avail < max_n ? avail : max_n;
            if (take == 0u || take < min_n) break;
            if (atomicCAS(ctl + kind, c, c + take) == c) { hh = head; n = take; break; }
        }
    }
    h = (uint32_t) __builtin_amdgcn_readfirstlane((int) hh);
    return (uint32_t) __builtin_amdgcn_readfirstlane((int) n);
}

// the id at ring position pos of `kind` (taken: the entry is emptied)
__device__ __forceinline__ uint32_t sq_take(uint16_t *q, int kind, uint32_t pos)
{
    sq_vu16 *e = (sq_vu16 *) q + kind * DRT_SQ_RING + (pos & (DRT_SQ_RING - 1u));
    uint32_t id = *e;
    while (id == kSqEmpty) { __builtin_amdgcn_s_sleep(1); id = *e; }
    *e = (uint16_t) kSqEmpty;
    return id;
}

// ids of the lanes with `pred` (what they wrote to their records before must be visible: sq_fence first)
__device__ __forceinline__ void sq_push(unsigned long long *ctl, uint16_t *q, int kind, bool pred, uint32_t id, uint32_t lane)
{
    const uint64_t m = __ballot(pred);
    if (!m) return;
    const int leader = __ffsll((long long) m) - 1;
    uint32_t tail = 0;
    if ((int) lane == leader) tail = (uint32_t) (atomicAdd(ctl + kind, (unsigned long long) __popcll(m) << 32) >> 32);
    tail = (uint32_t) __builtin_amdgcn_readlane((int) tail, leader);
    if (pred) {
        const uint32_t rank = (uint32_t) __popcll(m & ((1ull << lane) - 1ull));
        ((sq_vu16 *) q)[kind * DRT_SQ_RING + ((tail + rank) & (DRT_SQ_RING - 1u))] = (uint16_t) id;
    }
}

// a batch's rays to the queues of what they need next - `dest`: the lane's queue kind, or SQ_KINDS for none.  ONE LDS atomic
// instruction reserves the positions of every kind (lane k reserves for kind k): the reservations of five sq_push calls in a
// row each waited for their own returning atomic
__device__ __forceinline__ void sq_push_all(unsigned long long *ctl, uint16_t *q, int dest, uint32_t id, uint32_t lane)
{
    uint64_t m[SQ_KINDS];
#pragma unroll
    for (int k = 0; k < SQ_KINDS; ++k) m[k] = __ballot(dest == k);
    uint32_t mine = 0;                                                       // lane k: rays for kind k
#pragma unroll
    for (int k = 0; k < SQ_KINDS; ++k) mine = lane == (uint32_t) k ? (uint32_t) __popcll(m[k]) : mine;
    uint32_t tail = 0;
    if (lane < (uint32_t) SQ_KINDS && mine) tail = (uint32_t) (atomicAdd(ctl + lane, (unsigned long long) mine << 32) >> 32);
    if (dest < SQ_KINDS) {
        uint32_t t = 0; uint64_t mm = 0;
#pragma unroll
        for (int k = 0; k < SQ_KINDS; ++k) {
            const uint32_t tk = (uint32_t) __builtin_amdgcn_readlane((int) tail, k);
            if (dest == k) { t = tk; mm = m[k]; }
        }
        const uint32_t rank = (uint32_t) __popcll(mm & ((1ull << lane) - 1ull));
        ((sq_vu16 *) q)[dest * DRT_SQ_RING + ((t + rank) & (DRT_SQ_RING - 1u))] = (uint16_t) id;
    }
}

}  // namespace

// MG: the supergrid's majorants do not fit LDS next to the records (64^3 cells: a 512^3 grid at the reference's factor 8) - LDS
// holds one BIT per cell (non-empty: two thirds of the cells a flight crosses answer without a load) and the majorants of the
// others are read from global memory (1 MB, L2-resident); the cell steps are then unpredicated in both passes (the geometry of
// the 8 steps first, their loads together) and a flight's first cells are not stepped by the lanes that set it up
// QUAD (adjoint kernels): quadratic DRT (use_drt without use_drt_subsampling, the paper's comparison estimator: volpathsimple.py:143-150
// calls backpropagate_scattering_drt at EVERY vertex of the main path).  The main path is suspended in the middle of its collision
// block: its state goes into the record's global half (the reservoir's slots, which this estimator does not use, + three more
// quads), the record runs the DRT walk along the current segment and the recursive path from the selected vertex exactly as the
// subsampled estimator does at the end of a path, and at the end of the recursion the main path is restored - with the alt sampler
// advanced by the detour's draws - and resumes with the second half of the block (SP_QSCAT2).
// TAILM (adjoint kernels): the tail launch (Params::tail_mode) - it starts from the records of the tail pool instead of the ray queues.  An instantiation of
// its own: with the pool's prologue compiled into the main kernels those came out 3 KB larger and 8 % slower (instruction cache; profiles/r05_sq_experiments.txt)
// ROUNDS (primal kernels of launches in index order over a THIN medium, Params::sq_rounds): the regeneration block hands the records whose ray was over
// at once their next ray in the same block.  An instantiation of its own: compiled into the others the loop cost the headline's primal launch 0.15 ms
// without running once (profiles/r06_sq_instruction_budget.txt)
template <bool ADJ, bool COUNT, bool ENV, bool MG, bool QUAD = false, bool TAILM = false, bool ROUNDS = false>
__global__ void __launch_bounds__(DRT_SQ_THREADS) trace_sq_kernel(const Params P)
{
    constexpr int NWV = DRT_SQ_THREADS / 64;
    constexpr int R4 = 7;                                                    // uint4 per ray record in LDS
    static_assert(ADJ || !QUAD, "the primal pass of the quadratic estimator is the ordinary one");
    // SOLO: the tail launch runs a batch's rays to their ENDS in the registers of the lanes that loaded them - transitions, the next flight walked to
    // its end right where it is set up, its collision, round again - without a queue hop in between: a launch's last paths are latency, and a lone
    // ray's hop through the queues (store the record, another wave finds it, loads it) costs more than the work it carries.  (Not with majorants in
    // L2, MG: those flights are not stepped by the lanes that set them up.)
    constexpr bool SOLO = TAILM && !MG;
    constexpr int NB = QUAD ? 9 : 6;                                         // uint4 of part b of the global record (adjoint)
    constexpr int NC = ADJ ? 3 + NB : 3;                                     // uint4 per ray in global memory (Params::sq_cold)
    // LDS record: [0] {tn.x, tn.y, tn.z, cell} [1] {td.x, td.y, td.z, steps left (9 bits per axis) + direction signs}
    // [2] {tau, tmax, t, acc} - the flight (a finished flight leaves its cell's majorant, 0: left the segment, in [0].x) -
    // [3] {rd, wmax} [4] {wo, wt} [5] {G.state, G.inc}: the generator the current walk draws from (the alt sampler in the main
    // path's DRT walk, the sampler everywhere else) [6] {DRT walk: ro | other walks: adjsum, steps of the walk, -; flags}
    // global record, part a ([records][3] per workgroup): [0] {ro, si_t} [1] {beta, nt0} [2] {result, ray index}; part b, adjoint
    // ([records][6]; the main path's: a recursive path only reads r_si_t, r_o, r_cw at its end): [0] {dL, r_si_t}
    // [1] {sampler clone, r_depth, -} [2] {r_o, r_wsum.x} [3] {r_d, r_wsum.y} [4] {r_cw, r_wsum.z} [5] {the other generator}.
    // Records are contiguous (a batch holds arbitrary ids: one or two 128-byte lines per ray and part)
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
    const int NRAY = (int) P.sq_rays;                                        // records of this launch (a multiple of 64)
    const int n_cells = P.gx * P.gy * P.gz;
    const int mg_words = MG ? (n_cells + 31) / 32 : (n_cells + 1) / 2;
    uint4 *rec4 = (uint4 *) lds;
    uint32_t *mg_lds = lds + NRAY * R4 * 4;
    uint16_t *q_lds = (uint16_t *) (mg_lds + ((mg_words + 3) & ~3));
    unsigned long long *ctl = (unsigned long long *) (q_lds + SQ_KINDS * DRT_SQ_RING);
    unsigned long long *pool = ctl + SQ_KINDS;                                // [0] next, [1] end of the workgroup's reserved positions of the ray queues
    uint32_t *misc = (uint32_t *) (pool + 2);                                 // [0] dead records, [1] bits of the largest majorant, [2] lock of the pool, [3] ray queues tried,
                                                                             // [4] the workgroup hands its last records to the tail pool, [5] waves that have left for it
    uint32_t *recst = misc + 8;                                              // record-stream state per wave (emit_record)
#if DRT_SQ_PROFILE == 6
    uint32_t *pdbg = recst + NWV * 8;
    for (int w = threadIdx.x; w < 160; w += blockDim.x) pdbg[w] = 0u;
#endif
    for (int i = threadIdx.x; i < SQ_KINDS * DRT_SQ_RING; i += blockDim.x) {
        const int k = i - SQ_REGEN * DRT_SQ_RING;                            // every record starts in the ring of free records
        q_lds[i] = (uint16_t) (k >= 0 && k < NRAY ? k : (int) kSqEmpty);
    }
    if (threadIdx.x < SQ_KINDS) ctl[threadIdx.x] = threadIdx.x == SQ_REGEN ? ((unsigned long long) NRAY << 32) : 0ull;
    if (threadIdx.x < 8) misc[threadIdx.x] = (threadIdx.x == 3 && TAILM) ? 8u                // (tail mode: the ray queues count as drained)
                                           : (threadIdx.x == 4 && !TAILM && P.tail_pool) ? (uint32_t) DRT_SQ_TAIL_PUSH : 0u;
    if (threadIdx.x < 2) pool[threadIdx.x] = 0ull;
    for (int w = threadIdx.x; w < NWV * 8; w += blockDim.x) recst[w] = 0u;
    __syncthreads();
    if constexpr (MG) {
        for (int w = threadIdx.x; w < mg_words; w += blockDim.x) mg_lds[w] = P.mocc[w];
        // the largest cell majorant: every cell's is (scale x its largest sigma_t) rounded up to bf16 (majorant_grid_kernel), so
        // the global majorant rounded up the same way bounds them all (the early-out below only needs a bound)
        uint32_t b = __float_as_uint(P.majorant[0]);
        if (b & 0xffffu) b = (b | 0xffffu) + 1u;
        if (threadIdx.x == 0) misc[1] = b;
    } else {                                                                 // (the grid's values are bf16-representable: exact)
        uint32_t top = 0u;                                                   // (non-negative floats order like their bit patterns)
        for (int w = threadIdx.x; w < mg_words; w += blockDim.x) {
            const uint32_t a = __float_as_uint(P.mgrid[2 * w]), b = 2 * w + 1 < n_cells ? __float_as_uint(P.mgrid[2 * w + 1]) : 0u;
            mg_lds[w] = (a >> 16) | (b & 0xffff0000u);
            top = max(top, max(a, b));
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) top = max(top, (uint32_t) __shfl_down((int) top, off, 64));
        if ((threadIdx.x & 63u) == 0u && top) atomicMax(misc + 1, top);
    }
    __syncthreads();
    // A flight whose target optical depth exceeds (largest majorant) x (length of its segment) cannot end in a collision
    // whatever cells it crosses: it is not walked (flight set-up below; the bound is drt_super.hip's).
    const float mmax = __uint_as_float(__builtin_amdgcn_readfirstlane((int) misc[1]));
    const uint32_t *occ = nullptr;   // (tentative collisions lie in non-empty supergrid cells: the voxel bitmask would rarely say "empty")
    // (DRT_SQ_REGEN_FINISH) "thin": a flight across the whole box has a fair chance (> e^-3) of an optical-depth target beyond that bound - only then
    // is it worth looking at a ray's first target in the regeneration block
    const float box_diag = sqrtf((P.bmax[0] - P.bmin[0]) * (P.bmax[0] - P.bmin[0]) + (P.bmax[1] - P.bmin[1]) * (P.bmax[1] - P.bmin[1]) +
                                 (P.bmax[2] - P.bmin[2]) * (P.bmax[2] - P.bmin[2]));
    const bool thin = !ADJ && mmax * box_diag < 3.0f;

    const uint32_t lane = threadIdx.x & 63u;
    const int wave = __builtin_amdgcn_readfirstlane((int) (threadIdx.x >> 6));
    const uint16_t *mg16 = (const uint16_t *) mg_lds;
    uint32_t cnt[C_COUNT];
#pragma unroll
    for (int i = 0; i < C_COUNT; ++i) cnt[i] = 0;
#if DRT_SQ_PROFILE
    // experiment build (tools/mk_variant.sh NAME -DDRT_SQ_PROFILE=1 drt_sq.hip): the counting kernels' slots hold, summed over
    // waves, [0] lane cell steps, [1] wave cell steps, [2] collision batches, [3] rays in them, [4] transition batches, [5] rays
    // in them, [6] regeneration batches, [7] records in them, [8] polls
#define SQ_COUNT(slot) do { } while (0)
#define SQ_PROF(slot, v) do { if (DRT_SQ_PROFILE == 1 && COUNT) { const uint32_t v_ = (uint32_t) (v); if (lane == 0) cnt[slot] += v_; } } while (0)
    // DRT_SQ_PROFILE=2: shader clock (units of 64 cycles) per wave spent - [0] cell steps, [1] the walker's queue work (refill,
    // results, hand-back), [2] loading a batch (collision batches), [3] collision code, [4] loading a batch (transition batches:
    // LDS + global memory), [5] transition / regeneration code, [6] flight set-up, [7] storing a batch + pushes, [8] looking for work
    uint64_t pt_last = __builtin_readcyclecounter();
#define SQ_STAMP(slot) do { if (DRT_SQ_PROFILE == 2 && COUNT) { const uint64_t t_ = __builtin_readcyclecounter(); if (lane == 0) cnt[slot] += (uint32_t) ((t_ - pt_last) >> 6); pt_last = t_; } } while (0)
    // DRT_SQ_PROFILE=3 / 4: per block of the transition pass, the waves that ran it (3) / the lanes that needed it (4) -
    // [0] DRT vertex, [1] NEE walk finished, [2] phase sampling, [3] loop head, [4] real collision / escape, [5] emitter
    // direction, [6] end of a path, [7] flight set-up, [8] passes of the transition loop (profiles/r04_sq_experiments.txt)
#define SQ_BLK(slot, pred) do { if (DRT_SQ_PROFILE >= 3 && COUNT) { const uint64_t m_ = __ballot(pred); if (m_ && lane == 0) cnt[slot] += DRT_SQ_PROFILE == 3 ? 1u : (uint32_t) __popcll(m_); } } while (0)
#else
#define SQ_STAMP(slot) do { } while (0)
#define SQ_COUNT(slot) do { if (COUNT) cnt[slot]++; } while (0)
#define SQ_PROF(slot, v) do { } while (0)
#define SQ_BLK(slot, pred) do { } while (0)
#endif

    // uniform supergrid constants
    const int gx = P.gx, gy = P.gy, gz = P.gz;
    const float fgx = (float) gx, fgy = (float) gy, fgz = (float) gz;
    const int lin_y = gx, lin_z = gx * gy;

    uint32_t *rec = recst + wave * 8;                                          // record-stream state of this wave (emit_record)
    uint4 *cold_a = (uint4 *) P.sq_cold + (size_t) blockIdx.x * NC * NRAY;    // [NRAY][3] uint4 of this workgroup
    uint4 *cold_b = cold_a + 3 * NRAY;                                        // [NRAY][6] (adjoint; QUAD: [9], the last three: the suspended main path)
    const uint32_t xcc = sq_xcc_id();
    // (with a ray order the queue positions cover whole units: the last unit may reach past the launch's last ray)
    const uint64_t span = P.order ? (uint64_t) P.order_units * P.order_unit : P.n_rays - P.ray_first;
    const uint64_t n_runs = (span + DRT_SQ_RUN - 1) / DRT_SQ_RUN;
    // queue x serves the runs x, x + 8, ...; a workgroup starts on the queue of the XCD it runs on (L2 locality) and moves on
    // to the next ones when that one is drained: every ray is traced whatever the placement of the workgroups.  The
    // positions a workgroup has reserved (Params::sq_chunk at a time) are handed out from LDS under a lock: any wave starts rays.
    int polls = 0;
    if constexpr (TAILM) {
        // tail mode (Params::tail_mode): this launch finishes the records the main launch's drained workgroups wrote to the pool - workgroup b takes
        // the entries b, b + gridDim.x, ... into free records, each into the queue it was taken from; no ray is started (the ray queues count as drained)
        {
            const uint32_t n_pool = min(*P.tail_count, P.tail_cap);
            const uint32_t n_tail = n_pool > blockIdx.x ? min((uint32_t) NRAY, (n_pool - blockIdx.x + gridDim.x - 1u) / gridDim.x) : 0u;
#pragma unroll 1
            for (uint32_t k = (uint32_t) wave; k < n_tail; k += NWV) {
                uint32_t hq;
                if (!sq_pop(ctl, SQ_REGEN, 1u, 1u, lane, hq)) break;
                uint32_t id_k = 0;
                if (lane == 0u) id_k = sq_take(q_lds, SQ_REGEN, hq);
                id_k = (uint32_t) __builtin_amdgcn_readfirstlane((int) id_k);
                const uint4 *src = P.tail_pool + ((size_t) blockIdx.x + (size_t) k * gridDim.x) * kSqTailQuads;
                uint4 v = make_uint4(0u, 0u, 0u, 0u);
                if (lane < (ADJ ? 11u + NB : 11u)) v = src[lane];
                if (lane < 7u) rec4[R4 * id_k + lane] = v;
                else if (lane >= 8u && lane < 11u) cold_a[3 * id_k + (lane - 8u)] = v;
                else if (ADJ && lane >= 11u && lane < 11u + NB) cold_b[NB * id_k + (lane - 11u)] = v;
                const int kd = __builtin_amdgcn_readlane((int) v.x, 7);
                __threadfence_block();
                sq_fence();
                sq_push(ctl, q_lds, kd, lane == 0u, id_k, lane);
            }
        }
        // (every entry has its record before any wave enters the loop: there the free records are retired at once - the ray queues count as drained)
        __syncthreads();
    }
#if DRT_SQ_PROFILE == 6
    const unsigned long long pt_start6 = __builtin_amdgcn_s_memrealtime();
#endif
#if DRT_SQ_PROFILE == 5
    // experiment build: when do the workgroups' ray queues run dry, when do their waves end (100 MHz clock; the counting kernels' slots:
    // [0] 2^62 - first start, [1] 2^62 - first end, [2] last end, [3] sum of (end - start), [4] waves, [5] sum of (dry - start), [6] 2^62 - first dry)
    const unsigned long long pt_start = __builtin_amdgcn_s_memrealtime(); unsigned long long pt_drained = 0;
#endif

    for (;;) {   // (once, unless a hand-over to the tail pool finds more records than the workgroup's share: then the loop is resumed, see behind it)
    for (;;) {
        // ---- what is there to do? ---------------------------------------------------------------------------
        uint32_t qn = 0;
        if (lane < (uint32_t) SQ_KINDS) { const unsigned long long c = ((sq_vu64 *) ctl)[lane]; qn = (uint32_t) (c >> 32) - (uint32_t) c; }
        const uint32_t n_walk = (uint32_t) __builtin_amdgcn_readlane((int) qn, SQ_WALK), n_coll = (uint32_t) __builtin_amdgcn_readlane((int) qn, SQ_COLL);
        const uint32_t n_ta = (uint32_t) __builtin_amdgcn_readlane((int) qn, SQ_TA), n_tb = (uint32_t) __builtin_amdgcn_readlane((int) qn, SQ_TB);
        const uint32_t n_regen = (uint32_t) __builtin_amdgcn_readlane((int) qn, SQ_REGEN);
        const uint32_t dead = (uint32_t) __builtin_amdgcn_readfirstlane((int) ((sq_vu32 *) misc)[0]);
        if (dead >= (uint32_t) NRAY) break;
        const bool drained = (uint32_t) __builtin_amdgcn_readfirstlane((int) ((sq_vu32 *) misc)[3]) >= 8u;
        if constexpr (!TAILM) {
            // a drained workgroup's last paths are latency: nothing on this CU can hide them.  Their records go to the tail pool (behind the loop) and
            // the workgroup ends; the tail launch finishes them while the partition passes of the gradient reduction run on the CUs this frees.
            // (`live` counts the records in the queues or in a wave's registers; the two LDS reads are not one snapshot: it may be low by one batch)
            // (the hand-over flag is the top bit of the dead-record count: every wave leaves through the check above)
            if (drained && ((uint32_t) NRAY - dead - n_regen) - 1u < (uint32_t) __builtin_amdgcn_readfirstlane((int) ((sq_vu32 *) misc)[4])) {
                if (lane == 0u) atomicOr(misc, 0x80000000u);
                continue;
            }
        }
#ifdef DRT_EXP_DROP_TAIL
        // timing experiment (wrong results): a drained workgroup with at most DRT_EXP_DROP_TAIL live records ends at once - what a launch costs
        // WITHOUT the latency of its last paths (the upper bound of what a tail pool can hide behind the reductions)
        if (drained && (uint32_t) NRAY - dead - n_regen <= (uint32_t) DRT_EXP_DROP_TAIL) break;
#endif
#if DRT_SQ_PROFILE == 5
        if (drained && !pt_drained) pt_drained = __builtin_amdgcn_s_memrealtime();
#endif
        int kind = -1; uint32_t min_n = DRT_SQ_BATCH;
        if (n_coll >= DRT_SQ_BATCH) kind = SQ_COLL;
        else if (n_ta >= DRT_SQ_BATCH) kind = SQ_TA;
        else if (n_tb >= DRT_SQ_BATCH) kind = SQ_TB;
        else if (n_regen >= DRT_SQ_REGEN_MIN || (drained && n_regen)) { kind = SQ_REGEN; min_n = 1; }
        else if (n_walk) kind = SQ_WALK;
        else if (n_coll | n_ta | n_tb | n_regen) {
            if (polls < ((DRT_SQ_TAIL_FAST && drained) ? 0 : DRT_SQ_MAXPOLL)) { ++polls; SQ_PROF(8, 1); __builtin_amdgcn_s_sleep(4); continue; }
            const uint32_t best = max(max(n_coll, n_regen), max(n_ta, n_tb));            // the fullest queue
            kind = n_coll == best ? SQ_COLL : n_ta == best ? SQ_TA : n_tb == best ? SQ_TB : SQ_REGEN;
            min_n = 1;
        } else { SQ_PROF(8, 1); if (DRT_SQ_TAIL_FAST >= 2 && drained) __builtin_amdgcn_s_sleep(1); else __builtin_amdgcn_s_sleep(8); continue; }

        SQ_STAMP(8);
        if (kind == SQ_WALK) {
            // ================= walk: posted flights -> supergrid cells -> results ==============================
            bool fly = false, walked = false;
            uint32_t slot = 0;
            float tnx = kInf, tny = kInf, tnz = kInf, tdx = kInf, tdy = kInf, tdz = kInf, t = 0.0f, acc = 0.0f, tau = 0.0f, tmax = 0.0f;
            int cell = 0, sx = 0, sy = 0, sz = 0;
            uint32_t rem = 0;
            for (;;) {
                const uint64_t flym = __ballot(fly);
                const int nfree = 64 - __popcll(flym);
                if (nfree >= DRT_SQ_REFILL_MIN) {
                    uint32_t h0;
                    const uint32_t got = sq_pop(ctl, SQ_WALK, (uint32_t) nfree, 1u, lane, h0);
                    if (got) {
                        const uint32_t frank = (uint32_t) __popcll(~flym & ((1ull << lane) - 1ull));   // my rank among the free lanes
                        if (!fly && frank < got) {
                            slot = sq_take(q_lds, SQ_WALK, h0 + frank);
                            const uint4 *sp = rec4 + R4 * slot;
                            const uint4 q0 = sp[0], q1 = sp[1], q2 = sp[2];
                            tnx = __uint_as_float(q0.x); tny = __uint_as_float(q0.y); tnz = __uint_as_float(q0.z); cell = (int) q0.w;
                            tdx = __uint_as_float(q1.x); tdy = __uint_as_float(q1.y); tdz = __uint_as_float(q1.z); rem = q1.w;
                            tau = __uint_as_float(q2.x); tmax = __uint_as_float(q2.y); t = __uint_as_float(q2.z); acc = __uint_as_float(q2.w);
                            sx = (rem & (1u << 27)) ? -1 : 1; sy = (rem & (1u << 28)) ? -lin_y : lin_y; sz = (rem & (1u << 29)) ? -lin_z : lin_z;
                            fly = true;
                        }
                    }
                }
                if (!__ballot(fly)) break;                                       // nothing to walk (any more)
                walked = true;
                SQ_STAMP(1);
                // (primal kernels: the steps are not predicated on `fly`, as in drt_super.hip)
                constexpr bool kLoose = !ADJ || MG;
                bool fin = false; float res_mc = 0.0f, res_t = 0.0f, res_acc = 0.0f;
#pragma unroll
                for (int k = 0; k < DRT_SQ_K; ++k) {
#if DRT_SQ_PROFILE == 1
                    { const int nf = __popcll(__ballot(fly)); SQ_PROF(0, nf); SQ_PROF(1, 1); }
#endif
                    // one supergrid cell (oracle: the loop of sample_collision).  Crossing times are finite or +inf, never NaN.
                    const float tmin = fminf(fminf(tnx, tny), tnz);
                    const float texit = fminf(tmin, tmax);
                    const uint32_t ci = kLoose ? min((uint32_t) cell, (uint32_t) (n_cells - 1)) : (uint32_t) cell;
                    float mc;
                    if constexpr (MG) mc = ((mg_lds[ci >> 5] >> (ci & 31u)) & 1u) ? P.mgrid[ci] : 0.0f;
                    else mc = __uint_as_float((uint32_t) mg16[ci] << 16);
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
                SQ_STAMP(0);
                if (__ballot(fin)) {
                    // result: where the last cell was entered, the optical depth up to there, its majorant (0: left the segment)
                    if (fin) {
                        uint4 *sp = rec4 + R4 * slot;
                        sp[0].x = __float_as_uint(res_mc);
                        sp[2].z = __float_as_uint(kLoose ? res_t : t); sp[2].w = __float_as_uint(kLoose ? res_acc : acc);
                    }
                    sq_fence();
                    sq_push(ctl, q_lds, SQ_COLL, fin, slot, lane);
                }
                // a full heavy batch is waiting and this wave has little under way: leave
                const int nfly = __popcll(__ballot(fly));
                if (nfly <= DRT_SQ_LEAVE_MAX) {
                    uint32_t hn = 0;
                    if (lane >= (uint32_t) SQ_COLL && lane < (uint32_t) SQ_KINDS) { const unsigned long long c = ((sq_vu64 *) ctl)[lane]; hn = (uint32_t) (c >> 32) - (uint32_t) c; }
                    if (__ballot(hn >= DRT_SQ_BATCH)) break;
                }
            }
            if (__ballot(fly)) {                                                 // flights still under way: back to their records
                if (fly) {
                    uint4 *sp = rec4 + R4 * slot;
                    sp[0] = make_uint4(__float_as_uint(tnx), __float_as_uint(tny), __float_as_uint(tnz), (uint32_t) cell);
                    sp[1].w = rem;
                    sp[2].z = __float_as_uint(t); sp[2].w = __float_as_uint(acc);
                }
                sq_fence();
                sq_push(ctl, q_lds, SQ_WALK, fly, slot, lane);
            }
            if (walked) polls = 0;
            SQ_STAMP(1);
            continue;
        }

        // ================= a heavy batch: up to 64 rays of one kind ==========================================
        uint32_t h0;
        const uint32_t nb = sq_pop(ctl, kind, 64u, min_n, lane, h0);
        if (!nb) continue;                                                       // (another wave was faster)
        polls = 0;
        const bool act = lane < nb;
        uint32_t id = 0;
        if (act) id = sq_take(q_lds, kind, h0 + lane);
        sq_fence();
        uint4 *R = rec4 + R4 * id;

        // ---- per-ray state (registers of this batch only) ----------------------------------------------------
        int ph = SP_NONE, fl = SF_WAIT;
        bool rec_mode = false, rec_first = false, escaped = false, has_scattered = false, scat_once = false, pc_on = false;
        int depth = 0, pc_it = 0;
        uint32_t li = 0, pc_steps = 0;
        V3 ro = v3(0, 0, 0), rd = v3(0, 0, 1), wo = v3(0, 0, 0);
        float si_t = kInf, wmax = 0.0f, wt = 0.0f, nt0 = 0.0f, adjsum = 0.0f;
        float beta[3] = { 1, 1, 1 }, result[3] = { 0, 0, 0 }, dL[3] = { 0, 0, 0 };
        Pcg32 S; S.state = 0; S.inc = 1;
        Pcg32 A; A.state = 0; A.inc = 1;
        uint64_t Cst = 0;
        int r_depth = -1; float r_si_t = kInf; V3 r_o = ro, r_d = rd;
        float r_wsum[3] = { 0, 0, 0 }, r_cw[3] = { 0, 0, 0 };
        bool b_live = true;                                                     // part b of the global record is in registers (adjoint)
        // QUAD: the suspended main path {origin, segment end} {radiance, collision distance} {flags: scatter | escaped | has_scattered |
        // scat_once | pc_on | pc_it << 8; the sampler's increment}
        V3 q_ro = v3(0, 0, 0); float q_si_t = 0.0f, q_wt = 0.0f, q_result[3] = { 0, 0, 0 }; uint32_t q_flags = 0, q_sinc_lo = 1, q_sinc_hi = 0;
        bool walk_done = false;                                                 // the flight set up here ended within its first cells
        bool unit_empty = false;                                                // regeneration batches: the ray's pixel crosses only empty supergrid cells (Params::unit_empty)
#if DRT_SQ_PROFILE == 6
        uint32_t t_start = 0;                                                   // 100 MHz clock at the ray's start (0: none)
#endif
        float c_lm = 0.0f, c_tau = 0.0f, c_t = 0.0f, c_acc = 0.0f;              // the finished flight (collision batches)
        float w_tdx = kInf, w_tdy = kInf, w_tdz = kInf; uint32_t w_rem = 0;      // the walk's direction share of the DDA

        if (kind != SQ_REGEN) {
            if (act) {
                const uint4 q3 = R[3], q4 = R[4], q5 = R[5], q6 = R[6];
                rd = v3(__uint_as_float(q3.x), __uint_as_float(q3.y), __uint_as_float(q3.z)); wmax = __uint_as_float(q3.w);
                wo = v3(__uint_as_float(q4.x), __uint_as_float(q4.y), __uint_as_float(q4.z)); wt = __uint_as_float(q4.w);
                const uint32_t f = q6.w;
                ph = (int) (f & 15u); fl = (int) ((f >> 4) & 3u);
                rec_mode = (f >> 6) & 1u; rec_first = (f >> 7) & 1u; escaped = (f >> 8) & 1u; has_scattered = (f >> 9) & 1u;
                scat_once = (f >> 10) & 1u; pc_on = (f >> 11) & 1u;
                depth = (int) ((f >> 12) & 1023u); pc_it = (int) (f >> 22);
                const bool drtw = ph == SP_DRT || ph == SP_DRT_END;             // (in / just out of the DRT walk)
                const bool gA = ADJ && !rec_mode && drtw;                       // the record's generator is the alt sampler
                {
                    const uint64_t gs = ((uint64_t) q5.y << 32) | q5.x, gi = ((uint64_t) q5.w << 32) | q5.z;
                    if (gA) { A.state = gs; A.inc = gi; } else { S.state = gs; S.inc = gi; }
                }
                if (drtw) ro = v3(__uint_as_float(q6.x), __uint_as_float(q6.y), __uint_as_float(q6.z));
                else { adjsum = __uint_as_float(q6.x); pc_steps = q6.y; }
#if DRT_SQ_PROFILE == 6
                if (!ADJ) t_start = q6.z;
#endif
                if (kind == SQ_COLL) {
                    const uint4 q1 = R[1], q2 = R[2];
                    c_lm = __uint_as_float(R[0].x); c_tau = __uint_as_float(q2.x); c_t = __uint_as_float(q2.z); c_acc = __uint_as_float(q2.w);
                    w_tdx = __uint_as_float(q1.x); w_tdy = __uint_as_float(q1.y); w_tdz = __uint_as_float(q1.z); w_rem = q1.w;
                }
                if (SOLO || kind != SQ_COLL) {                                  // transitions (SOLO: every batch): the rest of the ray, from global memory
                    const V3 ro_walk = ro;                                      // (in / just out of a DRT walk the record's copy is the current one: the same value)
                    const uint4 *ca = cold_a + 3 * id;
                    const uint4 c0 = ca[0], c1 = ca[1], c2 = ca[2];
                    ro = v3(__uint_as_float(c0.x), __uint_as_float(c0.y), __uint_as_float(c0.z)); si_t = __uint_as_float(c0.w);
                    beta[0] = __uint_as_float(c1.x); beta[1] = __uint_as_float(c1.y); beta[2] = __uint_as_float(c1.z); nt0 = __uint_as_float(c1.w);
                    result[0] = __uint_as_float(c2.x); result[1] = __uint_as_float(c2.y); result[2] = __uint_as_float(c2.z); li = c2.w;
                    if constexpr (ADJ) {
                      b_live = !rec_mode;
                      if (b_live) {
                        const uint4 *cb = cold_b + NB * id;
                        const uint4 c3 = cb[0], c4 = cb[1], c5 = cb[2], c6 = cb[3], c7 = cb[4], c8 = cb[5];
                        if constexpr (QUAD) {
                            const uint4 c9 = cb[6], c10 = cb[7], c11 = cb[8];
                            q_ro = v3(__uint_as_float(c9.x), __uint_as_float(c9.y), __uint_as_float(c9.z)); q_si_t = __uint_as_float(c9.w);
                            q_result[0] = __uint_as_float(c10.x); q_result[1] = __uint_as_float(c10.y); q_result[2] = __uint_as_float(c10.z); q_wt = __uint_as_float(c10.w);
                            q_flags = c11.x; q_sinc_lo = c11.y; q_sinc_hi = c11.z;
                        }
                        dL[0] = __uint_as_float(c3.x); dL[1] = __uint_as_float(c3.y); dL[2] = __uint_as_float(c3.z); r_si_t = __uint_as_float(c3.w);
                        Cst = ((uint64_t) c4.y << 32) | c4.x; r_depth = (int) c4.z;
#if DRT_SQ_PROFILE == 6
                        t_start = c4.w;
#endif
                        r_o = v3(__uint_as_float(c5.x), __uint_as_float(c5.y), __uint_as_float(c5.z)); r_wsum[0] = __uint_as_float(c5.w);
                        r_d = v3(__uint_as_float(c6.x), __uint_as_float(c6.y), __uint_as_float(c6.z)); r_wsum[1] = __uint_as_float(c6.w);
                        r_cw[0] = __uint_as_float(c7.x); r_cw[1] = __uint_as_float(c7.y); r_cw[2] = __uint_as_float(c7.z); r_wsum[2] = __uint_as_float(c7.w);
                        const uint64_t os = ((uint64_t) c8.y << 32) | c8.x, oi = ((uint64_t) c8.w << 32) | c8.z;
                        if (gA) { S.state = os; S.inc = oi; } else { A.state = os; A.inc = oi; }
                      }
                    }
                    if (drtw) ro = ro_walk;
                }
            }
        }

        if (kind == SQ_COLL) SQ_STAMP(2); else SQ_STAMP(4);
        // ================= (Fe) the collision a flight ended in (collision batches; SOLO: right behind the flight's walk) ===================
        auto collide = [&](bool on) {
            if (on) {
                const bool drt = ph == SP_DRT;
                const bool useA = ADJ && !rec_mode && drt;
                Pcg32 Rg; Rg.state = useA ? A.state : S.state; Rg.inc = useA ? A.inc : S.inc;
                // Medium::sample_interaction [M3-ext] (oracle: sample_collision): the walker left {entry distance of the
                // last cell, optical depth up to there, that cell's majorant (0: the flight left the segment)}
                const float lm = c_lm, tau = c_tau;
                const float lim = lm > 0.0f ? 1.0f / lm : 0.0f;
                const float dt = lm > 0.0f ? fmaf(tau - c_acc, lim, c_t) : kInf;
                bool inside; V3 p;
                if (drt) { wt += dt; inside = wt <= wmax; p = ray_at(ro, rd, wt); }
                else { inside = dt <= wmax; p = ray_at(wo, rd, dt); }
                const float sig = inside ? eval_sigma_t(P, p, occ) : 0.0f;
                fl = SF_NEXT;
                if (!inside) {                                              // left the segment
                    ph = drt ? SP_DRT_END : (ph == SP_DT) ? SP_ESC : (ph == SP_RT ? SP_RT_END : SP_RTA_END);
                } else if (drt) {                                           // Medium::sample_interaction_drt (:549-551); wo = {T, wsum, selected t}
                    SQ_COUNT(C_DRT);
                    const float w = wo.x * lim;
                    wo.y += w;
                    const float u2 = Rg.next_1d();
                    if (w > 0.0f && u2 * wo.y <= w) wo.z = wt;
                    wo.x *= (lm - sig) * lim;
                    if (wo.x == 0.0f) ph = SP_DRT_END;
                } else if (ph == SP_DT) {                                   // :348-367
                    SQ_COUNT(C_DT); ++pc_steps;
                    const float r = sig * lim;
                    const float u2 = Rg.next_1d();
                    if (!(u2 >= r)) { wt = wt + dt; ph = SP_SCAT; }          // mei.t
                    else { wo = p; wmax -= dt; wt += dt; }
                } else {                                                    // ratio tracking :465-502
                    SQ_COUNT(C_RT); ++pc_steps;
                    const float tr = (lm - sig) * lim;
                    if constexpr (ADJ) {
                        if (ph == SP_RTA && tr > 0.0f) {                    // :487-492
                            splat_sigma_t<true>(P, p, -(adjsum * lim) / tr, rec);
                            SQ_COUNT(C_RT_ADJ);
                        }
                    }
                    wt *= tr; wo = p; wmax -= dt;
                    if (wt == 0.0f) ph = (ph == SP_RT) ? SP_RT_END : SP_RTA_END;
                }
                if (useA) A.state = Rg.state; else S.state = Rg.state;
            }
        };
        if (kind == SQ_COLL) {
            SQ_PROF(2, 1); SQ_PROF(3, nb);
            collide(act);
        } else if (kind == SQ_REGEN) {
            // ================= (A) regeneration ===========================================================
            // Ray indices come from a wave-local pool refilled Params::sq_chunk at a time with ONE returning atomic on the
            // XCD's queue head.
            // (the adjoint kernels - and every kernel with DRT_SQ_REGEN_FINISH 0 - keep the block as it was: a source change in here moved the adjoint
            //  kernel's register allocation and cost it 0.19 ms at the headline, profiles/r06_sq_instruction_budget.txt)
            if constexpr (ADJ || DRT_SQ_REGEN_FINISH == 0) {
                SQ_PROF(6, 1); SQ_PROF(7, nb);
                const uint64_t wmask = __ballot(act);
                uint64_t first = 0; uint32_t got = 0, qx = 0, qs_ = 0;
                if (lane == 0) {
                    while (atomicCAS(misc + 2, 0u, 1u) != 0u) __builtin_amdgcn_s_sleep(1);
                    uint64_t pn = ((sq_vu64 *) pool)[0], pe = ((sq_vu64 *) pool)[1];
                    uint32_t qs = ((sq_vu32 *) misc)[3];
                    while (pn >= pe && qs < 8u) {                                // refill from the ray queues
                        const uint32_t x = (xcc + qs) & 7u;
                        const uint64_t len = (n_runs > x ? (n_runs - x + 7) / 8 : 0) * DRT_SQ_RUN;
                        const unsigned long long base = atomicAdd(P.queues + x, (unsigned long long) P.sq_chunk);
                        if (base < len) { pn = base; pe = base + P.sq_chunk < len ? base + P.sq_chunk : len; }
                        else ++qs;                                               // this queue is drained: next one
                    }
                    qx = (xcc + qs) & 7u; qs_ = qs;
                    const uint64_t want = (uint64_t) __popcll(wmask);
                    got = (uint32_t) (pe - pn < want ? pe - pn : want);
                    if (qs >= 8u) got = 0;
                    first = pn; pn += got;
                    ((sq_vu64 *) pool)[0] = pn; ((sq_vu64 *) pool)[1] = pe; ((sq_vu32 *) misc)[3] = qs;
                    sq_fence();
                    ((sq_vu32 *) misc)[2] = 0u;                                  // unlock
                }
                first = ((uint64_t) (uint32_t) __builtin_amdgcn_readfirstlane((int) (first >> 32)) << 32) | (uint32_t) __builtin_amdgcn_readfirstlane((int) first);
                got = (uint32_t) __builtin_amdgcn_readfirstlane((int) got); qx = (uint32_t) __builtin_amdgcn_readfirstlane((int) qx);
                qs_ = (uint32_t) __builtin_amdgcn_readfirstlane((int) qs_);
                if (qs_ >= 8u && !got) {                       // all eight queues are empty: these records are done
                    if (lane == 0) atomicAdd(misc, nb);
                    continue;
                }
                const uint32_t myr = (uint32_t) __popcll(wmask & ((1ull << lane) - 1ull));
                const bool take = act && myr < got;
                const uint64_t q = first + myr;
                if (act) ph = SP_IDLE;                                           // (no ray for this record: it stays free and draws again)
                if (take) {
                    uint64_t i = ((q / DRT_SQ_RUN) * 8 + qx) * DRT_SQ_RUN + (q % DRT_SQ_RUN);
                    if (P.order) {                                              // position -> unit of the order -> ray
                        const uint32_t g = (uint32_t) i, u = P.order_unit == 1u ? g : g / P.order_unit;
                        i = i < span ? (uint64_t) P.order[u] * P.order_unit + (g - u * P.order_unit) : P.n_rays;
                    }
                    if (P.unit_empty && i + P.ray_first < P.n_rays) unit_empty = P.unit_empty[(uint32_t) i / P.empty_unit] != 0;
                    i += P.ray_first;
                    if (i < P.n_rays) {
                        // ---- sample() prologue (:51-108) + reach_medium (:292-319) ----
                        li = (uint32_t) i;
    #if DRT_SQ_PROFILE == 6
                        t_start = (uint32_t) __builtin_amdgcn_s_memrealtime() | 1u;
    #endif
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
                        SQ_COUNT(C_RAYS);
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
                        ph = active ? SP_HEAD : SP_END;
                    }
                }
            } else {
                SQ_PROF(6, 1); SQ_PROF(7, nb);
                if (act) ph = SP_IDLE;                                           // (no ray for this record: it stays free and draws again)
                bool retired = false;
                // (primal kernels, DRT_SQ_REGEN_FINISH >= 2: ROUNDS - the records whose ray was over at once take another ray right here)
                for (int round = 0;; ++round) {
                const bool need = act && ph == SP_IDLE;
                const uint64_t wmask = __ballot(need);
                uint64_t first = 0; uint32_t got = 0, qx = 0, qs_ = 0;
                if (lane == 0) {
                    while (atomicCAS(misc + 2, 0u, 1u) != 0u) __builtin_amdgcn_s_sleep(1);
                    uint64_t pn = ((sq_vu64 *) pool)[0], pe = ((sq_vu64 *) pool)[1];
                    uint32_t qs = ((sq_vu32 *) misc)[3];
                    while (pn >= pe && qs < 8u) {                                // refill from the ray queues
                        const uint32_t x = (xcc + qs) & 7u;
                        const uint64_t len = (n_runs > x ? (n_runs - x + 7) / 8 : 0) * DRT_SQ_RUN;
                        const unsigned long long base = atomicAdd(P.queues + x, (unsigned long long) P.sq_chunk);
                        if (base < len) { pn = base; pe = base + P.sq_chunk < len ? base + P.sq_chunk : len; }
                        else ++qs;                                               // this queue is drained: next one
                    }
                    qx = (xcc + qs) & 7u; qs_ = qs;
                    const uint64_t want = (uint64_t) __popcll(wmask);
                    got = (uint32_t) (pe - pn < want ? pe - pn : want);
                    if (qs >= 8u) got = 0;
                    first = pn; pn += got;
                    ((sq_vu64 *) pool)[0] = pn; ((sq_vu64 *) pool)[1] = pe; ((sq_vu32 *) misc)[3] = qs;
                    sq_fence();
                    ((sq_vu32 *) misc)[2] = 0u;                                  // unlock
                }
                first = ((uint64_t) (uint32_t) __builtin_amdgcn_readfirstlane((int) (first >> 32)) << 32) | (uint32_t) __builtin_amdgcn_readfirstlane((int) first);
                got = (uint32_t) __builtin_amdgcn_readfirstlane((int) got); qx = (uint32_t) __builtin_amdgcn_readfirstlane((int) qx);
                qs_ = (uint32_t) __builtin_amdgcn_readfirstlane((int) qs_);
                if (qs_ >= 8u && !got) {                       // all eight queues are empty: these records are done
                    if (round == 0) { if (lane == 0) atomicAdd(misc, nb); retired = true; }
                    break;                                     // (a later round: the records that hold a ray go on, the others are retired by a later batch)
                }
                const uint32_t myr = (uint32_t) __popcll(wmask & ((1ull << lane) - 1ull));
                const bool take = need && myr < got;
                const uint64_t q = first + myr;
                if (take) {
                    uint64_t i = ((q / DRT_SQ_RUN) * 8 + qx) * DRT_SQ_RUN + (q % DRT_SQ_RUN);
                    if (P.order) {                                              // position -> unit of the order -> ray
                        const uint32_t g = (uint32_t) i, u = P.order_unit == 1u ? g : g / P.order_unit;
                        i = i < span ? (uint64_t) P.order[u] * P.order_unit + (g - u * P.order_unit) : P.n_rays;
                    }
                    unit_empty = false;
                    if (P.unit_empty && i + P.ray_first < P.n_rays) unit_empty = P.unit_empty[(uint32_t) i / P.empty_unit] != 0;
                    i += P.ray_first;
                    if (i < P.n_rays) {
                        // ---- sample() prologue (:51-108) + reach_medium (:292-319) ----
                        li = (uint32_t) i;
    #if DRT_SQ_PROFILE == 6
                        t_start = (uint32_t) __builtin_amdgcn_s_memrealtime() | 1u;
    #endif
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
                        SQ_COUNT(C_RAYS);
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
                        si_t = kInf;
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
                        ph = active ? SP_HEAD : SP_END;
                        {
                            // Rays that are over before they begin - they miss the medium's box, or their FIRST flight cannot collide (the pixel crosses
                            // only empty supergrid cells: unit_empty; or its target optical depth exceeds largest majorant x segment length: the flight
                            // set-up's early-out, which decides MOST rays of an optimisation that starts from a thin medium, scene_config.py:166-169) - are
                            // finished right here.  The blocks below would take such a ray through the loop head (one roulette draw; no roulette at depth
                            // 0 <= rr_depth), the flight set-up (one draw -> tau; the flight leaves the segment), the escape and the end of the path: two
                            // rounds of the pass loop for two draws, the path-cache entry "escaped", the emitter's radiance.  The same statements, the
                            // same draws, the same values - here.
                            bool over = !active;
                            if (active && DRT_SQ_EARLY_OUT && P.rr_depth >= 0 && (unit_empty || thin)) {
                                Pcg32 T = S;
                                (void) T.next_1d();                                 // :120 u_rr
                                const float tau0 = -drt_logf(1.0f - T.next_1d());   // the first flight's target optical depth
                                if (unit_empty || tau0 > (mmax * si_t) * 1.001f) {
                                    S = T; over = true; escaped = true;             // :244-245
                                    if (pc_on && 0 < (int) P.path_cache_cap)
                                        P.path_cache[(size_t) li * P.path_cache_cap * 2] = make_uint4(__float_as_uint(kInf), (uint32_t) S.state, (uint32_t) (S.state >> 32), 0u);
                                }
                            }
                            if (over) {
                                if (escaped && !P.hide_emitters) {                  // (depth 0: volpathsimple.py:263-285; mis_weight(1, 0) = 1)
                                    float Le[3];
                                    (void) emitter_eval_pdf<ENV>(P, rd, Le);
    #pragma unroll
                                    for (int k = 0; k < 3; ++k) result[k] += (beta[k] * 1.0f) * Le[k];
                                }
                                const size_t o3 = 3 * (size_t) li;
                                P.L_out[o3] = result[0]; P.L_out[o3 + 1] = result[1]; P.L_out[o3 + 2] = result[2];
                                if (P.ray_iters) P.ray_iters[li] = (uint8_t) 0;
                                ph = SP_IDLE;
                            }
                        }
                    }
                }
                // (rounds only in a THIN medium, where nearly every ray is over at once - an optimisation's first iterations: 431 -> 537 iterations/s at
                //  config 3's 16^3 level; in a thick one - the headline - they cost the primal launch 0.1 ms: profiles/r06_sq_instruction_budget.txt)
                if (DRT_SQ_REGEN_FINISH < 2 || !ROUNDS || !thin) break;
                // another round while at least DRT_SQ_REGEN_AGAIN of the batch's records are free again and the pool may hold more rays
                if (got < (uint32_t) __popcll(wmask) || __popcll(__ballot(act && ph == SP_IDLE)) < DRT_SQ_REGEN_AGAIN) break;
                }
                if (retired) continue;
            }
        } else {
            SQ_PROF(4, 1); SQ_PROF(5, nb);
        }

        if (kind == SQ_COLL) SQ_STAMP(3); else SQ_STAMP(5);
        // ================= (B) path transitions (transition / regeneration batches), (Fs) the next flight ==============
        // A pass takes every ray of the batch to its next walk (or to the end of its path); rays whose walk comes out of
        // the path cache (adjoint pass), or whose next flight cannot collide, go round once more.
        // ---- NEE walk finished (:388-403): at the head of a pass and, in the adjoint kernels, once more behind the emitter direction
        // block - a main path whose walks come out of the path cache then does a whole bounce (phase sampling, loop head, collision,
        // emitter direction, this block) in ONE pass
        float nee_pdf = 0.0f; bool nee_pdf_ok = false;                          // ENV: the emitter density of the direction the NEE block of THIS pass sampled
        auto rt_end_block = [&](bool behind_nee) {
            if constexpr (!ADJ) {
                if (ph == SP_RT_END && pc_on && pc_it < (int) P.path_cache_cap)
                    P.path_cache[((size_t) li * P.path_cache_cap + pc_it) * 2 + 1] =
                        make_uint4(__float_as_uint(wt), (uint32_t) S.state, (uint32_t) (S.state >> 32), pc_steps);
            }
            if (ph == SP_RT_END) {
                float val[3], contrib[3];
                // (recomputed from the direction; behind the NEE block of the same pass the density is the one it just evaluated)
                const float ds_pdf = (ENV && behind_nee && nee_pdf_ok) ? emitter_sample_value_with_pdf<ENV>(P, rd, nee_pdf, val)
                                                                       : emitter_sample_value<ENV>(P, rd, val);
                const float w = mis_weight(ds_pdf, kInvFourPi);             // :391
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    contrib[k] = ((beta[k] * kInvFourPi) * w) * (val[k] * wt);
                    result[k] = (ADJ && !rec_mode) ? result[k] - contrib[k] : result[k] + contrib[k];   // :211-214
                }
                ph = SP_PHASE;
                if constexpr (ADJ) {
                    if (!rec_mode) {                                        // replay with the clone (:393-401)
                        adjsum = (dL[0] * contrib[0] + dL[1] * contrib[1]) + dL[2] * contrib[2];
                        uint64_t tmp = S.state; S.state = Cst; Cst = tmp;
                        (void) S.next_1d(); (void) S.next_1d();             // same direction again (:418)
                        if (nt0 < kInf) { wo = ro; wmax = nt0; wt = 1.0f; ph = SP_RTA; fl = SF_NEW; }
                        else ph = SP_RTA_END;
                    }
                }
            }
        };
        // QUAD: the main path comes back from the DRT detour of a vertex (no vertex selected, or the recursive path has ended): the alt
        // sampler continues where the detour left it, everything else is as it was in the middle of the collision block
        auto quad_resume = [&](bool from_recursion) {
            if constexpr (QUAD) {
                if (from_recursion) { A.state = S.state; A.inc = S.inc; }       // (the recursion sampled with a copy of the alt sampler)
                S.state = Cst; S.inc = ((uint64_t) q_sinc_hi << 32) | q_sinc_lo;
                beta[0] = r_wsum[0]; beta[1] = r_wsum[1]; beta[2] = r_wsum[2];
                result[0] = q_result[0]; result[1] = q_result[1]; result[2] = q_result[2];
                depth = r_depth; ro = q_ro; rd = r_d; si_t = q_si_t; wt = q_wt;
                escaped = (q_flags >> 1) & 1u; has_scattered = (q_flags >> 2) & 1u; scat_once = (q_flags >> 3) & 1u;
                pc_on = (q_flags >> 4) & 1u; pc_it = (int) (q_flags >> 8);
                rec_mode = false; rec_first = false;
                ph = SP_QSCAT2;
            }
        };
        for (;;) {
            if ((SOLO || kind != SQ_COLL) && __ballot(sq_is_trans<QUAD>(ph))) {
                uint4 pce1 = make_uint4(0u, 0u, 0u, 0u); bool pce1_ok = false;   // this iteration's NEE entry of the path cache, read at the loop head
                SQ_BLK(8, sq_is_trans<QUAD>(ph)); SQ_BLK(0, ph == SP_DRT_END); SQ_BLK(1, ph == SP_RT_END || ph == SP_RTA_END);
                // ---- DRT vertex selected: enter the detached recursive path (:553-575, :610-655) -----
                if constexpr (ADJ) {
                    if (ph == SP_DRT_END) {
                        if (!(wo.z < kInf)) {                                   // no tentative collision (:558)
                            if constexpr (QUAD) quad_resume(false); else ph = SP_IDLE;
                        } else {
                            const V3 xp = ray_at(ro, rd, wo.z);
                            r_o = xp; ro = xp;
                            const float sig = eval_sigma_t(P, xp, occ);         // :553-554
                            r_si_t = sig;
                            SQ_COUNT(C_DRT);
                            const float w = P.use_drt_mis ? 1.0f / (1.0f + sig * sig) : 1.0f;
                            const float ww = w * wo.y;
                            r_cw[0] = ww * r_cw[0]; r_cw[1] = ww * r_cw[1]; r_cw[2] = ww * r_cw[2];
                            S = A;                                              // the recursion samples with alt_sampler
                            rec_mode = true; rec_first = true;
                            result[0] = result[1] = result[2] = 0.0f;
                            beta[0] = beta[1] = beta[2] = 1.0f;
                            depth = r_depth + 1;
                            escaped = false; scat_once = true; has_scattered = false;
                            ph = P.use_nee ? SP_NEE : SP_PHASE;                 // :621-624 NEE at x' whatever the depth
                        }
                    }
                }

                nee_pdf_ok = false;
                rt_end_block(false);
                if constexpr (ADJ) {
                    if (ph == SP_RTA_END) { S.state = Cst; ph = SP_PHASE; }     // back to the primary stream
                }

                // ---- phase sampling + new segment (:221-246) -------------------------------------------------------
                SQ_BLK(2, ph == SP_PHASE);
                if (ph == SP_PHASE) {
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
                    ph = active ? SP_HEAD : SP_END;
                }

                // ---- loop head: Russian roulette, start delta tracking (:116-127) -------------------------------------
                SQ_BLK(3, ph == SP_HEAD);
                if (ph == SP_HEAD) {
                    float q = fminf(fmaxf(beta[0], fmaxf(beta[1], beta[2])), 0.99f);
                    bool perform_rr = depth > P.rr_depth;
                    float u_rr = S.next_1d();
                    bool active = (beta[0] != 0.0f || beta[1] != 0.0f || beta[2] != 0.0f) && (!perform_rr || (u_rr < q));
                    if (perform_rr) { float iq = 1.0f / q; beta[0] *= iq; beta[1] *= iq; beta[2] *= iq; }
                    if (!active) ph = SP_END;
                    else if (ADJ && !rec_mode && pc_on && pc_it < (int) P.path_cache_cap) {
                        // the adjoint takes this iteration's delta-tracking walk from the primal pass of the same job
                        const uint4 *pce = P.path_cache + ((size_t) li * P.path_cache_cap + pc_it) * 2;
                        const uint4 e = pce[0];
                        pce1 = pce[1]; pce1_ok = true;                          // (adjacent: one round trip for both)
                        wt = __uint_as_float(e.x);                              // mei.t
                        S.state = ((uint64_t) e.z << 32) | e.y;
                        if (COUNT && !DRT_SQ_PROFILE) cnt[C_DT] += e.w;
                        ph = wt < kInf ? SP_SCAT : SP_ESC;
                    } else { wo = ro; wmax = si_t; wt = 0.0f; ph = SP_DT; fl = SF_NEW; pc_steps = 0; }
                }

                // ---- the walk found a real collision (wt = mei.t) or left the medium (:130-215, :244-245) -----------
                if constexpr (!ADJ) {                                           // path cache: what this iteration's walk returned
                    if ((ph == SP_SCAT || ph == SP_ESC) && pc_on && pc_it < (int) P.path_cache_cap)
                        P.path_cache[((size_t) li * P.path_cache_cap + pc_it) * 2] =
                            make_uint4(__float_as_uint(ph == SP_SCAT ? wt : kInf), (uint32_t) S.state, (uint32_t) (S.state >> 32), pc_steps);
                }
                SQ_BLK(4, ph == SP_SCAT || ph == SP_ESC);
                if (ph == SP_SCAT || ph == SP_ESC || (QUAD && ph == SP_QSCAT2)) {
                    const bool resumed = QUAD && ph == SP_QSCAT2;               // (back from the detour: the lookups again, not counted again)
                    const bool scat = resumed ? (q_flags & 1u) != 0u : ph == SP_SCAT;
                    const bool adj_lane = ADJ && !rec_mode;
                    float albedo[3] = { 1.0f, 1.0f, 1.0f }, mei_sig = 0.0f;
                    V3 mp = ro;
                    if (scat) {
                        mp = ray_at(ro, rd, wt);                                // :371
                        has_scattered = true;
                        if (adj_lane) { mei_sig = eval_sigma_t(P, mp, occ); if (!resumed) SQ_COUNT(C_DT); }   // :373-375
                        eval_albedo(P, mp, albedo);                             // :141
                        if (!resumed) SQ_COUNT(C_ALB);
                    }
                    bool detour = false;
                    if constexpr (QUAD) {
                        if (adj_lane && P.use_drt && !resumed) {
                            // backpropagate_scattering_drt at this vertex (:143-150, :543-581): suspend the main path ...
                            detour = true;
                            q_flags = (scat ? 1u : 0u) | (escaped ? 2u : 0u) | (has_scattered ? 4u : 0u) | (scat_once ? 8u : 0u) | (pc_on ? 16u : 0u) |
                                      ((uint32_t) min(pc_it, 1023) << 8);      // (the width of the record's field, below)
                            q_ro = ro; q_si_t = si_t; q_wt = wt; q_sinc_lo = (uint32_t) S.inc; q_sinc_hi = (uint32_t) (S.inc >> 32);
                            q_result[0] = result[0]; q_result[1] = result[1]; q_result[2] = result[2];
                            Cst = S.state; r_depth = depth; r_d = rd;
#pragma unroll
                            for (int k = 0; k < 3; ++k) { r_wsum[k] = beta[k]; r_cw[k] = dL[k] * beta[k]; }   // adj (:146)
                            // ... and walk the segment with sample_interaction_drt (:543-551), as the subsampled estimator does at the end of a path
                            wmax = isfinite(si_t) ? si_t : kLargest;
                            wt = 0.0f; wo = v3(1.0f, 0.0f, kInf);               // T, wsum, selected t
                            ph = SP_DRT; fl = SF_NEW;
                        }
                    }
                    if constexpr (ADJ) {
                        if (adj_lane && !detour) {
                            if (!QUAD && P.use_drt) {                           // DRTReservoir.update :745-753
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
                                splat_scatter<true>(P, mp, gs, ga, rec); SQ_COUNT(C_SC); SQ_COUNT(C_SC_ALB);
                            }
                            // backpropagate_transmittance: 4 resampled points on the segment (:181-189, :584-607)
                            const float tr_int = scat ? wt : si_t;
                            const float tr_g = -(((dL[0] * result[0] + dL[1] * result[1]) + dL[2] * result[2]) * (tr_int / 4.0f));
                            V3 pts[4];
#pragma unroll
                            for (int j = 0; j < 4; ++j) {
                                const float u = A.next_1d();                    // :595
                                pts[j] = ray_at(ro, rd, u * tr_int);
                                SQ_COUNT(C_TR);
                            }
                            if (tr_g != 0.0f) emit_records0<4>(P, pts, tr_g * P.scale, rec);
                        }
                    }
                    if (detour) { }                                             // (the rest of the block when the main path is back)
                    else if (scat) {
                        beta[0] *= albedo[0]; beta[1] *= albedo[1]; beta[2] *= albedo[2];   // :193
                        depth += 1;                                             // :199
                        ro = mp;
                        if (depth < P.max_depth) ph = P.use_nee ? SP_NEE : SP_PHASE;   // :200, :206-207
                        else {
                            ph = SP_END;           // killed inside the medium; its phase draws (:221-222) are unobservable ...
                            if constexpr (QUAD) {  // ... except on a recursive path of the quadratic estimator: the main path's alt sampler continues behind them
                                if (rec_mode) { (void) S.next_1d(); (void) S.next_1d(); (void) S.next_1d(); }
                            }
                        }
                    } else {
                        escaped = true;                                         // :245
                        ph = SP_END;
                    }
                }

                // ---- emitter direction + boundary exit for NEE (:406-433) ------------------------------------------
                SQ_BLK(5, ph == SP_NEE);
                if (ph == SP_NEE) {
                    if (ADJ && !rec_mode) Cst = S.state;                        // :383
                    float ux = S.next_1d(), uy = S.next_1d();                   // :418
                    rd = emitter_sample_dir<ENV>(P, ux, uy);
                    Hit h = box_hit(P, ro, rd);                                 // :427-428
                    if constexpr (ENV) { nee_pdf = envmap_pdf(P, rd); nee_pdf_ok = true; if (nee_pdf == 0.0f) h.valid = false; }   // sampling_worked :421-423
                    pc_steps = 0;
                    nt0 = h.valid ? h.t : kInf;
                    if (ADJ && !rec_mode && pc_on && pc_it < (int) P.path_cache_cap) {
                        // the value walk of the main path comes out of the path cache: transmittance, stream, steps
                        uint4 e = pce1;
                        if (!pce1_ok) e = P.path_cache[((size_t) li * P.path_cache_cap + pc_it) * 2 + 1];
                        wt = __uint_as_float(e.x);
                        S.state = ((uint64_t) e.z << 32) | e.y;
                        if (COUNT && !DRT_SQ_PROFILE) cnt[C_RT] += e.w;
                        ph = SP_RT_END;
                    } else if (h.valid) { wo = ro; wmax = h.t; wt = 1.0f; ph = SP_RT; fl = SF_NEW; }
                    else { wt = 0.0f; ph = SP_RT_END; }
                }
                // ---- end of a path (:249-287) -----------------------------------------------------
                if constexpr (ADJ && DRT_SQ_RT2) {
                    if (__ballot(ph == SP_RT_END)) rt_end_block(true);          // (the value walk came out of the path cache)
                }
                SQ_BLK(6, ph == SP_END);
                if (ph == SP_END) {
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
                        ph = SP_IDLE;
                    } else {
                        if (rec_mode) {
                            // result = Li': gradient splat at x' (:577-581)
                            if (!b_live) {                                      // (a recursive path loaded without the main path's state)
                                const uint4 *cb = cold_b + NB * id;
                                const uint4 c3 = cb[0], c5 = cb[2], c7 = cb[4];
#if DRT_SQ_PROFILE == 6
                                t_start = cb[1].w;
#endif
                                r_si_t = __uint_as_float(c3.w);
                                r_o = v3(__uint_as_float(c5.x), __uint_as_float(c5.y), __uint_as_float(c5.z));
                                r_cw[0] = __uint_as_float(c7.x); r_cw[1] = __uint_as_float(c7.y); r_cw[2] = __uint_as_float(c7.z);
                                if constexpr (QUAD) {                           // ... and everything of the suspended main path
                                    const uint4 c4 = cb[1], c6 = cb[3], c9 = cb[6], c10 = cb[7], c11 = cb[8];
                                    dL[0] = __uint_as_float(c3.x); dL[1] = __uint_as_float(c3.y); dL[2] = __uint_as_float(c3.z);
                                    Cst = ((uint64_t) c4.y << 32) | c4.x; r_depth = (int) c4.z;
                                    r_wsum[0] = __uint_as_float(c5.w); r_wsum[1] = __uint_as_float(c6.w); r_wsum[2] = __uint_as_float(c7.w);
                                    r_d = v3(__uint_as_float(c6.x), __uint_as_float(c6.y), __uint_as_float(c6.z));
                                    q_ro = v3(__uint_as_float(c9.x), __uint_as_float(c9.y), __uint_as_float(c9.z)); q_si_t = __uint_as_float(c9.w);
                                    q_result[0] = __uint_as_float(c10.x); q_result[1] = __uint_as_float(c10.y); q_result[2] = __uint_as_float(c10.z); q_wt = __uint_as_float(c10.w);
                                    q_flags = c11.x; q_sinc_lo = c11.y; q_sinc_hi = c11.z;
                                    b_live = true;                              // (stored with the resumed main path)
                                }
                            }
                            float alb[3];
                            eval_albedo(P, r_o, alb);                           // :578
                            SQ_COUNT(C_ALB);
                            float gs = 0.0f, ga[3];
#pragma unroll
                            for (int k = 0; k < 3; ++k) {
                                float a = r_cw[k] * result[k];
                                gs += a * alb[k];
                                ga[k] = a * r_si_t;
                            }
                            splat_scatter<true>(P, r_o, gs, ga, rec); SQ_COUNT(C_SC); SQ_COUNT(C_SC_ALB);
                            if constexpr (QUAD) quad_resume(true); else ph = SP_IDLE;
                        } else if (!QUAD && P.use_drt && r_depth >= 0) {        // :249-259, DRTReservoir.get :756-760
                            const float d = ((r_cw[0] + r_cw[1]) + r_cw[2]) / 3.0f;
                            const float ws = ((r_wsum[0] + r_wsum[1]) + r_wsum[2]) / 3.0f;
#pragma unroll
                            for (int k = 0; k < 3; ++k) r_cw[k] = (d != 0.0f ? (ws * r_cw[k]) / d : 0.0f) * dL[k];   // adjoint
                            // sample_interaction_drt along the selected segment (:543-551)
                            wmax = isfinite(r_si_t) ? r_si_t : kLargest;
                            ro = r_o; rd = r_d;
                            wt = 0.0f; wo = v3(1.0f, 0.0f, kInf);               // T, wsum, selected t
                            ph = SP_DRT; fl = SF_NEW;
                        } else {
                            ph = SP_IDLE;
                        }
                    }
                }
            }

            // ================= (Fs) the next flight: set up and post =========================================
            SQ_STAMP(5);
            {
                const bool setup = ph < SP_HEAD && fl != SF_WAIT;
                SQ_BLK(7, setup);
                if (__ballot(setup)) {
                    if (setup) {
                        const bool drt = ph == SP_DRT;
                        const bool useA = ADJ && !rec_mode && drt;
                        Pcg32 Rg; Rg.state = useA ? A.state : S.state; Rg.inc = useA ? A.inc : S.inc;
                        // the direction's share of the DDA (oracle: sample_collision): crossing-time increments 1 / |dg|, direction
                        // signs.  It is the same for every flight of a walk: kept in the record, recomputed for a walk's first flight.
                        float tdx, tdy, tdz; int sgx, sgy, sgz;
                        if (fl == SF_NEW) {
                            const float dgx = (rd.x * P.inv_ext[0]) * fgx, dgy = (rd.y * P.inv_ext[1]) * fgy, dgz = (rd.z * P.inv_ext[2]) * fgz;
                            if (dgx >= 1e-20f) { tdx = 1.0f / dgx; sgx = 1; } else if (dgx <= -1e-20f) { tdx = 1.0f / -dgx; sgx = -1; } else { tdx = kInf; sgx = 0; }
                            if (dgy >= 1e-20f) { tdy = 1.0f / dgy; sgy = 1; } else if (dgy <= -1e-20f) { tdy = 1.0f / -dgy; sgy = -1; } else { tdy = kInf; sgy = 0; }
                            if (dgz >= 1e-20f) { tdz = 1.0f / dgz; sgz = 1; } else if (dgz <= -1e-20f) { tdz = 1.0f / -dgz; sgz = -1; } else { tdz = kInf; sgz = 0; }
                        } else {
                            tdx = w_tdx; tdy = w_tdy; tdz = w_tdz;                  // (the walk's previous flight left them in the record)
                            sgx = tdx == kInf ? 0 : (w_rem & (1u << 27)) ? -1 : 1;
                            sgy = tdy == kInf ? 0 : (w_rem & (1u << 28)) ? -1 : 1;
                            sgz = tdz == kInf ? 0 : (w_rem & (1u << 29)) ? -1 : 1;
                        }
                        const float u = Rg.next_1d();
                        const float tau = -drt_logf(1.0f - u);
                        const V3 o = drt ? ray_at(ro, rd, wt) : wo;
                        const float tmax = drt ? wmax - wt : wmax;
                        const float gxf = ((o.x - P.bmin[0]) * P.inv_ext[0]) * fgx;
                        const float gyf = ((o.y - P.bmin[1]) * P.inv_ext[1]) * fgy;
                        const float gzf = ((o.z - P.bmin[2]) * P.inv_ext[2]) * fgz;
                        const float flx = fminf(fmaxf(floorf(gxf), 0.0f), (float) (gx - 1));
                        const float fly_ = fminf(fmaxf(floorf(gyf), 0.0f), (float) (gy - 1));
                        const float flz = fminf(fmaxf(floorf(gzf), 0.0f), (float) (gz - 1));
                        const int cx = (int) flx, cy = (int) fly_, cz = (int) flz;
                        const float tnx = sgx > 0 ? ((flx + 1.0f) - gxf) * tdx : sgx < 0 ? (gxf - flx) * tdx : kInf;
                        const float tny = sgy > 0 ? ((fly_ + 1.0f) - gyf) * tdy : sgy < 0 ? (gyf - fly_) * tdy : kInf;
                        const float tnz = sgz > 0 ? ((flz + 1.0f) - gzf) * tdz : sgz < 0 ? (gzf - flz) * tdz : kInf;
                        const uint32_t rx_ = (uint32_t) (sgx > 0 ? gx - 1 - cx : cx), ry_ = (uint32_t) (sgy > 0 ? gy - 1 - cy : cy),
                                       rz_ = (uint32_t) (sgz > 0 ? gz - 1 - cz : cz);
                        const uint32_t rem = rx_ | (ry_ << 9) | (rz_ << 18) | (sgx < 0 ? 1u << 27 : 0u) | (sgy < 0 ? 1u << 28 : 0u) | (sgz < 0 ? 1u << 29 : 0u);
                        // ... nor are the flights along the primary segment of a ray whose pixel crosses only empty cells (build_unit_empty): the delta-tracking
                        // walk of its first bounce-loop iteration, the DRT sampler's walk along that segment - optical depth exactly 0, they leave the segment
                        const bool all_empty = unit_empty && fl == SF_NEW && !rec_mode && (drt ? r_depth == 0 : (ph == SP_DT && depth == 0));
                        if (DRT_SQ_EARLY_OUT && (tau > (mmax * tmax) * 1.001f || all_empty)) {
                            // no cell of this segment can bring the optical depth to tau: the flight leaves the segment, as
                            // the epilogue above finds it after a walk (majorant 0 in the record: dt = inf, not inside)
                            if (drt) wt += kInf;
                            fl = SF_NEXT;
                            w_tdx = tdx; w_tdy = tdy; w_tdz = tdz; w_rem = rem;
                            ph = drt ? SP_DRT_END : (ph == SP_DT) ? SP_ESC : (ph == SP_RT ? SP_RT_END : SP_RTA_END);
                        } else {
                            // The flight's first cells right here (the walker's cell step, the same arithmetic in the same order):
                            // half of all flights end within four cells and never see the walkers' queue - their result goes into
                            // the record as a walker leaves it, and the ray to the collision queue.
                            float wnx = tnx, wny = tny, wnz = tnz, wt_ = 0.0f, wacc = 0.0f, res_mc = 0.0f;
                            int wcell = (cz * gy + cy) * gx + cx;
                            uint32_t wrem = rem;
                            const int sx = sgx < 0 ? -1 : 1, sy = sgy < 0 ? -lin_y : lin_y, sz = sgz < 0 ? -lin_z : lin_z;
                            bool wfly = true;
                            do {                                                    // (SOLO: to the flight's end)
#pragma unroll
                            for (int k = 0; k < (MG ? 0 : DRT_SQ_INLINE_K); ++k) {
                                const float tmin = fminf(fminf(wnx, wny), wnz);
                                const float texit = fminf(tmin, tmax);
                                const float mc = __uint_as_float((uint32_t) mg16[wcell] << 16);
                                const float nacc = wacc + mc * (texit - wt_);
                                const bool hit = mc > 0.0f && nacc >= tau;
                                const bool isx = wnx == tmin, isy = !isx && wny == tmin;
                                const uint32_t sh = isx ? 0u : isy ? 9u : 18u;
                                const bool end = !(texit < tmax) || ((wrem >> sh) & 511u) == 0u;
                                const float tnn = tmin + (isx ? tdx : isy ? tdy : tdz);
                                if (wfly && (hit || end)) { res_mc = hit ? mc : 0.0f; wfly = false; }
                                const bool go = wfly;
                                wacc = go ? nacc : wacc;
                                wt_ = go ? texit : wt_;
                                wrem = go ? wrem - (1u << sh) : wrem;
                                wcell += go ? (isx ? sx : isy ? sy : sz) : 0;
                                wnx = (go && isx) ? tnn : wnx; wny = (go && isy) ? tnn : wny; wnz = (go && !isx && !isy) ? tnn : wnz;
                            }
                            } while (SOLO && wfly);
                            if constexpr (SOLO) {                                   // the finished flight stays in registers: its collision follows below
                                c_lm = res_mc; c_tau = tau; c_t = wt_; c_acc = wacc;
                                w_tdx = tdx; w_tdy = tdy; w_tdz = tdz; w_rem = wrem;
                            }
                            R[0] = make_uint4(wfly ? __float_as_uint(wnx) : __float_as_uint(res_mc), __float_as_uint(wny), __float_as_uint(wnz), (uint32_t) wcell);
                            R[1] = make_uint4(__float_as_uint(tdx), __float_as_uint(tdy), __float_as_uint(tdz), wrem);
                            R[2] = make_uint4(__float_as_uint(tau), __float_as_uint(tmax), __float_as_uint(wt_), __float_as_uint(wacc));
                            fl = SF_WAIT;
                            walk_done = !wfly;
                        }
                        if (useA) A.state = Rg.state; else S.state = Rg.state;
                    }
                }
            }
            SQ_STAMP(6);
            if constexpr (SOLO) {
                if (__ballot(act && walk_done)) { collide(act && walk_done); walk_done = false; }
                // round again while a ray has a transition to make or a flight to set up
                if (!__ballot(act && (sq_is_trans<QUAD>(ph) || (ph < SP_HEAD && fl != SF_WAIT)))) break;
            } else {
                if (kind == SQ_COLL || __popcll(__ballot(sq_is_trans<QUAD>(ph))) < DRT_SQ_T_PASS) break;   // (what is left goes to the transition queue)
            }
        }

        // ================= store the rays, hand them on ====================================================
        const bool go_walk = act && ph < SP_HEAD;                               // (posted: fl == SF_WAIT)
        const bool go_trans = act && sq_is_trans<QUAD>(ph);            // (collision batches only: the walk ended)
        const bool go_free = act && ph == SP_IDLE;
#if DRT_SQ_PROFILE == 6
        if (COUNT && go_free && t_start != 0u) {
            const uint32_t now = (uint32_t) __builtin_amdgcn_s_memrealtime(), el = now - (uint32_t) pt_start6, age = now - t_start;
            const uint32_t b = min(31u, el / 25000u) * 5u;
            atomicAdd(pdbg + b, 1u); atomicAdd(pdbg + b + 1, age >> 7); atomicAdd(pdbg + b + 2, (uint32_t) pc_it);
            if (2u * age > el) atomicAdd(pdbg + b + 3, 1u);
            atomicMax(pdbg + b + 4, age >> 7);
        }
#endif
        if (go_walk || go_trans) {
            const bool drtw = ph == SP_DRT || ph == SP_DRT_END;
            const bool gA = ADJ && !rec_mode && drtw;
            const uint64_t gs = gA ? A.state : S.state, gi = gA ? A.inc : S.inc;
            R[3] = make_uint4(__float_as_uint(rd.x), __float_as_uint(rd.y), __float_as_uint(rd.z), __float_as_uint(wmax));
            R[4] = make_uint4(__float_as_uint(wo.x), __float_as_uint(wo.y), __float_as_uint(wo.z), __float_as_uint(wt));
            R[5] = make_uint4((uint32_t) gs, (uint32_t) (gs >> 32), (uint32_t) gi, (uint32_t) (gi >> 32));
            // (depth and pc_it have 10 bits each: sq_supported bounds max_depth by 1000; pc_it - the bounce-loop iteration, i.e. the path cache index
            //  of the MAIN path, used while pc_it < path_cache_cap <= 64 - keeps counting on recursive and quadratic-detour paths, where nothing reads
            //  it, and saturates at 1023; the main path's value comes back from q_flags when a detour ends)
            const uint32_t f = (uint32_t) ph | ((uint32_t) fl << 4) | (rec_mode ? 1u << 6 : 0u) | (rec_first ? 1u << 7 : 0u) | (escaped ? 1u << 8 : 0u) |
                               (has_scattered ? 1u << 9 : 0u) | (scat_once ? 1u << 10 : 0u) | (pc_on ? 1u << 11 : 0u) |
                               ((uint32_t) min(depth, 1023) << 12) | ((uint32_t) min(pc_it, 1023) << 22);
            // (component by component: `drtw ? make_uint4(..) : make_uint4(..)` stored adjsum in .x whatever drtw said - hipcc 7.2)
#if DRT_SQ_PROFILE == 6
            R[6] = make_uint4(drtw ? __float_as_uint(ro.x) : __float_as_uint(adjsum), drtw ? __float_as_uint(ro.y) : pc_steps,
                              drtw ? __float_as_uint(ro.z) : (ADJ ? 0u : t_start), f);
#else
            R[6] = make_uint4(drtw ? __float_as_uint(ro.x) : __float_as_uint(adjsum), drtw ? __float_as_uint(ro.y) : pc_steps,
                              drtw ? __float_as_uint(ro.z) : 0u, f);
#endif
            if (SOLO || kind != SQ_COLL) {
                uint4 *ca = cold_a + 3 * id;
                ca[0] = make_uint4(__float_as_uint(ro.x), __float_as_uint(ro.y), __float_as_uint(ro.z), __float_as_uint(si_t));
                ca[1] = make_uint4(__float_as_uint(beta[0]), __float_as_uint(beta[1]), __float_as_uint(beta[2]), __float_as_uint(nt0));
                ca[2] = make_uint4(__float_as_uint(result[0]), __float_as_uint(result[1]), __float_as_uint(result[2]), li);
                if constexpr (ADJ) {
                    if (b_live) {
                        uint4 *cb = cold_b + NB * id;
                        const uint64_t os = gA ? S.state : A.state, oi = gA ? S.inc : A.inc;
                        cb[0] = make_uint4(__float_as_uint(dL[0]), __float_as_uint(dL[1]), __float_as_uint(dL[2]), __float_as_uint(r_si_t));
#if DRT_SQ_PROFILE == 6
                        cb[1] = make_uint4((uint32_t) Cst, (uint32_t) (Cst >> 32), (uint32_t) r_depth, t_start);
#else
                        cb[1] = make_uint4((uint32_t) Cst, (uint32_t) (Cst >> 32), (uint32_t) r_depth, 0u);
#endif
                        cb[2] = make_uint4(__float_as_uint(r_o.x), __float_as_uint(r_o.y), __float_as_uint(r_o.z), __float_as_uint(r_wsum[0]));
                        cb[3] = make_uint4(__float_as_uint(r_d.x), __float_as_uint(r_d.y), __float_as_uint(r_d.z), __float_as_uint(r_wsum[1]));
                        cb[4] = make_uint4(__float_as_uint(r_cw[0]), __float_as_uint(r_cw[1]), __float_as_uint(r_cw[2]), __float_as_uint(r_wsum[2]));
                        cb[5] = make_uint4((uint32_t) os, (uint32_t) (os >> 32), (uint32_t) oi, (uint32_t) (oi >> 32));
                        if constexpr (QUAD) {
                            cb[6] = make_uint4(__float_as_uint(q_ro.x), __float_as_uint(q_ro.y), __float_as_uint(q_ro.z), __float_as_uint(q_si_t));
                            cb[7] = make_uint4(__float_as_uint(q_result[0]), __float_as_uint(q_result[1]), __float_as_uint(q_result[2]), __float_as_uint(q_wt));
                            cb[8] = make_uint4(q_flags, q_sinc_lo, q_sinc_hi, 0u);
                        }
                    }
                }
            }
            if (!SOLO && go_trans && kind == SQ_COLL) { R[1] = make_uint4(__float_as_uint(w_tdx), __float_as_uint(w_tdy), __float_as_uint(w_tdz), w_rem); }
        }
        if (SOLO || kind != SQ_COLL) __threadfence_block();                    // (the global part of the records)
        sq_fence();
        const int tk = sq_trans_kind<DRT_SQ_SPLIT == 2 || (DRT_SQ_SPLIT == 1 && ADJ)>(ph);
#if DRT_SQ_PUSH_ALL
        sq_push_all(ctl, q_lds, go_walk ? (walk_done ? SQ_COLL : SQ_WALK) : go_trans ? tk : go_free ? SQ_REGEN : SQ_KINDS, id, lane);
#else
        sq_push(ctl, q_lds, SQ_WALK, go_walk && !walk_done, id, lane);
        sq_push(ctl, q_lds, SQ_COLL, go_walk && walk_done, id, lane);
        sq_push(ctl, q_lds, SQ_TA, go_trans && tk == SQ_TA, id, lane);
        sq_push(ctl, q_lds, SQ_TB, go_trans && tk == SQ_TB, id, lane);
        sq_push(ctl, q_lds, SQ_REGEN, go_free, id, lane);
#endif
        SQ_STAMP(7);
    }

    {
        if (!TAILM && P.tail_pool) {
            // hand-over: every wave is out of the loop with no record in its registers (a batch is stored and queued before the loop head is seen again):
            // what the queues hold goes to the pool - wave k the entries of queue kind k, three records per round (lane = 20 x record + quad)
            __syncthreads();
            // The decision inside the loop came from two LDS reads that are not one snapshot (records retired between them count twice): here the
            // count is exact.  More than the workgroup's share of the pool (the host sizes it as workgroups x sq_tail_push()): nothing is handed
            // over or dropped - the flag is taken back, the threshold set to 0 and the workgroup finishes its records itself.
            if ((((sq_vu32 *) misc)[0] & 0x80000000u) != 0u) {
                uint32_t queued = 0;
#pragma unroll
                for (int k = 0; k <= SQ_TB; ++k) { const unsigned long long c = ((sq_vu64 *) ctl)[k]; queued += (uint32_t) (c >> 32) - (uint32_t) c; }
                if (queued > (uint32_t) (DRT_SQ_TAIL_PUSH + 64)) {
                    __syncthreads();
                    if (threadIdx.x == 0) { ((sq_vu32 *) misc)[4] = 0u; atomicAnd(misc, 0x7fffffffu); }
                    __syncthreads();
                    continue;
                }
            }
            if ((((sq_vu32 *) misc)[0] & 0x80000000u) != 0u && wave <= SQ_TB) {
                const unsigned long long c = ((sq_vu64 *) ctl)[wave];
                const uint32_t head = (uint32_t) c, nq = (uint32_t) (c >> 32) - head;
                uint32_t base = 0;
                if (lane == 0u && nq) base = atomicAdd(P.tail_count, nq);
                base = (uint32_t) __builtin_amdgcn_readfirstlane((int) base);
                const uint32_t sub = lane / (uint32_t) kSqTailQuads, quad = lane - sub * (uint32_t) kSqTailQuads;
#pragma unroll 1
                for (uint32_t r = sub; r < nq; r += 3u) {
                    if (sub < 3u && base + r < P.tail_cap && quad < (ADJ ? 11u + NB : 11u)) {
                        const uint32_t id_r = q_lds[wave * DRT_SQ_RING + ((head + r) & (DRT_SQ_RING - 1u))];
                        uint4 v;
                        if (quad < 7u) v = rec4[R4 * id_r + quad];
                        else if (quad == 7u) v = make_uint4((uint32_t) wave, 0u, 0u, 0u);
                        else if (quad < 11u) v = cold_a[3 * id_r + (quad - 8u)];
                        else v = ADJ ? cold_b[NB * id_r + (quad - 11u)] : make_uint4(0u, 0u, 0u, 0u);
                        P.tail_pool[(size_t) (base + r) * kSqTailQuads + quad] = v;
                    }
                }
            }
        }
        if constexpr (ADJ) close_records(P, rec);
    }
    break;
    }
#if DRT_SQ_PROFILE == 6
    __syncthreads();
    if (COUNT && threadIdx.x < 160) {
        const uint32_t v = pdbg[threadIdx.x];
        if (threadIdx.x % 5 == 4) atomicMax(g_sq_dbg + threadIdx.x, (unsigned long long) v);
        else if (v) atomicAdd(g_sq_dbg + threadIdx.x, (unsigned long long) v);
    }
#endif
#if DRT_SQ_PROFILE == 5
    if (COUNT && lane == 0) {
        const unsigned long long te = __builtin_amdgcn_s_memrealtime(), q62 = 1ull << 62;
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
#undef SQ_COUNT
#undef SQ_PROF
#undef SQ_STAMP
#undef SQ_BLK
}

// Records per workgroup that fit LDS next to this supergrid's majorants (a multiple of 64); 0: this supergrid cannot be
// served (the host keeps drt_super.hip).  *bytes: dynamic LDS of the launch
// *global_majorants: the bf16 majorants do not fit (with at least DRT_SQ_MIN_RAYS records) but one bit per cell does - the MG kernels
static uint32_t sq_rays_for(const Params &P, size_t *bytes, bool *global_majorants = nullptr)
{
    const size_t cells = (size_t) P.gx * P.gy * P.gz;
    const size_t nwv = DRT_SQ_THREADS / 64;
    const size_t cap = 160u * 1024u;
#ifndef DRT_SQ_FORCE_MG
#define DRT_SQ_FORCE_MG 0          // experiment: 1 = majorants from L2 for every supergrid (more records fit LDS)
#endif
    for (int mg = DRT_SQ_FORCE_MG; mg < 2; ++mg) {
        if (mg && !(P.mocc && P.majorant)) break;
        const size_t words = mg ? (cells + 31) / 32 : (cells + 1) / 2;
        const size_t fixed = (((words + 3) & ~(size_t) 3) + (size_t) SQ_KINDS * DRT_SQ_RING / 2 + 2 * SQ_KINDS + 4 + 8 + nwv * 8 + (DRT_SQ_PROFILE == 6 ? 160 : 0)) * 4;
        if (fixed >= cap) continue;
        size_t n = ((cap - fixed) / (7 * 16)) & ~(size_t) 63;
        if (n > DRT_SQ_MAX_RAYS) n = DRT_SQ_MAX_RAYS;
        if (n < DRT_SQ_MIN_RAYS) continue;
        if (bytes) *bytes = fixed + n * 7 * 16;
        if (global_majorants) *global_majorants = mg != 0;
        return (uint32_t) n;
    }
    return 0;
}

// the tail launch runs its records to their ends without queue hops (SOLO) unless the majorants are read from L2 (MG)
bool sq_tail_solo(const Params &P)
{
    bool mg = false;
    return sq_rays_for(P, nullptr, &mg) != 0 && !mg;
}

uint32_t sq_tail_push() { return DRT_SQ_TAIL_PUSH + 64; }   // (the hand-over's count of live records may be low by one batch)
size_t sq_tail_entry_quads() { return kSqTailQuads; }

size_t sq_cold_bytes(int n_cus) { return (size_t) n_cus * 12 * DRT_SQ_MAX_RAYS * sizeof(uint4); }   // (3 + 9: the quadratic estimator's adjoint records)

bool sq_supported(const Params &P)
{
    return P.mgrid && P.gx <= 511 && P.gy <= 511 && P.gz <= 511 && P.max_depth <= 1000 && sq_rays_for(P, nullptr) != 0;
}

hipError_t launch_trace_sq(const Params &Pin, bool adjoint, bool count, int n_cus, hipStream_t stream)
{
    if (Pin.n_rays <= Pin.ray_first) return hipSuccess;
    size_t lds = 0;
    bool mg = false;
    const uint32_t nray = sq_rays_for(Pin, &lds, &mg);
    if (!nray || !Pin.sq_cold) return hipErrorInvalidValue;
    Params P = Pin;
    P.sq_rays = nray;
    unsigned blocks = (unsigned) n_cus;                                         // one workgroup per CU
    const uint64_t need = (P.n_rays - P.ray_first + nray - 1) / nray;           // no more workgroups than groups of records
    if (need < blocks && !P.tail_mode) blocks = (unsigned) need;
    P.sq_chunk = DRT_SQ_CHUNK;
    if (!P.order && !P.tail_mode) {                                             // index order: fewer, larger refills (see DRT_SQ_CHUNK_MAX)
        const uint64_t c = (P.n_rays - P.ray_first) / (32ull * blocks);
        P.sq_chunk = (uint32_t) (c < DRT_SQ_CHUNK ? DRT_SQ_CHUNK : c > DRT_SQ_CHUNK_MAX ? DRT_SQ_CHUNK_MAX : c) & ~63u;
    }
    // the tail launch: the pool's (few thousand) records over a few workgroups (tail_mode 1: the partition passes of the reduction run beside it) or the chip (2)
    if (P.tail_mode == 1u) {
        const unsigned fit = (unsigned) ((P.tail_cap + nray - 1) / nray);           // (every pool entry needs a record)
        const unsigned want = fit > (unsigned) DRT_SQ_TAIL_BLOCKS ? fit : (unsigned) DRT_SQ_TAIL_BLOCKS;
        blocks = blocks < want ? blocks : want;
    }
    else if (!P.tail_mode && P.tail_pool && (uint64_t) P.tail_cap < (uint64_t) blocks * (DRT_SQ_TAIL_PUSH + 64)) P.tail_pool = nullptr;   // (capacity invariant of the hand-over)
    dim3 block(DRT_SQ_THREADS), grid(blocks);
    const bool env = P.env_pix != nullptr;
    hipError_t e = hipSuccess;
    const bool quad = adjoint && P.use_drt && !P.use_drt_subsampling;           // quadratic DRT: the QUAD instantiations of the adjoint kernels
    const bool tailm = P.tail_mode != 0u;
    const bool rounds = DRT_SQ_REGEN_FINISH >= 2 && P.sq_rounds != 0u && !P.order;
#define DRT_SQ_LAUNCH(A, C, E) do { if (tailm) DRT_SQ_LAUNCH_Q(A, C, E, true); else DRT_SQ_LAUNCH_Q(A, C, E, false); } while (0)
#define DRT_SQ_LAUNCH_Q(A, C, E, T) do { if (A && quad) { if (mg) DRT_SQ_LAUNCH_(A, C, E, true, A, T); else DRT_SQ_LAUNCH_(A, C, E, false, A, T); } \
                                    else { if (mg) DRT_SQ_LAUNCH_(A, C, E, true, false, T); else DRT_SQ_LAUNCH_(A, C, E, false, false, T); } } while (0)
#define DRT_SQ_LAUNCH_(A, C, E, M, Q, T) do { if (!(A) && !(C) && !(T) && rounds) DRT_SQ_LAUNCH_R(A, C, E, M, Q, T, (!(A) && !(C) && !(Q) && !(T))); \
                                              else DRT_SQ_LAUNCH_R(A, C, E, M, Q, T, false); } while (0)
#define DRT_SQ_LAUNCH_R(A, C, E, M, Q, T, R)                                                                         \
    do {                                                                                                          \
        auto kern = trace_sq_kernel<A, C, E, M, Q, T, R>;                                                         \
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
    const int variant = (adjoint ? 4 : 0) | (count ? 2 : 0) | (env ? 1 : 0);
    switch (variant) {
        case 0: DRT_SQ_LAUNCH(false, false, false); break;
        case 1: DRT_SQ_LAUNCH(false, false, true); break;
        case 2: DRT_SQ_LAUNCH(false, true, false); break;
        case 3: DRT_SQ_LAUNCH(false, true, true); break;
        case 4: DRT_SQ_LAUNCH(true, false, false); break;
        case 5: DRT_SQ_LAUNCH(true, false, true); break;
        case 6: DRT_SQ_LAUNCH(true, true, false); break;
        default: DRT_SQ_LAUNCH(true, true, true); break;
    }
#undef DRT_SQ_LAUNCH
#undef DRT_SQ_LAUNCH_Q
#undef DRT_SQ_LAUNCH_
#undef DRT_SQ_LAUNCH_R
    return hipGetLastError();
}

}  // namespace drt
