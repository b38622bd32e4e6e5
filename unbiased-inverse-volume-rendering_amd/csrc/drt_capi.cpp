// drt_capi.cpp -- the C ABI of libdrt_hip.so (include/drt_hip.h).
//
// Host-side state of one integrator instance bound to one GPU: the borrowed
// parameter pointers, the device-resident majorant, the emitter / sensor, and the
// stream all work is enqueued on.  No C++ exceptions cross the ABI.
#include "../../include/drt_hip.h"
#include "drt_launch.h"

#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <utility>
#include <vector>

struct drt_handle_s {
    int device = 0;
    hipStream_t stream = nullptr;
    drt_config cfg{};
    drt::Params base{};            // scene part of the kernel parameter block
    bool have_medium = false, have_emitter = false, have_sensor = false;
    float *d_majorant = nullptr;   // [2]
    // the global majorant as the HOST last saw it: drt_params_changed copies it to pinned memory behind the reduction and records an event; launches
    // look at it when the event has completed (never waiting) - a hint for kernel choice only (a thin medium: Params::sq_rounds), stale by design
    float *h_majorant = nullptr;   // pinned, [1]
    hipEvent_t ev_majorant = nullptr;
    bool majorant_pending = false;
    float majorant_seen = -1.0f;   // < 0: never seen
    uint32_t *d_scratch = nullptr; // [1]
    unsigned long long *d_counters = nullptr;   // [C_COUNT]
    float *d_gt = nullptr;         // gradient scratch, 4 planes (always zero between launches)
    unsigned long long *d_queues = nullptr;   // 8 per-XCD ray queue heads (wavefront kernel)
    void *d_sq_cold = nullptr;                // queued supergrid tracer: adjoint path state kept in global memory (drt_sq.hip)
    void *d_uempty = nullptr;                 // per pixel: its rays cross only empty supergrid cells (build_unit_empty; queued supergrid tracer)
    size_t uempty_bytes = 0;
    void *d_order = nullptr;                  // ray order of the supergrid tracer's current launch (build_super_order)
    size_t order_bytes = 0;
    uint64_t order_first = 0, order_end = 0;  // ... made by the primal launch over these rays of the job the path cache describes:
    uint32_t order_unit = 0;                  //     the adjoint launch over the same rays takes it as it is (0: none)
    void *d_tail = nullptr;                   // tail pool of the cooperative kernels: [counter, pad to 256 B][entries x 128 B]
    size_t tail_entries = 0;
    int n_cus = 256;
    float *d_sigma_b = nullptr;    // bricked copy of sigma_t (refreshed by drt_params_changed)
    uint32_t *d_occ = nullptr;     // empty-space bitmask (kOccWords words)
    float *d_mgrid = nullptr;      // majorant supergrid (refreshed by drt_params_changed)
    float *d_env = nullptr;        // envmap emitter: pixels | marginal CDF | conditional CDFs (one allocation)
    float4 *d_grid4 = nullptr;     // interleaved four-channel apron-brick copy (fused pass), built on demand
    size_t grid4_quads = 0;
    uint64_t grid4_version = 0;    // medium_version the copy was built at (0: never)
    uint64_t medium_version = 0;   // bumped whenever the parameter grids (may) have changed: drt_set_medium / drt_params_changed
    // deferred splatting (drt_deferred.hip): record streams in / tile-sorted, chunk fills, partition tables.
    // Two slots: sub-batch b traces into slot b % 2 while slot (b - 1) % 2 is reduced on the side stream.
    struct RecSlot {
        void *mem = nullptr;           // one allocation, carved up in ensure_deferred
        size_t bytes = 0;
        uint64_t rays = 0;             // ray count the current carving was sized for
        int bins = 0;
        bool tiny = false;
        uint32_t per_ray[2] = {0, 0};  // record capacity per ray (sigma_t, each colour plane) of the current carving
        drt::DeferredPlan plan{};
        size_t clear_bytes = 0;        // chunk fills + cursors: the prefix of mem zeroed before every launch
        hipEvent_t traced = nullptr, reduced = nullptr;
        bool busy = false;             // `reduced` is pending on the side stream
    } rec[2];
    hipStream_t side = nullptr;        // high-priority stream of the overlapped reductions / of the early histogram pass
    // early histogram (drt_deferred.hip: launch_deferred_early_histogram): run_backward names the plan of the launch that
    // follows; the adjoint launcher takes the histogram of the main launch's records on `side` next to the tail launch
    const drt::DeferredPlan *early_plan = nullptr;
    bool early_done = false;
    bool early_partition = false;             // ... and it was the whole partition (histogram .. scatter: the queued tracer's tail launch), not just the histogram
    void *d_sq_tail = nullptr;                // tail pool of the queued tracer's adjoint launches: 256-byte header {count, .., dummy cursors} + entries
    size_t sq_tail_bytes = 0;
    hipEvent_t ev_split = nullptr, ev_hist = nullptr;
    hipStream_t nerf_stream = nullptr;  // the nerf half of the fused pass runs beside the volpathsimple half (drt_fused_render_*)
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    uint32_t *d_nerf_bounds = nullptr; // 32 bytes: max |dL|, |L_in|, |emission|, non-finite flag, largest negative density of a nerf tile adjoint launch (drt_nerf_tile.hip)
    // path cache (drt_coop.hip): written by the primal launch of an H1 step, read by the adjoint launch of the
    // same job if nothing happened to the handle in between
    void *d_pcache = nullptr;          // [rays][kPathCacheCap][2] uint4 | [rays] hash words
    size_t pcache_bytes = 0;
    uint32_t pcache_cap = 0;           // bounce-loop iterations per ray of the cache as the last primal launch laid it out
    struct JobSig { uint64_t n_rays, ray_offset, chunk, stride; uint32_t spp, seed; const void *rays_o, *rays_d; uint64_t scene_version; bool valid; } pcache_sig{};
    bool order_valid = false;          // block_order of the last primal launch is usable
    bool perm_valid = false;           // ray_perm was written by the primal launch the path cache signature describes
    uint64_t order_rays = 0;           // ray count of the launch that produced the stored order (0: none)
    uint64_t scene_version = 0;        // bumped by every call that changes the medium / emitter / sensor / integrator state
    size_t mgrid_cells = 0;
    size_t sigma_b_floats = 0;
    size_t gt_floats = 0;
    bool counting = false;
    uint64_t chunk = 0, stride = 0;   // ray interleave (drt_set_ray_interleave)
    uint32_t debug_flags = 0;
    int occ_z = 0;
    bool timing = false;
    // HIP event pairs around every tracing launch while timing is on: [0] primal, [1] backward
    std::vector<std::pair<hipEvent_t, hipEvent_t>> timed[4];   // primal, adjoint, gradient reduction, whole backward pass
    std::string error;
};

using drt::dbg;

namespace {

thread_local std::string g_error;

int fail(drt_handle h, int code, const char *fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (h) h->error = buf;
    g_error = buf;
    return code;
}

#define DRT_HIP_CHECK(h, expr)                                                              \
    do {                                                                                    \
        hipError_t e_ = (expr);                                                             \
        if (e_ != hipSuccess)                                                               \
            return fail(h, DRT_ERR_HIP, "%s failed: %s", #expr, hipGetErrorString(e_));      \
    } while (0)

struct DeviceGuard {
    int prev = -1;
    bool ok = false;
    explicit DeviceGuard(int dev)
    {
        if (hipGetDevice(&prev) == hipSuccess && hipSetDevice(dev) == hipSuccess) ok = true;
    }
    ~DeviceGuard() { if (prev >= 0) (void) hipSetDevice(prev); }
};

int check_job(drt_handle h, const float *rays_o, const float *rays_d, uint64_t n_rays,
              uint64_t ray_offset, uint32_t spp, bool need_albedo = true)
{
    if (!h) return fail(nullptr, DRT_ERR_INVALID_ARGUMENT, "null handle");
    if (!h->have_medium) return fail(h, DRT_ERR_NOT_CONFIGURED, "no medium: call drt_set_medium first");
    if (!h->have_emitter) return fail(h, DRT_ERR_NOT_CONFIGURED, "no emitter: call drt_set_emitter_constant first");
    if (need_albedo && !h->base.albedo) return fail(h, DRT_ERR_NOT_CONFIGURED, "the medium has no albedo grid");
    if ((rays_o == nullptr) != (rays_d == nullptr))
        return fail(h, DRT_ERR_INVALID_ARGUMENT, "rays_o and rays_d must both be given or both be NULL");
    if (!rays_o && !h->have_sensor)
        return fail(h, DRT_ERR_NOT_CONFIGURED, "no rays and no sensor: call drt_set_sensor_perspective");
    if (spp == 0) return fail(h, DRT_ERR_INVALID_ARGUMENT, "spp must be > 0");
    // wavefront size limit of the reference (batched.py:378-388): ray indices are 32-bit
    uint64_t last = ray_offset + n_rays;   // one past the largest global index
    if (h->chunk && n_rays)
        last = ray_offset + ((n_rays - 1) / h->chunk) * h->stride + ((n_rays - 1) % h->chunk) + 1;
    if (last > 0xffffffffull)
        return fail(h, DRT_ERR_INVALID_ARGUMENT, "global ray index exceeds 2^32 - 1 (render in several passes)");
    if (!rays_o) {
        uint64_t total = (uint64_t) h->base.width * (uint64_t) h->base.height * spp;
        if (last > total)
            return fail(h, DRT_ERR_INVALID_ARGUMENT, "ray range exceeds width*height*spp");
    }
    return DRT_OK;
}

void fill_job(drt_handle h, drt::Params &P, const float *rays_o, const float *rays_d, uint64_t n_rays,
              uint64_t ray_offset, uint32_t spp, uint32_t seed)
{
    P = h->base;
    P.majorant = h->d_majorant;
    P.hide_emitters = h->cfg.hide_emitters; P.use_nee = h->cfg.use_nee; P.use_drt = h->cfg.use_drt;
    P.use_drt_subsampling = h->cfg.use_drt_subsampling; P.use_drt_mis = h->cfg.use_drt_mis;
    P.max_depth = h->cfg.max_depth; P.rr_depth = h->cfg.rr_depth;
    P.sensor_flow = rays_o ? 0 : 1;
    P.rays_o = rays_o; P.rays_d = rays_d;
    P.n_rays = n_rays; P.ray_offset = ray_offset; P.spp = spp; P.seed = seed;
    P.chunk = h->chunk; P.stride = h->stride;
    P.alt_seed = drt::host_alt_seed(seed, rays_o == nullptr);
    P.counters = h->counting ? h->d_counters : nullptr;
    P.debug_flags = h->debug_flags;
    P.order = nullptr; P.order_unit = 1; P.order_units = 0;
    P.unit_empty = nullptr; P.empty_unit = 0;
}

void clear_timings(drt_handle h)
{
    for (auto &v : h->timed) {
        for (auto &p : v) { (void) hipEventDestroy(p.first); (void) hipEventDestroy(p.second); }
        v.clear();
    }
}

// launch bracketed by an event pair on the handle's stream when timing is enabled
#ifndef DRT_PATH_CACHE_CAP
#define DRT_PATH_CACHE_CAP 64
#endif
#ifndef DRT_ORDER_ITERS
#define DRT_ORDER_ITERS 1           // adjoint launches of the supergrid tracer: units ordered by the primal pass's iteration counts too (0: as the primal launch)
#endif
constexpr uint32_t kPathCacheCap = DRT_PATH_CACHE_CAP;  // bounce-loop iterations cached per ray (headline: 2.4 on average)
constexpr uint64_t kHeavyFirstMaxBlocks = 12288;       // launches up to this many 256-ray blocks run heavy blocks first
constexpr uint64_t kPathCacheMaxRays = 1ull << 24;     // larger primal launches (reference renders) skip the cache
constexpr size_t kPathCacheMaxBytes = 36ull << 30;     // the cache never takes more than this (2^24 rays x 64 iterations x 32 B = 34 GB)

bool same_job(const drt_handle_s::JobSig &a, const drt_handle_s::JobSig &b)
{
    return a.n_rays == b.n_rays && a.ray_offset == b.ray_offset && a.chunk == b.chunk && a.stride == b.stride &&
           a.spp == b.spp && a.seed == b.seed && a.rays_o == b.rays_o && a.rays_d == b.rays_d && a.scene_version == b.scene_version;
}

drt_handle_s::JobSig job_sig(drt_handle h, const drt::Params &P)
{
    return drt_handle_s::JobSig{ P.n_rays, P.ray_offset, P.chunk, P.stride, P.spp, P.seed, P.rays_o, P.rays_d, h->scene_version, true };
}

// layout behind the path-cache entries: [rays] hash words | [blocks] cost | [blocks] order | [perm slots] u16 schedule | [perm slots] u8 keys
uint16_t *perm_base(uint32_t *ray_hash, uint64_t n_rays)
{
    const uint64_t n_blocks = (n_rays + 255) / 256;
    return (uint16_t *) (((uintptr_t) (ray_hash + n_rays + 2 * n_blocks) + 7) & ~(uintptr_t) 7);
}

// primal launch of the cooperative kernel: bind the cache for writing (best effort)
void bind_path_cache_write(drt_handle h, drt::Params &P)
{
    h->pcache_sig.valid = false;
    if (P.n_rays != h->order_rays) h->order_rays = 0;           // another launch shape re-carves the buffer: the stored order dies
    if (dbg(h->debug_flags, 1048576u) || P.n_rays > kPathCacheMaxRays) return;
    const size_t n_blocks = (size_t) ((P.n_rays + 255) / 256);
    const size_t perm_slots = ((size_t) P.n_rays + drt::kPermGroup - 1) / drt::kPermGroup * drt::kPermGroup;
    const size_t rest = (size_t) P.n_rays * sizeof(uint32_t) + 2 * n_blocks * sizeof(uint32_t) + perm_slots * 3 + 16;
    auto entry_bytes = [&](uint32_t cap) { return (size_t) P.n_rays * cap * 2 * sizeof(uint4); };
    // Depth of the cache: kPathCacheCap iterations per ray (2 KiB per ray: 17 GB at the headline's 8.4 M rays) where that fits what the buffer holds
    // already or HALF of what the device has free now, at most kPathCacheMaxBytes - the record streams of the backward pass are sized from the free
    // memory too (run_backward), and a handle on a smaller device, or one that shares its device with other ranks, must leave them room; otherwise
    // halved until it fits (from 4 iterations down: no cache - deep main paths then walk again in the adjoint pass, results unchanged).
    uint32_t cap = kPathCacheCap;
    if (entry_bytes(cap) + rest > h->pcache_bytes) {
        size_t free_b = 0, total_b = 0;
        if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) { (void) hipGetLastError(); free_b = 0; }
        size_t budget = h->pcache_bytes + free_b / 2;
        if (budget > kPathCacheMaxBytes) budget = kPathCacheMaxBytes;
        while (cap >= 4u && entry_bytes(cap) + rest > budget) cap /= 2u;
        if (cap < 4u) return;
    }
    const size_t entries = entry_bytes(cap), need = entries + rest;
    if (need > h->pcache_bytes) {
        if (h->d_pcache) { if (hipStreamSynchronize(h->stream) != hipSuccess) return; (void) hipFree(h->d_pcache); h->d_pcache = nullptr; h->pcache_bytes = 0; }
        h->order_rays = 0;
        if (hipMalloc(&h->d_pcache, need) != hipSuccess) { (void) hipGetLastError(); h->d_pcache = nullptr; return; }
        h->pcache_bytes = need;
    }
    h->pcache_cap = cap;
    P.path_cache = (uint4 *) h->d_pcache;
    P.ray_hash = (uint32_t *) ((char *) h->d_pcache + entries);
    P.path_cache_cap = cap; P.path_cache_mode = 1;
    P.block_cost = P.ray_hash + P.n_rays;
    if (hipMemsetAsync(P.block_cost, 0, n_blocks * sizeof(uint32_t), h->stream) != hipSuccess) { (void) hipGetLastError(); P.block_cost = nullptr; }
    if (!dbg(h->debug_flags, 4194304u)) P.ray_iters = (uint8_t *) (perm_base(P.ray_hash, P.n_rays) + perm_slots);   // written by the cooperative primal kernel only
    h->perm_valid = false;
    h->pcache_sig = job_sig(h, P);
}

// adjoint launch: read the cache if it was written by the primal pass of this very job
void bind_path_cache_read(drt_handle h, drt::Params &P, uint64_t job_rays)
{
    drt::Params J = P; J.n_rays = job_rays;
    if (!h->pcache_sig.valid || dbg(h->debug_flags, 1048576u) || !same_job(h->pcache_sig, job_sig(h, J))) return;
    const size_t entries = (size_t) job_rays * h->pcache_cap * 2 * sizeof(uint4);   // (the depth the primal pass of this job wrote)
    P.path_cache = (uint4 *) h->d_pcache;
    P.ray_hash = (uint32_t *) ((char *) h->d_pcache + entries);
    P.path_cache_cap = h->pcache_cap; P.path_cache_mode = 2;
    const bool no_lpt = dbg(h->debug_flags, 16777216u);   // test hook: plain XCD block map
    // (measured: film 184^2 x 32 spp, the per-rank share at 8 GPUs: adjoint 2.22 -> 1.70 ms; 256^2: 3.11 -> 2.85 ms; at
    //  the full 512^2 the XCD-contiguous block map is worth more than the order: 9.48 vs 10.02 ms)
    if (!no_lpt && P.ray_first == 0 && P.n_rays == job_rays && h->order_valid)
        P.block_order = P.ray_hash + job_rays + (job_rays + 255) / 256;
    if (h->perm_valid && !dbg(h->debug_flags, 4194304u)) P.ray_perm = perm_base(P.ray_hash, job_rays);
    // (the iteration counts of the primal pass: the supergrid tracer's adjoint launches order their rays by them)
    if (!dbg(h->debug_flags, 4194304u)) {
        const size_t perm_slots = ((size_t) job_rays + drt::kPermGroup - 1) / drt::kPermGroup * drt::kPermGroup;
        P.ray_iters = (uint8_t *) (perm_base(P.ray_hash, job_rays) + perm_slots);
    }
}

struct EarlyCtx { drt_handle h; const drt::Params *P; };

// between the main and the tail launch of the adjoint tracer: snapshot the chunk cursors, histogram of the complete chunks on
// the side stream (the tail launch keeps few workgroups busy for as long as the job's longest path)
hipError_t early_histogram_between(void *ctx)
{
    EarlyCtx *c = (EarlyCtx *) ctx;
    drt_handle h = c->h;
    hipError_t e = drt::launch_deferred_split(*h->early_plan, h->stream);
    if (e == hipSuccess) e = hipEventRecord(h->ev_split, h->stream);
    if (e == hipSuccess) e = hipStreamWaitEvent(h->side, h->ev_split, 0);
    // (timed like the rest of the reduction: its own event pair on the side stream, summed into the reduction time)
    hipEvent_t ta = nullptr, tb = nullptr;
    if (e == hipSuccess && h->timing) { e = hipEventCreate(&ta); if (e == hipSuccess) e = hipEventCreate(&tb); if (e == hipSuccess) e = hipEventRecord(ta, h->side); }
    if (e == hipSuccess) e = drt::launch_deferred_early_histogram(*c->P, *h->early_plan, h->side);
    if (e == hipSuccess && h->timing) { e = hipEventRecord(tb, h->side); if (e == hipSuccess) h->timed[2].emplace_back(ta, tb); }
    if (e == hipSuccess) e = hipEventRecord(h->ev_hist, h->side);
    return e;
}

int timed_launch(drt_handle h, int which, const drt::Params &P, bool adjoint)
{
    hipEvent_t a = nullptr, b = nullptr;
    if (h->timing) {
        DRT_HIP_CHECK(h, hipEventCreate(&a));
        DRT_HIP_CHECK(h, hipEventCreate(&b));
        DRT_HIP_CHECK(h, hipEventRecord(a, h->stream));
    }
    // Kernel choice - which call reaches which kernel (DESIGN.md section 1 has the table):
    //   global majorant (majorant_resolution_factor 0): CoopTracer (drt_coop.hip: one ray per lane, wave-cooperative tracking rounds), both passes;
    //   majorant supergrid (the reference's default 8): the queued tracer (drt_sq.hip: rays are records, waves take batches of one kind of work),
    //   both passes, every estimator - where its records fit LDS next to the supergrid's majorants or cell bitmask (sq_supported: up to ~99^3 cells,
    //   max_depth <= 1000) and the adjoint's splats travel as records; otherwise (larger supergrids, the atomic gradient path of grids beyond 16384
    //   reduction tiles or without record memory): CoopTracer<SUPER> (drt_coop_super.hip: own-lane tracking steps through the supergrid).
    //   Every primal kernel writes the path cache the adjoint pass of the job reads.  Only the library flavour with test hooks also holds the older
    //   generations - the round-3 supergrid kernel (drt_super.hip; hook 4096), the round-2 state machine of whole flights (drt_wavefront.hip; hooks
    //   32 / 65536 / 134217728) and the plain per-lane Tracer (drt_kernels.hip; hooks 8 / 32768) - where the variant tests keep them in lock-step.
    //   Colour grids on their own lattice (drt_set_colour_resolution): the kernels of drt_own.hip - CoopTracer with either kind of majorant, compiled
    //   with the colour lookups and splats on that lattice; no tail pool, no queued tracer (correct first: the configurations it serves are rare).
    if (P.colour_own) {
        drt::Params Q = P;
        Q.tail_pool = nullptr; Q.tail_count = nullptr; Q.tail_cap = 0; Q.tail_mode = 0;
        if (P.mgrid) Q.ray_perm = nullptr;
        DRT_HIP_CHECK(h, drt::launch_trace_own(Q, adjoint, h->counting, h->stream));
        if (h->timing) {
            DRT_HIP_CHECK(h, hipEventRecord(b, h->stream));
            h->timed[which].emplace_back(a, b);
        }
        return DRT_OK;
    }
    const bool quadratic = h->cfg.use_drt && !h->cfg.use_drt_subsampling;
    // (the records' global halves, ~44 MB, are allocated only by a launch that will run the queued kernel: not when a test hook or the atomic
    //  gradient path routes this launch to the older kernels)
    const bool super_path = P.mgrid && !dbg(h->debug_flags, (134217728u | 8u | 32u | 32768u | 65536u)) && (!adjoint || P.rec_buf[0] != nullptr);
    bool sq_ok = super_path && !dbg(h->debug_flags, 4096u) && drt::sq_supported(P);
    if (sq_ok && !h->d_sq_cold && hipMalloc(&h->d_sq_cold, drt::sq_cold_bytes(h->n_cus)) != hipSuccess) {
        (void) hipGetLastError(); h->d_sq_cold = nullptr; sq_ok = false;
    }
    // Production: the queued tracer or - supergrids it does not take, the atomic gradient path - CoopTracer<SUPER> below.  The flavour with test
    // hooks also keeps the round-3 kernel (drt_super.hip; hook 4096) and the round-2 state machine (drt_wavefront.hip; hooks 32 / 65536 / 134217728).
#ifdef DRT_TEST_HOOKS
    const bool super3 = super_path && !sq_ok && !quadratic && dbg(h->debug_flags, 4096u) && drt::super_supported(P);
#else
    const bool super3 = false;
#endif
    const bool super = (super_path && sq_ok) || super3;
    if (super) {
        drt::Params Q = P;
        Q.queues = h->d_queues;
        const uint64_t span = P.n_rays - P.ray_first;
        // round 4: the queued tracer (drt_sq.hip) where the ray records fit LDS next to the majorants; test hook 4096 keeps
        // the round-3 kernel (drt_super.hip), which also serves what the queued one does not take
        const bool queued = sq_ok;
#ifndef DRT_SQ_UNIT_EMPTY
#define DRT_SQ_UNIT_EMPTY 1         // pixels whose rays cross only empty supergrid cells are flagged: their primary-segment flights are not walked
#endif
        // (sensor rays whose units are whole pixels: sub-batches, interleaved chunks and offsets that cut a pixel's rays get no flags)
        if (DRT_SQ_UNIT_EMPTY && queued && P.sensor_flow && P.mocc && P.spp && span < (1ull << 31) && P.ray_first % P.spp == 0 &&
            P.ray_offset % P.spp == 0 && P.chunk % P.spp == 0 && P.stride % P.spp == 0 && !dbg(h->debug_flags, 2147483648u)) {
            const uint32_t eunits = (uint32_t) ((span + P.spp - 1) / P.spp);
            if (eunits > h->uempty_bytes) {
                if (h->d_uempty) { DRT_HIP_CHECK(h, hipStreamSynchronize(h->stream)); (void) hipFree(h->d_uempty); h->d_uempty = nullptr; h->uempty_bytes = 0; }
                if (hipMalloc(&h->d_uempty, eunits) == hipSuccess) h->uempty_bytes = eunits; else { (void) hipGetLastError(); h->d_uempty = nullptr; }
            }
            if (h->d_uempty) {
                DRT_HIP_CHECK(h, drt::build_unit_empty(P, P.spp, eunits, (uint8_t *) h->d_uempty, h->stream));
                Q.unit_empty = (const uint8_t *) h->d_uempty; Q.empty_unit = P.spp;
            }
        }
        // Ray order: the longest paths first - units of one pixel's rays by the majorant optical depth along the pixel's ray
        // (drt_super.hip).  Best effort (no memory: index order); test hook 536870912: index order.
        const uint32_t unit = P.spp >= 4u ? P.spp : 16u;
        // (units of more rays than a CU traces at a time - the optimisation loop's primal launches, 1024 rays per pixel -
        //  are too coarse to be scheduled: measured 3-5 % slower in that order than in index order)
        // (and launches of fewer than 1.5 M rays - six rounds of the chip's 196 608 lanes - gain nothing: 512^2 x 4 spp 1.53 -> 1.69 ms,
        //  x 8 spp 1.89 -> 1.79 ms, x 32 spp 4.17 -> 3.75 ms; test hook 1073741824 orders launches from 4096 rays on)
        //  round 4, queued tracer: a rank's share of the headline at 8 GPUs, 512^2 x 32 spp / 8 = 1 M rays: step 3.83 -> 3.68 ms in that order)
        const uint64_t min_span = dbg(h->debug_flags, 1073741824u) ? 4096u : (1u << 20);
        if (span >= min_span && span < (1ull << 31) && unit <= 256u && !dbg(h->debug_flags, 536870912u)) {
            const uint32_t units = (uint32_t) ((span + unit - 1) / unit);
            const size_t need = drt::super_order_bytes(units);
            if (need > h->order_bytes) {
                if (h->d_order) { DRT_HIP_CHECK(h, hipStreamSynchronize(h->stream)); (void) hipFree(h->d_order); h->d_order = nullptr; h->order_bytes = 0; h->order_unit = 0; }
                if (hipMalloc(&h->d_order, need) == hipSuccess) h->order_bytes = need; else { (void) hipGetLastError(); h->d_order = nullptr; }
            }
            if (h->d_order) {
                // (adjoint launches behind the primal pass of the same job: the primal pass counted every ray's bounce-loop iterations -
                //  units with a long main path start first, whatever the optical depth along their pixel's ray says)
                const uint8_t *iters = (DRT_ORDER_ITERS && adjoint && P.path_cache_mode == 2 && sq_ok) ? P.ray_iters : nullptr;
                const bool reuse = !iters && adjoint && P.path_cache_mode == 2 && h->order_unit == unit && h->order_first == P.ray_first && h->order_end == P.n_rays;
                if (!reuse) DRT_HIP_CHECK(h, drt::build_super_order(P, unit, units, h->d_order, h->stream, iters));
                h->order_unit = (!adjoint && P.path_cache_mode == 1) || reuse ? unit : 0u;
                h->order_first = P.ray_first; h->order_end = P.n_rays;
                Q.order = (const uint32_t *) h->d_order; Q.order_unit = unit; Q.order_units = units;
            }
        }
        Q.ray_perm = nullptr; Q.block_order = nullptr;
        // thin medium (as far as the host has seen its majorant - drt_params_changed - without waiting for anything)?  Primal launches in index order then
        // run the ROUNDS kernels (drt_sq.hip): an optimisation's first iterations, where nearly every ray is over before it begins
        if (h->majorant_pending && hipEventQuery(h->ev_majorant) == hipSuccess) { h->majorant_seen = *h->h_majorant; h->majorant_pending = false; }
        else if (h->majorant_pending) (void) hipGetLastError();                        // (hipErrorNotReady is not an error)
        {
            const float dx = P.bmax[0] - P.bmin[0], dy = P.bmax[1] - P.bmin[1], dz = P.bmax[2] - P.bmin[2];
            Q.sq_rounds = (!adjoint && h->majorant_seen >= 0.0f && h->majorant_seen * std::sqrt(dx * dx + dy * dy + dz * dz) < 3.0f) ? 1u : 0u;
        }
        DRT_HIP_CHECK(h, hipMemsetAsync(h->d_queues, 0, 8 * sizeof(unsigned long long), h->stream));
        if (queued) {
            Q.sq_cold = h->d_sq_cold;
#ifndef DRT_SQ_TAIL
#define DRT_SQ_TAIL 1               // adjoint launches of the queued tracer: drained workgroups hand their last records to a tail pool; 0: they finish them
#endif
            // Tail pool (adjoint, deferred splats, the reduction of this very launch follows on h->stream): the last paths of a launch are latency - a
            // few records per CU, 0.45 ms of the headline's adjoint launch.  Drained workgroups write them to the pool and end; the partition
            // passes of the gradient reduction (histogram, offsets, scan, scatter: they do not touch the gradient grids) run on a side stream
            // BESIDE the tail launch, which finishes the pooled records with its splats as direct atomics; tile_reduce follows both.
            // Below 2 M rays this overlapped kind does not pay (a rank's share of the headline at 8 GPUs, 1 M rays: 3.19 ms per step without the pool,
            // 3.31 with it - such a launch IS its longest path, a second launch only adds its own start; at 4 GPUs, 2 M rays: 4.00 / 4.07 ms): those
            // launches take the SOLO kind below.
            // (test hooks: 268435456 no tail pool, 1073741824 launches from 4096 rays on have one)
            const uint64_t tail_min = dbg(h->debug_flags, 1073741824u) ? 4096u : (1u << 21);
            const bool big = adjoint && h->early_plan && Q.rec_buf[0] && span >= tail_min;
            // Round 5, SOLO tails: where the majorants fit LDS the tail launch runs its records to their ends in registers, without queue hops
            // (drt_sq.hip: SOLO) - a lone ray's bounce then costs its lookups and one wave's instructions, not eight hand-overs.  That pays
            // without anything running beside it: the primal launch and the small adjoint launches (a rank's share at 8 GPUs) hand their last
            // records to a tail launch over the whole chip.  (test hook 268435456: no tail pool of either kind)
#ifndef DRT_SQ_TAIL_SOLO
#define DRT_SQ_TAIL_SOLO 3          // bit 0: primal launches, bit 1: adjoint launches below the size of the overlapped tail
#endif
            const bool solo = drt::sq_tail_solo(Q) && span >= 8192u && !big &&
                              (adjoint ? ((DRT_SQ_TAIL_SOLO & 2) != 0 && Q.rec_buf[0] != nullptr) : (DRT_SQ_TAIL_SOLO & 1) != 0);
#ifndef DRT_SQ_TAIL_THIN
#define DRT_SQ_TAIL_THIN 0          // 0: primal launches over a thin medium (the ROUNDS kernels) make no tail launch - their few real paths are short, a second
                                    // launch over the whole chip costs more than they do (config3_as_reproduce 302-305 -> 308-310 iterations/s); 1: they do
#endif
            const bool tail = DRT_SQ_TAIL && (big || solo) && !dbg(h->debug_flags, 268435456u) && (DRT_SQ_TAIL_THIN || !Q.sq_rounds);
            if (tail) {
                const size_t cap = (size_t) h->n_cus * drt::sq_tail_push();
                const size_t need_b = 256 + cap * drt::sq_tail_entry_quads() * sizeof(uint4);
                if (need_b > h->sq_tail_bytes) {
                    if (h->d_sq_tail) { DRT_HIP_CHECK(h, hipStreamSynchronize(h->stream)); (void) hipFree(h->d_sq_tail); h->d_sq_tail = nullptr; h->sq_tail_bytes = 0; }
                    if (hipMalloc(&h->d_sq_tail, need_b) == hipSuccess) h->sq_tail_bytes = need_b; else { (void) hipGetLastError(); h->d_sq_tail = nullptr; }
                }
                if (h->d_sq_tail) {
                    DRT_HIP_CHECK(h, hipMemsetAsync(h->d_sq_tail, 0, 256, h->stream));
                    Q.tail_count = (uint32_t *) h->d_sq_tail; Q.tail_pool = (uint4 *) ((char *) h->d_sq_tail + 256); Q.tail_cap = (uint32_t) cap; Q.tail_mode = 0;
                }
            }
            DRT_HIP_CHECK(h, drt::launch_trace_sq(Q, adjoint, h->counting, h->n_cus, h->stream));
            if (tail && Q.tail_pool) {
                hipEvent_t ta = nullptr, tb = nullptr;
                if (big) {
                    if (!h->side) {
                        int lo = 0, hi = 0;
                        DRT_HIP_CHECK(h, hipDeviceGetStreamPriorityRange(&lo, &hi));
                        DRT_HIP_CHECK(h, hipStreamCreateWithPriority(&h->side, hipStreamNonBlocking, hi));
                    }
                    if (!h->ev_split) {
                        DRT_HIP_CHECK(h, hipEventCreateWithFlags(&h->ev_split, hipEventDisableTiming));
                        DRT_HIP_CHECK(h, hipEventCreateWithFlags(&h->ev_hist, hipEventDisableTiming));
                    }
                    // the main launch's records are complete (the tail launch emits none): partition them on the side stream ...
                    DRT_HIP_CHECK(h, hipEventRecord(h->ev_split, h->stream));
                    DRT_HIP_CHECK(h, hipStreamWaitEvent(h->side, h->ev_split, 0));
                    if (h->timing) { DRT_HIP_CHECK(h, hipEventCreate(&ta)); DRT_HIP_CHECK(h, hipEventCreate(&tb)); DRT_HIP_CHECK(h, hipEventRecord(ta, h->side)); }
                    DRT_HIP_CHECK(h, drt::launch_deferred_reduce(Q, *h->early_plan, h->side, nullptr, false, 1));
                    if (h->timing) { DRT_HIP_CHECK(h, hipEventRecord(tb, h->side)); h->timed[2].emplace_back(ta, tb); }
                    DRT_HIP_CHECK(h, hipEventRecord(h->ev_hist, h->side));
                }
                // ... beside the tail launch: the pool's records to their ends, splats as direct atomics into the caller's grids (no chunk is
                // handed out: the cursors it touches are dummies in the pool's header - the partition of the main launch's records is under way)
                drt::Params T = Q;
                T.tail_mode = big ? 1 : 2;                                 // (2: nothing runs beside it - over the whole chip)
                if (big) {                                                 // (a SOLO tail of a small launch appends to the record streams: their reduction follows it)
                    T.rec_cursor = (uint32_t *) h->d_sq_tail + 8;
                    T.rec_cap_chunks[0] = T.rec_cap_chunks[1] = 0;
                }
                T.order = nullptr; T.unit_empty = nullptr;
                DRT_HIP_CHECK(h, drt::launch_trace_sq(T, adjoint, h->counting, h->n_cus, h->stream));
                if (big) { h->early_done = true; h->early_partition = true; }
            }
        }
#ifdef DRT_TEST_HOOKS
        else DRT_HIP_CHECK(h, drt::launch_trace_super(Q, adjoint, h->counting, h->n_cus, h->stream));
#endif
        if (h->timing) {
            DRT_HIP_CHECK(h, hipEventRecord(b, h->stream));
            h->timed[which].emplace_back(a, b);
        }
        return DRT_OK;
    }
    // (hooks only: 134217728 = supergrid scenes in the round-2 kernels - the state machine of whole flights for the primal pass, 65536 = that
    //  kernel for any primal pass, 32 = for the adjoint too)
    const bool sm_primal = drt::kTestHooks && !adjoint && ((P.mgrid != nullptr && dbg(h->debug_flags, 134217728u)) || dbg(h->debug_flags, 65536u)) && !dbg(h->debug_flags, 8u);
    const bool sm_adjoint = drt::kTestHooks && adjoint && dbg(h->debug_flags, 32u) && !quadratic && !dbg(h->debug_flags, 8u);
    const bool wavefront = sm_primal || sm_adjoint;
    const bool coop = !wavefront && !dbg(h->debug_flags, (adjoint ? 32768u : 8u));
    if (coop) {
        // tail pool: room for 1/8 of the launch's rays (a workgroup sends at most DRT_TAIL_PUSH = 24 of its 256; a full pool only means that the rest stays where it is), kept while it is big enough;
        // without it (allocation failed) the kernels simply finish every path where it is
        drt::Params PT = P;
        PT.tail_pool = nullptr; PT.tail_count = nullptr; PT.tail_cap = 0; PT.tail_mode = 0;
        // (launches of fewer than 1.5 M rays finish every path where it is: the tail launch costs them the job's longest
        //  path once more - 512^2 x 4 spp, an eighth of the headline: adjoint 1.81 -> 1.65 ms without it, x 8 spp 2.22 / 2.20,
        //  x 32 spp 5.78 / 6.88; test hook 1073741824: small launches are scheduled like large ones)
        const bool tail_pays = P.n_rays - P.ray_first >= (3u << 19) || dbg(h->debug_flags, 1073741824u);
        if (adjoint && !P.mgrid && P.n_rays > P.ray_first && tail_pays) {
            const size_t want = (((size_t) (P.n_rays - P.ray_first) / 8 + 255) / 256) * 256;
            if (want > h->tail_entries) {
                if (h->d_tail) { DRT_HIP_CHECK(h, hipStreamSynchronize(h->stream)); (void) hipFree(h->d_tail); h->d_tail = nullptr; h->tail_entries = 0; }
                if (hipMalloc(&h->d_tail, 256 + want * 128) == hipSuccess) h->tail_entries = want; else { (void) hipGetLastError(); h->d_tail = nullptr; }
            }
            if (h->d_tail && want >= 256) {
                PT.tail_count = (uint32_t *) h->d_tail; PT.tail_pool = (uint4 *) ((char *) h->d_tail + 256); PT.tail_cap = (uint32_t) want;
            }
        }
        if (P.mgrid) {
            // supergrid: rays stay in image order - neighbouring pixels walk the same supergrid cells, and sorting them
            // by path length costs more than it saves (24.6 vs 18.8 ms at majorant_resolution_factor 8)
            drt::Params Q = PT;
            Q.ray_perm = nullptr;
            DRT_HIP_CHECK(h, drt::launch_trace_coop(Q, adjoint, h->counting, h->stream));
        } else {
            // (test hook 268435456: no early histogram pass)
            const bool early = adjoint && h->early_plan && PT.rec_buf[0] && PT.tail_pool && !dbg(h->debug_flags, 268435456u);
            if (early && !h->side) {
                int lo = 0, hi = 0;
                DRT_HIP_CHECK(h, hipDeviceGetStreamPriorityRange(&lo, &hi));
                DRT_HIP_CHECK(h, hipStreamCreateWithPriority(&h->side, hipStreamNonBlocking, hi));
            }
            if (early && !h->ev_split) {
                DRT_HIP_CHECK(h, hipEventCreateWithFlags(&h->ev_split, hipEventDisableTiming));
                DRT_HIP_CHECK(h, hipEventCreateWithFlags(&h->ev_hist, hipEventDisableTiming));
            }
            EarlyCtx ctx{ h, &PT };
            bool called = false;
            DRT_HIP_CHECK(h, drt::launch_trace_coop(PT, adjoint, h->counting, h->stream, early ? early_histogram_between : nullptr, &ctx, &called));
            h->early_done = called;
        }
    }
#ifdef DRT_TEST_HOOKS
    else if (!wavefront) {
        drt::Params Q = P;
        Q.ray_perm = nullptr;
        DRT_HIP_CHECK(h, drt::launch_trace(Q, adjoint, h->counting, h->stream));
    }
    else {
        drt::Params Q = P;
        Q.queues = h->d_queues;
        DRT_HIP_CHECK(h, hipMemsetAsync(h->d_queues, 0, 8 * sizeof(unsigned long long), h->stream));
        DRT_HIP_CHECK(h, drt::launch_trace_wavefront(Q, adjoint, h->counting, h->n_cus, h->stream));
    }
#endif
    if (h->timing) {
        DRT_HIP_CHECK(h, hipEventRecord(b, h->stream));
        h->timed[which].emplace_back(a, b);
    }
    return DRT_OK;
}

// Deferred splatting is used for the one-ray-per-lane adjoint kernel when the grid has at most kMaxBins
// tiles and the record streams fit the memory budget; otherwise (and with debug bit 128) the tracer
// adds its splats to the apron scratch with atomics and untile_gradients_kernel reduces that.
constexpr int kNoRecordMemory = -1000;                           // internal: record streams could not be allocated
constexpr uint64_t kRecBudgetBytes = 48ull << 30;               // record streams (emitted + tile-sorted) per sub-batch

bool want_deferred(drt_handle h, const drt::Params &P)
{
    if (dbg(h->debug_flags, (128u | 2u | 32u))) return false;          // 128: atomic path; 2: per-lane atomics; 32: state machine
    const int ntx = (P.rx + drt::kTileX - 1) / drt::kTileX, nty = (P.ry + drt::kTileY - 1) / drt::kTileY,
              ntz = (P.rz + drt::kTileZ - 1) / drt::kTileZ;
    return (int64_t) ntx * nty * ntz <= drt::kMaxBins;
}

int ensure_deferred(drt_handle h, drt_handle_s::RecSlot &R, drt::Params &P, uint64_t n_rays, uint32_t per_ray_sigma,
                    uint32_t per_ray_colour, bool simulate_no_memory = false)
{
    using namespace drt;
    if (simulate_no_memory) return kNoRecordMemory;
    DeferredPlan &D = R.plan;
    const int ntx = (P.rx + kTileX - 1) / kTileX, nty = (P.ry + kTileY - 1) / kTileY, ntz = (P.rz + kTileZ - 1) / kTileZ;
    const int n_bins = ntx * nty * ntz;
    const bool tiny = dbg(h->debug_flags, 256u) != 0;                // test hook: force the out-of-chunks path
    if (!R.mem || n_rays > R.rays || n_bins != R.bins || tiny != R.tiny || per_ray_sigma > R.per_ray[0] ||
        per_ray_colour > R.per_ray[1]) {
        // capacity: every wave may leave one chunk group partly filled per stream, plus the expected volume;
        // beyond it the tracer falls back to direct atomics (emit_record): a performance choice only
        const uint64_t waves = (n_rays + 63) / 64;
        const uint32_t per_ray[2] = { per_ray_sigma, per_ray_colour };
        uint64_t chunks[2];
        for (int s = 0; s < 2; ++s)
            chunks[s] = tiny ? 2 * rec_group(s) : (rec_group(s) + 1) * waves + (n_rays * per_ray[s] + kRecChunk - 1) / kRecChunk;
        const size_t quads[2] = { 1, 2 };                          // float4s per record
        size_t off = 0;
        auto carve = [&](size_t bytes) { size_t o = off; off += (bytes + 255) & ~(size_t) 255; return o; };
        size_t o_cursor = carve(32 * sizeof(uint32_t));          // cursors [8] + vmax [5] (+3) + debug checksums [5 x u64]
        size_t o_count[2], o_in[2], o_out[2];
        for (int s = 0; s < 2; ++s) o_count[s] = carve(chunks[s] * sizeof(uint32_t));
        const size_t clear = off;
        size_t o_hist = carve((size_t) 2 * kPartWGs * n_bins * sizeof(uint32_t));
        size_t o_base = carve((size_t) 2 * (n_bins + 1) * sizeof(uint32_t));
        size_t o_unit = carve((size_t) 2 * (n_bins + 1) * sizeof(uint32_t));
        for (int s = 0; s < 2; ++s) {
            o_in[s] = carve(chunks[s] * kRecChunk * quads[s] * sizeof(float4));
            o_out[s] = carve(chunks[s] * kRecChunk * quads[s] * sizeof(float4));
        }
        if (off > R.bytes) {
            if (R.mem) {
                DRT_HIP_CHECK(h, hipStreamSynchronize(h->stream));
                if (h->side) DRT_HIP_CHECK(h, hipStreamSynchronize(h->side));
                (void) hipFree(R.mem); R.mem = nullptr; R.bytes = 0;
            }
            if (dbg(h->debug_flags, 262144u) || hipMalloc(&R.mem, off) != hipSuccess) {   // not enough memory for the record streams (bit 262144: simulate):
                (void) hipGetLastError();                         // this job uses the atomic path instead
                R.mem = nullptr; R.rays = 0;
                return kNoRecordMemory;
            }
            R.bytes = off;
        }
        char *b = (char *) R.mem;
        D.cursor = (uint32_t *) (b + o_cursor);
        D.vmax = D.cursor + 8;
        uint64_t max_chunks = 0;
        for (int s = 0; s < 2; ++s) {
            D.chunk_count[s] = (uint32_t *) (b + o_count[s]);
            D.in[s] = (float4 *) (b + o_in[s]); D.out[s] = (float4 *) (b + o_out[s]);
            D.cap_chunks[s] = (uint32_t) chunks[s];
            if (chunks[s] > max_chunks) max_chunks = chunks[s];
        }
        D.hist = (uint32_t *) (b + o_hist); D.bin_base = (uint32_t *) (b + o_base); D.unit_start = (uint32_t *) (b + o_unit);
        D.n_bins = n_bins; D.ntx = ntx; D.nty = nty; D.ntz = ntz;
        D.max_units = (uint32_t) (max_chunks * kRecChunk / kUnitRecords + (uint64_t) n_bins + 1);
        R.clear_bytes = clear; R.rays = n_rays; R.bins = n_bins; R.tiny = tiny;
        R.per_ray[0] = per_ray_sigma; R.per_ray[1] = per_ray_colour;
    }
    // the tile grid of THIS medium: a slot sized for another grid with the same NUMBER of tiles keeps its buffers, not that grid's tile arrangement
    // (7 x 25 x 25 voxels: 1 x 2 x 1 tiles, then 25 x 3 x 5: 1 x 1 x 2 - the records went to the wrong tiles and part of the gradient was lost;
    //  found by tests/test_gpu_fuzz.py::test_random_sequence_on_one_handle_matches_the_oracle, round 6)
    D.ntx = ntx; D.nty = nty; D.ntz = ntz;
    DRT_HIP_CHECK(h, hipMemsetAsync(R.mem, 0, R.clear_bytes, h->stream));
    for (int s = 0; s < 2; ++s) { P.rec_buf[s] = D.in[s]; P.rec_chunk_count[s] = D.chunk_count[s]; P.rec_cap_chunks[s] = D.cap_chunks[s]; }
    P.rec_cursor = D.cursor;
    return DRT_OK;
}

int timed_reduce(drt_handle h, const drt::Params &P, const drt::DeferredPlan &D, hipStream_t stream, bool early_hist = false, int phase = 0);
int timed_untile(drt_handle h, const drt::Params &P);

// Adjoint launch + gradient reduction of one job.  Deferred path: the job is cut into sub-batches of rays
// whose record streams fit the memory budget.  With DRT_PIPELINE set, large jobs are cut into
// >= kPipeBatches pieces and pipelined over two record slots (tracer of sub-batch b on the caller's stream,
// partition + reduction of sub-batch b - 1 on a side stream; the caller's stream waits for every
// reduction at the end) - an experiment that measured slower than running them back to back.
constexpr uint64_t kPipeBatches = 4, kPipeMinRays = 1ull << 20;

template <class Launch>
int run_backward(drt_handle h, drt::Params &P, uint32_t per_ray_sigma, uint32_t per_ray_colour, Launch launch)
{
    hipEvent_t t0 = nullptr, t1 = nullptr;
    if (h->timing) {
        DRT_HIP_CHECK(h, hipEventCreate(&t0));
        DRT_HIP_CHECK(h, hipEventCreate(&t1));
        DRT_HIP_CHECK(h, hipEventRecord(t0, h->stream));
    }
    if (!want_deferred(h, P)) {
        int rc = launch(P);
        if (rc) return rc;
        rc = timed_untile(h, P);
        if (rc) return rc;
    } else {
        const uint64_t n_rays = P.n_rays;
        const uint64_t bytes_per_ray = 32ull * (uint64_t) per_ray_sigma + 64ull * (uint64_t) per_ray_colour + 1024ull;   // streams in + sorted, chunk slack
        // measured on the headline workload: overlapping costs more than it hides (tracer 15.0 -> 19.8 ms
        // with the reductions alongside, step 22.1 -> 24.7 ms), so the overlap is opt-in
        const bool forced = dbg(h->debug_flags, 2048u);         // test hook: overlap the reductions with the next sub-batch's tracer
        const bool want_pipe = forced;
        const uint64_t want_pipe_slots = want_pipe ? 2 : 1;
        uint64_t budget = dbg(h->debug_flags, 16384u) ? (8ull << 20) : kRecBudgetBytes;   // test hook: 8 MB -> many sub-batches
        {   // never ask for more than the device can give next to the caller's (torch's) allocations: what the slots
            // hold already plus 80 % of what is free now; smaller budgets only mean more ray sub-batches
            size_t free_b = 0, total_b = 0;
            if (hipMemGetInfo(&free_b, &total_b) == hipSuccess) {
                const uint64_t have = (uint64_t) h->rec[0].bytes + (uint64_t) h->rec[1].bytes;
                const uint64_t room = have + (uint64_t) ((double) free_b * 0.8);
                if (room < budget) budget = room;
            } else (void) hipGetLastError();
            const uint64_t floor_b = 256ull * bytes_per_ray * want_pipe_slots;     // at least one workgroup of rays per slot
            if (budget < floor_b) budget = floor_b;
        }
        uint64_t batch = (budget / want_pipe_slots) / bytes_per_ray;
        const bool pipe = want_pipe && (forced || n_rays >= kPipeMinRays);
        if (pipe && batch > (n_rays + kPipeBatches - 1) / kPipeBatches) batch = (n_rays + kPipeBatches - 1) / kPipeBatches;
        // whole ray-schedule groups (kPermGroup rays = 4 workgroups) and XCD runs per sub-batch
        batch = dbg(h->debug_flags, 16384u) ? (batch + drt::kPermGroup - 1) / drt::kPermGroup * drt::kPermGroup : (batch + 65535) / 65536 * 65536;
        const bool overlap = pipe;
        if (overlap && !h->side) {
            int lo = 0, hi = 0;
            DRT_HIP_CHECK(h, hipDeviceGetStreamPriorityRange(&lo, &hi));
            DRT_HIP_CHECK(h, hipStreamCreateWithPriority(&h->side, hipStreamNonBlocking, hi));
        }
        int b = 0;
        for (uint64_t first = 0; first < n_rays; first += batch, ++b) {
            auto &R = h->rec[overlap ? (b & 1) : 0];
            const uint64_t count = n_rays - first < batch ? n_rays - first : batch;
            if (R.busy) { DRT_HIP_CHECK(h, hipStreamWaitEvent(h->stream, R.reduced, 0)); R.busy = false; }
            int rc = ensure_deferred(h, R, P, count, per_ray_sigma, per_ray_colour,
                                     dbg(h->debug_flags, 524288u) && first > 0);   // test hook: memory runs out after the first sub-batch
            if (rc == kNoRecordMemory) {
                // no memory for the record streams (now): the rays that are left, [first, n_rays), take the
                // atomic path (splats into the apron scratch + untile) - slower, same gradients
                for (int s2 = 0; s2 < 2; ++s2) P.rec_buf[s2] = nullptr;
                P.rec_cursor = nullptr; P.ray_first = first; P.n_rays = n_rays;
                rc = launch(P);
                if (rc) return rc;
                rc = timed_untile(h, P);
                if (rc) return rc;
                break;
            }
            if (rc) return rc;
            P.ray_first = first; P.n_rays = first + count;
            h->early_plan = overlap ? nullptr : &R.plan; h->early_done = false; h->early_partition = false;
            rc = launch(P);
            if (rc) { h->early_plan = nullptr; return rc; }
            if (!overlap) {
                const bool early = h->early_done, part = h->early_partition;
                h->early_plan = nullptr; h->early_done = false; h->early_partition = false;
                if (early) DRT_HIP_CHECK(h, hipStreamWaitEvent(h->stream, h->ev_hist, 0));
                rc = timed_reduce(h, P, R.plan, h->stream, early && !part, part ? 2 : 0);
                if (rc) return rc;
                continue;
            }
            if (!R.traced) { DRT_HIP_CHECK(h, hipEventCreateWithFlags(&R.traced, hipEventDisableTiming)); DRT_HIP_CHECK(h, hipEventCreateWithFlags(&R.reduced, hipEventDisableTiming)); }
            DRT_HIP_CHECK(h, hipEventRecord(R.traced, h->stream));
            DRT_HIP_CHECK(h, hipStreamWaitEvent(h->side, R.traced, 0));
            rc = timed_reduce(h, P, R.plan, h->side);
            if (rc) return rc;
            DRT_HIP_CHECK(h, hipEventRecord(R.reduced, h->side));
            R.busy = true;
        }
        for (auto &R : h->rec)
            if (R.busy) { DRT_HIP_CHECK(h, hipStreamWaitEvent(h->stream, R.reduced, 0)); R.busy = false; }
        P.ray_first = 0; P.n_rays = n_rays;
    }
    if (h->timing) {
        DRT_HIP_CHECK(h, hipEventRecord(t1, h->stream));
        h->timed[3].emplace_back(t0, t1);
    }
    return DRT_OK;
}

int timed_reduce(drt_handle h, const drt::Params &P, const drt::DeferredPlan &D, hipStream_t stream, bool early_hist, int phase)
{
    hipEvent_t a = nullptr, b = nullptr;
    if (h->timing) {
        DRT_HIP_CHECK(h, hipEventCreate(&a));
        DRT_HIP_CHECK(h, hipEventCreate(&b));
        DRT_HIP_CHECK(h, hipEventRecord(a, stream));
    }
    DRT_HIP_CHECK(h, drt::launch_deferred_reduce(P, D, stream, nullptr, early_hist, phase));
    if (h->timing) {
        DRT_HIP_CHECK(h, hipEventRecord(b, stream));
        h->timed[2].emplace_back(a, b);
    }
    return DRT_OK;
}

int timed_nerf(drt_handle h, int which, const drt::Params &P, bool adjoint, hipStream_t st = nullptr)
{
    if (!st) st = h->stream;
    hipEvent_t a = nullptr, b = nullptr;
    if (h->timing) {
        DRT_HIP_CHECK(h, hipEventCreate(&a));
        DRT_HIP_CHECK(h, hipEventCreate(&b));
        DRT_HIP_CHECK(h, hipEventRecord(a, st));
    }
    DRT_HIP_CHECK(h, P.colour_own ? drt::launch_nerf_own(P, adjoint, h->counting, st) : drt::launch_nerf(P, adjoint, h->counting, st));
    if (h->timing) {
        DRT_HIP_CHECK(h, hipEventRecord(b, st));
        h->timed[which].emplace_back(a, b);
    }
    return DRT_OK;
}

int timed_untile(drt_handle h, const drt::Params &P)
{
    hipEvent_t a = nullptr, b = nullptr;
    if (h->timing) {
        DRT_HIP_CHECK(h, hipEventCreate(&a));
        DRT_HIP_CHECK(h, hipEventCreate(&b));
        DRT_HIP_CHECK(h, hipEventRecord(a, h->stream));
    }
    DRT_HIP_CHECK(h, drt::launch_untile(P, h->stream));
    if (h->timing) {
        DRT_HIP_CHECK(h, hipEventRecord(b, h->stream));
        h->timed[2].emplace_back(a, b);
    }
    return DRT_OK;
}

}  // namespace

extern "C" {

const char *drt_version(void) { return drt::kTestHooks ? "drt-hip 0.2 (gfx950, test hooks)" : "drt-hip 0.2 (gfx950)"; }

const char *drt_last_error(drt_handle h)
{
    if (h) return h->error.c_str();
    return g_error.c_str();
}

int drt_create(const drt_config *cfg, int device, drt_handle *out)
{
    if (!cfg || !out) return fail(nullptr, DRT_ERR_INVALID_ARGUMENT, "drt_create: null argument");
    *out = nullptr;
    if (cfg->max_depth < 0) return fail(nullptr, DRT_ERR_INVALID_ARGUMENT, "max_depth must be >= 0");
    int n_dev = 0;
    if (hipGetDeviceCount(&n_dev) != hipSuccess || n_dev <= 0)
        return fail(nullptr, DRT_ERR_NO_DEVICE, "no HIP device available (this library has no CPU fallback)");
    if (device < 0 || device >= n_dev)
        return fail(nullptr, DRT_ERR_INVALID_ARGUMENT, "device %d out of range [0,%d)", device, n_dev);
    drt_handle h = new (std::nothrow) drt_handle_s();
    if (!h) return fail(nullptr, DRT_ERR_HIP, "out of host memory");
    h->device = device;
    h->cfg = *cfg;
    DeviceGuard g(device);
    if (!g.ok) { delete h; return fail(nullptr, DRT_ERR_HIP, "hipSetDevice(%d) failed", device); }
    hipError_t e = hipMalloc(&h->d_majorant, 2 * sizeof(float));
    if (e == hipSuccess) e = hipMalloc(&h->d_scratch, sizeof(uint32_t));
    if (e == hipSuccess) e = hipMalloc(&h->d_queues, 8 * sizeof(unsigned long long));
    if (e == hipSuccess) { hipDeviceProp_t prop; e = hipGetDeviceProperties(&prop, device); if (e == hipSuccess) h->n_cus = prop.multiProcessorCount; }
    if (e == hipSuccess) e = hipMalloc(&h->d_occ, drt::kOccWords * sizeof(uint32_t));
    if (e == hipSuccess) e = hipMalloc(&h->d_counters, drt::C_COUNT * sizeof(unsigned long long));
    if (e == hipSuccess) e = hipMemset(h->d_counters, 0, drt::C_COUNT * sizeof(unsigned long long));
    if (e != hipSuccess) {
        int rc = fail(nullptr, DRT_ERR_HIP, "device allocation failed: %s", hipGetErrorString(e));
        drt_destroy(h);
        return rc;
    }
    *out = h;
    return DRT_OK;
}

int drt_destroy(drt_handle h)
{
    if (!h) return DRT_OK;
    DeviceGuard g(h->device);
    if (h->d_majorant) (void) hipFree(h->d_majorant);
    if (h->h_majorant) (void) hipHostFree(h->h_majorant);
    if (h->ev_majorant) (void) hipEventDestroy(h->ev_majorant);
    if (h->d_scratch) (void) hipFree(h->d_scratch);
    if (h->d_counters) (void) hipFree(h->d_counters);
    if (h->d_gt) (void) hipFree(h->d_gt);
    if (h->d_queues) (void) hipFree(h->d_queues);
    if (h->d_order) (void) hipFree(h->d_order);
    if (h->d_uempty) (void) hipFree(h->d_uempty);
    if (h->d_sq_tail) (void) hipFree(h->d_sq_tail);
    if (h->d_sq_cold) (void) hipFree(h->d_sq_cold);
    if (h->d_tail) (void) hipFree(h->d_tail);
    if (h->d_sigma_b) (void) hipFree(h->d_sigma_b);
    if (h->d_mgrid) (void) hipFree(h->d_mgrid);
    if (h->d_occ) (void) hipFree(h->d_occ);
    if (h->d_env) (void) hipFree(h->d_env);
    if (h->d_grid4) (void) hipFree(h->d_grid4);
    for (auto &R : h->rec) {
        if (R.mem) (void) hipFree(R.mem);
        if (R.traced) (void) hipEventDestroy(R.traced);
        if (R.reduced) (void) hipEventDestroy(R.reduced);
    }
    if (h->side) (void) hipStreamDestroy(h->side);
    if (h->nerf_stream) (void) hipStreamDestroy(h->nerf_stream);
    if (h->ev_fork) (void) hipEventDestroy(h->ev_fork);
    if (h->ev_join) (void) hipEventDestroy(h->ev_join);
    if (h->d_nerf_bounds) (void) hipFree(h->d_nerf_bounds);
    if (h->ev_split) (void) hipEventDestroy(h->ev_split);
    if (h->ev_hist) (void) hipEventDestroy(h->ev_hist);
    if (h->d_pcache) (void) hipFree(h->d_pcache);
    clear_timings(h);
    delete h;
    return DRT_OK;
}

int drt_release_scratch(drt_handle h)
{
    if (!h) return fail(nullptr, DRT_ERR_INVALID_ARGUMENT, "null handle");
    DeviceGuard g(h->device);
    DRT_HIP_CHECK(h, hipStreamSynchronize(h->stream));
    if (h->side) DRT_HIP_CHECK(h, hipStreamSynchronize(h->side));
    for (auto &R : h->rec) {
        if (R.mem) (void) hipFree(R.mem);
        R.mem = nullptr; R.bytes = 0; R.rays = 0; R.bins = 0; R.busy = false;
    }
    if (h->d_pcache) (void) hipFree(h->d_pcache);
    h->d_pcache = nullptr; h->pcache_bytes = 0; h->pcache_sig.valid = false; h->order_valid = false; h->order_rays = 0;
    if (h->d_tail) (void) hipFree(h->d_tail);
    h->d_tail = nullptr; h->tail_entries = 0;
    return DRT_OK;
}

int drt_set_stream(drt_handle h, void *hip_stream)
{
    if (!h) return fail(nullptr, DRT_ERR_INVALID_ARGUMENT, "null handle");
    h->stream = (hipStream_t) hip_stream;
    return DRT_OK;
}

int drt_set_ray_interleave(drt_handle h, uint64_t chunk_rays, uint64_t stride_rays)
{
    if (!h) return fail(nullptr, DRT_ERR_INVALID_ARGUMENT, "null handle");
    if (chunk_rays && stride_rays < chunk_rays)
        return fail(h, DRT_ERR_INVALID_ARGUMENT, "stride_rays must be >= chunk_rays");
    h->chunk = chunk_rays; h->stride = chunk_rays ? stride_rays : 0;
    return DRT_OK;
}

int drt_synchronize(drt_handle h)
{
    if (!h) return fail(nullptr, DRT_ERR_INVALID_ARGUMENT, "null handle");
    DeviceGuard g(h->device);
    DRT_HIP_CHECK(h, hipStreamSynchronize(h->stream));
    return DRT_OK;
}

int drt_params_changed(drt_handle h)
{
    if (!h) return fail(nullptr, DRT_ERR_INVALID_ARGUMENT, "null handle");
    h->scene_version++; h->medium_version++;
    if (!h->have_medium) return fail(h, DRT_ERR_NOT_CONFIGURED, "no medium set");
    DeviceGuard g(h->device);
    size_t n = (size_t) h->base.rx * h->base.ry * h->base.rz;
    // (with a supergrid the global majorant comes out of the supergrid pass: its cells cover every voxel - one pass over the grid
    //  less per optimisation step)
    if (h->base.mgrid)
        DRT_HIP_CHECK(h, drt::launch_majorant_grid(h->base.sigma_t, h->base.rx, h->base.ry, h->base.rz, h->base.gx, h->base.gy,
                                                   h->base.gz, h->base.scale, h->d_mgrid, (uint32_t *) h->base.mocc, h->stream,
                                                   h->d_scratch, h->d_majorant, (uint32_t *) h->base.mocc_dil));
    else
        DRT_HIP_CHECK(h, drt::launch_majorant(h->base.sigma_t, n, h->base.scale, h->d_scratch, h->d_majorant, h->stream));
    // (best effort: without the pinned word or the event the hint simply stays unknown)
    if (!h->h_majorant && hipHostMalloc((void **) &h->h_majorant, sizeof(float), hipHostMallocDefault) != hipSuccess) { (void) hipGetLastError(); h->h_majorant = nullptr; }
    if (h->h_majorant && !h->ev_majorant && hipEventCreateWithFlags(&h->ev_majorant, hipEventDisableTiming) != hipSuccess) { (void) hipGetLastError(); h->ev_majorant = nullptr; }
    if (h->h_majorant && h->ev_majorant && !h->majorant_pending) {        // (one copy in flight at a time: the word is read only behind its event)
        if (hipMemcpyAsync(h->h_majorant, h->d_majorant, sizeof(float), hipMemcpyDeviceToHost, h->stream) == hipSuccess &&
            hipEventRecord(h->ev_majorant, h->stream) == hipSuccess) h->majorant_pending = true;
        else (void) hipGetLastError();
    }
    DRT_HIP_CHECK(h, drt::launch_occupancy(h->base.sigma_t, h->base.rx, h->base.ry, h->base.rz, h->base.occ_shift, h->base.occ_x,
                                           h->base.occ_y, h->occ_z, h->d_occ, h->base.occ_words, h->stream));
    DRT_HIP_CHECK(h, drt::launch_brick_sigma(h->base.sigma_t, h->d_sigma_b, h->base.rx, h->base.ry, h->base.rz,
                                             h->base.sb_ystride, h->base.sb_zstride / h->base.sb_ystride, h->stream));
    return DRT_OK;
}

int drt_set_medium(drt_handle h, const float *sigma_t, const float *albedo, const int32_t res[3],
                   const float bbox_min[3], const float bbox_max[3], float scale,
                   int32_t majorant_resolution_factor)
{
    if (!h) return fail(nullptr, DRT_ERR_INVALID_ARGUMENT, "null handle");
    if (!sigma_t || !res || !bbox_min || !bbox_max)   /* albedo may be NULL for the nerf integrator */
        return fail(h, DRT_ERR_INVALID_ARGUMENT, "drt_set_medium: null argument");
    for (int a = 0; a < 3; ++a) {
        if (res[a] < 1) return fail(h, DRT_ERR_INVALID_ARGUMENT, "grid resolution must be >= 1");
        if (!(bbox_max[a] > bbox_min[a])) return fail(h, DRT_ERR_INVALID_ARGUMENT, "empty bounding box");
        if (!std::isfinite(bbox_min[a]) || !std::isfinite(bbox_max[a]) || !std::isfinite(bbox_max[a] - bbox_min[a]))
            return fail(h, DRT_ERR_INVALID_ARGUMENT, "bounding box is not finite");
    }
    if (!std::isfinite(scale)) return fail(h, DRT_ERR_INVALID_ARGUMENT, "medium scale is not finite");
    if (scale < 0.0f)                 // (densities of the other sign under the identity activation belong into the grid, where the kernels' bounds look for them)
        return fail(h, DRT_ERR_INVALID_ARGUMENT, "medium scale must be >= 0");
    if ((uint64_t) res[0] * res[1] * res[2] > 0x7fffffffull / 3)
        return fail(h, DRT_ERR_UNSUPPORTED, "grid too large for 32-bit voxel indexing");
    if (majorant_resolution_factor < 0)
        return fail(h, DRT_ERR_INVALID_ARGUMENT, "majorant_resolution_factor must be >= 0");
    drt::Params &B = h->base;
    B.sigma_t = sigma_t; B.albedo = albedo;
    B.rx = res[0]; B.ry = res[1]; B.rz = res[2];
    B.crx = res[0]; B.cry = res[1]; B.crz = res[2]; B.colour_own = 0;     // (until drt_set_colour_resolution says otherwise)
    for (int a = 0; a < 3; ++a) {
        B.bmin[a] = bbox_min[a]; B.bmax[a] = bbox_max[a];
        B.inv_ext[a] = 1.0f / (bbox_max[a] - bbox_min[a]);
    }
    B.scale = scale;
    // gradient scratch: 4 planes in the apron layout (drt_device.h: make_grad_indices), 16/3 x the grid each
    {
        size_t nbx = ((size_t) res[0] + 2) / 3;
        size_t plane = nbx * (size_t) res[1] * (size_t) res[2] * 16;
        if (plane > 0x7fffffffull) return fail(h, DRT_ERR_UNSUPPORTED, "grid too large for the gradient scratch");
        if (plane * 4 != h->gt_floats) {
            DeviceGuard g(h->device);
            if (h->d_gt) { DRT_HIP_CHECK(h, hipStreamSynchronize(h->stream)); (void) hipFree(h->d_gt); h->d_gt = nullptr; h->gt_floats = 0; }
            DRT_HIP_CHECK(h, hipMalloc(&h->d_gt, plane * 4 * sizeof(float)));
            DRT_HIP_CHECK(h, hipMemsetAsync(h->d_gt, 0, plane * 4 * sizeof(float), h->stream));
            h->gt_floats = plane * 4;
        }
        B.gt = h->d_gt; B.gt_plane = (uint32_t) plane; B.gt_nbx = (int) nbx;
    }
    // majorant supergrid (0 = global majorant only)
    if (majorant_resolution_factor > 0) {
        int G[3];
        for (int a = 0; a < 3; ++a) { G[a] = res[a] / majorant_resolution_factor; if (G[a] < 1) G[a] = 1; }
        size_t cells = (size_t) G[0] * G[1] * G[2];
        if (cells != h->mgrid_cells) {
            DeviceGuard g(h->device);
            if (h->d_mgrid) { DRT_HIP_CHECK(h, hipStreamSynchronize(h->stream)); (void) hipFree(h->d_mgrid); h->d_mgrid = nullptr; h->mgrid_cells = 0; }
            DRT_HIP_CHECK(h, hipMalloc(&h->d_mgrid, (cells + 2 * ((cells + 31) / 32)) * sizeof(float)));   // majorants | non-empty bitmask | the same, dilated by one cell
            h->mgrid_cells = cells;
        }
        B.mgrid = h->d_mgrid; B.gx = G[0]; B.gy = G[1]; B.gz = G[2];
        B.mocc = (const uint32_t *) (h->d_mgrid + cells); B.mocc_words = (int) ((cells + 31) / 32);
        B.mocc_dil = B.mocc + B.mocc_words;
    } else {
        B.mgrid = nullptr; B.gx = B.gy = B.gz = 0; B.mocc = nullptr; B.mocc_words = 0; B.mocc_dil = nullptr;
    }
    // empty-space bitmask: cells of 2^shift voxels, at most kOccWords*32 cells
    {
        int shift = 3;
        for (;; ++shift) {
            size_t ox = ((size_t) res[0] >> shift) + 1, oy = ((size_t) res[1] >> shift) + 1, oz = ((size_t) res[2] >> shift) + 1;
            if (ox * oy * oz <= (size_t) drt::kOccWords * 32) {
                B.occ_shift = shift; B.occ_x = (int) ox; B.occ_y = (int) oy;
                B.occ_words = (int) ((ox * oy * oz + 31) / 32);
                h->occ_z = (int) oz;
                break;
            }
        }
        B.occ = h->d_occ;
    }
    // apron-brick sigma_t copy: one 128-byte line per (3x3x1)-voxel base-corner block (eval_sigma_t)
    {
        size_t nbx = ((size_t) res[0] + 2) / 3, nby = ((size_t) res[1] + 2) / 3;
        size_t floats = nbx * nby * (size_t) res[2] * 32;
        if (nbx * nby * (size_t) res[2] > 0x7ffffffull || res[0] > 21000 || res[1] > 21000 || nbx * nby >= (1u << 24) ||
            res[2] >= (1 << 24))                                   // eval_sigma_t indexes with 24-bit multiplies
            return fail(h, DRT_ERR_UNSUPPORTED, "grid too large for the apron-brick copy");
        if (floats != h->sigma_b_floats) {
            DeviceGuard g(h->device);
            if (h->d_sigma_b) { DRT_HIP_CHECK(h, hipStreamSynchronize(h->stream)); (void) hipFree(h->d_sigma_b); h->d_sigma_b = nullptr; h->sigma_b_floats = 0; }
            DRT_HIP_CHECK(h, hipMalloc(&h->d_sigma_b, floats * sizeof(float)));
            h->sigma_b_floats = floats;
        }
        B.sigma_b = h->d_sigma_b;
        B.sb_ystride = (int) nbx; B.sb_zstride = (int) (nby * nbx);
    }
    h->have_medium = true; h->scene_version++;
    return drt_params_changed(h);
}

int drt_set_colour_resolution(drt_handle h, const int32_t res[3])
{
    if (!h) return fail(nullptr, DRT_ERR_INVALID_ARGUMENT, "null handle");
    if (!h->have_medium) return fail(h, DRT_ERR_NOT_CONFIGURED, "drt_set_colour_resolution: no medium set");
    drt::Params &B = h->base;
    if (!res || (res[0] == 0 && res[1] == 0 && res[2] == 0)) {        // back to sigma_t's lattice
        B.crx = B.rx; B.cry = B.ry; B.crz = B.rz; B.colour_own = 0;
        h->scene_version++;
        return DRT_OK;
    }
    for (int a = 0; a < 3; ++a)
        if (res[a] < 1) return fail(h, DRT_ERR_INVALID_ARGUMENT, "colour grid resolution must be >= 1");
    if ((uint64_t) res[0] * res[1] * res[2] > 0x7fffffffull / 3)
        return fail(h, DRT_ERR_UNSUPPORTED, "colour grid too large for 32-bit voxel indexing");
    B.crx = res[0]; B.cry = res[1]; B.crz = res[2];
    B.colour_own = (B.crx != B.rx || B.cry != B.ry || B.crz != B.rz) ? 1 : 0;
    h->scene_version++;
    return DRT_OK;
}

int drt_set_emitter_constant(drt_handle h, const float radiance[3])
{
    if (!h) return fail(nullptr, DRT_ERR_INVALID_ARGUMENT, "null handle");
    if (!radiance) return fail(h, DRT_ERR_INVALID_ARGUMENT, "null radiance");
    for (int k = 0; k < 3; ++k) h->base.Le[k] = radiance[k];
    h->base.env_pix = nullptr; h->base.env_marg = h->base.env_cond = nullptr; h->base.env_gmarg = h->base.env_gcond = nullptr;
    h->have_emitter = true; h->scene_version++;
    return DRT_OK;
}

int drt_set_emitter_envmap(drt_handle h, const float *pixels, int32_t width, int32_t height,
                           const float to_world[9], float scale)
{
    if (!h) return fail(nullptr, DRT_ERR_INVALID_ARGUMENT, "null handle");
    if (!pixels || !to_world) return fail(h, DRT_ERR_INVALID_ARGUMENT, "drt_set_emitter_envmap: null argument");
    if (width < 2 || height < 2 || (int64_t) width * height > (1 << 28))
        return fail(h, DRT_ERR_INVALID_ARGUMENT, "drt_set_emitter_envmap: bad resolution %d x %d", width, height);
    DeviceGuard g(h->device);
    const size_t w = (size_t) width, hh = (size_t) height, n_pix = 3 * w * hh;
    std::vector<float> pix(n_pix);
    DRT_HIP_CHECK(h, hipStreamSynchronize(h->stream));          // a previous map may still be in use
    DRT_HIP_CHECK(h, hipMemcpy(pix.data(), pixels, n_pix * sizeof(float), hipMemcpyDeviceToHost));

    // Importance-sampling tables (DESIGN.md "envmap"): piecewise constant over texels, weight = peak
    // luminance of the 3x3 neighbourhood (covers the bilinear footprint: pdf > 0 wherever the lookup
    // can be > 0) x sin(theta_row); double accumulation, one rounding to float per entry.
    std::vector<float> marg(hh + 1), cond(hh * (w + 1));
    std::vector<double> lum(w * hh), rowsum(hh);
    double lmax = 0.0;
    for (size_t i = 0; i < w * hh; ++i) {
        const float *p = pix.data() + 3 * i;
        double l = 0.212671 * (double) p[0] + 0.715160 * (double) p[1] + 0.072169 * (double) p[2];
        lum[i] = l > 0.0 ? l : 0.0;
        if (lum[i] > lmax) lmax = lum[i];
    }
    const bool uniform = !(lmax > 0.0);                          // all-black map: uniform weights
    double total = 0.0;
    for (int j = 0; j < height; ++j) {
        const double st = sin(3.14159265358979323846 * ((double) j + 0.5) / (double) height);
        float *c = cond.data() + (size_t) j * (w + 1);
        double run = 0.0;
        for (int i = 0; i < width; ++i) {
            double m = uniform ? 1.0 : 0.0;
            if (!uniform)
                for (int dj = -1; dj <= 1; ++dj)
                    for (int di = -1; di <= 1; ++di) {
                        int jj = j + dj, ii = (i + di + width) % width;
                        jj = jj < 0 ? 0 : (jj > height - 1 ? height - 1 : jj);
                        double l = lum[(size_t) jj * w + ii];
                        if (l > m) m = l;
                    }
            run += m * st;
            c[i + 1] = (float) run;
        }
        rowsum[j] = run;
        total += run;
        c[0] = 0.0f;
        for (int i = 0; i < width; ++i)
            c[i + 1] = run > 0.0 ? (float)((double) c[i + 1] / run) : (float)((double)(i + 1) / (double) width);
        c[w] = 1.0f;
    }
    double run = 0.0;
    marg[0] = 0.0f;
    for (int j = 0; j < height; ++j) { run += rowsum[j]; marg[j + 1] = (float)(run / total); }
    marg[hh] = 1.0f;

    // guide tables (cdf_find_guided, drt_device.h): guide[k] = largest index in [0, n - 1] with cdf[index] <= k / n
    auto make_guide = [](const float *cdf, size_t n, uint32_t *guide) {
        size_t idx = 0;
        for (size_t k = 0; k <= n; ++k) {
            const float x = (float) ((double) k / (double) n);
            while (idx + 1 < n && cdf[idx + 1] <= x) ++idx;
            guide[k] = (uint32_t) idx;
        }
    };
    std::vector<uint32_t> gmarg(hh + 1), gcond(hh * (w + 1));
    make_guide(marg.data(), hh, gmarg.data());
    for (size_t j = 0; j < hh; ++j) make_guide(cond.data() + j * (w + 1), w, gcond.data() + j * (w + 1));

    // device layout: float4 texels {r, g, b, density} | conditional CDF rows | their guide rows | marginal CDF | its guide; the rows of the
    // two [h][w + 1] tables are padded to whole 128-byte lines (a guided search touches neighbouring entries of ONE row)
    const size_t cstride = ((w + 1) + 31) & ~(size_t) 31;
    std::vector<float> pix4(4 * w * hh), condp(hh * cstride, 1.0f);
    std::vector<uint32_t> gcondp(hh * cstride, 0u);
    for (size_t j = 0; j < hh; ++j) {
        const float *c = cond.data() + j * (w + 1);
        // the texel's density in uv space, with the float operations the device lookup made per query before round 5 (drt_device.h):
        // (pmf(row) * pmf(col | row)) * (w * h), each difference and product rounded to float
        const float pm = marg[j + 1] - marg[j];
        for (size_t i = 0; i < w; ++i) {
            const float pc = c[i + 1] - c[i];
            float *t = pix4.data() + 4 * (j * w + i);
            t[0] = pix[3 * (j * w + i)]; t[1] = pix[3 * (j * w + i) + 1]; t[2] = pix[3 * (j * w + i) + 2];
            t[3] = (pm * pc) * ((float) width * (float) height);
        }
        std::copy(c, c + w + 1, condp.begin() + j * cstride);
        std::copy(gcond.begin() + j * (w + 1), gcond.begin() + (j + 1) * (w + 1), gcondp.begin() + j * cstride);
    }
    if (h->d_env) { (void) hipFree(h->d_env); h->d_env = nullptr; }
    auto pad32 = [](size_t n) { return (n + 31) & ~(size_t) 31; };
    const size_t o_cond = pad32(pix4.size()), o_gcond = o_cond + condp.size(), o_marg = o_gcond + gcondp.size(), o_gmarg = o_marg + pad32(marg.size());
    const size_t n_all = o_gmarg + pad32(gmarg.size());
    DRT_HIP_CHECK(h, hipMalloc(&h->d_env, n_all * sizeof(float)));
    DRT_HIP_CHECK(h, hipMemcpy(h->d_env, pix4.data(), pix4.size() * sizeof(float), hipMemcpyHostToDevice));
    DRT_HIP_CHECK(h, hipMemcpy(h->d_env + o_cond, condp.data(), condp.size() * sizeof(float), hipMemcpyHostToDevice));
    DRT_HIP_CHECK(h, hipMemcpy(h->d_env + o_gcond, gcondp.data(), gcondp.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
    DRT_HIP_CHECK(h, hipMemcpy(h->d_env + o_marg, marg.data(), marg.size() * sizeof(float), hipMemcpyHostToDevice));
    DRT_HIP_CHECK(h, hipMemcpy(h->d_env + o_gmarg, gmarg.data(), gmarg.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
    h->base.env_pix = (const float4 *) h->d_env;
    h->base.env_cond = h->d_env + o_cond;
    h->base.env_gcond = (const uint32_t *) (h->d_env + o_gcond);
    h->base.env_marg = h->d_env + o_marg;
    h->base.env_gmarg = (const uint32_t *) (h->d_env + o_gmarg);
    h->base.env_cstride = (int) cstride;
    h->base.env_w = width; h->base.env_h = height; h->base.env_scale = scale;
    for (int k = 0; k < 9; ++k) h->base.env_R[k] = to_world[k];
    for (int k = 0; k < 3; ++k) h->base.Le[k] = 0.0f;
    h->have_emitter = true; h->scene_version++;
    return DRT_OK;
}

int drt_set_sensor_perspective(drt_handle h, const float origin[3], const float left[3],
                               const float up[3], const float dir[3], float tan_x, float tan_y,
                               int32_t width, int32_t height)
{
    if (!h) return fail(nullptr, DRT_ERR_INVALID_ARGUMENT, "null handle");
    if (!origin || !left || !up || !dir) return fail(h, DRT_ERR_INVALID_ARGUMENT, "null sensor frame");
    if (width < 1 || height < 1) return fail(h, DRT_ERR_INVALID_ARGUMENT, "film size must be >= 1");
    drt::Params &B = h->base;
    bool same = h->have_sensor && B.tan_x == tan_x && B.tan_y == tan_y && B.width == width && B.height == height;
    for (int k = 0; k < 3; ++k)
        same = same && B.cam_o[k] == origin[k] && B.cam_left[k] == left[k] && B.cam_up[k] == up[k] && B.cam_dir[k] == dir[k];
    for (int k = 0; k < 3; ++k) {
        B.cam_o[k] = origin[k]; B.cam_left[k] = left[k]; B.cam_up[k] = up[k]; B.cam_dir[k] = dir[k];
    }
    B.tan_x = tan_x; B.tan_y = tan_y; B.width = width; B.height = height;
    h->have_sensor = true;
    if (!same) h->scene_version++;                              // (the host layer re-sends the sensor with every sample())
    return DRT_OK;
}

int drt_render_primal(drt_handle h, const float *rays_o, const float *rays_d, uint64_t n_rays,
                      uint64_t ray_offset, uint32_t spp, uint32_t seed, float *L_out)
{
    if (h && n_rays == 0) return DRT_OK;    /* empty batch: nothing to enqueue */
    int rc = check_job(h, rays_o, rays_d, n_rays, ray_offset, spp);
    if (rc) return rc;
    if (!L_out && n_rays) return fail(h, DRT_ERR_INVALID_ARGUMENT, "null L_out");
    DeviceGuard g(h->device);
    drt::Params P;
    fill_job(h, P, rays_o, rays_d, n_rays, ray_offset, spp, seed);
    P.L_out = L_out;
    h->pcache_sig.valid = false;
    bind_path_cache_write(h, P);                                 // every primal kernel records its walks
    {   // the block order left by the previous primal launch of the same shape predicts this one's heavy blocks
        // (same sensor, next step); an order is only ever a schedule, never a result
        const bool no_lpt = dbg(h->debug_flags, 16777216u);   // test hook: plain XCD block map
        const uint64_t n_blocks = (n_rays + 255) / 256;
        (void) n_blocks;
        if (!no_lpt && P.block_cost && h->order_rays == n_rays && !P.mgrid &&
            !dbg(h->debug_flags, (8u | 65536u)))
            P.block_order = P.block_cost + n_blocks;
    }
    rc = timed_launch(h, 0, P, false);
    h->order_valid = false;
    h->perm_valid = false;
    // the cooperative primal kernel (global majorant) and the state machine (supergrid) write the sort keys
    // (only the global-majorant adjoint kernel takes its rays through the schedule: supergrid scenes skip the sort - 0.065 ms of the
    //  factor-8 headline's step)
    if (rc == DRT_OK && P.ray_iters && !P.mgrid && !dbg(h->debug_flags, 8u)) {   // (the plain per-lane primal kernel, bit 8, does not)
        const bool coop_costs = P.block_cost && !P.mgrid && !dbg(h->debug_flags, 65536u);        // the cooperative primal filled block_cost
        DRT_HIP_CHECK(h, drt::launch_ray_perm(P.ray_iters, P.n_rays, perm_base(P.ray_hash, P.n_rays), coop_costs ? P.block_cost : nullptr, h->stream));
        h->perm_valid = true;
    }
    if (rc == DRT_OK && P.block_cost && !P.mgrid && !dbg(h->debug_flags, (8u | 65536u))) {   // cooperative primal: it filled block_cost
        const uint32_t n_blocks = (uint32_t) ((P.n_rays + 255) / 256);
        DRT_HIP_CHECK(h, drt::launch_block_order(P.block_cost, n_blocks, P.block_cost + n_blocks, n_blocks <= kHeavyFirstMaxBlocks, h->stream));
        h->order_valid = true; h->order_rays = n_rays;
    }
    return rc;
}

int drt_render_backward(drt_handle h, const float *rays_o, const float *rays_d, uint64_t n_rays,
                        uint64_t ray_offset, uint32_t spp, uint32_t seed, const float *dL,
                        const float *L_in, float *grad_sigma_t, float *grad_albedo)
{
    if (h && n_rays == 0) return DRT_OK;
    int rc = check_job(h, rays_o, rays_d, n_rays, ray_offset, spp);
    if (rc) return rc;
    if (n_rays && (!dL || !L_in || !grad_sigma_t || !grad_albedo))
        return fail(h, DRT_ERR_INVALID_ARGUMENT, "drt_render_backward: null dL / L_in / gradient buffer");
    DeviceGuard g(h->device);
    drt::Params P;
    fill_job(h, P, rays_o, rays_d, n_rays, ray_offset, spp, seed);
    P.dL = dL; P.L_in = L_in; P.g_sigma = grad_sigma_t; P.g_albedo = grad_albedo;
    // capacity: 48 sigma_t and 6 colour records per ray (headline workload: 12.3 and 1.4); beyond it the
    // tracer falls back to direct atomics (emit_record), so this is a performance choice only
    const uint64_t job_rays = n_rays;
    rc = run_backward(h, P, 48, 6, [&](drt::Params &Q) {
        if (!dbg(h->debug_flags, 32u) || dbg(h->debug_flags, 8u)) bind_path_cache_read(h, Q, job_rays);      // ... and every adjoint kernel but it
        return timed_launch(h, 1, Q, true);
    });
    h->pcache_sig.valid = false;
    return rc;
}

static int nerf_fill(drt_handle h, drt::Params &P, const drt_nerf_config *cfg, const float *emission, bool fused_half = false)
{
    P.nerf_fused_half = fused_half ? 1 : 0;
    if (!cfg || !emission) return fail(h, DRT_ERR_INVALID_ARGUMENT, "nerf: null config / emission grid");
    if (cfg->queries_per_ray < 2) return fail(h, DRT_ERR_INVALID_ARGUMENT, "queries_per_ray must be >= 2");
    P.emission = emission; P.nerf_queries = cfg->queries_per_ray; P.nerf_jitter = cfg->jittering_enabled ? 1 : 0;
    P.nerf_relu = cfg->activation_relu ? 1 : 0; P.hide_emitters = cfg->hide_emitters ? 1 : 0;
    return DRT_OK;
}

static int ensure_grid4(drt_handle h, drt::Params &P, const float *rgb);

// The nerf adjoint of a filled job.  Sensor rays: drt_nerf_tile.hip (a workgroup per pixel tile, the splats pre-reduced in an LDS window and
// flushed with atomics into the caller's grids: no records, no sub-batches); explicit ray batches - the optimisation loop's random pixels -
// and test hook 512: the record path (nerf_kernel + drt_deferred.hip).  g4: lookups from the four-channel copy (the fused pass).
static int nerf_backward(drt_handle h, drt::Params &P, const drt_nerf_config *cfg, bool g4, hipStream_t st = nullptr)
{
    if (!st) st = h->stream;
    if (drt::nerf_tile_supported(P) && !dbg(h->debug_flags, 512u)) {
        hipEvent_t ev[4] = { nullptr, nullptr, nullptr, nullptr };   // (tracer slot and whole-pass slot: the pass is this launch)
        if (h->timing) {
            for (auto &e : ev) DRT_HIP_CHECK(h, hipEventCreate(&e));
            DRT_HIP_CHECK(h, hipEventRecord(ev[0], st));
            DRT_HIP_CHECK(h, hipEventRecord(ev[2], st));
        }
        if (!h->d_nerf_bounds) DRT_HIP_CHECK(h, hipMalloc((void **) &h->d_nerf_bounds, 32));
        DRT_HIP_CHECK(h, drt::launch_nerf_tile_adjoint(P, g4, h->counting, h->d_nerf_bounds, st));
        if (h->timing) {
            DRT_HIP_CHECK(h, hipEventRecord(ev[1], st));
            DRT_HIP_CHECK(h, hipEventRecord(ev[3], st));
            h->timed[1].emplace_back(ev[0], ev[1]);
            h->timed[3].emplace_back(ev[2], ev[3]);
        }
        return DRT_OK;
    }
    const uint32_t q = (uint32_t) cfg->queries_per_ray;          // at most one splat per query and plane
    return run_backward(h, P, q, q, [&](drt::Params &Q) { return timed_nerf(h, 1, Q, true); });
}

static int nerf_primal(drt_handle h, const drt_nerf_config *cfg, const float *emission, const float *rays_o,
                       const float *rays_d, uint64_t n_rays, uint64_t ray_offset, uint32_t spp, uint32_t seed,
                       float *L_out, bool fused_half, hipStream_t st = nullptr)
{
    if (h && n_rays == 0) return DRT_OK;
    int rc = check_job(h, rays_o, rays_d, n_rays, ray_offset, spp, false);
    if (rc) return rc;
    if (!L_out) return fail(h, DRT_ERR_INVALID_ARGUMENT, "null L_out");
    DeviceGuard g(h->device);
    drt::Params P;
    fill_job(h, P, rays_o, rays_d, n_rays, ray_offset, spp, seed);
    rc = nerf_fill(h, P, cfg, emission, fused_half);
    if (rc) return rc;
    P.L_out = L_out;
    if (!fused_half) h->pcache_sig.valid = false;
    return timed_nerf(h, 0, P, false, st);
}

int drt_nerf_render_primal(drt_handle h, const drt_nerf_config *cfg, const float *emission, const float *rays_o,
                           const float *rays_d, uint64_t n_rays, uint64_t ray_offset, uint32_t spp, uint32_t seed,
                           float *L_out)
{
    return nerf_primal(h, cfg, emission, rays_o, rays_d, n_rays, ray_offset, spp, seed, L_out, false);
}

int drt_nerf_render_backward(drt_handle h, const drt_nerf_config *cfg, const float *emission, const float *rays_o,
                             const float *rays_d, uint64_t n_rays, uint64_t ray_offset, uint32_t spp, uint32_t seed,
                             const float *dL, const float *L_in, float *grad_sigma_t, float *grad_emission)
{
    if (h && n_rays == 0) return DRT_OK;
    int rc = check_job(h, rays_o, rays_d, n_rays, ray_offset, spp, false);
    if (rc) return rc;
    if (!dL || !L_in || !grad_sigma_t || !grad_emission)
        return fail(h, DRT_ERR_INVALID_ARGUMENT, "drt_nerf_render_backward: null dL / L_in / gradient buffer");
    DeviceGuard g(h->device);
    drt::Params P;
    fill_job(h, P, rays_o, rays_d, n_rays, ray_offset, spp, seed);
    rc = nerf_fill(h, P, cfg, emission);
    if (rc) return rc;
    P.dL = dL; P.L_in = L_in; P.g_sigma = grad_sigma_t; P.g_albedo = grad_emission;
    // (sensor rays: sigma_t and the emission of a query from ONE 256-byte block of a four-channel copy made for this call)
    const bool tile = drt::nerf_tile_supported(P) && !dbg(h->debug_flags, 512u);
    const bool g4 = tile && ensure_grid4(h, P, emission) == DRT_OK;   // (no memory / a grid beyond the copy's index range: the separate lookups)
    if (tile && !g4) (void) hipGetLastError();
    return nerf_backward(h, P, cfg, g4);
}

// ---- nerf + volpathsimple over one set of grids (BASELINE config 5): two dense passes on two streams, see drt_fused_render_* below ------------
// the interleaved four-channel apron-brick copy [sigma_t, r, g, b] (eval4): (re)built when the parameter grids changed since the last copy
// `rgb`: the colour grid of the copy - the medium's albedo (the fused pass: the copy is kept until the parameters change) or the caller's emission
// grid of a stand-alone nerf call (no version to go by: copied on every call, 0.46 ms at 256^3 against the 3 ms the four-channel lookups save)
static int ensure_grid4(drt_handle h, drt::Params &P, const float *rgb)
{
    const bool own = rgb == nullptr || rgb == h->base.albedo;
    if (!rgb) rgb = h->base.albedo;
    const drt::Params &B = h->base;
    if (B.colour_own) return fail(h, DRT_ERR_UNSUPPORTED, "the four-channel copy needs the colour grid on sigma_t's lattice");
    const size_t nbx = ((size_t) B.rx + 2) / 3, quads = nbx * (size_t) B.ry * (size_t) B.rz * 16;
    if (quads > 0x7fffffffull) return fail(h, DRT_ERR_UNSUPPORTED, "grid too large for the four-channel copy");
    if (quads != h->grid4_quads) {
        if (h->d_grid4) { DRT_HIP_CHECK(h, hipStreamSynchronize(h->stream)); (void) hipFree(h->d_grid4); h->d_grid4 = nullptr; h->grid4_quads = 0; }
        DRT_HIP_CHECK(h, hipMalloc(&h->d_grid4, quads * sizeof(float4)));
        h->grid4_quads = quads; h->grid4_version = 0;
    }
    if (!own || h->grid4_version != h->medium_version) {   // the parameter grids (may) have changed since the copy was made
        DRT_HIP_CHECK(h, drt::launch_brick_grid4(B.sigma_t, rgb, h->d_grid4, B.rx, B.ry, B.rz, (int) nbx, h->stream));
        h->grid4_version = own ? h->medium_version : 0;
    }
    P.grid4 = h->d_grid4; P.g4_nbx = (int) nbx;
    return DRT_OK;
}

// The fused pass = the two integrators over the SAME rays and the same four grids, each on its own copy of the sampler stream (round 5: as two
// dense passes - rounds 3/4 ran both halves in one kernel, drt_fused*.hip, whose nerf half emitted one gradient record per query: 907 M
// records per step of config 5, 25 of the adjoint pass's 46 ms in the reduction; and whose volpathsimple half ran one path per lane at a third
// of the lanes).  Now: the nerf march as the stand-alone integrator with emission = the medium's albedo grid (primal: nerf_kernel; adjoint for
// sensor rays: drt_nerf_tile.hip, lookups from the four-channel copy, splats pre-reduced in LDS), and the volpathsimple half through
// drt_render_primal / drt_render_backward - i.e. the queued supergrid tracer or the wave-cooperative tracer, path cache, record streams and
// ONE tile_reduce.  Results: radiance of both halves bit-exact as before (the same statements per ray), gradients up to summation order.
// the nerf half on its own stream beside the volpathsimple half: fork behind what the handle's stream holds so far (the caller's zeroed gradient
// buffers, the four-channel copy), join before the call returns.  The two halves write disjoint radiance buffers and add to the gradient grids
// with atomics only (the window flush of drt_nerf_tile.hip, tile_reduce's flush, direct splats): no order between them is needed.
static int fused_fork(drt_handle h)
{
    if (!h->nerf_stream) {
        DRT_HIP_CHECK(h, hipStreamCreateWithFlags(&h->nerf_stream, hipStreamNonBlocking));
        DRT_HIP_CHECK(h, hipEventCreateWithFlags(&h->ev_fork, hipEventDisableTiming));
        DRT_HIP_CHECK(h, hipEventCreateWithFlags(&h->ev_join, hipEventDisableTiming));
    }
    DRT_HIP_CHECK(h, hipEventRecord(h->ev_fork, h->stream));
    DRT_HIP_CHECK(h, hipStreamWaitEvent(h->nerf_stream, h->ev_fork, 0));
    return DRT_OK;
}

static int fused_join(drt_handle h)
{
    DRT_HIP_CHECK(h, hipEventRecord(h->ev_join, h->nerf_stream));
    DRT_HIP_CHECK(h, hipStreamWaitEvent(h->stream, h->ev_join, 0));
    return DRT_OK;
}

int drt_fused_render_primal(drt_handle h, const drt_nerf_config *cfg, const float *rays_o, const float *rays_d, uint64_t n_rays,
                            uint64_t ray_offset, uint32_t spp, uint32_t seed, float *L_nerf_out, float *L_drt_out)
{
    if (h && n_rays == 0) return DRT_OK;
    int rc = check_job(h, rays_o, rays_d, n_rays, ray_offset, spp);
    if (rc) return rc;
    if (!L_nerf_out || !L_drt_out) return fail(h, DRT_ERR_INVALID_ARGUMENT, "drt_fused_render_primal: null output buffer");
    if (!cfg) return fail(h, DRT_ERR_INVALID_ARGUMENT, "fused: null nerf config");
    {
        DeviceGuard g(h->device);
        rc = fused_fork(h);
        if (rc) return rc;
    }
    rc = nerf_primal(h, cfg, h->base.albedo, rays_o, rays_d, n_rays, ray_offset, spp, seed, L_nerf_out, true, h->nerf_stream);
    int rc2 = rc ? rc : drt_render_primal(h, rays_o, rays_d, n_rays, ray_offset, spp, seed, L_drt_out);   // (its path cache serves the backward pass)
    {
        DeviceGuard g(h->device);
        const int rj = fused_join(h);
        if (!rc2) rc2 = rj;
    }
    return rc2;
}

int drt_fused_render_backward(drt_handle h, const drt_nerf_config *cfg, const float *rays_o, const float *rays_d, uint64_t n_rays,
                              uint64_t ray_offset, uint32_t spp, uint32_t seed, const float *dL_nerf, const float *L_nerf_in,
                              const float *dL_drt, const float *L_drt_in, float *grad_sigma_t, float *grad_rgb)
{
    if (h && n_rays == 0) return DRT_OK;
    int rc = check_job(h, rays_o, rays_d, n_rays, ray_offset, spp);
    if (rc) return rc;
    if (!dL_nerf || !L_nerf_in || !dL_drt || !L_drt_in || !grad_sigma_t || !grad_rgb)
        return fail(h, DRT_ERR_INVALID_ARGUMENT, "drt_fused_render_backward: null dL / L_in / gradient buffer");
    hipEvent_t t0 = nullptr, t1 = nullptr;
    size_t n_pass = 0;
    {
        DeviceGuard g(h->device);
        drt::Params P;
        fill_job(h, P, rays_o, rays_d, n_rays, ray_offset, spp, seed);
        rc = nerf_fill(h, P, cfg, h->base.albedo, true);
        if (rc) return rc;
        P.dL = dL_nerf; P.L_in = L_nerf_in; P.g_sigma = grad_sigma_t; P.g_albedo = grad_rgb;
        const bool tile = drt::nerf_tile_supported(P) && !dbg(h->debug_flags, 512u);
        const bool g4 = tile && ensure_grid4(h, P, nullptr) == DRT_OK;   // (sigma_t and the colour of a query from ONE 256-byte block; else: separate lookups)
        if (tile && !g4) (void) hipGetLastError();
        if (h->timing) {
            DRT_HIP_CHECK(h, hipEventCreate(&t0));
            DRT_HIP_CHECK(h, hipEventCreate(&t1));
            DRT_HIP_CHECK(h, hipEventRecord(t0, h->stream));
            n_pass = h->timed[3].size();
        }
        // the tile kernel (one workgroup of 16 waves and 139 KiB of LDS per CU) leaves room for the wave-cooperative tracer's workgroups beside it;
        // the record path of explicit ray batches shares the handle's record streams with the volpathsimple half: one after the other
        bool forked = false;
        if (tile) {
            rc = fused_fork(h);
            forked = rc == DRT_OK;
            if (!rc) rc = nerf_backward(h, P, cfg, g4, h->nerf_stream);
        } else rc = nerf_backward(h, P, cfg, false);
        if (rc) {                                                    // (nothing of the other stream may outlive the call)
            if (forked) (void) fused_join(h);
            if (t0) { (void) hipEventDestroy(t0); (void) hipEventDestroy(t1); }
            return rc;
        }
        // (launched first: its workgroups need almost a whole CU's LDS, which the other half's many small workgroups would not leave free)
    }
    rc = drt_render_backward(h, rays_o, rays_d, n_rays, ray_offset, spp, seed, dL_drt, L_drt_in, grad_sigma_t, grad_rgb);
    {
        DeviceGuard g(h->device);
        if (h->nerf_stream) { const int rj = fused_join(h); if (!rc) rc = rj; }
        if (h->timing && t0) {                                   // the whole pass as ONE entry (the halves overlap)
            DRT_HIP_CHECK(h, hipEventRecord(t1, h->stream));
            while (h->timed[3].size() > n_pass) { (void) hipEventDestroy(h->timed[3].back().first); (void) hipEventDestroy(h->timed[3].back().second); h->timed[3].pop_back(); }
            h->timed[3].emplace_back(t0, t1);
        }
    }
    return rc;
}

int drt_batch_sample_rays_range(drt_handle h, const float *sensors, int32_t n_sensors, uint32_t batch_first,
                                uint32_t batch_count, uint32_t spp, uint32_t sub_seed_pixels, uint32_t sub_seed_rays,
                                float *rays_o, float *rays_d, uint32_t *sensor_idx, uint32_t *pixels)
{
    if (!h) return fail(nullptr, DRT_ERR_INVALID_ARGUMENT, "null handle");
    if (!sensors || n_sensors < 1 || spp == 0) return fail(h, DRT_ERR_INVALID_ARGUMENT, "drt_batch_sample_rays: bad sensors / spp");
    if (((uint64_t) batch_first + batch_count) * spp > 0xffffffffull)
        return fail(h, DRT_ERR_INVALID_ARGUMENT, "batch_size * spp exceeds 2^32 - 1");
    if (batch_count && (!rays_o || !rays_d)) return fail(h, DRT_ERR_INVALID_ARGUMENT, "null ray buffers");
    DeviceGuard g(h->device);
    DRT_HIP_CHECK(h, drt::launch_batch_raygen(sensors, n_sensors, batch_first, batch_count, spp, sub_seed_pixels, sub_seed_rays,
                                              rays_o, rays_d, sensor_idx, pixels, h->stream));
    return DRT_OK;
}

int drt_batch_sample_rays(drt_handle h, const float *sensors, int32_t n_sensors, uint32_t batch_size, uint32_t spp,
                          uint32_t sub_seed_pixels, uint32_t sub_seed_rays, float *rays_o, float *rays_d,
                          uint32_t *sensor_idx, uint32_t *pixels)
{
    return drt_batch_sample_rays_range(h, sensors, n_sensors, 0, batch_size, spp, sub_seed_pixels, sub_seed_rays, rays_o,
                                       rays_d, sensor_idx, pixels);
}

int drt_film_develop(drt_handle h, const float *L, uint64_t n_pixels, uint32_t spp, float *image)
{
    if (!h) return fail(nullptr, DRT_ERR_INVALID_ARGUMENT, "null handle");
    if (spp == 0 || (n_pixels && (!L || !image))) return fail(h, DRT_ERR_INVALID_ARGUMENT, "drt_film_develop: bad argument");
    DeviceGuard g(h->device);
    DRT_HIP_CHECK(h, drt::launch_film_develop(L, n_pixels, spp, image, h->stream));
    return DRT_OK;
}

int drt_adam_step_clamped(void *hip_stream, float *p, const float *g, float *m, float *v, uint64_t n, double beta_1, double beta_2,
                          double epsilon, double lr_t, float lo, float hi)
{
    if (!p || !g || !m || !v) return fail(nullptr, DRT_ERR_INVALID_ARGUMENT, "drt_adam_step_clamped: null buffer");
    if ((((uintptr_t) p) | ((uintptr_t) g) | ((uintptr_t) m) | ((uintptr_t) v)) & 15u)
        return fail(nullptr, DRT_ERR_INVALID_ARGUMENT, "drt_adam_step_clamped: buffers must be 16-byte aligned");
    if (!(lo <= hi)) return fail(nullptr, DRT_ERR_INVALID_ARGUMENT, "drt_adam_step_clamped: lo > hi");
    return drt::launch_adam_step(p, g, m, v, n, beta_1, beta_2, epsilon, lr_t, (hipStream_t) hip_stream, lo, hi) == hipSuccess ? DRT_OK : DRT_ERR_HIP;
}

int drt_adam_step(void *hip_stream, float *p, const float *g, float *m, float *v, uint64_t n, double beta_1, double beta_2,
                  double epsilon, double lr_t)
{
    if (n && (!p || !g || !m || !v)) return DRT_ERR_INVALID_ARGUMENT;
    if (((uintptr_t) p | (uintptr_t) g | (uintptr_t) m | (uintptr_t) v) % 16) return DRT_ERR_INVALID_ARGUMENT;
    return drt::launch_adam_step(p, g, m, v, n, beta_1, beta_2, epsilon, lr_t, (hipStream_t) hip_stream) == hipSuccess ? DRT_OK : DRT_ERR_HIP;
}

int drt_grad_support_mask(void *hip_stream, const float *sigma_t, const int32_t res[3], uint64_t sparse_offset_floats, uint32_t channels,
                          uint64_t n_blocks, uint32_t block_floats, uint32_t *bits_scratch, uint8_t *mask)
{
    if (!sigma_t || !res || !bits_scratch || (n_blocks && !mask)) return DRT_ERR_INVALID_ARGUMENT;
    if (res[0] <= 0 || res[1] <= 0 || res[2] <= 0 || channels == 0 || block_floats == 0) return DRT_ERR_INVALID_ARGUMENT;
    return drt::launch_support_mask(sigma_t, res[0], res[1], res[2], sparse_offset_floats, channels, n_blocks, block_floats, bits_scratch, mask,
                                    (hipStream_t) hip_stream) == hipSuccess ? DRT_OK : DRT_ERR_HIP;
}

int drt_grad_block_mask(void *hip_stream, const float *buf, uint64_t n_blocks, uint32_t block_floats, uint8_t *mask)
{
    if (n_blocks && (!buf || !mask)) return DRT_ERR_INVALID_ARGUMENT;
    if (block_floats != 64 && block_floats != 128 && block_floats != 256) return DRT_ERR_INVALID_ARGUMENT;
    if ((uintptr_t) buf % 16) return DRT_ERR_INVALID_ARGUMENT;
    return drt::launch_block_mask(buf, n_blocks, block_floats, mask, (hipStream_t) hip_stream) == hipSuccess ? DRT_OK : DRT_ERR_HIP;
}

int drt_grad_block_positions(void *hip_stream, const uint8_t *mask, uint64_t n_blocks, int32_t *pos, int32_t *count, uint32_t *scratch)
{
    if (n_blocks && (!mask || !pos || !scratch)) return DRT_ERR_INVALID_ARGUMENT;
    return drt::launch_block_positions(mask, n_blocks, pos, count, scratch, (hipStream_t) hip_stream) == hipSuccess ? DRT_OK : DRT_ERR_HIP;
}

int drt_grad_pack(void *hip_stream, const float *flat, const int32_t *pos, uint64_t n_blocks, uint32_t block_floats, float *packed, float *check)
{
    if (n_blocks && (!flat || !pos || !packed || !check)) return DRT_ERR_INVALID_ARGUMENT;
    if (block_floats != 64 && block_floats != 128 && block_floats != 256) return DRT_ERR_INVALID_ARGUMENT;
    if (((uintptr_t) flat | (uintptr_t) packed) % 16) return DRT_ERR_INVALID_ARGUMENT;
    return drt::launch_grad_pack(const_cast<float *>(flat), pos, n_blocks, block_floats, packed, check, false, (hipStream_t) hip_stream) == hipSuccess
               ? DRT_OK : DRT_ERR_HIP;
}

int drt_grad_unpack(void *hip_stream, const float *packed, const int32_t *pos, uint64_t n_blocks, uint32_t block_floats, float *flat)
{
    if (n_blocks && (!flat || !pos || !packed)) return DRT_ERR_INVALID_ARGUMENT;
    if (block_floats != 64 && block_floats != 128 && block_floats != 256) return DRT_ERR_INVALID_ARGUMENT;
    if (((uintptr_t) flat | (uintptr_t) packed) % 16) return DRT_ERR_INVALID_ARGUMENT;
    return drt::launch_grad_pack(flat, pos, n_blocks, block_floats, const_cast<float *>(packed), nullptr, true, (hipStream_t) hip_stream) == hipSuccess
               ? DRT_OK : DRT_ERR_HIP;
}

int drt_film_backward(drt_handle h, const float *grad_image, uint64_t n_pixels, uint32_t spp, float *dL)
{
    if (!h) return fail(nullptr, DRT_ERR_INVALID_ARGUMENT, "null handle");
    if (spp == 0 || (n_pixels && (!grad_image || !dL))) return fail(h, DRT_ERR_INVALID_ARGUMENT, "drt_film_backward: bad argument");
    DeviceGuard g(h->device);
    DRT_HIP_CHECK(h, drt::launch_film_backward(grad_image, n_pixels, spp, dL, h->stream));
    return DRT_OK;
}

int drt_debug_eval(drt_handle h, int op, const float *in, uint64_t n, float *out)
{
    if (!h) return fail(nullptr, DRT_ERR_INVALID_ARGUMENT, "null handle");
    if (n && (!in || !out)) return fail(h, DRT_ERR_INVALID_ARGUMENT, "drt_debug_eval: null buffer");
    if ((op == 3 || op == 4 || op == 5) && !h->have_medium) return fail(h, DRT_ERR_NOT_CONFIGURED, "no medium set");
    if (op == 7 && !h->have_sensor) return fail(h, DRT_ERR_NOT_CONFIGURED, "no sensor set");
    DeviceGuard g(h->device);
    drt::Params P = h->base;
    P.majorant = h->d_majorant;
    DRT_HIP_CHECK(h, drt::launch_debug_eval(P, op, in, n, out, h->stream));
    return DRT_OK;
}

int drt_set_debug_flags(drt_handle h, uint32_t flags)
{
    if (!h) return fail(nullptr, DRT_ERR_INVALID_ARGUMENT, "null handle");
    if (!drt::kTestHooks && flags)
        return fail(h, DRT_ERR_UNSUPPORTED, "this is the production library: test hooks are compiled out (use libdrt_hip_hooks.so)");
    h->debug_flags = flags; h->scene_version++;
    return DRT_OK;
}

int drt_nerf_tile_stats(drt_handle h, uint64_t *lds_lane_adds)
{
    if (!h || !lds_lane_adds) return fail(h, DRT_ERR_INVALID_ARGUMENT, "drt_nerf_tile_stats: null argument");
    *lds_lane_adds = 0;
    if (!h->d_nerf_bounds) return DRT_OK;                          // (no tile launch yet)
    DeviceGuard g(h->device);
    DRT_HIP_CHECK(h, hipStreamSynchronize(h->stream));
    if (h->nerf_stream) DRT_HIP_CHECK(h, hipStreamSynchronize(h->nerf_stream));
    unsigned long long v = 0;
    DRT_HIP_CHECK(h, hipMemcpy(&v, h->d_nerf_bounds + 6, sizeof v, hipMemcpyDeviceToHost));
    *lds_lane_adds = (uint64_t) v;
    return DRT_OK;
}

int drt_enable_counters(drt_handle h, int enable)
{
    if (!h) return fail(nullptr, DRT_ERR_INVALID_ARGUMENT, "null handle");
    h->counting = enable != 0;
    return DRT_OK;
}

int drt_reset_counters(drt_handle h)
{
    if (!h) return fail(nullptr, DRT_ERR_INVALID_ARGUMENT, "null handle");
    DeviceGuard g(h->device);
    DRT_HIP_CHECK(h, hipMemsetAsync(h->d_counters, 0, drt::C_COUNT * sizeof(unsigned long long), h->stream));
    return DRT_OK;
}

int drt_get_counters(drt_handle h, drt_counters *out)
{
    if (!h || !out) return fail(h, DRT_ERR_INVALID_ARGUMENT, "drt_get_counters: null argument");
    DeviceGuard g(h->device);
    unsigned long long host[drt::C_COUNT];
    DRT_HIP_CHECK(h, hipMemcpyAsync(host, h->d_counters, sizeof host, hipMemcpyDeviceToHost, h->stream));
    DRT_HIP_CHECK(h, hipStreamSynchronize(h->stream));
    out->n_rays = host[drt::C_RAYS]; out->n_dt = host[drt::C_DT]; out->n_rt = host[drt::C_RT];
    out->n_drt = host[drt::C_DRT]; out->n_alb = host[drt::C_ALB]; out->n_tr = host[drt::C_TR];
    out->n_rt_adj = host[drt::C_RT_ADJ]; out->n_sc = host[drt::C_SC]; out->n_sc_alb = host[drt::C_SC_ALB];
    return DRT_OK;
}

int drt_enable_timing(drt_handle h, int enable)
{
    if (!h) return fail(nullptr, DRT_ERR_INVALID_ARGUMENT, "null handle");
    DeviceGuard g(h->device);
    DRT_HIP_CHECK(h, hipStreamSynchronize(h->stream));
    clear_timings(h);
    h->timing = enable != 0;
    return DRT_OK;
}

int drt_read_timings(drt_handle h, int backward, float *out_ms, int capacity)
{
    if (!h) return fail(nullptr, DRT_ERR_INVALID_ARGUMENT, "null handle");
    if (capacity > 0 && !out_ms) return fail(h, DRT_ERR_INVALID_ARGUMENT, "null out_ms");
    DeviceGuard g(h->device);
    DRT_HIP_CHECK(h, hipStreamSynchronize(h->stream));
    if (backward < 0 || backward > 3) return fail(h, DRT_ERR_INVALID_ARGUMENT, "drt_read_timings: kind must be 0..3");
    auto &v = h->timed[backward];
    int n = (int) v.size();
    for (int i = 0; i < n && i < capacity; ++i)
        DRT_HIP_CHECK(h, hipEventElapsedTime(out_ms + i, v[i].first, v[i].second));
    return n;
}

}  // extern "C"
