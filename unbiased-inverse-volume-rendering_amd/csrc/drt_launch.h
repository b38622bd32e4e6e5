// drt_launch.h -- host-side launch interface between the C ABI (drt_capi.cpp)
// and the kernels (drt_kernels.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "drt_device.h"

namespace drt {

hipError_t launch_trace(const Params &P, bool adjoint, bool count, hipStream_t stream);
hipError_t launch_majorant_grid(const float *sigma_t, int rx, int ry, int rz, int gx, int gy, int gz, float scale,
                                float *out, uint32_t *mask, hipStream_t stream, uint32_t *max_bits = nullptr, float *majorant = nullptr,
                                uint32_t *mask_dil = nullptr);
hipError_t launch_occupancy(const float *sigma_t, int rx, int ry, int rz, int shift, int ox, int oy, int oz,
                            uint32_t *occ, int words, hipStream_t stream);
hipError_t launch_brick_sigma(const float *src, float *dst, int rx, int ry, int rz, int nbx, int nby,
                              hipStream_t stream);
hipError_t launch_nerf(const Params &P, bool adjoint, bool count, hipStream_t stream);
// drt_own.hip: the same kernels with the colour grids on their own lattice (Params::colour_own)
hipError_t launch_nerf_own(const Params &P, bool adjoint, bool count, hipStream_t stream);
hipError_t launch_trace_own(const Params &P, bool adjoint, bool count, hipStream_t stream);
// drt_nerf_tile.hip: the nerf adjoint for sensor rays - a workgroup per pixel tile, its splats pre-reduced in an LDS window of 16^3 voxels, no records
// (g4: lookups from Params::grid4 - the fused pass, emission = the medium's albedo grid - instead of sigma_b + Params::emission)
bool nerf_tile_supported(const Params &P);
// bounds: 32 bytes of device scratch (the fixed-point units of the LDS window follow from max |dL|, max |L_in|, max |emission|, reduced there first)
hipError_t launch_nerf_tile_adjoint(const Params &P, bool g4, bool count, uint32_t *bounds, hipStream_t stream);
hipError_t launch_trace_wavefront(const Params &P, bool adjoint, bool count, int n_cus, hipStream_t stream);
// supergrid scenes (majorant_resolution_factor > 0): lane-level state machine stepping one supergrid cell at a time, the
// majorant grid in LDS (drt_super.hip); the adjoint needs the record streams (deferred splatting)
bool super_supported(const Params &P);
hipError_t launch_trace_super(const Params &P, bool adjoint, bool count, int n_cus, hipStream_t stream);
// Ray order for launch_trace_super (Params::order): units of `unit` consecutive rays of [P.ray_first, P.n_rays) sorted by a
// cost key - the majorant optical depth along the unit's first ray through the supergrid -, most expensive first.
// work: super_order_bytes(units) bytes; the permutation is the first `units` words of it.
// the same scenes as work queues inside a compute unit: rays live in LDS records and belong to no lane, waves take batches of
// one kind of work (drt_sq.hip; round 4).  Uses the ray order and the XCD queues of launch_trace_super; the adjoint needs
// Params::sq_cold (sq_cold_bytes(n_cus) bytes)
bool sq_supported(const Params &P);
size_t sq_cold_bytes(int n_cus);
hipError_t launch_trace_sq(const Params &P, bool adjoint, bool count, int n_cus, hipStream_t stream);
// tail pool of the queued tracer's adjoint launches (Params::tail_pool / tail_count / tail_cap / tail_mode): a drained workgroup writes its last
// <= sq_tail_push() records to the pool and ends; launch_trace_sq with tail_mode = 1 finishes them (its splats as direct atomics)
uint32_t sq_tail_push();
bool sq_tail_solo(const Params &P);       // the tail launch finishes its records in registers, no queue hops (supergrids whose majorants fit LDS)
size_t sq_tail_entry_quads();
size_t super_order_bytes(uint32_t units);
// flags[u] = 1: every ray of unit u (the `unit` = spp rays of one pixel, sensor rays only) crosses only empty supergrid cells (Params::unit_empty)
hipError_t build_unit_empty(const Params &P, uint32_t unit, uint32_t units, uint8_t *flags, hipStream_t stream);
hipError_t build_super_order(const Params &P, uint32_t unit, uint32_t units, void *work, hipStream_t stream, const uint8_t *iters = nullptr);
// one ray per lane with wave-cooperative tracking loops (drt_coop.hip); global majorant only (P.mgrid == nullptr)
hipError_t launch_trace_coop_super(const Params &P, bool adjoint, bool count, hipStream_t stream);
// `between` (optional): called on the host after the main launch has been enqueued and before the tail launch (adjoint of the
// specialised kernels with a tail pool); returns whether it was called through *called
typedef hipError_t (*coop_between_fn)(void *ctx);
hipError_t launch_trace_coop(const Params &P, bool adjoint, bool count, hipStream_t stream, coop_between_fn between = nullptr,
                             void *between_ctx = nullptr, bool *called = nullptr);
hipError_t launch_ray_perm(const uint8_t *iters, uint64_t n_rays, uint16_t *perm, uint32_t *block_cost, hipStream_t stream);
hipError_t launch_block_order(const uint32_t *cost, uint32_t n_blocks, uint32_t *order, bool heavy_first, hipStream_t stream);
hipError_t launch_untile(const Params &P, hipStream_t stream);
// the interleaved four-channel apron-brick copy [sigma_t, r, g, b] of the medium (eval4; drt_nerf_tile.hip)
hipError_t launch_brick_grid4(const float *sigma_t, const float *rgb, float4 *dst, int rx, int ry, int rz, int nbx, hipStream_t stream);

// Deferred splatting (drt_deferred.hip): record streams -> tile partition -> LDS reduction.
constexpr int kTileX = 32, kTileY = 16, kTileZ = 16;   // base-corner cells per tile; LDS tile = 33 x 17 x 17 floats
constexpr int kMaxBins = 16384;                          // tiles per grid the one-pass partition handles (64 KiB LDS histogram): 512^3
constexpr uint32_t kUnitRecords = 16384;                 // records per reduce workgroup
#ifndef DRT_PART_WGS
#define DRT_PART_WGS 256
#endif
#ifndef DRT_PART_THREADS
#define DRT_PART_THREADS 1024
#endif
// Partition workgroups (histogram / scatter): every (workgroup, tile) pair owns an output sub-range, i.e.
// one partially written line at a time; few, large workgroups keep that working set (WGs x tiles x 128 B)
// inside the 256 MB memory-side cache.
constexpr int kPartWGs = DRT_PART_WGS;
constexpr int kPartThreads = DRT_PART_THREADS;
struct DeferredPlan {
    float4 *in[2], *out[2];          // record streams as emitted / tile-sorted (stream 0: 1 float4 per record, stream 1: 2)
    uint32_t *chunk_count[2];        // valid records per chunk of in[s]
    uint32_t cap_chunks[2];
    uint32_t *cursor;                // [0..1] chunks handed out (may exceed the capacity), [4..5] overflowed splats
    uint32_t *hist;                  // [2][kPartWGs][n_bins] counts, then exclusive offsets
    uint32_t *bin_base;              // [2][n_bins + 1]
    uint32_t *unit_start;            // [2][n_bins + 1] first reduce unit of every tile
    uint32_t *vmax;                  // [5] bit pattern of max |value| per reduce plane (zeroed per launch):
                                     //     [0] stream 0; [1..4] stream 1's sigma_t, r, g, b
    int n_bins, ntx, nty, ntz;
    uint32_t max_units;              // launch bound of the reduce kernel (any stream)
};
// ev: optional 5 events recorded before/after the stages (histogram | offsets+scan | scatter | reduce)
// early_hist: the histogram of the chunks below the split (launch_deferred_early_histogram) has been taken already
hipError_t launch_deferred_reduce(const Params &P, const DeferredPlan &D, hipStream_t stream, hipEvent_t *ev = nullptr, bool early_hist = false, int phase = 0);
// The adjoint tracer's tail launch keeps few workgroups busy for as long as the job's longest path: between the main and
// the tail launch the record streams' chunk cursors are snapshot (`split`, on `stream`), and the histogram pass over the
// chunks below the split - everything the main launch wrote - runs on `side` next to the tail launch.
hipError_t launch_deferred_split(const DeferredPlan &D, hipStream_t stream);
hipError_t launch_deferred_early_histogram(const Params &P, const DeferredPlan &D, hipStream_t side);
hipError_t launch_majorant(const float *sigma_t, size_t n, float scale, uint32_t *scratch_bits,
                           float *majorant, hipStream_t stream);
hipError_t launch_batch_raygen(const float *sensors, int n_sensors, uint32_t batch_first, uint32_t batch_size, uint32_t spp,
                               uint32_t seed_pixels, uint32_t seed_rays, float *rays_o, float *rays_d, uint32_t *sensor_idx,
                               uint32_t *pixels, hipStream_t stream);
hipError_t launch_adam_step(float *p, const float *g, float *m, float *v, uint64_t n, double b1, double b2, double eps, double lr_t,
                            hipStream_t stream, float lo = -__builtin_huge_valf(), float hi = __builtin_huge_valf());
hipError_t launch_support_mask(const float *sigma_t, int rx, int ry, int rz, uint64_t sparse_off, uint32_t ch, uint64_t n_blocks,
                               uint32_t block_floats, uint32_t *bits, uint8_t *mask, hipStream_t stream);
hipError_t launch_block_mask(const float *buf, uint64_t n_blocks, uint32_t block_floats, uint8_t *mask, hipStream_t stream);
// packing of the one-collective gradient all-reduce: ranks of the set's blocks (group_count: ceil(n_blocks / 1024) words of scratch), the
// gather into the packed buffer fused with the check of the blocks outside the set, and the scatter back
hipError_t launch_block_positions(const uint8_t *mask, uint64_t n_blocks, int32_t *pos, int32_t *count, uint32_t *group_count, hipStream_t stream);
hipError_t launch_grad_pack(float *flat, const int32_t *pos, uint64_t n_blocks, uint32_t block_floats, float *packed, float *check, bool unpack,
                            hipStream_t stream);
hipError_t launch_film_develop(const float *L, uint64_t n_pixels, uint32_t spp, float *image,
                               hipStream_t stream);
hipError_t launch_film_backward(const float *grad_image, uint64_t n_pixels, uint32_t spp, float *dL,
                                hipStream_t stream);
uint32_t host_alt_seed(uint32_t seed, bool sensor_flow);
hipError_t launch_debug_eval(const Params &P, int op, const float *in, uint64_t n, float *out, hipStream_t stream);

}  // namespace drt
