/*
 * drt_hip.h -- C ABI of libdrt_hip.so: the MI355X (gfx950) differential ratio
 * tracking integrator.  This is the drop-in boundary for the hot path of
 * rgl-epfl/unbiased-inverse-volume-rendering: what the reference reaches through
 * `integrator.sample(mode, scene, sampler, ray, dL, state_in, ...)`
 * (python/integrators/volpathsimple.py:38-49) and the primal -> dL -> adjoint
 * harness around it (python/batched.py:134-197, 212-326).
 *
 * Conventions
 *  - plain C, no C++/torch types; every call returns 0 on success or a negative
 *    drt_status; `drt_last_error` returns a human-readable message.
 *  - all tensor arguments are caller-owned DEVICE pointers (HIP, fp32) on the
 *    handle's device; the library never takes ownership and never allocates
 *    caller-visible memory.  Grids use Mitsuba's VolumeGrid layout (Z,Y,X,C).
 *  - all work is enqueued on the handle's stream (drt_set_stream) and is
 *    asynchronous with respect to the host, like Dr.Jit kernels until dr.eval().
 *  - one handle is bound to one device; a handle is not thread-safe, distinct
 *    handles are independent.
 *  - the library needs a GPU: there is no CPU fallback behind this ABI.
 */
#ifndef DRT_HIP_H
#define DRT_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct drt_handle_s *drt_handle;

typedef enum drt_status {
    DRT_OK = 0,
    DRT_ERR_INVALID_ARGUMENT = -1,
    DRT_ERR_NOT_CONFIGURED = -2,   /* medium / emitter / sensor missing */
    DRT_ERR_HIP = -3,              /* a HIP runtime call failed */
    DRT_ERR_NO_DEVICE = -4,
    DRT_ERR_UNSUPPORTED = -5
} drt_status;

/* Integrator properties.  Replaces mi.Properties read in
 * VolpathSimpleIntegrator.__init__ (volpathsimple.py:19-36) plus max_depth /
 * rr_depth of the RBIntegrator base (used at :118,200). */
typedef struct drt_config {
    int32_t hide_emitters;        /* default 0 */
    int32_t use_nee;              /* default 1 */
    int32_t use_drt;              /* default 1 */
    int32_t use_drt_subsampling;  /* default 1 */
    int32_t use_drt_mis;          /* default 1 */
    int32_t max_depth;
    int32_t rr_depth;             /* IntegratorConfig.create sets max_depth + 1000 (opt_config.py:105-106) */
} drt_config;

/* Event counters for the algorithmic-bytes roofline. */
typedef struct drt_counters {
    uint64_t n_rays;
    uint64_t n_dt;      /* sigma_t lookups, delta tracking (volpathsimple.py:348,375) */
    uint64_t n_rt;      /* sigma_t lookups, ratio tracking (:469) */
    uint64_t n_drt;     /* sigma_t lookups, sample_interaction_drt (:550,554) */
    uint64_t n_alb;     /* albedo lookups (:141,578) */
    uint64_t n_tr;      /* transmittance-resampling splats (:594-607) */
    uint64_t n_rt_adj;  /* ratio-tracking adjoint splats (:487-492) */
    uint64_t n_sc;      /* sigma_t scattering splats (:170,580) */
    uint64_t n_sc_alb;  /* albedo scattering splats (:170,580) */
} drt_counters;

/* mi.load_dict({'type': 'volpathsimple', ...}) / mi.register_integrator factory
 * (volpathsimple.py:769, opt_config.py:97-108). */
int drt_create(const drt_config *cfg, int device, drt_handle *out);
int drt_destroy(drt_handle h);
/* Python exceptions / asserts of the reference (opt_config.py:98-104, util.py:83-85). */
const char *drt_last_error(drt_handle h);

/* Library-owned scratch grows with the largest job seen (splat record streams: ~2 KB per ray of a backward
 * sub-batch, capped by what the device has free; path cache: 0.5 KB per ray) and is kept for reuse.  This call
 * synchronises the stream and returns it to the device (it is re-allocated on demand); results never depend
 * on it.  drt_destroy frees it too. */
int drt_release_scratch(drt_handle h);

/* Stream on which all later calls enqueue work (hipStream_t, NULL = default). */
int drt_set_stream(drt_handle h, void *hip_stream);
int drt_synchronize(drt_handle h);

/* Image-tile sharding across GPUs (no counterpart in the single-GPU reference;
 * SURVEY.md 8e): local ray i of later render calls has the global index
 * ray_offset + (i / chunk_rays) * stride_rays + (i % chunk_rays).  chunk_rays = 0
 * restores the contiguous mapping ray_offset + i.  chunk_rays should be a
 * multiple of spp so that a pixel's samples stay on one rank. */
int drt_set_ray_interleave(drt_handle h, uint64_t chunk_rays, uint64_t stride_rays);

/* The single medium of the scene: util.get_single_medium (python/util.py:75-86) +
 * the `heterogeneous` medium / `gridvolume` parameters of the fixture
 * (tests/test_integrators.py:79-111).  sigma_t: (Z,Y,X,1); albedo: (Z,Y,X,3);
 * res = {X,Y,Z}; scale finite and >= 0, the box finite and not empty.  Pointers are borrowed until the next drt_set_medium.
 * Also (re)computes the majorant on device. */
int drt_set_medium(drt_handle h, const float *sigma_t, const float *albedo, const int32_t res[3],
                   const float bbox_min[3], const float bbox_max[3], float scale,
                   int32_t majorant_resolution_factor);
/* The COLOUR grids - `albedo`, and the `emission` grid of the drt_nerf_* / drt_fused_* calls - on their own lattice res = {X,Y,Z}
 * (the same box): Mitsuba's GridVolume::eval interpolates every grid on its own resolution, and the reference's janga-smoke pairs a
 * 264 x 136 x 136 density with 256 x 128 x 128 albedo / emission grids (python/scene_config.py:108-110).  The colour gradient buffers of
 * the backward calls then have that shape.  NULL or {0,0,0}: sigma_t's lattice (what drt_set_medium leaves; call this after it).  Scenes
 * whose lattices differ run the kernels of csrc/drt_own.hip: correct (parity: tests/test_gpu_lattice.py), not the tuned path. */
int drt_set_colour_resolution(drt_handle h, const int32_t res[3]);
/* params.update(opt) after an optimizer step (python/optimize.py:354) and
 * medium.set_majorant_resolution_factor (:195-199): refresh the majorant from
 * the (same) parameter buffers.  No host synchronisation. */
int drt_params_changed(drt_handle h);

/* `constant` emitter (tests/test_integrators.py:73-77); the integrator only
 * supports infinite emitters (volpathsimple.py:16). */
int drt_set_emitter_constant(drt_handle h, const float radiance[3]);

/* `envmap` emitter (python/scene_config.py:102,152,210,262,313; used at
 * volpathsimple.py:273 pdf_direction, :284 eval, :419 sample_emitter_direction).
 * `pixels`: DEVICE pointer to a lat-long RGB bitmap [height][width][3] f32 (row 0 =
 * the +Y pole); the library takes its own copy and builds the importance-sampling
 * tables (synchronises the stream), so the caller's buffer may be released.
 * `to_world`: row-major 3x3 rotation; radiance = bilinear lookup x scale.
 * Local direction (sin phi sin theta, cos theta, -cos phi sin theta) <-> uv =
 * (phi / 2pi, theta / pi).  Replaces a previously set constant emitter and vice
 * versa.  No envmap gradients (volpathsimple.py:283 TODO). */
int drt_set_emitter_envmap(drt_handle h, const float *pixels, int32_t width, int32_t height,
                           const float to_world[9], float scale);

/* `perspective` sensor + box-filter hdrfilm used by mi.render
 * (tests/test_integrators.py:46-67; python/optimize.py:44,129,345). */
int drt_set_sensor_perspective(drt_handle h, const float origin[3], const float left[3],
                               const float up[3], const float dir[3], float tan_x, float tan_y,
                               int32_t width, int32_t height);

/* sample(mode=Primal) over a ray batch: volpathsimple.py:38-290 as called from
 * render_batch_primal (batched.py:163-173) / render_batch_backward step (1)
 * (batched.py:255-264).  Ray i has global index ray_offset + i, pixel
 * (ray_offset + i) / spp and the PCG32 stream tea32(seed, ray_offset + i).
 *   rays_o/rays_d != NULL : batched flow, [n][3] each (batched.py:426-467)
 *   rays_o/rays_d == NULL : mi.render flow, rays generated from the sensor with
 *                           the film position drawn from the ray's own stream.
 * L_out: [n][3]. */
int drt_render_primal(drt_handle h, const float *rays_o, const float *rays_d, uint64_t n_rays,
                      uint64_t ray_offset, uint32_t spp, uint32_t seed, float *L_out);

/* sample(mode=Backward, dL, state_in=L_in): render_batch_backward step (2)
 * (batched.py:309-326).  Same rays / seed as the primal call that produced L_in.
 * ACCUMULATES (+=) into grad_sigma_t (Z,Y,X,1) and grad_albedo (Z,Y,X,3) - the
 * reference's dr.grad(params[k]) after scatter_reduce(Add) (volpathsimple.py:170,489,580,607).
 * When this call directly follows the drt_render_primal call of the same job on this handle (same
 * ray range, seed, spp, interleave and ray buffers, no set_* / params_changed call in between - the
 * H1 sequence of batched.py:255-326), the walks recorded by that primal pass are reused instead
 * of being traced again (path cache; per-ray hashes guard explicit ray buffers that were refilled).
 * Any other order is equally valid and simply traces the paths again. */
int drt_render_backward(drt_handle h, const float *rays_o, const float *rays_d, uint64_t n_rays,
                        uint64_t ray_offset, uint32_t spp, uint32_t seed, const float *dL,
                        const float *L_in, float *grad_sigma_t, float *grad_albedo);

/* NeRFIntegrator properties (python/integrators/nerf.py:30-35; density_noise_std is
 * effectively unsupported in the reference, nerf.py:160-162, and is not exposed). */
typedef struct drt_nerf_config {
    int32_t hide_emitters;      /* default 0 */
    int32_t queries_per_ray;    /* default 128 */
    int32_t jittering_enabled;  /* default 1 */
    int32_t activation_relu;    /* 0 identity (default), 1 relu */
} drt_nerf_config;

/* NeRFIntegrator.sample(mode=Primal / Backward) (python/integrators/nerf.py:47-148): emissive ray
 * marching through the medium set by drt_set_medium (its albedo may be NULL) with the emission
 * grid `emission` (Z,Y,X,3) = medium.get_emission (nerf.py:164).  Ray / seed conventions as for
 * drt_render_*.  The backward call accumulates into grad_sigma_t (Z,Y,X,1) and grad_emission
 * (Z,Y,X,3) (dr.backward_from, nerf.py:122-129). */
int drt_nerf_render_primal(drt_handle h, const drt_nerf_config *cfg, const float *emission, const float *rays_o,
                           const float *rays_d, uint64_t n_rays, uint64_t ray_offset, uint32_t spp, uint32_t seed,
                           float *L_out);
int drt_nerf_render_backward(drt_handle h, const drt_nerf_config *cfg, const float *emission, const float *rays_o,
                             const float *rays_d, uint64_t n_rays, uint64_t ray_offset, uint32_t spp, uint32_t seed,
                             const float *dL, const float *L_in, float *grad_sigma_t, float *grad_emission);

/* BASELINE config 5: the `nerf` march and volpathsimple scattering over ONE set of grids [sigma_t, r, g, b] in one call.
 * The reference's scenes bind ONE asset as the medium's albedo and emission grid (python/scene_config.py:109-110), so the
 * colour grid given to drt_set_medium as `albedo` is both.  Per ray, the pass computes NeRFIntegrator.sample (nerf.py:47-148;
 * `cfg`) and VolpathSimpleIntegrator.sample (volpathsimple.py:38-290; the handle's drt_config) from the same camera ray, each
 * on its own copy of the same PCG32 stream and bit-identical to its stand-alone call; the backward pass accumulates BOTH
 * integrators' gradients into grad_sigma_t (Z,Y,X,1) and grad_rgb (Z,Y,X,3) (albedo gradient + emission gradient: one
 * parameter).  Round 5: two dense passes over the rays instead of one kernel - the nerf march (adjoint of sensor rays:
 * drt_nerf_tile.hip, lookups from an interleaved 16-byte-voxel apron-brick copy of sigma_t + colour, voxel gradients
 * pre-reduced in LDS) and the volpathsimple half through the production tracers of drt_render_* (queued supergrid tracer /
 * wave-cooperative tracer, path cache, deferred records) - any emitter, any kind of majorant.  Ray / seed conventions as
 * for drt_render_*. */
int drt_fused_render_primal(drt_handle h, const drt_nerf_config *cfg, const float *rays_o, const float *rays_d, uint64_t n_rays,
                            uint64_t ray_offset, uint32_t spp, uint32_t seed, float *L_nerf_out, float *L_drt_out);
int drt_fused_render_backward(drt_handle h, const drt_nerf_config *cfg, const float *rays_o, const float *rays_d, uint64_t n_rays,
                              uint64_t ray_offset, uint32_t spp, uint32_t seed, const float *dL_nerf, const float *L_nerf_in,
                              const float *dL_drt, const float *L_drt_in, float *grad_sigma_t, float *grad_rgb);

/* sample_batch_pixels + sample_batch_rays of the batched (ray-centric) render op
 * (python/batched.py:397-467).  `sensors`: DEVICE array of n_sensors x 16 floats {origin[3], left[3],
 * up[3], dir[3], tan_x, tan_y, width, height}.  For every batch entry b a (sensor, pixel) pair is
 * drawn from lane b of the PCG32 wavefront seeded with sub_seed_pixels; ray r = b*spp + j gets its
 * sub-pixel offset from lane r of the wavefront seeded with sub_seed_rays.  Outputs: rays_o / rays_d
 * [batch_size*spp][3], sensor_idx [batch_size] and pixels [batch_size][2] (x, y) (may be NULL). */
int drt_batch_sample_rays(drt_handle h, const float *sensors, int32_t n_sensors, uint32_t batch_size, uint32_t spp,
                          uint32_t sub_seed_pixels, uint32_t sub_seed_rays, float *rays_o, float *rays_d,
                          uint32_t *sensor_idx, uint32_t *pixels);

/* The same for the batch entries [batch_first, batch_first + batch_count) only: the share of one rank when the
 * `batch_size` pixel list is dealt across GPUs (no counterpart in the single-GPU reference).
 * The samplers' lanes stay the GLOBAL entry / ray indices, so the union over ranks equals the unsharded
 * batch bit for bit; outputs are local: rays_o / rays_d [batch_count*spp][3], sensor_idx [batch_count],
 * pixels [batch_count][2].  Trace the rays with ray_offset = batch_first * spp. */
int drt_batch_sample_rays_range(drt_handle h, const float *sensors, int32_t n_sensors, uint32_t batch_first,
                                uint32_t batch_count, uint32_t spp, uint32_t sub_seed_pixels, uint32_t sub_seed_rays,
                                float *rays_o, float *rays_d, uint32_t *sensor_idx, uint32_t *pixels);

/* Box-filter film: image[p] = mean over the pixel's spp samples
 * (block.put + film.develop, batched.py:176-197).  L: [n_pixels*spp][3]. */
int drt_film_develop(drt_handle h, const float *L, uint64_t n_pixels, uint32_t spp, float *image);
/* Its adjoint: dL[i] = grad_image[i / spp] / spp (batched.py:298-306). */
int drt_film_backward(drt_handle h, const float *grad_image, uint64_t n_pixels, uint32_t spp,
                      float *dL);

/* Multi-GPU gradient exchange (no handle: works on any gradient buffer of the current device, on `hip_stream`).
 * mask[b] = 1 if block b (block_floats = 64 | 128 | 256 consecutive floats, buf 16-byte aligned) holds anything but
 * zeros (NaN / inf count), else 0.  The host side (distributed.py) all-reduces the masks (MAX) and then only the
 * blocks that are non-zero on some rank - the reference is single-GPU, this replaces nothing in it. */
int drt_grad_block_mask(void *hip_stream, const float *buf, uint64_t n_blocks, uint32_t block_floats, uint8_t *mask);

/* The blocks of the flat gradient buffer that CAN be non-zero, from sigma_t alone (no handle; multi-GPU: the packing set of the ONE
 * gradient all-reduce per backward, computed before the adjoint pass - distributed.gradient_support; the reference is single-GPU).
 * sigma_t (Z,Y,X,1), res = {X,Y,Z}.  The buffer holds one per-voxel plane of `channels` floats per voxel (the albedo gradient)
 * starting at float `sparse_offset_floats`: a block wholly inside it gets mask 1 iff one of its voxels lies within one step (3x3x3
 * neighbourhood) of a non-zero sigma_t voxel - a scattering vertex has sigma_t(x) > 0 (volpathsimple.py:152-172, 577-581) and a
 * trilinear footprint; every other block gets 1.  bits_scratch: ceil(X / 32) * Y * Z words of device scratch. */
int drt_grad_support_mask(void *hip_stream, const float *sigma_t, const int32_t res[3], uint64_t sparse_offset_floats, uint32_t channels,
                          uint64_t n_blocks, uint32_t block_floats, uint32_t *bits_scratch, uint8_t *mask);

/* Packing of that ONE all-reduce (no handle; distributed._allreduce_flat - the reference is single-GPU).
 * drt_grad_block_positions: pos[b] = rank of block b among the blocks with mask[b] != 0, -1 for the others; *count (device) = their
 * number.  scratch: ceil(n_blocks / 1024) words.
 * drt_grad_pack: one pass over the flat buffer - block b of the set is copied to packed[pos[b] * block_floats ...]; every block outside
 * the set is tested and the number of those that hold anything but zeros is ADDED to *check (the float that rides at the end of the
 * packed buffer: a non-zero sum over the ranks says the set was too small).  drt_grad_unpack: the summed blocks back to their places.
 * flat / packed 16-byte aligned, block_floats = 64 | 128 | 256. */
int drt_grad_block_positions(void *hip_stream, const uint8_t *mask, uint64_t n_blocks, int32_t *pos, int32_t *count, uint32_t *scratch);
int drt_grad_pack(void *hip_stream, const float *flat, const int32_t *pos, uint64_t n_blocks, uint32_t block_floats, float *packed, float *check);
int drt_grad_unpack(void *hip_stream, const float *packed, const int32_t *pos, uint64_t n_blocks, uint32_t block_floats, float *flat);

/* One Adam step on a parameter grid in a single pass (no handle; N2: mi.ad.Adam as python/optimize.py:329,352-354 uses it):
 * m = beta_1 m + (1 - beta_1) g;  v = beta_2 v + (1 - beta_2) g^2;  p -= lr_t m / (sqrt(v) + epsilon), where the caller folds the
 * bias corrections into lr_t = lr sqrt(1 - beta_2^t) / (1 - beta_1^t).  All four buffers: n floats, 16-byte aligned. */
int drt_adam_step(void *hip_stream, float *p, const float *g, float *m, float *v, uint64_t n, double beta_1, double beta_2,
                  double epsilon, double lr_t);
/* The same with the parameter's valid range applied to the updated value in the same pass: opt.step() followed by
 * enforce_valid_params (python/optimize.py:169-179, 352-353: sigma_t >= 0, albedo in [0, 1]) - torch.clamp's semantics (a NaN
 * stays a NaN); lo = -inf / hi = +inf: that side is open. */
int drt_adam_step_clamped(void *hip_stream, float *p, const float *g, float *m, float *v, uint64_t n, double beta_1, double beta_2,
                          double epsilon, double lr_t, float lo, float hi);

/* Event counting (off by default; enabling selects a counting build of the kernels). */
int drt_enable_counters(drt_handle h, int enable);
int drt_reset_counters(drt_handle h);
int drt_get_counters(drt_handle h, drt_counters *out);   /* synchronises the stream */
/* The nerf adjoint of sensor rays (csrc/drt_nerf_tile.hip) is bound by the LDS atomic rate, not by HBM: with counting enabled, its last launch's
 * LDS lane-adds (8 per non-zero plane of a query's splat, after the zero skips) - the numerator of that kernel's roofline in bench.py.
 * Synchronises the handle's streams.  0 when no such launch ran with counting enabled. */
int drt_nerf_tile_stats(drt_handle h, uint64_t *lds_lane_adds);

/* Kernel timing for the roofline leg of bench.py: while enabled, every tracing
 * launch is bracketed by a HIP event pair recorded on the handle's stream.
 * drt_enable_timing (either value) synchronises and discards recorded pairs.
 * drt_read_timings synchronises, writes up to `capacity` durations (ms, launch
 * order) of the primal (backward = 0) or adjoint (backward = 1) tracing launches
 * or of the gradient reductions that follow each adjoint launch (backward = 2; on
 * a side stream when sub-batches are pipelined) or of whole backward passes,
 * first launch to last reduction (backward = 3), and returns the number recorded
 * (>= 0) or a negative drt_status. */
int drt_enable_timing(drt_handle h, int enable);
int drt_read_timings(drt_handle h, int backward, float *out_ms, int capacity);

/* Test hook: evaluate one device primitive per item (6 floats in, 6 floats out) so
 * that the parity tests can compare the [M3-ext] building blocks bit for bit with
 * the oracle.  op: 0 log, 1 sincos(2 pi u), 2 square_to_uniform_sphere, 3 sigma_t(p),
 * 4 albedo(p), 5 box hit (o,d) -> valid,t,n, 6 PCG32 floats of (seed,index) bit
 * patterns, 7 sensor ray (pixel bits, ux, uy), 8 mis_weight / div / sqrt / fma,
 * 9 majorant supergrid cell (index bits), 10 exp, 11 atan2(y, x), 12 envmap eval(d)
 * rgb + pdf_direction(d), 13 envmap sample_direction(u1, u2) -> d, pdf, 14 Medium::sample_interaction_drt
 * (E2) from o along d to the box exit with the stream PCG32(tea32(0x5eed, item)) -> valid, t', W, maxt. */
int drt_debug_eval(drt_handle h, int op, const float *in, uint64_t n, float *out);

/* Profiling ablations / kernel selection for experiments; 0 in production.
 * bit 0 (1): skip the gradient atomics; bit 1 (2): per-lane (uncoalesced) atomics; bit 3 (8): force the
 * one-ray-per-lane tracing kernels; bit 4 (16): no empty-space bitmask; bit 5 (32): state-machine
 * kernel for the adjoint too; bit 7 (128): gradient splats as atomics into the apron scratch (the path
 * used when the grid has more than 4096 tiles or the record streams exceed the memory budget) instead
 * of deferred records; bit 8 (256): two-chunk record streams (exercises the out-of-chunks fallback);
 * bit 9 (512): the nerf adjoint of sensor rays through the record path (nerf_kernel + deferred splatting, as explicit ray batches go)
 * instead of the LDS-window kernel drt_nerf_tile.hip; bit 10 (1024): reduction without the flush (timing only); bit 11 (2048):
 * overlap the tracer of ray sub-batch b with the reduction of sub-batch b - 1 on a side stream (also
 * DRT_PIPELINE in the environment; measured slower); bit 13
 * (8192): exact checksum of the flushed sums; bit 14 (16384): 8 MB record budget, i.e. many ray
 * sub-batches (test hook); bit 15 (32768): plain one-ray-per-lane adjoint kernel instead of the
 * wave-cooperative tracking loops (drt_coop.hip); bit 16 (65536): state-machine kernel for the primal
 * (default: the cooperative kernel, which also writes the path cache); bit 20 (1048576): no path
 * cache (the adjoint pass walks its primal path again); bit 18 (262144): pretend that the record
 * streams cannot be allocated (the job then takes the atomic path, as it does when hipMalloc fails); bit 19
 * (524288): pretend that they cannot be (re)allocated from the second ray sub-batch on (the remaining rays take
 * the atomic path); bit 21 (2097152): generic tracing kernels instead of the ones specialised for the registered
 * `volpathsimple-drt` estimator; bit 25 (33554432): no workgroup hand-off of sparse waves' paths (every wave runs
 * its own paths to the end); bit 26 (67108864): none in the primal pass only; bit 27 (134217728): supergrid scenes in
 * the older kernels instead of drt_super.hip; bit 28 (268435456): no early histogram pass beside the adjoint's tail launch (queued tracer: no tail pool);
 * bit 29 (536870912): the supergrid tracer takes its rays in index order (production: thick pixels first); bit 30
 * (1073741824): launches of fewer than 1.5 M rays are scheduled like large ones (ray order, tail launch) - the small scenes
 * of the tests then cover those schedules; bit 31 (2147483648): the queued supergrid tracer walks every flight (production: the
 * primary-segment flights of pixels whose rays cross only empty supergrid cells are ended at their set-up). */
int drt_set_debug_flags(drt_handle h, uint32_t flags);

const char *drt_version(void);

#ifdef __cplusplus
}
#endif
#endif
