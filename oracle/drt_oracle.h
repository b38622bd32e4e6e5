/*
 * drt_oracle.h -- CPU ORACLE (test infrastructure, NOT the product).
 *
 * Scalar, per-ray restatement in plain C of the reference's differential
 * ratio tracking integrator:
 *     /root/reference/python/integrators/volpathsimple.py   (whole file)
 *     /root/reference/python/batched.py:212-326             (primal -> dL -> adjoint sequence)
 * plus the slice of the Mitsuba 3 branch `unbiased-inverse-volume-rendering`
 * that those files call (Medium::sample_interaction[_drt], GridVolume::eval,
 * PCG32 `independent` sampler, sample_tea_32, constant / envmap emitter, isotropic
 * phase, AABB fast-path intersection).
 *
 * PARITY UNPINNED: Mitsuba 3 / Dr.Jit are third-party dependencies that are
 * absent from /root/reference and cannot be imported in the authoring
 * container (no wheel, no network, branch pinned by name only in
 * README.md:97-104); the reference tests store no golden vectors
 * (tests/test_integrators.py:222-347 compare two live Mitsuba runs).
 * The Mitsuba-side semantics are therefore restated from the call sites
 * and public upstream behaviour; the oracle is pinned by analytic
 * known-answer tests and finite differences (tests/test_oracle_*.py), not
 * by reference outputs.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library.  The product (unbiased-inverse-volume-rendering_amd/)
 * never links, imports or calls it.
 */
#ifndef DRT_ORACLE_H
#define DRT_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Integrator properties: volpathsimple.py:19-36 (+ max_depth / rr_depth of the
 * RBIntegrator base, used at :118,200). */
typedef struct drto_config {
    int32_t hide_emitters;
    int32_t use_nee;
    int32_t use_drt;
    int32_t use_drt_subsampling;
    int32_t use_drt_mis;
    int32_t max_depth;
    int32_t rr_depth;
} drto_config;

/* Heterogeneous medium inside an axis-aligned box (tests/test_integrators.py:79-111).
 * Grids are Mitsuba VolumeGrid tensors, shape (Z,Y,X,C), x fastest. */
typedef struct drto_medium {
    const float *sigma_t;   /* (Z,Y,X,1) */
    const float *albedo;    /* (Z,Y,X,3) */
    int32_t res[3];         /* X, Y, Z */
    float bbox_min[3];
    float bbox_max[3];
    float scale;            /* medium `scale` (density_scale) */
    int32_t majorant_factor; /* majorant_resolution_factor (scene_config.py:36); 0 = global majorant */
    /* The colour grids - `albedo`, and the `emission` grid of drto_nerf_render - on their OWN lattice (X, Y, Z), as Mitsuba
     * interpolates every GridVolume on its own resolution: the reference's janga-smoke pairs a 264 x 136 x 136 density with
     * 256 x 128 x 128 albedo / emission grids (python/scene_config.py:108-110).  All zero: the lattice of sigma_t (`res`). */
    int32_t res_colour[3];
} drto_medium;

/* The scene's single infinite emitter (volpathsimple.py:16).
 *   pixels == NULL : `constant` emitter of `radiance` (tests/test_integrators.py:73-77).
 *   pixels != NULL : `envmap` emitter (scene_config.py:102,152,210,262,313) [M3-ext]: lat-long RGB
 *                    bitmap [height][width][3] (row 0 = +Y pole), times `scale`, rotated by the 3x3
 *                    row-major `to_world`.  Local direction (sin phi sin theta, cos theta,
 *                    -cos phi sin theta) <-> uv = (phi / 2pi, theta / pi), bilinear lookup (wrap in u,
 *                    clamp in v).  Importance sampling: piecewise-constant over texels, weight =
 *                    max luminance of the 3x3 neighbourhood x sin(theta_row) (the build's own
 *                    marginal/conditional CDF; Mitsuba's Hierarchical2D warp is not restated -
 *                    same estimator expectation, different sample placement). */
typedef struct drto_emitter {
    float radiance[3];
    const float *pixels;
    int32_t width, height;
    float to_world[9];
    float scale;
} drto_emitter;

/* Perspective sensor after look_at: world-space orthonormal frame.
 * left = normalize(cross(up, dir)), up' = cross(dir, left). */
typedef struct drto_sensor {
    float origin[3];
    float left[3];
    float up[3];
    float dir[3];
    float tan_x;            /* tan(fov_x / 2) */
    float tan_y;            /* tan_x * height / width */
    int32_t width, height;
} drto_sensor;

/* Event counters: one unit = one trilinear lookup or splat. */
typedef struct drto_counters {
    uint64_t n_rays;
    uint64_t n_dt;       /* sigma_t lookups in delta tracking (A4) incl. attached re-eval */
    uint64_t n_rt;       /* sigma_t lookups in ratio tracking (A8), all runs */
    uint64_t n_drt;      /* sigma_t lookups in sample_interaction_drt (E2) incl. re-eval */
    uint64_t n_alb;      /* albedo lookups (A5, A9) */
    uint64_t n_tr;       /* transmittance-resampling splats (A6) */
    uint64_t n_rt_adj;   /* ratio-tracking adjoint splats (A8 second run) */
    uint64_t n_sc;       /* sigma_t scattering-gradient splats (A5 + A9) */
    uint64_t n_sc_alb;   /* albedo scattering-gradient splats (A5 + A9) */
} drto_counters;

/* A render job.  Ray i (global index = ray_offset + i) uses the primary PCG32
 * stream seeded with tea32(seed, ray_offset + i).
 *   sensor != NULL : mi.render flow - pixel = index / spp, film position drawn
 *                    from the ray's own stream (2 draws) before sample().
 *   sensor == NULL : batched flow (batched.py:426-467) - rays_o/rays_d given. */
typedef struct drto_job {
    const drto_config  *cfg;
    const drto_medium  *medium;
    const drto_emitter *emitter;
    const drto_sensor  *sensor;
    const float *rays_o;    /* [n][3] or NULL */
    const float *rays_d;    /* [n][3] or NULL */
    uint64_t n_rays;        /* rays in this job (shard) */
    uint64_t ray_offset;    /* global index of the first ray */
    uint32_t spp;
    uint32_t seed;
    int32_t  n_threads;     /* OpenMP threads (0 = default) */
    int32_t  grad_cache_log2; /* > 0: per-thread write-combining cache of 2^n voxels in front of the shared gradient
                               * grids; -1: tile-binned accumulation (per-thread record buckets by z layer, reduced without
                               * atomics) - both for the timed CPU-baseline leg, both change the fp64 summation order only;
                               * 0: atomic adds into the shared grids */
} drto_job;

/* sample(Primal): L_out[n][3].  volpathsimple.py:38-290 */
int drto_render_primal(const drto_job *job, float *L_out, drto_counters *cnt);

/* sample(Backward) given dL[n][3] and state_in = L_in[n][3] (batched.py:309-318).
 * Accumulates (+=) into grad_sigma_t (Z,Y,X,1) and grad_albedo (Z,Y,X,3). */
int drto_render_backward(const drto_job *job, const float *dL, const float *L_in,
                         double *grad_sigma_t, double *grad_albedo, drto_counters *cnt);

/* Whole H1 step (batched.py:255-326) with the test loss mean((img-0.5)^2)
 * (tests/test_integrators.py:119): primal -> image -> dL -> adjoint.
 * image_out[n/spp][3] and L_scratch[n][3] are caller-allocated. */
int drto_h1_step(const drto_job *job, float *L_scratch, float *image_out, double *loss_out,
                 double *grad_sigma_t, double *grad_albedo, drto_counters *cnt);

/* NeRFIntegrator properties (python/integrators/nerf.py:30-35). */
typedef struct drto_nerf_config {
    int32_t hide_emitters;
    int32_t queries_per_ray;
    int32_t jittering_enabled;
    int32_t activation_relu;     /* 0 identity, 1 relu (nerf.py:38-44) */
} drto_nerf_config;

/* NeRFIntegrator.sample (nerf.py:47-148) over a job: adjoint = 0 writes L_out; adjoint = 1 takes
 * dL / L_in and accumulates into grad_sigma_t (Z,Y,X,1) and grad_emission (Z,Y,X,3).
 * `emission` is the (Z,Y,X,3) grid behind medium.get_emission (nerf.py:164). */
int drto_nerf_render(const drto_job *job, const drto_nerf_config *ncfg, const float *emission, int adjoint,
                     const float *dL, const float *L_in, float *L_out, double *grad_sigma_t,
                     double *grad_emission, drto_counters *cnt);

/* Independent textbook delta-tracking path tracer (no NEE, no MIS, own RNG use):
 * plays the role of Mitsuba's builtin `volpath` in tests/test_integrators.py:222-257. */
int drto_render_textbook(const drto_job *job, float *L_out);

/* N1: batched (ray-centric) pixel / ray sampling, python/batched.py:397-467. */
void drto_batch_sample_rays(const drto_sensor *sensors, int n_sensors, uint32_t batch_size, uint32_t spp,
                            uint32_t sub_seed_pixels, uint32_t sub_seed_rays, float *rays_o, float *rays_d,
                            uint32_t *sensor_idx, uint32_t *pixels);

/* --- test hooks on the primitives ---------------------------------------- */
uint32_t drto_tea32(uint32_t v0, uint32_t v1, uint32_t *out_v1);
void     drto_pcg32_floats(uint32_t seed, uint32_t index, int n, float *out);
void     drto_pcg32_raw(uint64_t initstate, uint64_t initseq, int n, uint32_t *out);
void     drto_uniform_sphere(float ux, float uy, float out[3]);
float    drto_logf(float x);
float    drto_expf(float x);
void     drto_sincos_2pi(float u, float *s, float *c);
float    drto_atan2f(float y, float x);
/* envmap primitives (test hooks): radiance towards world direction d; solid-angle pdf of sampling d;
 * sample_direction(u1, u2) -> d, pdf, radiance / pdf; the importance-sampling tables
 * (marginal [h+1], conditional [h][w+1]; either may be NULL). */
int      drto_envmap_eval(const drto_emitter *e, const float d[3], float out[3]);
float    drto_envmap_pdf(const drto_emitter *e, const float d[3]);
int      drto_envmap_sample(const drto_emitter *e, float u1, float u2, float d[3], float *pdf, float weight[3]);
int      drto_envmap_tables(const drto_emitter *e, float *marginal, float *conditional);
float    drto_eval_sigma_t(const drto_medium *m, const float p[3]);
void     drto_eval_albedo(const drto_medium *m, const float p[3], float out[3]);
float    drto_majorant(const drto_medium *m);
/* supergrid: writes gx*gy*gz cell majorants (x fastest) into out (may be NULL) and returns the
 * cell count; dims[3] receives (gx,gy,gz).  0 cells when majorant_factor == 0. */
int      drto_majorant_grid(const drto_medium *m, int32_t dims[3], float *out);
/* mean ratio-tracking transmittance estimate over n independent walks (A8). */
double   drto_ratio_tracking_mean(const drto_medium *m, const float o[3], const float d[3],
                                  float tmax, uint32_t seed, int n);
/* E2 (Medium::sample_interaction_drt, call site volpathsimple.py:549-551): n independent walks from o along
 * d to the box exit, walk i on the stream PCG32(tea32(seed, first + i)); per walk valid / t' / W; returns maxt. */
float    drto_sample_interaction_drt(const drto_medium *m, const float o[3], const float d[3], uint32_t seed,
                                     uint32_t first, int n, int32_t *valid, float *t_out, float *W_out);
/* returns 1 and fills t / normal if the ray hits the medium box surface (E4). */
int      drto_box_hit(const drto_medium *m, const float o[3], const float d[3],
                      float *t, float n[3]);
void     drto_sensor_ray(const drto_sensor *s, uint32_t pixel, float ux, float uy,
                         float o[3], float d[3]);
/* alt-sampler seed derived from lane 0's draw (volpathsimple.py:99-107). */
uint32_t drto_alt_seed(uint32_t seed, int sensor_flow);

#ifdef __cplusplus
}
#endif
#endif
