// drt_device.h -- device-side primitives of the gfx950 DRT integrator.
//
// Everything here follows the arithmetic specification in DESIGN.md: each fp32
// operation is individually rounded (the library is compiled with
// -ffp-contract=off), fused multiply-adds appear only as explicit fmaf(), and the
// elementary functions (log, sincos) are fixed polynomials.  This is what makes a
// ray's primal radiance independent of how rays are scheduled onto wavefronts.
//
// Reference semantics being implemented (python/integrators/volpathsimple.py and
// the Mitsuba 3 calls it makes) are cited per function.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace drt {

struct V3 { float x, y, z; };

__device__ __forceinline__ V3 v3(float x, float y, float z) { return V3{ x, y, z }; }
// Ray3f::operator()(t) = fmadd(d, t, o)
__device__ __forceinline__ V3 ray_at(V3 o, V3 d, float t)
{
    return V3{ fmaf(d.x, t, o.x), fmaf(d.y, t, o.y), fmaf(d.z, t, o.z) };
}

// Test hooks (kernel-variant selection, ablations, simulated out-of-memory; include/drt_hip.h: drt_set_debug_flags)
// exist only in the build with -DDRT_TEST_HOOKS (libdrt_hip_hooks.so, used by the variant tests and profiling
// ablations); in the production library every test compiles to `false` and drt_set_debug_flags rejects non-zero flags.
#ifdef DRT_TEST_HOOKS
constexpr bool kTestHooks = true;
#else
constexpr bool kTestHooks = false;
#endif
__host__ __device__ __forceinline__ constexpr bool dbg(uint32_t flags, uint32_t bits) { return kTestHooks && (flags & bits) != 0u; }

// Experiment switches (timing / profiling builds made by tools/mk_variant.sh; several of them give WRONG results on purpose)
// must never leak into a shipped library: they only compile with -DDRT_EXPERIMENT_BUILD.
#if !defined(DRT_EXPERIMENT_BUILD) && (defined(DRT_FAST_MATH) || defined(DRT_ENV_EXP) || defined(DRT_EXP_ALB) || defined(DRT_EXP_DROP_TAIL) || defined(DRT_NT_STATS) || \
                                       (defined(DRT_SQ_PROFILE) && DRT_SQ_PROFILE != 0))
#error "experiment switch without -DDRT_EXPERIMENT_BUILD (tools/mk_variant.sh adds it): not for a shipped library"
#endif
constexpr float kInvFourPi = 0.07957747154594767f;
constexpr float kFourPi    = 12.566370614359172f;
constexpr float kHalfPi    = 1.5707963267948966f;
constexpr float kLargest   = 3.4028234663852886e38f;                 // dr.largest(Float)
constexpr float kRayEps    = 1500.0f * 5.9604644775390625e-8f;       // math::RayEpsilon<float>
constexpr float kInf       = __builtin_huge_valf();

// ---------------------------------------------------------------------------
// sample_tea_32 + PCG32 (`independent` sampler); call sites
// volpathsimple.py:71,99,105-107,120,222,348,359,383,418,470,536,595,632
// ---------------------------------------------------------------------------
__host__ __device__ __forceinline__ void tea32(uint32_t v0, uint32_t v1, uint32_t &o0, uint32_t &o1)
{
    uint32_t sum = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        sum += 0x9e3779b9u;
        v0 += ((v1 << 4) + 0xa341316cu) ^ (v1 + sum) ^ ((v1 >> 5) + 0xc8013ea4u);
        v1 += ((v0 << 4) + 0xad90777du) ^ (v0 + sum) ^ ((v0 >> 5) + 0x7e95761eu);
    }
    o0 = v0; o1 = v1;
}

struct Pcg32 {
    uint64_t state, inc;

    __host__ __device__ __forceinline__ uint32_t next_u32()
    {
        uint64_t old = state;
        state = old * 0x5851f42d4c957f2dull + inc;
        uint32_t xs = (uint32_t)(((old >> 18) ^ old) >> 27);
        uint32_t rot = (uint32_t)(old >> 59);
        return (xs >> rot) | (xs << ((0u - rot) & 31u));
    }
    // Sampler::seed(seed, wavefront): lane `index` -> PCG32(initstate, initseq) = tea32(seed, index)
    __host__ __device__ __forceinline__ void seed(uint32_t seed_value, uint32_t index)
    {
        uint32_t v0, v1;
        tea32(seed_value, index, v0, v1);
        state = 0;
        inc = ((uint64_t) v1 << 1) | 1ull;
        next_u32();
        state += (uint64_t) v0;
        next_u32();
    }
    // next_float32: 23 mantissa bits, [0,1)
    __host__ __device__ __forceinline__ float next_1d()
    {
        uint32_t bits = (next_u32() >> 9) | 0x3f800000u;
        float f;
        __builtin_memcpy(&f, &bits, 4);
        return f - 1.0f;
    }
};

// ---------------------------------------------------------------------------
// fixed elementary functions
// ---------------------------------------------------------------------------
__device__ __forceinline__ float drt_logf(float x)
{
#ifdef DRT_FAST_MATH
    // pricing experiment only (what bit-exactness costs): the hardware's v_log_f32.  Never a shipped flavour's default.
    return __builtin_amdgcn_logf(x) * 0.69314718055994530942f;
#endif
    uint32_t ix = __float_as_uint(x);
    int e = (int)(ix >> 23) - 126;
    float m = __uint_as_float((ix & 0x007fffffu) | 0x3f000000u);
    float f;
    if (m < 0.70710678118654752440f) { e -= 1; f = (m + m) - 1.0f; }
    else { f = m - 1.0f; }
    float z = f * f;
    float p = 7.0376836292e-2f;
    p = fmaf(p, f, -1.1514610310e-1f);
    p = fmaf(p, f, 1.1676998740e-1f);
    p = fmaf(p, f, -1.2420140846e-1f);
    p = fmaf(p, f, 1.4249322787e-1f);
    p = fmaf(p, f, -1.6668057665e-1f);
    p = fmaf(p, f, 2.0000714765e-1f);
    p = fmaf(p, f, -2.4999993993e-1f);
    p = fmaf(p, f, 3.3333331174e-1f);
    float y = (f * z) * p;
    float fe = (float) e;
    y = fmaf(-2.12194440e-4f, fe, y);
    y = fmaf(-0.5f, z, y);
    float r = f + y;
    r = fmaf(0.693359375f, fe, r);
    return r;
}

// exp for x <= 0: Cephes expf (ln2 split range reduction, degree-5 polynomial, exact 2^n scale)
__device__ __forceinline__ float drt_expf(float x)
{
    if (x < -87.0f) return 0.0f;
    float z = floorf(fmaf(1.44269504088896341f, x, 0.5f));
    x = fmaf(-0.693359375f, z, x);
    x = fmaf(2.12194440e-4f, z, x);
    int n = (int) z;
    float x2 = x * x;
    float p = 1.9875691500e-4f;
    p = fmaf(p, x, 1.3981999507e-3f);
    p = fmaf(p, x, 8.3334519073e-3f);
    p = fmaf(p, x, 4.1665795894e-2f);
    p = fmaf(p, x, 1.6666665459e-1f);
    p = fmaf(p, x, 5.0000001201e-1f);
    float r = fmaf(p, x2, x) + 1.0f;
    return r * __uint_as_float((uint32_t)(n + 127) << 23);
}

__device__ __forceinline__ void drt_sincos_2pi(float u, float &s_out, float &c_out)
{
    float a = u * 4.0f;
    int q = (int) a;
    float f = a - (float) q;
    bool swap = f > 0.5f;
    float g = swap ? (1.0f - f) : f;
    float x = g * kHalfPi;
    float x2 = x * x;
    float ps = -1.9515295891e-4f;
    ps = fmaf(ps, x2, 8.3321608736e-3f);
    ps = fmaf(ps, x2, -1.6666654611e-1f);
    float s = fmaf(x * x2, ps, x);
    float pc = 2.443315711809948e-5f;
    pc = fmaf(pc, x2, -1.388731625493765e-3f);
    pc = fmaf(pc, x2, 4.166664568298827e-2f);
    float c = fmaf(x2 * x2, pc, fmaf(-0.5f, x2, 1.0f));
    if (swap) { float t = s; s = c; c = t; }
    q &= 3;
    float sq = (q == 0) ? s : (q == 1) ? c : (q == 2) ? -s : -c;
    float cq = (q == 0) ? c : (q == 1) ? -s : (q == 2) ? -c : s;
    s_out = sq; c_out = cq;
}

// warp::square_to_uniform_sphere: isotropic phase (volpathsimple.py:221,630) and
// constant emitter direction sampling (:419)
__device__ __forceinline__ V3 square_to_uniform_sphere(float ux, float uy)
{
    float z = fmaf(-2.0f, uy, 1.0f);
    float r = sqrtf(fmaxf(0.0f, fmaf(-z, z, 1.0f)));
    float s, c;
    drt_sincos_2pi(ux, s, c);
    return V3{ r * c, r * s, z };
}

// atan2 with a specified instruction sequence (Cephes atanf polynomial on min/max in [0,1]);
// atan2(0,0) = 0.  Same sequence as the oracle's, so envmap lookups are bit-identical.
__device__ __forceinline__ float drt_atan2f(float y, float x)
{
    float ax = fabsf(x), ay = fabsf(y);
    float mx = fmaxf(ax, ay), mn = fminf(ax, ay);
    float t = mx != 0.0f ? mn / mx : 0.0f;
    float y0 = 0.0f, z = t;
    if (t > 0.4142135624f) { y0 = 0.78539816339744831f; z = (t - 1.0f) / (t + 1.0f); }
    float zz = z * z;
    float p = fmaf(fmaf(fmaf(8.05374449538e-2f, zz, -1.38776856032e-1f), zz, 1.99777106478e-1f), zz, -3.33329491539e-1f);
    float a = y0 + fmaf(p * zz, z, z);
    if (ay > ax) a = 1.57079632679489662f - a;
    if (x < 0.0f) a = 3.14159265358979323846f - a;
    return y < 0.0f ? -a : a;
}

// mi.ad.common.mis_weight: power heuristic, non-finite -> 0 (volpathsimple.py:278,391)
__device__ __forceinline__ float mis_weight(float a, float b)
{
    float a2 = a * a, b2 = b * b;
    float w = a2 / (a2 + b2);
    return isfinite(w) ? w : 0.0f;
}

// ---------------------------------------------------------------------------
// kernel parameter block (passed by value in the kernarg segment)
// ---------------------------------------------------------------------------
struct Params {
    // medium
    const float *sigma_t;      // (Z,Y,X,1), caller's layout (majorant reduction, re-bricking)
    const float *sigma_b;      // library-owned apron-brick copy read by the tracking loops (eval_sigma_t)
    int sb_ystride, sb_zstride; // bricks per row (nbx) / per z slab (nby*nbx)
    // empty-space bitmask: bit c of occ is 0 iff every voxel a lookup with base corner inside
    // cell c (2^occ_shift voxels per axis) can touch is exactly zero -> the lookup is 0 without a fetch
    const uint32_t *occ;
    int occ_shift, occ_x, occ_y, occ_words;
    const float *albedo;       // (Z,Y,X,3)
    // library-owned interleaved FOUR-CHANNEL apron-brick copy [sigma_t, r, g, b] (16-byte voxels) of sigma_t and the
    // colour grid (albedo = emission, scene_config.py:109-110) for the fused nerf + volpathsimple pass (eval4), or nullptr
    const float4 *grid4;
    int g4_nbx;                // lines per grid row = ceil(rx / 3)
    const float *majorant;     // device: [0] = scale*max(sigma_t), [1] = 1/[0] (0 if [0]==0)
    const float *mgrid;        // majorant supergrid, one majorant per cell (x fastest), or nullptr
    const uint32_t *mocc;      // bit c = supergrid cell c has a non-zero majorant (the DDA skips the others without a load)
    const uint32_t *mocc_dil;  // bit c = cell c or one of its 26 neighbours has (build_unit_empty)
    int mocc_words;
    int gx, gy, gz;
    int rx, ry, rz;
    float bmin[3], bmax[3], inv_ext[3];
    float scale;
    float Le[3];
    // `envmap` emitter (env_pix != nullptr) instead of the constant Le: lat-long RGB bitmap
    // [env_h][env_w][3] (library-owned device copy), row-major to_world rotation, scale, and the
    // importance-sampling tables (marginal CDF over rows [h+1], conditional CDFs [h][w+1])
    const float4 *env_pix;                   // [env_h][env_w] texels {r, g, b, density in uv space} (envmap_taps)
    const float *env_marg, *env_cond;
    const uint32_t *env_gmarg, *env_gcond;   // guide tables of the two CDF families (cdf_find_guided): [h + 1], [h][env_cstride]
    int env_w, env_h, env_cstride;           // env_cstride: floats per row of env_cond / env_gcond (w + 1 rounded up to whole 128-byte lines)
    float env_R[9], env_scale;
    // nerf integrator (python/integrators/nerf.py): emission grid (Z,Y,X,3) and properties
    const float *emission;
    int nerf_queries, nerf_jitter, nerf_relu;
    // 1: this nerf march is one half of the fused nerf + volpathsimple pass (drt_fused_render_*): its rays are counted by the other half
    int nerf_fused_half;
    // integrator flags
    int hide_emitters, use_nee, use_drt, use_drt_subsampling, use_drt_mis, max_depth, rr_depth;
    // sensor (mi.render flow)
    int sensor_flow;
    float cam_o[3], cam_left[3], cam_up[3], cam_dir[3];
    float tan_x, tan_y;
    int width, height;
    // job
    const float *rays_o, *rays_d;
    uint64_t n_rays, ray_offset;   // local rays [ray_first, n_rays) are traced by this launch (sub-batches of one job)
    uint64_t ray_first;
    uint64_t chunk, stride;    // global index of local ray i = ray_offset + (i/chunk)*stride + i%chunk (chunk 0: + i)
    uint32_t spp, seed, alt_seed;
    // outputs / adjoint inputs
    float *L_out;
    const float *dL, *L_in;
    float *g_sigma, *g_albedo;
    // library-owned gradient scratch: 4 planes [sigma_t, r, g, b] in an APRON layout: one 64-byte
    // line per base corner block (3 voxels along x) holds all 8 corners of a splat -> one atomic
    // request per splat and plane (see make_grad_indices); reduced into the caller's grids by
    // untile_gradients_kernel
    float *gt;                 // plane c at gt + c * gt_plane
    uint32_t gt_plane;         // floats per plane = nbx * ry * rz * 16
    int gt_nbx;                // lines per grid row = ceil(rx / 3)
    // Deferred splatting (adjoint, drt_deferred.hip): instead of atomics the tracer appends records to two
    // streams, in chunks of kRecChunk records handed out by rec_cursor[s]; the records are then partitioned
    // by grid tile and reduced in LDS.  rec_buf[0] == nullptr: the atomic path (apron scratch) is used.
    //   stream 0: 16-byte records {p.x, p.y, p.z, g_sigma_t}                     (float4)
    //   stream 1: 32-byte records {p.x, p.y, p.z, g_sigma_t, g_r, g_g, g_b, 0}   (2 x float4): the splats that carry
    //             all four gradient channels at one point - scatter events (sigma_t + albedo, volpathsimple.py:170,580)
    //             and nerf queries (sigma_t + emission, nerf.py:122-129) - one append instead of four
    float4 *rec_buf[2];
    uint32_t *rec_chunk_count[2];   // valid records per chunk (zeroed per launch)
    uint32_t rec_cap_chunks[2];     // chunks available per stream
    uint32_t *rec_cursor;           // [0..1] chunks handed out, [4..5] splats that overflowed (direct atomics)
    // Path cache (drt_coop.hip): the primal pass of an H1 step records, per ray and bounce-loop iteration,
    // what its delta-tracking walk and its NEE transmittance walk returned (distance / transmittance, the
    // sampler state behind them, their step counts); the adjoint pass of the SAME job reads them instead of
    // walking again.  [n_rays][path_cache_cap][2] uint4; mode 0 off, 1 write, 2 read.  ray_hash: one word
    // per ray (explicit rays: hash of origin and direction) that must match for a ray's entries to be used.
    // heavy-first scheduling (drt_coop.hip): the primal pass sums the tracking steps of every 256-ray block into
    // block_cost, block_order (blocks by descending cost) then tells the adjoint launch which block a workgroup takes
    uint32_t *block_cost;
    const uint32_t *block_order;
    // ray -> lane schedule inside every group of kPermGroup rays: the primal pass records how many bounce-loop
    // iterations each ray ran, ray_perm_kernel sorts every group by it (longest first), and the adjoint pass of the
    // same job hands ray ray_perm[slot] of the group to thread `slot`, so that a wave's 64 rays leave the bounce
    // loop together (a schedule only: every ray's result is independent of the lane that traces it)
    uint8_t *ray_iters;            // [rays] written by the cooperative primal kernel
    const uint16_t *ray_perm;      // [rays rounded up to kPermGroup] read by the cooperative adjoint kernel
    uint4 *path_cache;
    uint32_t *ray_hash;
    uint32_t path_cache_cap, path_cache_mode;
    // tail pool of the specialised cooperative kernels (CoopTracer::wg_handoff): the last <= kTailPush live paths of a
    // workgroup are written here (32 words each) and finished by a second, dense launch of the same kernel in tail mode
    uint4 *tail_pool;              // [tail_cap][8] uint4 or nullptr
    uint32_t *tail_count;          // entries written (zeroed before the main launch)
    uint32_t tail_cap, tail_mode;  // capacity in entries; 1 = this launch finishes the pool's paths
    unsigned long long *queues;     // 8 per-XCD ray queue heads (wavefront kernel), zeroed per launch
    // Ray order of the supergrid tracer (drt_super.hip, build_super_order): queue position g takes ray
    // ray_first + order[g / order_unit] * order_unit + g % order_unit - units of consecutive rays, the expensive ones first, so
    // that the longest paths of a launch start at its beginning instead of running on alone at its end (a schedule
    // only: every ray's result is independent of when it is traced).  nullptr: rays in index order
    const uint32_t *order;
    uint32_t order_unit, order_units;
    // queued supergrid tracer: unit_empty[(i - ray_first) / empty_unit] != 0 - every ray of that unit (one pixel's sensor rays) crosses only
    // supergrid cells whose majorant is 0 (build_unit_empty): the flights along its primary segment are not walked.  nullptr: no flags
    const uint8_t *unit_empty;
    uint32_t empty_unit;
    // queued supergrid tracer (drt_sq.hip): per workgroup [3 | 9][sq_rays] uint4 of path state that only the path
    // transitions use (throughput, radiance; adjoint: dL, the sampler clone, the DRT reservoir); library-owned, L2-resident.
    // sq_rays: ray records per workgroup (set by launch_trace_sq: what fits LDS next to the majorants)
    void *sq_cold;
    uint32_t sq_rays;
    uint32_t sq_chunk;              // queue positions a workgroup reserves per refill of its ray pool (set by launch_trace_sq)
    unsigned long long *counters;   // 9 x u64 or nullptr
    uint32_t debug_flags;           // ablation switches for profiling (drt_set_debug_flags); 0 in production
    // The colour grids - `albedo`, the nerf integrator's `emission` - on their OWN lattice (drt_set_colour_resolution): Mitsuba interpolates every
    // GridVolume on its own resolution, and the reference's janga-smoke pairs a 264 x 136 x 136 density with 256 x 128 x 128 albedo / emission grids
    // (python/scene_config.py:108-110).  colour_own = 0: the lattice of sigma_t (crx = rx, ...), every kernel as before.  colour_own = 1: the job
    // runs the kernels of drt_own.hip - the only translation unit compiled with DRT_COLOUR_OWN, in which eval_rgb and the colour splats read
    // crx / cry / crz; in every other unit these fields are never read.  (At the end of the block: no other field moves.)
    int crx, cry, crz, colour_own;
    uint32_t sq_rounds;             // queued tracer, primal launches in index order: the medium is thin as far as the host knows (drt_capi.cpp: majorant read-back) - the ROUNDS kernels
};

// ---------------------------------------------------------------------------
// envmap emitter [M3-ext] (volpathsimple.py:273,284,419; include/drt_hip.h: drt_set_emitter_envmap)
// local direction (sin phi sin theta, cos theta, -cos phi sin theta) <-> uv = (phi/2pi, theta/pi)
// ---------------------------------------------------------------------------
constexpr float kInvTwoPi = 0.15915494309189535f;
constexpr float kInvPi = 0.31830988618379069f;
constexpr float kTwoPiSq = 19.739208802178716f;
constexpr float kOneMinusEps = 0.99999994f;

// The map is ONE table of float4 texels {r, g, b, density}: `density` = the texel's probability density in uv space, pmf(row) * pmf(col | row)
// * w * h - the float product of the table differences, computed once per texel on the host with the very operations the lookup used to
// make (drt_set_emitter_envmap) - so that Emitter::eval (bilinear, :284) and Emitter::pdf_direction (piecewise constant, :273) of ONE direction
// come from the same four taps: the density's texel (floor(u w), floor(v h)) is always one of the bilinear footprint's four (round 4 read the
// radiance from an RGB table and the density from two CDF tables: 16 scalar loads over three tables, and the direction -> uv conversion twice).
struct EnvTaps { float4 t00, t01, t10, t11; float fx, fy; int i0, i1, j0, j1; };

// bilinear footprint at uv in [0,1)^2: texel centres at ((i+.5)/w, (j+.5)/h), wrap in u, clamp in v
__device__ __forceinline__ EnvTaps envmap_taps(const Params &P, float u, float v)
{
    const int w = P.env_w, h = P.env_h;
    float px = fmaf(u, (float) w, -0.5f), py = fmaf(v, (float) h, -0.5f);
    float fx0 = floorf(px), fy0 = floorf(py);
    EnvTaps T;
    T.fx = px - fx0; T.fy = py - fy0;
    int i0 = (int) fx0, j0 = (int) fy0;
    int i1 = i0 + 1, j1 = j0 + 1;
    if (i0 < 0) i0 += w;
    if (i1 >= w) i1 -= w;
    j0 = min(max(j0, 0), h - 1);
    j1 = min(max(j1, 0), h - 1);
    T.i0 = i0; T.i1 = i1; T.j0 = j0; T.j1 = j1;
    T.t00 = P.env_pix[(size_t) j0 * w + i0]; T.t01 = P.env_pix[(size_t) j0 * w + i1];
    T.t10 = P.env_pix[(size_t) j1 * w + i0]; T.t11 = P.env_pix[(size_t) j1 * w + i1];
    return T;
}

__device__ __forceinline__ void envmap_bilinear(const Params &P, const EnvTaps &T, float out[3])
{
    const float wx0 = 1.0f - T.fx, wy0 = 1.0f - T.fy, fx = T.fx, fy = T.fy;
    const float p00[3] = { T.t00.x, T.t00.y, T.t00.z }, p01[3] = { T.t01.x, T.t01.y, T.t01.z };
    const float p10[3] = { T.t10.x, T.t10.y, T.t10.z }, p11[3] = { T.t11.x, T.t11.y, T.t11.z };
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        float a = fmaf(wx0, p00[k], fx * p01[k]);
        float b = fmaf(wx0, p10[k], fx * p11[k]);
        out[k] = fmaf(wy0, a, fy * b) * P.env_scale;
    }
}

__device__ __forceinline__ void envmap_dir_to_uv(const Params &P, V3 d, float &u, float &v, float &sin_theta)
{
    const float *R = P.env_R;                                // local = R^T d
    float lx = fmaf(R[6], d.z, fmaf(R[3], d.y, R[0] * d.x));
    float ly = fmaf(R[7], d.z, fmaf(R[4], d.y, R[1] * d.x));
    float lz = fmaf(R[8], d.z, fmaf(R[5], d.y, R[2] * d.x));
    float st = sqrtf(fmaf(lx, lx, lz * lz));
    float uu = drt_atan2f(lx, -lz) * kInvTwoPi;
    if (uu < 0.0f) uu += 1.0f;
    if (uu >= 1.0f) uu = 0.0f;
    float vv = drt_atan2f(st, ly) * kInvPi;
    vv = fminf(fmaxf(vv, 0.0f), kOneMinusEps);
    u = uu; v = vv; sin_theta = st;
}

// Emitter::eval for an escaped ray of direction d (volpathsimple.py:284)
__device__ __forceinline__ void envmap_eval(const Params &P, V3 d, float out[3])
{
#if defined(DRT_ENV_EXP) && (DRT_ENV_EXP & 4)
    out[0] = out[1] = out[2] = 0.5f + d.x * 0.1f; return;    // timing experiment only
#endif
    float u, v, st;
    envmap_dir_to_uv(P, d, u, v, st);
    const EnvTaps T = envmap_taps(P, u, v);
    envmap_bilinear(P, T, out);
}

// Emitter::pdf_direction (volpathsimple.py:273): the density texel alone (one 16-byte load)
__device__ __forceinline__ float envmap_pdf(const Params &P, V3 d)
{
#if defined(DRT_ENV_EXP) && (DRT_ENV_EXP & 2)
    return kInvFourPi + d.y * 1e-3f;                          // timing experiment only
#endif
    float u, v, st;
    envmap_dir_to_uv(P, d, u, v, st);
    int i = min((int)(u * (float) P.env_w), P.env_w - 1), j = min((int)(v * (float) P.env_h), P.env_h - 1);
    float den = kTwoPiSq * st;
    return den > 0.0f ? P.env_pix[(size_t) j * P.env_w + i].w / den : 0.0f;
}

// both of ONE direction: one direction -> uv conversion, the four taps of the bilinear footprint (the density's texel is one of them)
__device__ __forceinline__ float envmap_eval_pdf(const Params &P, V3 d, float out[3])
{
    float u, v, st;
    envmap_dir_to_uv(P, d, u, v, st);
    const EnvTaps T = envmap_taps(P, u, v);
    envmap_bilinear(P, T, out);
    const int i = min((int)(u * (float) P.env_w), P.env_w - 1), j = min((int)(v * (float) P.env_h), P.env_h - 1);
    float dens;
    if ((i == T.i0 || i == T.i1) && (j == T.j0 || j == T.j1))
        dens = i == T.i0 ? (j == T.j0 ? T.t00.w : T.t10.w) : (j == T.j0 ? T.t01.w : T.t11.w);
    else dens = P.env_pix[(size_t) j * P.env_w + i].w;        // (cannot happen: floor(u w) lies in the footprint; kept for safety)
    const float den = kTwoPiSq * st;
    return den > 0.0f ? dens / den : 0.0f;
}

// largest k in [0, n-1] with cdf[k] <= x (cdf[0] = 0, cdf[n] = 1, x in [0,1))
__device__ __forceinline__ int cdf_find(const float *cdf, int n, float x)
{
    int lo = 0, hi = n;
    while (hi - lo > 1) {
        int mid = (lo + hi) >> 1;
        if (cdf[mid] <= x) lo = mid; else hi = mid;
    }
    return lo;
}

// The same index through a guide table: guide[k] = largest index with cdf <= k / n, so the answer for x in
// [(k - 1) / n, (k + 1) / n) lies in [guide[k - 1], guide[k + 1]] - a bracket of two or three entries where the density is
// high, i.e. where the samples fall - and the bisection starts there instead of at [0, n]: ~4 dependent loads instead of
// log2(n) + 1 = 9 / 10 per CDF (the bracket is one interval wider on both sides than x * n says: its rounding cannot matter).
__device__ __forceinline__ int cdf_find_guided(const float *cdf, const uint32_t *guide, int n, float x)
{
    const int k = min((int) (x * (float) n), n - 1);
    int lo = (int) guide[max(k - 1, 0)], hi = min((int) guide[min(k + 1, n)] + 1, n);
    if (hi - lo <= 4) {
        // the usual bracket (two to four entries where the samples fall): its inner entries in ONE round of independent loads instead
        // of two dependent bisection steps - the largest index of a non-decreasing table with cdf <= x is lo + (entries <= x behind lo)
        const float c1 = cdf[min(lo + 1, n)], c2 = cdf[min(lo + 2, n)], c3 = cdf[min(lo + 3, n)];
        return lo + ((lo + 1 < hi && c1 <= x) ? 1 : 0) + ((lo + 2 < hi && c2 <= x) ? 1 : 0) + ((lo + 3 < hi && c3 <= x) ? 1 : 0);
    }
    while (hi - lo > 1) {
        int mid = (lo + hi) >> 1;
        if (cdf[mid] <= x) lo = mid; else hi = mid;
    }
    return lo;
}

// Scene::sample_emitter_direction (volpathsimple.py:419).  The pdf and the radiance are evaluated
// FROM THE DIRECTION (envmap_pdf / envmap_eval, as for an escaped ray), so NEE and the escape-side
// MIS weight see the same density and the state machine only has to keep the direction.
__device__ __forceinline__ V3 envmap_sample_dir(const Params &P, float u1, float u2)
{
#if defined(DRT_ENV_EXP) && (DRT_ENV_EXP & 1)
    return square_to_uniform_sphere(u1, u2);                  // timing experiment only
#endif
    const int w = P.env_w, h = P.env_h;
    int j = cdf_find_guided(P.env_marg, P.env_gmarg, h, u2);
    const float *c = P.env_cond + (size_t) j * P.env_cstride;          // (rows padded to whole 128-byte lines)
    int i = cdf_find_guided(c, P.env_gcond + (size_t) j * P.env_cstride, w, u1);
    float dv = fminf((u2 - P.env_marg[j]) / (P.env_marg[j + 1] - P.env_marg[j]), kOneMinusEps);
    float du = fminf((u1 - c[i]) / (c[i + 1] - c[i]), kOneMinusEps);
    float u = ((float) i + du) / (float) w, v = ((float) j + dv) / (float) h;
    float sp, cp, st, ct;
    drt_sincos_2pi(u, sp, cp);
    drt_sincos_2pi(0.5f * v, st, ct);
    float lx = sp * st, ly = ct, lz = -(cp * st);
    const float *R = P.env_R;                                // world = R local
    V3 d;
    d.x = fmaf(R[2], lz, fmaf(R[1], ly, R[0] * lx));
    d.y = fmaf(R[5], lz, fmaf(R[4], ly, R[3] * lx));
    d.z = fmaf(R[8], lz, fmaf(R[7], ly, R[6] * lx));
    return d;
}

// The emitter dispatch is a compile-time switch (ENV = an envmap is bound): the constant-emitter
// kernels - every BASELINE configuration - carry none of the envmap code or its registers.
// emitter direction sample: the direction for (u1, u2) ...
template <bool ENV>
__device__ __forceinline__ V3 emitter_sample_dir(const Params &P, float u1, float u2)
{
    if constexpr (ENV) return envmap_sample_dir(P, u1, u2);
    else return square_to_uniform_sphere(u1, u2);
}

// ... and, from the direction, ds.pdf and radiance / pdf
template <bool ENV>
__device__ __forceinline__ float emitter_sample_value(const Params &P, V3 d, float val[3])
{
    if constexpr (ENV) {
        float Le[3];
        const float p = envmap_eval_pdf(P, d, Le);
#pragma unroll
        for (int k = 0; k < 3; ++k) val[k] = p > 0.0f ? Le[k] / p : 0.0f;
        return p;
    } else {
#pragma unroll
        for (int k = 0; k < 3; ++k) val[k] = P.Le[k] * kFourPi;
        return kInvFourPi;
    }
}

// the same with the direction's density already known (the NEE block of the same pass evaluated it for this very direction)
template <bool ENV>
__device__ __forceinline__ float emitter_sample_value_with_pdf(const Params &P, V3 d, float p, float val[3])
{
    if constexpr (ENV) {
        float Le[3];
        envmap_eval(P, d, Le);
#pragma unroll
        for (int k = 0; k < 3; ++k) val[k] = p > 0.0f ? Le[k] / p : 0.0f;
        return p;
    } else return emitter_sample_value<ENV>(P, d, val);
}

template <bool ENV>
__device__ __forceinline__ float emitter_pdf(const Params &P, V3 d)
{
    if constexpr (ENV) return envmap_pdf(P, d);
    else return kInvFourPi;
}

// radiance towards an escaped ray AND the emitter's density of its direction (the escape side of the MIS weight, :273-284)
template <bool ENV>
__device__ __forceinline__ float emitter_eval_pdf(const Params &P, V3 d, float Le[3])
{
    if constexpr (ENV) return envmap_eval_pdf(P, d, Le);
    else { Le[0] = P.Le[0]; Le[1] = P.Le[1]; Le[2] = P.Le[2]; return kInvFourPi; }
}

// emitter radiance towards an escaped ray
template <bool ENV>
__device__ __forceinline__ void emitter_eval(const Params &P, V3 d, float Le[3])
{
    if constexpr (ENV) envmap_eval(P, d, Le);
    else { Le[0] = P.Le[0]; Le[1] = P.Le[1]; Le[2] = P.Le[2]; }
}

// Medium::sample_interaction with a majorant supergrid (scene_config.py:36, optimize.py:182-199) [M3-ext]: 3-D DDA
// through the cells the ray crosses, accumulating majorant * length until tau = -log(1-u) is reached.  Returns the
// distance (inf if the ray leaves [0, tmax] first) and the local majorant / reciprocal at the collision.
// `mocc`: bitmask of the cells with a non-zero majorant as this wave reads it (LDS copy inside the tracing kernels,
// nullptr: every cell is loaded): empty cells contribute nothing to the optical depth, so the walk through empty
// space - most of a sparse volume - runs on LDS bit tests instead of a chain of dependent L2 loads.
__device__ __forceinline__ float dda_collision(const Params &P, const float *mg, const uint32_t *mocc, V3 o, V3 d, float tmax,
                                               float u, float &m_out, float &im_out)
{
    const float tau = -drt_logf(1.0f - u);
    float gxf = ((o.x - P.bmin[0]) * P.inv_ext[0]) * (float) P.gx, dgx = (d.x * P.inv_ext[0]) * (float) P.gx;
    float gyf = ((o.y - P.bmin[1]) * P.inv_ext[1]) * (float) P.gy, dgy = (d.y * P.inv_ext[1]) * (float) P.gy;
    float gzf = ((o.z - P.bmin[2]) * P.inv_ext[2]) * (float) P.gz, dgz = (d.z * P.inv_ext[2]) * (float) P.gz;
    float flx = fminf(fmaxf(floorf(gxf), 0.0f), (float)(P.gx - 1));
    float fly = fminf(fmaxf(floorf(gyf), 0.0f), (float)(P.gy - 1));
    float flz = fminf(fmaxf(floorf(gzf), 0.0f), (float)(P.gz - 1));
    int cx = (int) flx, cy = (int) fly, cz = (int) flz;
    // crossing times: td = 1 / |dg| (the same for every flight of a walk), first crossing = cells to the next plane * td;
    // |dg| < 1e-20: parallel (every crossing time stays finite or +inf) - as oracle/drt_oracle.c sample_collision
    float tnx, tny, tnz, tdx, tdy, tdz; int sx, sy, sz;
    if (dgx >= 1e-20f) { tdx = 1.0f / dgx; tnx = ((flx + 1.0f) - gxf) * tdx; sx = 1; }
    else if (dgx <= -1e-20f) { tdx = 1.0f / -dgx; tnx = (gxf - flx) * tdx; sx = -1; }
    else { tnx = kInf; tdx = kInf; sx = 0; }
    if (dgy >= 1e-20f) { tdy = 1.0f / dgy; tny = ((fly + 1.0f) - gyf) * tdy; sy = 1; }
    else if (dgy <= -1e-20f) { tdy = 1.0f / -dgy; tny = (gyf - fly) * tdy; sy = -1; }
    else { tny = kInf; tdy = kInf; sy = 0; }
    if (dgz >= 1e-20f) { tdz = 1.0f / dgz; tnz = ((flz + 1.0f) - gzf) * tdz; sz = 1; }
    else if (dgz <= -1e-20f) { tdz = 1.0f / -dgz; tnz = (gzf - flz) * tdz; sz = -1; }
    else { tnz = kInf; tdz = kInf; sz = 0; }
    float t = 0.0f, acc = 0.0f;
    for (;;) {
        int a = (tny < tnx) ? 1 : 0;
        float tmin = (tny < tnx) ? tny : tnx;
        if (tnz < tmin) { a = 2; tmin = tnz; }
        float texit = fminf(tmin, tmax);
        const int cell = (cz * P.gy + cy) * P.gx + cx;
        float mc = 0.0f;
        if (!mocc || ((mocc[cell >> 5] >> (cell & 31)) & 1u)) mc = mg[cell];
        if (mc > 0.0f) {
            float dtau = mc * (texit - t);
            if (acc + dtau >= tau) { float im = 1.0f / mc; m_out = mc; im_out = im; return fmaf(tau - acc, im, t); }
            acc += dtau;
        }
        t = texit;
        if (!(texit < tmax)) break;
        if (a == 0) { cx += sx; if (cx < 0 || cx >= P.gx) break; tnx += tdx; }
        else if (a == 1) { cy += sy; if (cy < 0 || cy >= P.gy) break; tny += tdy; }
        else { cz += sz; if (cz < 0 || cz >= P.gz) break; tnz += tdz; }
    }
    m_out = 0.0f; im_out = 0.0f;
    return kInf;
}

enum CounterSlot { C_RAYS = 0, C_DT, C_RT, C_DRT, C_ALB, C_TR, C_RT_ADJ, C_SC, C_SC_ALB, C_COUNT };

// ---------------------------------------------------------------------------
// GridVolume::eval: trilinear, clamp, cell-centred (q = p_local*res - 0.5);
// get_scattering_coefficients / get_albedo (volpathsimple.py:141,375,554,578,598)
// ---------------------------------------------------------------------------
struct Stencil {
    int x0, x1, y0, y1, z0, z1;      // y*/z* premultiplied by their strides
    float wx0, wx1, wy0, wy1, wz0, wz1;
};

__device__ __forceinline__ void axis_setup(float p, float bmin, float inv_ext, int res,
                                           int &i0, int &i1, float &w0, float &w1)
{
    float l = (p - bmin) * inv_ext;
    float q = fmaf(l, (float) res, -0.5f);
    float fl = floorf(q);
    float fw = q - fl;
    fl = fminf(fmaxf(fl, -1.0f), (float) res);
    int i = (int) fl;
    i0 = min(max(i, 0), res - 1);
    i1 = min(max(i + 1, 0), res - 1);
    w1 = fw; w0 = 1.0f - fw;
}

__device__ __forceinline__ Stencil make_stencil(const Params &P, V3 p)
{
    Stencil s;
    axis_setup(p.x, P.bmin[0], P.inv_ext[0], P.rx, s.x0, s.x1, s.wx0, s.wx1);
    axis_setup(p.y, P.bmin[1], P.inv_ext[1], P.ry, s.y0, s.y1, s.wy0, s.wy1);
    axis_setup(p.z, P.bmin[2], P.inv_ext[2], P.rz, s.z0, s.z1, s.wz0, s.wz1);
    s.y0 *= P.rx; s.y1 *= P.rx;
    int sz = P.rx * P.ry;
    s.z0 *= sz; s.z1 *= sz;
    return s;
}

__device__ __forceinline__ void stencil_weights(const Stencil &s, float w[8]);

// Gradient scratch addressing.  The atomic path retires one request per 64-byte line per wave
// instruction (tools/ubench), so the scratch is laid out such that a splat's whole 2x2x2 footprint
// is ONE line: line (bx, y0, z0) with bx = x0 / 3 holds, for the base corners x0 in [3bx, 3bx+2],
// the 16 floats  [(dz*2 + dy)*4 + (x - 3bx)],  x in [3bx, 3bx+3]  of voxels (x, y0+dy, z0+dz).
// A voxel therefore has up to 8 slots (2 along x at block seams, 2 along y, 2 along z); they are
// summed by untile_gradients_kernel.  Storage 16/3 = 5.3x the grid per plane - HBM is plentiful.
// Clamped corners (x1 == x0 at the grid border) fold onto the x0 slot.
__device__ __forceinline__ void make_grad_indices(const Params &P, V3 p, int idx[8], float w[8])
{
    Stencil s;
    axis_setup(p.x, P.bmin[0], P.inv_ext[0], P.rx, s.x0, s.x1, s.wx0, s.wx1);
    axis_setup(p.y, P.bmin[1], P.inv_ext[1], P.ry, s.y0, s.y1, s.wy0, s.wy1);
    axis_setup(p.z, P.bmin[2], P.inv_ext[2], P.rz, s.z0, s.z1, s.wz0, s.wz1);
    stencil_weights(s, w);
    const uint32_t bx = ((uint32_t) s.x0 * 43691u) >> 17;            // x0 / 3
    const int base = (int) ((((uint32_t) s.z0 * (uint32_t) P.ry + (uint32_t) s.y0) * (uint32_t) P.gt_nbx + bx) << 4)
                   + (s.x0 - 3 * (int) bx);
    const int dx = s.x1 - s.x0, dy = (s.y1 - s.y0) << 2, dz = (s.z1 - s.z0) << 3;   // 0 when clamped
    idx[0] = base;           idx[1] = base + dx;
    idx[2] = base + dy;      idx[3] = base + dy + dx;
    idx[4] = base + dz;      idx[5] = base + dz + dx;
    idx[6] = base + dz + dy; idx[7] = base + dz + dy + dx;
}

__device__ __forceinline__ float trilerp8(const Stencil &s, float d0, float d1, float d2, float d3,
                                          float d4, float d5, float d6, float d7)
{
    float v00 = fmaf(s.wx0, d0, s.wx1 * d1);
    float v01 = fmaf(s.wx0, d2, s.wx1 * d3);
    float v10 = fmaf(s.wx0, d4, s.wx1 * d5);
    float v11 = fmaf(s.wx0, d6, s.wx1 * d7);
    float v0 = fmaf(s.wy0, v00, s.wy1 * v01);
    float v1 = fmaf(s.wy0, v10, s.wy1 * v11);
    return fmaf(s.wz0, v0, s.wz1 * v1);
}

// sigma_t lookup from the library-owned APRON-BRICK copy (Params::sigma_b).  One 128-byte line
// holds the 4x4x2 voxels [3bx, 3bx+3] x [3by, 3by+3] x [z, z+1] (indices clamped): every trilinear
// footprint whose base corner (x0,y0,z0) has x0 in [3bx,3bx+2], y0 in [3by,3by+2], z0 = z lies in
// exactly ONE line (an ordinary 4x4x2 brick layout averaged 2.3 lines, the caller's linear layout
// 4.1), and the two x-neighbours of each corner pair are adjacent floats (4 x 8-byte loads).
// Storage: 32/9 = 3.6x the grid; the values are copies, so results are unchanged bit for bit.
// `occ` = the empty-space bitmask as this wave reads it (LDS copy inside the tracing kernels).
__device__ __forceinline__ float eval_sigma_t(const Params &P, V3 p, const uint32_t *occ)
{
    Stencil s;
    axis_setup(p.x, P.bmin[0], P.inv_ext[0], P.rx, s.x0, s.x1, s.wx0, s.wx1);
    axis_setup(p.y, P.bmin[1], P.inv_ext[1], P.ry, s.y0, s.y1, s.wy0, s.wy1);
    axis_setup(p.z, P.bmin[2], P.inv_ext[2], P.rz, s.z0, s.z1, s.wz0, s.wz1);
    // index arithmetic with 24-bit multiplies (v_mul_u32_u24 / v_mad_u32_u24 issue at the full rate, v_mul_lo_u32
    // at less than half of it - tools/ubench/valu_issue.hip); every operand below is < 2^24 (drt_set_medium checks)
    if (occ) {
        const uint32_t c = __umul24(__umul24((uint32_t) s.z0 >> P.occ_shift, (uint32_t) P.occ_y) + ((uint32_t) s.y0 >> P.occ_shift),
                                    (uint32_t) P.occ_x) + ((uint32_t) s.x0 >> P.occ_shift);
        if (!((occ[c >> 5] >> (c & 31u)) & 1u)) return 0.0f;     // all 8 corners are exactly 0
    }
    // x0 / 3 and x0 % 3 for x0 < 65536 / 3 (multiply-shift)
    const uint32_t bx = __umul24((uint32_t) s.x0, 43691u) >> 17, by = __umul24((uint32_t) s.y0, 43691u) >> 17;
    const uint32_t ox = (uint32_t) s.x0 - 3u * bx, oy = (uint32_t) s.y0 - 3u * by;
    const float *g = P.sigma_b + ((size_t) (__umul24((uint32_t) s.z0, (uint32_t) P.sb_zstride) + __umul24(by, (uint32_t) P.sb_ystride) + bx) << 5)
                   + (oy << 2) + ox;
    // the line stores clamped neighbours itself, so +1 / +4 / +16 are always the right corners
    float d0 = g[0], d1 = g[1], d2 = g[4], d3 = g[5], d4 = g[16], d5 = g[17], d6 = g[20], d7 = g[21];
    // lower clamp (floor(q) = -1): both corners of that axis are voxel 0 (axis_setup), not 0 and 1.  Only lookups
    // within half a voxel of the box surface get here: one wave-level test keeps the 15 selects out of the common path
    const bool border = s.x1 == s.x0 || s.y1 == s.y0 || s.z1 == s.z0;
    if (__builtin_expect(__ballot(border) != 0ull, 0)) {
        if (s.x1 == s.x0) { d1 = d0; d3 = d2; d5 = d4; d7 = d6; }
        if (s.y1 == s.y0) { d2 = d0; d3 = d1; d6 = d4; d7 = d5; }
        if (s.z1 == s.z0) { d4 = d0; d5 = d1; d6 = d2; d7 = d3; }
    }
    return trilerp8(s, d0, d1, d2, d3, d4, d5, d6, d7) * P.scale;
}

// true iff the empty-space bitmask says that every voxel a lookup at p can touch is exactly zero
__device__ __forceinline__ bool occ_empty(const Params &P, V3 p, const uint32_t *occ)
{
    int x0, y0, z0, i1; float w0, w1;
    axis_setup(p.x, P.bmin[0], P.inv_ext[0], P.rx, x0, i1, w0, w1);
    axis_setup(p.y, P.bmin[1], P.inv_ext[1], P.ry, y0, i1, w0, w1);
    axis_setup(p.z, P.bmin[2], P.inv_ext[2], P.rz, z0, i1, w0, w1);
    const uint32_t c = __umul24(__umul24((uint32_t) z0 >> P.occ_shift, (uint32_t) P.occ_y) + ((uint32_t) y0 >> P.occ_shift),
                                (uint32_t) P.occ_x) + ((uint32_t) x0 >> P.occ_shift);
    return !((occ[c >> 5] >> (c & 31u)) & 1u);
}

// sigma_t AND the colour grid at p from the interleaved four-channel apron-brick copy (Params::grid4): block
// (bx = x0 / 3, y0, z0) = 256 bytes = TWO 128-byte lines holds the voxels [3bx, 3bx+3] x {y0, y0+1} x {z0, z0+1}
// (indices clamped) as float4 {sigma_t, r, g, b}, slot (dz*2 + dy)*4 + (x - 3bx): every trilinear footprint is eight
// float4 loads from one block (its z0 slab in one line, its z1 slab in the next), where the separate layouts
// need 1 line (sigma_t bricks) + 4..8 lines ((Z,Y,X,3) rows).  Storage 16/3 x the four-channel grid.  The values are
// copies and the interpolation is trilerp8 with the same stencil: results equal eval_sigma_t / eval_rgb bit for bit.
__device__ __forceinline__ void eval4(const Params &P, V3 p, float &sigma_t, float rgb[3])
{
    Stencil s;
    axis_setup(p.x, P.bmin[0], P.inv_ext[0], P.rx, s.x0, s.x1, s.wx0, s.wx1);
    axis_setup(p.y, P.bmin[1], P.inv_ext[1], P.ry, s.y0, s.y1, s.wy0, s.wy1);
    axis_setup(p.z, P.bmin[2], P.inv_ext[2], P.rz, s.z0, s.z1, s.wz0, s.wz1);
    const uint32_t bx = __umul24((uint32_t) s.x0, 43691u) >> 17, ox = (uint32_t) s.x0 - 3u * bx;
    const float4 *g = P.grid4 + ((size_t) ((uint32_t) s.z0 * (uint32_t) P.ry + (uint32_t) s.y0) * (uint32_t) P.g4_nbx + bx) * 16 + ox;
    float4 d0 = g[0], d1 = g[1], d2 = g[4], d3 = g[5], d4 = g[8], d5 = g[9], d6 = g[12], d7 = g[13];
    // lower clamp (floor(q) = -1): both corners of that axis are voxel 0 (see eval_sigma_t)
    const bool border = s.x1 == s.x0 || s.y1 == s.y0 || s.z1 == s.z0;
    if (__builtin_expect(__ballot(border) != 0ull, 0)) {
        if (s.x1 == s.x0) { d1 = d0; d3 = d2; d5 = d4; d7 = d6; }
        if (s.y1 == s.y0) { d2 = d0; d3 = d1; d6 = d4; d7 = d5; }
        if (s.z1 == s.z0) { d4 = d0; d5 = d1; d6 = d2; d7 = d3; }
    }
    sigma_t = trilerp8(s, d0.x, d1.x, d2.x, d3.x, d4.x, d5.x, d6.x, d7.x) * P.scale;
    rgb[0] = trilerp8(s, d0.y, d1.y, d2.y, d3.y, d4.y, d5.y, d6.y, d7.y);
    rgb[1] = trilerp8(s, d0.z, d1.z, d2.z, d3.z, d4.z, d5.z, d6.z, d7.z);
    rgb[2] = trilerp8(s, d0.w, d1.w, d2.w, d3.w, d4.w, d5.w, d6.w, d7.w);
}

// the stencil of a lookup into a COLOUR grid: on sigma_t's lattice everywhere but in drt_own.hip (DRT_COLOUR_OWN), where it is the grids' own
__device__ __forceinline__ Stencil make_stencil_colour(const Params &P, V3 p)
{
#ifdef DRT_COLOUR_OWN
    Stencil s;
    axis_setup(p.x, P.bmin[0], P.inv_ext[0], P.crx, s.x0, s.x1, s.wx0, s.wx1);
    axis_setup(p.y, P.bmin[1], P.inv_ext[1], P.cry, s.y0, s.y1, s.wy0, s.wy1);
    axis_setup(p.z, P.bmin[2], P.inv_ext[2], P.crz, s.z0, s.z1, s.wz0, s.wz1);
    s.y0 *= P.crx; s.y1 *= P.crx;
    int sz = P.crx * P.cry;
    s.z0 *= sz; s.z1 *= sz;
    return s;
#else
    return make_stencil(P, p);
#endif
}

__device__ __forceinline__ void eval_rgb(const Params &P, const float *g, V3 p, float out[3])
{
    Stencil s = make_stencil_colour(P, p);
    int a = s.z0 + s.y0, b = s.z0 + s.y1, c = s.z1 + s.y0, d = s.z1 + s.y1;
    int i0 = 3 * (a + s.x0), i1 = 3 * (a + s.x1), i2 = 3 * (b + s.x0), i3 = 3 * (b + s.x1);
    int i4 = 3 * (c + s.x0), i5 = 3 * (c + s.x1), i6 = 3 * (d + s.x0), i7 = 3 * (d + s.x1);
#pragma unroll
    for (int ch = 0; ch < 3; ++ch)
        out[ch] = trilerp8(s, g[i0 + ch], g[i1 + ch], g[i2 + ch], g[i3 + ch],
                           g[i4 + ch], g[i5 + ch], g[i6 + ch], g[i7 + ch]);
}

__device__ __forceinline__ void eval_albedo(const Params &P, V3 p, float out[3])
{
#if defined(DRT_EXP_ALB) && DRT_EXP_ALB == 1       // timing experiment: no albedo loads at all (results are wrong)
    out[0] = 0.8f; out[1] = 0.65f; out[2] = 0.45f + 1e-9f * p.x; return;
#endif
    eval_rgb(P, P.albedo, p, out);
}

// Reverse mode of the trilinear gather = 8-corner scatter-add.  gfx950 has a
// hardware fp32 global atomic add (global_atomic_add_f32, device scope); the
// library is built with -munsafe-fp-atomics so atomicAdd lowers to it.
__device__ __forceinline__ void stencil_weights(const Stencil &s, float w[8])
{
    float zy00 = s.wz0 * s.wy0, zy01 = s.wz0 * s.wy1;
    float zy10 = s.wz1 * s.wy0, zy11 = s.wz1 * s.wy1;
    w[0] = zy00 * s.wx0; w[1] = zy00 * s.wx1; w[2] = zy01 * s.wx0; w[3] = zy01 * s.wx1;
    w[4] = zy10 * s.wx0; w[5] = zy10 * s.wx1; w[6] = zy11 * s.wx0; w[7] = zy11 * s.wx1;
}

__device__ __forceinline__ void stencil_indices(const Stencil &s, int idx[8])
{
    int a = s.z0 + s.y0, b = s.z0 + s.y1, c = s.z1 + s.y0, d = s.z1 + s.y1;
    idx[0] = a + s.x0; idx[1] = a + s.x1; idx[2] = b + s.x0; idx[3] = b + s.x1;
    idx[4] = c + s.x0; idx[5] = c + s.x1; idx[6] = d + s.x0; idx[7] = d + s.x1;
}

// Cooperative scatter.  Measured on MI355X (tools/ubench/atomic_*.hip): the fp32
// atomic path retires ~21 G requests/s chip-wide, where one request = one 64-byte
// line touched by one wave instruction, whatever the number of lanes that hit it
// (16 lanes on 16 consecutive floats cost the same as 1).  A lane that issues its
// 8 corners as 8 instructions therefore pays 8 requests; if the lanes that are
// active at the splat site share the work so that the two x-neighbours of a corner
// pair leave in the SAME instruction, the pair costs one request.  The active
// lanes stage (index, value) records in a wave-private LDS area and then, in groups
// of up to 8 lanes, walk the group's records: lane j of the group adds corner j.
constexpr int kCoopDwords = 32;   // per lane: 8 indices + 3 x 8 values
constexpr int kPermGroup = 1024;  // rays per sort group of the adjoint's ray -> lane schedule (a multiple of the workgroup size)
constexpr int kOccWords = 1024;   // empty-space bitmask: at most 32768 cells = 4 KiB of LDS per workgroup

// Lanes of ONE wavefront exchange records through LDS.  The hardware executes a wave's DS
// instructions in order, so all that is needed is (1) that the compiler neither reorders nor caches
// LDS accesses across this point - wavefront-scope fences alone were observed NOT to guarantee that
// (stale records, wrong gradients) - and (2) that outstanding LDS operations have landed.
__device__ __forceinline__ void coop_stage_sync()
{
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    asm volatile("" ::: "memory");
}

template <int NCH>
__device__ __forceinline__ void coop_scatter(float *dst, uint32_t chan_stride, const int idx[8],
                                             const float (&val)[NCH][8], uint32_t *rec)
{
    const uint64_t mask = __ballot(1);
    const uint32_t lane = __lane_id();
    const uint32_t rank = (uint32_t) __popcll(mask & ((1ull << lane) - 1ull));
    const uint32_t k = (uint32_t) __popcll(mask);
    uint32_t *mine = rec + rank * kCoopDwords;
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        mine[c] = (uint32_t) idx[c];
#pragma unroll
        for (int ch = 0; ch < NCH; ++ch) mine[8 + 8 * ch + c] = __float_as_uint(val[ch][c]);
    }
    coop_stage_sync();
    const uint32_t g0 = rank & ~7u, j = rank & 7u;
    const uint32_t m = min(8u, k - g0);
    for (uint32_t t = 0; t < m; ++t) {
        const uint32_t *src = rec + (g0 + t) * kCoopDwords;
        for (uint32_t c = j; c < 8; c += m) {
            float *p = dst + src[c];
#pragma unroll
            for (int ch = 0; ch < NCH; ++ch) atomicAdd(p + (size_t) ch * chan_stride, __uint_as_float(src[8 + 8 * ch + c]));
        }
    }
    coop_stage_sync();
}

// ---------------------------------------------------------------------------
// Deferred splatting: record append.  `st` = the wave's LDS state {cur[2], -, -, end[2]} (absolute record
// slots of the stream's open chunk group).  Called under divergence: the active lanes compact themselves
// (ballot / prefix), the first one advances the cursor - taking fresh chunks from the global counter
// when the open ones fill up, so chunks are always full except each wave's last - and every lane
// stores its record with one (stream 0) or two (stream 1) 16-byte writes.  Out of chunks: the splat goes
// straight to the caller's grids with atomics (slow, correct; counted in rec_cursor[4+s]).
// ---------------------------------------------------------------------------
constexpr uint32_t kRecChunk = 256;
constexpr uint32_t kRecStreams = 2;
// chunks handed out per allocation (every wave fills several: fewer same-address returning atomics on the cursor)
__host__ __device__ constexpr uint32_t rec_group(int s) { return s == 0 ? 4u : 2u; }

__device__ __forceinline__ void splat_direct(const Params &P, int plane, V3 p, float v)
{
    Stencil st = make_stencil(P, p);
    float w[8];
    stencil_weights(st, w);
    float *dst = plane == 0 ? P.g_sigma : P.g_albedo + (plane - 1);
    const int stride = plane == 0 ? 1 : 3;
    const int idx[8] = { st.z0 + st.y0 + st.x0, st.z0 + st.y0 + st.x1, st.z0 + st.y1 + st.x0, st.z0 + st.y1 + st.x1,
                         st.z1 + st.y0 + st.x0, st.z1 + st.y0 + st.x1, st.z1 + st.y1 + st.x0, st.z1 + st.y1 + st.x1 };
#pragma unroll
    for (int k = 0; k < 8; ++k) atomicAdd(dst + (size_t) stride * idx[k], w[k] * v);
}

// The wave state lives in LDS; the pointer arrives as a generic one (it travels through a struct), so it
// is cast back to the LDS address space: plain ds_read / ds_write, which one wave executes in order,
// instead of FLAT accesses (observed to lose an update now and then under the nerf kernel's emission rate).
typedef __attribute__((address_space(3))) volatile uint32_t lds_u32;

// S = 0: {p, v0};  S = 1: {p, v0, c[0..2]} (c may be nullptr for S = 0)
template <int S>
__device__ __forceinline__ void emit_record(const Params &P, V3 p, float v0, const float *c, uint32_t *st_)
{
    lds_u32 *st = (lds_u32 *) st_;
    const uint64_t mask = __ballot(1);
    const uint32_t lane = __lane_id();
    const uint32_t rank = (uint32_t) __popcll(mask & ((1ull << lane) - 1ull));
    const uint32_t n = (uint32_t) __popcll(mask);
    uint32_t base0 = 0, base1 = 0, split = 0;
    if (rank == 0) {
        const uint32_t cur = st[S], end = st[4 + S];
        base0 = cur;
        if (cur + n <= end) { split = n; st[S] = cur + n; }
        else {
            split = end - cur;
            constexpr uint32_t G = rec_group(S);
            if (end) {                                                              // the open chunks are full
                const uint32_t last = (end - 1u) / kRecChunk;
                for (uint32_t j = 0; j < G; ++j) P.rec_chunk_count[S][last - j] = kRecChunk;
            }
            const uint32_t ch = atomicAdd(P.rec_cursor + S, G);
            if (ch + G <= P.rec_cap_chunks[S]) {
                base1 = ch * kRecChunk;
                st[S] = base1 + (n - split); st[4 + S] = base1 + G * kRecChunk;
            } else {
                base1 = 0xffffffffu;                                                // out of chunks
                st[S] = end;
                atomicAdd(P.rec_cursor + 4 + S, n - split);
            }
        }
    }
    base0 = (uint32_t) __builtin_amdgcn_readfirstlane((int) base0);
    base1 = (uint32_t) __builtin_amdgcn_readfirstlane((int) base1);
    split = (uint32_t) __builtin_amdgcn_readfirstlane((int) split);
    uint32_t slot = 0xffffffffu;
    if (rank < split) slot = base0 + rank;
    else if (base1 != 0xffffffffu) slot = base1 + (rank - split);
    if (slot != 0xffffffffu) {
        if constexpr (S == 0) P.rec_buf[0][slot] = make_float4(p.x, p.y, p.z, v0);
        else {
            float4 *dst = P.rec_buf[1] + 2 * (size_t) slot;
            dst[0] = make_float4(p.x, p.y, p.z, v0);
            dst[1] = make_float4(c[0], c[1], c[2], 0.0f);
        }
    } else {
        if (v0 != 0.0f) splat_direct(P, 0, p, v0);
        if constexpr (S == 1) {
#pragma unroll
            for (int k = 0; k < 3; ++k) if (c[k] != 0.0f) splat_direct(P, 1 + k, p, c[k]);
        }
    }
}

// N records of stream 0 per active lane with ONE reservation (the four resampled points of
// backpropagate_transmittance, volpathsimple.py:584-607: one cursor transaction instead of four; a lane's records are
// consecutive slots, 64 contiguous bytes, unless the reservation straddles two chunk groups)
template <int N>
__device__ __forceinline__ void emit_records0(const Params &P, const V3 (&p)[N], float v0, uint32_t *st_)
{
    lds_u32 *st = (lds_u32 *) st_;
    const uint64_t mask = __ballot(1);
    const uint32_t lane = __lane_id();
    const uint32_t rank = (uint32_t) __popcll(mask & ((1ull << lane) - 1ull)) * N;
    const uint32_t n = (uint32_t) __popcll(mask) * N;
    uint32_t base0 = 0, base1 = 0, split = 0;
    if (rank == 0) {
        const uint32_t cur = st[0], end = st[4];
        base0 = cur;
        if (cur + n <= end) { split = n; st[0] = cur + n; }
        else {
            split = end - cur;
            constexpr uint32_t G = rec_group(0);
            if (end) {                                                              // the open chunks are full
                const uint32_t last = (end - 1u) / kRecChunk;
                for (uint32_t j = 0; j < G; ++j) P.rec_chunk_count[0][last - j] = kRecChunk;
            }
            const uint32_t ch = atomicAdd(P.rec_cursor, G);
            if (ch + G <= P.rec_cap_chunks[0]) {
                base1 = ch * kRecChunk;
                st[0] = base1 + (n - split); st[4] = base1 + G * kRecChunk;
            } else {
                base1 = 0xffffffffu;                                                // out of chunks
                st[0] = end;
                atomicAdd(P.rec_cursor + 4, n - split);
            }
        }
    }
    base0 = (uint32_t) __builtin_amdgcn_readfirstlane((int) base0);
    base1 = (uint32_t) __builtin_amdgcn_readfirstlane((int) base1);
    split = (uint32_t) __builtin_amdgcn_readfirstlane((int) split);
#pragma unroll
    for (int j = 0; j < N; ++j) {
        const uint32_t r = rank + (uint32_t) j;
        uint32_t slot = 0xffffffffu;
        if (r < split) slot = base0 + r;
        else if (base1 != 0xffffffffu) slot = base1 + (r - split);
        if (slot != 0xffffffffu) P.rec_buf[0][slot] = make_float4(p[j].x, p[j].y, p[j].z, v0);
        else splat_direct(P, 0, p[j], v0);
    }
}

// end of the wave: publish the fill of the chunks still open
__device__ __forceinline__ void close_records(const Params &P, uint32_t *st_)
{
    lds_u32 *st = (lds_u32 *) st_;
    if (__lane_id() < kRecStreams) {
        const int s = (int) __lane_id();
        const uint32_t cur = st[s], end = st[4 + s];
        if (end) {
            const uint32_t G = rec_group(s), first = end / kRecChunk - G;
            for (uint32_t j = 0; j < G; ++j) {
                const uint32_t lo = (first + j) * kRecChunk;
                P.rec_chunk_count[s][first + j] = cur <= lo ? 0u : (cur - lo < kRecChunk ? cur - lo : kRecChunk);
            }
        }
    }
}

// DEFER: `rec` is the wave's record state (emit_record); otherwise the cooperative-scatter staging area
template <bool DEFER = false>
__device__ __forceinline__ void splat_sigma_t(const Params &P, V3 p, float g, uint32_t *rec)
{
    if (g == 0.0f) return;            // adding exact zeros changes nothing: skip the requests
    if constexpr (DEFER) {
        if (dbg(P.debug_flags, 1u)) return;
        emit_record<0>(P, p, g * P.scale, nullptr, rec);
        return;
    }
    float w[8]; int idx[8];
    make_grad_indices(P, p, idx, w);
    float gs = g * P.scale;
    if (dbg(P.debug_flags, 1u)) return;   // ablation: no gradient atomics
    if (dbg(P.debug_flags, 2u)) {         // ablation: one lane, eight instructions (round-1 v1 behaviour)
#pragma unroll
        for (int k = 0; k < 8; ++k) atomicAdd(P.gt + idx[k], w[k] * gs);
        return;
    }
    float val[1][8];
#pragma unroll
    for (int k = 0; k < 8; ++k) val[0][k] = w[k] * gs;
    coop_scatter<1>(P.gt, 0, idx, val, rec);
}

#ifdef DRT_COLOUR_OWN
// drt_own.hip: the colour grids have their own lattice, which the tile partition, the LDS tiles and the apron scratch (all laid out on
// sigma_t's) do not know - the 8 x 3 corner products go straight to the caller's colour gradient grid (fp32 atomics).  Slow and correct: this
// unit serves the configurations whose grids differ in resolution, not the benchmark.
__device__ __forceinline__ void splat_colour_own(const Params &P, V3 p, const float g[3])
{
    if (g[0] == 0.0f && g[1] == 0.0f && g[2] == 0.0f) return;
    if (dbg(P.debug_flags, 1u)) return;
    const Stencil st = make_stencil_colour(P, p);
    float w[8];
    stencil_weights(st, w);
    const int idx[8] = { st.z0 + st.y0 + st.x0, st.z0 + st.y0 + st.x1, st.z0 + st.y1 + st.x0, st.z0 + st.y1 + st.x1,
                         st.z1 + st.y0 + st.x0, st.z1 + st.y0 + st.x1, st.z1 + st.y1 + st.x0, st.z1 + st.y1 + st.x1 };
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        float *dst = P.g_albedo + 3 * (size_t) idx[k];
#pragma unroll
        for (int c = 0; c < 3; ++c) if (g[c] != 0.0f) atomicAdd(dst + c, w[k] * g[c]);
    }
}
#endif

template <bool DEFER = false>
__device__ __forceinline__ void splat_albedo(const Params &P, V3 p, const float g[3], uint32_t *rec)
{
    static_assert(!DEFER, "deferred colour splats travel with their sigma_t splat: splat_scatter");
#ifdef DRT_COLOUR_OWN
    splat_colour_own(P, p, g);
    return;
#endif
    if (g[0] == 0.0f && g[1] == 0.0f && g[2] == 0.0f) return;   // e.g. nerf queries in empty space (weight 0)
    float w[8]; int idx[8];
    make_grad_indices(P, p, idx, w);
    if (dbg(P.debug_flags, 1u)) return;
    if (dbg(P.debug_flags, 2u)) {
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            float *dst = P.gt + P.gt_plane + idx[k];
            atomicAdd(dst, w[k] * g[0]);
            atomicAdd(dst + P.gt_plane, w[k] * g[1]);
            atomicAdd(dst + 2 * (size_t) P.gt_plane, w[k] * g[2]);
        }
        return;
    }
    float val[3][8];
#pragma unroll
    for (int k = 0; k < 8; ++k) { val[0][k] = w[k] * g[0]; val[1][k] = w[k] * g[1]; val[2][k] = w[k] * g[2]; }
    coop_scatter<3>(P.gt + P.gt_plane, P.gt_plane, idx, val, rec);
}

// A sigma_t splat and a colour (albedo / emission) splat at the SAME point: scatter events
// (volpathsimple.py:170,580) and nerf queries (nerf.py:122-129).  Deferred: ONE 32-byte record when any colour
// channel is non-zero, the 16-byte sigma_t record otherwise (nerf queries in empty space), nothing if all are zero.
template <bool DEFER = false>
__device__ __forceinline__ void splat_scatter(const Params &P, V3 p, float gs, const float ga[3], uint32_t *rec)
{
#ifdef DRT_COLOUR_OWN
    splat_sigma_t<DEFER>(P, p, gs, rec);         // (its record / its scratch line: sigma_t's lattice, as everywhere)
    splat_colour_own(P, p, ga);
    return;
#endif
    if constexpr (DEFER) {
        if (dbg(P.debug_flags, 1u)) return;
        const bool colour = ga[0] != 0.0f || ga[1] != 0.0f || ga[2] != 0.0f;
        if (colour) emit_record<1>(P, p, gs * P.scale, ga, rec);
        else if (gs != 0.0f) emit_record<0>(P, p, gs * P.scale, nullptr, rec);
    } else {
        splat_sigma_t<false>(P, p, gs, rec);
        splat_albedo<false>(P, p, ga, rec);
    }
}

// ---------------------------------------------------------------------------
// scene.ray_intersect with use_bbox_fast_path: nearest hit with the medium box
// surface at t > 0 (volpathsimple.py:234,265,298,307,428,637)
// ---------------------------------------------------------------------------
struct Hit { bool valid; float t; V3 p; V3 n; };

__device__ __forceinline__ Hit box_hit(const Params &P, V3 o, V3 d)
{
    Hit h; h.valid = false; h.t = kInf; h.p = v3(0, 0, 0); h.n = v3(0, 0, 0);
    float tn = -kInf, tf = kInf;
    int an = 0, af = 0;
    const float oo[3] = { o.x, o.y, o.z }, dd[3] = { d.x, d.y, d.z };
    bool miss = false;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        if (dd[a] != 0.0f) {
            // (through the reciprocal direction, as BoundingBox::ray_intersect: one IEEE division per axis instead of two)
            const float rcp = 1.0f / dd[a];
            float t0 = (P.bmin[a] - oo[a]) * rcp;
            float t1 = (P.bmax[a] - oo[a]) * rcp;
            if (t0 > t1) { float t = t0; t0 = t1; t1 = t; }
            if (t0 > tn) { tn = t0; an = a; }
            if (t1 < tf) { tf = t1; af = a; }
        } else if (oo[a] < P.bmin[a] || oo[a] > P.bmax[a]) {
            miss = true;
        }
    }
    if (miss || !(tn <= tf)) return h;
    float t, sgn; int ax;
    if (tn > 0.0f) { t = tn; ax = an; sgn = (dd[ax] > 0.0f) ? -1.0f : 1.0f; }
    else if (tf > 0.0f) { t = tf; ax = af; sgn = (dd[ax] > 0.0f) ? 1.0f : -1.0f; }
    else return h;
    if (!isfinite(t)) return h;
    h.valid = true; h.t = t; h.p = ray_at(o, d, t);
    h.n = v3(ax == 0 ? sgn : 0.0f, ax == 1 ? sgn : 0.0f, ax == 2 ? sgn : 0.0f);
    return h;
}

// SurfaceInteraction::spawn_ray -> offset_p(d)
__device__ __forceinline__ V3 offset_p(const Hit &h, V3 d)
{
    float mag = (1.0f + fmaxf(fabsf(h.p.x), fmaxf(fabsf(h.p.y), fabsf(h.p.z)))) * kRayEps;
    float dn = h.n.x * d.x + h.n.y * d.y + h.n.z * d.z;
    if (dn < 0.0f) mag = -mag;
    return V3{ fmaf(mag, h.n.x, h.p.x), fmaf(mag, h.n.y, h.p.y), fmaf(mag, h.n.z, h.p.z) };
}

// perspective sensor ray (tests/test_integrators.py:46-67 fixture; Mitsuba look_at frame)
__device__ __forceinline__ void sensor_ray(const Params &P, uint32_t pixel, float ux, float uy,
                                           V3 &o, V3 &d)
{
    uint32_t py = pixel / (uint32_t) P.width, px = pixel - py * (uint32_t) P.width;
    float sx = ((float) px + ux) * (1.0f / (float) P.width);
    float sy = ((float) py + uy) * (1.0f / (float) P.height);
    float cx = fmaf(-2.0f, sx, 1.0f) * P.tan_x;
    float cy = fmaf(-2.0f, sy, 1.0f) * P.tan_y;
    float inv = 1.0f / sqrtf(fmaf(cx, cx, fmaf(cy, cy, 1.0f)));
    cx *= inv; cy *= inv; float cz = inv;
    o = v3(P.cam_o[0], P.cam_o[1], P.cam_o[2]);
    d = v3(fmaf(P.cam_left[0], cx, fmaf(P.cam_up[0], cy, P.cam_dir[0] * cz)),
           fmaf(P.cam_left[1], cx, fmaf(P.cam_up[1], cy, P.cam_dir[1] * cz)),
           fmaf(P.cam_left[2], cx, fmaf(P.cam_up[2], cy, P.cam_dir[2] * cz)));
}

}  // namespace drt
