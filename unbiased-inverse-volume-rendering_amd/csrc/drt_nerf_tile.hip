// drt_nerf_tile.hip -- the adjoint pass of NeRFIntegrator.sample (python/integrators/nerf.py:47-148, backward: :122-129) for sensor rays,
// with the voxel gradients PRE-REDUCED IN LDS (round 5).
//
// Why: the emission-absorption march splats a sigma_t gradient (and, where the query has weight, an emission gradient) at EVERY query -
// 128 per ray, 907 M per step of BASELINE config 5 - where the scattering integrator splats a dozen per ray.  As deferred records
// (nerf_kernel<ADJ, ., DEFER>, drt_deferred.hip) that is 14.5 GB of 16-byte records per step, written, histogrammed, scattered and read
// again: the reduction passes were 25 of the 46 ms of the fused adjoint pass (profiles/r04_fused_kernel_stats.csv).  But a march is
// COHERENT where a scattering path is not: the rays of a small pixel tile walk through the grid side by side, so at march step j
// all their queries lie within a few voxels of each other.  Here a workgroup owns an 8 x 8-pixel tile (a wave = 16 of its pixels x 4 samples) and adds
// every splat into a 16^3-voxel WINDOW of four-channel accumulators in LDS (torus addressing: voxel (x, y, z) lives in slot (z & 15, y & 15,
// x & 15) while the window covers it).  Every ray runs on by itself until a splat falls outside the window; when every ray of the workgroup
// waits (or is done) the window's non-zero accumulators are flushed to the caller's grids (one global atomic per voxel and channel -
// consecutive lanes flush consecutive x, one 64-byte request per 16 voxels) and the window moves to the waiting splat closest to the camera:
// that ray is inside by construction, so every phase makes progress and no splat ever bypasses the window (config 5: 11 phases per workgroup
// for 128 queries; ~10^8 global atomic lanes per step instead of 907 M x 8 corners; no records).
// The accumulators are 64-bit FIXED-POINT integers (ds_add_u64): measured on this chip (tools/ubench/lds_atomic_conflict_rate.hip) ds_add_f32
// retires 0.2 T lane-adds/s whatever the addresses - a first version with float accumulators spent 62 % of its wave-cycles waiting for the LDS,
// 72 ms per launch - where ds_add_u64 retires 1.4 - 3.2 T/s at 8 - 64 distinct addresses per instruction.  The unit is a power of two 2^44 below
// a bound of the launch's largest possible splat (from max |dL|, max |L_in|, max emission, the longest march step: nerf_tile_bounds_kernel), so a
// contribution converts exactly down to 2^-44 of that bound and 2^19 of them fit: sums inside a window are exact, whatever their order.
// Non-finite inputs cannot travel through fixed point: the bounds kernel flags them and the pass marks the gradient grids NaN.
//
// Arithmetic per ray: the statements of nerf_kernel<ADJ> (drt_kernels.hip) in the same order - same lookups, same weights
// (stencil_weights); gradients differ from the record path by summation order only.  Used for launches of sensor rays
// (Params::sensor_flow), any spp (a workgroup marches 16 samples of its 64 pixels); explicit ray batches keep the record path.
#include <atomic>
#include "drt_device.h"
#include "drt_launch.h"

#ifndef DRT_NT_SPW
#define DRT_NT_SPW 4               // samples per wave (lane -> ray map of the adjoint kernel): 4, or 1 = a wave is one sample of the tile's 64 pixels (round 5)
#endif
#ifndef DRT_NT_THREADS
#define DRT_NT_THREADS 1024        // threads per workgroup: the 64 pixels of a tile x (DRT_NT_THREADS / 64) samples
#endif

#ifdef DRT_NT_STATS
// experiment build: [0] window phases, [2] splats
__device__ unsigned long long g_nt_dbg[8];
extern "C" int drt_nt_debug_read(unsigned long long *out, int reset)
{
    hipError_t e = hipMemcpyFromSymbol(out, HIP_SYMBOL(g_nt_dbg), sizeof(g_nt_dbg));
    if (e == hipSuccess && reset) { static unsigned long long z[8]; e = hipMemcpyToSymbol(HIP_SYMBOL(g_nt_dbg), z, sizeof(z)); }
    return (int) e;
}
#define NT_STAT(slot, v) do { atomicAdd(g_nt_dbg + (slot), (unsigned long long) (v)); } while (0)
#else
#define NT_STAT(slot, v) do { } while (0)
#endif

namespace drt {

namespace {

constexpr int kWin = 16;                                   // window edge in voxels (a power of two)
constexpr int kWinSlots = kWin * kWin * kWin;              // 4096 voxels
// slot of voxel (x, y, z) = (z & 15) * kSZ + (y & 15) * kSY + (x & 15): row and slab strides that spread a splat neighbourhood over the LDS banks
// (with strides 16 / 256 the rows y, y + 2, ... and EVERY slab z of a column share their banks: the 64 lanes of an add instruction - a few voxels
// wide, a few deep - met in ~10 banks, SQ_LDS_IDX_ACTIVE was 97 cycles per instruction and the LDS was busy 85 % of the kernel's 72 ms)
constexpr int kSY = 17, kSZ = 16 * kSY + 5;
constexpr int kWinStore = kWin * kSZ;                      // 4432 accumulators of 8 bytes per channel: 4 channels = 138.5 KiB, one workgroup of 16 waves per CU
constexpr int kFixBits = 44;

struct NerfTile {
    uint32_t tiles_x;              // tiles of 8 x 8 pixels per film row
    uint32_t groups;               // workgroups per tile: each marches DRT_NT_THREADS / 64 of the pixels' samples
    uint32_t *bounds;              // [0] max |dL|, [1] max |L_in| over the launch's rays, [2] max |emission| over the grid (float bits), [3] a non-finite one was seen, [4] the largest negative density's magnitude
    uint32_t g4;                   // lookups from the four-channel copy (Params::grid4) instead of sigma_b + emission
    uint32_t count;
};

// caller's sigma_t (Z,Y,X,1) + colour (Z,Y,X,3) -> interleaved four-channel apron-brick copy (see eval4);
// one thread per stored float4
__global__ void __launch_bounds__(256) brick_grid4_kernel(const float *sigma_t, const float *rgb, float4 *dst, int rx, int ry,
                                                          int rz, int nbx)
{
    const size_t t = (size_t) blockIdx.x * blockDim.x + threadIdx.x;
    const size_t total = (size_t) nbx * ry * rz * 16;
    if (t >= total) return;
    const uint32_t slot = (uint32_t) (t & 15), line = (uint32_t) (t >> 4);
    const uint32_t bx = line % (uint32_t) nbx, r = line / (uint32_t) nbx;
    const uint32_t y0 = r % (uint32_t) ry, z0 = r / (uint32_t) ry;
    const int x = min((int) (3 * bx + (slot & 3)), rx - 1);
    const int y = min((int) (y0 + ((slot >> 2) & 1)), ry - 1);
    const int z = min((int) (z0 + (slot >> 3)), rz - 1);
    const size_t v = ((size_t) z * ry + y) * rx + x;
    dst[t] = make_float4(sigma_t[v], rgb[3 * v], rgb[3 * v + 1], rgb[3 * v + 2]);
}

// x * inv (|.| < 2^51) as a two's complement integer, rounded to nearest: the double 1.5 x 2^52 + n holds n in its low mantissa bits
// (5 vector instructions where the float -> int64 cast takes 12: the kernel is bound by vector-instruction issue, profiles/r05_fused_pmc_util.txt)
__device__ __forceinline__ unsigned long long fix64(float x, double inv)
{
    const double magic = 6755399441055744.0;
    const double d = fma((double) x, inv, magic);
    return (unsigned long long) __double_as_longlong(d) - (unsigned long long) __double_as_longlong(magic);
}

// eval4 (drt_device.h) for a footprint that is known already (unscaled indices): the splat below needs the same stencil
__device__ __forceinline__ void eval4_at(const Params &P, const Stencil &s, float &sigma_t, float rgb[3])
{
    const uint32_t bx = __umul24((uint32_t) s.x0, 43691u) >> 17, ox = (uint32_t) s.x0 - 3u * bx;
    const float4 *g = P.grid4 + ((size_t) ((uint32_t) s.z0 * (uint32_t) P.ry + (uint32_t) s.y0) * (uint32_t) P.g4_nbx + bx) * 16 + ox;
    float4 d0 = g[0], d1 = g[1], d2 = g[4], d3 = g[5], d4 = g[8], d5 = g[9], d6 = g[12], d7 = g[13];
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

// max |dL|, max |L_in| over the rays of the launch and max |emission| over the grid -> out[0..2] (float bits; zeroed by the caller):
// what the fixed-point units of the window follow from
// out[4]: the largest NEGATIVE density of the grid, as a magnitude (identity activation: a = exp(-sigma dt) > 1 there, throughput and weights can grow)
__global__ void __launch_bounds__(256) nerf_tile_bounds_kernel(const float *dL, const float *L_in, size_t n_ray_floats, const float *em, size_t n_em,
                                                               const float *sig, size_t n_sig, uint32_t *out)
{
    float m[4] = { 0.0f, 0.0f, 0.0f, 0.0f };
    const size_t stride = (size_t) gridDim.x * blockDim.x, i0 = (size_t) blockIdx.x * blockDim.x + threadIdx.x;
    bool bad = false;                                                   // a non-finite input: fixed point cannot carry it - out[3] makes the pass say so
    for (size_t i = i0; i < n_ray_floats; i += stride) {
        const float a = fabsf(dL[i]), b = fabsf(L_in[i]);
        bad = bad || !(a < kInf) || !(b < kInf);
        m[0] = fmaxf(m[0], a); m[1] = fmaxf(m[1], b);
    }
    for (size_t i = i0; i < n_em; i += stride) { const float a = fabsf(em[i]); bad = bad || !(a < kInf); m[2] = fmaxf(m[2], a); }
    for (size_t i = i0; i < n_sig; i += stride) { const float a = sig[i]; bad = bad || !(fabsf(a) < kInf); m[3] = fmaxf(m[3], -a); }
    if (__ballot(bad) && (threadIdx.x & 63) == 0) atomicOr(out + 3, 1u);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) m[k] = fmaxf(m[k], __shfl_down(m[k], off, 64));
        if ((threadIdx.x & 63) == 0 && m[k] > 0.0f) atomicMax(out + (k < 3 ? k : 4), __float_as_uint(m[k]));   // (non-negative floats order like their bits)
    }
}

template <bool G4>
__global__ void __launch_bounds__(DRT_NT_THREADS) nerf_tile_adjoint_kernel(const Params P, const NerfTile T)
{
    constexpr int NT = DRT_NT_THREADS;
    extern __shared__ __attribute__((aligned(16))) unsigned long long win[];   // [4][kWinStore]: sigma_t, r, g, b (two's complement fixed point)
    __shared__ int wctl[16];                                          // [0..2] min, [3..5] max of the waiting splats' corners, [6..8] direction signs, [9..14] footprint of the ray the window moves to
    __shared__ unsigned long long wkey[1];                           // the waiting splat closest to the camera: {distance bits, thread}
    __shared__ uint32_t occ_lds[kOccWords];
    __shared__ uint32_t wgain[2];                                     // negative densities: the workgroup's M1, W (float bits)
    const uint32_t t = threadIdx.x, lane = t & 63u;

    // ---- thread -> ray.  The kernel is bound by the LDS atomic unit, which serves the lanes of ONE add instruction that share an address one after the other
    //      (12 + 2 x (lanes per address - 1) clocks per ds_add_u64, tools/ubench/lds_atomic_conflict_rate.hip), so the map decides how many do:
    //        rays in index order (a pixel's 32 samples side by side in a wave; first version)      ~10 lanes per address, 18 us per march step
    //        round 5: lane = pixel of the 8 x 8 tile, wave = sample (64 pixel centres ~0.6 voxel apart)      4 - 8 lanes per address, launch 15.6 ms
    //        round 6: a wave = the 16 pixels of one stride-2 sub-lattice of the tile x 4 samples - pixels 1.2 voxels apart, a pixel's four samples a
    //                 jittered fraction of the march step apart in depth: SQ_LDS_ADDR_CONFLICT 1.83 -> 0.48 G, SQ_WAIT_INST_LDS 4.49 -> 1.27 G of 32.7 G
    //                 wave-cycles, launch 14.3 ms (2 / 8 / 16 samples per wave: 15.3 / 15.8 / 17.6 ms - beyond four the lanes' texel loads scatter and a
    //                 pixel's samples meet again in depth; profiles/r06_nerf_tile_experiments.txt) -------------------------------------------------------
    const uint32_t tile = blockIdx.x / T.groups, sg = blockIdx.x - tile * T.groups;
    const uint32_t bx = tile % T.tiles_x, by = tile / T.tiles_x;
#if DRT_NT_SPW == 4
    const uint32_t wv = t >> 6, pix = lane & 15u, q = wv & 3u;
    const uint32_t smp = sg * (NT / 64) + 4u * (wv >> 2) + (lane >> 4);
    const uint32_t px = bx * 8u + 2u * (pix & 3u) + (q & 1u), py = by * 8u + 2u * (pix >> 2) + (q >> 1);
#else
    const uint32_t smp = sg * (NT / 64) + (t >> 6);
    const uint32_t px = bx * 8u + (lane & 7u), py = by * 8u + (lane >> 3);
#endif
    bool job = smp < P.spp && px < (uint32_t) P.width && py < (uint32_t) P.height;
    uint64_t i = 0; uint32_t gi = 0;
    if (job) {
        const uint64_t g64 = ((uint64_t) py * (uint32_t) P.width + px) * P.spp + smp;
        gi = (uint32_t) g64;
        job = g64 >= P.ray_offset;
        const uint64_t rel = g64 - P.ray_offset;
        if (P.chunk) { const uint64_t c = rel / P.stride, r = rel - c * P.stride; job = job && r < P.chunk; i = c * P.chunk + r; }
        else i = rel;
        job = job && i >= P.ray_first && i < P.n_rays;
    }
    // fixed-point units: 2^(e - 44) with 2^e >= the bound of a sigma_t splat / of a colour splat (|dL_k| x weight), per WORKGROUP (the window's sums are
    // flushed as floats: every workgroup may count in its own unit)
    //   |ge_k| = |dL_k| |1 - a| T                                        <= Dmax M1,              M1 = max over the queries of max(a, 1) x T
    //   |gs|  <= sum_k |dL_k| (|em_k| dt a T + |result_k| dt a / (a + 1e-10)) <= 3 Dmax dt (Emax M1 + Lmax + Emax W),  W = sum over the queries of |1 - a| T
    //   (|result_k| <= |L_in| + sum |weight| |em_k|), dt <= 2 ext / (N - 1)
    // Without negative densities a <= 1 and T <= 1: M1 <= 1, W <= 1.  NEGATIVE densities under the identity activation (a projected optimisation has
    // none) make a = exp(-sigma dt) > 1 and let throughput and weights grow: the workgroup then marches its rays once for M1 and W (below) before it
    // marches them for the gradients.  (Until round 6 the bound was the launch's worst case exp(2 |sigma|max x diagonal): with a strongly negative
    // region anywhere in the grid the unit came out so coarse that ordinary gradients lost their digits - tests/test_gpu_fuzz.py found it.)
    float unit_s, unit_c; double inv_s, inv_c;
    const float Dmax = __uint_as_float(T.bounds[0]), Lmax = __uint_as_float(T.bounds[1]), Emax = __uint_as_float(T.bounds[2]);
    const float neg = P.nerf_relu ? 0.0f : __uint_as_float(T.bounds[4]);
    const float dt_max = 2.0f * sqrtf((P.bmax[0] - P.bmin[0]) * (P.bmax[0] - P.bmin[0]) + (P.bmax[1] - P.bmin[1]) * (P.bmax[1] - P.bmin[1]) +
                                      (P.bmax[2] - P.bmin[2]) * (P.bmax[2] - P.bmin[2])) / (float) (P.nerf_queries - 1);
    // Non-finite dL / L_in / emission / density values (as a diverged optimisation produces them), or bounds that overflow fp32: fixed point
    // cannot carry them.  The march is skipped and BOTH gradient grids are filled with NaN - every voxel, so that a caller (or a masked
    // all-reduce) that looks at any part of the grids sees that this gradient is void, as it would find NaN in the voxels the record path touches.
    if (T.bounds[3] || !(fabsf(P.scale) * 3.0f * Dmax * (2.0f * Emax + Lmax) * dt_max * 1.001f < kInf) || !(Dmax < kInf)) {
        const float nan = __uint_as_float(0x7fc00000u);
        const size_t nv = (size_t) P.rx * P.ry * P.rz, i0 = (size_t) blockIdx.x * NT + t, stride = (size_t) gridDim.x * NT;
        for (size_t v = i0; v < nv; v += stride) P.g_sigma[v] = nan;
        for (size_t v = i0; v < 3 * nv; v += stride) P.g_albedo[v] = nan;
        return;
    }
    if (__syncthreads_count(job) == 0) return;                       // (a launch over a window of the film: most tiles hold none of its rays)

    for (int w = t; w < 4 * kWinStore; w += NT) win[w] = 0ull;
    const uint32_t *occ = nullptr;
    if (!G4 && P.occ) {
        for (int w = t; w < P.occ_words; w += NT) occ_lds[w] = P.occ[w];
        occ = occ_lds;
    }
    __syncthreads();

    // ---- the ray (nerf.py:67-88) -------------------------------------------------------------------------------
    V3 o = v3(0, 0, 0), d = v3(0, 0, 1);
    float result[3] = { 0, 0, 0 }, dL[3] = { 0, 0, 0 };
    float throughput = 1.0f, step = 0.0f, jit = 0.0f, t_a = 0.0f, ent_t = 0.0f;
    bool active = false;
    if (job) {
        Pcg32 S; S.seed(P.seed, gi);
        const float ux = S.next_1d(), uy = S.next_1d();
        sensor_ray(P, gi / P.spp, ux, uy, o, d);
        result[0] = P.L_in[3 * i]; result[1] = P.L_in[3 * i + 1]; result[2] = P.L_in[3 * i + 2];
        dL[0] = P.dL[3 * i]; dL[1] = P.dL[3 * i + 1]; dL[2] = P.dL[3 * i + 2];
        Hit si = box_hit(P, o, d);
        active = si.valid;
        if (active) {
            ent_t = si.t;
            o = offset_p(si, d);
            si = box_hit(P, o, d);
            active = si.valid;
        }
        if (active) {
            const int N = P.nerf_queries;
            step = P.nerf_jitter ? (si.t - 0.0f) / (float) N : (si.t - 0.0f) / (float) (N - 1);
            jit = S.next_1d();
        }
    }
    // ---- negative densities: this workgroup's M1 and W (the march of the loop below, sigma_t only) ----------------------------------------------
    {
        float M1 = 1.0f, Wm = 1.0f;
        if (neg > 0.0f) {                                               // (workgroup-uniform)
            if (t < 2) wgain[t] = 0u;
            __syncthreads();
            float m1 = 0.0f, W = 0.0f;
            if (active) {
                const int N = P.nerf_queries;
                float thr = 1.0f, ta = 0.0f;
                for (int q = 0; q < N; ++q) {
                    const float t_b = P.nerf_jitter ? step * ((float) (q + 1) + jit) : step * (float) (q + 1);
                    const float dt = t_b - ta;
                    const V3 p = ray_at(o, d, t_b);
                    float raw;
                    if constexpr (G4) {
                        Stencil s4; float em4[3];
                        axis_setup(p.x, P.bmin[0], P.inv_ext[0], P.rx, s4.x0, s4.x1, s4.wx0, s4.wx1);
                        axis_setup(p.y, P.bmin[1], P.inv_ext[1], P.ry, s4.y0, s4.y1, s4.wy0, s4.wy1);
                        axis_setup(p.z, P.bmin[2], P.inv_ext[2], P.rz, s4.z0, s4.z1, s4.wz0, s4.wz1);
                        eval4_at(P, s4, raw, em4);
                    } else raw = eval_sigma_t(P, p, occ);
                    const bool last = !(q + 1 < N);
                    const float a = last ? 1.0f : drt_expf(-raw * dt);
                    m1 = fmaxf(m1, fmaxf(a, 1.0f) * thr);
                    W += fabsf(1.0f - a) * thr;
                    ta = t_b;
                    if (!last) thr *= a + 1e-10f;
                }
                if (!(thr < kInf) || !(W < kInf) || !(m1 < kInf)) m1 = kInf;   // (an overflow, or inf x 0 behind it: this workgroup's gradients are void)
            }
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) { m1 = fmaxf(m1, __shfl_xor(m1, off, 64)); W = fmaxf(W, __shfl_xor(W, off, 64)); }
            if (lane == 0) { atomicMax(wgain, __float_as_uint(m1)); atomicMax(wgain + 1, __float_as_uint(W)); }   // (non-negative floats order like their bits)
            __syncthreads();
            // (workgroup-uniform: kept in scalar registers, like the bounds they multiply)
            M1 = fmaxf(1.0f, __uint_as_float(__builtin_amdgcn_readfirstlane(wgain[0]))) * 1.001f;
            Wm = fmaxf(1.0f, __uint_as_float(__builtin_amdgcn_readfirstlane(wgain[1]))) * 1.001f;
        }
        const float Bs = fabsf(P.scale) * 3.0f * Dmax * (Emax * M1 + Emax * Wm + Lmax) * dt_max * 1.001f, Bc = Dmax * M1;
        if (!(Bs < kInf) || !(Bc < kInf)) {                             // this workgroup's rays overflow fp32: its share of the gradient is void - and so is the whole
            const float nan = __uint_as_float(0x7fc00000u);
            const size_t nv = (size_t) P.rx * P.ry * P.rz;
            for (size_t v = t; v < nv; v += NT) P.g_sigma[v] = nan;
            for (size_t v = t; v < 3 * nv; v += NT) P.g_albedo[v] = nan;
            return;
        }
        int es = 0, ec = 0;
        (void) frexpf(fmaxf(Bs, 1e-30f), &es); (void) frexpf(fmaxf(Bc, 1e-30f), &ec);
        es = max(es - kFixBits, -100); ec = max(ec - kFixBits, -100);
        unit_s = ldexpf(1.0f, es); inv_s = ldexp(1.0, -es); unit_c = ldexpf(1.0f, ec); inv_c = ldexp(1.0, -ec);
    }
    uint32_t n_adds = 0;                                               // (LDS lane-adds of this ray, counting launches only: bounds[6..7])
    int Wx = -(1 << 28), Wy = -(1 << 28), Wz = -(1 << 28);             // window origin (workgroup-uniform; none yet: the first splats all wait)
    const int N = P.nerf_queries;
    int j = 0;
    // the splat a ray holds while the window does not cover it
    bool pend = false, colour = false;
    Stencil st;
    st.x0 = st.x1 = st.y0 = st.y1 = st.z0 = st.z1 = 0; st.wx0 = st.wx1 = st.wy0 = st.wy1 = st.wz0 = st.wz1 = 0.0f;
    float v0 = 0.0f, ge[3] = { 0.0f, 0.0f, 0.0f };
    auto flush = [&]() {                                                // slot -> the voxel it holds under the current origin
        for (int l = t; l < kWinSlots; l += NT) {
            const int sx = l & 15, sy = (l >> 4) & 15, sz = l >> 8, s = sz * kSZ + sy * kSY + sx;
            const unsigned long long a0 = win[s], a1 = win[kWinStore + s], a2 = win[2 * kWinStore + s], a3 = win[3 * kWinStore + s];
            if ((a0 | a1 | a2 | a3) != 0ull) {
                const int x = Wx + ((sx - Wx) & 15), y = Wy + ((sy - Wy) & 15), z = Wz + ((sz - Wz) & 15);
                const size_t lin = ((size_t) z * (size_t) P.ry + (size_t) y) * (size_t) P.rx + (size_t) x;
                if (a0) atomicAdd(P.g_sigma + lin, (float) (long long) a0 * unit_s);
                if (a1) atomicAdd(P.g_albedo + 3 * lin, (float) (long long) a1 * unit_c);
                if (a2) atomicAdd(P.g_albedo + 3 * lin + 1, (float) (long long) a2 * unit_c);
                if (a3) atomicAdd(P.g_albedo + 3 * lin + 2, (float) (long long) a3 * unit_c);
                win[s] = 0ull; win[kWinStore + s] = 0ull; win[2 * kWinStore + s] = 0ull; win[3 * kWinStore + s] = 0ull;
            }
        }
    };

    // ---- the march (nerf.py:94-129), WINDOW-synchronous: every ray runs on by itself - lookup, weights, the splat into the window - until a
    //      splat falls outside the window; when every ray of the workgroup waits (or is done), the window is flushed and moved to the waiting splat
    //      that is closest to the camera (the rays of a tile are nearly parallel: nothing waits behind it), with the slack on the side the rays
    //      move to.  That ray is inside by construction: every phase makes progress, and no splat ever bypasses the window.
    //      (A first version marched all rays in lock-step, one barrier pair per query: 18 us per step - every step waited for the slowest wave's
    //       loads, and for the atomics of the 6 % of splats that the step-synchronous window could not cover.) ---------------------------------------
    for (;;) {
        for (;;) {
            if (pend) {
                if (!(st.x0 >= Wx && st.x1 < Wx + kWin && st.y0 >= Wy && st.y1 < Wy + kWin && st.z0 >= Wz && st.z1 < Wz + kWin)) break;
                float w[8];
                stencil_weights(st, w);
                const int sx0 = st.x0 & 15, sx1 = st.x1 & 15, sy0 = (st.y0 & 15) * kSY, sy1 = (st.y1 & 15) * kSY, sz0 = (st.z0 & 15) * kSZ, sz1 = (st.z1 & 15) * kSZ;
                const int sl[8] = { sz0 + sy0 + sx0, sz0 + sy0 + sx1, sz0 + sy1 + sx0, sz0 + sy1 + sx1,
                                    sz1 + sy0 + sx0, sz1 + sy0 + sx1, sz1 + sy1 + sx0, sz1 + sy1 + sx1 };
                if (v0 != 0.0f) {
#pragma unroll
                    for (int k = 0; k < 8; ++k) atomicAdd(win + sl[k], fix64(w[k] * v0, inv_s));
                    if (T.count) n_adds += 8u;
                }
                if (colour) {
#pragma unroll
                    for (int c = 0; c < 3; ++c) {
                        if (ge[c] != 0.0f) {
#pragma unroll
                            for (int k = 0; k < 8; ++k) atomicAdd(win + (c + 1) * kWinStore + sl[k], fix64(w[k] * ge[c], inv_c));
                            if (T.count) n_adds += 8u;
                        }
                    }
                }
                pend = false;
                NT_STAT(2, 1);
            }
            if (!(active && j < N)) break;
            // query j
            const float t_b = P.nerf_jitter ? step * ((float) (j + 1) + jit) : step * (float) (j + 1);
            const V3 p = ray_at(o, d, t_b);
            // the query's footprint (unscaled indices): the lookup's and, if the query splats, the splat's (no splat waits here: `st` is free)
            axis_setup(p.x, P.bmin[0], P.inv_ext[0], P.rx, st.x0, st.x1, st.wx0, st.wx1);
            axis_setup(p.y, P.bmin[1], P.inv_ext[1], P.ry, st.y0, st.y1, st.wy0, st.wy1);
            axis_setup(p.z, P.bmin[2], P.inv_ext[2], P.rz, st.z0, st.z1, st.wz0, st.wz1);
            float raw, em[3];
            if constexpr (G4) eval4_at(P, st, raw, em);
            else { raw = eval_sigma_t(P, p, occ); eval_rgb(P, P.emission, p, em); }
            const float dt = t_b - t_a;
            const float sigma = P.nerf_relu ? fmaxf(0.0f, raw) : raw;
            const bool last = !(j + 1 < N);
            const float a = last ? 1.0f : drt_expf(-sigma * dt);
            const float weight = (1.0f - a) * throughput;
            const float safe_a = a + 1e-10f;
#pragma unroll
            for (int k = 0; k < 3; ++k) result[k] = result[k] - weight * em[k];
            const float da = last ? 0.0f : -dt * a;
            float gs = 0.0f;
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                gs += dL[k] * (em[k] * (-da * throughput) + (result[k] / safe_a) * da);
                ge[k] = dL[k] * weight;
            }
            if (P.nerf_relu && !(raw > 0.0f)) gs = 0.0f;
            t_a = t_b;
            if (!last) throughput *= safe_a;
            ++j;
            colour = ge[0] != 0.0f || ge[1] != 0.0f || ge[2] != 0.0f;
            if (gs != 0.0f || colour) {                                         // (adding exact zeros changes nothing)
                v0 = gs * P.scale;
                pend = true;
            }
        }
        // ---- every ray waits or is done: the waiting splat closest to the camera, the bounding box of the waiting ones -------------------
        if (t < 8) wctl[t] = t < 3 ? 1 << 28 : t < 6 ? -(1 << 28) : 0;
        if (t == 0) { wkey[0] = ~0ull; NT_STAT(0, 1); }
        __syncthreads();                                                        // (... and the phase's LDS adds are done)
        // (a waiting splat's distance from the camera: ent_t + the t_b of its query, which is t_a by now; distances are positive: ordered like their bits)
        unsigned long long mine = pend ? (((unsigned long long) __float_as_uint(ent_t + t_a) << 32) | t) : ~0ull;
        unsigned long long best = mine;
        int mn[3] = { pend ? st.x0 : 1 << 28, pend ? st.y0 : 1 << 28, pend ? st.z0 : 1 << 28 };
        int mx[3] = { pend ? st.x1 : -(1 << 28), pend ? st.y1 : -(1 << 28), pend ? st.z1 : -(1 << 28) };
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            const unsigned long long o2 = __shfl_xor(best, off, 64);
            best = o2 < best ? o2 : best;
#pragma unroll
            for (int k = 0; k < 3; ++k) { mn[k] = min(mn[k], __shfl_xor(mn[k], off, 64)); mx[k] = max(mx[k], __shfl_xor(mx[k], off, 64)); }
        }
        if (lane == 0 && best != ~0ull) {
            atomicMin(wkey, best);
#pragma unroll
            for (int k = 0; k < 3; ++k) { atomicMin(wctl + k, mn[k]); atomicMax(wctl + 3 + k, mx[k]); }
        }
        flush();
        __syncthreads();
        const unsigned long long win_key = wkey[0];
        if (win_key == ~0ull) break;                                            // nothing waits: every ray is done (the window is flushed)
        if (mine == win_key) {                                                  // the ray the window moves to
            wctl[6] = d.x < 0.0f ? -1 : 1; wctl[7] = d.y < 0.0f ? -1 : 1; wctl[8] = d.z < 0.0f ? -1 : 1;
            wctl[9] = st.x0; wctl[10] = st.x1; wctl[11] = st.y0; wctl[12] = st.y1; wctl[13] = st.z0; wctl[14] = st.z1;
        }
        __syncthreads();
        // per axis: the box's corner on the side the rays come from, moved as far as that ray's footprint allows
        // (workgroup-uniform: scalar registers - 6 vector registers fewer, with the leaner bookkeeping of round 6 119 instead of 127)
        Wx = __builtin_amdgcn_readfirstlane(wctl[6] >= 0 ? max(wctl[0], wctl[10] - (kWin - 1)) : min(wctl[3] - (kWin - 1), wctl[9]));
        Wy = __builtin_amdgcn_readfirstlane(wctl[7] >= 0 ? max(wctl[1], wctl[12] - (kWin - 1)) : min(wctl[4] - (kWin - 1), wctl[11]));
        Wz = __builtin_amdgcn_readfirstlane(wctl[8] >= 0 ? max(wctl[2], wctl[14] - (kWin - 1)) : min(wctl[5] - (kWin - 1), wctl[13]));
        __syncthreads();                                                        // (wctl / wkey are reset by the next phase's end)
    }
    if (T.count && P.counters) {
        // (as nerf_kernel counts: one sigma_t + one colour lookup, one sigma_t + one colour splat per query)
        const uint32_t n_q = (uint32_t) j;                                        // (one lookup and one splat per query of the march)
        uint32_t vals[C_COUNT] = { job && !P.nerf_fused_half ? 1u : 0u, n_q, 0, 0, n_q, 0, 0, n_q, n_q };
#pragma unroll
        for (int s = 0; s < C_COUNT; ++s) {
            uint32_t v = vals[s];
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
            if (lane == 0 && v) atomicAdd(P.counters + s, (unsigned long long) v);
        }
        // the kernel's own ceiling is the LDS atomic rate: lane-adds of this launch (after the zero skips) -> bounds[6..7] (drt_nerf_tile_stats)
        uint32_t a = n_adds;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) a += __shfl_down(a, off, 64);
        if (lane == 0 && a) atomicAdd((unsigned long long *) (T.bounds + 6), (unsigned long long) a);
    }
}

}  // namespace

hipError_t launch_brick_grid4(const float *sigma_t, const float *rgb, float4 *dst, int rx, int ry, int rz, int nbx,
                              hipStream_t stream)
{
    const size_t total = (size_t) nbx * ry * rz * 16;
    hipLaunchKernelGGL(brick_grid4_kernel, dim3((unsigned) ((total + 255) / 256)), dim3(256), 0, stream, sigma_t, rgb, dst, rx, ry, rz, nbx);
    return hipGetLastError();
}

// sensor rays of whole samples-per-pixel groups, at most one workgroup of rays per pixel, a grid the flush can index, every grid on sigma_t's lattice
// (the window's four planes share one footprint; own colour lattice: drt_own.hip's nerf kernel)
bool nerf_tile_supported(const Params &P)
{
    return !P.colour_own && P.sensor_flow && P.spp >= 1 && P.width >= 1 && P.height >= 1 && P.g_sigma && P.g_albedo &&
           (uint64_t) P.width * (uint64_t) P.height * P.spp < (1ull << 32) && (!P.chunk || P.stride >= P.chunk);
}

hipError_t launch_nerf_tile_adjoint(const Params &P, bool g4, bool count, uint32_t *bounds, hipStream_t stream)
{
    if (P.n_rays <= P.ray_first) return hipSuccess;
    if (!nerf_tile_supported(P) || (g4 && !P.grid4) || !bounds || !P.emission) return hipErrorInvalidValue;
    NerfTile T;
    {
        hipError_t e = hipMemsetAsync(bounds, 0, 8 * sizeof(uint32_t), stream);
        if (e != hipSuccess) return e;
        const size_t n_em = (size_t) P.rx * P.ry * P.rz * 3;
        hipLaunchKernelGGL(nerf_tile_bounds_kernel, dim3(2048), dim3(256), 0, stream, P.dL + 3 * P.ray_first, P.L_in + 3 * P.ray_first,
                           (size_t) (P.n_rays - P.ray_first) * 3, P.emission, n_em, P.sigma_t, n_em / 3, bounds);
        T.bounds = bounds;
    }
    T.tiles_x = ((uint32_t) P.width + 7u) / 8u;
    const uint32_t tiles_y = ((uint32_t) P.height + 7u) / 8u;
    T.groups = (P.spp + DRT_NT_THREADS / 64 - 1) / (DRT_NT_THREADS / 64);
    T.g4 = g4 ? 1u : 0u; T.count = count ? 1u : 0u;
    const size_t lds = (size_t) 4 * kWinStore * sizeof(unsigned long long);
    auto set_lds = [&](const void *k) {
        static std::atomic<bool> done[2][64];
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 63;
        if (!done[g4 ? 1 : 0][dev] || dev == 63) {
            const hipError_t e = hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds);
            if (e != hipSuccess) return e;
            done[g4 ? 1 : 0][dev] = true;
        }
        return hipSuccess;
    };
    const dim3 grid(T.tiles_x * tiles_y * T.groups), block(DRT_NT_THREADS);
    if (g4) {
        const hipError_t e = set_lds((const void *) nerf_tile_adjoint_kernel<true>);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(nerf_tile_adjoint_kernel<true>, grid, block, lds, stream, P, T);
    } else {
        const hipError_t e = set_lds((const void *) nerf_tile_adjoint_kernel<false>);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(nerf_tile_adjoint_kernel<false>, grid, block, lds, stream, P, T);
    }
    return hipGetLastError();
}

}  // namespace drt
