// drt_order.hip -- what a launch of the supergrid tracers is told BEFORE it starts (majorant_resolution_factor > 0, the reference's default:
// python/scene_config.py:36): the order in which its rays are started (Params::order, build_super_order: units of one pixel's rays, the
// expensive ones first - a path of depth 64 takes about a millisecond of flight round trips however empty the chip is) and the pixels whose
// rays cross only empty supergrid cells (Params::unit_empty, build_unit_empty).  Schedules and proofs only: they enter no result - a ray's
// numbers depend on (seed, global index, scene) alone (volpathsimple.py:38-290 is traced by drt_sq.hip / drt_coop_tracer.h).
#include "drt_device.h"
#include "drt_launch.h"

namespace drt {

// ---- ray order (Params::order) --------------------------------------------------------------------------------------------
// A path of depth 64 takes about a millisecond of flight round trips however empty the chip is.  In index order the last
// such paths start when the queues run dry, and the launch then waits for them (measured on the headline scene at factor 8:
// launch time = 1.3 ms + 0.089 ms x spp in the primal pass, 1.9 ms + 0.187 ms x spp in the adjoint pass; the constant grows
// with max_depth: 0.5 / 0.9 / 1.3 ms at depth 4 / 16 / 64).  Units of consecutive rays are therefore started by descending
// cost: a stable-per-block counting sort over 64 keys.
namespace {

#ifndef DRT_ORDER_BLOCK
#define DRT_ORDER_BLOCK 2048
#endif
#ifndef DRT_ORDER_SCALE
#define DRT_ORDER_SCALE 8.0f
#endif
#ifndef DRT_ORDER_ITERS_SCALE
#define DRT_ORDER_ITERS_SCALE 2
#endif
constexpr int kOrderKeys = 64, kOrderBlock = DRT_ORDER_BLOCK, kOrderThreads = 256, kOrderFlatBelow = 8;

// key of a unit: the majorant optical depth along the unit's first ray (pixel centre for sensor rays) through the
// supergrid, on a logarithmic scale - paths get long where the medium is thick.  The box is clipped by the slab test; the
// key orders launches, it enters no result.  (Tried for the adjoint pass, where the primal pass of the job
// has counted every ray's bounce-loop iterations: rays sorted one by one by that count - 6.06 instead of 4.94 ms at 16 spp,
// lanes of a wave then move in lock-step and neighbours in the image are torn apart -, and units by their longest ray:
// 7.30 ms against 7.23 ms with this key.)
// `iters` (adjoint launches behind the primal pass of the same job, round 5): Params::ray_iters, the bounce-loop iterations the primal
// pass counted for every ray.  A unit's key is then at least twice the count of its longest main path: measured with the queued tracer
// (tools/finish_age_profile.py, profiles/r05_finish_age.txt) the launch's last paths are paths of 15 - 40 iterations whose rays were
// started in its last fifth - the optical depth along the pixel's ray does not see them coming, the primal pass did.
__global__ void __launch_bounds__(kOrderThreads) order_keys_depth_kernel(const Params P, uint32_t unit, uint32_t units, uint8_t *keys, const uint8_t *iters)
{
    const uint32_t u = blockIdx.x * blockDim.x + threadIdx.x;
    if (u >= units) return;
    const uint64_t i = P.ray_first + (uint64_t) u * unit;
    V3 o, d;
    if (P.sensor_flow) {
        const uint64_t g64 = P.chunk ? P.ray_offset + (i / P.chunk) * P.stride + (i % P.chunk) : P.ray_offset + i;
        sensor_ray(P, (uint32_t) g64 / P.spp, 0.5f, 0.5f, o, d);
    } else {
        o = v3(P.rays_o[3 * i], P.rays_o[3 * i + 1], P.rays_o[3 * i + 2]);
        d = v3(P.rays_d[3 * i], P.rays_d[3 * i + 1], P.rays_d[3 * i + 2]);
    }
    const float oo[3] = { o.x, o.y, o.z }, dd[3] = { d.x, d.y, d.z };
    float t0 = 0.0f, t1 = kInf;
    bool miss = false;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        if (dd[a] != 0.0f) {
            const float rcp = 1.0f / dd[a];
            float ta = (P.bmin[a] - oo[a]) * rcp, tb = (P.bmax[a] - oo[a]) * rcp;
            if (ta > tb) { const float t = ta; ta = tb; tb = t; }
            t0 = fmaxf(t0, ta); t1 = fminf(t1, tb);
        } else if (oo[a] < P.bmin[a] || oo[a] > P.bmax[a]) miss = true;
    }
    // (16 samples of the majorant at the midpoints of equal pieces of the chord: independent loads, one memory latency - a
    //  cell-by-cell walk of dependent loads took longer than the sort it feeds.  Tried: 24 trilinear samples of sigma_t
    //  itself - headline at factor 8 635 instead of 655 Msamples/s: the majorants are what the flights see)
    float od = 0.0f;
    if (!miss && t0 < t1 && t1 < kInf) {
        constexpr int kSamples = 16;
        const int gn[3] = { P.gx, P.gy, P.gz };
        const float dt = (t1 - t0) * (1.0f / kSamples);
        float m[kSamples];
#pragma unroll
        for (int j = 0; j < kSamples; ++j) {
            const float t = fmaf((float) j + 0.5f, dt, t0);
            int c[3];
#pragma unroll
            for (int a = 0; a < 3; ++a) {
                const float gf = ((fmaf(dd[a], t, oo[a]) - P.bmin[a]) * P.inv_ext[a]) * (float) gn[a];
                c[a] = (int) fminf(fmaxf(floorf(gf), 0.0f), (float) (gn[a] - 1));
            }
            m[j] = P.mgrid[((size_t) c[2] * P.gy + c[1]) * P.gx + c[0]];
        }
#pragma unroll
        for (int j = 0; j < kSamples; ++j) od += m[j];
        od *= dt;
    }
    int k = (int) (DRT_ORDER_SCALE * log2f(1.0f + od));
    if (iters) {
        uint32_t longest = 0;
        const uint64_t end = i + unit < P.n_rays ? i + unit : P.n_rays;
        for (uint64_t r = i; r < end; ++r) longest = max(longest, (uint32_t) iters[r]);
        k = max(k, (int) (DRT_ORDER_ITERS_SCALE * longest));
    }
    k = k < 0 ? 0 : k > kOrderKeys - 1 ? kOrderKeys - 1 : k;
    keys[u] = (uint8_t) k;
}

// counts per (key, block), keys by descending cost: hist[(63 - key) * n_blocks + block]
__global__ void __launch_bounds__(kOrderThreads) order_hist_kernel(const uint8_t *keys, uint32_t units, uint32_t n_blocks, uint32_t *hist)
{
    __shared__ uint32_t h[kOrderKeys];
    if (threadIdx.x < kOrderKeys) h[threadIdx.x] = 0u;
    __syncthreads();
    const uint32_t first = blockIdx.x * kOrderBlock;
    for (uint32_t k = threadIdx.x; k < kOrderBlock && first + k < units; k += kOrderThreads) atomicAdd(&h[keys[first + k]], 1u);
    __syncthreads();
    if (threadIdx.x < kOrderKeys) hist[(size_t) (kOrderKeys - 1 - threadIdx.x) * n_blocks + blockIdx.x] = h[threadIdx.x];
}

// exclusive scan of the n counts in place (one workgroup: n = 64 x blocks <= a few hundred thousand)
__global__ void __launch_bounds__(1024) order_scan_kernel(uint32_t *hist, uint32_t n)
{
    __shared__ uint32_t part[1024];
    const uint32_t per = (n + 1023u) / 1024u, lo = threadIdx.x * per, hi = lo + per < n ? lo + per : n;
    uint32_t sum = 0;
    for (uint32_t k = lo; k < hi; ++k) sum += hist[k];
    part[threadIdx.x] = sum;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {
        const uint32_t v = (int) threadIdx.x >= off ? part[threadIdx.x - off] : 0u;
        __syncthreads();
        part[threadIdx.x] += v;
        __syncthreads();
    }
    uint32_t run = part[threadIdx.x] - sum;
    for (uint32_t k = lo; k < hi; ++k) { const uint32_t v = hist[k]; hist[k] = run; run += v; }
}

// units -> their places (inside a block of 2048 units and one key in any order: neighbours stay neighbours)
__global__ void __launch_bounds__(kOrderThreads) order_scatter_kernel(const uint8_t *keys, uint32_t units, uint32_t n_blocks, const uint32_t *offs, uint32_t *order)
{
    __shared__ uint32_t cur[kOrderKeys];
    if (threadIdx.x < kOrderKeys) cur[threadIdx.x] = offs[(size_t) (kOrderKeys - 1 - threadIdx.x) * n_blocks + blockIdx.x];
    __syncthreads();
    const uint32_t first = blockIdx.x * kOrderBlock;
    // no unit reaches optical depth 1 (key 8): paths are short everywhere, nothing to bring forward - index order (an
    // order by chord length made the optimisation loop's launches over its thin starting medium 9 % slower)
    const bool flat = offs[(size_t) (kOrderKeys - kOrderFlatBelow) * n_blocks] == 0u;
    for (uint32_t k = threadIdx.x; k < kOrderBlock && first + k < units; k += kOrderThreads) {
        const uint32_t place = atomicAdd(&cur[keys[first + k]], 1u);
        order[flat ? first + k : place] = first + k;
    }
}

// launches of up to kOrderSmall units (the optimisation loop's: 10^4 pixels of 1024 rays): the three passes in ONE workgroup
// (every launch of the tracer pays for its order: 4 kernels were 3 % of the loop's iteration)
constexpr uint32_t kOrderSmall = 65536;
__global__ void __launch_bounds__(1024) order_small_kernel(const uint8_t *keys, uint32_t units, uint32_t *order)
{
    __shared__ uint32_t h[kOrderKeys], cur[kOrderKeys];
    if (threadIdx.x < kOrderKeys) h[threadIdx.x] = 0u;
    __syncthreads();
    for (uint32_t k = threadIdx.x; k < units; k += 1024u) atomicAdd(&h[keys[k]], 1u);
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t run = 0;
        for (int key = kOrderKeys - 1; key >= 0; --key) { cur[key] = run; run += h[key]; }
    }
    __syncthreads();
    const bool flat = cur[kOrderFlatBelow - 1] == 0u;            // = units with key >= kOrderFlatBelow
    for (uint32_t k0 = 0; k0 < units; k0 += 1024u) {              // (round by round: neighbours stay neighbours)
        const uint32_t k = k0 + threadIdx.x;
        if (k < units) {
            const uint32_t place = atomicAdd(&cur[keys[k]], 1u);
            order[flat ? k : place] = k;
        }
        __syncthreads();
    }
}

}  // namespace

// ---- pixels whose rays cross only EMPTY supergrid cells (Params::unit_empty, round 5) ------------------------------------------------
// Measured on the headline scene (tools/finish_age_profile.py, profiles/r05_finish_age.txt): 37 % of the camera rays enter the medium's
// box and leave it through cells whose majorant is 0 - no collision is possible, yet each of them is started, walked cell by cell through
// ~45 cells, taken through a collision batch and a transition batch (and, in the adjoint pass, walked once more by the DRT sampler along
// the same segment): the launch's last quarter.  A flight through empty cells accumulates an optical depth of exactly 0 and ends by
// leaving the segment whatever its target depth: the flight set-up of the queued tracer ends such a flight at once (its early-out, the
// same code path as "target depth > largest majorant x length"), bit for bit what the walk would have returned.
// The proof is per PIXEL (sensor rays; one thread each): the pixel's centre ray is sampled every half cell along its chord through the box
// grown by one cell, and every sample's cell and its 26 neighbours must be empty.  Any point of any ray through the pixel that lies inside
// the box is within a quarter cell (the footprint bound below) of the centre ray's point at the same distance, which lies inside the grown
// box and within a quarter cell of a sample: its cell is one of the 27.  Rays of wider pixels, or explicit rays, get no flag.
__global__ void __launch_bounds__(256) unit_empty_kernel(const Params P, uint32_t unit, uint32_t units, uint8_t *flags)
{
    const uint32_t u = blockIdx.x * blockDim.x + threadIdx.x;
    if (u >= units) return;
    const uint64_t i = P.ray_first + (uint64_t) u * unit;
    const uint64_t g64 = P.chunk ? P.ray_offset + (i / P.chunk) * P.stride + (i % P.chunk) : P.ray_offset + i;
    V3 o, d;
    sensor_ray(P, (uint32_t) (g64 / P.spp), 0.5f, 0.5f, o, d);
    const float oo[3] = { o.x, o.y, o.z }, dd[3] = { d.x, d.y, d.z };
    const int gn[3] = { P.gx, P.gy, P.gz };
    float cell[3], cmin = kInf;
#pragma unroll
    for (int a = 0; a < 3; ++a) { cell[a] = (P.bmax[a] - P.bmin[a]) / (float) gn[a]; cmin = fminf(cmin, cell[a]); }
    float t0 = 0.0f, t1 = kInf;
    bool miss = false;
#pragma unroll
    for (int a = 0; a < 3; ++a) {                                   // the box grown by one cell
        const float lo = P.bmin[a] - cell[a], hi = P.bmax[a] + cell[a];
        if (dd[a] != 0.0f) {
            const float rcp = 1.0f / dd[a];
            float ta = (lo - oo[a]) * rcp, tb = (hi - oo[a]) * rcp;
            if (ta > tb) { const float t = ta; ta = tb; tb = t; }
            t0 = fmaxf(t0, ta); t1 = fminf(t1, tb);
        } else if (oo[a] < lo || oo[a] > hi) miss = true;
    }
    uint8_t empty = 0;
    // footprint: a ray through the pixel deviates from the centre ray by at most the pixel's half diagonal (in tangent units, generous);
    // judged at the far end of the grown box along this ray, or - if the centre ray misses it - at the box's farthest corner
    const float spread = 1.5f * (P.tan_x / (float) P.width + P.tan_y / (float) P.height);
    float far2 = 0.0f;
#pragma unroll
    for (int a = 0; a < 3; ++a) { const float e = fmaxf(fabsf(P.bmin[a] - cell[a] - oo[a]), fabsf(P.bmax[a] + cell[a] - oo[a])); far2 += e * e; }
    const bool crosses = !miss && t0 < t1 && t1 < kInf;
    if ((crosses ? t1 : sqrtf(far2)) * spread <= 0.25f * cmin) {
        bool any = false;
        if (crosses) {
            const float step = 0.5f * cmin;
            const int n = (int) ((t1 - t0) / step) + 2;
            for (int j = 0; j < n && !any; ++j) {
                const float t = fminf(t0 + (float) j * step, t1);
                int c[3];
#pragma unroll
                for (int a = 0; a < 3; ++a) {
                    const float gf = ((fmaf(dd[a], t, oo[a]) - P.bmin[a]) * P.inv_ext[a]) * (float) gn[a];
                    c[a] = (int) fminf(fmaxf(floorf(gf), 0.0f), (float) (gn[a] - 1));
                }
                const uint32_t cc = (uint32_t) ((c[2] * gn[1] + c[1]) * gn[0] + c[0]);
                any = ((P.mocc_dil[cc >> 5] >> (cc & 31u)) & 1u) != 0u;          // the sample's cell and its 26 neighbours
            }
        }
        // (a centre ray that misses the grown box: no ray of the pixel has a point inside the box - they all escape at once)
        empty = any ? 0 : 1;
    }
    flags[u] = empty;
}

hipError_t build_unit_empty(const Params &P, uint32_t unit, uint32_t units, uint8_t *flags, hipStream_t stream)
{
    if (!units) return hipSuccess;
    hipLaunchKernelGGL(unit_empty_kernel, dim3((units + 255) / 256), dim3(256), 0, stream, P, unit, units, flags);
    return hipGetLastError();
}

static inline size_t order_align(size_t n) { return (n + 255) & ~(size_t) 255; }

size_t super_order_bytes(uint32_t units)
{
    const size_t n_blocks = ((size_t) units + kOrderBlock - 1) / kOrderBlock;
    return order_align((size_t) units * 4) + order_align((size_t) units) + order_align(n_blocks * kOrderKeys * 4);
}

hipError_t build_super_order(const Params &P, uint32_t unit, uint32_t units, void *work, hipStream_t stream, const uint8_t *iters)
{
    if (!units) return hipSuccess;
    const uint32_t n_blocks = (units + kOrderBlock - 1) / kOrderBlock;
    uint32_t *order = (uint32_t *) work;
    uint8_t *keys = (uint8_t *) work + order_align((size_t) units * 4);
    uint32_t *hist = (uint32_t *) (keys + order_align((size_t) units));
    const unsigned key_blocks = (units + kOrderThreads - 1) / kOrderThreads;
    hipLaunchKernelGGL(order_keys_depth_kernel, dim3(key_blocks), dim3(kOrderThreads), 0, stream, P, unit, units, keys, iters);
    if (units <= kOrderSmall) {
        hipLaunchKernelGGL(order_small_kernel, dim3(1), dim3(1024), 0, stream, keys, units, order);
        return hipGetLastError();
    }
    hipLaunchKernelGGL(order_hist_kernel, dim3(n_blocks), dim3(kOrderThreads), 0, stream, keys, units, n_blocks, hist);
    hipLaunchKernelGGL(order_scan_kernel, dim3(1), dim3(1024), 0, stream, hist, n_blocks * (uint32_t) kOrderKeys);
    hipLaunchKernelGGL(order_scatter_kernel, dim3(n_blocks), dim3(kOrderThreads), 0, stream, keys, units, n_blocks, hist, order);
    return hipGetLastError();
}

}  // namespace drt
