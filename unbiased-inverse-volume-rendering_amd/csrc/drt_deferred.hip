// drt_deferred.hip -- deferred, tile-binned gradient splatting for the adjoint pass.
//
// The adjoint tracer emits ~14 trilinear splats per ray (volpathsimple.py:170,489,580,594,607).  As
// global fp32 atomics they cost one L2 atomic request per splat and plane, and the chip retires only
// ~21 G requests/s: 8.6 of the adjoint's 21.8 ms.  Here the tracer appends
// records to two streams instead (emit_record, drt_device.h) - stream 0: 16 bytes {p, g_sigma_t}; stream 1:
// 32 bytes {p, g_sigma_t, g_r, g_g, g_b} for the splats that carry all four channels at one point (scatter
// events, nerf queries) - and this file turns them into gradients with streaming passes only:
//
//   bin_histogram  records -> per-(workgroup, tile) counts            (LDS histogram)
//   bin_offsets    counts  -> exclusive offsets, tile bases, reduce units
//   bin_scatter    records -> tile-sorted copy                         (LDS cursors)
//   tile_reduce    workgroups walk runs of <= kUnitRecords-record units of ONE tile and ONE gradient plane
//                  (stream 0 -> sigma_t; stream 1 -> sigma_t, r, g, b: four passes over the sorted records, which
//                  stay in the memory-side cache): trilinear weights recomputed (same axis_setup / stencil_weights
//                  as the lookup), 8 LDS adds per record into a (32+1)x(16+1)x(16+1) tile, then one coalesced
//                  flush of the non-zero tile entries into the caller's gradient grid (+=, fp32 atomics: ~10^7
//                  per launch instead of ~10^9 lane-atomics).
//
// The LDS accumulators are 64-bit FIXED POINT: on gfx950 ds_add_f32 retires 0.2 T lane-atomics/s but
// ds_add_u64 3.4 T/s (tools/ubench/lds_atomic_rate.hip; the fp32 version of this kernel took 5.3 ms,
// 4.9 of them in ds_add_f32).  Every float product w*v is scaled by a power of two chosen from the
// plane's max |v| (found by the histogram pass) so that |w*v| * scale < 2^30 and rounded to an
// integer: quantisation 2^-31 max|v| per add, exact and order-independent sums inside a tile.
// A non-finite value poisons its plane's maximum: the tile then takes a float path that PROPAGATES the
// NaN / inf into the gradient exactly as the atomic path would (no silent clamping).
//
// A tile is addressed by the BASE corner cell of the splat, so its footprint is the tile plus a
// one-voxel apron on the high side.  The sigma_t values already carry the medium `scale`.  Sum order differs
// from the atomic path (as it does between two runs of the atomic path); the parity tolerance on gradients
// is unchanged.
#include <atomic>
#include "drt_device.h"
#include "drt_launch.h"

#ifndef DRT_PART_NT
#define DRT_PART_NT 5              // non-temporal accesses of the passes that stream the records: bit 0 the histogram's and the scatter's loads of the emitted records,
                                   // bit 2 tile_reduce's loads of stream 0's sorted records (read once; stream 1's are read by four planes and stay cached),
                                   // bit 1 the scatter's stores.  Measured, alternating runs on one box (headline Msamples/s / reductions ms; default = 0):
                                   // 0: 939-943 / 1.99   1: 948-962 / 1.82-1.95   4: 947 / 1.94   5: 955-960 / 1.76-1.88   2: 883 / 2.54   3: 886 / 2.52
                                   // - the loads leave the L2 to the scatter's open output lines; non-temporal STORES lose their write combining
#endif
#ifndef DRT_PART_UNROLL
#define DRT_PART_UNROLL 4
#endif

namespace drt {

namespace {

typedef float nt_f4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 nt_load4(const float4 *p) { const nt_f4 v = __builtin_nontemporal_load((const nt_f4 *) p); return make_float4(v.x, v.y, v.z, v.w); }
__device__ __forceinline__ void nt_store4(float4 v, float4 *p) { nt_f4 w = { v.x, v.y, v.z, v.w }; __builtin_nontemporal_store(w, (nt_f4 *) p); }

constexpr int kLdsTile = (kTileX + 1) * (kTileY + 1) * (kTileZ + 1);

// Workgroup barrier AFTER no-return LDS atomics.  __syncthreads() alone compiles to a bare s_barrier here
// (the memory model treats LDS as in-order and waits for nothing), and the reads that follow were
// observed to overtake ds_add_u64 operations still in flight: a few records' worth of gradient lost per
// launch, one launch in ten.  Drain this wave's LDS queue first.
__device__ __forceinline__ void lds_atomics_barrier()
{
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __syncthreads();
}
constexpr int kPartUnroll = DRT_PART_UNROLL;   // chunks a partition thread keeps in flight
constexpr uint32_t kRideRecords = 4096;  // stream-1 records per tile that plane 0 takes along (a quarter of a reduce unit)
constexpr uint32_t kReduceWGs = 1024;   // per plane: 4 workgroups per CU, looping over the reduce units (512 .. 4096: the same)

struct Cell { int x0, x1, y0, y1, z0, z1; float w[8]; };

// base corner cell + weights of a record position: the SAME arithmetic as the lookups / the atomic path
__device__ __forceinline__ void cell_of(const Params &P, float px, float py, float pz, int &x0, int &y0, int &z0)
{
    int i1; float w0, w1;
    axis_setup(px, P.bmin[0], P.inv_ext[0], P.rx, x0, i1, w0, w1);
    axis_setup(py, P.bmin[1], P.inv_ext[1], P.ry, y0, i1, w0, w1);
    axis_setup(pz, P.bmin[2], P.inv_ext[2], P.rz, z0, i1, w0, w1);
}

__device__ __forceinline__ int bin_of(const Params &P, const DeferredPlan &D, float4 r)
{
    int x0, y0, z0;
    cell_of(P, r.x, r.y, r.z, x0, y0, z0);
    return ((z0 / kTileZ) * D.nty + (y0 / kTileY)) * D.ntx + (x0 / kTileX);
}

// max |v| over finite values; a non-finite value returns +inf bits so that the plane is reduced on the
// propagating float path (tile_reduce)
__device__ __forceinline__ float plane_max(float vmax, float v)
{
    const float a = fabsf(v);
    if (!(a <= 3.0e38f)) return kInf;                            // inf or NaN
    return a > vmax ? a : vmax;
}

// part 0: every chunk; part 1: the chunks below the split (D.cursor[2 + S], launch_deferred_split) - what the adjoint
// tracer's main launch wrote -; part 2: the chunks from the split on, ADDED to part 1's counts
template <int S>
__device__ __forceinline__ void bin_histogram_stream(const Params &P, const DeferredPlan &D, uint32_t *h, int part)
{
    constexpr int kPlanes = S == 0 ? 1 : 4, kQuads = S == 0 ? 1 : 2;
    for (int b = threadIdx.x; b < D.n_bins; b += blockDim.x) h[b] = 0;
    __syncthreads();
    const uint32_t all = min(D.cursor[S], D.cap_chunks[S]), split = min(D.cursor[2 + S], all);
    const uint32_t lo = part == 2 ? split : 0u, used = part == 1 ? split : all;
    float vmax[kPlanes];
#pragma unroll
    for (int c = 0; c < kPlanes; ++c) vmax[c] = 0.0f;
    static_assert(kRecChunk == 256, "one record per thread and chunk");
    // chunk -> workgroup map (shared with the scatter pass): workgroup g owns chunks g*kSub + sub (mod stride)
    constexpr uint32_t kSub = kPartThreads / 256;                  // chunks a workgroup reads side by side
    const uint32_t sub = threadIdx.x >> 8, rec = threadIdx.x & 255u;
    for (uint32_t c0 = blockIdx.x * kSub + sub; c0 < used; c0 += kPartUnroll * kSub * gridDim.x) {
        float4 r[kPartUnroll], q[kPartUnroll]; bool ok[kPartUnroll];
#pragma unroll
        for (int k = 0; k < kPartUnroll; ++k) {                   // kPartUnroll chunks in flight per thread
            const uint32_t c = c0 + k * kSub * gridDim.x;
            ok[k] = c >= lo && c < used && rec < D.chunk_count[S][c];
            if (ok[k]) {
                const float4 *src = D.in[S] + ((size_t) c * kRecChunk + rec) * kQuads;
                r[k] = DRT_PART_NT & 1 ? nt_load4(src) : src[0];
                if constexpr (S == 1) q[k] = DRT_PART_NT & 1 ? nt_load4(src + 1) : src[1];
            }
        }
#pragma unroll
        for (int k = 0; k < kPartUnroll; ++k) {
            if (!ok[k]) continue;
            atomicAdd(&h[bin_of(P, D, r[k])], 1u);
            vmax[0] = plane_max(vmax[0], r[k].w);
            if constexpr (S == 1) { vmax[1] = plane_max(vmax[1], q[k].x); vmax[2] = plane_max(vmax[2], q[k].y); vmax[3] = plane_max(vmax[3], q[k].z); }
        }
    }
    __shared__ uint32_t wg_vmax[4];                              // one global atomic per workgroup and plane, not per wave
    if (threadIdx.x < 4) wg_vmax[threadIdx.x] = 0u;
    __syncthreads();
#pragma unroll
    for (int c = 0; c < kPlanes; ++c) {
        float v = vmax[c];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) v = fmaxf(v, __shfl_down(v, off, 64));
        if ((threadIdx.x & 63) == 0 && v > 0.0f) atomicMax(&wg_vmax[c], __float_as_uint(v));
    }
    lds_atomics_barrier();
    if (threadIdx.x < kPlanes && wg_vmax[threadIdx.x]) atomicMax(D.vmax + (S == 0 ? 0 : 1) + threadIdx.x, wg_vmax[threadIdx.x]);
    uint32_t *dst = D.hist + ((size_t) S * gridDim.x + blockIdx.x) * D.n_bins;
    if (part == 2) { for (int b = threadIdx.x; b < D.n_bins; b += blockDim.x) dst[b] += h[b]; }
    else for (int b = threadIdx.x; b < D.n_bins; b += blockDim.x) dst[b] = h[b];
}

// per tile: exclusive prefix over the partition workgroups (in place) and the tile total.  One wave
// per (stream, tile): lane l owns workgroups [l * per, (l + 1) * per) - independent loads, a wave scan
// of the lane sums, independent stores.
__global__ void __launch_bounds__(256) bin_offsets_kernel(const DeferredPlan D, int n_wgs)
{
    const int lane = threadIdx.x & 63, s = blockIdx.y;
    const int b = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (b >= D.n_bins) return;
    uint32_t *col = D.hist + (size_t) s * n_wgs * D.n_bins + b;
    constexpr int kPer = kPartWGs / 64;
    uint32_t v[kPer], sum = 0;
#pragma unroll
    for (int k = 0; k < kPer; ++k) { v[k] = col[(size_t) (lane * kPer + k) * D.n_bins]; sum += v[k]; }
    uint32_t incl = sum;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) { uint32_t t = __shfl_up(incl, off, 64); if (lane >= off) incl += t; }
    uint32_t run = incl - sum;
#pragma unroll
    for (int k = 0; k < kPer; ++k) { col[(size_t) (lane * kPer + k) * D.n_bins] = run; run += v[k]; }
    if (lane == 63) D.bin_base[(size_t) s * (D.n_bins + 1) + b] = incl;      // total for now
}

// one workgroup per stream: exclusive scan of the tile totals -> bin_base, and of the reduce-unit counts
__global__ void __launch_bounds__(1024) bin_scan_kernel(const DeferredPlan D)
{
    __shared__ uint32_t sa[1024], sb[1024];
    const int s = blockIdx.x, t = threadIdx.x;
    uint32_t *base = D.bin_base + (size_t) s * (D.n_bins + 1);
    uint32_t *ustart = D.unit_start + (size_t) s * (D.n_bins + 1);
    const int per = (D.n_bins + 1023) / 1024;
    uint32_t tot[kMaxBins / 1024], ua = 0, ub = 0;
    for (int k = 0; k < per; ++k) {
        const int b = t * per + k;
        tot[k] = b < D.n_bins ? base[b] : 0u;
        ua += tot[k]; ub += (tot[k] + kUnitRecords - 1) / kUnitRecords;
    }
    sa[t] = ua; sb[t] = ub;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {
        uint32_t a = t >= off ? sa[t - off] : 0u, b = t >= off ? sb[t - off] : 0u;
        __syncthreads();
        sa[t] += a; sb[t] += b;
        __syncthreads();
    }
    uint32_t ra = sa[t] - ua, rb = sb[t] - ub;                    // exclusive
    for (int k = 0; k < per; ++k) {
        const int b = t * per + k;
        if (b < D.n_bins) { base[b] = ra; ustart[b] = rb; }
        ra += tot[k]; rb += (tot[k] + kUnitRecords - 1) / kUnitRecords;
    }
    if (t == 1023) { base[D.n_bins] = sa[t]; ustart[D.n_bins] = sb[t]; }
}

// both streams in one launch (blockIdx.y): the small stream's workgroups fill the tail of the large one's
__global__ void __launch_bounds__(kPartThreads) bin_histogram_kernel(const Params P, const DeferredPlan D, int part)
{
    extern __shared__ uint32_t h[];
    if (blockIdx.y == 0) bin_histogram_stream<0>(P, D, h, part); else bin_histogram_stream<1>(P, D, h, part);
}

// snapshot of the chunk cursors: the chunks below it are complete once the launch that precedes this kernel has ended
__global__ void rec_split_kernel(const DeferredPlan D)
{
    if (threadIdx.x < kRecStreams) D.cursor[2 + threadIdx.x] = min(D.cursor[threadIdx.x], D.cap_chunks[threadIdx.x]);
}

template <int S>
__device__ __forceinline__ void bin_scatter_stream(const Params &P, const DeferredPlan &D, uint32_t *cur)
{
    constexpr int kQuads = S == 0 ? 1 : 2;
    const uint32_t *off = D.hist + ((size_t) S * gridDim.x + blockIdx.x) * D.n_bins;
    const uint32_t *base = D.bin_base + (size_t) S * (D.n_bins + 1);
    for (int b = threadIdx.x; b < D.n_bins; b += blockDim.x) cur[b] = base[b] + off[b];
    __syncthreads();
    const uint32_t used = min(D.cursor[S], D.cap_chunks[S]);
    float4 *dst = D.out[S];
    constexpr uint32_t kSub = kPartThreads / 256;
    const uint32_t sub = threadIdx.x >> 8, rec = threadIdx.x & 255u;
    for (uint32_t c0 = blockIdx.x * kSub + sub; c0 < used; c0 += kPartUnroll * kSub * gridDim.x) {   // same map as the histogram
        float4 r[kPartUnroll], q[kPartUnroll]; bool ok[kPartUnroll];
#pragma unroll
        for (int k = 0; k < kPartUnroll; ++k) { r[k] = make_float4(0.f, 0.f, 0.f, 0.f); q[k] = r[k]; }
#pragma unroll
        for (int k = 0; k < kPartUnroll; ++k) {
            const uint32_t c = c0 + k * kSub * gridDim.x;
            ok[k] = c < used && rec < D.chunk_count[S][c];
            if (ok[k]) {
                const float4 *src = D.in[S] + ((size_t) c * kRecChunk + rec) * kQuads;
                r[k] = DRT_PART_NT & 1 ? nt_load4(src) : src[0];
                if constexpr (S == 1) q[k] = DRT_PART_NT & 1 ? nt_load4(src + 1) : src[1];
            }
        }
        uint32_t slot[kPartUnroll];
#pragma unroll
        for (int k = 0; k < kPartUnroll; ++k) slot[k] = ok[k] ? atomicAdd(&cur[bin_of(P, D, r[k])], 1u) : 0u;
#pragma unroll
        for (int k = 0; k < kPartUnroll; ++k) if (ok[k]) {
            float4 *o = dst + (size_t) slot[k] * kQuads;
            if (DRT_PART_NT & 2) { nt_store4(r[k], o); if constexpr (S == 1) nt_store4(q[k], o + 1); }
            else { o[0] = r[k]; if constexpr (S == 1) o[1] = q[k]; }
        }
    }
}

__global__ void __launch_bounds__(kPartThreads) bin_scatter_kernel(const Params P, const DeferredPlan D)
{
    extern __shared__ uint32_t cur[];
    if (blockIdx.y == 0) bin_scatter_stream<0>(P, D, cur); else bin_scatter_stream<1>(P, D, cur);
}

// blockIdx.y = reduce plane: 0: sigma_t of stream 0 AND of stream 1 (for every tile that has stream-0 records: the
// workgroup that starts such a tile's first unit also adds stream 1's sigma_t values of that tile - one zero / flush
// of the LDS tile instead of two); 1: sigma_t of stream 1 for the tiles WITHOUT stream-0 records; 2..4: r, g, b of stream 1
#ifndef DRT_REDUCE_TRANSPOSE
#define DRT_REDUCE_TRANSPOSE 1     // 0: every unit in arrival order
#endif
#ifndef DRT_REDUCE_THREADS
#define DRT_REDUCE_THREADS 512      // (256: 1.15 ms per headline launch, 512 / 1024: 0.85 ms - more waves per CU behind the LDS tile)
#endif
__global__ void __launch_bounds__(DRT_REDUCE_THREADS) tile_reduce_kernel(const Params P, const DeferredPlan D)
{
    extern __shared__ unsigned long long tile[];                 // kLdsTile signed 64-bit fixed-point accumulators
    const int plane = blockIdx.y;
    const int s = plane == 0 ? 0 : 1, ch = plane == 0 ? 0 : plane - 1;        // stream, channel of the record
    const int quads = s == 0 ? 1 : 2;
    // scale = 2^(30 - e) with max|v| < 2^e: every product |w * v| * scale < 2^30 is rounded to a 32-bit integer (one
    // multiply + one conversion; the 64-bit route through double cost ~12 instructions per corner, i.e. most of this
    // kernel's VALU time) and sign-extended into the 64-bit accumulator, which holds 2^33 such adds.  Quantisation
    // 2^-31 max|v| per add (an fp32 atomic add rounds at 2^-24 of the running sum).  The two sigma_t planes share one
    // scale (plane 0 adds values of both streams)
    const float vmax = plane <= 1 ? fmaxf(__uint_as_float(D.vmax[0]), __uint_as_float(D.vmax[1])) : __uint_as_float(D.vmax[plane]);
    if (vmax == 0.0f) return;                                    // nothing but zeros in this plane
    const bool finite = vmax <= 3.0e38f;                         // a NaN / inf anywhere in the plane: float path below
    int e = 0;
    (void) frexpf(finite ? vmax : 1.0f, &e);
    const float scale = ldexpf(1.0f, 30 - e);
    const double inv_scale = ldexp(1.0, e - 30);
    const uint32_t *base = D.bin_base + (size_t) s * (D.n_bins + 1);
    const uint32_t *ustart = D.unit_start + (size_t) s * (D.n_bins + 1);
    const uint32_t *base0 = D.bin_base, *base1 = D.bin_base + (D.n_bins + 1);
    const uint32_t n_units = ustart[D.n_bins];
    // persistent workgroups (empty ones are not free to dispatch), each a CONTIGUOUS run of units: the
    // units of a heavy tile follow one another, so the LDS tile is zeroed / flushed once per run
    const uint32_t u_begin = (uint32_t) (((uint64_t) n_units * blockIdx.x) / gridDim.x);
    const uint32_t u_end = (uint32_t) (((uint64_t) n_units * (blockIdx.x + 1)) / gridDim.x);
    float *dst = ch == 0 ? P.g_sigma : P.g_albedo + (ch - 1);
    const int stride = ch == 0 ? 1 : 3;
    const float4 *src = D.out[s];
    int b = -1, X0 = 0, Y0 = 0, Z0 = 0;
    bool live = false;                                            // the LDS tile holds sums of tile b that must be flushed

    // one record {position r.xyz, value}: 8 LDS adds (or, on the non-finite path, 8 float atomics into the grid)
    auto add_record = [&](const float4 r, const float val) {
        if (val == 0.0f) return;                                  // adding exact zeros changes nothing
        Stencil st;
        axis_setup(r.x, P.bmin[0], P.inv_ext[0], P.rx, st.x0, st.x1, st.wx0, st.wx1);
        axis_setup(r.y, P.bmin[1], P.inv_ext[1], P.ry, st.y0, st.y1, st.wy0, st.wy1);
        axis_setup(r.z, P.bmin[2], P.inv_ext[2], P.rz, st.z0, st.z1, st.wz0, st.wz1);
        float w[8];
        stencil_weights(st, w);
        if (!finite) {
            // a non-finite gradient value somewhere in this plane: add the float products straight to the
            // caller's grid, as the atomic path does - NaN / inf propagate instead of being clamped away
            const int gx[2] = { st.x0, st.x1 }, gy[2] = { st.y0, st.y1 }, gz[2] = { st.z0, st.z1 };
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                const size_t vox = ((size_t) gz[c >> 2] * P.ry + gy[(c >> 1) & 1]) * P.rx + gx[c & 1];
                atomicAdd(dst + (size_t) stride * vox, w[c] * val);
            }
            return;
        }
        const int x0 = st.x0 - X0, x1 = st.x1 - X0;
        const int y0 = (st.y0 - Y0) * (kTileX + 1), y1 = (st.y1 - Y0) * (kTileX + 1);
        const int z0 = (st.z0 - Z0) * ((kTileX + 1) * (kTileY + 1)), z1 = (st.z1 - Z0) * ((kTileX + 1) * (kTileY + 1));
        const int o[8] = { z0 + y0 + x0, z0 + y0 + x1, z0 + y1 + x0, z0 + y1 + x1,
                           z1 + y0 + x0, z1 + y0 + x1, z1 + y1 + x0, z1 + y1 + x1 };
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            const float pv = w[c] * val;                           // the float product the atomic path adds (|pv| <= vmax)
            atomicAdd(&tile[o[c]], (unsigned long long) (long long) __float2int_rn(pv * scale));
        }
    };

    for (uint32_t u = u_begin; u <= u_end; ++u) {
        int nb = -1;
        if (u < u_end) {
            int lo = b < 0 ? 0 : b, hi = D.n_bins;                // largest tile with ustart[tile] <= u
            while (hi - lo > 1) { int mid = (lo + hi) >> 1; if (ustart[mid] <= u) lo = mid; else hi = mid; }
            nb = lo;
        }
        if (nb == b && nb < 0) break;                             // no units at all for this workgroup
        if (nb != b) {
            if (b >= 0 && live && finite && !dbg(P.debug_flags, 1024u)) {   // flush the finished tile (+=)
                lds_atomics_barrier();
                for (int j = threadIdx.x; j < kLdsTile; j += blockDim.x) {
                    const long long q = (long long) tile[j];
                    if (q == 0) continue;
                    const float v = (float) ((double) q * inv_scale);
                    const int lx = j % (kTileX + 1), ly = (j / (kTileX + 1)) % (kTileY + 1), lz = j / ((kTileX + 1) * (kTileY + 1));
                    const size_t vox = ((size_t) (Z0 + lz) * P.ry + (Y0 + ly)) * P.rx + (X0 + lx);
                    atomicAdd(dst + (size_t) stride * vox, v);
                }
                __syncthreads();
            }
            if (nb < 0) break;
            b = nb;
            X0 = (b % D.ntx) * kTileX; Y0 = ((b / D.ntx) % D.nty) * kTileY; Z0 = (b / (D.ntx * D.nty)) * kTileZ;
            // stream 1's sigma_t values of tile b ride along with stream 0's when they are few (scatter events next to
            // the transmittance splats); when they are many (nerf queries) they keep their own plane and units
            const bool ride = base0[b + 1] > base0[b] && base1[b + 1] - base1[b] <= kRideRecords;
            live = !(plane == 1 && ride);
            if (live && finite) {
                for (int j = threadIdx.x; j < kLdsTile; j += blockDim.x) tile[j] = 0ull;
                __syncthreads();
            }
            if (plane == 0 && ride && u == ustart[b]) {           // first unit of this tile: stream 1's sigma_t values come along
                const float4 *src1 = D.out[1];
                for (uint32_t i1 = base1[b] + threadIdx.x; i1 < base1[b + 1]; i1 += blockDim.x) {
                    const float4 r = src1[2 * (size_t) i1];
                    add_record(r, r.w);
                }
            }
        }
        if (!live) continue;
        const uint32_t first = base[b] + (u - ustart[b]) * kUnitRecords;
        const uint32_t last = min(first + kUnitRecords, base[b + 1]);
        // Records that arrive together may share their voxels - one march step of neighbouring nerf rays: 64 lanes on the same
        // LDS words, the adds serialise (nerf: 3.75 ms per plane and sub-batch).  Every wave looks at the unit's first 64 records
        // (the same records, so the same answer in every wave): when a quarter of them fall into their neighbour's cell the unit
        // is read TRANSPOSED - one contiguous segment per thread, so that the records a wave adds together lie n_rows apart in
        // arrival order (nerf 174 -> 203, fused nerf + DRT 142 -> 163 Msamples/s).  Scatter-event records do not (headline:
        // coalesced order 2.42 ms, transposed 2.66 ms).
        bool transposed = false;
        if (DRT_REDUCE_TRANSPOSE && last - first >= 4096u) {
            const uint32_t lane = threadIdx.x & 63u;
            const float4 r = src[(size_t) (first + lane) * quads];
            int x0, y0, z0;
            cell_of(P, r.x, r.y, r.z, x0, y0, z0);
            const int id = (z0 * P.ry + y0) * P.rx + x0, left = __shfl_up(id, 1, 64);
            transposed = __popcll(__ballot(lane > 0u && id == left)) >= 16;
        }
        if (transposed) {
            const uint32_t n_rows = (last - first + blockDim.x - 1) / blockDim.x;
            for (uint32_t r0 = 0; r0 < n_rows; r0 += 4) {
                float4 rr[4]; float vv[4]; bool ok[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const uint32_t i = first + threadIdx.x * n_rows + r0 + k;
                    ok[k] = r0 + k < n_rows && i < last;
                    if (ok[k]) {
                        const float4 *rp = src + (size_t) i * quads;
                        rr[k] = (DRT_PART_NT & 4) && s == 0 ? nt_load4(rp) : rp[0];
                        vv[k] = rr[k].w;
                        if (s == 1 && ch > 0) { const float4 c4 = rp[1]; vv[k] = ch == 1 ? c4.x : (ch == 2 ? c4.y : c4.z); }
                    }
                }
#pragma unroll
                for (int k = 0; k < 4; ++k) if (ok[k]) add_record(rr[k], vv[k]);
            }
            continue;
        }
        for (uint32_t i0 = first + threadIdx.x; i0 < last; i0 += 4 * blockDim.x) {
            float4 rr[4]; float vv[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) if (i0 + k * blockDim.x < last) {
                const float4 *rp = src + (size_t) (i0 + k * blockDim.x) * quads;
                rr[k] = (DRT_PART_NT & 4) && s == 0 ? nt_load4(rp) : rp[0];
                vv[k] = rr[k].w;
                if (s == 1 && ch > 0) { const float4 c4 = rp[1]; vv[k] = ch == 1 ? c4.x : (ch == 2 ? c4.y : c4.z); }
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                if (i0 + k * blockDim.x >= last) break;
                add_record(rr[k], vv[k]);
            }
        }
    }
}

}  // namespace

hipError_t launch_deferred_split(const DeferredPlan &D, hipStream_t stream)
{
    hipLaunchKernelGGL(rec_split_kernel, dim3(1), dim3(64), 0, stream, D);
    return hipGetLastError();
}

hipError_t launch_deferred_early_histogram(const Params &P, const DeferredPlan &D, hipStream_t side)
{
    const size_t lds = (size_t) D.n_bins * sizeof(uint32_t);
    hipLaunchKernelGGL(bin_histogram_kernel, dim3(kPartWGs, kRecStreams), dim3(kPartThreads), lds, side, P, D, 1);
    return hipGetLastError();
}

// phase 0: everything; 1: the partition only (histogram, offsets, scan, scatter: the passes that do not touch the caller's gradient grids - the
// queued tracer runs them on a side stream beside its tail launch); 2: tile_reduce only (the partition has been made)
hipError_t launch_deferred_reduce(const Params &P, const DeferredPlan &D, hipStream_t stream, hipEvent_t *ev, bool early_hist, int phase)
{
    const size_t lds = (size_t) D.n_bins * sizeof(uint32_t);
    auto mark = [&](int k) { if (ev) (void) hipEventRecord(ev[k], stream); };
    mark(0);
    if (phase != 2) {
        hipLaunchKernelGGL(bin_histogram_kernel, dim3(kPartWGs, kRecStreams), dim3(kPartThreads), lds, stream, P, D, early_hist ? 2 : 0);
        mark(1);
        hipLaunchKernelGGL(bin_offsets_kernel, dim3((D.n_bins + 3) / 4, kRecStreams), dim3(256), 0, stream, D, kPartWGs);
        hipLaunchKernelGGL(bin_scan_kernel, dim3(kRecStreams), dim3(1024), 0, stream, D);
        mark(2);
        hipLaunchKernelGGL(bin_scatter_kernel, dim3(kPartWGs, kRecStreams), dim3(kPartThreads), lds, stream, P, D);
        mark(3);
    }
    if (phase == 1) return hipGetLastError();
    {   // (the attribute belongs to the function on a device: set once per device)
        static std::atomic<bool> attr_set[64];                   // (zero-initialised; handles may be driven from several host threads)
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 63;
        if (!attr_set[dev] || dev == 63) {
            const hipError_t attr = hipFuncSetAttribute((const void *) tile_reduce_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, kLdsTile * 8);
            if (attr != hipSuccess) return attr;
            attr_set[dev] = true;
        }
    }
    const uint32_t wgs = D.max_units < kReduceWGs ? D.max_units : kReduceWGs;
    hipLaunchKernelGGL(tile_reduce_kernel, dim3(wgs, 5), dim3(DRT_REDUCE_THREADS), (size_t) kLdsTile * 8, stream, P, D);
    mark(4);
    return hipGetLastError();
}

}  // namespace drt
