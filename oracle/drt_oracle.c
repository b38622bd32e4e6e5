/*
 * drt_oracle.c -- CPU ORACLE (test infrastructure, NOT the product).
 * See drt_oracle.h for scope, the list of reference files restated here and
 * the "parity unpinned" statement.  Build: oracle/Makefile (gcc, -ffp-contract=off).
 *
 * Every float operation is written out in a fixed order (explicit fmaf where a
 * fused op is meant) so that an independent implementation following the same
 * specification (DESIGN.md "Arithmetic specification") reproduces the primal
 * radiance bit for bit.
 */
#include "drt_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* ------------------------------------------------------------------------- */
/* small helpers                                                              */
/* ------------------------------------------------------------------------- */
typedef struct { float x, y, z; } v3;

static inline uint32_t f2u(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
static inline float u2f(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }
static inline v3 v3_make(float x, float y, float z) { v3 r = { x, y, z }; return r; }
/* Ray3f::operator()(t) = fmadd(d, t, o) [M3-ext] */
static inline v3 ray_at(v3 o, v3 d, float t)
{
    return v3_make(fmaf(d.x, t, o.x), fmaf(d.y, t, o.y), fmaf(d.z, t, o.z));
}
static inline float max3f(float a, float b, float c) { return fmaxf(a, fmaxf(b, c)); }

#define DRT_INV_FOURPI 0.07957747154594767f   /* 1/(4 pi) */
#define DRT_FOURPI     12.566370614359172f
#define DRT_HALF_PI    1.5707963267948966f
#define DRT_LARGEST    3.4028234663852886e38f /* dr.largest(Float) */
#define DRT_RAY_EPS    (1500.0f * 5.9604644775390625e-8f) /* math::RayEpsilon<float> [M3-ext] */

/* ------------------------------------------------------------------------- */
/* E7: sample_tea_32 + PCG32 `independent` sampler [M3-ext]                   */
/* call sites volpathsimple.py:71,99,105-107,120,222,348,359,383,418,470,536,595,632 */
/* ------------------------------------------------------------------------- */
static inline void tea32(uint32_t v0, uint32_t v1, uint32_t *o0, uint32_t *o1)
{
    uint32_t sum = 0;
    for (int i = 0; i < 4; ++i) {
        sum += 0x9e3779b9u;
        v0 += ((v1 << 4) + 0xa341316cu) ^ (v1 + sum) ^ ((v1 >> 5) + 0xc8013ea4u);
        v1 += ((v0 << 4) + 0xad90777du) ^ (v0 + sum) ^ ((v0 >> 5) + 0x7e95761eu);
    }
    *o0 = v0; *o1 = v1;
}

typedef struct { uint64_t state, inc; } pcg32;

static inline uint32_t pcg32_next_u32(pcg32 *r)
{
    uint64_t old = r->state;
    r->state = old * 0x5851f42d4c957f2dull + r->inc;
    uint32_t xs = (uint32_t)(((old >> 18) ^ old) >> 27);
    uint32_t rot = (uint32_t)(old >> 59);
    return (xs >> rot) | (xs << ((0u - rot) & 31u));
}
/* PCG32::seed(initstate, initseq) */
static inline void pcg32_seed(pcg32 *r, uint64_t initstate, uint64_t initseq)
{
    r->state = 0;
    r->inc = (initseq << 1) | 1ull;
    pcg32_next_u32(r);
    r->state += initstate;
    pcg32_next_u32(r);
}
/* Sampler::seed(seed, wavefront): lane `index` gets PCG32(tea32(seed, index)) */
static inline void sampler_seed(pcg32 *r, uint32_t seed, uint32_t index)
{
    uint32_t v0, v1;
    tea32(seed, index, &v0, &v1);
    pcg32_seed(r, (uint64_t) v0, (uint64_t) v1);
}
/* next_float32: 23 random mantissa bits in [0,1) */
static inline float next_1d(pcg32 *r)
{
    return u2f((pcg32_next_u32(r) >> 9) | 0x3f800000u) - 1.0f;
}

/* ------------------------------------------------------------------------- */
/* deterministic elementary functions (specified op by op)         */
/* ------------------------------------------------------------------------- */
/* natural log for normal positive x (Cephes logf polynomial, Horner in fmaf). */
static inline float drt_logf(float x)
{
    uint32_t ix = f2u(x);
    int e = (int)(ix >> 23) - 126;                 /* x = m * 2^e, m in [0.5,1) */
    float m = u2f((ix & 0x007fffffu) | 0x3f000000u);
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

/* exp for x <= 0 (Cephes expf: range reduction by ln2 split, degree-5 polynomial, exact 2^n scale). */
static inline float drt_expf(float x)
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
    return r * u2f((uint32_t)(n + 127) << 23);     /* n in [-126, 1] here */
}

/* sin/cos of 2*pi*u for u in [0,1): exact quadrant split, Cephes minimax on [0,pi/4]. */
static inline void drt_sincos_2pi(float u, float *s_out, float *c_out)
{
    float a = u * 4.0f;
    int q = (int) a;                 /* 0..3 */
    float f = a - (float) q;         /* exact, [0,1) */
    int swap = f > 0.5f;
    float g = swap ? (1.0f - f) : f; /* [0,0.5] */
    float x = g * DRT_HALF_PI;       /* [0,pi/4] */
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
    float sq, cq;
    switch (q & 3) {
        case 0: sq = s;  cq = c;  break;
        case 1: sq = c;  cq = -s; break;
        case 2: sq = -s; cq = -c; break;
        default: sq = -c; cq = s; break;
    }
    *s_out = sq; *c_out = cq;
}

/* warp::square_to_uniform_sphere [M3-ext]; used by the isotropic phase (E6,
 * volpathsimple.py:221,630) and the constant emitter (E5, :419). */
static inline v3 square_to_uniform_sphere(float ux, float uy)
{
    float z = fmaf(-2.0f, uy, 1.0f);
    float r = sqrtf(fmaxf(0.0f, fmaf(-z, z, 1.0f)));
    float s, c;
    drt_sincos_2pi(ux, &s, &c);
    return v3_make(r * c, r * s, z);
}

/* mi.ad.common.mis_weight (E8): power heuristic, non-finite -> 0 */
static inline float mis_weight(float a, float b)
{
    float a2 = a * a, b2 = b * b;
    float w = a2 / (a2 + b2);
    return isfinite(w) ? w : 0.0f;
}

/* atan2 with a specified instruction sequence (Cephes atanf polynomial on min/max in [0,1]); the
 * HIP kernels evaluate the same sequence, so envmap lookups are bit-identical.  atan2(0,0) = 0. */
static inline float drt_atan2f(float y, float x)
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

/* ------------------------------------------------------------------------- */
/* envmap emitter (E5) [M3-ext]: see drto_emitter in drt_oracle.h             */
/* ------------------------------------------------------------------------- */
typedef struct {
    const float *pix;            /* [h][w][3], NULL = constant emitter */
    int w, h;
    float R[9];                  /* to_world, row-major */
    float scale;
    float *marg;                 /* [h+1] marginal CDF over rows */
    float *cond;                 /* [h][w+1] conditional CDFs over columns */
} envmap_t;

#define DRT_INV_TWOPI 0.15915494309189535f
#define DRT_INV_PI    0.31830988618379069f
#define DRT_TWO_PI_SQ 19.739208802178716f    /* 2 pi^2 */
#define DRT_ONE_MINUS_EPS 0.99999994f

/* Importance-sampling tables.  Double accumulation, one rounding to float per entry; the HIP
 * library's host code (drt_capi.cpp) performs the same loop. */
static int envmap_build(envmap_t *e)
{
    const int w = e->w, h = e->h;
    e->marg = (float *) malloc(sizeof(float) * (size_t)(h + 1));
    e->cond = (float *) malloc(sizeof(float) * (size_t) h * (size_t)(w + 1));
    double *lum = (double *) malloc(sizeof(double) * (size_t) w * h);
    double *rowsum = (double *) malloc(sizeof(double) * (size_t) h);
    if (!e->marg || !e->cond || !lum || !rowsum) { free(lum); free(rowsum); return -4; }
    double lmax = 0.0;
    for (size_t i = 0; i < (size_t) w * h; ++i) {
        const float *p = e->pix + 3 * i;
        double l = 0.212671 * (double) p[0] + 0.715160 * (double) p[1] + 0.072169 * (double) p[2];
        lum[i] = l > 0.0 ? l : 0.0;
        if (lum[i] > lmax) lmax = lum[i];
    }
    const int uniform = !(lmax > 0.0);                   /* all-black map: uniform weights */
    double total = 0.0;
    for (int j = 0; j < h; ++j) {
        const double st = sin(3.14159265358979323846 * ((double) j + 0.5) / (double) h);
        float *c = e->cond + (size_t) j * (w + 1);
        double run = 0.0;
        for (int i = 0; i < w; ++i) {
            double m = uniform ? 1.0 : 0.0;
            if (!uniform)
                for (int dj = -1; dj <= 1; ++dj) for (int di = -1; di <= 1; ++di) {
                    int jj = j + dj, ii = (i + di + w) % w;
                    jj = jj < 0 ? 0 : (jj > h - 1 ? h - 1 : jj);
                    double l = lum[(size_t) jj * w + ii];
                    if (l > m) m = l;
                }
            run += m * st;
            c[i + 1] = (float) run;                      /* unnormalised for now */
        }
        rowsum[j] = run;
        total += run;
        c[0] = 0.0f;
        for (int i = 0; i < w; ++i)
            c[i + 1] = run > 0.0 ? (float)((double) c[i + 1] / run) : (float)((double)(i + 1) / (double) w);
        c[w] = 1.0f;
    }
    double run = 0.0;
    e->marg[0] = 0.0f;
    for (int j = 0; j < h; ++j) { run += rowsum[j]; e->marg[j + 1] = (float)(run / total); }
    e->marg[h] = 1.0f;
    free(lum); free(rowsum);
    return 0;
}

static void envmap_free(envmap_t *e) { free(e->marg); free(e->cond); e->marg = e->cond = NULL; }

/* bilinear lookup at uv in [0,1)^2: texel centres at ((i+.5)/w, (j+.5)/h), wrap in u, clamp in v */
static inline void envmap_lookup(const envmap_t *e, float u, float v, float out[3])
{
    float px = fmaf(u, (float) e->w, -0.5f), py = fmaf(v, (float) e->h, -0.5f);
    float fx0 = floorf(px), fy0 = floorf(py);
    float fx = px - fx0, fy = py - fy0;
    int i0 = (int) fx0, j0 = (int) fy0;
    int i1 = i0 + 1, j1 = j0 + 1;
    if (i0 < 0) i0 += e->w;
    if (i1 >= e->w) i1 -= e->w;
    j0 = j0 < 0 ? 0 : (j0 > e->h - 1 ? e->h - 1 : j0);
    j1 = j1 < 0 ? 0 : (j1 > e->h - 1 ? e->h - 1 : j1);
    const float *p00 = e->pix + 3 * ((size_t) j0 * e->w + i0), *p01 = e->pix + 3 * ((size_t) j0 * e->w + i1);
    const float *p10 = e->pix + 3 * ((size_t) j1 * e->w + i0), *p11 = e->pix + 3 * ((size_t) j1 * e->w + i1);
    float wx0 = 1.0f - fx, wy0 = 1.0f - fy;
    for (int k = 0; k < 3; ++k) {
        float a = fmaf(wx0, p00[k], fx * p01[k]);
        float b = fmaf(wx0, p10[k], fx * p11[k]);
        out[k] = fmaf(wy0, a, fy * b) * e->scale;
    }
}

/* world direction -> uv, sin(theta) */
static inline void envmap_dir_to_uv(const envmap_t *e, v3 d, float *u, float *v, float *sin_theta)
{
    const float *R = e->R;                               /* local = R^T d */
    float lx = fmaf(R[6], d.z, fmaf(R[3], d.y, R[0] * d.x));
    float ly = fmaf(R[7], d.z, fmaf(R[4], d.y, R[1] * d.x));
    float lz = fmaf(R[8], d.z, fmaf(R[5], d.y, R[2] * d.x));
    float st = sqrtf(fmaf(lx, lx, lz * lz));
    float uu = drt_atan2f(lx, -lz) * DRT_INV_TWOPI;
    if (uu < 0.0f) uu += 1.0f;
    if (uu >= 1.0f) uu = 0.0f;
    float vv = drt_atan2f(st, ly) * DRT_INV_PI;
    vv = fminf(fmaxf(vv, 0.0f), DRT_ONE_MINUS_EPS);
    *u = uu; *v = vv; *sin_theta = st;
}

/* texel probability density in uv space: pmf(row) * pmf(col | row) * w * h */
static inline float envmap_pdf_uv(const envmap_t *e, int i, int j)
{
    const float *c = e->cond + (size_t) j * (e->w + 1);
    float pm = e->marg[j + 1] - e->marg[j], pc = c[i + 1] - c[i];
    return (pm * pc) * ((float) e->w * (float) e->h);
}

/* Emitter::eval for an escaped ray of direction d (volpathsimple.py:284) */
static inline void envmap_eval(const envmap_t *e, v3 d, float out[3])
{
    float u, v, st;
    envmap_dir_to_uv(e, d, &u, &v, &st);
    envmap_lookup(e, u, v, out);
}

/* Emitter::pdf_direction (volpathsimple.py:273) */
static inline float envmap_pdf(const envmap_t *e, v3 d)
{
    float u, v, st;
    envmap_dir_to_uv(e, d, &u, &v, &st);
    int i = (int)(u * (float) e->w), j = (int)(v * (float) e->h);
    if (i > e->w - 1) i = e->w - 1;
    if (j > e->h - 1) j = e->h - 1;
    float den = DRT_TWO_PI_SQ * st;
    return den > 0.0f ? envmap_pdf_uv(e, i, j) / den : 0.0f;
}

/* largest k in [0, n-1] with cdf[k] <= x (cdf[0] = 0, cdf[n] = 1, x in [0,1)) */
static inline int cdf_find(const float *cdf, int n, float x)
{
    int lo = 0, hi = n;                                  /* invariant: cdf[lo] <= x < cdf[hi] */
    while (hi - lo > 1) {
        int mid = (lo + hi) >> 1;
        if (cdf[mid] <= x) lo = mid; else hi = mid;
    }
    return lo;
}

/* Scene::sample_emitter_direction (volpathsimple.py:419): direction, pdf, radiance / pdf.  The pdf
 * and the radiance are evaluated FROM THE DIRECTION (as pdf_direction / eval do for an escaped
 * ray), so NEE and the escape-side MIS weight see the same density. */
static inline v3 envmap_sample_dir(const envmap_t *e, float u1, float u2)
{
    int j = cdf_find(e->marg, e->h, u2);
    const float *c = e->cond + (size_t) j * (e->w + 1);
    int i = cdf_find(c, e->w, u1);
    float dv = fminf((u2 - e->marg[j]) / (e->marg[j + 1] - e->marg[j]), DRT_ONE_MINUS_EPS);
    float du = fminf((u1 - c[i]) / (c[i + 1] - c[i]), DRT_ONE_MINUS_EPS);
    float u = ((float) i + du) / (float) e->w, v = ((float) j + dv) / (float) e->h;
    float sp, cp, st, ct;
    drt_sincos_2pi(u, &sp, &cp);
    drt_sincos_2pi(0.5f * v, &st, &ct);
    float lx = sp * st, ly = ct, lz = -(cp * st);
    const float *R = e->R;                               /* world = R local */
    v3 d;
    d.x = fmaf(R[2], lz, fmaf(R[1], ly, R[0] * lx));
    d.y = fmaf(R[5], lz, fmaf(R[4], ly, R[3] * lx));
    d.z = fmaf(R[8], lz, fmaf(R[7], ly, R[6] * lx));
    return d;
}

static inline void envmap_sample(const envmap_t *e, float u1, float u2, v3 *d, float *pdf, float weight[3])
{
    *d = envmap_sample_dir(e, u1, u2);
    float p = envmap_pdf(e, *d), Le[3];
    envmap_eval(e, *d, Le);
    *pdf = p;
    for (int k = 0; k < 3; ++k) weight[k] = p > 0.0f ? Le[k] / p : 0.0f;
}

static int envmap_from_emitter(envmap_t *e, const drto_emitter *em)
{
    if (em->width < 2 || em->height < 2) return -2;
    e->pix = em->pixels; e->w = em->width; e->h = em->height; e->scale = em->scale;
    for (int k = 0; k < 9; ++k) e->R[k] = em->to_world[k];
    return envmap_build(e);
}

/* ------------------------------------------------------------------------- */
/* scene                                                                      */
/* ------------------------------------------------------------------------- */
typedef struct {
    drto_config cfg;
    const float *sigma_t, *albedo;
    int rx, ry, rz;
    int crx, cry, crz;           /* the colour grids' own lattice (albedo, nerf emission); = rx, ry, rz unless drto_medium::res_colour says otherwise */
    int own_colour;              /* ... it differs from sigma_t's */
    v3 bmin, bmax, inv_ext;
    float scale, majorant, inv_majorant;
    /* majorant supergrid (majorant_resolution_factor > 0): gx*gy*gz cells, [2*cell] = majorant,
     * [2*cell+1] = 1/majorant (0 if empty) */
    int gx, gy, gz;
    float *mgrid;
    float Le[3];
    envmap_t env;                /* env.pix != NULL: envmap emitter instead of the constant Le */
} scene_t;

/* Thread-local write-combining cache in front of the shared gradient grids (job->grad_cache_log2 > 0; used by
 * the timed CPU-baseline leg only): direct-mapped on the voxel index, a hit accumulates privately, an eviction
 * costs ONE atomic add.  Rays of neighbouring pixels splat into the same voxels, so most adds never reach the
 * shared grid - without it 256 threads mostly wait for each other's cache lines (`omp atomic` on doubles). */
typedef struct { uint32_t tag; double v[4]; } gcache_line;   /* [0] sigma_t, [1..3] albedo rgb of voxel `tag` */

/* Tile-binned gradient accumulation for the timed CPU-baseline leg (job->grad_cache_log2 == -1), the CPU counterpart of the device's deferred
 * splatting: a thread appends its splats as records {p, value(s)} to ITS bucket of the base corner's z layer (sigma_t records by sigma_t's
 * lattice, colour records by the colour lattice); behind the ray loop the layers are reduced in two sweeps - even layers, then odd ones: a
 * record of layer z adds to layers z and z + 1, so layers of one parity never touch the same voxels - each layer by one thread with plain adds.
 * No atomics, no shared cache lines in the hot loop; the fp64 sums differ from the other modes by their order only. */
typedef struct { float p[3]; float v[3]; } grec_t;           /* sigma_t record: v[0]; colour record: v[0..2] */
typedef struct gbucket { grec_t *r; uint32_t n, cap; } gbucket_t;
static inline void gbucket_push(gbucket_t *b, v3 p, float v0, float v1, float v2)
{
    if (b->n == b->cap) {
        uint32_t cap = b->cap ? b->cap * 2u : 256u;
        grec_t *r = (grec_t *) realloc(b->r, (size_t) cap * sizeof(grec_t));
        if (!r) return;                                      /* (out of memory: the record is lost - a timing leg, checked nowhere) */
        b->r = r; b->cap = cap;
    }
    grec_t *q = &b->r[b->n++];
    q->p[0] = p.x; q->p[1] = p.y; q->p[2] = p.z; q->v[0] = v0; q->v[1] = v1; q->v[2] = v2;
}

typedef struct {
    const scene_t *sc;
    double *g_sigma, *g_albedo;   /* NULL in primal */
    drto_counters cnt;
    uint32_t alt_seed;
    uint32_t ray_index;
    gcache_line *gcache;          /* NULL: every splat goes to the shared grids with atomics */
    uint32_t gcache_mask;
    struct gbucket *gbin;         /* job->grad_cache_log2 == -1 (timed CPU-baseline leg): this thread's splat records by z layer, see run_job */
} ctx_t;

/* E3: GridVolume::eval, trilinear, clamp, cell-centred (q = p*res - 0.5) */
typedef struct { int idx[8]; float wx0, wx1, wy0, wy1, wz0, wz1; } stencil_t;

static inline void axis_setup(float p, float bmin, float inv_ext, int res,
                              int *i0, int *i1, float *w0, float *w1)
{
    float l = (p - bmin) * inv_ext;
    float q = fmaf(l, (float) res, -0.5f);
    float fl = floorf(q);
    float fw = q - fl;
    fl = fminf(fmaxf(fl, -1.0f), (float) res);
    int i = (int) fl;
    int a = i < 0 ? 0 : (i > res - 1 ? res - 1 : i);
    int b = i + 1 < 0 ? 0 : (i + 1 > res - 1 ? res - 1 : i + 1);
    *i0 = a; *i1 = b; *w1 = fw; *w0 = 1.0f - fw;
}

static inline void make_stencil(const scene_t *sc, v3 p, stencil_t *s)
{
    int x0, x1, y0, y1, z0, z1;
    axis_setup(p.x, sc->bmin.x, sc->inv_ext.x, sc->rx, &x0, &x1, &s->wx0, &s->wx1);
    axis_setup(p.y, sc->bmin.y, sc->inv_ext.y, sc->ry, &y0, &y1, &s->wy0, &s->wy1);
    axis_setup(p.z, sc->bmin.z, sc->inv_ext.z, sc->rz, &z0, &z1, &s->wz0, &s->wz1);
    int sy = sc->rx, sz = sc->rx * sc->ry;
    s->idx[0] = z0 * sz + y0 * sy + x0; s->idx[1] = z0 * sz + y0 * sy + x1;
    s->idx[2] = z0 * sz + y1 * sy + x0; s->idx[3] = z0 * sz + y1 * sy + x1;
    s->idx[4] = z1 * sz + y0 * sy + x0; s->idx[5] = z1 * sz + y0 * sy + x1;
    s->idx[6] = z1 * sz + y1 * sy + x0; s->idx[7] = z1 * sz + y1 * sy + x1;
}

static inline float trilerp(const stencil_t *s, const float *data, int stride, int ch)
{
    float d[8];
    for (int k = 0; k < 8; ++k) d[k] = data[(size_t) s->idx[k] * stride + ch];
    float v00 = fmaf(s->wx0, d[0], s->wx1 * d[1]);
    float v01 = fmaf(s->wx0, d[2], s->wx1 * d[3]);
    float v10 = fmaf(s->wx0, d[4], s->wx1 * d[5]);
    float v11 = fmaf(s->wx0, d[6], s->wx1 * d[7]);
    float v0 = fmaf(s->wy0, v00, s->wy1 * v01);
    float v1 = fmaf(s->wy0, v10, s->wy1 * v11);
    return fmaf(s->wz0, v0, s->wz1 * v1);
}

/* get_scattering_coefficients: sigma_t = scale * trilerp(sigma_t.data) */
static inline float eval_sigma_t(const scene_t *sc, v3 p)
{
    stencil_t s;
    make_stencil(sc, p, &s);
    return trilerp(&s, sc->sigma_t, 1, 0) * sc->scale;
}
/* the same on the colour grids' lattice (GridVolume::eval interpolates every grid on its own resolution): identical to
 * make_stencil when the lattices are equal */
static inline void make_stencil_colour(const scene_t *sc, v3 p, stencil_t *s)
{
    int x0, x1, y0, y1, z0, z1;
    axis_setup(p.x, sc->bmin.x, sc->inv_ext.x, sc->crx, &x0, &x1, &s->wx0, &s->wx1);
    axis_setup(p.y, sc->bmin.y, sc->inv_ext.y, sc->cry, &y0, &y1, &s->wy0, &s->wy1);
    axis_setup(p.z, sc->bmin.z, sc->inv_ext.z, sc->crz, &z0, &z1, &s->wz0, &s->wz1);
    int sy = sc->crx, sz = sc->crx * sc->cry;
    s->idx[0] = z0 * sz + y0 * sy + x0; s->idx[1] = z0 * sz + y0 * sy + x1;
    s->idx[2] = z0 * sz + y1 * sy + x0; s->idx[3] = z0 * sz + y1 * sy + x1;
    s->idx[4] = z1 * sz + y0 * sy + x0; s->idx[5] = z1 * sz + y0 * sy + x1;
    s->idx[6] = z1 * sz + y1 * sy + x0; s->idx[7] = z1 * sz + y1 * sy + x1;
}
/* get_albedo */
static inline void eval_albedo(const scene_t *sc, v3 p, float out[3])
{
    stencil_t s;
    make_stencil_colour(sc, p, &s);
    for (int c = 0; c < 3; ++c) out[c] = trilerp(&s, sc->albedo, 3, c);
}

static inline void stencil_weights(const stencil_t *s, float w[8])
{
    float zy00 = s->wz0 * s->wy0, zy01 = s->wz0 * s->wy1;
    float zy10 = s->wz1 * s->wy0, zy11 = s->wz1 * s->wy1;
    w[0] = zy00 * s->wx0; w[1] = zy00 * s->wx1; w[2] = zy01 * s->wx0; w[3] = zy01 * s->wx1;
    w[4] = zy10 * s->wx0; w[5] = zy10 * s->wx1; w[6] = zy11 * s->wx0; w[7] = zy11 * s->wx1;
}

static inline void shared_add(double *dst, double v)
{
#ifdef _OPENMP
#pragma omp atomic
#endif
    *dst += v;
}
/* (colour grids on their own lattice: their voxels are cached under tag | 0x80000000 and only use v[1..3]) */
static inline void gcache_evict(ctx_t *c, gcache_line *l)
{
    if (l->tag == 0xffffffffu) return;
    if (l->v[0] != 0.0) shared_add(&c->g_sigma[l->tag], l->v[0]);
    const size_t cv = l->tag & 0x7fffffffu;
    for (int ch = 0; ch < 3; ++ch)
        if (l->v[1 + ch] != 0.0) shared_add(&c->g_albedo[cv * 3 + ch], l->v[1 + ch]);
    l->v[0] = l->v[1] = l->v[2] = l->v[3] = 0.0;
}
static inline double *gcache_slot(ctx_t *c, uint32_t voxel)
{
    gcache_line *l = &c->gcache[(voxel * 2654435761u >> 7) & c->gcache_mask];
    if (l->tag != voxel) { gcache_evict(c, l); l->tag = voxel; }
    return l->v;
}
static void gcache_flush(ctx_t *c)
{
    if (!c->gcache) return;
    for (uint32_t i = 0; i <= c->gcache_mask; ++i) { gcache_evict(c, &c->gcache[i]); c->gcache[i].tag = 0xffffffffu; }
}

/* reverse-mode of the trilinear gather = 8-corner scatter_reduce(Add) (E3/E9) */
static inline void splat_sigma_t(ctx_t *c, v3 p, float g)
{
    stencil_t s; float w[8];
    if (c->gbin) {                                           /* binned mode: the record, by its base corner's z layer */
        int z0, z1; float w0, w1;
        axis_setup(p.z, c->sc->bmin.z, c->sc->inv_ext.z, c->sc->rz, &z0, &z1, &w0, &w1);
        gbucket_push(&c->gbin[z0], p, g, 0.0f, 0.0f);
        return;
    }
    make_stencil(c->sc, p, &s);
    stencil_weights(&s, w);
    float gs = g * c->sc->scale;
    for (int k = 0; k < 8; ++k) {
        double v = (double)(w[k] * gs);
        if (c->gcache) gcache_slot(c, (uint32_t) s.idx[k])[0] += v;
        else shared_add(&c->g_sigma[s.idx[k]], v);
    }
}
static inline void splat_albedo(ctx_t *c, v3 p, const float g[3])
{
    stencil_t s; float w[8];
    if (c->gbin) {
        int z0, z1; float w0, w1;
        axis_setup(p.z, c->sc->bmin.z, c->sc->inv_ext.z, c->sc->crz, &z0, &z1, &w0, &w1);
        gbucket_push(&c->gbin[c->sc->rz + z0], p, g[0], g[1], g[2]);   /* (colour layers behind sigma_t's) */
        return;
    }
    make_stencil_colour(c->sc, p, &s);
    stencil_weights(&s, w);
    const uint32_t tag_bit = c->sc->own_colour ? 0x80000000u : 0u;
    for (int k = 0; k < 8; ++k) {
        double *slot = c->gcache ? gcache_slot(c, (uint32_t) s.idx[k] | tag_bit) : NULL;
        for (int ch = 0; ch < 3; ++ch) {
            double v = (double)(w[k] * g[ch]);
            if (slot) slot[1 + ch] += v;
            else shared_add(&c->g_albedo[(size_t) s.idx[k] * 3 + ch], v);
        }
    }
}

/* E4: scene.ray_intersect with use_bbox_fast_path: nearest hit with the box
 * surface at t > 0 (entry face from outside, exit face from inside). */
typedef struct { int valid; float t; v3 p; v3 n; } si_t;

static inline si_t box_hit(const scene_t *sc, v3 o, v3 d)
{
    si_t si; si.valid = 0; si.t = INFINITY; si.p = v3_make(0, 0, 0); si.n = v3_make(0, 0, 0);
    float tn = -INFINITY, tf = INFINITY;
    int an = 0, af = 0;
    const float oo[3] = { o.x, o.y, o.z }, dd[3] = { d.x, d.y, d.z };
    const float lo[3] = { sc->bmin.x, sc->bmin.y, sc->bmin.z };
    const float hi[3] = { sc->bmax.x, sc->bmax.y, sc->bmax.z };
    for (int a = 0; a < 3; ++a) {
        if (dd[a] != 0.0f) {
            /* slab distances through the reciprocal direction, as Mitsuba's BoundingBox::ray_intersect forms them
             * (d_rcp = rcp(ray.d); t = (plane - o) * d_rcp): one IEEE division per axis instead of two */
            float rcp = 1.0f / dd[a];
            float t0 = (lo[a] - oo[a]) * rcp;
            float t1 = (hi[a] - oo[a]) * rcp;
            if (t0 > t1) { float t = t0; t0 = t1; t1 = t; }
            if (t0 > tn) { tn = t0; an = a; }
            if (t1 < tf) { tf = t1; af = a; }
        } else if (oo[a] < lo[a] || oo[a] > hi[a]) {
            return si;
        }
    }
    if (!(tn <= tf)) return si;
    float t; int ax; float sgn;
    if (tn > 0.0f) { t = tn; ax = an; sgn = dd[ax] > 0.0f ? -1.0f : 1.0f; }
    else if (tf > 0.0f) { t = tf; ax = af; sgn = dd[ax] > 0.0f ? 1.0f : -1.0f; }
    else return si;
    if (!isfinite(t)) return si;
    si.valid = 1; si.t = t; si.p = ray_at(o, d, t);
    float n[3] = { 0, 0, 0 }; n[ax] = sgn;
    si.n = v3_make(n[0], n[1], n[2]);
    return si;
}

/* SurfaceInteraction::spawn_ray -> offset_p(d) [M3-ext] */
static inline v3 offset_p(const si_t *si, v3 d)
{
    float mag = (1.0f + max3f(fabsf(si->p.x), fabsf(si->p.y), fabsf(si->p.z))) * DRT_RAY_EPS;
    float dn = si->n.x * d.x + si->n.y * d.y + si->n.z * d.z;
    if (dn < 0.0f) mag = -mag;
    return v3_make(fmaf(mag, si->n.x, si->p.x), fmaf(mag, si->n.y, si->p.y), fmaf(mag, si->n.z, si->p.z));
}

/* E1: free-flight distance of Medium::sample_interaction with a global
 * majorant: t = -log(1-u)/majorant (multiplication by the reciprocal). */
static inline float sample_distance(const scene_t *sc, float u)
{
    if (sc->majorant == 0.0f) return INFINITY;
    return -drt_logf(1.0f - u) * sc->inv_majorant;
}

/* E1 with a majorant supergrid [M3-ext] (scene_config.py:36, optimize.py:182-199): the free-flight
 * distance is sampled against piecewise-constant local majorants by a 3-D DDA through the
 * supergrid: walk the cells the ray crosses, accumulating majorant * length until the target
 * optical depth tau = -log(1-u) is reached.  Returns the distance (INFINITY if the ray leaves
 * [0,tmax] first) and the local majorant / its reciprocal at the collision. */
#ifndef DRTO_DDA_VISIT            /* experiment hooks (tools/dda_stats.c includes this file with them defined) */
#define DRTO_DDA_VISIT(sc, cx, cy, cz, m) ((void) 0)
#define DRTO_DDA_END(sc) ((void) 0)
#endif
static float sample_collision(const scene_t *sc, v3 o, v3 d, float tmax, float u, float *m_out, float *im_out)
{
    if (!sc->mgrid) {
        *m_out = sc->majorant; *im_out = sc->inv_majorant;
        return sample_distance(sc, u);
    }
    const float tau = -drt_logf(1.0f - u);
    const int G[3] = { sc->gx, sc->gy, sc->gz };
    const float oo[3] = { o.x, o.y, o.z }, dd[3] = { d.x, d.y, d.z };
    const float lo[3] = { sc->bmin.x, sc->bmin.y, sc->bmin.z };
    const float ie[3] = { sc->inv_ext.x, sc->inv_ext.y, sc->inv_ext.z };
    int cell[3], step[3]; float tnext[3], tdelta[3];
    for (int a = 0; a < 3; ++a) {
        float g = ((oo[a] - lo[a]) * ie[a]) * (float) G[a];
        float dg = (dd[a] * ie[a]) * (float) G[a];
        float fl = fminf(fmaxf(floorf(g), 0.0f), (float)(G[a] - 1));
        cell[a] = (int) fl;
        /* crossing times: tdelta = 1/|dg| (one division per axis: d is the same for every flight of a walk), first
         * crossing = distance to the next plane in cells * tdelta.  |dg| < 1e-20: the ray cannot cross a plane of this
         * axis inside the box - parallel (keeps every crossing time finite or +inf, never NaN) */
        if (dg >= 1e-20f) { tdelta[a] = 1.0f / dg; tnext[a] = ((fl + 1.0f) - g) * tdelta[a]; step[a] = 1; }
        else if (dg <= -1e-20f) { tdelta[a] = 1.0f / -dg; tnext[a] = (g - fl) * tdelta[a]; step[a] = -1; }
        else { tnext[a] = INFINITY; tdelta[a] = INFINITY; step[a] = 0; }
    }
    float t = 0.0f, acc = 0.0f;
    for (;;) {
        int a = 0;
        if (tnext[1] < tnext[a]) a = 1;
        if (tnext[2] < tnext[a]) a = 2;
        float texit = fminf(tnext[a], tmax);
        const float *mc = sc->mgrid + 2 * (size_t)((cell[2] * G[1] + cell[1]) * G[0] + cell[0]);
        float m = mc[0];
        DRTO_DDA_VISIT(sc, cell[0], cell[1], cell[2], m);
        if (m > 0.0f) {
            float dtau = m * (texit - t);
            if (acc + dtau >= tau) {
                DRTO_DDA_END(sc);
                *m_out = m; *im_out = mc[1];
                return fmaf(tau - acc, mc[1], t);
            }
            acc += dtau;
        }
        t = texit;
        if (!(texit < tmax)) break;
        cell[a] += step[a];
        if (cell[a] < 0 || cell[a] >= G[a]) break;
        tnext[a] += tdelta[a];
    }
    DRTO_DDA_END(sc);
    *m_out = 0.0f; *im_out = 0.0f;
    return INFINITY;
}

typedef struct { v3 o, d; float maxt; } ray_t;

/* ------------------------------------------------------------------------- */
/* A8: estimate_transmittance -- ratio tracking  (volpathsimple.py:436-504)   */
/* adj == NULL: primal.  adj != NULL: also back-propagate (lines 483-492).    */
/* ------------------------------------------------------------------------- */
static float estimate_transmittance(ctx_t *c, v3 o, v3 d, float tmax, pcg32 *S, const float *adj)
{
    const scene_t *sc = c->sc;
    float T = 1.0f;
    for (;;) {
        float maj, imaj;
        float dt = sample_collision(sc, o, d, tmax, next_1d(S), &maj, &imaj);
        if (!(dt <= tmax)) break;                       /* :480-481 */
        v3 p = ray_at(o, d, dt);
        float sig = eval_sigma_t(sc, p);
        float tr = (maj - sig) * imaj;                  /* sigma_n / majorant :473-476 */
        c->cnt.n_rt++;
        if (adj && tr > 0.0f) {                          /* :487-492 */
            float a = (adj[0] + adj[1]) + adj[2];
            c->cnt.n_rt_adj++;
            splat_sigma_t(c, p, -(a * imaj) / tr);
        }
        T *= tr;                                         /* :495 */
        o = p; tmax -= dt;                               /* :497-499 */
        if (T == 0.0f) break;                            /* :502 */
    }
    return T;
}

/* A7: sample_emitter (volpathsimple.py:406-433).  Returns emitter_val * transmittance (RGB) in
 * out[] and ds.pdf in *ds_pdf. */
static void sample_emitter(ctx_t *c, v3 p, pcg32 *S, const float *adj, float out[3], float *ds_pdf)
{
    const scene_t *sc = c->sc;
    float ux = next_1d(S), uy = next_1d(S);              /* :418 */
    v3 wd;
    float pdf, val[3];
    if (sc->env.pix) {
        envmap_sample(&sc->env, ux, uy, &wd, &pdf, val); /* envmap::sample_direction */
    } else {
        wd = square_to_uniform_sphere(ux, uy);           /* constant::sample_direction */
        pdf = DRT_INV_FOURPI;
        for (int k = 0; k < 3; ++k) val[k] = sc->Le[k] * DRT_FOURPI;    /* radiance / pdf */
    }
    *ds_pdf = pdf;
    float T = 0.0f;
    if (pdf != 0.0f) {                                   /* sampling_worked :421-423 */
        si_t si = box_hit(sc, p, wd);                    /* mei.spawn_ray: no offset (n=0) :427-428 */
        if (si.valid) T = estimate_transmittance(c, p, wd, si.t, S, adj);
    }
    for (int k = 0; k < 3; ++k) out[k] = val[k] * T;
}

/* A7: sample_emitter_for_nee (volpathsimple.py:380-403) */
static void sample_emitter_for_nee(ctx_t *c, v3 p, pcg32 *S, const float beta[3], const float *dL,
                                   float contrib[3])
{
    pcg32 clone = *S;                                    /* :383 */
    float emitted[3], ds_pdf;
    sample_emitter(c, p, S, NULL, emitted, &ds_pdf);     /* :385 */
    float w = mis_weight(ds_pdf, DRT_INV_FOURPI);        /* ds.pdf vs phase_pdf :391 */
    for (int k = 0; k < 3; ++k)
        contrib[k] = ((beta[k] * DRT_INV_FOURPI) * w) * emitted[k];
    if (dL) {                                            /* :393-401 */
        float adj[3] = { dL[0] * contrib[0], dL[1] * contrib[1], dL[2] * contrib[2] };
        float unused[3], unused_pdf;
        sample_emitter(c, p, &clone, adj, unused, &unused_pdf);
    }
}

/* ------------------------------------------------------------------------- */
/* A4: sample_real_interaction -- delta tracking  (volpathsimple.py:323-377)  */
/* ------------------------------------------------------------------------- */
typedef struct { int valid; float t; v3 p; float sigma_t; } mei_t;

static mei_t sample_real_interaction(ctx_t *c, const ray_t *ray, pcg32 *S, int attached)
{
    const scene_t *sc = c->sc;
    mei_t mei; mei.valid = 0; mei.t = INFINITY; mei.p = v3_make(0, 0, 0); mei.sigma_t = 0.0f;
    v3 ro = ray->o; float rmaxt = ray->maxt; float running_t = 0.0f;
    for (;;) {
        float maj, imaj;
        float dt = sample_collision(sc, ro, ray->d, rmaxt, next_1d(S), &maj, &imaj);   /* :348 */
        if (!(dt <= rmaxt)) break;                       /* :358 escaped */
        v3 p = ray_at(ro, ray->d, dt);
        float sig = eval_sigma_t(sc, p);
        c->cnt.n_dt++;
        float r = sig * imaj;                            /* :354 */
        float u = next_1d(S);                            /* :359 */
        if (!(u >= r)) {                                 /* real collision */
            mei.valid = 1; mei.t = running_t + dt;       /* :351 */
            break;
        }
        ro = p; rmaxt -= dt; running_t += dt;            /* :364-367 */
    }
    if (mei.valid) {
        mei.p = ray_at(ray->o, ray->d, mei.t);           /* :371 */
        if (attached) {                                  /* :373-375 */
            mei.sigma_t = eval_sigma_t(sc, mei.p);
            c->cnt.n_dt++;
        }
    }
    return mei;
}

/* ------------------------------------------------------------------------- */
/* E2: Medium::sample_interaction_drt [M3-ext]: ratio tracking along          */
/* [0, maxt]; every tentative collision x_i carries weight T_i / majorant;    */
/* one kept by weighted reservoir sampling; returns W = sum_i T_i / majorant  */
/* so that E[W f(x')] = int_0^maxt T(t) f(t) dt  (call site :549-551).         */
/* ------------------------------------------------------------------------- */
static int sample_interaction_drt(ctx_t *c, const ray_t *ray, pcg32 *A, float *t_out, float *W_out)
{
    const scene_t *sc = c->sc;
    float t = 0.0f, T = 1.0f, wsum = 0.0f, tsel = INFINITY;
    int valid = 0;
    for (;;) {
        float maj, imaj;
        if (sc->mgrid) t += sample_collision(sc, ray_at(ray->o, ray->d, t), ray->d, ray->maxt - t, next_1d(A), &maj, &imaj);
        else { t += sample_distance(sc, next_1d(A)); maj = sc->majorant; imaj = sc->inv_majorant; }
        if (!(t <= ray->maxt)) break;
        float sig = eval_sigma_t(sc, ray_at(ray->o, ray->d, t));
        c->cnt.n_drt++;
        float w = T * imaj;
        wsum += w;
        float u = next_1d(A);
        if (w > 0.0f && u * wsum <= w) { tsel = t; valid = 1; }
        T *= (maj - sig) * imaj;
        if (T == 0.0f) break;
    }
    *t_out = tsel; *W_out = wsum;
    return valid;
}

/* ------------------------------------------------------------------------- */
/* A11/A12: PathState / DRTPathState / DRTReservoir (volpathsimple.py:660-765) */
/* ------------------------------------------------------------------------- */
typedef struct { int depth; si_t si; float last_pdf; int escaped; int active; } pstate_t;
typedef struct {
    int depth; si_t si; ray_t ray; int active;
    float wsum[3], cw[3];
} reservoir_t;

static void drt_sample(ctx_t *c, pcg32 *S, int adjoint, ray_t ray, const float *dL,
                       const float *state_in, const pstate_t *ps, float out[3]);

/* A10: sample_recursive (volpathsimple.py:610-655) */
static void sample_recursive(ctx_t *c, pcg32 *A, v3 p, int depth, float Li[3])
{
    const scene_t *sc = c->sc;
    Li[0] = Li[1] = Li[2] = 0.0f;
    if (sc->cfg.use_nee) {                               /* :621-624 */
        const float one[3] = { 1.0f, 1.0f, 1.0f };
        float nee[3];
        sample_emitter_for_nee(c, p, A, one, NULL, nee);
        for (int k = 0; k < 3; ++k) Li[k] += nee[k];
    }
    (void) next_1d(A);                                   /* phase.sample sample1 :632 */
    float ux = next_1d(A), uy = next_1d(A);
    v3 wo = square_to_uniform_sphere(ux, uy);
    ray_t rr; rr.o = p; rr.d = wo;
    si_t sn = box_hit(sc, p, wo);                        /* :637 */
    rr.maxt = sn.valid ? sn.t : DRT_LARGEST;             /* :639-640 */
    pstate_t ps;
    ps.depth = depth + 1; ps.si = sn; ps.last_pdf = DRT_INV_FOURPI; ps.escaped = 0;
    /* DEVIATION: an invalid si_next (fp corner case) deactivates the
     * recursive path instead of tracking forever with maxt = largest. */
    ps.active = (ps.depth < sc->cfg.max_depth) && sn.valid;  /* :647 */
    float Lr[3];
    drt_sample(c, A, 0, rr, NULL, NULL, &ps, Lr);        /* :651 */
    for (int k = 0; k < 3; ++k) Li[k] += Lr[k];
}

/* A9: backpropagate_scattering_drt, final/quadratic branch (volpathsimple.py:543-581) */
static void drt_backprop(ctx_t *c, pcg32 *A, const ray_t *ray, const si_t *si, int depth,
                         const float adj[3])
{
    const scene_t *sc = c->sc;
    ray_t sub = *ray;
    sub.maxt = isfinite(si->t) ? si->t : DRT_LARGEST;    /* :544-545 */
    float tp, W;
    if (!sample_interaction_drt(c, &sub, A, &tp, &W)) return;   /* :550,558 */
    v3 p = ray_at(sub.o, sub.d, tp);
    float sig = eval_sigma_t(sc, p);                     /* :553-554 attached */
    c->cnt.n_drt++;
    float Li[3];
    sample_recursive(c, A, p, depth, Li);                /* :565-568 */
    float w = sc->cfg.use_drt_mis ? 1.0f / (1.0f + sig * sig) : 1.0f;   /* :571-575 */
    float alb[3];
    eval_albedo(sc, p, alb);                             /* :578 */
    c->cnt.n_alb++;
    float ww = w * W;
    float gs = 0.0f, ga[3];
    for (int k = 0; k < 3; ++k) {
        float a = (ww * adj[k]) * Li[k];
        gs += a * alb[k];
        ga[k] = a * sig;
    }
    splat_sigma_t(c, p, gs); c->cnt.n_sc++;              /* :577-581 */
    splat_albedo(c, p, ga);  c->cnt.n_sc_alb++;
}

/* A6: backpropagate_transmittance (volpathsimple.py:584-607) */
static void backprop_transmittance(ctx_t *c, pcg32 *A, const ray_t *ray, float interval,
                                   const float dL[3], const float result[3])
{
    float adjw = (dL[0] * result[0] + dL[1] * result[1]) + dL[2] * result[2];
    float g = -(adjw * (interval / 4.0f));               /* contribs -= sigma_t; inv_pdf = interval/n */
    for (int j = 0; j < 4; ++j) {
        float t = next_1d(A) * interval;                 /* :595 */
        splat_sigma_t(c, ray_at(ray->o, ray->d, t), g);
        c->cnt.n_tr++;
    }
}

/* ------------------------------------------------------------------------- */
/* A2: VolpathSimpleIntegrator.sample (volpathsimple.py:38-290)               */
/* ------------------------------------------------------------------------- */
static void drt_sample(ctx_t *c, pcg32 *S, int adjoint, ray_t ray, const float *dL,
                       const float *state_in, const pstate_t *ps, float out[3])
{
    const scene_t *sc = c->sc;
    const drto_config *cfg = &sc->cfg;
    float result[3] = { 0, 0, 0 };
    float beta[3] = { 1.0f, 1.0f, 1.0f };
    if (adjoint) { result[0] = state_in[0]; result[1] = state_in[1]; result[2] = state_in[2]; }

    int active, depth, escaped;
    si_t si;
    if (ps) {                                            /* :61-67 */
        active = ps->active; depth = ps->depth; si = ps->si; escaped = ps->escaped;
    } else {
        active = 1; depth = 0; escaped = 0;
        (void) next_1d(S);                               /* :71 colour-channel draw */
        /* A3: reach_medium (:292-319) */
        si = box_hit(sc, ray.o, ray.d);
        if (!si.valid) { escaped = 1; active = 0; }
        else {
            ray.o = offset_p(&si, ray.d);                /* :306 */
            si_t sn = box_hit(sc, ray.o, ray.d);         /* :307 */
            if (!sn.valid) active = 0;                   /* :310 */
            else { ray.maxt = sn.t; si = sn; }           /* :316-317 */
        }
    }
    int has_scattered = ps ? (active && !escaped) : 0;   /* :84-89 */
    float last_pdf = ps ? ps->last_pdf : 1.0f;

    reservoir_t R;                                       /* :94-96 */
    memset(&R, 0, sizeof R);
    R.depth = -1; R.active = active;

    pcg32 A; A.state = 0; A.inc = 1;
    if (active) (void) next_1d(S);                       /* :99 alt_seed_rnd */
    if (adjoint) sampler_seed(&A, c->alt_seed, c->ray_index);   /* :100-107 */

    while (active) {                                     /* :114 */
        /* Russian roulette (:117-121); the draw is consumed every iteration */
        float q = fminf(max3f(beta[0], beta[1], beta[2]), 0.99f);
        int perform_rr = depth > cfg->rr_depth;
        float u_rr = next_1d(S);
        active = (beta[0] != 0.0f || beta[1] != 0.0f || beta[2] != 0.0f)
                 && (!perform_rr || (u_rr < q));
        if (perform_rr) { float iq = 1.0f / q; beta[0] *= iq; beta[1] *= iq; beta[2] *= iq; }
        if (!active) break;      /* everything below is masked by `active` */

        mei_t mei = sample_real_interaction(c, &ray, S, adjoint);   /* :126 */
        int did_escape = !mei.valid, did_scatter = mei.valid;       /* :130-134 */
        has_scattered |= did_scatter;

        float albedo[3] = { 1.0f, 1.0f, 1.0f };          /* :141 */
        if (did_scatter) { eval_albedo(sc, mei.p, albedo); c->cnt.n_alb++; }

        if (adjoint) {
            if (cfg->use_drt) {                          /* :143-150 */
                float adj[3] = { dL[0] * beta[0], dL[1] * beta[1], dL[2] * beta[2] };
                if (cfg->use_drt_subsampling) {          /* :521-539, DRTReservoir.update :745-753 */
                    float u = next_1d(&A);
                    float m = 0.0f;
                    for (int k = 0; k < 3; ++k) { R.wsum[k] += beta[k]; m += beta[k] / R.wsum[k]; }
                    m = m / 3.0f;
                    if (u <= m) {
                        R.cw[0] = beta[0]; R.cw[1] = beta[1]; R.cw[2] = beta[2];
                        R.depth = depth; R.si = si; R.ray = ray; R.active = 1;
                    }
                } else {
                    drt_backprop(c, &A, &ray, &si, depth, adj);
                }
            }
            if ((!cfg->use_drt || cfg->use_drt_mis) && did_scatter) {   /* :152-172 */
                float w = 1.0f;
                if (cfg->use_drt && cfg->use_drt_mis) {
                    float s2 = mei.sigma_t * mei.sigma_t;
                    w = s2 / (1.0f + s2);
                }
                float inv_pdf = 1.0f / mei.sigma_t;
                float gs = 0.0f, ga[3];
                for (int k = 0; k < 3; ++k) {
                    float Li = result[k] / fmaxf(1e-8f, albedo[k]);     /* :167 */
                    float a = ((w * dL[k]) * Li) * inv_pdf;
                    gs += a * albedo[k];
                    ga[k] = a * mei.sigma_t;
                }
                splat_sigma_t(c, mei.p, gs); c->cnt.n_sc++;
                splat_albedo(c, mei.p, ga);  c->cnt.n_sc_alb++;
            }
            /* :181-189 */
            backprop_transmittance(c, &A, &ray, did_escape ? si.t : mei.t, dL, result);
        }

        beta[0] *= albedo[0]; beta[1] *= albedo[1]; beta[2] *= albedo[2];   /* :193 */
        if (did_scatter) depth += 1;                     /* :199 */
        active = did_scatter && (depth < cfg->max_depth);/* :200 */

        if (cfg->use_nee && did_scatter && active) {     /* :206-215 */
            float nee[3];
            sample_emitter_for_nee(c, mei.p, S, beta, adjoint ? dL : NULL, nee);
            for (int k = 0; k < 3; ++k) result[k] = adjoint ? result[k] - nee[k] : result[k] + nee[k];
        }

        if (did_scatter) {                               /* :221-230 */
            (void) next_1d(S);
            float ux = next_1d(S), uy = next_1d(S);
            ray.o = mei.p; ray.d = square_to_uniform_sphere(ux, uy); ray.maxt = DRT_LARGEST;
            last_pdf = DRT_INV_FOURPI;
        }
        si = box_hit(sc, ray.o, ray.d);                  /* :233-235 (did_scatter | did_escape) */
        ray.maxt = isfinite(si.t) ? si.t : DRT_LARGEST;
        if (did_scatter && !si.valid) active = 0;        /* :240-241 accidental escape */
        if (did_escape) {                                /* :244-245 */
            if (si.valid) ray.o = offset_p(&si, ray.d);
            escaped = 1;
        }
    }

    if (adjoint && cfg->use_drt && cfg->use_drt_subsampling && R.active && R.depth >= 0) {  /* :249-259 */
        float d = ((R.cw[0] + R.cw[1]) + R.cw[2]) / 3.0f;      /* DRTReservoir.get :756-760 */
        float ws = ((R.wsum[0] + R.wsum[1]) + R.wsum[2]) / 3.0f;
        float adj[3];
        for (int k = 0; k < 3; ++k) adj[k] = (d != 0.0f ? (ws * R.cw[k]) / d : 0.0f) * dL[k];
        drt_backprop(c, &A, &R.ray, &R.si, R.depth, adj);
    }

    if (!adjoint) {                                      /* :263-287 envmap, primal only */
        if (escaped && !(depth <= 0 && cfg->hide_emitters)) {
            float w = 1.0f, Le[3] = { sc->Le[0], sc->Le[1], sc->Le[2] };
            if (cfg->use_nee) {
                float epdf = 0.0f;                                      /* :273-277 */
                if (has_scattered) epdf = sc->env.pix ? envmap_pdf(&sc->env, ray.d) : DRT_INV_FOURPI;
                w = mis_weight(last_pdf, epdf);
            }
            if (sc->env.pix) envmap_eval(&sc->env, ray.d, Le);          /* :284 */
            for (int k = 0; k < 3; ++k) result[k] += (beta[k] * w) * Le[k];
        }
    }
    out[0] = result[0]; out[1] = result[1]; out[2] = result[2];
}

/* ------------------------------------------------------------------------- */
/* A17: NeRFIntegrator.sample (python/integrators/nerf.py:47-148)             */
/* emission-absorption ray marching, queries_per_ray jittered queries,        */
/* PRB-style backward.  `emission` grid (Z,Y,X,3) = medium.get_emission.       */
/* ------------------------------------------------------------------------- */
static int scene_init(scene_t *sc, const drto_job *job);
static void scene_free(scene_t *sc);
static inline void job_ray(const drto_job *job, uint64_t i, pcg32 *S, ray_t *ray);
static void cnt_add(drto_counters *a, const drto_counters *b);

typedef struct {
    int hide_emitters, queries, jitter, relu;
    const float *emission;
    double *g_emission;
} nerf_t;

static void nerf_sample(ctx_t *c, const nerf_t *nf, pcg32 *S, int adjoint, ray_t ray, const float *dL,
                        const float *state_in, float out[3])
{
    const scene_t *sc = c->sc;
    float result[3] = { 0, 0, 0 };
    if (adjoint) { result[0] = state_in[0]; result[1] = state_in[1]; result[2] = state_in[2]; }
    float throughput = 1.0f, weights_sum = 0.0f;
    si_t si = box_hit(sc, ray.o, ray.d);                       /* nerf.py:67-79 */
    int active = si.valid, escaped = !active;
    if (active) {
        ray.o = offset_p(&si, ray.d);
        si = box_hit(sc, ray.o, ray.d);
        active = si.valid;
    }
    if (active) {
        const int N = nf->queries;
        float step = nf->jitter ? (si.t - 0.0f) / (float) N : (si.t - 0.0f) / (float)(N - 1);   /* nerf.py:6-10,82 */
        float t_a = 0.0f;
        float jit = next_1d(S);                               /* :88 one jitter per ray */
        for (int j = 0; j < N; ++j) {                         /* :94-129 */
            float t_b = nf->jitter ? step * ((float)(j + 1) + jit) : step * (float)(j + 1);   /* :12-17 */
            float dt = t_b - t_a;
            v3 p = ray_at(ray.o, ray.d, t_b);                 /* query_medium :151-165 */
            float raw = eval_sigma_t(sc, p);
            c->cnt.n_dt++;
            float sigma = nf->relu ? fmaxf(0.0f, raw) : raw;
            float em[3];
            { stencil_t s; make_stencil_colour(sc, p, &s); for (int k = 0; k < 3; ++k) em[k] = trilerp(&s, nf->emission, 3, k); }
            c->cnt.n_alb++;
            int last = !(j + 1 < N);
            float a = last ? 1.0f : drt_expf(-sigma * dt);    /* :104-106 */
            float weight = (1.0f - a) * throughput;
            float safe_a = a + 1e-10f;
            for (int k = 0; k < 3; ++k)
                result[k] = adjoint ? result[k] - weight * em[k] : result[k] + weight * em[k];   /* :110-113 */
            if (adjoint) {                                    /* :122-129 */
                /* d/d sigma of  dL * (em * weight + (result / det(safe_a)) * safe_a); zero at the last step */
                float gs = 0.0f, ge[3];
                float da = last ? 0.0f : -dt * a;             /* d a / d sigma */
                for (int k = 0; k < 3; ++k) {
                    gs += dL[k] * (em[k] * (-da * throughput) + (result[k] / safe_a) * da);
                    ge[k] = dL[k] * weight;
                }
                if (nf->relu && !(raw > 0.0f)) gs = 0.0f;
                splat_sigma_t(c, p, gs); c->cnt.n_sc++;
                {
                    stencil_t s; float w[8];
                    make_stencil_colour(sc, p, &s); stencil_weights(&s, w);
                    for (int q = 0; q < 8; ++q) for (int k = 0; k < 3; ++k) {
                        double v = (double)(w[q] * ge[k]);
#ifdef _OPENMP
#pragma omp atomic
#endif
                        nf->g_emission[(size_t) s.idx[q] * 3 + k] += v;
                    }
                    c->cnt.n_sc_alb++;
                }
            }
            t_a = t_b;
            if (!last) { throughput *= safe_a; weights_sum += weight; }   /* :117-120 (masked by still_walking) */
        }
    }
    /* composite with the background emitter (:131-146; executes in both modes, :144) */
    int active_e = escaped || active;
    if (nf->hide_emitters) active_e = active_e && (weights_sum > 0.0f);
    if (active_e) {
        float Le[3] = { sc->Le[0], sc->Le[1], sc->Le[2] };
        if (sc->env.pix) envmap_eval(&sc->env, ray.d, Le);
        for (int k = 0; k < 3; ++k) result[k] += (1.0f - weights_sum) * Le[k];
    }
    out[0] = result[0]; out[1] = result[1]; out[2] = result[2];
}

int drto_nerf_render(const drto_job *job, const drto_nerf_config *ncfg, const float *emission, int adjoint,
                     const float *dL, const float *L_in, float *L_out, double *grad_sigma_t,
                     double *grad_emission, drto_counters *cnt)
{
    scene_t sc;
    drto_job j2 = *job;
    drto_config dummy; memset(&dummy, 0, sizeof dummy);
    if (!j2.cfg) j2.cfg = &dummy;
    if (scene_init(&sc, &j2)) return -1;
    if (!emission || !ncfg || ncfg->queries_per_ray < 2) { scene_free(&sc); return -2; }
    nerf_t nf; nf.hide_emitters = ncfg->hide_emitters; nf.queries = ncfg->queries_per_ray;
    nf.jitter = ncfg->jittering_enabled; nf.relu = ncfg->activation_relu;
    nf.emission = emission; nf.g_emission = grad_emission;
    drto_counters total; memset(&total, 0, sizeof total);
#ifdef _OPENMP
    int nt = job->n_threads > 0 ? job->n_threads : omp_get_max_threads();
#pragma omp parallel num_threads(nt)
#endif
    {
        ctx_t c; memset(&c, 0, sizeof c);
        c.sc = &sc; c.g_sigma = grad_sigma_t;
#ifdef _OPENMP
#pragma omp for schedule(dynamic, 256)
#endif
        for (int64_t i = 0; i < (int64_t) job->n_rays; ++i) {
            pcg32 S; ray_t ray;
            job_ray(job, (uint64_t) i, &S, &ray);
            c.cnt.n_rays++;
            float L[3];
            if (adjoint) nerf_sample(&c, &nf, &S, 1, ray, dL + 3 * i, L_in + 3 * i, L);
            else {
                nerf_sample(&c, &nf, &S, 0, ray, NULL, NULL, L);
                L_out[3 * i] = L[0]; L_out[3 * i + 1] = L[1]; L_out[3 * i + 2] = L[2];
            }
        }
#ifdef _OPENMP
#pragma omp critical
#endif
        cnt_add(&total, &c.cnt);
    }
    if (cnt) cnt_add(cnt, &total);
    scene_free(&sc);
    return 0;
}

/* ------------------------------------------------------------------------- */
/* harness: scene setup, ray generation, job loops                            */
/* ------------------------------------------------------------------------- */
static int scene_init(scene_t *sc, const drto_job *job)
{
    const drto_medium *m = job->medium;
    if (!job->cfg || !m || !job->emitter || !m->sigma_t) return -1;
    sc->cfg = *job->cfg;
    sc->sigma_t = m->sigma_t; sc->albedo = m->albedo;
    sc->rx = m->res[0]; sc->ry = m->res[1]; sc->rz = m->res[2];
    sc->crx = sc->rx; sc->cry = sc->ry; sc->crz = sc->rz;
    if (m->res_colour[0] > 0 && m->res_colour[1] > 0 && m->res_colour[2] > 0) { sc->crx = m->res_colour[0]; sc->cry = m->res_colour[1]; sc->crz = m->res_colour[2]; }
    else if (m->res_colour[0] | m->res_colour[1] | m->res_colour[2]) return -2;
    sc->own_colour = sc->crx != sc->rx || sc->cry != sc->ry || sc->crz != sc->rz;
    sc->bmin = v3_make(m->bbox_min[0], m->bbox_min[1], m->bbox_min[2]);
    sc->bmax = v3_make(m->bbox_max[0], m->bbox_max[1], m->bbox_max[2]);
    sc->inv_ext = v3_make(1.0f / (sc->bmax.x - sc->bmin.x), 1.0f / (sc->bmax.y - sc->bmin.y),
                          1.0f / (sc->bmax.z - sc->bmin.z));
    sc->scale = m->scale;
    float mx = 0.0f;
    size_t n = (size_t) sc->rx * sc->ry * sc->rz;
    for (size_t i = 0; i < n; ++i) mx = fmaxf(mx, m->sigma_t[i]);
    sc->majorant = mx * m->scale;                         /* global majorant = scale * max(grid) */
    sc->inv_majorant = sc->majorant != 0.0f ? 1.0f / sc->majorant : 0.0f;
    for (int k = 0; k < 3; ++k) sc->Le[k] = job->emitter->radiance[k];
    sc->mgrid = NULL; sc->gx = sc->gy = sc->gz = 0;
    memset(&sc->env, 0, sizeof(sc->env));
    if (job->emitter->pixels) {
        int rc = envmap_from_emitter(&sc->env, job->emitter);
        if (rc) return rc;
    }
    if (m->majorant_factor > 0) {
        /* cell (I,J,K) covers 1/G of the box per axis; its majorant is scale * max over every
         * voxel a trilinear lookup inside the cell can touch, padded by one voxel:
         * [floor(I*res/G) - 1, ceil((I+1)*res/G)] per axis, clamped. */
        int f = m->majorant_factor;
        int R[3] = { sc->rx, sc->ry, sc->rz }, G[3];
        for (int a = 0; a < 3; ++a) { G[a] = R[a] / f; if (G[a] < 1) G[a] = 1; }
        sc->gx = G[0]; sc->gy = G[1]; sc->gz = G[2];
        sc->mgrid = (float *) malloc(sizeof(float) * 2 * (size_t) G[0] * G[1] * G[2]);
        if (!sc->mgrid) return -4;
        for (int K = 0; K < G[2]; ++K) for (int J = 0; J < G[1]; ++J) for (int I = 0; I < G[0]; ++I) {
            int c[3] = { I, J, K }, lo[3], hi[3];
            for (int a = 0; a < 3; ++a) {
                lo[a] = (int)(((long long) c[a] * R[a]) / G[a]) - 1;
                hi[a] = (int)((((long long)(c[a] + 1)) * R[a] + G[a] - 1) / G[a]);
                if (lo[a] < 0) lo[a] = 0;
                if (hi[a] > R[a] - 1) hi[a] = R[a] - 1;
            }
            float mxc = 0.0f;
            for (int z = lo[2]; z <= hi[2]; ++z) for (int y = lo[1]; y <= hi[1]; ++y) for (int x = lo[0]; x <= hi[0]; ++x)
                mxc = fmaxf(mxc, m->sigma_t[((size_t) z * R[1] + y) * R[0] + x]);
            float mm = mxc * m->scale;
            {   /* rounded UP to the next bf16-representable value: a majorant only has to bound (at most 0.8 % looser),
                 * and the device keeps the supergrid as 16-bit values in on-chip memory */
                uint32_t b; memcpy(&b, &mm, 4);
                if (b & 0xffffu) b = (b | 0xffffu) + 1u;
                memcpy(&mm, &b, 4);
            }
            float *dst = sc->mgrid + 2 * (size_t)((K * G[1] + J) * G[0] + I);
            dst[0] = mm; dst[1] = mm != 0.0f ? 1.0f / mm : 0.0f;
        }
    }
    return 0;
}

static void scene_free(scene_t *sc) { free(sc->mgrid); sc->mgrid = NULL; envmap_free(&sc->env); }

/* perspective sensor (tests/test_integrators.py:46-67): sample position in
 * [0,1]^2, (0,0) = top-left; camera x axis = `left`. */
static inline void sensor_ray(const drto_sensor *s, uint32_t pixel, float ux, float uy, v3 *o, v3 *d)
{
    uint32_t py = pixel / (uint32_t) s->width, px = pixel - py * (uint32_t) s->width;
    float sx = ((float) px + ux) * (1.0f / (float) s->width);
    float sy = ((float) py + uy) * (1.0f / (float) s->height);
    float cx = fmaf(-2.0f, sx, 1.0f) * s->tan_x;
    float cy = fmaf(-2.0f, sy, 1.0f) * s->tan_y;
    float inv = 1.0f / sqrtf(fmaf(cx, cx, fmaf(cy, cy, 1.0f)));
    cx *= inv; cy *= inv; float cz = inv;
    *o = v3_make(s->origin[0], s->origin[1], s->origin[2]);
    *d = v3_make(fmaf(s->left[0], cx, fmaf(s->up[0], cy, s->dir[0] * cz)),
                 fmaf(s->left[1], cx, fmaf(s->up[1], cy, s->dir[1] * cz)),
                 fmaf(s->left[2], cx, fmaf(s->up[2], cy, s->dir[2] * cz)));
}

static inline void job_ray(const drto_job *job, uint64_t i, pcg32 *S, ray_t *ray)
{
    uint32_t gi = (uint32_t)(job->ray_offset + i);
    sampler_seed(S, job->seed, gi);
    if (job->sensor) {
        float ux = next_1d(S), uy = next_1d(S);
        sensor_ray(job->sensor, gi / job->spp, ux, uy, &ray->o, &ray->d);
    } else {
        ray->o = v3_make(job->rays_o[3 * i], job->rays_o[3 * i + 1], job->rays_o[3 * i + 2]);
        ray->d = v3_make(job->rays_d[3 * i], job->rays_d[3 * i + 1], job->rays_d[3 * i + 2]);
    }
    ray->maxt = DRT_LARGEST;
}

/* volpathsimple.py:99-107: alt_seed = tea32(bits(lane-0 draw), 1)[0].  Lane 0's
 * draw index is fixed (2nd draw of its stream; 4th in the sensor flow). */
uint32_t drto_alt_seed(uint32_t seed, int sensor_flow)
{
    pcg32 S;
    sampler_seed(&S, seed, 0);
    int skip = sensor_flow ? 3 : 1;
    for (int k = 0; k < skip; ++k) (void) next_1d(&S);
    float u = next_1d(&S);
    uint32_t v0, v1;
    tea32(f2u(u), 1u, &v0, &v1);
    return v0;
}

static void cnt_add(drto_counters *a, const drto_counters *b)
{
    a->n_rays += b->n_rays; a->n_dt += b->n_dt; a->n_rt += b->n_rt; a->n_drt += b->n_drt;
    a->n_alb += b->n_alb; a->n_tr += b->n_tr; a->n_rt_adj += b->n_rt_adj;
    a->n_sc += b->n_sc; a->n_sc_alb += b->n_sc_alb;
}

static int run_job(const drto_job *job, int adjoint, const float *dL, const float *L_in,
                   float *L_out, double *g_sigma, double *g_albedo, drto_counters *cnt)
{
    scene_t sc;
    if (scene_init(&sc, job)) return -1;
    if (!sc.albedo) { scene_free(&sc); return -2; }
    uint32_t alt_seed = drto_alt_seed(job->seed, job->sensor != NULL);
    drto_counters total; memset(&total, 0, sizeof total);
#ifdef _OPENMP
    int nt = job->n_threads > 0 ? job->n_threads : omp_get_max_threads();
#else
    int nt = 1;
#endif
    /* binned mode (see gbucket_t): [thread][sigma_t layers | colour layers] buckets */
    const int binned = adjoint && job->grad_cache_log2 == -1;
    const int n_layers = sc.rz + sc.crz;
    gbucket_t *bins = binned ? (gbucket_t *) calloc((size_t) nt * n_layers, sizeof(gbucket_t)) : NULL;
#ifdef _OPENMP
#pragma omp parallel num_threads(nt)
#endif
    {
        ctx_t c; memset(&c, 0, sizeof c);
        c.sc = &sc; c.g_sigma = g_sigma; c.g_albedo = g_albedo; c.alt_seed = alt_seed;
#ifdef _OPENMP
        if (bins) c.gbin = bins + (size_t) omp_get_thread_num() * n_layers;
#else
        if (bins) c.gbin = bins;
#endif
        if (adjoint && job->grad_cache_log2 > 0 && job->grad_cache_log2 <= 24) {
            const uint32_t lines = 1u << job->grad_cache_log2;
            c.gcache = (gcache_line *) malloc((size_t) lines * sizeof(gcache_line));
            if (c.gcache) {
                c.gcache_mask = lines - 1u;
                for (uint32_t k = 0; k < lines; ++k) { c.gcache[k].tag = 0xffffffffu; c.gcache[k].v[0] = c.gcache[k].v[1] = c.gcache[k].v[2] = c.gcache[k].v[3] = 0.0; }
            }
        }
#ifdef _OPENMP
#pragma omp for schedule(dynamic, 256)
#endif
        for (int64_t i = 0; i < (int64_t) job->n_rays; ++i) {
            pcg32 S; ray_t ray;
            job_ray(job, (uint64_t) i, &S, &ray);
            c.ray_index = (uint32_t)(job->ray_offset + (uint64_t) i);
            c.cnt.n_rays++;
            float L[3];
            if (adjoint) drt_sample(&c, &S, 1, ray, dL + 3 * i, L_in + 3 * i, NULL, L);
            else {
                drt_sample(&c, &S, 0, ray, NULL, NULL, NULL, L);
                L_out[3 * i] = L[0]; L_out[3 * i + 1] = L[1]; L_out[3 * i + 2] = L[2];
            }
        }
        gcache_flush(&c);
        free(c.gcache);
        if (bins) {
            /* every thread's records are in: reduce the layers, even ones first (the implicit barriers of the two loops separate the sweeps) */
            for (int parity = 0; parity < 2; ++parity) {
#ifdef _OPENMP
#pragma omp for schedule(dynamic, 1)
#endif
                for (int l = parity; l < n_layers + (n_layers & 1); l += 2) {
                    /* (sigma_t layers and colour layers interleave in one index range: layer l of sigma_t and layer l' of the colour grid
                     *  write different grids) */
                    if (l >= n_layers) continue;
                    const int colour = l >= sc.rz;
                    for (int t = 0; t < nt; ++t) {
                        const gbucket_t *b = &bins[(size_t) t * n_layers + l];
                        for (uint32_t i = 0; i < b->n; ++i) {
                            const grec_t *q = &b->r[i];
                            stencil_t s; float w[8];
                            const v3 p = v3_make(q->p[0], q->p[1], q->p[2]);
                            if (!colour) {
                                make_stencil(&sc, p, &s); stencil_weights(&s, w);
                                const float gs = q->v[0] * sc.scale;
                                for (int k = 0; k < 8; ++k) g_sigma[s.idx[k]] += (double) (w[k] * gs);
                            } else {
                                make_stencil_colour(&sc, p, &s); stencil_weights(&s, w);
                                for (int k = 0; k < 8; ++k)
                                    for (int ch = 0; ch < 3; ++ch) g_albedo[(size_t) s.idx[k] * 3 + ch] += (double) (w[k] * q->v[ch]);
                            }
                        }
                    }
                }
            }
        }
#ifdef _OPENMP
#pragma omp critical
#endif
        cnt_add(&total, &c.cnt);
    }
    if (bins) {
        for (size_t i = 0; i < (size_t) nt * n_layers; ++i) free(bins[i].r);
        free(bins);
    }
    if (cnt) cnt_add(cnt, &total);
    scene_free(&sc);
    return 0;
}

int drto_render_primal(const drto_job *job, float *L_out, drto_counters *cnt)
{
    return run_job(job, 0, NULL, NULL, L_out, NULL, NULL, cnt);
}

int drto_render_backward(const drto_job *job, const float *dL, const float *L_in,
                         double *grad_sigma_t, double *grad_albedo, drto_counters *cnt)
{
    return run_job(job, 1, dL, L_in, NULL, grad_sigma_t, grad_albedo, cnt);
}

int drto_h1_step(const drto_job *job, float *L, float *image, double *loss_out,
                 double *grad_sigma_t, double *grad_albedo, drto_counters *cnt)
{
    int rc = drto_render_primal(job, L, cnt);             /* batched.py:255-264 */
    if (rc) return rc;
    uint64_t npix = job->n_rays / job->spp;
    double loss = 0.0;
    float inv_spp = 1.0f / (float) job->spp;
    /* box film: image[p] = mean_spp L (batched.py:272-297); loss = mean((img-0.5)^2) */
    for (uint64_t p = 0; p < npix; ++p)
        for (int k = 0; k < 3; ++k) {
            float s = 0.0f;
            for (uint32_t j = 0; j < job->spp; ++j) s += L[3 * (p * job->spp + j) + k];
            float v = s * inv_spp;
            image[3 * p + k] = v;
            loss += ((double) v - 0.5) * ((double) v - 0.5);
        }
    loss /= (double)(npix * 3);
    if (loss_out) *loss_out = loss;
    /* dL_i = dloss/dimage[p(i)] / spp  (batched.py:298-306) */
    float *dL = (float *) malloc(sizeof(float) * 3 * job->n_rays);
    if (!dL) return -3;
    float gscale = 2.0f / (float)(npix * 3);
    for (uint64_t i = 0; i < job->n_rays; ++i)
        for (int k = 0; k < 3; ++k)
            dL[3 * i + k] = (gscale * (image[3 * (i / job->spp) + k] - 0.5f)) * inv_spp;
    rc = drto_render_backward(job, dL, L, grad_sigma_t, grad_albedo, cnt);  /* batched.py:309-318 */
    free(dL);
    return rc;
}

/* Textbook analog delta-tracking path tracer: escape -> Le, collide -> albedo,
 * uniform phase.  No NEE/MIS; its own RNG consumption order. */
int drto_render_textbook(const drto_job *job, float *L_out)
{
    scene_t sc;
    if (scene_init(&sc, job)) return -1;
#ifdef _OPENMP
    int nt = job->n_threads > 0 ? job->n_threads : omp_get_max_threads();
#pragma omp parallel for schedule(dynamic, 256) num_threads(nt)
#endif
    for (int64_t i = 0; i < (int64_t) job->n_rays; ++i) {
        pcg32 S; ray_t ray;
        job_ray(job, (uint64_t) i, &S, &ray);
        float beta[3] = { 1, 1, 1 }, L[3] = { 0, 0, 0 };
        si_t si = box_hit(&sc, ray.o, ray.d);
        int inside = 0;
        if (si.valid) {
            ray.o = offset_p(&si, ray.d);
            si_t sn = box_hit(&sc, ray.o, ray.d);
            if (sn.valid) { inside = 1; ray.maxt = sn.t; }
        }
        int depth = 0, alive = 1;
        while (inside && alive) {
            float t = 0.0f; int real = 0;
            for (;;) {
                t += sample_distance(&sc, next_1d(&S));
                if (!(t <= ray.maxt)) break;
                float sig = eval_sigma_t(&sc, ray_at(ray.o, ray.d, t));
                if (next_1d(&S) * sc.majorant < sig) { real = 1; break; }
            }
            if (!real) break;                             /* escaped: add Le below */
            v3 p = ray_at(ray.o, ray.d, t);
            float a[3]; eval_albedo(&sc, p, a);
            beta[0] *= a[0]; beta[1] *= a[1]; beta[2] *= a[2];
            if (++depth >= sc.cfg.max_depth) { alive = 0; break; }
            float ux = next_1d(&S), uy = next_1d(&S);
            ray.o = p; ray.d = square_to_uniform_sphere(ux, uy);
            si_t sn = box_hit(&sc, ray.o, ray.d);
            if (!sn.valid) { alive = 0; break; }
            ray.maxt = sn.t;
        }
        if (alive) {
            float Le[3] = { sc.Le[0], sc.Le[1], sc.Le[2] };
            if (sc.env.pix) envmap_eval(&sc.env, ray.d, Le);
            for (int k = 0; k < 3; ++k) L[k] = beta[k] * Le[k];
        }
        L_out[3 * i] = L[0]; L_out[3 * i + 1] = L[1]; L_out[3 * i + 2] = L[2];
    }
    scene_free(&sc);
    return 0;
}

/* ------------------------------------------------------------------------- */
/* test hooks                                                                 */
/* ------------------------------------------------------------------------- */
static void scene_from_medium(scene_t *sc, const drto_medium *m)
{
    drto_config cfg; memset(&cfg, 0, sizeof cfg);
    drto_emitter em; memset(&em, 0, sizeof em);
    drto_job job; memset(&job, 0, sizeof job);
    job.cfg = &cfg; job.medium = m; job.emitter = &em;
    scene_init(sc, &job);
}
/* N1: sample_batch_pixels + sample_batch_rays (python/batched.py:397-467).
 * sampler 0 (wavefront = batch_size, seed sub_seed_pixels): sensor_idx = uint(n_sensors * next_1d()),
 * pixel = uint(film_size * next_2d()); ray sampler (wavefront = batch_size*spp, seed sub_seed_rays):
 * sub-pixel offset = next_2d(); pos = (pixel + offset) / film_size; ray = sensor.sample_ray(pos). */
void drto_batch_sample_rays(const drto_sensor *sensors, int n_sensors, uint32_t batch_size, uint32_t spp,
                            uint32_t sub_seed_pixels, uint32_t sub_seed_rays, float *rays_o, float *rays_d,
                            uint32_t *sensor_idx, uint32_t *pixels)
{
    for (uint32_t b = 0; b < batch_size; ++b) {
        pcg32 S0; sampler_seed(&S0, sub_seed_pixels, b);
        float us = next_1d(&S0), ux = next_1d(&S0), uy = next_1d(&S0);
        uint32_t si = (uint32_t)((float) n_sensors * us);
        if (si >= (uint32_t) n_sensors) si = (uint32_t) n_sensors - 1;
        const drto_sensor *sn = sensors + si;
        uint32_t px = (uint32_t)((float) sn->width * ux), py = (uint32_t)((float) sn->height * uy);
        if (sensor_idx) sensor_idx[b] = si;
        if (pixels) { pixels[2 * b] = px; pixels[2 * b + 1] = py; }
        for (uint32_t j = 0; j < spp; ++j) {
            uint32_t r = b * spp + j;
            pcg32 S1; sampler_seed(&S1, sub_seed_rays, r);
            float ox = next_1d(&S1), oy = next_1d(&S1);
            v3 o, d;
            sensor_ray(sn, py * (uint32_t) sn->width + px, ox, oy, &o, &d);
            rays_o[3 * r] = o.x; rays_o[3 * r + 1] = o.y; rays_o[3 * r + 2] = o.z;
            rays_d[3 * r] = d.x; rays_d[3 * r + 1] = d.y; rays_d[3 * r + 2] = d.z;
        }
    }
}

uint32_t drto_tea32(uint32_t v0, uint32_t v1, uint32_t *out_v1)
{
    uint32_t a, b; tea32(v0, v1, &a, &b); if (out_v1) *out_v1 = b; return a;
}
void drto_pcg32_floats(uint32_t seed, uint32_t index, int n, float *out)
{
    pcg32 S; sampler_seed(&S, seed, index);
    for (int i = 0; i < n; ++i) out[i] = next_1d(&S);
}
void drto_pcg32_raw(uint64_t initstate, uint64_t initseq, int n, uint32_t *out)
{
    pcg32 S; pcg32_seed(&S, initstate, initseq);
    for (int i = 0; i < n; ++i) out[i] = pcg32_next_u32(&S);
}
void drto_uniform_sphere(float ux, float uy, float out[3])
{
    v3 d = square_to_uniform_sphere(ux, uy);
    out[0] = d.x; out[1] = d.y; out[2] = d.z;
}
float drto_atan2f(float y, float x) { return drt_atan2f(y, x); }
int drto_envmap_eval(const drto_emitter *em, const float d[3], float out[3])
{
    envmap_t e; memset(&e, 0, sizeof(e));
    e.pix = em->pixels; e.w = em->width; e.h = em->height; e.scale = em->scale;
    for (int k = 0; k < 9; ++k) e.R[k] = em->to_world[k];
    envmap_eval(&e, v3_make(d[0], d[1], d[2]), out);
    return 0;
}
float drto_envmap_pdf(const drto_emitter *em, const float d[3])
{
    envmap_t e;
    if (envmap_from_emitter(&e, em)) return -1.0f;
    float p = envmap_pdf(&e, v3_make(d[0], d[1], d[2]));
    envmap_free(&e);
    return p;
}
int drto_envmap_sample(const drto_emitter *em, float u1, float u2, float d[3], float *pdf, float weight[3])
{
    envmap_t e;
    int rc = envmap_from_emitter(&e, em);
    if (rc) return rc;
    v3 dd;
    envmap_sample(&e, u1, u2, &dd, pdf, weight);
    d[0] = dd.x; d[1] = dd.y; d[2] = dd.z;
    envmap_free(&e);
    return 0;
}
int drto_envmap_tables(const drto_emitter *em, float *marginal, float *conditional)
{
    envmap_t e;
    int rc = envmap_from_emitter(&e, em);
    if (rc) return rc;
    if (marginal) memcpy(marginal, e.marg, sizeof(float) * (size_t)(e.h + 1));
    if (conditional) memcpy(conditional, e.cond, sizeof(float) * (size_t) e.h * (size_t)(e.w + 1));
    envmap_free(&e);
    return 0;
}
float drto_logf(float x) { return drt_logf(x); }
float drto_expf(float x) { return drt_expf(x); }
void drto_sincos_2pi(float u, float *s, float *c) { drt_sincos_2pi(u, s, c); }
float drto_eval_sigma_t(const drto_medium *m, const float p[3])
{
    scene_t sc; scene_from_medium(&sc, m);
    float r = eval_sigma_t(&sc, v3_make(p[0], p[1], p[2]));
    scene_free(&sc);
    return r;
}
void drto_eval_albedo(const drto_medium *m, const float p[3], float out[3])
{
    scene_t sc; scene_from_medium(&sc, m);
    eval_albedo(&sc, v3_make(p[0], p[1], p[2]), out);
    scene_free(&sc);
}
float drto_majorant(const drto_medium *m)
{
    scene_t sc; scene_from_medium(&sc, m);
    scene_free(&sc);
    return sc.majorant;
}
int drto_majorant_grid(const drto_medium *m, int32_t dims[3], float *out)
{
    scene_t sc; scene_from_medium(&sc, m);
    int n = sc.mgrid ? sc.gx * sc.gy * sc.gz : 0;
    dims[0] = sc.gx; dims[1] = sc.gy; dims[2] = sc.gz;
    if (out) for (int i = 0; i < n; ++i) out[i] = sc.mgrid[2 * i];
    scene_free(&sc);
    return n;
}
double drto_ratio_tracking_mean(const drto_medium *m, const float o[3], const float d[3],
                                float tmax, uint32_t seed, int n)
{
    scene_t sc; scene_from_medium(&sc, m);
    ctx_t c; memset(&c, 0, sizeof c); c.sc = &sc;
    double acc = 0.0;
    for (int i = 0; i < n; ++i) {
        pcg32 S; sampler_seed(&S, seed, (uint32_t) i);
        acc += (double) estimate_transmittance(&c, v3_make(o[0], o[1], o[2]),
                                               v3_make(d[0], d[1], d[2]), tmax, &S, NULL);
    }
    scene_free(&sc);
    return acc / (double) n;
}
/* E2 test hook: n independent walks from o along d to the box exit; walk i draws from the stream
 * PCG32(tea32(seed, first + i)).  Outputs per walk: valid, t', W.  Returns maxt. */
float drto_sample_interaction_drt(const drto_medium *m, const float o[3], const float d[3], uint32_t seed,
                                  uint32_t first, int n, int32_t *valid, float *t_out, float *W_out)
{
    scene_t sc; scene_from_medium(&sc, m);
    ctx_t c; memset(&c, 0, sizeof c); c.sc = &sc;
    ray_t r; r.o = v3_make(o[0], o[1], o[2]); r.d = v3_make(d[0], d[1], d[2]);
    si_t si = box_hit(&sc, r.o, r.d);
    r.maxt = si.valid ? si.t : 0.0f;
    for (int i = 0; i < n; ++i) {
        pcg32 A; sampler_seed(&A, seed, first + (uint32_t) i);
        valid[i] = sample_interaction_drt(&c, &r, &A, &t_out[i], &W_out[i]);
    }
    scene_free(&sc);
    return r.maxt;
}
int drto_box_hit(const drto_medium *m, const float o[3], const float d[3], float *t, float n[3])
{
    scene_t sc; scene_from_medium(&sc, m);
    si_t si = box_hit(&sc, v3_make(o[0], o[1], o[2]), v3_make(d[0], d[1], d[2]));
    *t = si.t; n[0] = si.n.x; n[1] = si.n.y; n[2] = si.n.z;
    scene_free(&sc);
    return si.valid;
}
void drto_sensor_ray(const drto_sensor *s, uint32_t pixel, float ux, float uy, float o[3], float d[3])
{
    v3 oo, dd; sensor_ray(s, pixel, ux, uy, &oo, &dd);
    o[0] = oo.x; o[1] = oo.y; o[2] = oo.z; d[0] = dd.x; d[1] = dd.y; d[2] = dd.z;
}
