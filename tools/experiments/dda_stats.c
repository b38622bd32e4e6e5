/* Experiment (not a test, not the product): what would empty-space skipping save in the supergrid DDA?
 * Includes the oracle with its DDA hooks defined, records the cells every flight visits and replays them under
 * several skipping schemes, counting operations.  Build + run: tools/experiments/dda_stats.py */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
struct scene_s;
static void dda_visit(const void *sc, int cx, int cy, int cz, float m);
static void dda_end(const void *sc);
#define DRTO_DDA_VISIT(sc, cx, cy, cz, m) dda_visit(sc, cx, cy, cz, m)
#define DRTO_DDA_END(sc) dda_end(sc)
#include "../../oracle/drt_oracle.c"

#define MAXC 4096
static __thread int n_vis; static __thread short vis[MAXC][3]; static __thread unsigned char vis_ne[MAXC];
static unsigned char *g_df = NULL; static const float *g_df_for = NULL; static int g_G[3];
/* stats */
enum { S_FLIGHTS, S_CELLS, S_EMPTY, S_DF1_J, S_DF1_S, S_DF2_J, S_DF2_S, S_DF3_J, S_DF3_S, S_M2_J, S_M2_S, S_M4_J, S_M4_S, S_M8_J, S_M8_S,
       S_ORTH_S, S_ODF_J, S_ODF_S, S_DD3_J, S_DD3_S, S_DD7_J, S_DD7_S, S_DD15_J, S_DD15_S, S_DD15M2_J, S_DD15M2_S, S_DD15M3_J, S_DD15M3_S, S_JLEN, S_MD_E_J, S_MD_E_S, S_MD_K4_J, S_MD_K4_S, S_MD_K8_J, S_MD_K8_S, S_MD_S_J, S_MD_S_S, S_MD_K8R_J, S_MD_K8R_S, S_FD_K8_J, S_FD_K8_S, S_FD_K4_J, S_FD_K4_S, S_LEAD_EMPTY, S_TRAIL_EMPTY, S_ALL_EMPTY_FLIGHTS, S_ALL_EMPTY_CELLS, S_N };
static unsigned long long g_stat[S_N];
static unsigned long long g_runhist[64];      /* empty-run length histogram (capped 63) */
static unsigned long long g_lenhist[64];

static short *g_ddf = NULL; static short *g_mdf = NULL; static int g_MG[3];  /* [8][cells] */
static void build_df(const scene_t *sc)
{
#pragma omp critical
    {
        if (g_df_for != sc->mgrid) {
            int gx = sc->gx, gy = sc->gy, gz = sc->gz; g_G[0] = gx; g_G[1] = gy; g_G[2] = gz;
            free(g_df); g_df = (unsigned char *) malloc((size_t) gx * gy * gz);
            for (int z = 0; z < gz; ++z) for (int y = 0; y < gy; ++y) for (int x = 0; x < gx; ++x) {
                int d = 0;
                if (sc->mgrid[2 * ((z * gy + y) * gx + x)] > 0.0f) { g_df[(z * gy + y) * gx + x] = 255; continue; }
                for (d = 1; d < 64; ++d) {
                    int ok = 1;
                    for (int zz = z - d; zz <= z + d && ok; ++zz) for (int yy = y - d; yy <= y + d && ok; ++yy) for (int xx = x - d; xx <= x + d; ++xx) {
                        if (zz < 0 || yy < 0 || xx < 0 || zz >= gz || yy >= gy || xx >= gx) continue;
                        if (sc->mgrid[2 * ((zz * gy + yy) * gx + xx)] > 0.0f) { ok = 0; break; }
                    }
                    if (!ok) break;
                }
                g_df[(z * gy + y) * gx + x] = (unsigned char) (d - 1);
            }
            {
                size_t n = (size_t) gx * gy * gz; free(g_ddf); g_ddf = (short *) malloc(sizeof(short) * 8 * n);
                for (int o = 0; o < 8; ++o) {
                    int sx = (o & 1) ? -1 : 1, sy = (o & 2) ? -1 : 1, sz = (o & 4) ? -1 : 1;
                    for (int zi = 0; zi < gz; ++zi) for (int yi = 0; yi < gy; ++yi) for (int xi = 0; xi < gx; ++xi) {
                        int x = sx > 0 ? gx - 1 - xi : xi, y = sy > 0 ? gy - 1 - yi : yi, z = sz > 0 ? gz - 1 - zi : zi;
                        size_t c = ((size_t) z * gy + y) * gx + x;
                        if (sc->mgrid[2 * c] > 0.0f) { g_ddf[o * n + c] = -1; continue; }
                        int best = 30000;
                        for (int m = 1; m < 8; ++m) {
                            int xx = x + ((m & 1) ? sx : 0), yy = y + ((m & 2) ? sy : 0), zz = z + ((m & 4) ? sz : 0);
                            int v = 30000;
                            if (xx >= 0 && yy >= 0 && zz >= 0 && xx < gx && yy < gy && zz < gz) v = g_ddf[o * n + ((size_t) zz * gy + yy) * gx + xx];
                            if (v < best) best = v;
                        }
                        g_ddf[o * n + c] = (short) (best >= 30000 ? 30000 : best + 1);
                    }
                }
            }
            {
                int mx = (gx + 1) / 2, my = (gy + 1) / 2, mz = (gz + 1) / 2; g_MG[0] = mx; g_MG[1] = my; g_MG[2] = mz;
                size_t n = (size_t) mx * my * mz; free(g_mdf); g_mdf = (short *) malloc(sizeof(short) * 8 * n);
                unsigned char *ne = (unsigned char *) calloc(n, 1);
                for (int z = 0; z < gz; ++z) for (int y = 0; y < gy; ++y) for (int x = 0; x < gx; ++x)
                    if (sc->mgrid[2 * (((size_t) z * gy + y) * gx + x)] > 0.0f) ne[((size_t) (z / 2) * my + y / 2) * mx + x / 2] = 1;
                for (int o = 0; o < 8; ++o) {
                    int sx = (o & 1) ? -1 : 1, sy = (o & 2) ? -1 : 1, sz = (o & 4) ? -1 : 1;
                    for (int zi = 0; zi < mz; ++zi) for (int yi = 0; yi < my; ++yi) for (int xi = 0; xi < mx; ++xi) {
                        int x = sx > 0 ? mx - 1 - xi : xi, y = sy > 0 ? my - 1 - yi : yi, z = sz > 0 ? mz - 1 - zi : zi;
                        size_t c = ((size_t) z * my + y) * mx + x;
                        if (ne[c]) { g_mdf[o * n + c] = -1; continue; }
                        int best = 30000;
                        for (int m = 1; m < 8; ++m) {
                            int xx = x + ((m & 1) ? sx : 0), yy = y + ((m & 2) ? sy : 0), zz = z + ((m & 4) ? sz : 0);
                            int v = 30000;
                            if (xx >= 0 && yy >= 0 && zz >= 0 && xx < mx && yy < my && zz < mz) v = g_mdf[o * n + ((size_t) zz * my + yy) * mx + xx];
                            if (v < best) best = v;
                        }
                        g_mdf[o * n + c] = (short) (best >= 30000 ? 30000 : best + 1);
                    }
                }
                free(ne);
            }
            g_df_for = sc->mgrid;
        }
    }
}

static void dda_visit(const void *scv, int cx, int cy, int cz, float m)
{
    (void) scv;
    if (n_vis < MAXC) { vis[n_vis][0] = (short) cx; vis[n_vis][1] = (short) cy; vis[n_vis][2] = (short) cz; vis_ne[n_vis] = m > 0.0f; ++n_vis; }
}

static void replay_df(int dmin, unsigned long long *J, unsigned long long *S, const int *G)
{
    int i = 0;
    while (i < n_vis) {
        int D = vis_ne[i] ? 0 : g_df[(vis[i][2] * G[1] + vis[i][1]) * G[0] + vis[i][0]];
        if (!vis_ne[i] && D >= dmin) {
            int j = i;
            while (j < n_vis && abs(vis[j][0] - vis[i][0]) <= D && abs(vis[j][1] - vis[i][1]) <= D && abs(vis[j][2] - vis[i][2]) <= D) ++j;
            ++*J; i = j;
        } else { ++*S; ++i; }
    }
}
static int flight_oct(void)
{
    int o = 0;
    for (int i = 1; i < n_vis; ++i) { if (vis[i][0] < vis[i-1][0]) o |= 1; if (vis[i][1] < vis[i-1][1]) o |= 2; if (vis[i][2] < vis[i-1][2]) o |= 4; }
    return o;
}
/* directional distance jumps: cap = largest stored distance, dmin = smallest distance worth a jump; orth: distance "infinite" ends the flight for free */
static void replay_ddf(int cap, int dmin, unsigned long long *J, unsigned long long *S, const int *G, unsigned long long *jlen)
{
    int o = flight_oct(); size_t n = (size_t) G[0] * G[1] * G[2];
    int i = 0;
    while (i < n_vis) {
        int D = g_ddf[o * n + ((size_t) vis[i][2] * G[1] + vis[i][1]) * G[0] + vis[i][0]];
        if (D >= 30000) { ++*S; return; }        /* nothing ahead: the step that sees it ends the flight */
        if (D > cap) D = cap;
        if (D >= dmin) {
            int j = i;
            while (j < n_vis && abs(vis[j][0] - vis[i][0]) <= D && abs(vis[j][1] - vis[i][1]) <= D && abs(vis[j][2] - vis[i][2]) <= D) ++j;
            ++*J; if (jlen) *jlen += j - i; i = j;
        } else { ++*S; ++i; }
    }
}
/* fine-cell reach from the macro-2 directional distances (cap 15 macro cells); -1: not jumpable */
static int md_reach(int o, int i)
{
    size_t n = (size_t) g_MG[0] * g_MG[1] * g_MG[2];
    int D = g_mdf[o * n + ((size_t) (vis[i][2] / 2) * g_MG[1] + vis[i][1] / 2) * g_MG[0] + vis[i][0] / 2];
    if (D < 0) return -1;
    if (D > 15) D = 15;
    return 2 * D;       /* (the +1 of cells at the near end of their macro cell is not used) */
}
static int fd_reach(int o, int i, const int *G)
{
    size_t n = (size_t) G[0] * G[1] * G[2];
    int D = g_ddf[o * n + ((size_t) vis[i][2] * G[1] + vis[i][1]) * G[0] + vis[i][0]];
    if (D < 0) return -1;
    return D > 15 ? 15 : D;
}
/* K = 0: a jump whenever the cell allows one; K > 0: jumps only every K steps (and at the start); K < 0: at the start only.  rep: jumps per check */
static void replay_sched(int fine, int K, int dmin, int rep, unsigned long long *J, unsigned long long *S, const int *G)
{
    int o = flight_oct();
    int i = 0, since = 0;
    while (i < n_vis) {
        int check = K == 0 || (K > 0 && since % K == 0) || (K < 0 && since == 0);
        int jumped = 0;
        if (check) for (int r = 0; r < rep && i < n_vis; ++r) {
            int D = fine ? fd_reach(o, i, G) : md_reach(o, i);
            if (D < dmin) break;
            int j = i;
            while (j < n_vis && abs(vis[j][0] - vis[i][0]) <= D && abs(vis[j][1] - vis[i][1]) <= D && abs(vis[j][2] - vis[i][2]) <= D) ++j;
            ++*J; i = j; jumped = 1;
        }
        (void) jumped;
        if (i < n_vis) { ++*S; ++i; ++since; }
    }
}
static void replay_macro(int M, unsigned long long *J, unsigned long long *S, const scene_t *sc)
{
    int i = 0;
    while (i < n_vis) {
        int mx = vis[i][0] / M, my = vis[i][1] / M, mz = vis[i][2] / M, empty = 1;
        if (vis_ne[i]) empty = 0;
        for (int z = mz * M; z < (mz + 1) * M && z < sc->gz && empty; ++z) for (int y = my * M; y < (my + 1) * M && y < sc->gy && empty; ++y)
            for (int x = mx * M; x < (mx + 1) * M && x < sc->gx; ++x) if (sc->mgrid[2 * ((z * sc->gy + y) * sc->gx + x)] > 0.0f) { empty = 0; break; }
        if (empty) {
            int j = i;
            while (j < n_vis && vis[j][0] / M == mx && vis[j][1] / M == my && vis[j][2] / M == mz) ++j;
            ++*J; i = j;
        } else { ++*S; ++i; }
    }
}

static void dda_end(const void *scv)
{
    const scene_t *sc = (const scene_t *) scv;
    if (g_df_for != sc->mgrid) build_df(sc);
    unsigned long long st[S_N]; memset(st, 0, sizeof st);
    st[S_FLIGHTS] = 1; st[S_CELLS] = n_vis;
    int run = 0, ne = 0;
    for (int i = 0; i < n_vis; ++i) {
        if (!vis_ne[i]) { ++st[S_EMPTY]; ++run; }
        else { ++ne; if (run) {
#pragma omp atomic
            g_runhist[run > 63 ? 63 : run]++;
            } run = 0; }
    }
    if (run) {
#pragma omp atomic
        g_runhist[run > 63 ? 63 : run]++;
    }
#pragma omp atomic
    g_lenhist[n_vis > 63 ? 63 : n_vis]++;
    int lead = 0; while (lead < n_vis && !vis_ne[lead]) ++lead;
    st[S_LEAD_EMPTY] = lead;
    if (lead == n_vis) { st[S_ALL_EMPTY_FLIGHTS] = 1; st[S_ALL_EMPTY_CELLS] = n_vis; }
    else { int tr = 0; while (!vis_ne[n_vis - 1 - tr]) ++tr; st[S_TRAIL_EMPTY] = tr; }
    replay_df(1, &st[S_DF1_J], &st[S_DF1_S], g_G);
    replay_df(2, &st[S_DF2_J], &st[S_DF2_S], g_G);
    replay_df(3, &st[S_DF3_J], &st[S_DF3_S], g_G);
    { unsigned long long dummy = 0; replay_ddf(0, 1, &dummy, &st[S_ORTH_S], g_G, NULL); }
    replay_ddf(3, 1, &st[S_DD3_J], &st[S_DD3_S], g_G, NULL);
    replay_ddf(7, 1, &st[S_DD7_J], &st[S_DD7_S], g_G, NULL);
    replay_ddf(15, 1, &st[S_DD15_J], &st[S_DD15_S], g_G, NULL);
    replay_ddf(15, 2, &st[S_DD15M2_J], &st[S_DD15M2_S], g_G, &st[S_JLEN]);
    replay_ddf(15, 3, &st[S_DD15M3_J], &st[S_DD15M3_S], g_G, NULL);
    replay_sched(0, 0, 2, 1, &st[S_MD_E_J], &st[S_MD_E_S], g_G);
    replay_sched(0, 4, 2, 1, &st[S_MD_K4_J], &st[S_MD_K4_S], g_G);
    replay_sched(0, 8, 2, 1, &st[S_MD_K8_J], &st[S_MD_K8_S], g_G);
    replay_sched(0, -1, 2, 4, &st[S_MD_S_J], &st[S_MD_S_S], g_G);
    replay_sched(0, 8, 2, 2, &st[S_MD_K8R_J], &st[S_MD_K8R_S], g_G);
    replay_sched(1, 8, 2, 1, &st[S_FD_K8_J], &st[S_FD_K8_S], g_G);
    replay_sched(1, 4, 2, 1, &st[S_FD_K4_J], &st[S_FD_K4_S], g_G);
    replay_macro(2, &st[S_M2_J], &st[S_M2_S], sc);
    replay_macro(4, &st[S_M4_J], &st[S_M4_S], sc);
    replay_macro(8, &st[S_M8_J], &st[S_M8_S], sc);
    for (int k = 0; k < S_N; ++k) if (st[k]) {
#pragma omp atomic
        g_stat[k] += st[k];
    }
    n_vis = 0;
}

void dda_stats_get(unsigned long long *stat, unsigned long long *runhist, unsigned long long *lenhist)
{
    memcpy(stat, g_stat, sizeof g_stat); memcpy(runhist, g_runhist, sizeof g_runhist); memcpy(lenhist, g_lenhist, sizeof g_lenhist);
    memset(g_stat, 0, sizeof g_stat); memset(g_runhist, 0, sizeof g_runhist); memset(g_lenhist, 0, sizeof g_lenhist);
}
