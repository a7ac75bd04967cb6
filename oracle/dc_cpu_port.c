/* TEST / BENCH INFRASTRUCTURE (NOT product code) -- the CPU comparator timed beside the GPU
 * (bench.py `cpu_baseline`, kind "port"; SURVEY.md section 8d item 1: "C/OpenMP, fp32 compute +
 * fp64 accumulate, OMP 1 / all cores").  It runs the SAME algorithm as the HIP kernel
 * (bpl-next_amd/csrc/dc_kernels.hip.h), written for host cores:
 *
 *   - a handle owns every buffer (no allocation per evaluation);
 *   - at creation, as bplhip_set_fixtures does: the fixtures are sorted by (home, away) pair
 *     (runs of equal pairs become contiguous) and the data-only sums are taken (sum of goals per
 *     team and role, sum of lgamma terms);
 *   - per evaluation: per-team float32 tables {exp(att+ha), exp(-def)}, {exp(att), exp(-def)};
 *     rho from the maxima over the unique pairs; ONE fused OpenMP pass over the sorted goal
 *     bytes [+ f32 weights] in which the two rates and the four score-class tau terms are
 *     computed once per run piece and every fixture is only classified (branch-free byte
 *     compares the compiler vectorises) -- the HIP kernel's per-lane scheme; per-thread float64
 *     accumulators [3T + 4] (no atomics, no critical section), a fixed-order reduction of the
 *     per-thread blocks, and the float64 epilogue (priors, Jacobians, adjoint of the bounds,
 *     chain rule).
 *
 * Mathematics: SURVEY.md Appendix A (bpl/dixon_coles.py:39-84, bpl/extended_dixon_coles.py:78-248,
 * bpl/_util.py:17-93).  Checked against the float64 oracle in tests/test_oracle.py at float32
 * tolerances.  Only tests/ and bench.py's cpu_baseline leg may load it. */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define HALF_LOG_2PI 0.9189385332046727
#define LN2 0.6931471805599453
#define RATE_CLIP 15.0f
#define LOG_RATE_CLIP 2.70805020110221
#define SIG_LO 1.1754943508222875e-38
#define SIG_HI (1.0 - 1.1920928955078125e-07)
#define PAD 8 /* doubles of padding between per-thread blocks (false sharing) */

typedef struct {
    int model, T, K, D, nthreads, P, nacc;
    int64_t n;
    const uint16_t *h, *a;
    const uint8_t *x, *y;
    const float* w;     /* or NULL */
    const double* xs;   /* [T,K] standardised, or NULL */
    uint32_t* pairs;    /* unique home | away << 16, in sorted order = run order */
    int64_t* run_off;   /* [P + 1] first sorted fixture of each pair's run */
    uint8_t *sx, *sy;   /* goals sorted by pair */
    float* sw;          /* weights sorted by pair, or NULL */
    double *cA, *cD, *cH, lgsum;
    float *tabH, *tabA; /* [T][2] */
    double* acc;        /* [nthreads][nacc + PAD] */
    double *att, *def, *ha, *g; /* [T] each, g = [3T + 4] reduced */
    int o_adec, o_ddec, o_ha, o_bA, o_bD, o_hadec, o_mha, o_sat, o_sdt, o_sh, o_u, o_corr, o_md, o_sa, o_sd;
} port_t;

static double sigmoid(double v) {
    if (v >= 0) return 1.0 / (1.0 + exp(-v));
    const double e = exp(v);
    return e / (1.0 + e);
}
static double softplus(double v) { return fmax(v, 0.0) + log1p(exp(-fabs(v))); }
static void clipped_sigmoid(double v, double* s, double* ds) {
    const double t = sigmoid(v);
    if (t < SIG_LO) { *s = SIG_LO; *ds = 0.0; }
    else if (t > SIG_HI) { *s = SIG_HI; *ds = 0.0; }
    else { *s = t; *ds = t * (1.0 - t); }
}

void dcp_destroy(port_t* p) {
    if (!p) return;
    free(p->pairs); free(p->cA); free(p->tabH); free(p->acc); free(p->att);
    free(p->run_off); free(p->sx); free(p->sw);
    free(p);
}

/* The arrays are borrowed (kept alive by the caller).  w: float32 weights or NULL. */
port_t* dcp_create(int model, int64_t n, int T, int K, const uint16_t* h, const uint16_t* a,
                   const uint8_t* x, const uint8_t* y, const float* w, const double* xs,
                   int nthreads) {
    port_t* p = (port_t*)calloc(1, sizeof *p);
    if (!p) return NULL;
    p->model = model; p->T = T; p->K = model == 1 ? K : 0; p->n = n;
    p->h = h; p->a = a; p->x = x; p->y = y; p->w = w; p->xs = xs;
    K = p->K;
    if (model == 0) {
        p->o_adec = 0; p->o_corr = T; p->o_ddec = T + 1; p->o_ha = 2 * T + 1; p->o_md = 2 * T + 2;
        p->o_sa = 2 * T + 3; p->o_sd = 2 * T + 4; p->D = 2 * T + 5;
    } else {
        p->o_bA = 0; p->o_corr = K; p->o_bD = K + 1; p->o_hadec = 2 * K + 1; p->o_md = 2 * K + 1 + T;
        p->o_mha = 2 * K + 2 + T; p->o_sat = 2 * K + 3 + T; p->o_sdt = 2 * K + 3 + 2 * T;
        p->o_sa = 2 * K + 3 + 3 * T; p->o_sd = p->o_sa + 1; p->o_sh = p->o_sa + 2; p->o_u = p->o_sa + 3;
        p->D = 3 * T + 2 * K + 7;
    }
#ifdef _OPENMP
    p->nthreads = nthreads > 0 ? nthreads : omp_get_max_threads();
#else
    p->nthreads = 1;
#endif
    p->nacc = 3 * T + 4;
    p->cA = (double*)calloc(3 * (size_t)T, sizeof(double));
    p->cD = p->cA + T; p->cH = p->cA + 2 * T;
    p->tabH = (float*)calloc(4 * (size_t)T, sizeof(float));
    p->tabA = p->tabH + 2 * T;
    p->acc = (double*)calloc((size_t)p->nthreads * (p->nacc + PAD), sizeof(double));
    p->att = (double*)calloc(3 * (size_t)T + p->nacc, sizeof(double));
    p->def = p->att + T; p->ha = p->att + 2 * T; p->g = p->att + 3 * T;
    int64_t* cnt = (int64_t*)calloc((size_t)T * T + 1, sizeof(int64_t));
    p->sx = (uint8_t*)malloc(2 * (size_t)n + 64);
    p->sy = p->sx ? p->sx + n : NULL;
    p->sw = w ? (float*)malloc(sizeof(float) * (size_t)n + 64) : NULL;
    if (!p->cA || !p->tabH || !p->acc || !p->att || !cnt || !p->sx || (w && !p->sw)) {
        free(cnt); dcp_destroy(p); return NULL;
    }
    double lg[256];
    for (int k = 0; k < 256; ++k) lg[k] = lgamma((double)k + 1.0);
    for (int64_t i = 0; i < n; ++i) {  /* data-only sums, once */
        const double wi = w ? (double)w[i] : 1.0;
        p->cA[h[i]] += wi * x[i]; p->cA[a[i]] += wi * y[i];
        p->cD[a[i]] += wi * x[i]; p->cD[h[i]] += wi * y[i];
        p->cH[h[i]] += wi * x[i];
        p->lgsum += wi * (lg[x[i]] + lg[y[i]]);
        cnt[(size_t)h[i] * T + a[i] + 1] += 1;
    }
    /* counting sort by pair (stable) */
    int P = 0;
    for (size_t k = 0; k < (size_t)T * T; ++k) P += cnt[k + 1] > 0;
    p->pairs = (uint32_t*)malloc(sizeof(uint32_t) * (size_t)(P > 0 ? P : 1));
    p->run_off = (int64_t*)malloc(sizeof(int64_t) * ((size_t)P + 1));
    if (!p->pairs || !p->run_off) { free(cnt); dcp_destroy(p); return NULL; }
    p->P = 0;
    int64_t off = 0;
    for (size_t k = 0; k < (size_t)T * T; ++k) {
        const int64_t c = cnt[k + 1];
        cnt[k + 1] = off;  /* becomes the write cursor of cell k */
        if (c > 0) {
            p->pairs[p->P] = (uint32_t)(k / T) | ((uint32_t)(k % T) << 16);
            p->run_off[p->P++] = off;
        }
        off += c;
    }
    p->run_off[p->P] = off;
    for (int64_t i = 0; i < n; ++i) {
        const int64_t d = cnt[(size_t)h[i] * T + a[i] + 1]++;
        p->sx[d] = x[i]; p->sy[d] = y[i];
        if (w) p->sw[d] = w[i];
    }
    free(cnt);
    return p;
}

int dcp_latent_dim(const port_t* p) { return p->D; }
int dcp_threads(const port_t* p) { return p->nthreads; }

/* one evaluation: U and grad[D]; aux[4] = {rho, LB, UB, q} may be NULL */
int dcp_eval(port_t* p, const double* z, double* U_out, double* grad, double* aux) {
    const int T = p->T, K = p->K, clip = p->model == 1;
    double *att = p->att, *def = p->def, *ha = p->ha;
    const double s_a = exp(z[p->o_sa]), s_d = exp(z[p->o_sd]), m = z[p->o_md];
    double s_h = 0.0, mha = 0.0;
    for (int t = 0; t < T; ++t) {
        if (p->model == 0) {
            att[t] = s_a * z[p->o_adec + t];
            def[t] = m + s_d * z[p->o_ddec + t];
            ha[t] = z[p->o_ha];
        } else {
            double apm = 0.0, dpm = m;
            for (int k = 0; k < K; ++k) {
                apm += p->xs[(size_t)t * K + k] * z[p->o_bA + k];
                dpm += p->xs[(size_t)t * K + k] * z[p->o_bD + k];
            }
            s_h = exp(z[p->o_sh]); mha = z[p->o_mha];
            att[t] = apm + z[p->o_sat + t] * s_a;
            def[t] = dpm + z[p->o_sdt + t] * s_d;
            ha[t] = mha + s_h * z[p->o_hadec + t];
        }
        const float edn = expf((float)-def[t]);
        p->tabH[2 * t] = expf((float)(att[t] + ha[t])); p->tabH[2 * t + 1] = edn;
        p->tabA[2 * t] = expf((float)att[t]);           p->tabA[2 * t + 1] = edn;
    }
    double q, dq;
    clipped_sigmoid(z[p->o_corr], &q, &dq);
    /* bounds over the unique pairs (bpl/_util.py:23-30): max over fixtures = max over pairs */
    float M = 0.f, Lh = 0.f, La = 0.f;
    uint32_t pP = 0, pQ = 0, pR = 0;
    for (int k = 0; k < p->P; ++k) {
        const int hh = p->pairs[k] & 0xFFFF, aa = p->pairs[k] >> 16;
        float l1 = p->tabH[2 * hh] * p->tabA[2 * aa + 1], l2 = p->tabA[2 * aa] * p->tabH[2 * hh + 1];
        if (clip) { l1 = fminf(l1, RATE_CLIP); l2 = fminf(l2, RATE_CLIP); }
        if (l1 * l2 > M || k == 0) { M = l1 * l2; pP = p->pairs[k]; }
        if (l1 > Lh || k == 0) { Lh = l1; pQ = p->pairs[k]; }
        if (l2 > La || k == 0) { La = l2; pR = p->pairs[k]; }
    }
    const double UB = M > 1.0f ? 1.0 / (double)M : 1.0;
    const double LB = -1.0 / (double)fmaxf(Lh, La);
    const double rho_d = LB + q * (UB - LB);
    const float rho = (float)rho_d;

    /* ---- the fused pass over the (pair-sorted) fixtures */
    const int nacc = p->nacc, stride = nacc + PAD;
    const int64_t n = p->n;
    const uint8_t *sx = p->sx, *sy = p->sy;
    const float* sw = p->sw;
    const float *tabH = p->tabH, *tabA = p->tabA;
#ifdef _OPENMP
#pragma omp parallel num_threads(p->nthreads)
#endif
    {
#ifdef _OPENMP
        const int tid = omp_get_thread_num(), nt = omp_get_num_threads();
#else
        const int tid = 0, nt = 1;
#endif
        double* acc = p->acc + (size_t)tid * stride;
        for (int k = 0; k < nacc; ++k) acc[k] = 0.0;
        double* aA = acc; double* aD = acc + T; double* aH = acc + 2 * T;
        double slam = 0.0, slog = 0.0, su = 0.0, sclip = 0.0;
        /* this thread's contiguous slice of the sorted fixtures, walked run piece by run piece */
        const int64_t i0 = n * tid / nt, i1 = n * (tid + 1) / nt;
        int r = 0;
        {   /* first run that reaches past i0 (binary search) */
            int lo = 0, hi2 = p->P;
            while (lo < hi2) { const int mid = (lo + hi2) / 2; if (p->run_off[mid + 1] <= i0) lo = mid + 1; else hi2 = mid; }
            r = lo;
        }
        for (int64_t i = i0; i < i1; ++r) {
            const int64_t e = p->run_off[r + 1] < i1 ? p->run_off[r + 1] : i1;
            const int hh = p->pairs[r] & 0xFFFF, aa = p->pairs[r] >> 16;
            float l1 = tabH[2 * hh] * tabA[2 * aa + 1];
            float l2 = tabA[2 * aa] * tabH[2 * hh + 1];
            const float r1 = l1, r2 = l2;
            const int c1 = clip && l1 > RATE_CLIP, c2 = clip && l2 > RATE_CLIP;
            if (c1) l1 = RATE_CLIP;
            if (c2) l2 = RATE_CLIP;
            /* classify: (weighted) counts of the four low-score classes, total weight, goal sums */
            float n00 = 0.f, n10 = 0.f, n01 = 0.f, n11 = 0.f, nall = 0.f, gx = 0.f, gy = 0.f;
            if (!sw) {
                uint32_t k00 = 0, k10 = 0, k01 = 0, k11 = 0, kx = 0, ky = 0;
                for (int64_t j = i; j < e; ++j) {
                    const uint8_t xx = sx[j], yy = sy[j];
                    k00 += (xx | yy) == 0;
                    k10 += (xx == 1) & (yy == 0);
                    k01 += (xx == 0) & (yy == 1);
                    k11 += (xx == 1) & (yy == 1);
                    kx += xx; ky += yy;
                }
                n00 = (float)k00; n10 = (float)k10; n01 = (float)k01; n11 = (float)k11;
                nall = (float)(e - i); gx = (float)kx; gy = (float)ky;
            } else {
                for (int64_t j = i; j < e; ++j) {
                    const uint8_t xx = sx[j], yy = sy[j];
                    const float wj = sw[j];
                    n00 += (xx | yy) == 0 ? wj : 0.f;
                    n10 += ((xx == 1) & (yy == 0)) ? wj : 0.f;
                    n01 += ((xx == 0) & (yy == 1)) ? wj : 0.f;
                    n11 += ((xx == 1) & (yy == 1)) ? wj : 0.f;
                    nall += wj; gx += wj * xx; gy += wj * yy;
                }
            }
            /* tau (bpl/_util.py:58-91): arg = 1 + rho c, c = -l1 l2 | l2 | l1 | -1 */
            const float cc[4] = {-l1 * l2, l2, l1, -1.0f};
            const float nn[4] = {n00, n10, n01, n11};
            float uu[4];
            for (int k = 0; k < 4; ++k) {
                const float t = 1.0f + rho * cc[k];
                uu[k] = t > 0.0f ? cc[k] / t : 0.0f;  /* dlog tau / d rho */
                if (nn[k] != 0.0f) slog += (double)(nn[k] * logf(fmaxf(t, 0.0f)));
                su += (double)(nn[k] * uu[k]);
            }
            slam += (double)(nall * (l1 + l2));
            /* -(dL/d eta) without the goal counts: l - rho u [x == 0] (home), [y == 0] (away);
             * a clipped rate has zero gradient (cancel the goal count added in the epilogue) and
             * k eta -> k log 15 in the value */
            float gh = nall * l1 - rho * (n00 * uu[0] + n01 * uu[2]);
            float ga = nall * l2 - rho * (n00 * uu[0] + n10 * uu[1]);
            if (c1) { gh = gx; sclip += (double)gx * (log((double)r1) - LOG_RATE_CLIP); }
            if (c2) { ga = gy; sclip += (double)gy * (log((double)r2) - LOG_RATE_CLIP); }
            aA[hh] += gh; aH[hh] += gh; aD[aa] += gh;
            aA[aa] += ga; aD[hh] += ga;
            i = e;
        }
        acc[3 * T] = slam; acc[3 * T + 1] = slog; acc[3 * T + 2] = su; acc[3 * T + 3] = sclip;
#ifdef _OPENMP
#pragma omp barrier
        /* fixed-order reduction of the per-thread blocks */
#pragma omp for schedule(static)
#endif
        for (int k = 0; k < nacc; ++k) {
            double s2 = 0.0;
            for (int t2 = 0; t2 < nt; ++t2) s2 += p->acc[(size_t)t2 * stride + k];
            p->g[k] = s2;
        }
    }
    const double* raw = p->g;  /* rawA | rawD | rawH | SLAM SLOG SU SCLIP */
    const double SLAM = raw[3 * T], SLOG = raw[3 * T + 1], G_rho = raw[3 * T + 2], SCLIP = raw[3 * T + 3];

    /* ---- epilogue (float64): dL/d attack_t etc. = goal counts - raw sums, bounds adjoint */
    double* g_att = p->acc;  /* reuse thread 0's block as scratch: [3T] */
    double* g_def = g_att + T;
    double* g_ha = g_att + 2 * T;
    double Llin = 0.0;
    for (int t = 0; t < T; ++t) {
        g_att[t] = p->cA[t] - raw[t];
        g_def[t] = -(p->cD[t] - raw[T + t]);
        g_ha[t] = p->cH[t] - raw[2 * T + t];
        Llin += att[t] * p->cA[t] - def[t] * p->cD[t] + ha[t] * p->cH[t];
    }
    if (p->P > 0) {
        if (M > 1.0f) {
            const double v = G_rho * q * (-UB);
            const int hh = pP & 0xFFFF, aa = pP >> 16;
            const int c1 = clip && p->tabH[2 * hh] * p->tabA[2 * aa + 1] > RATE_CLIP;
            const int c2 = clip && p->tabA[2 * aa] * p->tabH[2 * hh + 1] > RATE_CLIP;
            if (!c1) { g_att[hh] += v; g_ha[hh] += v; g_def[aa] -= v; }
            if (!c2) { g_att[aa] += v; g_def[hh] -= v; }
        }
        const double v = G_rho * (1.0 - q) * (-LB);
        if (Lh >= La) {
            const int hh = pQ & 0xFFFF, aa = pQ >> 16;
            if (!(clip && p->tabH[2 * hh] * p->tabA[2 * aa + 1] > RATE_CLIP)) { g_att[hh] += v; g_ha[hh] += v; g_def[aa] -= v; }
        } else {
            const int hh = pR & 0xFFFF, aa = pR >> 16;
            if (!(clip && p->tabA[2 * aa] * p->tabH[2 * hh + 1] > RATE_CLIP)) { g_att[aa] += v; g_def[hh] -= v; }
        }
    }
    const double zc = z[p->o_corr];
    double Lp = log(q) + log1p(-q) + log(6.0) - softplus(zc) - softplus(-zc);
    Lp += -0.5 * s_a * s_a - HALF_LOG_2PI + LN2 + z[p->o_sa];
    Lp += -0.5 * s_d * s_d - HALF_LOG_2PI + LN2 + z[p->o_sd];
    Lp += -0.5 * m * m - HALF_LOG_2PI;
    double sum_gd = 0, sum_gh = 0, dot_a = 0, dot_d = 0, dot_h = 0;
    if (p->model == 0) {
        const double gam = z[p->o_ha];
        Lp += -0.5 * ((gam - 0.1) / 0.2) * ((gam - 0.1) / 0.2) - log(0.2) - HALF_LOG_2PI;
        for (int t = 0; t < T; ++t) {
            const double ad = z[p->o_adec + t], dd = z[p->o_ddec + t];
            Lp += -0.5 * ad * ad - 0.5 * dd * dd - 2.0 * HALF_LOG_2PI;
            grad[p->o_adec + t] = -(s_a * g_att[t] - ad);
            grad[p->o_ddec + t] = -(s_d * g_def[t] - dd);
            sum_gd += g_def[t]; sum_gh += g_ha[t];
            dot_a += ad * g_att[t]; dot_d += dd * g_def[t];
        }
        grad[p->o_ha] = -(sum_gh - (gam - 0.1) / 0.04);
    } else {
        Lp += -0.5 * ((mha - 0.1) / 0.2) * ((mha - 0.1) / 0.2) - log(0.2) - HALF_LOG_2PI;
        Lp += -0.5 * s_h * s_h - HALF_LOG_2PI + LN2 + z[p->o_sh];
        double u, du;
        const double zu = z[p->o_u];
        clipped_sigmoid(zu, &u, &du);
        Lp += log(u) + 3.0 * log1p(-u) + log(20.0) - softplus(zu) - softplus(-zu);
        const double rp = 2.0 * u - 1.0, vv = 1.0 - rp * rp;
        double dL_drp = 0.0;
        for (int t = 0; t < T; ++t) {
            const double sa = z[p->o_sat + t], sd = z[p->o_sdt + t], hd = z[p->o_hadec + t];
            const double e = sd - rp * sa;
            Lp += -0.5 * sa * sa - 0.5 * e * e / vv - 0.5 * log(vv) - 0.5 * hd * hd - 3.0 * HALF_LOG_2PI;
            grad[p->o_sat + t] = -(s_a * g_att[t] - sa + rp * e / vv);
            grad[p->o_sdt + t] = -(s_d * g_def[t] - e / vv);
            grad[p->o_hadec + t] = -(s_h * g_ha[t] - hd);
            sum_gd += g_def[t]; sum_gh += g_ha[t];
            dot_a += sa * g_att[t]; dot_d += sd * g_def[t]; dot_h += hd * g_ha[t];
            dL_drp += e * sa / vv - rp * e * e / (vv * vv) + rp / vv;
        }
        for (int k = 0; k < K; ++k) {
            const double ba = z[p->o_bA + k], bd = z[p->o_bD + k];
            Lp += -0.5 * ba * ba - 0.5 * bd * bd - 2.0 * HALF_LOG_2PI;
            double sA = 0.0, sD = 0.0;
            for (int t = 0; t < T; ++t) {
                sA += p->xs[(size_t)t * K + k] * g_att[t];
                sD += p->xs[(size_t)t * K + k] * g_def[t];
            }
            grad[p->o_bA + k] = -(sA - ba);
            grad[p->o_bD + k] = -(sD - bd);
        }
        grad[p->o_mha] = -(sum_gh - (mha - 0.1) / 0.04);
        grad[p->o_sh] = -(s_h * dot_h - s_h * s_h + 1.0);
        grad[p->o_u] = -(2.0 * dL_drp * du + (1.0 / u - 3.0 / (1.0 - u)) * du + (1.0 - 2.0 * sigmoid(zu)));
    }
    grad[p->o_md] = -(sum_gd - m);
    grad[p->o_sa] = -(s_a * dot_a - s_a * s_a + 1.0);
    grad[p->o_sd] = -(s_d * dot_d - s_d * s_d + 1.0);
    grad[p->o_corr] = -(G_rho * (UB - LB) * dq + (1.0 / q - 1.0 / (1.0 - q)) * dq + (1.0 - 2.0 * sigmoid(zc)));
    *U_out = -(Lp + Llin - SLAM - p->lgsum + SLOG - SCLIP);
    if (aux) { aux[0] = rho_d; aux[1] = LB; aux[2] = UB; aux[3] = q; }
    return 0;
}

/* `count` evaluations of zs[i % n_z] back to back (the timing loop stays out of Python) */
double dcp_eval_many(port_t* p, const double* zs, int n_z, int count, double* U_last, double* grad_last) {
    double U = 0.0;
    for (int i = 0; i < count; ++i) dcp_eval(p, zs + (size_t)(i % n_z) * p->D, &U, grad_last, NULL);
    *U_last = U;
    return U;
}
