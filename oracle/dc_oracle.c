/* ORACLE (test infrastructure, NOT product code) -- plain C float64 restatement of the
 * Dixon-Coles potential energy and gradient, the C twin of oracle/dc_oracle.py.
 *
 * PARITY UNPINNED: see the header of oracle/dc_oracle.py (the reference holds no golden
 * numbers for this path; numpyro/jax are not installed).  Pinned against dc_oracle.py
 * (itself cross-checked by autograd of a literal transcription and finite differences)
 * in tests/test_oracle.py.  Used only by tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg ("port": this is the CPU path timed beside the GPU).
 *
 * Follows: bpl/dixon_coles.py:39-84, bpl/extended_dixon_coles.py:78-248,
 * bpl/_util.py:17-31 (bounds), bpl/_util.py:35-93 (tau), numpyro 0.13.2 semantics
 * (SURVEY.md Appendix A).  The structure mirrors what XLA:CPU executes for the
 * reference: materialise the two rate vectors, three global max reductions, the masked
 * tau terms, then the scatter-add of the adjoint into the per-team vectors.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define HALF_LOG_2PI 0.9189385332046727
#define LN2 0.6931471805599453
#define RATE_CLIP 15.0
#define SIG_LO 1.1754943508222875e-38
#define SIG_HI (1.0 - 1.1920928955078125e-07)

typedef struct {
    int model, T, K, D;
    int o_adec, o_ddec, o_ha;
    int o_bA, o_bD, o_hadec, o_mha, o_sat, o_sdt, o_sh, o_u;
    int o_corr, o_md, o_sa, o_sd;
} layout_t;

static layout_t make_layout(int model, int T, int K) {
    layout_t L;
    memset(&L, 0, sizeof L);
    L.model = model; L.T = T; L.K = K;
    if (model == 0) {
        L.K = 0; L.o_adec = 0; L.o_corr = T; L.o_ddec = T + 1; L.o_ha = 2 * T + 1;
        L.o_md = 2 * T + 2; L.o_sa = 2 * T + 3; L.o_sd = 2 * T + 4; L.D = 2 * T + 5;
    } else {
        L.o_bA = 0; L.o_corr = K; L.o_bD = K + 1; L.o_hadec = 2 * K + 1;
        L.o_md = 2 * K + 1 + T; L.o_mha = 2 * K + 2 + T; L.o_sat = 2 * K + 3 + T;
        L.o_sdt = 2 * K + 3 + 2 * T; L.o_sa = 2 * K + 3 + 3 * T; L.o_sd = L.o_sa + 1;
        L.o_sh = L.o_sa + 2; L.o_u = L.o_sa + 3; L.D = 3 * T + 2 * K + 7;
    }
    return L;
}

static double sigmoid(double x) {
    if (x >= 0) return 1.0 / (1.0 + exp(-x));
    double e = exp(x);
    return e / (1.0 + e);
}
static double softplus(double x) { return fmax(x, 0.0) + log1p(exp(-fabs(x))); }
static void clipped_sigmoid(double x, double* v, double* dv) {
    double s = sigmoid(x);
    if (s < SIG_LO) { *v = SIG_LO; *dv = 0.0; }
    else if (s > SIG_HI) { *v = SIG_HI; *dv = 0.0; }
    else { *v = s; *dv = s * (1.0 - s); }
}

int dco_latent_dim(int model, int T, int K) { return make_layout(model, T, K).D; }

/* Returns 0 on success. w, xs (standardised covariates [T,K]) may be NULL. aux[4] =
 * {rho, LB, UB, q} may be NULL. nthreads <= 1 -> scalar loop. */
int dco_potential_grad(int model, int64_t n, int T, int K, const uint16_t* h,
                       const uint16_t* a, const uint8_t* x, const uint8_t* y,
                       const double* w, const double* xs, const double* z, double* U_out,
                       double* grad, double* aux, int nthreads) {
    const layout_t L = make_layout(model, T, K);
    const int clip = model == 1;
    double* att = (double*)malloc(sizeof(double) * 6 * (size_t)T);
    double* lam = (double*)malloc(sizeof(double) * 2 * (size_t)n);
    if (!att || !lam) { free(att); free(lam); return -4; }
    double *def = att + T, *ha = att + 2 * T, *g_att = att + 3 * T, *g_def = att + 4 * T,
           *g_ha = att + 5 * T;
    double* lh = lam;
    double* la = lam + n;
    double lg[256];
    for (int k = 0; k < 256; ++k) lg[k] = lgamma((double)k + 1.0);
    (void)nthreads;

    /* constrained / deterministic sites */
    const double s_a = exp(z[L.o_sa]), s_d = exp(z[L.o_sd]), m = z[L.o_md];
    double s_h = 0, mha = 0;
    if (model == 0) {
        for (int t = 0; t < T; ++t) {
            att[t] = s_a * z[L.o_adec + t];
            def[t] = m + s_d * z[L.o_ddec + t];
            ha[t] = z[L.o_ha];
        }
    } else {
        s_h = exp(z[L.o_sh]); mha = z[L.o_mha];
        for (int t = 0; t < T; ++t) {
            double apm = 0.0, dpm = m;
            for (int k = 0; k < K; ++k) {
                apm += xs[(size_t)t * K + k] * z[L.o_bA + k];
                dpm += xs[(size_t)t * K + k] * z[L.o_bD + k];
            }
            att[t] = apm + z[L.o_sat + t] * s_a;
            def[t] = dpm + z[L.o_sdt + t] * s_d;
            ha[t] = mha + s_h * z[L.o_hadec + t];
        }
    }
    double q, dq;
    clipped_sigmoid(z[L.o_corr], &q, &dq);

    /* pass 1: rates + the three global max reductions (bpl/_util.py:23-30) */
    double M = 0.0, Lh = 0.0, La = 0.0;
    int64_t iP = 0, iQ = 0, iR = 0;
#ifdef _OPENMP
#pragma omp parallel num_threads(nthreads > 1 ? nthreads : 1)
#endif
    {
        double M_ = 0.0, Lh_ = 0.0, La_ = 0.0;
        int64_t iP_ = -1, iQ_ = -1, iR_ = -1;
#ifdef _OPENMP
#pragma omp for schedule(static) nowait
#endif
        for (int64_t i = 0; i < n; ++i) {
            double l1 = exp(att[h[i]] - def[a[i]] + ha[h[i]]);
            double l2 = exp(att[a[i]] - def[h[i]]);
            if (clip) { l1 = fmin(l1, RATE_CLIP); l2 = fmin(l2, RATE_CLIP); }
            lh[i] = l1; la[i] = l2;
            if (l1 * l2 > M_ || iP_ < 0) { M_ = l1 * l2; iP_ = i; }
            if (l1 > Lh_ || iQ_ < 0) { Lh_ = l1; iQ_ = i; }
            if (l2 > La_ || iR_ < 0) { La_ = l2; iR_ = i; }
        }
#ifdef _OPENMP
#pragma omp critical
#endif
        {
            if (iP_ >= 0 && (M_ > M || (M_ == M && iP_ < iP) )) { M = M_; iP = iP_; }
            if (iQ_ >= 0 && (Lh_ > Lh || (Lh_ == Lh && iQ_ < iQ))) { Lh = Lh_; iQ = iQ_; }
            if (iR_ >= 0 && (La_ > La || (La_ == La && iR_ < iR))) { La = La_; iR = iR_; }
        }
    }
    const double UB = M > 1.0 ? 1.0 / M : 1.0;
    const double LB = -1.0 / fmax(Lh, La);
    const double rho = LB + q * (UB - LB);

    /* pass 2: Poisson + tau value, adjoint wrt eta, scatter-add to the teams */
    for (int t = 0; t < 3 * T; ++t) g_att[t] = 0.0;
    double Llik = 0.0, G_rho = 0.0;
#ifdef _OPENMP
#pragma omp parallel num_threads(nthreads > 1 ? nthreads : 1)
#endif
    {
        double* loc = (double*)calloc(3 * (size_t)T, sizeof(double));
        double Ll = 0.0, Gr = 0.0;
#ifdef _OPENMP
#pragma omp for schedule(static) nowait
#endif
        for (int64_t i = 0; i < n; ++i) {
            const double wi = w ? w[i] : 1.0;
            const double l1 = lh[i], l2 = la[i];
            const int xi = x[i], yi = y[i];
            const int c1 = clip && l1 >= RATE_CLIP && exp(att[h[i]] - def[a[i]] + ha[h[i]]) > RATE_CLIP;
            const int c2 = clip && l2 >= RATE_CLIP && exp(att[a[i]] - def[h[i]]) > RATE_CLIP;
            Ll += wi * (xi * log(l1) - l1 - lg[xi] + yi * log(l2) - l2 - lg[yi]);
            double bar1 = wi * (xi / l1 - 1.0), bar2 = wi * (yi / l2 - 1.0);
            if (xi <= 1 && yi <= 1) {
                double c, d1 = 0.0, d2 = 0.0; /* c: arg = 1 + rho c ; d = dc/dlam */
                if (xi == 0 && yi == 0) { c = -l1 * l2; d1 = -l2; d2 = -l1; }
                else if (xi == 1 && yi == 0) { c = l2; d2 = 1.0; }
                else if (xi == 0 && yi == 1) { c = l1; d1 = 1.0; }
                else { c = -1.0; }
                const double arg = 1.0 + rho * c;
                if (arg > 0.0) {
                    Ll += wi * log(arg);
                    Gr += wi * c / arg;
                    bar1 += wi * rho * d1 / arg;
                    bar2 += wi * rho * d2 / arg;
                } else {
                    Ll += wi * log(0.0); /* -inf (tol = 0, bpl/_util.py:42) */
                }
            }
            /* rho-coupling through the arg-extremal fixtures (Appendix A.3) is added
             * after the reduction (needs the complete G_rho) */
            const double g1 = c1 ? 0.0 : bar1 * l1, g2 = c2 ? 0.0 : bar2 * l2;
            loc[h[i]] += g1;          loc[T + a[i]] -= g1;  loc[2 * T + h[i]] += g1;
            loc[a[i]] += g2;          loc[T + h[i]] -= g2;
        }
#ifdef _OPENMP
#pragma omp critical
#endif
        {
            for (int t = 0; t < 3 * T; ++t) g_att[t] += loc[t];
            Llik += Ll; G_rho += Gr;
        }
        free(loc);
    }
    {   /* adjoint of the bounds */
        const int cP1 = clip && exp(att[h[iP]] - def[a[iP]] + ha[h[iP]]) > RATE_CLIP;
        const int cP2 = clip && exp(att[a[iP]] - def[h[iP]]) > RATE_CLIP;
        if (M > 1.0) {
            const double v = G_rho * q * (-UB);
            if (!cP1) { g_att[h[iP]] += v; g_ha[h[iP]] += v; g_def[a[iP]] -= v; }
            if (!cP2) { g_att[a[iP]] += v; g_def[h[iP]] -= v; }
        }
        const double v = G_rho * (1.0 - q) * (-LB);
        if (Lh >= La) {
            const int cQ = clip && exp(att[h[iQ]] - def[a[iQ]] + ha[h[iQ]]) > RATE_CLIP;
            if (!cQ) { g_att[h[iQ]] += v; g_ha[h[iQ]] += v; g_def[a[iQ]] -= v; }
        } else {
            const int cR = clip && exp(att[a[iR]] - def[h[iR]]) > RATE_CLIP;
            if (!cR) { g_att[a[iR]] += v; g_def[h[iR]] -= v; }
        }
    }

    /* priors + Jacobians + chain rule */
    const double zc = z[L.o_corr];
    double Lp = log(q) + log1p(-q) + log(6.0) - softplus(zc) - softplus(-zc);
    const double g_corr = G_rho * (UB - LB) * dq + (1.0 / q - 1.0 / (1.0 - q)) * dq +
                          (1.0 - 2.0 * sigmoid(zc));
    Lp += -0.5 * s_a * s_a - HALF_LOG_2PI + LN2 + z[L.o_sa];
    Lp += -0.5 * s_d * s_d - HALF_LOG_2PI + LN2 + z[L.o_sd];
    Lp += -0.5 * m * m - HALF_LOG_2PI;
    double sum_gd = 0, sum_gh = 0, dot_a = 0, dot_d = 0, dot_h = 0;
    if (model == 0) {
        const double gam = z[L.o_ha];
        Lp += -0.5 * ((gam - 0.1) / 0.2) * ((gam - 0.1) / 0.2) - log(0.2) - HALF_LOG_2PI;
        for (int t = 0; t < T; ++t) {
            const double ad = z[L.o_adec + t], dd = z[L.o_ddec + t];
            Lp += -0.5 * ad * ad - 0.5 * dd * dd - 2.0 * HALF_LOG_2PI;
            grad[L.o_adec + t] = -(s_a * g_att[t] - ad);
            grad[L.o_ddec + t] = -(s_d * g_def[t] - dd);
            sum_gd += g_def[t]; sum_gh += g_ha[t];
            dot_a += ad * g_att[t]; dot_d += dd * g_def[t];
        }
        grad[L.o_ha] = -(sum_gh - (gam - 0.1) / 0.04);
    } else {
        Lp += -0.5 * ((mha - 0.1) / 0.2) * ((mha - 0.1) / 0.2) - log(0.2) - HALF_LOG_2PI;
        Lp += -0.5 * s_h * s_h - HALF_LOG_2PI + LN2 + z[L.o_sh];
        double u, du;
        const double zu = z[L.o_u];
        clipped_sigmoid(zu, &u, &du);
        Lp += log(u) + 3.0 * log1p(-u) + log(20.0) - softplus(zu) - softplus(-zu);
        const double rp = 2.0 * u - 1.0, vv = 1.0 - rp * rp;
        double dL_drp = 0.0;
        for (int t = 0; t < T; ++t) {
            const double sa = z[L.o_sat + t], sd = z[L.o_sdt + t], hd = z[L.o_hadec + t];
            const double e = sd - rp * sa;
            Lp += -0.5 * sa * sa - 0.5 * e * e / vv - 0.5 * log(vv) - 0.5 * hd * hd -
                  3.0 * HALF_LOG_2PI;
            grad[L.o_sat + t] = -(s_a * g_att[t] - sa + rp * e / vv);
            grad[L.o_sdt + t] = -(s_d * g_def[t] - e / vv);
            grad[L.o_hadec + t] = -(s_h * g_ha[t] - hd);
            sum_gd += g_def[t]; sum_gh += g_ha[t];
            dot_a += sa * g_att[t]; dot_d += sd * g_def[t]; dot_h += hd * g_ha[t];
            dL_drp += e * sa / vv - rp * e * e / (vv * vv) + rp / vv;
        }
        for (int k = 0; k < K; ++k) {
            const double ba = z[L.o_bA + k], bd = z[L.o_bD + k];
            Lp += -0.5 * ba * ba - 0.5 * bd * bd - 2.0 * HALF_LOG_2PI;
            double sA = 0.0, sD = 0.0;
            for (int t = 0; t < T; ++t) {
                sA += xs[(size_t)t * K + k] * g_att[t];
                sD += xs[(size_t)t * K + k] * g_def[t];
            }
            grad[L.o_bA + k] = -(sA - ba);
            grad[L.o_bD + k] = -(sD - bd);
        }
        grad[L.o_mha] = -(sum_gh - (mha - 0.1) / 0.04);
        grad[L.o_sh] = -(s_h * dot_h - s_h * s_h + 1.0);
        grad[L.o_u] = -(2.0 * dL_drp * du + (1.0 / u - 3.0 / (1.0 - u)) * du +
                        (1.0 - 2.0 * sigmoid(zu)));
    }
    grad[L.o_md] = -(sum_gd - m);
    grad[L.o_sa] = -(s_a * dot_a - s_a * s_a + 1.0);
    grad[L.o_sd] = -(s_d * dot_d - s_d * s_d + 1.0);
    grad[L.o_corr] = -g_corr;
    *U_out = -(Lp + Llik);
    if (aux) { aux[0] = rho; aux[1] = LB; aux[2] = UB; aux[3] = q; }
    free(att);
    free(lam);
    return 0;
}

int dco_max_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
