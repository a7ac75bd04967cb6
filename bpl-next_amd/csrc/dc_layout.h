// dc_layout.h -- latent-vector layout and per-team parameter maps shared by the HIP
// kernels and the host-side helpers (constrain, init).  float64 throughout.
//
// Layout = numpyro's flat order (sorted site names) of the models declared at
//   basic    bpl/dixon_coles.py:39-84
//   extended bpl/extended_dixon_coles.py:78-248
#pragma once
#include <math.h>
#include <stdint.h>

#if defined(__HIPCC__)
#define DC_HD __host__ __device__ __forceinline__
#else
#define DC_HD inline
#endif

namespace dc {

constexpr int MODEL_BASIC = 0;
constexpr int MODEL_EXTENDED = 1;

// numpyro SigmoidTransform clips expit() to [finfo.tiny, 1 - finfo.eps]; the reference
// runs in float32 (JAX default), so these are the float32 constants.
constexpr double SIG_LO = 1.1754943508222875e-38;
constexpr double SIG_HI = 1.0 - 1.1920928955078125e-07;
constexpr double RATE_CLIP = 15.0;  // bpl/extended_dixon_coles.py:197-198
constexpr double LOG_RATE_CLIP = 2.70805020110221;
constexpr double HALF_LOG_2PI = 0.9189385332046727;
constexpr double LN2 = 0.6931471805599453;

struct Layout {
    int model, T, K, D;
    // basic
    int o_adec, o_ddec, o_ha;
    // extended
    int o_bA, o_bD, o_hadec, o_mha, o_sat, o_sdt, o_sh, o_u;
    // both
    int o_corr, o_md, o_sa, o_sd;
};

DC_HD Layout make_layout(int model, int T, int K) {
    Layout L{};
    L.model = model;
    L.T = T;
    L.K = K;
    if (model == MODEL_BASIC) {
        L.K = 0;
        L.o_adec = 0;
        L.o_corr = T;
        L.o_ddec = T + 1;
        L.o_ha = 2 * T + 1;
        L.o_md = 2 * T + 2;
        L.o_sa = 2 * T + 3;
        L.o_sd = 2 * T + 4;
        L.D = 2 * T + 5;
    } else {
        L.o_bA = 0;
        L.o_corr = K;
        L.o_bD = K + 1;
        L.o_hadec = 2 * K + 1;
        L.o_md = 2 * K + 1 + T;
        L.o_mha = 2 * K + 2 + T;
        L.o_sat = 2 * K + 3 + T;
        L.o_sdt = 2 * K + 3 + 2 * T;
        L.o_sa = 2 * K + 3 + 3 * T;
        L.o_sd = L.o_sa + 1;
        L.o_sh = L.o_sa + 2;
        L.o_u = L.o_sa + 3;
        L.D = 3 * T + 2 * K + 7;
    }
    return L;
}

DC_HD double sigmoid(double x) {
    if (x >= 0) return 1.0 / (1.0 + exp(-x));
    double e = exp(x);
    return e / (1.0 + e);
}
DC_HD double softplus(double x) { return fmax(x, 0.0) + log1p(exp(-fabs(x))); }

// clipped expit: value and derivative (0 where the clip binds)
DC_HD void clipped_sigmoid(double x, double* v, double* dv) {
    double s = sigmoid(x);
    if (s < SIG_LO) {
        *v = SIG_LO;
        *dv = 0.0;
    } else if (s > SIG_HI) {
        *v = SIG_HI;
        *dv = 0.0;
    } else {
        *v = s;
        *dv = s * (1.0 - s);
    }
}

// attack_t, defence_t, home_advantage_t of team t (constrained / deterministic sites).
//   basic    bpl/dixon_coles.py:51-61 (LocScaleReparam(centered=0))
//   extended bpl/extended_dixon_coles.py:124-146, 175-187
DC_HD void team_params(const Layout& L, const double* z, const double* xs, int t,
                       double* attack, double* defence, double* ha) {
    if (L.model == MODEL_BASIC) {
        *attack = exp(z[L.o_sa]) * z[L.o_adec + t];
        *defence = z[L.o_md] + exp(z[L.o_sd]) * z[L.o_ddec + t];
        *ha = z[L.o_ha];
    } else {
        double apm = 0.0, dpm = z[L.o_md];
        for (int k = 0; k < L.K; ++k) {
            double xv = xs[(size_t)t * L.K + k];
            apm += xv * z[L.o_bA + k];
            dpm += xv * z[L.o_bD + k];
        }
        *attack = apm + z[L.o_sat + t] * exp(z[L.o_sa]);
        *defence = dpm + z[L.o_sdt + t] * exp(z[L.o_sd]);
        *ha = z[L.o_mha] + exp(z[L.o_sh]) * z[L.o_hadec + t];
    }
}

}  // namespace dc
