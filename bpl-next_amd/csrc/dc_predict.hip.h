// dc_predict.hip.h -- the predict path on the device:
// `predict_score_proba` of bpl/dixon_coles.py:139-163 / bpl/extended_dixon_coles.py:360-399:
//     mean over posterior draws s of  exp(corr_term_s) * Poisson(x; lh_s) * Poisson(y; la_s)
// with the rates of `_calculate_expected_goals` (bpl/dixon_coles.py:126-137) and the tau
// term of bpl/_util.py:35-93 evaluated per draw with that draw's corr_coef (tol = 0).
// One thread per requested (home, away, x, y) entry, a float64 loop over the S draws; the
// draws ([S,T] tables, 8 B elements) stay L2 resident.  base.py's grid / outcome / n-goals
// methods call this with (max_goals+1)^2 scorelines per fixture, like the reference does.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace dcp {

struct PredictArgs {
    int S, T;
    const double* attack;    // [S,T]
    const double* defence;   // [S,T]
    const double* home_adv;  // [S] (ha_stride = 0) or [S,T] (ha_stride = T)
    int ha_stride;
    const double* corr;      // [S]
    long long M;
    const uint16_t* h;
    const uint16_t* a;
    const uint16_t* x;       // goals as given (may exceed 255 in a query)
    const uint16_t* y;
    double* out;             // [M]
};

__global__ __launch_bounds__(256) void predict_score_proba(PredictArgs A) {
    const long long m = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (m >= A.M) return;
    const int h = A.h[m], a = A.a[m], x = A.x[m], y = A.y[m];
    const double lgx = lgamma((double)x + 1.0), lgy = lgamma((double)y + 1.0);
    const bool low = x <= 1 && y <= 1;
    double acc = 0.0;
    for (int s = 0; s < A.S; ++s) {
        const size_t r = (size_t)s * A.T;
        const double ha = A.ha_stride ? A.home_adv[r + h] : A.home_adv[s];
        const double eh = A.attack[r + h] - A.defence[r + a] + ha;
        const double ea = A.attack[r + a] - A.defence[r + h];
        const double lh = exp(eh), la = exp(ea);
        // exp(Poisson.log_prob) = exp(k log(rate) - lgamma(k+1) - rate)
        double p = exp(x * eh - lh - lgx) * exp(y * ea - la - lgy);
        if (low) {
            const double rho = A.corr[s];
            const double c = x == 0 ? (y == 0 ? -lh * la : lh) : (y == 0 ? la : -1.0);
            p *= fmax(1.0 + rho * c, 0.0);  // exp(log(clip(., 0)))
        }
        acc += p;
    }
    A.out[m] = acc / (double)A.S;
}

}  // namespace dcp
