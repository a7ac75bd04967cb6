// dc_predict.hip.h -- the predict path on the device:
// `predict_score_proba` of bpl/dixon_coles.py:139-163 / bpl/extended_dixon_coles.py:360-399:
//     mean over posterior draws s of  exp(corr_term_s) * Poisson(x; lh_s) * Poisson(y; la_s)
// with the rates of `_calculate_expected_goals` (bpl/dixon_coles.py:126-137) and the tau
// term of bpl/_util.py:35-93 evaluated per draw with that draw's corr_coef (tol = 0).
//
// Two kernels:
//   predict_score_grid   THE predict primitive: the whole (G+1) x (G+1) scoreline grid of a
//       fixture (predict_score_grid_proba, bpl/base.py:74-111, from which the outcome, n-goals
//       and sampling methods are reductions).  Without tau the grid of one draw is the outer
//       product of two Poisson pmf vectors, so the mean over draws is a [16 x S] x [S x 16]
//       contraction per 16 x 16 tile: ONE WAVE PER FIXTURE on the matrix pipe
//       (v_mfma_f32_16x16x4_f32, exact float32 fma chain, four draws per instruction).  Lane l
//       computes exactly the operands it has to supply -- A[x = l & 15][k = l >> 4] =
//       pmf(x; home rate of draw k), B[k][y = l & 15] = pmf(y; away rate of draw k): two v_exp
//       each, no LDS, no cross-lane traffic.  The float32 accumulators are folded into float64
//       every 64 draws.  tau only changes the four low-score cells: their (tau - 1) corrections
//       are accumulated in float64 on the lanes that hold pmf(0) / pmf(1).  The posterior is
//       kept in float32 (the reference's own dtype): 4 B x S x (2T + 2), L2 resident.
//   predict_score_proba  arbitrary (home, away, x, y) entries, one thread each, float64 loop
//       over the draws (scorelines beyond the grid, e.g. x > 63).
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

// ---- the grid kernel
struct GridArgs {
    int S, T;
    const float* attack;     // [S,T] float32
    const float* defence;    // [S,T]
    const float* home_adv;   // [S] (ha_stride = 0) or [S,T] (ha_stride = T)
    int ha_stride;
    const float* corr;       // [S]
    int M, G;                // fixtures, max_goals
    const uint16_t* h;       // [M]
    const uint16_t* a;       // [M]
    double* out;             // [M, G+1, G+1]
};
constexpr int GRID_MAX_GOALS = 63;
typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void predict_score_grid(GridArgs A) {
    __shared__ float lg[GRID_MAX_GOALS + 1];  // lgamma(k + 1)
    for (int k = threadIdx.x; k <= GRID_MAX_GOALS; k += blockDim.x) lg[k] = (float)lgamma((double)k + 1.0);
    __syncthreads();
    const int lane = threadIdx.x & 63, d = lane >> 4, i = lane & 15;
    const int f = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (f >= A.M) return;  // (wave uniform)
    const int h = A.h[f], a = A.a[f], G = A.G, S = A.S, T = A.T;
    const int G1 = G + 1, nt = (G + 16) / 16;
    double* out = A.out + (size_t)f * G1 * G1;
    const double inv_s = 1.0 / (double)S;
    for (int tx = 0; tx < nt; ++tx)
        for (int ty = 0; ty < nt; ++ty) {
            const int x = 16 * tx + i, y = 16 * ty + i;
            const float fx = (float)x, fy = (float)y;
            const float lgx = lg[min(x, GRID_MAX_GOALS)], lgy = lg[min(y, GRID_MAX_GOALS)];
            const bool low_tile = tx == 0 && ty == 0;
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
            double accd[4] = {0.0, 0.0, 0.0, 0.0};
            double c_a = 0.0, c_b = 0.0;  // lane i = 0: cells (0,0), (0,1); lane i = 1: (1,0), (1,1)
            int fold = 0;
            for (int s0 = 0; s0 < S; s0 += 4) {
                const int sdr = s0 + d;
                const bool valid = sdr < S;
                const int s = valid ? sdr : S - 1;
                const size_t r = (size_t)s * T;
                const float ha = A.ha_stride ? A.home_adv[r + h] : A.home_adv[s];
                const float eh = A.attack[r + h] - A.defence[r + a] + ha;
                const float ea = A.attack[r + a] - A.defence[r + h];
                const float lh = __expf(eh), la = __expf(ea);
                // exp(Poisson.log_prob(k)) = exp(k log(rate) - rate - lgamma(k + 1))
                const float pa = valid && x <= G ? __expf(fmaf(fx, eh, -lh) - lgx) : 0.f;
                const float pb = valid && y <= G ? __expf(fmaf(fy, ea, -la) - lgy) : 0.f;
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(pa, pb, acc, 0, 0, 0);
                if (low_tile && i < 2 && valid) {
                    // exp(log(clip(1 + rho c, 0))) - 1 for the lane's two cells (bpl/_util.py:58-91)
                    const float rho = A.corr[s];
                    if (i == 0) {  // pa = pmf_h(0), pb = pmf_a(0); pmf_a(1) = pb * la
                        c_a += (double)(pa * pb) * ((double)fmaxf(1.f - rho * lh * la, 0.f) - 1.0);
                        c_b += (double)(pa * pb * la) * ((double)fmaxf(1.f + rho * lh, 0.f) - 1.0);
                    } else {       // pa = pmf_h(1), pb = pmf_a(1); pmf_a(0) = exp(-la)
                        c_a += (double)(pa * __expf(-la)) * ((double)fmaxf(1.f + rho * la, 0.f) - 1.0);
                        c_b += (double)(pa * pb) * ((double)fmaxf(1.f - rho, 0.f) - 1.0);
                    }
                }
                if (++fold == 16) {  // 64 draws per float32 accumulation
#pragma unroll
                    for (int j = 0; j < 4; ++j) accd[j] += (double)acc[j];
                    acc = f32x4{0.f, 0.f, 0.f, 0.f};
                    fold = 0;
                }
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) accd[j] += (double)acc[j];
            if (low_tile) {  // the four corrections: sum over the four draw groups d
                c_a += __shfl_xor(c_a, 16);
                c_b += __shfl_xor(c_b, 16);
                c_a += __shfl_xor(c_a, 32);
                c_b += __shfl_xor(c_b, 32);
                // cell (x, y) lives on lane (d = x / 4, i = y), register j = x % 4: (0,0) and (1,0)
                // on lane 0 (j = 0, 1), (0,1) and (1,1) on lane 1
                const double c00 = __shfl(c_a, 0), c01 = __shfl(c_b, 0), c10 = __shfl(c_a, 1),
                             c11 = __shfl(c_b, 1);
                if (lane == 0) { accd[0] += c00; accd[1] += c10; }
                if (lane == 1) { accd[0] += c01; accd[1] += c11; }
            }
            // D[row = 4 (lane >> 4) + j][col = lane & 15]
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int xo = 16 * tx + 4 * d + j;
                if (xo <= G && y <= G) out[(size_t)xo * G1 + y] = accd[j] * inv_s;
            }
        }
}

}  // namespace dcp
