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
// One WAVE per fixture, blocks of 64 posterior draws.
//   lane = draw:  the posterior's float32 copies are TEAM-major ([T][S]: a team's draws are
//     contiguous), so a block is one coalesced round of loads (the next block's are in flight while
//     this one is worked on).  Each lane works out its draw's two Poisson pmf vectors for the tile's
//     16 goal counts -- one exp, then pmf(k+1) = pmf(k) rate / (k+1) -- and parks them in a
//     wave-private LDS strip [draw][goal]; the four tau corrections of the low scorelines are plain
//     per-draw products here as well.
//   lane = (goal, draw % 4):  per group of four draws the two pmf vectors are the A and B operands of
//     v_mfma_f32_16x16x4_f32 -- the grid is a sum over draws of OUTER PRODUCTS, a genuine rank-4
//     update per instruction -- two LDS reads and one MFMA per step, float32 for 64 draws, then
//     folded into float64.
// History (24 320 fixtures x 1000 draws x 16 x 16, profiles/r02/): gathers attack[s, h] ... per group
// of four draws inside the loop, pmf entries by one exp each, tau corrections in float64 on every
// step: 1010 us; coalesced blocks + prefetch: 624 us; pmf by recurrence in the draw layout and the
// corrections once per draw: see profiles/r02/kernels.md.
struct GridArgs {
    int S, T;
    const float* attack;     // [T,S] float32, team-major
    const float* defence;    // [T,S]
    const float* home_adv;   // [S] (ha_stride = 0) or [T,S] (ha_stride != 0)
    int ha_stride;
    const float* corr;       // [S]
    int M, G;                // fixtures, max_goals
    const uint16_t* h;       // [M]
    const uint16_t* a;       // [M]
    double* out;             // [M, G+1, G+1]
};
constexpr int GRID_MAX_GOALS = 63;
constexpr int GRID_WAVES = 4;
constexpr int GRID_ROW = 17;   // floats per draw in the strip (16 goal counts + 1: conflict-free both ways)
typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(64 * GRID_WAVES) void predict_score_grid(GridArgs A) {
    __shared__ float lg[GRID_MAX_GOALS + 2];          // lgamma(k + 1)
    __shared__ float rk[GRID_MAX_GOALS + 2];          // 1 / (k + 1)
    __shared__ float strip[GRID_WAVES][2][64 * GRID_ROW];  // per wave: pmf_home, pmf_away of a block [draw][goal]
    for (int k = threadIdx.x; k <= GRID_MAX_GOALS + 1; k += blockDim.x) {
        lg[k] = (float)lgamma((double)k + 1.0);
        rk[k] = (float)(1.0 / ((double)k + 1.0));
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, d = lane >> 4, i = lane & 15;
    const int f = blockIdx.x * GRID_WAVES + wave;
    if (f >= A.M) return;  // (wave uniform)
    const int h = A.h[f], a = A.a[f], G = A.G, S = A.S;
    const int G1 = G + 1, nt = (G + 16) / 16;
    double* out = A.out + (size_t)f * G1 * G1;
    const double inv_s = 1.0 / (double)S;
    const float* att_h = A.attack + (size_t)h * S;
    const float* att_a = A.attack + (size_t)a * S;
    const float* def_h = A.defence + (size_t)h * S;
    const float* def_a = A.defence + (size_t)a * S;
    const float* ha_p = A.ha_stride ? A.home_adv + (size_t)h * S : A.home_adv;
    float* pmH = strip[wave][0];
    float* pmA = strip[wave][1];
    // a block's raw values for this lane's draw (clamped index: the loads are unconditional)
    struct Raw { float ah, aa, dh, da, ha, rho; };
    auto load_raw = [&](int s0) {
        const int s = min(s0 + lane, S - 1);
        Raw r;
        r.ah = att_h[s]; r.aa = att_a[s]; r.dh = def_h[s]; r.da = def_a[s]; r.ha = ha_p[s]; r.rho = A.corr[s];
        return r;
    };
    for (int tx = 0; tx < nt; ++tx)
        for (int ty = 0; ty < nt; ++ty) {
            const int x0 = 16 * tx, y0 = 16 * ty;
            const bool low_tile = tx == 0 && ty == 0;
            double accd[4] = {0.0, 0.0, 0.0, 0.0};
            double c00 = 0.0, c01 = 0.0, c10 = 0.0, c11 = 0.0;  // this lane's draws: tau corrections
            Raw nxt = load_raw(0);
            for (int s0 = 0; s0 < S; s0 += 64) {
                const Raw cur = nxt;
                if (s0 + 64 < S) nxt = load_raw(s0 + 64);     // in flight while this block is worked on
                {   // lane = draw: the tile's 16 entries of both pmf vectors
                    const bool valid = s0 + lane < S;
                    const float eh = cur.ah - cur.da + cur.ha, ea = cur.aa - cur.dh;
                    const float lh = __expf(eh), la = __expf(ea);
                    // exp(Poisson.log_prob(k)) = exp(k log(rate) - rate - lgamma(k + 1)) at the tile's first
                    // count, then pmf(k + 1) = pmf(k) rate / (k + 1)
                    float ph = valid ? __expf(fmaf((float)x0, eh, -lh) - lg[x0]) : 0.f;
                    float pa = valid ? __expf(fmaf((float)y0, ea, -la) - lg[y0]) : 0.f;
                    if (low_tile) {
                        // exp(log(clip(1 + rho c, 0))) - 1 for the four low scorelines (bpl/_util.py:58-91)
                        const float rho = cur.rho, p1h = ph * lh, p1a = pa * la;
                        c00 += (double)((ph * pa) * (fmaxf(1.f - rho * lh * la, 0.f) - 1.f));
                        c01 += (double)((ph * p1a) * (fmaxf(1.f + rho * lh, 0.f) - 1.f));
                        c10 += (double)((p1h * pa) * (fmaxf(1.f + rho * la, 0.f) - 1.f));
                        c11 += (double)((p1h * p1a) * (fmaxf(1.f - rho, 0.f) - 1.f));
                    }
                    float* rowH = pmH + lane * GRID_ROW;
                    float* rowA = pmA + lane * GRID_ROW;
#pragma unroll
                    for (int k = 0; k < 16; ++k) {
                        rowH[k] = x0 + k <= G ? ph : 0.f;
                        rowA[k] = y0 + k <= G ? pa : 0.f;
                        ph *= lh * rk[x0 + k];
                        pa *= la * rk[y0 + k];
                    }
                }
                f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int g = 0; g < 16; ++g) {   // lane = (goal count i, draw 4 g + d)
                    const int sl = 4 * g + d;
                    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(pmH[sl * GRID_ROW + i], pmA[sl * GRID_ROW + i], acc, 0, 0, 0);
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) accd[j] += (double)acc[j];   // 64 draws per float32 accumulation
            }
            if (low_tile) {
                double c4[4] = {c00, c01, c10, c11};
                dc::wave_sum4_f64(c4);
                // cell (x, y) lives on lane (d = x / 4, i = y), register j = x % 4: (0,0) and (1,0)
                // on lane 0 (j = 0, 1), (0,1) and (1,1) on lane 1
                if (lane == 0) { accd[0] += c4[0]; accd[1] += c4[2]; }
                if (lane == 1) { accd[0] += c4[1]; accd[1] += c4[3]; }
            }
            // D[row = 4 (lane >> 4) + j][col = lane & 15]
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int xo = 16 * tx + 4 * d + j, y = y0 + i;
                if (xo <= G && y <= G) out[(size_t)xo * G1 + y] = accd[j] * inv_s;
            }
        }
}

}  // namespace dcp
