// dc_predict.hip.h -- the predict path on the device:
// `predict_score_proba` of bpl/dixon_coles.py:139-163 / bpl/extended_dixon_coles.py:360-399 /
// bpl/neutral_dixon_coles.py:425-488 / bpl/neutral_dixon_coles_WC.py (same, + confederations):
//     mean over posterior draws s of  exp(corr_term_s) * Poisson(x; lh_s) * Poisson(y; la_s)
// with the rates of `_calculate_expected_goals` and the tau term of bpl/_util.py:35-93 evaluated
// per draw with that draw's corr_coef (tol = 0).  Two rate forms (template parameter VENUE):
//   VENUE = 0  (bpl/dixon_coles.py:126-137, bpl/extended_dixon_coles.py:335-358)
//       log lh = attack[h] - defence[a] + home_advantage(scalar or [h]);  log la = attack[a] - defence[h]
//   VENUE = 1  (bpl/neutral_dixon_coles.py:399-423, bpl/neutral_dixon_coles_WC.py: + confederations,
//               bpl/dynamic_dixon_coles.py:336-361 with the tables of one gameweek)
//       on = 1 - neutral_venue,  dc = confederation_strength[home_conf] - confederation_strength[away_conf]
//       log lh = attack[h] - defence[a] + on (home_attack[h] - away_defence[a]) + dc
//       log la = attack[a] - defence[h] + on (away_attack[a] - home_defence[h]) - dc
//
// Two kernels:
//   predict_score_grid   THE predict primitive: the whole (G+1) x (G+1) scoreline grid of a
//       fixture (predict_score_grid_proba, bpl/base.py:74-111, from which the outcome, n-goals
//       and sampling methods are reductions).  Without tau the grid of one draw is the outer
//       product of two Poisson pmf vectors, so the mean over draws is a [16 x S] x [S x 16]
//       contraction per 16 x 16 tile: ONE WAVE PER FIXTURE on the matrix pipe
//       (v_mfma_f32_16x16x4_f32, exact float32 fma chain, four draws per instruction).
//   predict_score_proba  arbitrary (home, away, x, y) entries, one thread each, float64 loop
//       over the draws (scorelines beyond the grid, e.g. x > 63).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace dcp {

struct PredictArgs {
    int S, T, C;
    const double* attack;    // [S,T]
    const double* defence;   // [S,T]
    const double* home_adv;  // VENUE = 0: [S] (ha_stride = 0) or [S,T] (ha_stride = T)
    int ha_stride;
    const double* home_attack;   // VENUE = 1: [S,T] each
    const double* away_attack;
    const double* home_defence;
    const double* away_defence;
    const double* conf;      // [S,C] or null
    const double* corr;      // [S]
    long long M;
    const uint16_t* h;
    const uint16_t* a;
    const uint16_t* x;       // goals as given (may exceed 255 in a query)
    const uint16_t* y;
    const uint8_t* neutral;  // VENUE = 1: [M]
    const uint16_t* hc;      // VENUE = 1 with confederations: [M] each
    const uint16_t* ac;
    double* out;             // [M]
};

template <bool VENUE>
__global__ __launch_bounds__(256) void predict_score_proba(PredictArgs A) {
    const long long m = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (m >= A.M) return;
    const int h = A.h[m], a = A.a[m], x = A.x[m], y = A.y[m];
    const double lgx = lgamma((double)x + 1.0), lgy = lgamma((double)y + 1.0);
    const bool low = x <= 1 && y <= 1;
    double on = 0.0;
    int hc = 0, ac = 0;
    if constexpr (VENUE) {
        on = A.neutral[m] ? 0.0 : 1.0;
        if (A.conf) { hc = A.hc[m]; ac = A.ac[m]; }
    }
    double acc = 0.0;
    for (int s = 0; s < A.S; ++s) {
        const size_t r = (size_t)s * A.T;
        double eh = A.attack[r + h] - A.defence[r + a], ea = A.attack[r + a] - A.defence[r + h];
        if constexpr (VENUE) {
            eh += on * A.home_attack[r + h] - on * A.away_defence[r + a];
            ea += on * A.away_attack[r + a] - on * A.home_defence[r + h];
            if (A.conf) {
                const double dc = A.conf[(size_t)s * A.C + hc] - A.conf[(size_t)s * A.C + ac];
                eh += dc;
                ea -= dc;
            }
        } else {
            eh += A.ha_stride ? A.home_adv[r + h] : A.home_adv[s];
        }
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

// ---- the grid kernel (round 3: no LDS traffic in the loop)
// One WAVE per fixture, blocks of 64 posterior draws, two lane layouts:
//   lane = draw:  the posterior's float32 copies are TEAM-major ([T][S]: a team's draws are
//     contiguous), so a block is one coalesced round of loads (the next block's are in flight while
//     this one is worked on).  Each lane reduces its draw to FOUR numbers, in base-2 units:
//     E_h = log2 lh, Q_h = -lh log2(e) (and the away pair) -- pmf(k; l) = 2^(k E + Q) / k!  -- and
//     books the four tau corrections of the low scorelines as plain per-draw products.
//   lane = (goal count i = lane & 15, draw group k = lane >> 4):  v_mfma_f32_16x16x4_f32 takes
//     A[i][k] and B[k][i] from lane 16 k + i.  The order of the draws inside the sum is free, so group
//     k takes the 16 draws held by ITS OWN ROW of 16 lanes: for step g the operand of lane (i, k) is
//     2^(i E + Q) of the draw on lane 16 k + g -- a DPP row broadcast (row_newbcast:g, folded into the
//     VALU instruction that consumes it), one fma and one v_exp_f32 per operand.  No LDS round trip,
//     no bank conflicts, no barrier: round 2's kernel wrote both pmf vectors of every draw to a
//     wave-private LDS strip and read them back, 64 LDS operations per 16 MFMAs, and the LDS pipe
//     (shared by the CU's four SIMDs) was as busy as the matrix pipe.
//   The 1 / (x! y!) of the pmfs is applied ONCE, in float64, when the tile is written (with an
//     exact power-of-two offset c_k = rint(log2 k!) inside the exponent so that 2^(k E + Q - c_k)
//     stays in float32 range up to max_goals = 63): a float32 log2(k!) inside the exponent would be a
//     rounding error common to all draws (1.3e-6 relative at k = 15).
//   float32 MFMA accumulation over the 64 draws of a block, float64 across blocks.
// History (24 320 fixtures x 1000 draws x 16 x 16): gathers attack[s, h] ... per group of four draws
// inside the loop, pmf entries by one exp each, tau corrections in float64 on every step: 1010 us;
// coalesced blocks + prefetch: 624 us; pmf by recurrence in the draw layout through an LDS strip:
// 273 us (round 2); this kernel: see profiles/r03/kernels.md.
struct GridArgs {
    int S, T;
    const float* attack;     // [T,S] float32, team-major
    const float* defence;    // [T,S]
    const float* home_adv;   // VENUE = 0: [S] (ha_stride = 0) or [T,S] (ha_stride != 0)
    int ha_stride;
    const float* home_attack;   // VENUE = 1: [T,S] each
    const float* away_attack;
    const float* home_defence;
    const float* away_defence;
    const float* conf;       // [C,S] or null
    const float* corr;       // [S]
    int M, G;                // fixtures, max_goals
    const uint16_t* h;       // [M]
    const uint16_t* a;       // [M]
    const uint8_t* neutral;  // VENUE = 1: [M]
    const uint16_t* hc;      // with confederations: [M] each
    const uint16_t* ac;
    double* out;             // [M, G+1, G+1]
    const float* cexp;       // [64] c_k = rint(log2 k!)            (host-built once per context)
    const double* scale;     // [64] 2^c_k / k!
};
constexpr int GRID_MAX_GOALS = 63;
constexpr int GRID_WAVES = 4;
typedef float f32x4 __attribute__((ext_vector_type(4)));

// t = Q[lane G of the caller's row of 16] - c + E[lane G of the row] * f: the base-2 exponent of one
// pmf entry, two VALU instructions with the row broadcast folded in (the compiler's DPP combiner
// folds it into the subtraction only and spends a v_mov_b32_dpp on the fma)
template <int G> __device__ __forceinline__ float pmf_exponent(float E, float Q, float f, float c) {
    float t;
    asm("v_sub_f32_dpp %0, %1, %2 row_newbcast:%5 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f32_dpp %0, %3, %4 row_newbcast:%5 row_mask:0xf bank_mask:0xf"
        : "=&v"(t) : "v"(Q), "v"(c), "v"(E), "v"(f), "n"(G));
    return t;
}

template <bool VENUE>
__global__ __launch_bounds__(64 * GRID_WAVES) void predict_score_grid(GridArgs A) {
    const float* __restrict__ cexp = A.cexp;
    const double* __restrict__ scale = A.scale;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, d = lane >> 4, i = lane & 15;
    const int f = blockIdx.x * GRID_WAVES + wave;
    if (f >= A.M) return;  // (wave uniform)
    const int h = A.h[f], a = A.a[f], G = A.G, S = A.S;
    const int G1 = G + 1, nt = (G + 16) / 16;
    double* out = A.out + (size_t)f * G1 * G1;
    const double inv_s = 1.0 / (double)S;
    constexpr float LOG2E = 1.44269504088896341f;
    const float* att_h = A.attack + (size_t)h * S;
    const float* att_a = A.attack + (size_t)a * S;
    const float* def_h = A.defence + (size_t)h * S;
    const float* def_a = A.defence + (size_t)a * S;
    const float* ha_p = nullptr;
    const float *hat_h = nullptr, *aat_a = nullptr, *hdf_h = nullptr, *adf_a = nullptr, *cf_h = nullptr, *cf_a = nullptr;
    float on = 0.f;
    if constexpr (VENUE) {
        hat_h = A.home_attack + (size_t)h * S;
        aat_a = A.away_attack + (size_t)a * S;
        hdf_h = A.home_defence + (size_t)h * S;
        adf_a = A.away_defence + (size_t)a * S;
        on = A.neutral[f] ? 0.f : 1.f;
        if (A.conf) {
            cf_h = A.conf + (size_t)A.hc[f] * S;
            cf_a = A.conf + (size_t)A.ac[f] * S;
        }
    } else {
        ha_p = A.ha_stride ? A.home_adv + (size_t)h * S : A.home_adv;
    }
    // a block's log-rates for this lane's draw (clamped index: the loads are unconditional)
    struct Raw { float eh, ea, rho; };
    auto load_raw = [&](int s0) {
        const int s = min(s0 + lane, S - 1);
        Raw r;
        if constexpr (VENUE) {
            // same association as the float64 restatement: ((att - def) + on hat) - on adf (+- dc)
            const float ah = att_h[s], da = def_a[s], aa = att_a[s], dh = def_h[s];
            const float v0 = hat_h[s], v1 = adf_a[s], v2 = aat_a[s], v3 = hdf_h[s];
            r.eh = ah - da + on * v0 - on * v1;
            r.ea = aa - dh + on * v2 - on * v3;
            if (cf_h) {
                const float dc = cf_h[s] - cf_a[s];
                r.eh += dc;
                r.ea -= dc;
            }
        } else {
            r.eh = att_h[s] - def_a[s] + ha_p[s];
            r.ea = att_a[s] - def_h[s];
        }
        r.rho = A.corr[s];
        return r;
    };
    for (int tx = 0; tx < nt; ++tx)
        for (int ty = 0; ty < nt; ++ty) {
            const int x0 = 16 * tx, y0 = 16 * ty;
            const bool low_tile = tx == 0 && ty == 0;
            const float fx = (float)(x0 + i), fy = (float)(y0 + i);
            const float cx = cexp[min(x0 + i, GRID_MAX_GOALS)], cy = cexp[min(y0 + i, GRID_MAX_GOALS)];
            double accd[4] = {0.0, 0.0, 0.0, 0.0};
            // this lane's draws: tau corrections (float32 over the <= S / 64 blocks of a lane: each
            // term is at most a cell's own size, the rounding of the sum 1e-7 of it)
            float c00 = 0.f, c01 = 0.f, c10 = 0.f, c11 = 0.f;
            Raw nxt = load_raw(0);
            for (int s0 = 0; s0 < S; s0 += 64) {
                const Raw cur = nxt;
                if (s0 + 64 < S) nxt = load_raw(s0 + 64);     // in flight while this block is worked on
                // lane = draw
                const bool valid = s0 + lane < S;
                float Eh = cur.eh * LOG2E, Ea = cur.ea * LOG2E;
                const float lh = __builtin_amdgcn_exp2f(Eh), la = __builtin_amdgcn_exp2f(Ea);
                float Qh = valid ? -lh * LOG2E : -__builtin_inff();   // 2^-inf = 0: a draw beyond S adds nothing
                float Qa = -la * LOG2E;
                if (low_tile) {
                    // exp(log(clip(1 + rho c, 0))) - 1 for the four low scorelines (bpl/_util.py:58-91)
                    const float ph = __builtin_amdgcn_exp2f(Qh), pa = __builtin_amdgcn_exp2f(Qa);
                    const float rho = cur.rho, p1h = ph * lh, p1a = pa * la;
                    c00 = fmaf(ph * pa, fmaxf(1.f - rho * lh * la, 0.f) - 1.f, c00);
                    c01 = fmaf(ph * p1a, fmaxf(1.f + rho * lh, 0.f) - 1.f, c01);
                    c10 = fmaf(p1h * pa, fmaxf(1.f + rho * la, 0.f) - 1.f, c10);
                    c11 = fmaf(p1h * p1a, fmaxf(1.f - rho, 0.f) - 1.f, c11);
                }
                // (the asm below reads these four through DPP: two wait states after the VALU that
                // wrote them, which the compiler cannot see inside an asm block)
                asm volatile("s_nop 1" : "+v"(Eh), "+v"(Qh), "+v"(Ea), "+v"(Qa));
                // lane = (goal count, draw group): 16 rank-4 updates
                f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#define DCP_STEP(g)                                                                              \
    {                                                                                            \
        const float ta = pmf_exponent<g>(Eh, Qh, fx, cx);                                        \
        const float tb = pmf_exponent<g>(Ea, Qa, fy, cy);                                        \
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__builtin_amdgcn_exp2f(ta), __builtin_amdgcn_exp2f(tb), acc, 0, 0, 0); \
    }
                DCP_STEP(0) DCP_STEP(1) DCP_STEP(2) DCP_STEP(3) DCP_STEP(4) DCP_STEP(5) DCP_STEP(6) DCP_STEP(7)
                DCP_STEP(8) DCP_STEP(9) DCP_STEP(10) DCP_STEP(11) DCP_STEP(12) DCP_STEP(13) DCP_STEP(14) DCP_STEP(15)
#undef DCP_STEP
#pragma unroll
                for (int j = 0; j < 4; ++j) accd[j] += (double)acc[j];   // 64 draws per float32 accumulation
            }
            // D[row = 4 (lane >> 4) + j][col = lane & 15], scaled by 2^(c_x + c_y) / (x! y!)
            const double sy = scale[min(y0 + i, GRID_MAX_GOALS)];
#pragma unroll
            for (int j = 0; j < 4; ++j) accd[j] *= scale[min(x0 + 4 * d + j, GRID_MAX_GOALS)] * sy;
            if (low_tile) {
                double c4[4] = {(double)c00, (double)c01, (double)c10, (double)c11};
                dc::wave_sum4_f64(c4);
                // cell (x, y) lives on lane (d = x / 4, i = y), register j = x % 4: (0,0) and (1,0)
                // on lane 0 (j = 0, 1), (0,1) and (1,1) on lane 1  (0! = 1! = 1: no scale)
                if (lane == 0) { accd[0] += c4[0]; accd[1] += c4[2]; }
                if (lane == 1) { accd[0] += c4[1]; accd[1] += c4[3]; }
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int xo = x0 + 4 * d + j, y = y0 + i;
                if (xo <= G && y <= G) out[(size_t)xo * G1 + y] = accd[j] * inv_s;
            }
        }
}

}  // namespace dcp
