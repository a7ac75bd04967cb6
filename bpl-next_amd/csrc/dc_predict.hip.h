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

// ---- the grid kernel
// One WAVE per fixture, blocks of 64 posterior draws, two lane layouts with a wave-private LDS strip
// between them:
//   lane = draw:  the posterior's float32 copies are TEAM-major ([T][S]: a team's draws are
//     contiguous), so a block is one coalesced round of loads (the next block's are in flight while
//     this one is worked on).  Each lane works out its draw's two Poisson pmf vectors for the tile's
//     16 goal counts by the recurrence pmf(k + 1) = pmf(k) rate / (k + 1) -- both vectors at once, one
//     v_pk_mul_f32 for the two factors and one for the two updates per goal count, two v_exp_f32 per
//     draw in all -- and parks them in the strip, [goal][draw] (row stride 68 floats); the four tau
//     corrections of the low scorelines are plain per-draw products here as well.
//   lane = (goal count i = lane & 15, draw group k = lane >> 4):  v_mfma_f32_16x16x4_f32 takes
//     A[i][k] and B[k][i] from lane 16 k + i.  The order of the draws inside the sum is free, so group
//     k takes draws 16 k .. 16 k + 15 of the block: lane (i, k) reads 16 CONSECUTIVE floats of row i
//     of each strip -- four ds_read_b128 per operand per block (conflict free: the 16 lanes of a row
//     group cover all banks) instead of one ds_read_b32 per MFMA -- and the 16 MFMAs of the block run
//     from registers.
// What bounds it (tools/micro/grid_pipes.hip, profiles/r03/grid_pipes.txt): on a SIMD the VALU and the
// matrix pipe do NOT overlap -- a step costs MFMA (32 cycles for 16x16x4 f32) PLUS the VALU work
// that makes its operands -- so the operands must be cheap in VALU instructions: the recurrence costs 2
// packed multiplies per goal count for 64 draws, i.e. ~8 cycles per MFMA; an operand made in place by
// a DPP row broadcast + fma + v_exp_f32 (no LDS at all, tried first this round) costs 20 per operand,
// 72 per step: 317 us against round 2's 273.  The LDS pipe runs beside both.
// History (24 320 fixtures x 1000 draws x 16 x 16): gathers attack[s, h] ... per group of four draws
// inside the loop, pmf entries by one exp each, tau corrections in float64 on every step: 1010 us;
// coalesced blocks + prefetch: 624 us; pmf by recurrence through an LDS strip, one ds_read_b32 per
// operand per MFMA, recurrence constants from LDS: 273 us (round 2); this kernel: profiles/r03/kernels.md.
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
    float* out32;            // ... or, when not null, the same grids as float32 (the reference's own dtype: half the copy back)
    float rk[64];            // 1 / (k + 1): by value, i.e. in the kernarg segment -- wave-uniform scalar loads
    double rfact[16];        // 1 / k!, k < 16 (the first tile's cells are scaled at the end)
};
constexpr int GRID_MAX_GOALS = 63;
constexpr int GRID_WAVES = 4;
constexpr int GRID_ROW = 68;   // floats per goal count in a strip: 64 draws + 4 (16-byte aligned, rows 4 banks apart)
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <bool VENUE>
__global__ __launch_bounds__(64 * GRID_WAVES) void predict_score_grid(GridArgs A) {
    __shared__ __attribute__((aligned(16))) float strip[GRID_WAVES][2][16 * GRID_ROW];  // per wave: home, away [goal][draw]
    const float* rk = A.rk;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, d = lane >> 4, i = lane & 15;
    const int f = blockIdx.x * GRID_WAVES + wave;
    if (f >= A.M) return;  // (wave uniform)
    const int h = A.h[f], a = A.a[f], G = A.G, S = A.S;
    const int G1 = G + 1, nt = (G + 16) / 16;
    double* out = A.out + (size_t)f * G1 * G1;
    float* out32 = A.out32 ? A.out32 + (size_t)f * G1 * G1 : nullptr;   // (wave uniform)
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
    // a block's raw values for this lane's draw (clamped index: the loads are unconditional).  Kept AS
    // LOADED until the block is worked on: combining them here would make the wave wait for the
    // loads it has just issued, i.e. no prefetch at all.
    struct Raw { float ah, aa, dh, da, ha, rho, v0, v1, v2, v3, ch, ca; };
    // (uniform base + ONE 32-bit byte offset per lane: the loads take the scalar-base form and share
    // their address register -- six 64-bit address computations per block otherwise)
    auto at = [](const float* base, unsigned int boff) {
        return *reinterpret_cast<const float*>(reinterpret_cast<const char*>(base) + boff);
    };
    auto load_raw = [&](int s0) {
        const unsigned int b = (unsigned int)min(s0 + lane, S - 1) * 4u;
        Raw r{};
        r.ah = at(att_h, b); r.aa = at(att_a, b); r.dh = at(def_h, b); r.da = at(def_a, b);
        if constexpr (VENUE) {
            r.v0 = at(hat_h, b); r.v1 = at(adf_a, b); r.v2 = at(aat_a, b); r.v3 = at(hdf_h, b);
            if (cf_h) { r.ch = at(cf_h, b); r.ca = at(cf_a, b); }
        } else {
            r.ha = at(ha_p, b);
        }
        r.rho = at(A.corr, b);
        return r;
    };
    // log-rates of a draw
    auto log_rates = [&](const Raw& r, float* eh, float* ea) {
        if constexpr (VENUE) {
            // same association as the float64 restatement: ((att - def) + on hat) - on adf (+- dc)
            *eh = r.ah - r.da + on * r.v0 - on * r.v1;
            *ea = r.aa - r.dh + on * r.v2 - on * r.v3;
            if (cf_h) {
                const float dc = r.ch - r.ca;
                *eh += dc;
                *ea -= dc;
            }
        } else {
            *eh = r.ah - r.da + r.ha;
            *ea = r.aa - r.dh;
        }
    };
    float* stH = strip[wave][0];
    float* stA = strip[wave][1];
    for (int tx = 0; tx < nt; ++tx)
        for (int ty = 0; ty < nt; ++ty) {
            const int x0 = 16 * tx, y0 = 16 * ty;
            const bool low_tile = tx == 0 && ty == 0;
            // (wave-uniform constants of the recurrence: scalar loads, hoisted out of the draw loop)
            f32x2 inv[16];
#pragma unroll
            for (int k = 0; k < 16; ++k) inv[k] = f32x2{rk[min(x0 + k, GRID_MAX_GOALS)], rk[min(y0 + k, GRID_MAX_GOALS)]};
            double accd[4] = {0.0, 0.0, 0.0, 0.0};
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};   // float32 over FOLD blocks of 64 draws, then folded into accd
            constexpr int FOLD = 4;
            int in_acc = 0;
            // this lane's draws: tau corrections (float32 over the <= S / 64 blocks of a lane: each
            // term is at most a cell's own size, the rounding of the sum 1e-7 of it)
            float c00 = 0.f, c01 = 0.f, c10 = 0.f, c11 = 0.f;
            Raw nxt = load_raw(0);
            for (int s0 = 0; s0 < S; s0 += 64) {
                const Raw cur = nxt;
                if (s0 + 64 < S) {
                    nxt = load_raw(s0 + 64);     // in flight while this block is worked on
                    // (opaque: or the compiler sinks the loads to their first use, the next iteration)
                    asm volatile("" : "+v"(nxt.ah), "+v"(nxt.aa), "+v"(nxt.dh), "+v"(nxt.da), "+v"(nxt.rho));
                }
                {   // lane = draw: both pmf vectors of the tile into the strip
                    const bool valid = s0 + lane < S;
                    float eh, ea;
                    log_rates(cur, &eh, &ea);
                    const float lh = __builtin_amdgcn_exp2f(eh * LOG2E), la = __builtin_amdgcn_exp2f(ea * LOG2E);
                    const f32x2 rate = {lh, la};
                    // pmf(0) = exp(-rate); a draw beyond S adds nothing
                    f32x2 p = {valid ? __builtin_amdgcn_exp2f(-lh * LOG2E) : 0.f, valid ? __builtin_amdgcn_exp2f(-la * LOG2E) : 0.f};
                    if (low_tile) {
                        // exp(log(clip(1 + rho c, 0))) - 1 for the four low scorelines (bpl/_util.py:58-91)
                        const float ph = p.x, pa = p.y, rho = cur.rho, p1h = ph * lh, p1a = pa * la;
                        c00 = fmaf(ph * pa, fmaxf(1.f - rho * lh * la, 0.f) - 1.f, c00);
                        c01 = fmaf(ph * p1a, fmaxf(1.f + rho * lh, 0.f) - 1.f, c01);
                        c10 = fmaf(p1h * pa, fmaxf(1.f + rho * la, 0.f) - 1.f, c10);
                        c11 = fmaf(p1h * p1a, fmaxf(1.f - rho, 0.f) - 1.f, c11);
                    } else {
                        // a later tile starts at pmf(x0), pmf(y0): the same recurrence from 0
                        for (int k = 0; k < max(x0, y0); ++k) {
                            const f32x2 step = rate * f32x2{rk[k], rk[k]};
                            if (k < x0) p.x *= step.x;
                            if (k < y0) p.y *= step.y;
                        }
                    }
                    if (low_tile) {
                        // the first tile carries e^-rate rate^k WITHOUT the 1 / k! (one v_pk_mul_f32 per goal
                        // count for both vectors instead of two); 1 / (x! y!) multiplies the finished cell.
                        // e^-r r^k <= 1.3e11 for k <= 15 and any r: no overflow, products included.
#pragma unroll
                        for (int k = 0; k < 16; ++k) {
                            stH[k * GRID_ROW + lane] = p.x;
                            stA[k * GRID_ROW + lane] = p.y;
                            p *= rate;
                        }
                    } else {
#pragma unroll
                        for (int k = 0; k < 16; ++k) {
                            stH[k * GRID_ROW + lane] = p.x;
                            stA[k * GRID_ROW + lane] = p.y;
                            p *= rate * inv[k];    // two v_pk_mul_f32 for both vectors
                        }
                    }
                }
                // lane = (goal count, draw group): this lane's 16 draws of row i, then 16 rank-4 updates
                // (the strip is wave-private and LDS operations of a wave complete in order: no barrier)
                f32x4 ah[4], aa[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    ah[q] = *reinterpret_cast<const f32x4*>(stH + i * GRID_ROW + 16 * d + 4 * q);
                    aa[q] = *reinterpret_cast<const f32x4*>(stA + i * GRID_ROW + 16 * d + 4 * q);
                }
#pragma unroll
                for (int g = 0; g < 16; ++g)
                    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(ah[g >> 2][g & 3], aa[g >> 2][g & 3], acc, 0, 0, 0);
                if (++in_acc == FOLD || s0 + 64 >= S) {   // (reading the accumulator waits for the matrix pipe)
#pragma unroll
                    for (int j = 0; j < 4; ++j) accd[j] += (double)acc[j];
                    acc = f32x4{0.f, 0.f, 0.f, 0.f};
                    in_acc = 0;
                }
            }
            if (low_tile) {   // the factorials the first tile left out: cell (x, y) = (4 d + j, i)
#pragma unroll
                for (int j = 0; j < 4; ++j) accd[j] *= A.rfact[4 * d + j] * A.rfact[i];
            }
            if (low_tile) {
                double c4[4] = {(double)c00, (double)c01, (double)c10, (double)c11};
                dc::wave_sum4_f64(c4);
                // cell (x, y) lives on lane (d = x / 4, i = y), register j = x % 4: (0,0) and (1,0)
                // on lane 0 (j = 0, 1), (0,1) and (1,1) on lane 1
                if (lane == 0) { accd[0] += c4[0]; accd[1] += c4[2]; }
                if (lane == 1) { accd[0] += c4[1]; accd[1] += c4[3]; }
            }
            // D[row = 4 (lane >> 4) + j][col = lane & 15]
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int xo = x0 + 4 * d + j, y = y0 + i;
                if (xo <= G && y <= G) {
                    if (out32) out32[(size_t)xo * G1 + y] = (float)(accd[j] * inv_s);
                    else out[(size_t)xo * G1 + y] = accd[j] * inv_s;
                }
            }
        }
}

}  // namespace dcp
