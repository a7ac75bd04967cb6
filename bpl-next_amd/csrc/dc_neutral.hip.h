// dc_neutral.hip.h -- gfx950 kernels for the neutral-venue Dixon-Coles model
// (bpl/neutral_dixon_coles.py:102-283): the extended model's
// rho-correlated attack/defence plus four per-team non-centred offsets (home_attack,
// away_attack, home_defence, away_defence) that are switched off at neutral venues, an
// always-weighted likelihood, no rate clip.
//
// First correct path: float64, the fixture passes are the dynamic model's with ONE
// "gameweek" (dc_dynamic.hip.h: dyn_pass1 = rates + maxima, dyn_pass2 = value + adjoint
// into LDS-private per-team accumulators, now with per-fixture weights); only the z-side
// is model specific:
//   neu_cells     per team: constrained sites -> the six-entry cell record of dyn_pass*
//   neu_epilogue  bounds adjoint, priors + Jacobians, chain rule to z (one workgroup; the
//                 z side is O(T))
// Roofline: HBM-bound stream of 11 B per fixture (u16,u16,u8,u8,u8 neutral, f32 weight).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "dc_dynamic.hip.h"
#include "dc_kernels.hip.h"  // DPP wave reductions

namespace dcn {

using dc::HALF_LOG_2PI;
using dc::LN2;

// flat latent layout: sorted site names (numpyro), D = 6T + 2K + C + 13 (C confederations,
// World-Cup variant bpl/neutral_dixon_coles_WC.py; C = 0 for the plain neutral model)
struct NeuLayout {
    int T, K, C, D;
    int o_conf;
    int o_bA, o_aat, o_adf, o_corr, o_bD, o_hat, o_hdf, o_maa, o_mad, o_md, o_mha, o_mhd, o_sat,
        o_sdt, o_s_att, o_s_aa, o_s_ad, o_s_def, o_s_ha, o_s_hd, o_u;
};
inline NeuLayout make_neu_layout(int T, int K, int C = 0) {
    NeuLayout L{};
    L.T = T; L.K = K; L.C = C;
    int o = 0;
    L.o_bA = o; o += K;        // attack_coefficients
    L.o_aat = o; o += T;       // away_attack_decentered
    L.o_adf = o; o += T;       // away_defence_decentered
    L.o_conf = o; o += C;      // confederation_strength_decentered
    L.o_corr = o; o += 1;      // corr_coef_raw
    L.o_bD = o; o += K;        // defence_coefficients
    L.o_hat = o; o += T;       // home_attack_decentered
    L.o_hdf = o; o += T;       // home_defence_decentered
    L.o_maa = o; o += 1;       // mean_away_attack
    L.o_mad = o; o += 1;       // mean_away_defence
    L.o_md = o; o += 1;        // mean_defence
    L.o_mha = o; o += 1;       // mean_home_attack
    L.o_mhd = o; o += 1;       // mean_home_defence
    L.o_sat = o; o += T;       // standardised_attack
    L.o_sdt = o; o += T;       // standardised_defence
    L.o_s_att = o; o += 1;     // std_attack
    L.o_s_aa = o; o += 1;      // std_away_attack
    L.o_s_ad = o; o += 1;      // std_away_defence
    L.o_s_def = o; o += 1;     // std_defence
    L.o_s_ha = o; o += 1;      // std_home_attack
    L.o_s_hd = o; o += 1;      // std_home_defence
    L.o_u = o; o += 1;         // u
    L.D = o;
    return L;
}

struct NeuArgs {
    dcd::DynArgs F;   // fixtures, cells, scratch (acc | sc), z / potential / grad / aux
    NeuLayout L;
};

// ---- per team: constrained sites -> cell record (dcd::P_*)
__global__ __launch_bounds__(256) void neu_cells(NeuArgs A) {
    const NeuLayout& L = A.L;
    const double* z = A.F.z;
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    // this launch also clears the evaluation's scratch (acc | sc | cacc)
    for (size_t i = t; i < A.F.scratch_n; i += (size_t)gridDim.x * blockDim.x) A.F.acc[i] = 0.0;
    if (t >= L.T) return;
    double att = 0.0, def = z[L.o_md];
    for (int k = 0; k < L.K; ++k) {
        const double xv = A.F.xs[(size_t)t * L.K + k];
        att += xv * z[L.o_bA + k];
        def += xv * z[L.o_bD + k];
    }
    att += z[L.o_sat + t] * exp(z[L.o_s_att]);
    def += z[L.o_sdt + t] * exp(z[L.o_s_def]);
    const double hat = z[L.o_mha] + exp(z[L.o_s_ha]) * z[L.o_hat + t];
    const double aat = z[L.o_maa] + exp(z[L.o_s_aa]) * z[L.o_aat + t];
    const double hdf = z[L.o_mhd] + exp(z[L.o_s_hd]) * z[L.o_hdf + t];
    const double adf = z[L.o_mad] + exp(z[L.o_s_ad]) * z[L.o_adf + t];
    double* P = A.F.cells + (size_t)t * dcd::P_N;
    P[dcd::P_AH] = att + hat;
    P[dcd::P_AA] = att + aat;
    P[dcd::P_BH] = def + hdf;
    P[dcd::P_BA] = def + adf;
    P[dcd::P_ATT] = att;
    P[dcd::P_DEF] = def;
}

// ---- epilogue: one workgroup
constexpr int NEU_EPI = 256;
// sums: 0 dL/d rho_p | 1 sum sa G_att | 2 sum sd G_def | 3 sum G_def | 4..7 sum G_x
// (hat, aat, hdf, adf) | 8..11 sum dec_x G_x | 12 log-density of the team sites | 13.. cov
constexpr int NEU_SUMS = 13;

__global__ __launch_bounds__(NEU_EPI) void neu_epilogue(NeuArgs A) {
    extern __shared__ double sums[];  // [NEU_SUMS + 2K]
    const NeuLayout& L = A.L;
    const int T = L.T, K = L.K;
    const int tid = threadIdx.x;
    const double* z = A.F.z;
    double* grad = A.F.grad;
    // this thread's first team: requested before anything else, so the global round trips
    // overlap with the scalar sites and the coupling fetch below
    double pG[dcd::A_N], pz[6];
    {
        const int t0 = tid < T ? tid : 0;
        const double* Ac = A.F.acc + (size_t)t0 * dcd::A_N;
#pragma unroll
        for (int j = 0; j < dcd::A_N; ++j) pG[j] = Ac[j];
        pz[0] = z[L.o_sat + t0]; pz[1] = z[L.o_sdt + t0]; pz[2] = z[L.o_hat + t0];
        pz[3] = z[L.o_aat + t0]; pz[4] = z[L.o_hdf + t0]; pz[5] = z[L.o_adf + t0];
    }
    for (int i = tid; i < NEU_SUMS + 2 * K; i += NEU_EPI) sums[i] = 0.0;
    // scalar sites first, in parallel lanes of one wave (a float64 libm call costs ~1 us of
    // dependent instructions: they must not run one after another on one lane):
    //   lanes 0..5 exp(std sites) | lanes 6, 7 sigmoid sites u, corr_coef_raw
    __shared__ double scal[6 + 2 * 6];
    if (tid < 6) {
        const int o = tid == 0 ? L.o_s_att : tid == 1 ? L.o_s_def : tid == 2 ? L.o_s_ha
                    : tid == 3 ? L.o_s_aa : tid == 4 ? L.o_s_hd : L.o_s_ad;
        scal[tid] = exp(z[o]);
    } else if (tid < 8) {
        const dcd::SigSite ss = dcd::sig_site(z[tid == 6 ? L.o_u : L.o_corr]);
        double* q = scal + 6 + (tid - 6) * 6;
        q[0] = ss.v; q[1] = ss.dv; q[2] = ss.log_v; q[3] = ss.log_1mv; q[4] = ss.sig; q[5] = ss.sp_sum;
    }
    __syncthreads();
    const double* su_ = scal + 6;       // u site
    const double* sc_ = scal + 12;      // corr_coef_raw site
    const dcd::Bounds b = dcd::bounds_from(A.F, sc_[0], sc_[1], sc_[4]);
    // adjoint of the bounds: one table per workgroup in LDS
    __shared__ dcd::Coupling C;
    __shared__ dcd::CouplingFix CF[2];
    dcd::build_coupling(A.F, b, &C, CF, tid);
    const int cn = C.n;
    auto coupled = [&](int cell, int which, double base) {
        double v = base;
        for (int e = 0; e < cn; ++e)
            if (C.cell[e] == cell && C.which[e] == which) v += C.val[e];
        return v;
    };
    const double s_att = scal[0], s_def = scal[1], s_ha = scal[2], s_aa = scal[3], s_hd = scal[4],
                 s_ad = scal[5];
    const double u = su_[0], du = su_[1], su = su_[4];
    const double rp = 2.0 * u - 1.0, vv = 1.0 - rp * rp, log_vv = log(vv);

    double loc[NEU_SUMS];
#pragma unroll
    for (int i = 0; i < NEU_SUMS; ++i) loc[i] = 0.0;
    for (int t = tid; t < T; t += NEU_EPI) {
        const double* Ac = A.F.acc + (size_t)t * dcd::A_N;
        double G6[dcd::A_N];
#pragma unroll
        for (int j = 0; j < dcd::A_N; ++j) G6[j] = t == tid ? pG[j] : Ac[j];
        bool hit = false;  // (at most three fixtures' teams carry a bounds adjoint)
        for (int e = 0; e < cn; ++e) hit = hit || (C.cell[e] == t && C.which[e] < dcd::A_N);
        if (hit) {
#pragma unroll
            for (int j = 0; j < dcd::A_N; ++j) G6[j] = coupled(t, j, G6[j]);
        }
        const double G_att = G6[dcd::A_ATT], G_def = G6[dcd::A_DEF], G_hat = G6[dcd::A_HATT],
                     G_adf = G6[dcd::A_ADEF], G_aat = G6[dcd::A_AATT], G_hdf = G6[dcd::A_HDEF];
        const bool first = t == tid;
        const double sa = first ? pz[0] : z[L.o_sat + t], sd = first ? pz[1] : z[L.o_sdt + t];
        const double e = sd - rp * sa;
        grad[L.o_sat + t] = -(s_att * G_att - sa + rp * e / vv);
        grad[L.o_sdt + t] = -(s_def * G_def - e / vv);
        const double hat = first ? pz[2] : z[L.o_hat + t], aat = first ? pz[3] : z[L.o_aat + t],
                     hdf = first ? pz[4] : z[L.o_hdf + t], adf = first ? pz[5] : z[L.o_adf + t];
        grad[L.o_hat + t] = -(s_ha * G_hat - hat);
        grad[L.o_aat + t] = -(s_aa * G_aat - aat);
        grad[L.o_hdf + t] = -(s_hd * G_hdf - hdf);
        grad[L.o_adf + t] = -(s_ad * G_adf - adf);
        loc[0] += e * sa / vv - rp * e * e / (vv * vv) + rp / vv;
        loc[1] += sa * G_att;
        loc[2] += sd * G_def;
        loc[3] += G_def;
        loc[4] += G_hat; loc[5] += G_aat; loc[6] += G_hdf; loc[7] += G_adf;
        loc[8] += hat * G_hat; loc[9] += aat * G_aat; loc[10] += hdf * G_hdf; loc[11] += adf * G_adf;
        loc[12] += -0.5 * sa * sa - HALF_LOG_2PI - 0.5 * e * e / vv - 0.5 * log_vv - HALF_LOG_2PI
                   - 0.5 * (hat * hat + aat * aat + hdf * hdf + adf * adf) - 4.0 * HALF_LOG_2PI;
        for (int k = 0; k < K; ++k) {
            const double xv = A.F.xs[(size_t)t * K + k];
            atomicAdd(&sums[NEU_SUMS + k], xv * G_att);
            atomicAdd(&sums[NEU_SUMS + K + k], xv * G_def);
        }
    }
    // wave sums by DPP (same-address LDS atomics from many lanes serialise badly), then one
    // LDS atomic per wave and value
    if ((tid & ~63) < T) {
#pragma unroll
        for (int i = 0; i < NEU_SUMS; ++i) {
            const double v = dc::wave_sum_f64(loc[i]);
            if ((tid & 63) == 0) atomicAdd(&sums[i], v);
        }
    }
    __syncthreads();

    for (int k = tid; k < 2 * K; k += NEU_EPI) {  // covariate coefficients ~ N(0,1)
        const int o = k < K ? L.o_bA + k : L.o_bD + k - K;
        grad[o] = -(sums[NEU_SUMS + k] - z[o]);
    }
    for (int cf = tid; cf < L.C; cf += NEU_EPI) {  // confederation strengths ~ N(0,1) (loc 0, scale 1)
        const double G = coupled(cf, dcd::A_N, A.F.cacc[cf]);
        grad[L.o_conf + cf] = -(G - z[L.o_conf + cf]);
    }
    // scalar sites, one per lane of wave 0: 0..5 HalfNormal stds | 6..9 Normal means | 10 u | 11 corr
    if (tid < 64) {
        double Lp = 0.0;
        if (tid < 6) {  // HalfNormal(scale) in log space: std_attack / std_defence scale 0.5, others 1
            const int o = tid == 0 ? L.o_s_att : tid == 1 ? L.o_s_def : tid == 2 ? L.o_s_ha
                        : tid == 3 ? L.o_s_aa : tid == 4 ? L.o_s_hd : L.o_s_ad;
            const double sv = tid == 0 ? s_att : tid == 1 ? s_def : tid == 2 ? s_ha
                            : tid == 3 ? s_aa : tid == 4 ? s_hd : s_ad;
            const double scale = tid < 2 ? 0.5 : 1.0;
            const double dotG = tid == 0 ? sums[1] : tid == 1 ? sums[2] : sums[6 + tid];
            const double r = sv / scale;
            Lp = LN2 - (tid < 2 ? -LN2 : 0.0) - HALF_LOG_2PI - 0.5 * r * r + z[o];  // log(0.5) = -ln 2
            grad[o] = -(sv * dotG - r * r + 1.0);
        } else if (tid < 10) {
            const int j = tid - 6;
            const int o = j == 0 ? L.o_mha : j == 1 ? L.o_maa : j == 2 ? L.o_mhd : L.o_mad;
            const double mu = (j & 1) ? -0.1 : 0.1;
            const double mean = z[o], r = (mean - mu) / 0.2;
            Lp = -0.5 * r * r + 1.6094379124341003 - HALF_LOG_2PI;
            grad[o] = -(sums[4 + j] - (mean - mu) / 0.04);
        } else if (tid == 10) {  // u ~ Beta(2,4) through the sigmoid
            Lp = su_[2] + 3.0 * su_[3] + 2.995732273553991 - su_[5];
            grad[L.o_u] = -((1.0 / u - 3.0 / (1.0 - u)) * du + 2.0 * sums[0] * du + (1.0 - 2.0 * su));
        } else if (tid == 11) {  // corr_coef_raw ~ Beta(2,2)
            Lp = sc_[2] + sc_[3] + 1.791759469228055 - sc_[5];
            grad[L.o_corr] = -((1.0 / b.q - 1.0 / (1.0 - b.q)) * b.dq + (1.0 - 2.0 * b.sq) +
                               b.G_rho * (b.UB - b.LB) * b.dq);
        } else if (tid == 12) {
            const double m = z[L.o_md];
            Lp = -0.5 * m * m - HALF_LOG_2PI;
            grad[L.o_md] = -(sums[3] - m);
        }
        for (int k = tid; k < 2 * K; k += 64) {
            const int o = k < K ? L.o_bA + k : L.o_bD + k - K;
            Lp += -0.5 * z[o] * z[o] - HALF_LOG_2PI;
        }
        for (int cf = tid; cf < L.C; cf += 64) {
            const double v = z[L.o_conf + cf];
            Lp += -0.5 * v * v - HALF_LOG_2PI;
        }
        Lp = dcd::wave_sum(Lp);
        if (tid == 0) {
            const double Ltot = sums[12] + A.F.sc[dcd::SC_U] - A.F.lgsum + Lp;
            A.F.potential[0] = -Ltot;
            if (A.F.aux) {
                A.F.aux[0] = b.rho;
                A.F.aux[1] = b.LB;
                A.F.aux[2] = b.UB;
                A.F.aux[3] = b.q;
            }
        }
    }
}

}  // namespace dcn
