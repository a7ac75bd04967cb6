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

struct alignas(16) FusedFixture {  // one 16-byte load per fixture
    uint16_t h, a;
    uint8_t x, y, nv, hc, ac, pad[3];
    float w;
};
static_assert(sizeof(FusedFixture) == 16, "FusedFixture is one dwordx4");

struct NeuArgs {
    dcd::DynArgs F;   // fixtures, cells, scratch (acc | sc), z / potential / grad / aux
    NeuLayout L;
    const FusedFixture* fxp;   // neu_big: the fixtures again, one 16-byte record each (weight 1 when unweighted)
    int runs;                  // neu_big: 1 = per-run arithmetic (the host checked the run tables' capacity)
};
// neu_big, round 4: a wave's table of the (venue, home, away[, confederations]) runs of its part of the slice
constexpr int NEU_RUNS_MAX = 32;   // runs per wave (host: max over the waves' parts, else the per-fixture form)
constexpr int NEU_RUN_W = 10;      // key | first fixture | W, WX, WY, W00, W10, W01, W11 | spare

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

// dcd::sig_site through the short float64 routines (lean_math.hip.h: <= 2 ulp; a libm call is ~1 us of
// dependent instructions on the one lane that runs it)
__device__ inline dcd::SigSite sig_site_lean(double zr) {
    const double az = fabs(zr), ez = dc::lean::exp(-az), l1 = dc::lean::log1p_pos(ez);
    const double sp_pos = az + l1;                 // softplus(|z|)
    const double s_abs = dc::lean::rcp(1.0 + ez);
    dcd::SigSite r;
    r.sig = zr >= 0 ? s_abs : 1.0 - s_abs;
    r.sp_sum = sp_pos + l1;
    r.v = r.sig;
    r.dv = r.sig * (1.0 - r.sig);
    r.log_v = zr >= 0 ? -l1 : -sp_pos;             // log sigmoid(z)   = -softplus(-z)
    r.log_1mv = zr >= 0 ? -sp_pos : -l1;           // log(1-sigmoid(z)) = -softplus(z)
    if (r.sig < dc::SIG_LO || r.sig > dc::SIG_HI) {
        r.v = r.sig < dc::SIG_LO ? dc::SIG_LO : dc::SIG_HI;
        r.dv = 0.0;
        r.log_v = log(r.v);
        r.log_1mv = log1p(-r.v);
    }
    return r;
}

// BLOCK threads of ONE workgroup; SC1: the accumulators and scratch words were produced by other
// workgroups of THIS launch (neu_big's last-arriving workgroup): L1-bypassing loads
#ifdef DC_STAMPS
#define EPI_STAMP(k) do { if (threadIdx.x == 0 && epi_stamp) epi_stamp[k] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define EPI_STAMP(k) do { } while (0)
#endif
// LDSACC (neu_big): A.F.acc is this workgroup's LDS copy -- the bounds' adjoint entries are added into it by one
// lane each instead of being searched by every team
template <int BLOCK, bool SC1, bool LDSACC = false>
__device__ __forceinline__ void epilogue_body(const NeuArgs& A, double* sums, unsigned long long* epi_stamp = nullptr,
                                              double* lds_acc = nullptr, const double* pre_scal = nullptr) {
    constexpr int NEU_EPI = BLOCK;
    const NeuLayout& L = A.L;
    EPI_STAMP(0);
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
        for (int j = 0; j < dcd::A_N; ++j) pG[j] = SC1 ? dc::ld_sc1(&Ac[j]) : Ac[j];
        pz[0] = z[L.o_sat + t0]; pz[1] = z[L.o_sdt + t0]; pz[2] = z[L.o_hat + t0];
        pz[3] = z[L.o_aat + t0]; pz[4] = z[L.o_hdf + t0]; pz[5] = z[L.o_adf + t0];
    }
    // (SC1: the maxima, dL/d rho and the arg-extremal fixtures' indices in the same round of loads)
    double pre_sc[4] = {0.0, 0.0, 0.0, 0.0};
    unsigned long long pre_idx[3] = {0ull, 0ull, 0ull};
    if (SC1) {
        const unsigned long long* scu = reinterpret_cast<const unsigned long long*>(A.F.sc);
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            pre_sc[j] = dc::ld_sc1(&A.F.sc[dcd::SC_MAXP + j]);
            pre_idx[j] = __hip_atomic_load(&scu[dcd::SC_IDXP + j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        static_assert(dcd::SC_MAXH == dcd::SC_MAXP + 1 && dcd::SC_MAXA == dcd::SC_MAXP + 2, "maxima in a row");
        pre_sc[3] = dc::ld_sc1(&A.F.sc[dcd::SC_GRHO]);
    }
    for (int i = tid; i < NEU_SUMS + 2 * K; i += NEU_EPI) sums[i] = 0.0;
    // scalar sites first, in parallel lanes of one wave (a float64 libm call costs ~1 us of
    // dependent instructions: they must not run one after another on one lane):
    //   lanes 0..5 exp(std sites) | lanes 6, 7 sigmoid sites u, corr_coef_raw
    __shared__ double scal_own[6 + 2 * 6];
    const double* scal = LDSACC ? pre_scal : scal_own;   // (neu_big: worked out in its barrier's shadow, by every workgroup)
    if (LDSACC) {
    } else if (tid < 6) {
        const int o = tid == 0 ? L.o_s_att : tid == 1 ? L.o_s_def : tid == 2 ? L.o_s_ha
                    : tid == 3 ? L.o_s_aa : tid == 4 ? L.o_s_hd : L.o_s_ad;
        scal_own[tid] = dc::lean::exp(z[o]);
    } else if (tid < 8) {
        const dcd::SigSite ss = sig_site_lean(z[tid == 6 ? L.o_u : L.o_corr]);
        double* q = scal_own + 6 + (tid - 6) * 6;
        q[0] = ss.v; q[1] = ss.dv; q[2] = ss.log_v; q[3] = ss.log_1mv; q[4] = ss.sig; q[5] = ss.sp_sum;
    }
    __syncthreads();
    EPI_STAMP(1);
    const double* su_ = scal + 6;       // u site
    const double* sc_ = scal + 12;      // corr_coef_raw site
    dcd::Bounds b;
    if (SC1) {
        b.M = pre_sc[0]; b.Lh = pre_sc[1]; b.La = pre_sc[2];
        b.q = sc_[0]; b.dq = sc_[1]; b.sq = sc_[4];
        b.UB = b.M > 1.0 ? 1.0 / b.M : 1.0;
        b.LB = -1.0 / fmax(b.Lh, b.La);
        b.rho = b.LB + b.q * (b.UB - b.LB);
        b.G_rho = pre_sc[3];
    } else {
        b = dcd::bounds_from(A.F, sc_[0], sc_[1], sc_[4]);
    }
    // adjoint of the bounds: one table per workgroup in LDS
    __shared__ dcd::Coupling C;
    __shared__ dcd::CouplingFix CF[2];
    dcd::build_coupling<SC1>(A.F, b, &C, CF, tid, SC1 ? pre_idx : nullptr);
    EPI_STAMP(2);
    const int cn = C.n;
    if (LDSACC) {   // (one entry per lane; build_coupling ends with a barrier)
        if (tid < cn && C.which[tid] < dcd::A_N)
            atomicAdd(&lds_acc[C.cell[tid] * dcd::A_N + C.which[tid]], C.val[tid]);   // (the LDS array itself: through
                                                                                       // A.F.acc it is a generic pointer)
        __syncthreads();
    }
    auto coupled = [&](int cell, int which, double base) {
        double v = base;
        for (int e = 0; e < cn; ++e)
            if (C.cell[e] == cell && C.which[e] == which) v += C.val[e];
        return v;
    };
    const double s_att = scal[0], s_def = scal[1], s_ha = scal[2], s_aa = scal[3], s_hd = scal[4],
                 s_ad = scal[5];
    const double u = su_[0], du = su_[1], su = su_[4];
    const double rp = 2.0 * u - 1.0, vv = 1.0 - rp * rp, log_vv = dc::lean::log(vv);
    const double ivv = dc::lean::rcp(vv);   // (one reciprocal: a float64 division is ~30 dependent instructions)

    double loc[NEU_SUMS];
#pragma unroll
    for (int i = 0; i < NEU_SUMS; ++i) loc[i] = 0.0;
    for (int t = tid; t < T; t += NEU_EPI) {
        const double* Ac = A.F.acc + (size_t)t * dcd::A_N;
        double G6[dcd::A_N];
#pragma unroll
        for (int j = 0; j < dcd::A_N; ++j) G6[j] = LDSACC ? Ac[j] : t == tid ? pG[j] : SC1 ? dc::ld_sc1(&Ac[j]) : Ac[j];
        if (!LDSACC) {   // (at most three fixtures' teams carry a bounds adjoint: ONE pass over the table, every LDS read of it
            // independent of the others -- entry by entry behind a search per adjoint, these reads were a chain
            // of ~120 dependent LDS round trips: 9 of the epilogue's 15 us)
            double adj[dcd::A_N];
#pragma unroll
            for (int j = 0; j < dcd::A_N; ++j) adj[j] = 0.0;
#pragma unroll
            for (int e = 0; e < 18; ++e) {
                const bool mine = e < cn && C.cell[e] == t;
                const int wh = C.which[e];
                const double v = C.val[e];
#pragma unroll
                for (int j = 0; j < dcd::A_N; ++j) adj[j] += (mine && wh == j) ? v : 0.0;
            }
#pragma unroll
            for (int j = 0; j < dcd::A_N; ++j) G6[j] += adj[j];
        }
        const double G_att = G6[dcd::A_ATT], G_def = G6[dcd::A_DEF], G_hat = G6[dcd::A_HATT],
                     G_adf = G6[dcd::A_ADEF], G_aat = G6[dcd::A_AATT], G_hdf = G6[dcd::A_HDEF];
        const bool first = t == tid;
        const double sa = first ? pz[0] : z[L.o_sat + t], sd = first ? pz[1] : z[L.o_sdt + t];
        const double e = sd - rp * sa;
        grad[L.o_sat + t] = -(s_att * G_att - sa + rp * e * ivv);
        grad[L.o_sdt + t] = -(s_def * G_def - e * ivv);
        const double hat = first ? pz[2] : z[L.o_hat + t], aat = first ? pz[3] : z[L.o_aat + t],
                     hdf = first ? pz[4] : z[L.o_hdf + t], adf = first ? pz[5] : z[L.o_adf + t];
        grad[L.o_hat + t] = -(s_ha * G_hat - hat);
        grad[L.o_aat + t] = -(s_aa * G_aat - aat);
        grad[L.o_hdf + t] = -(s_hd * G_hdf - hdf);
        grad[L.o_adf + t] = -(s_ad * G_adf - adf);
        loc[0] += e * sa * ivv - rp * e * e * (ivv * ivv) + rp * ivv;
        loc[1] += sa * G_att;
        loc[2] += sd * G_def;
        loc[3] += G_def;
        loc[4] += G_hat; loc[5] += G_aat; loc[6] += G_hdf; loc[7] += G_adf;
        loc[8] += hat * G_hat; loc[9] += aat * G_aat; loc[10] += hdf * G_hdf; loc[11] += adf * G_adf;
        loc[12] += -0.5 * sa * sa - HALF_LOG_2PI - 0.5 * e * e * ivv - 0.5 * log_vv - HALF_LOG_2PI
                   - 0.5 * (hat * hat + aat * aat + hdf * hdf + adf * adf) - 4.0 * HALF_LOG_2PI;
        for (int k = 0; k < K; ++k) {
            const double xv = A.F.xs[(size_t)t * K + k];
            atomicAdd(&sums[NEU_SUMS + k], xv * G_att);
            atomicAdd(&sums[NEU_SUMS + K + k], xv * G_def);
        }
    }
    EPI_STAMP(3);
    // wave sums by DPP (same-address LDS atomics from many lanes serialise badly), then one
    // LDS atomic per wave and value
    if ((tid & ~63) < T) {
        dc::wave_sumN_f64(loc);   // (the thirteen chains interleaved step by step, not one after another)
        if ((tid & 63) == 0) {
#pragma unroll
            for (int i = 0; i < NEU_SUMS; ++i) atomicAdd(&sums[i], loc[i]);
        }
    }
    __syncthreads();
    EPI_STAMP(4);

    for (int k = tid; k < 2 * K; k += NEU_EPI) {  // covariate coefficients ~ N(0,1)
        const int o = k < K ? L.o_bA + k : L.o_bD + k - K;
        grad[o] = -(sums[NEU_SUMS + k] - z[o]);
    }
    for (int cf = tid; cf < L.C; cf += NEU_EPI) {  // confederation strengths ~ N(0,1) (loc 0, scale 1)
        const double G = coupled(cf, dcd::A_N, SC1 ? dc::ld_sc1(&A.F.cacc[cf]) : A.F.cacc[cf]);
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
            const double Ltot = sums[12] + (SC1 ? dc::ld_sc1(&A.F.sc[dcd::SC_U]) : A.F.sc[dcd::SC_U]) - A.F.lgsum + Lp;
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


__global__ __launch_bounds__(NEU_EPI) void neu_epilogue(NeuArgs A) {
    extern __shared__ double sums[];  // [NEU_SUMS + 2K]
    epilogue_body<NEU_EPI, false>(A, sums);
}

// ---- the whole evaluation in ONE launch for any number of fixtures (the multi-launch path above is
// four launches: 58 us at N = 1e6, 16 of them the one-workgroup epilogue and its launch boundary).
// Every workgroup (one per CU, all resident) works the T cell records out for itself in LDS -- the
// z side is O(T): no hand-off -- and takes one contiguous slice of the fixtures:
//   2  rates (kept in LDS for phase 3), the weighted Poisson part of the value, maxima
//   -- grid barrier (dcd::tree_arrive / tree_wait on the context's counters)
//   3  tau terms and the adjoints into LDS-private accumulators (a wave that sits on one
//      (venue, home, away) run -- the fixtures are sorted by it -- adds its sums once), flushed with
//      6T + C global float64 atomics per workgroup
//   -- arrival ticket: the last workgroup runs the epilogue (epilogue_body, L1-bypassing loads) and puts
//      the scratch and the counters back to zero for the next launch
constexpr int NEU_BIG_BLOCK = 512;    // (1024 threads leave 128 VGPRs per lane: the epilogue alone wants 254 -- 118 spilled, its
                                      // team loop and phase 3 ran at the speed of scratch memory)
constexpr int NEU_BIG_GROUPS = 16;   // copies of the global scratch (see neu_big)
#ifdef DC_STAMPS  // diagnostic build: thread 0 of the LAST workgroup overwrites grad[o_sat + 0..15] with the 100 MHz
                  // clock at points of its way through the launch (tools/neutral_big_stamps.py)
#define NEU_STAMP(k) do { if (threadIdx.x == 0) nstamp_[k] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define NEU_STAMP(k) do { } while (0)
#endif
__host__ __device__ inline size_t big_lds_bytes(const NeuLayout& L, long long rate_cap) {
    return ((size_t)L.T * (2 * dcd::P_N + 2 * dcd::A_N) + 2 * (size_t)L.C + dcd::SC_N + 4 * (size_t)rate_cap + NEU_SUMS +
            2 * (size_t)L.K + 2) * 8;
}
__global__ __launch_bounds__(NEU_BIG_BLOCK) void neu_big(NeuArgs A) {
    constexpr int WAVES = NEU_BIG_BLOCK / 64;
    extern __shared__ __attribute__((aligned(16))) double big_lds[];   // fixtures [cap] (16 B) | rates [cap][2] | cells [T][P_N] | accumulators [T][A_N] + [C] | epilogue sums
    __shared__ unsigned long long shm[3 * (WAVES > NEU_BIG_GROUPS ? WAVES : NEU_BIG_GROUPS)];
    __shared__ double shr[2 * WAVES];
    __shared__ int s_ok, s_last, s_slow;
    __shared__ double s_scal[6 + 2 * 6];
    const NeuLayout& L = A.L;
    const dcd::DynArgs& F = A.F;
    const int T = L.T, K = L.K, C = L.C;
    const double* z = F.z;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const unsigned int nb = gridDim.x;
    FusedFixture* const lfx = reinterpret_cast<FusedFixture*>(big_lds);   // (16-byte aligned: first)
    double* const lrate = big_lds + 2 * (size_t)F.rate_cap;
    double* const lcell = lrate + 2 * (size_t)F.rate_cap;
    double* const lexp = lcell + (size_t)T * dcd::P_N;   // [T][P_N]: exp(+-record) -- a rate is a product of two
    double* const lacc = lexp + (size_t)T * dcd::P_N;    // [T][A_N] then [C]
    double* const lconf = lacc + (size_t)T * dcd::A_N;
    double* const lred = lconf + C;                      // [scratch_n]: the copies of the global scratch, summed
    double* const sums = lred + F.scratch_n;
    // Global scratch: NEU_BIG_GROUPS copies of acc | sc | cacc, workgroup b adds into copy b % GROUPS -- hundreds
    // of atomics on ONE address are served one after another at the memory side (~35 ns each: 256
    // workgroups x one maximum = 9 us, and the loads that follow queue behind them), 16 per address are not
    // felt.  The arg-extremal indices (rare, one per wave at most) all go to copy 0.
    const int grp = (int)(blockIdx.x % NEU_BIG_GROUPS);
    double* const gacc = F.acc + (size_t)grp * F.scratch_n;
    double* const gsc = gacc + (size_t)T * dcd::A_N;
    double* const gcacc = gsc + dcd::SC_N;
    unsigned long long* scu = reinterpret_cast<unsigned long long*>(F.sc);      // copy 0: the indices
    unsigned long long* gscu = reinterpret_cast<unsigned long long*>(gsc);
    const unsigned int failed = __hip_atomic_load(F.tickets + dcd::TK_FAIL, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#ifdef DC_STAMPS
    unsigned long long nstamp_[16] = {};
#endif
    NEU_STAMP(0);
    const long long share = (F.n + nb - 1) / nb;
    const long long i_lo = (long long)blockIdx.x * share < F.n ? (long long)blockIdx.x * share : F.n;
    const long long i_hi = i_lo + share < F.n ? i_lo + share : F.n;

    if (tid == 0) s_slow = 0;
    __syncthreads();
    // ---- this workgroup's fixtures into LDS (one 16-byte record each, every load in flight before the first
    // store: read from memory round by round, each round of phases 2 and 3 waited a memory latency)
    const int n_mine = (int)(i_hi - i_lo);
    {
        static_assert(sizeof(FusedFixture) == sizeof(uint4), "one dwordx4");
        const uint4* src = reinterpret_cast<const uint4*>(A.fxp + i_lo);
        uint4* dst = reinterpret_cast<uint4*>(lfx);
        for (int k0 = 0; k0 < n_mine; k0 += NEU_BIG_BLOCK * 4) {
            uint4 w[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int k = k0 + j * NEU_BIG_BLOCK + tid;
                w[j] = src[k < n_mine ? k : n_mine - 1];
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int k = k0 + j * NEU_BIG_BLOCK + tid;
                if (k < n_mine) dst[k] = w[j];
            }
        }
    }
    NEU_STAMP(1);
    // ---- cells of every team, accumulators to zero
    {
        const double e_att = dc::lean::exp(z[L.o_s_att]), e_def = dc::lean::exp(z[L.o_s_def]),
                     e_ha = dc::lean::exp(z[L.o_s_ha]), e_aa = dc::lean::exp(z[L.o_s_aa]),
                     e_hd = dc::lean::exp(z[L.o_s_hd]), e_ad = dc::lean::exp(z[L.o_s_ad]);
        const double m_def = z[L.o_md], m_ha = z[L.o_mha], m_aa = z[L.o_maa], m_hd = z[L.o_mhd], m_ad = z[L.o_mad];
        for (int t = tid; t < T; t += NEU_BIG_BLOCK) {
            double att = 0.0, def = m_def;
            for (int k = 0; k < K; ++k) {
                const double xv = F.xs[(size_t)t * K + k];
                att += xv * z[L.o_bA + k];
                def += xv * z[L.o_bD + k];
            }
            att += z[L.o_sat + t] * e_att;
            def += z[L.o_sdt + t] * e_def;
            const double hat = m_ha + e_ha * z[L.o_hat + t];
            const double aat = m_aa + e_aa * z[L.o_aat + t];
            const double hdf = m_hd + e_hd * z[L.o_hdf + t];
            const double adf = m_ad + e_ad * z[L.o_adf + t];
            double* P = lcell + (size_t)t * dcd::P_N;
            double* E = lexp + (size_t)t * dcd::P_N;
            const double rec[dcd::P_N] = {att + hat, att + aat, def + hdf, def + adf, att, def};
            static_assert(dcd::P_AH == 0 && dcd::P_AA == 1 && dcd::P_BH == 2 && dcd::P_BA == 3 && dcd::P_ATT == 4 &&
                          dcd::P_DEF == 5, "record order");
            bool far = false;
#pragma unroll
            for (int j = 0; j < dcd::P_N; ++j) {
                P[j] = rec[j];
                // (attack-type entries enter a rate with +, defence-type ones with -; beyond +-300 a factor alone
                // could overflow although the rate does not: the workgroup then takes exp of the difference)
                E[j] = dc::lean::exp((j == dcd::P_AH || j == dcd::P_AA || j == dcd::P_ATT) ? rec[j] : -rec[j]);
                far = far || fabs(rec[j]) > 300.0;
            }
            if (far) s_slow = 1;
        }
        for (int k = tid; k < T * dcd::A_N + C; k += NEU_BIG_BLOCK) lacc[k] = 0.0;
    }
    __syncthreads();
    const bool slow = s_slow != 0 || C != 0;   // (confederation strengths shift the exponent per fixture: exact form)

    NEU_STAMP(2);
    // ---- Round 4: PER-RUN ARITHMETIC (A.runs).  The fixtures are sorted by (venue, home, away[, confederations])
    // and every fixture of a run has the same rates and the same four tau arguments: per fixture the kernel spent
    // ~20 float64 operations on the rates and ~50 (a log and a reciprocal for a third of them) on the adjoints --
    // phases 2 and 3 were 3.1 + 5.2 us of a 26.7 us launch, bound by float64 issue.  What the value and the
    // adjoints need from a run's fixtures is LINEAR in seven sums (the weights of all its fixtures, of their goals
    // on either side, of its (0,0) / (1,0) / (0,1) / (1,1) scorelines).  So a wave walks its part of the slice once,
    // adds those seven per lane while a step stays inside one run (as it carried the adjoints before), reduces them
    // at the run's end and files one record per run; then LANE r works run r out -- rates, Poisson part and
    // maxima in front of the barrier, tau terms and adjoints behind it (its rates stay in registers).  The
    // fixtures are still streamed and classified on every evaluation.  (dc_vec got the same treatment: DESIGN.md 4a.)
    const int per_wave_r = (n_mine + WAVES - 1) / WAVES;
    const int rw0 = wave * per_wave_r < n_mine ? wave * per_wave_r : n_mine;
    const int rw1 = rw0 + per_wave_r < n_mine ? rw0 + per_wave_r : n_mine;
    double* const myruns = lrate + (size_t)wave * NEU_RUNS_MAX * NEU_RUN_W;
    int n_runs = 0;                      // (wave-uniform)
    unsigned long long r_key = 0ull;     // lane r < n_runs: its run
    long long r_first = 0;
    double rW = 0.0, rWX = 0.0, rWY = 0.0, rW00 = 0.0, rW10 = 0.0, rW01 = 0.0, rW11 = 0.0, r_lh = 0.0, r_la = 0.0;
    if (A.runs) {
        unsigned long long run_key = ~0ull;
        long long run_first = 0;
        double sv[7] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
        auto file_run = [&](unsigned long long kk, long long first, double (&t)[7]) {   // (wave-uniform call, totals in every lane)
            if (lane == 0 && n_runs < NEU_RUNS_MAX) {
                double* R = myruns + (size_t)n_runs * NEU_RUN_W;
                R[0] = __longlong_as_double((long long)kk);
                R[1] = __longlong_as_double(first);
#pragma unroll
                for (int j = 0; j < 7; ++j) R[2 + j] = t[j];
            }
            ++n_runs;
        };
        auto flush_carried = [&]() {
            if (run_key == ~0ull) return;
            dc::wave_sumN_f64(sv);
            file_run(run_key, run_first, sv);
            run_key = ~0ull;
#pragma unroll
            for (int j = 0; j < 7; ++j) sv[j] = 0.0;
        };
        for (int base = rw0; base < rw1; base += 64) {   // (wave-uniform trip count)
            const int k = base + lane;
            const bool active = k < rw1;
            unsigned long long key = 0ull;
            double cv[7] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
            if (active) {
                const FusedFixture f = lfx[k];
                const int x = f.x, y = f.y;
                const double wi = (double)f.w;
                key = ((unsigned long long)f.h << 33) | ((unsigned long long)f.a << 17) |
                      ((unsigned long long)(f.nv != 0) << 16) | ((unsigned long long)f.hc << 8) | (unsigned long long)f.ac;
                cv[0] = wi; cv[1] = wi * x; cv[2] = wi * y;
                const bool low = x <= 1 && y <= 1;
                cv[3] = low && x == 0 && y == 0 ? wi : 0.0;
                cv[4] = low && x == 1 && y == 0 ? wi : 0.0;
                cv[5] = low && x == 0 && y == 1 ? wi : 0.0;
                cv[6] = low && x == 1 && y == 1 ? wi : 0.0;
            }
            // (the active lanes are a prefix of the wave: lane 0 is active here)
            const unsigned long long k0 = ((unsigned long long)(unsigned int)__builtin_amdgcn_readfirstlane((int)(key >> 32)) << 32) |
                                          (unsigned int)__builtin_amdgcn_readfirstlane((int)key);
            if (__all(!active || key == k0)) {   // one run in this step
                if (k0 != run_key) {
                    flush_carried();
                    run_key = k0;
                    run_first = i_lo + base;
                }
#pragma unroll
                for (int j = 0; j < 7; ++j) sv[j] += cv[j];
                continue;
            }
            flush_carried();
            // a step that straddles runs: one record per run of the step
            const unsigned long long prev = __shfl_up(key, 1, 64);
            unsigned long long rest = __ballot(active && (lane == 0 || key != prev));
            while (rest) {   // (wave-uniform)
                const int l0 = __ffsll((long long)rest) - 1;
                rest &= rest - 1;
                const int l1 = rest ? __ffsll((long long)rest) - 1 : 64;
                const bool in = active && lane >= l0 && lane < l1;
                double t[7];
#pragma unroll
                for (int j = 0; j < 7; ++j) t[j] = in ? cv[j] : 0.0;
                dc::wave_sumN_f64(t);
                const unsigned long long kk = ((unsigned long long)(unsigned int)__builtin_amdgcn_readlane((int)(key >> 32), l0) << 32) |
                                              (unsigned int)__builtin_amdgcn_readlane((int)key, l0);
                file_run(kk, i_lo + base + l0, t);
            }
        }
        flush_carried();
        if (n_runs > NEU_RUNS_MAX) n_runs = NEU_RUNS_MAX;   // (cannot happen: the host counted)
        if (lane < n_runs) {
            const double* R = myruns + (size_t)lane * NEU_RUN_W;
            r_key = (unsigned long long)__double_as_longlong(R[0]);
            r_first = __double_as_longlong(R[1]);
            rW = R[2]; rWX = R[3]; rWY = R[4]; rW00 = R[5]; rW10 = R[6]; rW01 = R[7]; rW11 = R[8];
        }
    }
    // ---- phase 2: rates, Poisson part of the value, maxima
    double Ui = 0.0;
    {
        double mP = 0.0, mH = 0.0, mA = 0.0;
        if (A.runs && lane < n_runs) {   // lane r: run r
            const int h = (int)(r_key >> 33) & 0xFFFF, a = (int)(r_key >> 17) & 0xFFFF, hc_ = (int)(r_key >> 8) & 0xFF,
                      ac_ = (int)r_key & 0xFF;
            const bool nvf = (r_key >> 16) & 1;
            const int oh_att = h * dcd::P_N + (nvf ? dcd::P_ATT : dcd::P_AH), oa_def = a * dcd::P_N + (nvf ? dcd::P_DEF : dcd::P_BA);
            const int oa_att = a * dcd::P_N + (nvf ? dcd::P_ATT : dcd::P_AA), oh_def = h * dcd::P_N + (nvf ? dcd::P_DEF : dcd::P_BH);
            double eh = lcell[oh_att] - lcell[oa_def];
            double ea = lcell[oa_att] - lcell[oh_def];
            if (C) {  // bpl/neutral_dixon_coles_WC.py:188-203
                const double d = F.cs[hc_] - F.cs[ac_];
                eh += d;
                ea -= d;
            }
            double lh = lexp[oh_att] * lexp[oa_def], la = lexp[oa_att] * lexp[oh_def];
            if (slow) {   // (workgroup-uniform)
                lh = dc::lean::exp(eh);
                la = dc::lean::exp(ea);
            }
            r_lh = lh;
            r_la = la;
            Ui += rWX * eh - rW * lh + rWY * ea - rW * la;
            mP = lh * la;
            mH = lh;
            mA = la;
        }
        for (int k = tid; !A.runs && k < n_mine; k += NEU_BIG_BLOCK) {
            const FusedFixture f = lfx[k];
            const int h = f.h, a = f.a, x = f.x, y = f.y;
            const bool nvf = f.nv != 0;
            const double wi = (double)f.w;
            const int oh_att = h * dcd::P_N + (nvf ? dcd::P_ATT : dcd::P_AH), oa_def = a * dcd::P_N + (nvf ? dcd::P_DEF : dcd::P_BA);
            const int oa_att = a * dcd::P_N + (nvf ? dcd::P_ATT : dcd::P_AA), oh_def = h * dcd::P_N + (nvf ? dcd::P_DEF : dcd::P_BH);
            double eh = lcell[oh_att] - lcell[oa_def];
            double ea = lcell[oa_att] - lcell[oh_def];
            if (C) {  // bpl/neutral_dixon_coles_WC.py:188-203
                const double d = F.cs[f.hc] - F.cs[f.ac];
                eh += d;
                ea -= d;
            }
            double lh = lexp[oh_att] * lexp[oa_def], la = lexp[oa_att] * lexp[oh_def];
            if (slow) {   // (workgroup-uniform)
                lh = dc::lean::exp(eh);
                la = dc::lean::exp(ea);
            }
            lrate[2 * k] = lh;
            lrate[2 * k + 1] = la;
            Ui += wi * (x * eh - lh + y * ea - la);
            mP = fmax(mP, lh * la);
            mH = fmax(mH, lh);
            mA = fmax(mA, la);
        }
        dc::wave_max3_f64(mP, mH, mA);
        if (lane == 0) {  // (positive doubles order like their bit patterns)
            shm[wave * 3 + 0] = (unsigned long long)__double_as_longlong(mP);
            shm[wave * 3 + 1] = (unsigned long long)__double_as_longlong(mH);
            shm[wave * 3 + 2] = (unsigned long long)__double_as_longlong(mA);
        }
        __syncthreads();
        if (tid < 3) {
            unsigned long long m = 0;
            for (int w = 0; w < WAVES; ++w) m = shm[w * 3 + tid] > m ? shm[w * 3 + tid] : m;
            if (m) atomicMax(&gscu[dcd::SC_MAXP + tid], m);
        }
    }
    NEU_STAMP(3);
    dcd::tree_arrive(F.tickets, dcd::TB_2, blockIdx.x, nb);
    // (in the barrier's shadow: the scalar sites for the epilogue, whichever workgroup will run it -- lanes 0..5
    // exp(std sites) | lanes 6, 7 the sigmoid sites u, corr_coef_raw; and corr_coef_raw's sigmoid for phase 3)
    if (tid < 6) {
        const int o = tid == 0 ? L.o_s_att : tid == 1 ? L.o_s_def : tid == 2 ? L.o_s_ha
                    : tid == 3 ? L.o_s_aa : tid == 4 ? L.o_s_hd : L.o_s_ad;
        s_scal[tid] = dc::lean::exp(z[o]);
    } else if (tid < 8) {
        const dcd::SigSite ss = sig_site_lean(z[tid == 6 ? L.o_u : L.o_corr]);
        double* qq = s_scal + 6 + (tid - 6) * 6;
        qq[0] = ss.v; qq[1] = ss.dv; qq[2] = ss.log_v; qq[3] = ss.log_1mv; qq[4] = ss.sig; qq[5] = ss.sp_sum;
    }
    double q, dq, sq;
    {
        const double z_corr = z[L.o_corr];
        const double ezc = dc::lean::exp(-fabs(z_corr));
        const double sc_abs = dc::lean::rcp(1.0 + ezc);
        sq = z_corr >= 0 ? sc_abs : 1.0 - sc_abs;
        q = sq < dc::SIG_LO ? dc::SIG_LO : sq > dc::SIG_HI ? dc::SIG_HI : sq;
        dq = (sq < dc::SIG_LO || sq > dc::SIG_HI) ? 0.0 : sq * (1.0 - sq);
    }
    auto give_up = [&]() {   // a bounded wait expired: NaN outputs (the header's promise), potential AND gradient
        if (blockIdx.x == 0) {
            if (tid == 0) F.potential[0] = __builtin_nan("");
            for (int i = tid; i < L.D; i += NEU_BIG_BLOCK) F.grad[i] = __builtin_nan("");
        }
    };
    if (!dcd::tree_wait(F.tickets, dcd::TB_2, failed, &s_ok, F.fault)) { give_up(); return; }
    NEU_STAMP(4);

    // ---- phase 3: tau terms, adjoints
    {
        // the three maxima: over the copies (one load per lane of wave 0, folded through LDS)
        static_assert(dcd::SC_MAXH == dcd::SC_MAXP + 1 && dcd::SC_MAXA == dcd::SC_MAXP + 2, "maxima in a row");
        if (tid < 3 * NEU_BIG_GROUPS) {
            const int g_ = tid / 3, j_ = tid - 3 * g_;
            shm[tid] = (unsigned long long)__double_as_longlong(
                dc::ld_sc1(F.acc + (size_t)g_ * F.scratch_n + (size_t)T * dcd::A_N + dcd::SC_MAXP + j_));
        }
        __syncthreads();
        unsigned long long mx[3] = {0ull, 0ull, 0ull};
#pragma unroll
        for (int g_ = 0; g_ < NEU_BIG_GROUPS; ++g_)
#pragma unroll
            for (int j_ = 0; j_ < 3; ++j_) mx[j_] = shm[3 * g_ + j_] > mx[j_] ? shm[3 * g_ + j_] : mx[j_];
        const double M = __longlong_as_double((long long)mx[0]), Lh = __longlong_as_double((long long)mx[1]),
                     La = __longlong_as_double((long long)mx[2]);
        const double UB = M > 1.0 ? 1.0 / M : 1.0;
        const double LB = -1.0 / fmax(Lh, La);
        const double rho = LB + q * (UB - LB);
        (void)dq;
        NEU_STAMP(8);
        double ui = 0.0;
        // Each WAVE takes a contiguous part of the slice, 64 fixtures per step: consecutive steps of a wave mostly
        // stay inside one (venue, home, away) run, whose adjoints it then carries in registers (one partial sum
        // per lane) and adds ONCE, when the run ends -- a DPP sum and a set of LDS atomics per step were 1.1 us
        // per step and workgroup
        const int per_wave = (n_mine + WAVES - 1) / WAVES;
        const int w0 = wave * per_wave < n_mine ? wave * per_wave : n_mine;
        const int w1 = w0 + per_wave < n_mine ? w0 + per_wave : n_mine;
        if (A.runs && lane < n_runs) {   // ---- lane r: run r's tau terms and adjoints, from its seven sums
            const double lh = r_lh, la = r_la;
            double gh = rWX - rW * lh, ga = rWY - rW * la;
            auto cls = [&](double Wc, double cc, bool to_h, bool to_a) {
                if (!(Wc > 0.0)) return;
                const double arg = 1.0 + rho * cc;
                if (arg > 0.0) {
                    Ui += Wc * dc::lean::log(arg);
                    const double uu = cc * dc::lean::rcp(arg);
                    ui += Wc * uu;
                    if (to_h) gh += rho * Wc * uu;
                    if (to_a) ga += rho * Wc * uu;
                } else {
                    Ui += Wc * log(0.0);  // -inf (tol = 0, bpl/_util.py:42)
                }
            };
            cls(rW00, -lh * la, true, true);
            cls(rW10, la, false, true);
            cls(rW01, lh, true, false);
            cls(rW11, -1.0, false, false);
            // arg-extremal fixtures: the smallest index among those attaining a maximum -- the run's first fixture
            // (a run cut by a wave or workgroup boundary proposes each piece's first: the maximum of ~0 - i keeps the smallest)
            if (lh * la == M) atomicMax(&scu[dcd::SC_IDXP], ~0ull - (unsigned long long)r_first);
            if (lh == Lh) atomicMax(&scu[dcd::SC_IDXQ], ~0ull - (unsigned long long)r_first);
            if (la == La) atomicMax(&scu[dcd::SC_IDXR], ~0ull - (unsigned long long)r_first);
            const unsigned long long kk = r_key;
            const int h_ = (int)(kk >> 33) & 0xFFFF, a_ = (int)(kk >> 17) & 0xFFFF, hc_ = (int)(kk >> 8) & 0xFF, ac_ = (int)kk & 0xFF;
            const bool nv_ = (kk >> 16) & 1;
            double* Ah = lacc + h_ * dcd::A_N;
            double* Aa = lacc + a_ * dcd::A_N;
            atomicAdd(&Ah[dcd::A_ATT], gh);
            atomicAdd(&Aa[dcd::A_DEF], -gh);
            atomicAdd(&Aa[dcd::A_ATT], ga);
            atomicAdd(&Ah[dcd::A_DEF], -ga);
            if (!nv_) {
                atomicAdd(&Ah[dcd::A_HATT], gh);
                atomicAdd(&Aa[dcd::A_ADEF], -gh);
                atomicAdd(&Aa[dcd::A_AATT], ga);
                atomicAdd(&Ah[dcd::A_HDEF], -ga);
            }
            if (C) {
                atomicAdd(&lconf[hc_], gh - ga);
                atomicAdd(&lconf[ac_], ga - gh);
            }
        }
        unsigned long long run_key = ~0ull;          // (no fixture packs to this: bits 49.. are zero)
        double run_h = 0.0, run_a = 0.0;             // this lane's share of the carried run's sums
        for (int base = w0; !A.runs && base < w1; base += 64) {  // (wave-uniform trip count)
            const int k = base + lane;
            const long long i = i_lo + k;
            const bool active = k < w1;
            int h = 0, a = 0, nv = 0, hcv = 0, acv = 0;
            double gh = 0.0, ga = 0.0;
            bool hitP = false, hitQ = false, hitR = false;
            if (active) {
                const FusedFixture f = lfx[k];
                h = f.h; a = f.a; nv = f.nv; hcv = f.hc; acv = f.ac;
                const int x = f.x, y = f.y;
                const double wi = (double)f.w;
                const double lh = lrate[2 * k], la = lrate[2 * k + 1];
                gh = x - lh;
                ga = y - la;
                if (x <= 1 && y <= 1) {
                    const double cc = x == 0 ? (y == 0 ? -lh * la : lh) : (y == 0 ? la : -1.0);
                    const double arg = 1.0 + rho * cc;
                    if (arg > 0.0) {
                        Ui += wi * dc::lean::log(arg);
                        const double uu = cc * dc::lean::rcp(arg);
                        ui += wi * uu;
                        if (x == 0) gh += rho * uu;
                        if (y == 0) ga += rho * uu;
                    } else {
                        Ui += wi * log(0.0);  // -inf (tol = 0, bpl/_util.py:42)
                    }
                }
                gh *= wi;
                ga *= wi;
                // arg-extremal fixtures: smallest index among those attaining the maximum
                // (stored as ~0 - i under atomicMax, so the zeroed word means "none")
                hitP = lh * la == M;
                hitQ = lh == Lh;
                hitR = la == La;
            }
            // arg-extremal fixtures: smallest index among those attaining the maximum (stored as ~0 - i under
            // atomicMax, so the zeroed word means "none").  One atomic per wave and maximum, from its lowest
            // lane that attains it: the fixtures are sorted by (venue, home, away), every fixture of the
            // extremal run ties, and a thousand same-address atomics are served one after another at the memory
            // side (the last workgroup spent 15 us in this phase at N = 1e6)
            {
                const unsigned long long bP = __ballot(hitP), bQ = __ballot(hitQ), bR = __ballot(hitR);
                if (bP && lane == __ffsll((long long)bP) - 1) atomicMax(&scu[dcd::SC_IDXP], ~0ull - (unsigned long long)i);
                if (bQ && lane == __ffsll((long long)bQ) - 1) atomicMax(&scu[dcd::SC_IDXQ], ~0ull - (unsigned long long)i);
                if (bR && lane == __ffsll((long long)bR) - 1) atomicMax(&scu[dcd::SC_IDXR], ~0ull - (unsigned long long)i);
            }
            const unsigned long long am = __ballot(active);
            if (am == 0ull) continue;
            // The fixtures are sorted by (venue, home, away[, confederations]): a wave's 64 consecutive ones
            // are one run or a few.  Per run: the sums by DPP, ONE set of LDS atomics from its first lane --
            // 64 lanes adding to the same two teams' words are served one after another by the LDS (the few
            // waves per workgroup that straddle a run boundary cost more than all the others together).
            auto add_adjoints = [&](unsigned long long kk, double sh, double sa) {
                const int h_ = (int)(kk >> 33) & 0xFFFF, a_ = (int)(kk >> 17) & 0xFFFF, hc_ = (int)(kk >> 8) & 0xFF,
                          ac_ = (int)kk & 0xFF;
                const bool nv_ = (kk >> 16) & 1;
                double* Ah = lacc + h_ * dcd::A_N;
                double* Aa = lacc + a_ * dcd::A_N;
                atomicAdd(&Ah[dcd::A_ATT], sh);
                atomicAdd(&Aa[dcd::A_DEF], -sh);
                atomicAdd(&Aa[dcd::A_ATT], sa);
                atomicAdd(&Ah[dcd::A_DEF], -sa);
                if (!nv_) {
                    atomicAdd(&Ah[dcd::A_HATT], sh);
                    atomicAdd(&Aa[dcd::A_ADEF], -sh);
                    atomicAdd(&Aa[dcd::A_AATT], sa);
                    atomicAdd(&Ah[dcd::A_HDEF], -sa);
                }
                if (C) {
                    atomicAdd(&lconf[hc_], sh - sa);
                    atomicAdd(&lconf[ac_], sa - sh);
                }
            };
            auto flush_run = [&]() {   // (wave-uniform)
                if (run_key == ~0ull) return;
                double sh = run_h, sa = run_a;
                dc::wave_sum2_f64(sh, sa);
                if (lane == 0) add_adjoints(run_key, sh, sa);
                run_key = ~0ull;
                run_h = run_a = 0.0;
            };
            const unsigned long long key = ((unsigned long long)h << 33) | ((unsigned long long)a << 17) |
                                           ((unsigned long long)(nv != 0) << 16) | ((unsigned long long)hcv << 8) |
                                           (unsigned long long)acv;
            // (the active lanes are a prefix of the wave: lane 0 is active here)
            const unsigned long long k0 = ((unsigned long long)(unsigned int)__builtin_amdgcn_readfirstlane((int)(key >> 32)) << 32) |
                                          (unsigned int)__builtin_amdgcn_readfirstlane((int)key);
            if (__all(!active || key == k0)) {   // one run in this step
                if (k0 != run_key) {
                    flush_run();
                    run_key = k0;
                }
                run_h += gh;   // (zero on the inactive lanes)
                run_a += ga;
                continue;
            }
            flush_run();
            // a step that straddles runs: per run the sums by DPP and one set of atomics from its first lane (64
            // lanes adding to the same two teams' words are served one after another by the LDS)
            const unsigned long long prev = __shfl_up(key, 1, 64);
            const bool head = active && (lane == 0 || key != prev);
            unsigned long long rest = __ballot(head);
            if (__popcll(rest) <= 8) {
                while (rest) {   // (wave-uniform)
                    const int l0 = __ffsll((long long)rest) - 1;
                    rest &= rest - 1;
                    const int l1 = rest ? __ffsll((long long)rest) - 1 : 64;
                    const bool in = active && lane >= l0 && lane < l1;
                    double sh = in ? gh : 0.0, sa = in ? ga : 0.0;
                    dc::wave_sum2_f64(sh, sa);
                    if (lane == l0) add_adjoints(key, sh, sa);
                }
            } else if (active) {   // short runs (many pairs, few fixtures each): every lane for itself
                add_adjoints(key, gh, ga);
            }
        }
        if (run_key != ~0ull) {   // (wave-uniform) the run this wave was still carrying
            double sh = run_h, sa = run_a;
            dc::wave_sum2_f64(sh, sa);
            if (lane == 0) {
                const unsigned long long kk = run_key;
                const int h_ = (int)(kk >> 33) & 0xFFFF, a_ = (int)(kk >> 17) & 0xFFFF, hc_ = (int)(kk >> 8) & 0xFF,
                          ac_ = (int)kk & 0xFF;
                const bool nv_ = (kk >> 16) & 1;
                double* Ah = lacc + h_ * dcd::A_N;
                double* Aa = lacc + a_ * dcd::A_N;
                atomicAdd(&Ah[dcd::A_ATT], sh);
                atomicAdd(&Aa[dcd::A_DEF], -sh);
                atomicAdd(&Aa[dcd::A_ATT], sa);
                atomicAdd(&Ah[dcd::A_DEF], -sa);
                if (!nv_) {
                    atomicAdd(&Ah[dcd::A_HATT], sh);
                    atomicAdd(&Aa[dcd::A_ADEF], -sh);
                    atomicAdd(&Aa[dcd::A_AATT], sa);
                    atomicAdd(&Ah[dcd::A_HDEF], -sa);
                }
                if (C) {
                    atomicAdd(&lconf[hc_], sh - sa);
                    atomicAdd(&lconf[ac_], sa - sh);
                }
            }
        }
        NEU_STAMP(9);
        double both[2] = {Ui, ui};
        dc::wave_sumN_f64(both);
        if (lane == 0) {
            shr[wave * 2] = both[0];
            shr[wave * 2 + 1] = both[1];
        }
        __syncthreads();
        if (tid < 2) {
            double v = 0.0;
            for (int w = 0; w < WAVES; ++w) v += shr[w * 2 + tid];
            if (v != 0.0) atomicAdd(&gsc[tid == 0 ? dcd::SC_U : dcd::SC_GRHO], v);
        }
        for (int k = tid; k < T * dcd::A_N; k += NEU_BIG_BLOCK) {
            const double v = lacc[k];
            if (v != 0.0) atomicAdd(&gacc[k], v);
        }
        for (int k = tid; k < C; k += NEU_BIG_BLOCK) {
            const double v = lconf[k];
            if (v != 0.0) atomicAdd(&gcacc[k], v);
        }
        // (the maxima travel to the epilogue in LDS: the last workgroup has them like every other)
        if (tid == 0) {
            lred[(size_t)T * dcd::A_N + dcd::SC_MAXP] = M;
            lred[(size_t)T * dcd::A_N + dcd::SC_MAXH] = Lh;
            lred[(size_t)T * dcd::A_N + dcd::SC_MAXA] = La;
        }
    }
    // ---- arrive (atomics drained); the last workgroup runs the epilogue
    NEU_STAMP(5);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) s_last = dcd::tree_arrive_one(F.tickets, dcd::TB_FINAL, blockIdx.x, nb, false);
    __syncthreads();
    if (!s_last) return;
    dcd::tree_reset(F.tickets);   // everyone is past the barrier: the counters go back to zero for the next launch
    NEU_STAMP(6);
    // the copies of the scratch, summed into LDS (every load of a thread in flight at once); the epilogue
    // then reads LDS only.  Words: acc | sc (maxima: put there above; SC_U, SC_GRHO: sums; indices: copy 0) | cacc
    {
        const int o_sc = T * dcd::A_N;
        for (int k = tid; k < (int)F.scratch_n; k += NEU_BIG_BLOCK) {
            const int j = k - o_sc;
            const bool is_max = j >= dcd::SC_MAXP && j <= dcd::SC_MAXA, is_idx = j >= dcd::SC_IDXP && j <= dcd::SC_IDXR;
            double v[NEU_BIG_GROUPS];
#pragma unroll
            for (int g_ = 0; g_ < NEU_BIG_GROUPS; ++g_) v[g_] = dc::ld_sc1(F.acc + (size_t)g_ * F.scratch_n + k);
            double sum = 0.0;
#pragma unroll
            for (int g_ = 0; g_ < NEU_BIG_GROUPS; ++g_) sum += v[g_];
            if (is_idx) lred[k] = v[0];          // (bit patterns of the indices: copy 0 only)
            else if (!is_max) lred[k] = sum;
        }
    }
    __syncthreads();
    NeuArgs E = A;
    E.F.acc = lred;
    E.F.sc = lred + (size_t)T * dcd::A_N;
    E.F.cacc = E.F.sc + dcd::SC_N;
#ifdef DC_STAMPS
    __shared__ unsigned long long epi_st[8];
    epilogue_body<NEU_BIG_BLOCK, false, true>(E, sums, epi_st, lred, s_scal);
#else
    epilogue_body<NEU_BIG_BLOCK, false, true>(E, sums, nullptr, lred, s_scal);
#endif
    NEU_STAMP(7);
#ifdef DC_STAMPS
    __syncthreads();
    if (tid == 0) {
        for (int k = 0; k < 8 && k < T; ++k) F.grad[L.o_sat + k] = (double)nstamp_[k];
        for (int k = 0; k < 5 && 8 + k < T; ++k) F.grad[L.o_sat + 8 + k] = (double)epi_st[k];
        for (int k = 0; k < 2 && 13 + k < T; ++k) F.grad[L.o_sat + 13 + k] = (double)nstamp_[8 + k];
    }
#endif
    // everything this launch accumulated is read: back to zero for the next one (plain stores: the words are
    // next touched by the NEXT launch's atomics, and the kernel boundary writes them back first)
    __syncthreads();
    for (size_t k = tid; k < F.scratch_n * NEU_BIG_GROUPS; k += NEU_BIG_BLOCK) F.acc[k] = 0.0;
}

// ---- the whole evaluation in ONE workgroup (one workgroup per chain), everything in LDS.
// The four-launch path above is bound by its three kernel boundaries and by global round trips
// between dependent steps (a 16 us epilogue that moves 23 KB).  At the sizes the neutral model
// is usually fitted on (hundreds to a few thousand internationals) one CU holds all of it: z,
// the cell records and the per-fixture adjoints live in LDS, six barriers replace the launches.
// Wave 0 is the SCALAR wave (one scalar site per lane, every transcendental of the z side);
// waves 1..15 are the 960 workers (teams, fixtures).
//   A  workers: z -> LDS; cell records, one thread per team, straight from global z (each
//      computes the six exp(std) itself: waiting for the scalar wave would cost a barrier)
//      scalar wave: exp / sigmoid of its sites (what step D needs)
//   C  workers: rates; maxima by DPP, one ds_max per wave
//      scalar wave: the rest of its chain (log1p, log-density, constant gradient parts) -- off
//      the critical path, it is needed from step F on
//   D  bounds; value + adjoint (gh, ga) of every fixture -> LDS
//   E  one 16-lane ROW per slot of (at most) 64 incidence entries of ONE team (host-built
//      schedule, the first round's entries requested at kernel entry): gather the adjoints into
//      six partial sums by four DPP steps inside the row; then fold each team's slots and add
//      the bounds' adjoint.  No atomics (LDS float64 atomics retire about one lane per cycle: 8
//      per fixture took 3 us of the first version's 11), and the sums have a fixed order.
//   F  the sums over teams, one per wave
//   G  team and scalar-site gradients, potential (every global store is after the last
//      barrier: a barrier waits for the stores before it)
// The arg-extremal fixtures travel as ONE packed word per bound (ds_max_u64 on
// {~index, h, a, venue, confederations}): no second look at the fixture columns.
constexpr int FUSED_BLOCK = 1024;
constexpr int FUSED_WAVES = FUSED_BLOCK / 64;
constexpr int FUSED_WORKERS = FUSED_BLOCK - 64;
constexpr int FUSED_ROWS = FUSED_BLOCK / 16; // 16-lane rows: slots per round of step E
constexpr int FUSED_MAX_N = 1 << 13;         // (LDS binds first for most shapes)
constexpr int FUSED_MAX_T = 4095, FUSED_MAX_C = 255;
constexpr int FUSED_PRE = 4;                 // incidence entries per lane and slot (64 per slot)
constexpr int FUSED_CONF_COPIES = 8;         // private copies of the confederation accumulators
constexpr int FUSED_SITES = 13;  // scalar sites: 6 stds, u, corr_coef_raw, 4 means, mean_defence
// fixed LDS words
enum {
    FX_STD = 0,      // [6] exp(std sites): att, def, ha, aa, hd, ad
    FX_Q = 6,        // corr_coef_raw site: clipped sigmoid
    FX_RP = 7, FX_IVV, FX_LOGVV,   // rho' = 2u-1, 1/(1-rho'^2), log(1-rho'^2)
    FX_LP = 10,      // log-density of the scalar sites, coefficients and confederation strengths
    FX_MAX = 12,     // [3] maxima (bit patterns)
    FX_KEY = 15,     // [3] packed arg-extremal fixtures
    FX_MUL = 18,     // [13] gradient of scalar site l: -(mul[l] * dot_l + pre[l])
    FX_PRE = 31,     // [13]
    FX_WSUM = 44,    // [waves][2] per-wave sums over fixtures: value, d/d rho (added in wave order)
    FX_SUMS = 44 + 2 * FUSED_WAVES,  // [NEU_SUMS + 2K]
    FX_N = FX_SUMS
};

// incidence entry of a team: fixture index << 2 | venue neutral << 1 | team is the away side
constexpr uint32_t INC_NONE = 0xFFFFFFFFu;

struct FusedArgs {
    NeuLayout L;
    int n;
    int n_slots;                 // slots of step E (each: <= 64 incidence entries of one team)
    const FusedFixture* fx;      // [n]
    const uint32_t* sched;       // [rounds * FUSED_PRE][FUSED_BLOCK] entry of (round, k, thread) or INC_NONE
    const int* slot_off;         // [T + 1] first slot of each team
    const double* xs;            // [T, K] or nullptr
    double lgsum;
    const double* z;             // [chains, D]
    double* potential;           // [chains]
    double* grad;                // [chains, D]
    double* aux;                 // [chains, 4] or nullptr
    int stop_after;              // diagnostic build only: leave after this step (0: run to the end)
    // NUTS-aware instantiation (one persistent chain per workgroup): z, potential, gradient and aux
    // live in the chain's state block; the scalar wave books the leaf after the gradient
    double* nuts;                // [chains][nuts_stride] or nullptr
    size_t nuts_stride;
    int max_depth;
    nd::Persist persist;
};

__host__ __device__ inline size_t fused_leaf_doubles(const NeuLayout& L) { return (size_t)L.D + 8; }  // grad | potential | aux
__host__ __device__ inline size_t fused_lds_doubles(const NeuLayout& L, long long n, int n_slots) {
    return FX_N + NEU_SUMS + 2 * (size_t)L.K + L.D + (size_t)L.T * dcd::P_N + (size_t)L.T * dcd::A_N +
           (size_t)(FUSED_CONF_COPIES + 1) * L.C + 2 * (size_t)n + (size_t)n_slots * dcd::A_N +
           (L.T + 2) / 2;
}

#ifdef DC_STAMPS  // diagnostic build: the phase timeline (10 ns ticks since entry) replaces grad[0..9]
#define NEU_STAMP_DECL unsigned long long stamp_[10] = {}
#define NEU_STAMP(k)                                      \
    do {                                                  \
        stamp_[k] = __builtin_amdgcn_s_memrealtime();     \
        if (A.stop_after == (k)) return;                  \
    } while (0)
#define NEU_STAMP_FLUSH for (int k_ = 0; k_ < 10; ++k_) grad[k_] = (double)(stamp_[k_] - stamp_[0])
#else
#define NEU_STAMP_DECL do { } while (0)
#define NEU_STAMP(k) do { } while (0)
#define NEU_STAMP_FLUSH do { } while (0)
#endif

template <bool NUTS>
__global__ __launch_bounds__(FUSED_BLOCK) void neu_fused(FusedArgs A) {
    extern __shared__ double lds[];
    NEU_STAMP_DECL;
    NEU_STAMP(0);
    const NeuLayout& L = A.L;
    const int T = L.T, K = L.K, C = L.C, D = L.D, N = A.n;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wid = tid - 64;                          // worker index (waves 1..15), < 0 on the scalar wave
    double* fixed = lds;
    double* sums = fixed + FX_SUMS;                    // [NEU_SUMS + 2K]
    double* zs = sums + NEU_SUMS + 2 * K;              // [D]
    double* cells = zs + D;                            // [T][P_N]
    double* accF = cells + T * dcd::P_N;               // [T][A_N] then [C]: adjoints, bounds included
    double* cacc = accF + T * dcd::A_N + C;            // [FUSED_CONF_COPIES][C]
    double* gb = cacc + FUSED_CONF_COPIES * C;         // [N][2] weighted (gh, ga)
    double* part = gb + 2 * (size_t)N;                 // [n_slots][A_N] partial sums of step E
    int* slot_off = reinterpret_cast<int*>(part + (size_t)A.n_slots * dcd::A_N);  // [T + 1]
    unsigned long long* fixed_u = reinterpret_cast<unsigned long long*>(fixed);

    double* ns = NUTS ? A.nuts + (size_t)blockIdx.x * A.nuts_stride : nullptr;
    if (NUTS && ns[nd::H_S_DONE] != 0.0) return;  // chain finished (uniform over the workgroup)
    const double* z = NUTS ? nd::vec(ns, D, nd::V_ZN) : A.z + (size_t)blockIdx.x * D;
    double* grad = NUTS ? nd::vec(ns, D, nd::V_GRAD) : A.grad + (size_t)blockIdx.x * D;

    // ---- A
    // every global word this thread needs before the first barrier is requested here
    const FusedFixture first = A.fx[wid >= 0 && wid < N ? wid : 0];
    uint32_t pre[FUSED_PRE];
#pragma unroll
    for (int r = 0; r < FUSED_PRE; ++r) pre[r] = A.sched[r * FUSED_BLOCK + tid];
    // scalar wave: its site's latent value (kept across the first barrier)
    //   0..5 stds | 6 u | 7 corr_coef_raw | 8..11 means (home attack, away attack, home defence,
    //   away defence) | 12 mean_defence
    double zv = 0.0, ez = 0.0, sg = 0.0, v_site = 0.0, dv_site = 0.0;
    const bool sigm = lane == 6 || lane == 7;
    if (wave == 0) {
        const int o = lane == 0 ? L.o_s_att : lane == 1 ? L.o_s_def : lane == 2 ? L.o_s_ha
                    : lane == 3 ? L.o_s_aa : lane == 4 ? L.o_s_hd : lane == 5 ? L.o_s_ad
                    : lane == 6 ? L.o_u : lane == 7 ? L.o_corr : lane == 8 ? L.o_mha
                    : lane == 9 ? L.o_maa : lane == 10 ? L.o_mhd : lane == 11 ? L.o_mad : L.o_md;
        zv = z[lane < FUSED_SITES ? o : 0];
        // one exp for lanes 0..7 TOGETHER (a float64 libm call is ~0.3 us of dependent
        // instructions; on diverged lanes they would run one after another)
        ez = dc::lean::exp(sigm ? -fabs(zv) : zv);
        const double s_abs = dc::lean::rcp(1.0 + ez);
        sg = zv >= 0 ? s_abs : 1.0 - s_abs;
        v_site = sg < dc::SIG_LO ? dc::SIG_LO : sg > dc::SIG_HI ? dc::SIG_HI : sg;
        dv_site = (sg < dc::SIG_LO || sg > dc::SIG_HI) ? 0.0 : sg * (1.0 - sg);
        if (lane < 6) fixed[FX_STD + lane] = ez;
        if (lane == 7) fixed[FX_Q] = v_site;
        if (lane >= 8 && lane < 14) fixed_u[FX_MAX + lane - 8] = 0ull;   // maxima and keys
    } else {
        for (int i = wid; i < D; i += FUSED_WORKERS) zs[i] = z[i];
        for (int i = wid; i <= T; i += FUSED_WORKERS) slot_off[i] = A.slot_off[i];
        for (int i = wid; i < FUSED_CONF_COPIES * C; i += FUSED_WORKERS) cacc[i] = 0.0;
        // cell records from global z: the team's six values, the five means and the six stds in
        // one round of loads
        for (int t = wid; t < T; t += FUSED_WORKERS) {
            const double sat = z[L.o_sat + t], sdt = z[L.o_sdt + t], zhat = z[L.o_hat + t],
                         zaat = z[L.o_aat + t], zhdf = z[L.o_hdf + t], zadf = z[L.o_adf + t];
            const double md = z[L.o_md], mha = z[L.o_mha], maa = z[L.o_maa], mhd = z[L.o_mhd], mad = z[L.o_mad];
            const double l_att = z[L.o_s_att], l_def = z[L.o_s_def], l_ha = z[L.o_s_ha], l_aa = z[L.o_s_aa],
                         l_hd = z[L.o_s_hd], l_ad = z[L.o_s_ad];
            double att = 0.0, def = md;
            for (int k = 0; k < K; ++k) {
                const double xv = A.xs[(size_t)t * K + k];
                att += xv * z[L.o_bA + k];
                def += xv * z[L.o_bD + k];
            }
            att += sat * dc::lean::exp(l_att);
            def += sdt * dc::lean::exp(l_def);
            const double hat = mha + dc::lean::exp(l_ha) * zhat;
            const double aat = maa + dc::lean::exp(l_aa) * zaat;
            const double hdf = mhd + dc::lean::exp(l_hd) * zhdf;
            const double adf = mad + dc::lean::exp(l_ad) * zadf;
            double* P = cells + t * dcd::P_N;
            P[dcd::P_AH] = att + hat;
            P[dcd::P_AA] = att + aat;
            P[dcd::P_BH] = def + hdf;
            P[dcd::P_BA] = def + adf;
            P[dcd::P_ATT] = att;
            P[dcd::P_DEF] = def;
        }
    }
    __syncthreads();
    NEU_STAMP(1);
    // ---- C
    auto etas = [&](const FusedFixture& f, double* eh, double* ea) {
        const double* Ph = cells + f.h * dcd::P_N;
        const double* Pa = cells + f.a * dcd::P_N;
        const bool nvf = f.nv != 0;
        *eh = Ph[nvf ? dcd::P_ATT : dcd::P_AH] - Pa[nvf ? dcd::P_DEF : dcd::P_BA];
        *ea = Pa[nvf ? dcd::P_ATT : dcd::P_AA] - Ph[nvf ? dcd::P_DEF : dcd::P_BH];
        if (C) {  // bpl/neutral_dixon_coles_WC.py:188-203
            const double d = zs[L.o_conf + f.hc] - zs[L.o_conf + f.ac];
            *eh += d;
            *ea -= d;
        }
    };
    double eh0 = 0.0, ea0 = 0.0, lh0 = 0.0, la0 = 0.0;  // this worker's first fixture, kept for step D
    if (wave == 0) {
        // the rest of the scalar chain: one log1p for the two sigmoid sites, then per site its
        // log-density and the constant part of its gradient
        const double l1 = dc::lean::log1p_pos(sigm ? ez : 0.0);
        double Lp = 0.0;
        // confederation strengths and coefficients ~ N(0,1): lanes stride over them
        for (int k = lane; k < 2 * K + C; k += 64) {
            const double v = zs[k < K ? L.o_bA + k : k < 2 * K ? L.o_bD + k - K : L.o_conf + k - 2 * K];
            Lp += -0.5 * v * v - HALF_LOG_2PI;
        }
        double mul = 1.0, prec = 0.0;
        if (lane < 6) {  // HalfNormal(scale) in log space: std_attack / std_defence scale 0.5, others 1
            const double r = lane < 2 ? 2.0 * ez : ez;
            Lp += LN2 - (lane < 2 ? -LN2 : 0.0) - HALF_LOG_2PI - 0.5 * r * r + zv;
            mul = ez;
            prec = 1.0 - r * r;
        } else if (sigm) {
            // log v, log(1-v), softplus(z) + softplus(-z) of the (clipped) sigmoid
            const double az = fabs(zv), sp_pos = az + l1;
            double log_v = zv >= 0 ? -l1 : -sp_pos, log_1mv = zv >= 0 ? -sp_pos : -l1;
            if (dv_site == 0.0) {
                log_v = log(v_site);
                log_1mv = log1p(-v_site);
            }
            const double v = v_site;
            if (lane == 6) {  // u ~ Beta(2,4); rho' = 2u - 1, 1 - rho'^2 = 4 u (1 - u)
                const double rp = 2.0 * v - 1.0, vv = 1.0 - rp * rp;
                fixed[FX_RP] = rp;
                fixed[FX_IVV] = dc::lean::rcp(vv);
                fixed[FX_LOGVV] = 2.0 * LN2 + log_v + log_1mv;
                Lp += log_v + 3.0 * log_1mv + 2.995732273553991 - (sp_pos + l1);
                mul = 2.0 * dv_site;
                prec = (dc::lean::rcp(v) - 3.0 * dc::lean::rcp(1.0 - v)) * dv_site + (1.0 - 2.0 * sg);
            } else {          // corr_coef_raw ~ Beta(2,2)
                Lp += log_v + log_1mv + 1.791759469228055 - (sp_pos + l1);
                mul = dv_site;
                prec = (dc::lean::rcp(v) - dc::lean::rcp(1.0 - v)) * dv_site + (1.0 - 2.0 * sg);
            }
        } else if (lane < 12) {  // means ~ N(+-0.1, 0.2)
            const double mu = (lane & 1) ? -0.1 : 0.1, r = (zv - mu) / 0.2;
            Lp += -0.5 * r * r + 1.6094379124341003 - HALF_LOG_2PI;
            prec = -(zv - mu) / 0.04;
        } else if (lane == 12) {  // mean_defence ~ N(0,1)
            Lp += -0.5 * zv * zv - HALF_LOG_2PI;
            prec = -zv;
        }
        if (lane < FUSED_SITES) {
            fixed[FX_MUL + lane] = mul;
            fixed[FX_PRE + lane] = prec;
        }
        Lp = dc::wave_sum_f64(Lp);
        if (lane == 0) {
            fixed[FX_LP] = Lp;
            fixed[FX_WSUM] = 0.0;        // (the scalar wave has no fixtures)
            fixed[FX_WSUM + 1] = 0.0;
        }
    } else {
        double mP = 0.0, mH = 0.0, mA = 0.0;
        for (int i = wid; i < N; i += FUSED_WORKERS) {
            FusedFixture f = first;
            if (i != wid) f = A.fx[i];
            double eh, ea;
            etas(f, &eh, &ea);
            const double lh = dc::lean::exp(eh), la = dc::lean::exp(ea);
            if (i == wid) { eh0 = eh; ea0 = ea; lh0 = lh; la0 = la; }
            mP = fmax(mP, lh * la);
            mH = fmax(mH, lh);
            mA = fmax(mA, la);
        }
        if (wid - lane < N) {  // (waves without fixtures: nothing to offer)
            dc::wave_max3_f64(mP, mH, mA);
            // (positive doubles order like their bit patterns)
            if (lane < 3)
                atomicMax(&fixed_u[FX_MAX + lane],
                          (unsigned long long)__double_as_longlong(lane == 0 ? mP : lane == 1 ? mH : mA));
        }
    }
    __syncthreads();
    NEU_STAMP(2);
    // ---- D
    // (the bounds cost two float64 divisions: only the waves that use them work them out -- four
    // waves share a SIMD, and what all sixteen repeat is paid four times over)
    auto bounds = [&]() {
        dcd::Bounds r;
        r.M = fixed[FX_MAX]; r.Lh = fixed[FX_MAX + 1]; r.La = fixed[FX_MAX + 2];
        r.q = fixed[FX_Q];
        r.UB = r.M > 1.0 ? 1.0 / r.M : 1.0;
        r.LB = -1.0 / fmax(r.Lh, r.La);
        r.rho = r.LB + r.q * (r.UB - r.LB);
        r.G_rho = 0.0;
        return r;
    };
    if (wave != 0 && wid - lane < N) {
        const dcd::Bounds b = bounds();
        double* cmine = cacc + (wave % FUSED_CONF_COPIES) * C;
        double Ui = 0.0, ui = 0.0;
        for (int i = wid; i < N; i += FUSED_WORKERS) {
            FusedFixture f = first;
            double eh = eh0, ea = ea0, lh = lh0, la = la0;
            if (i != wid) {
                f = A.fx[i];
                etas(f, &eh, &ea);
                lh = dc::lean::exp(eh);
                la = dc::lean::exp(ea);
            }
            const double w = (double)f.w;
            double Uf = f.x * eh - lh + f.y * ea - la;
            double gh = f.x - lh, ga = f.y - la;
            if (f.x <= 1 && f.y <= 1) {
                const double c = f.x == 0 ? (f.y == 0 ? -lh * la : lh) : (f.y == 0 ? la : -1.0);
                const double arg = 1.0 + b.rho * c;
                if (arg > 0.0) {
                    Uf += dc::lean::log(arg);
                    const double u = c / arg;
                    ui += w * u;
                    if (f.x == 0) gh += b.rho * u;
                    if (f.y == 0) ga += b.rho * u;
                } else {
                    Uf += log(0.0);  // -inf (tol = 0, bpl/_util.py:42)
                }
            }
            Ui += w * Uf;
            gh *= w;
            ga *= w;
            gb[2 * i] = gh;
            gb[2 * i + 1] = ga;
            // arg-extremal fixtures: the smallest index among those attaining a maximum wins
            const unsigned long long key =
                ((unsigned long long)(0x7FFFFF - i) << 41) | ((unsigned long long)f.h << 29) |
                ((unsigned long long)f.a << 17) | ((unsigned long long)(f.nv != 0) << 16) |
                ((unsigned long long)f.hc << 8) | (unsigned long long)f.ac;
            if (lh * la == b.M) atomicMax(&fixed_u[FX_KEY], key);
            if (lh == b.Lh) atomicMax(&fixed_u[FX_KEY + 1], key);
            if (la == b.La) atomicMax(&fixed_u[FX_KEY + 2], key);
            if (C) {
                atomicAdd(&cmine[f.hc], gh - ga);
                atomicAdd(&cmine[f.ac], ga - gh);
            }
        }
        double both[2] = {Ui, ui};
        dc::wave_sumN_f64(both);
        if (lane == 0) {
            fixed[FX_WSUM + 2 * wave] = both[0];
            fixed[FX_WSUM + 2 * wave + 1] = both[1];
        }
    } else if (wave != 0 && lane == 0) {
        fixed[FX_WSUM + 2 * wave] = 0.0;
        fixed[FX_WSUM + 2 * wave + 1] = 0.0;
    }
    __syncthreads();
    NEU_STAMP(3);
    // ---- E
    {
        // gather: row `tid / 16` takes slots row, row + 64, ...; lane k-th entries 16 apart
        const int sub = lane & 15, row = tid >> 4;
        for (int slot = row, round = 0; slot < A.n_slots; slot += FUSED_ROWS, ++round) {
            uint32_t e[FUSED_PRE];
#pragma unroll
            for (int r = 0; r < FUSED_PRE; ++r)
                e[r] = round == 0 ? pre[r] : A.sched[(size_t)(round * FUSED_PRE + r) * FUSED_BLOCK + tid];
            double s6[dcd::A_N];
#pragma unroll
            for (int j = 0; j < dcd::A_N; ++j) s6[j] = 0.0;
#pragma unroll
            for (int r = 0; r < FUSED_PRE; ++r) {
                const bool have = e[r] != INC_NONE;
                const int i = have ? (int)(e[r] >> 2) : 0;
                const bool away = e[r] & 1, nvf = e[r] & 2;
                const double gh = have ? gb[2 * i] : 0.0, ga = have ? gb[2 * i + 1] : 0.0;
                const double up = away ? ga : gh, down = away ? gh : ga;  // own rate | the opponent's
                const double upv = nvf ? 0.0 : up, downv = nvf ? 0.0 : down;
                s6[dcd::A_ATT] += up;
                s6[dcd::A_DEF] -= down;
                s6[dcd::A_HATT] += away ? 0.0 : upv;
                s6[dcd::A_AATT] += away ? upv : 0.0;
                s6[dcd::A_ADEF] -= away ? downv : 0.0;
                s6[dcd::A_HDEF] -= away ? 0.0 : downv;
            }
            dc::row_sum_f64(s6);
            if (sub < dcd::A_N) {
                double v = s6[0];
#pragma unroll
                for (int j = 1; j < dcd::A_N; ++j) v = sub == j ? s6[j] : v;
                part[slot * dcd::A_N + sub] = v;
            }
        }
    }
    __syncthreads();
    NEU_STAMP(4);
    auto bounds_with_adjoint = [&]() {
        dcd::Bounds r = bounds();
#pragma unroll
        for (int wv = 0; wv < FUSED_WAVES; ++wv) r.G_rho += fixed[FX_WSUM + 2 * wv + 1];
        return r;
    };
    dcd::Bounds b{};  // (wave 0 always takes part here: it keeps its copy for step G)
    if (tid - lane < T * dcd::A_N || tid - lane < C) {
        b = bounds_with_adjoint();
        // the bounds' adjoint reaches (at most) two fixtures: the one with the largest rate
        // product (when it binds, M > 1) through both rates, the one with the largest single
        // rate through that rate
        const bool lb_home = b.Lh >= b.La;
        const unsigned long long k0 = b.M > 1.0 ? fixed_u[FX_KEY] : 0ull;
        const unsigned long long k1 = fixed_u[lb_home ? FX_KEY + 1 : FX_KEY + 2];
        const double vP = b.G_rho * b.q * (-b.UB), vL = b.G_rho * (1.0 - b.q) * (-b.LB);
        auto rate_adjoint = [&](unsigned long long key, bool home_rate, double v, int cell, int which) {
            if (key == 0ull) return 0.0;
            const int fh = (int)(key >> 29) & 0xFFF, fa = (int)(key >> 17) & 0xFFF;
            const bool fnv = (key >> 16) & 1;
            const int fhc = (int)(key >> 8) & 0xFF, fac = (int)key & 0xFF;
            const int up = home_rate ? fh : fa, down = home_rate ? fa : fh;   // attack side / defence side
            if (which == dcd::A_N) {
                const int cu = home_rate ? fhc : fac, cd = home_rate ? fac : fhc;
                return (cell == cu ? v : 0.0) - (cell == cd ? v : 0.0);
            }
            double r = 0.0;
            if (which == dcd::A_ATT && cell == up) r += v;
            if (which == dcd::A_DEF && cell == down) r -= v;
            if (!fnv) {
                if (which == (home_rate ? dcd::A_HATT : dcd::A_AATT) && cell == up) r += v;
                if (which == (home_rate ? dcd::A_ADEF : dcd::A_HDEF) && cell == down) r -= v;
            }
            return r;
        };
        auto bounds_adjoint = [&](int cell, int which) {
            return rate_adjoint(k0, true, vP, cell, which) + rate_adjoint(k0, false, vP, cell, which) +
                   rate_adjoint(k1, lb_home, vL, cell, which);
        };
        for (int e = tid; e < T * dcd::A_N; e += FUSED_BLOCK) {  // fold each team's slots, in slot order
            const int t = e / dcd::A_N, j = e - t * dcd::A_N;
            double v = 0.0;
            for (int sl = slot_off[t]; sl < slot_off[t + 1]; ++sl) v += part[sl * dcd::A_N + j];
            accF[e] = v + bounds_adjoint(t, j);
        }
        for (int cf = tid; cf < C; cf += FUSED_BLOCK) {
            double v = 0.0;
            for (int cpy = 0; cpy < FUSED_CONF_COPIES; ++cpy) v += cacc[cpy * C + cf];
            accF[T * dcd::A_N + cf] = v + bounds_adjoint(cf, dcd::A_N);
        }
    }
    __syncthreads();
    NEU_STAMP(5);
    // ---- F: the NEU_SUMS + 2K sums over teams, one per wave
    const double rp = fixed[FX_RP], ivv = fixed[FX_IVV];
    {
        const double log_vv = fixed[FX_LOGVV];
        for (int s = wave; s < NEU_SUMS + 2 * K; s += FUSED_WAVES) {
            double v = 0.0;
            for (int t = lane; t < T; t += 64) {
                const double* G6 = accF + t * dcd::A_N;
                const double sa = zs[L.o_sat + t], sd = zs[L.o_sdt + t];
                const double e = sd - rp * sa;
                double term;
                switch (s) {
                    case 0: term = e * sa * ivv - rp * e * e * (ivv * ivv) + rp * ivv; break;
                    case 1: term = sa * G6[dcd::A_ATT]; break;
                    case 2: term = sd * G6[dcd::A_DEF]; break;
                    case 3: term = G6[dcd::A_DEF]; break;
                    case 4: term = G6[dcd::A_HATT]; break;
                    case 5: term = G6[dcd::A_AATT]; break;
                    case 6: term = G6[dcd::A_HDEF]; break;
                    case 7: term = G6[dcd::A_ADEF]; break;
                    case 8: term = zs[L.o_hat + t] * G6[dcd::A_HATT]; break;
                    case 9: term = zs[L.o_aat + t] * G6[dcd::A_AATT]; break;
                    case 10: term = zs[L.o_hdf + t] * G6[dcd::A_HDEF]; break;
                    case 11: term = zs[L.o_adf + t] * G6[dcd::A_ADEF]; break;
                    case 12: {
                        const double hat = zs[L.o_hat + t], aat = zs[L.o_aat + t], hdf = zs[L.o_hdf + t],
                                     adf = zs[L.o_adf + t];
                        term = -0.5 * sa * sa - HALF_LOG_2PI - 0.5 * e * e * ivv - 0.5 * log_vv - HALF_LOG_2PI -
                               0.5 * (hat * hat + aat * aat + hdf * hdf + adf * adf) - 4.0 * HALF_LOG_2PI;
                        break;
                    }
                    default: {
                        const int k = s - NEU_SUMS;
                        term = k < K ? A.xs[(size_t)t * K + k] * G6[dcd::A_ATT]
                                     : A.xs[(size_t)t * K + k - K] * G6[dcd::A_DEF];
                    }
                }
                v += term;
            }
            v = dc::wave_sum_f64(v);
            if (lane == 0) sums[s] = v;
        }
    }
    __syncthreads();
    NEU_STAMP(6);
    // ---- G: gradients and the potential.  NUTS: every entry also goes to the leaf's LDS strip (the leaf
    // would otherwise read the gradient back through L2), and the scalar wave requests the chain's
    // state before its own entries (the loads are in flight while the gradient is written)
    double* gL = lds + fused_lds_doubles(L, N, A.n_slots);   // NUTS: grad[D] | potential | aux[4]
    auto put = [&](int o, double v) {
        grad[o] = v;
        if (NUTS) gL[o] = v;
    };
    nd::LeafState<nd::LEAF_NE_MAX> leaf{};
    if (NUTS && wave == 0) leaf = nd::leaf_prefetch<nd::LEAF_NE_MAX>(ns, D, A.max_depth, lane);
    if (wave != 0) {
        const double s_att = fixed[FX_STD], s_def = fixed[FX_STD + 1], s_ha = fixed[FX_STD + 2],
                     s_aa = fixed[FX_STD + 3], s_hd = fixed[FX_STD + 4], s_ad = fixed[FX_STD + 5];
        for (int t = wid; t < T; t += FUSED_WORKERS) {
            const double* G6 = accF + t * dcd::A_N;
            const double sa = zs[L.o_sat + t], sd = zs[L.o_sdt + t];
            const double e = sd - rp * sa;
            put(L.o_sat + t, -(s_att * G6[dcd::A_ATT] - sa + rp * e * ivv));
            put(L.o_sdt + t, -(s_def * G6[dcd::A_DEF] - e * ivv));
            put(L.o_hat + t, -(s_ha * G6[dcd::A_HATT] - zs[L.o_hat + t]));
            put(L.o_aat + t, -(s_aa * G6[dcd::A_AATT] - zs[L.o_aat + t]));
            put(L.o_hdf + t, -(s_hd * G6[dcd::A_HDEF] - zs[L.o_hdf + t]));
            put(L.o_adf + t, -(s_ad * G6[dcd::A_ADEF] - zs[L.o_adf + t]));
        }
        for (int k = wid; k < 2 * K; k += FUSED_WORKERS) {  // covariate coefficients ~ N(0,1)
            const int o = k < K ? L.o_bA + k : L.o_bD + k - K;
            put(o, -(sums[NEU_SUMS + k] - zs[o]));
        }
        for (int cf = wid; cf < C; cf += FUSED_WORKERS)  // confederation strengths ~ N(0,1)
            put(L.o_conf + cf, -(accF[T * dcd::A_N + cf] - zs[L.o_conf + cf]));
    } else if (lane < FUSED_SITES) {
        // the sum each scalar site's gradient takes (sites in the lane order of step A)
        const int o = lane == 0 ? L.o_s_att : lane == 1 ? L.o_s_def : lane == 2 ? L.o_s_ha
                    : lane == 3 ? L.o_s_aa : lane == 4 ? L.o_s_hd : lane == 5 ? L.o_s_ad
                    : lane == 6 ? L.o_u : lane == 7 ? L.o_corr : lane == 8 ? L.o_mha
                    : lane == 9 ? L.o_maa : lane == 10 ? L.o_mhd : lane == 11 ? L.o_mad : L.o_md;
        const int si = lane == 0 ? 1 : lane == 1 ? 2 : lane < 6 ? 6 + lane : lane == 6 ? 0
                     : lane < 12 ? lane - 4 : 3;                        // (lane 7 takes G_rho instead)
        const double dot = lane == 7 ? b.G_rho * (b.UB - b.LB) : sums[si];
        put(o, -(fixed[FX_MUL + lane] * dot + fixed[FX_PRE + lane]));
        if (lane == 0) {
            double U = 0.0;
#pragma unroll
            for (int wv = 0; wv < FUSED_WAVES; ++wv) U += fixed[FX_WSUM + 2 * wv];
            const double pot = -(sums[12] + U - A.lgsum + fixed[FX_LP]);
            if (NUTS) {
                ns[nd::H_LEAF_PE] = pot;
                ns[nd::H_LEAF_AUX0] = b.rho; ns[nd::H_LEAF_AUX1] = b.LB;
                ns[nd::H_LEAF_AUX2] = b.UB;  ns[nd::H_LEAF_AUX3] = b.q;
                gL[D] = pot;
                gL[D + 1] = b.rho; gL[D + 2] = b.LB; gL[D + 3] = b.UB; gL[D + 4] = b.q;
            } else {
                A.potential[blockIdx.x] = pot;
            }
            NEU_STAMP(7);
            NEU_STAMP_FLUSH;
            if (!NUTS && A.aux) {
                double* aux = A.aux + (size_t)blockIdx.x * 4;
                aux[0] = b.rho;
                aux[1] = b.LB;
                aux[2] = b.UB;
                aux[3] = b.q;
            }
        }
    }
    if (NUTS) {
        // ---- the leapfrog's bookkeeping (nuts_dev.hip.h: what kp_leaf does as a launch of its own),
        // on the scalar wave, from the LDS strip, once every wave has written its entries (the barrier
        // also waits for the global stores: the chain's next steps read the gradient from its state block)
        __syncthreads();
        if (wave == 0) {
            nd::leaf_prepare<true>(leaf);
            const bool sub_done = nd::nuts_leaf(ns, D, A.max_depth, lane, gL, leaf);
            if (sub_done) nd::persist_advance(ns, A.persist, blockIdx.x, lane);
        }
    }
}

}  // namespace dcn
