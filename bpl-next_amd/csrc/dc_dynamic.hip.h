// dc_dynamic.hip.h -- gfx950 kernels for the dynamic (time-varying, neutral-venue)
// Dixon-Coles model: bpl/dynamic_dixon_coles.py:63-247 with the INTENDED random walk
// (attack[g] = attack[g-1] + standardised_attack[g]*std_attack[g]; the reference's
// `.at[].set()` results are discarded) or, `random_walk = 0`,
// the behaviour of the code as written (attack = defence = 0).
//
// First correct path for BASELINE config 4 (T = 100, G = 50, D = 35 502): float64
// throughout, four small launches per evaluation, no fixture re-ordering:
//   dyn_cells     per-(gameweek, team) constrained sites from z (walk = cumulative sum)
//   dyn_pass1     per fixture: rates; global maxima for the rho bounds (atomicMax)
//   dyn_pass2     per fixture: Poisson + tau value and adjoint, float64 atomics into the
//                 six per-cell accumulators; arg-extremal fixtures (atomicMin of index)
//   dyn_epilogue  bounds adjoint, reverse cumulative sum over gameweeks (adjoint of the
//                 walk), priors + Jacobians, chain rule to z
// Roofline: HBM-bound stream of 9 B per fixture (u16,u16,u8,u8,u16,u8) + gathers from an
// L2-resident cell table; at config-4 size (N = 2500) it is launch-latency bound.
// Mathematics: SURVEY.md Appendix A.5 (+ Appendix A.1-A.3 for the shared pieces).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "dc_layout.h"

namespace dcd {

using dc::HALF_LOG_2PI;
using dc::LN2;

struct DynLayout {
    int G, T, K, D;
    int o_bA, o_aat, o_adf, o_corr, o_bD, o_hat, o_hdf, o_maa, o_mad, o_md, o_mha, o_mhd, o_sat,
        o_sdt, o_s_att, o_s_aa, o_s_ad, o_s_def, o_s_ha, o_s_hd, o_u;
};

inline DynLayout make_dyn_layout(int G, int T, int K) {
    DynLayout L{};
    L.G = G; L.T = T; L.K = K;
    int o = 0;
    const int GT = G * T;
    L.o_bA = o; o += K;
    L.o_aat = o; o += GT;
    L.o_adf = o; o += GT;
    L.o_corr = o; o += 1;
    L.o_bD = o; o += K;
    L.o_hat = o; o += GT;
    L.o_hdf = o; o += GT;
    L.o_maa = o; o += G;
    L.o_mad = o; o += G;
    L.o_md = o; o += 1;
    L.o_mha = o; o += G;
    L.o_mhd = o; o += G;
    L.o_sat = o; o += GT;
    L.o_sdt = o; o += GT;
    L.o_s_att = o; o += G;
    L.o_s_aa = o; o += G;
    L.o_s_ad = o; o += G;
    L.o_s_def = o; o += G;
    L.o_s_ha = o; o += G;
    L.o_s_hd = o; o += G;
    L.o_u = o; o += GT;
    L.D = o;
    return L;
}

// scratch scalars (doubles / u64 words), zeroed by a memset before every evaluation
enum { SC_U = 0, SC_GRHO, SC_MAXP, SC_MAXH, SC_MAXA, SC_IDXP, SC_IDXQ, SC_IDXR, SC_N = 8 };
// cell parameter record
enum { P_AH = 0, P_AA, P_BH, P_BA, P_ATT, P_DEF, P_N = 6 };
// cell accumulator record
enum { A_ATT = 0, A_DEF, A_HATT, A_ADEF, A_AATT, A_HDEF, A_N = 6 };

struct DynArgs {
    const uint16_t* h;
    const uint16_t* a;
    const uint8_t* x;
    const uint8_t* y;
    const uint16_t* gw;
    const uint8_t* nv;
    long long n;
    const double* xs;   // [T,K] standardised covariates or nullptr
    double lgsum;
    double* cells;      // [G*T][P_N]
    double* acc;        // [G*T][A_N]   (zeroed per evaluation)
    double* lam;        // [2][n]
    double* sc;         // [SC_N]       (zeroed per evaluation; maxima/indices as u64 bits)
    double* hyp;        // [6][G] exp(std_*)  order: att, def, ha, aa, hd, ad
    const double* z;
    double* potential;
    double* grad;
    double* aux;
    int random_walk;
    DynLayout L;
};

__device__ __forceinline__ double sig(double x) {
    return x >= 0 ? 1.0 / (1.0 + exp(-x)) : 1.0 - 1.0 / (1.0 + exp(x));
}
__device__ __forceinline__ double softplus(double x) {
    return fmax(x, 0.0) + log1p(exp(-fabs(x)));
}
__device__ __forceinline__ void clipped_sig(double x, double* v, double* dv, double* s) {
    const double t = sig(x);
    *s = t;
    if (t < dc::SIG_LO) { *v = dc::SIG_LO; *dv = 0.0; }
    else if (t > dc::SIG_HI) { *v = dc::SIG_HI; *dv = 0.0; }
    else { *v = t; *dv = t * (1.0 - t); }
}

// ---- per-(gameweek, team) constrained sites
__global__ __launch_bounds__(256) void dyn_cells(DynArgs A) {
    const DynLayout& L = A.L;
    const int G = L.G, T = L.T, K = L.K;
    const double* z = A.z;
    extern __shared__ double sh[];  // [6*G] exp(std)
    for (int i = threadIdx.x; i < 6 * G; i += blockDim.x) {
        const int j = i / G, g = i - j * G;
        const int o = j == 0 ? L.o_s_att : j == 1 ? L.o_s_def : j == 2 ? L.o_s_ha
                    : j == 3 ? L.o_s_aa : j == 4 ? L.o_s_hd : L.o_s_ad;
        const double v = exp(z[o + g]);
        sh[i] = v;
        if (blockIdx.x == 0) A.hyp[i] = v;
    }
    __syncthreads();
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= T) return;
    double att = 0.0, def = z[L.o_md];
    for (int k = 0; k < K; ++k) {
        const double xv = A.xs[(size_t)t * K + k];
        att += xv * z[L.o_bA + k];
        def += xv * z[L.o_bD + k];
    }
    for (int g = 0; g < G; ++g) {
        const int c = g * T + t;
        double a_ = 0.0, d_ = 0.0;
        if (A.random_walk) {
            att += z[L.o_sat + c] * sh[g];
            def += z[L.o_sdt + c] * sh[G + g];
            a_ = att;
            d_ = def;
        }
        const double hat = z[L.o_mha + g] + sh[2 * G + g] * z[L.o_hat + c];
        const double aat = z[L.o_maa + g] + sh[3 * G + g] * z[L.o_aat + c];
        const double hdf = z[L.o_mhd + g] + sh[4 * G + g] * z[L.o_hdf + c];
        const double adf = z[L.o_mad + g] + sh[5 * G + g] * z[L.o_adf + c];
        double* P = A.cells + (size_t)c * P_N;
        P[P_AH] = a_ + hat;
        P[P_AA] = a_ + aat;
        P[P_BH] = d_ + hdf;
        P[P_BA] = d_ + adf;
        P[P_ATT] = a_;
        P[P_DEF] = d_;
    }
}

__device__ __forceinline__ void fixture_etas(const DynArgs& A, long long i, int* ch, int* ca,
                                             int* neutral, double* eh, double* ea) {
    const int T = A.L.T;
    const int g = A.gw[i], h = A.h[i], a = A.a[i];
    *ch = g * T + h;
    *ca = g * T + a;
    *neutral = A.nv[i];
    const double* Ph = A.cells + (size_t)(*ch) * P_N;
    const double* Pa = A.cells + (size_t)(*ca) * P_N;
    if (*neutral) {
        *eh = Ph[P_ATT] - Pa[P_DEF];
        *ea = Pa[P_ATT] - Ph[P_DEF];
    } else {
        *eh = Ph[P_AH] - Pa[P_BA];
        *ea = Pa[P_AA] - Ph[P_BH];
    }
}

__device__ __forceinline__ unsigned long long wave_max_u64(unsigned long long v) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        const unsigned long long o = __shfl_xor(v, d, 64);
        v = o > v ? o : v;
    }
    return v;
}
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d, 64);
    return v;
}

// ---- pass 1: rates + maxima (positive doubles order like their bit patterns)
__global__ __launch_bounds__(256) void dyn_pass1(DynArgs A) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    unsigned long long mP = 0, mH = 0, mA = 0;
    if (i < A.n) {
        int ch, ca, nv;
        double eh, ea;
        fixture_etas(A, i, &ch, &ca, &nv, &eh, &ea);
        const double lh = exp(eh), la = exp(ea);
        A.lam[i] = lh;
        A.lam[A.n + i] = la;
        mP = (unsigned long long)__double_as_longlong(lh * la);
        mH = (unsigned long long)__double_as_longlong(lh);
        mA = (unsigned long long)__double_as_longlong(la);
    }
    mP = wave_max_u64(mP);
    mH = wave_max_u64(mH);
    mA = wave_max_u64(mA);
    if ((threadIdx.x & 63) == 0) {
        unsigned long long* sc = reinterpret_cast<unsigned long long*>(A.sc);
        atomicMax(&sc[SC_MAXP], mP);
        atomicMax(&sc[SC_MAXH], mH);
        atomicMax(&sc[SC_MAXA], mA);
    }
}

// ---- pass 2: value + adjoint per fixture
__global__ __launch_bounds__(256) void dyn_pass2(DynArgs A) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const unsigned long long* scu = reinterpret_cast<const unsigned long long*>(A.sc);
    const double M = __longlong_as_double((long long)scu[SC_MAXP]);
    const double Lh = __longlong_as_double((long long)scu[SC_MAXH]);
    const double La = __longlong_as_double((long long)scu[SC_MAXA]);
    double q, dq, s;
    clipped_sig(A.z[A.L.o_corr], &q, &dq, &s);
    const double UB = M > 1.0 ? 1.0 / M : 1.0;
    const double LB = -1.0 / fmax(Lh, La);
    const double rho = LB + q * (UB - LB);
    double Ui = 0.0, ui = 0.0;
    if (i < A.n) {
        int ch, ca, nv;
        double eh, ea;
        fixture_etas(A, i, &ch, &ca, &nv, &eh, &ea);
        const double lh = A.lam[i], la = A.lam[A.n + i];
        const int x = A.x[i], y = A.y[i];
        Ui = x * eh - lh + y * ea - la;
        double gh = x - lh, ga = y - la;
        if (x <= 1 && y <= 1) {
            const double c = x == 0 ? (y == 0 ? -lh * la : lh) : (y == 0 ? la : -1.0);
            const double arg = 1.0 + rho * c;
            if (arg > 0.0) {
                Ui += log(arg);
                ui = c / arg;
                if (x == 0) gh += rho * ui;
                if (y == 0) ga += rho * ui;
            } else {
                Ui += log(0.0);  // -inf (tol = 0, bpl/_util.py:42)
            }
        }
        double* Ah = A.acc + (size_t)ch * A_N;
        double* Aa = A.acc + (size_t)ca * A_N;
        atomicAdd(&Ah[A_ATT], gh);
        atomicAdd(&Aa[A_DEF], -gh);
        atomicAdd(&Aa[A_ATT], ga);
        atomicAdd(&Ah[A_DEF], -ga);
        if (!nv) {
            atomicAdd(&Ah[A_HATT], gh);
            atomicAdd(&Aa[A_ADEF], -gh);
            atomicAdd(&Aa[A_AATT], ga);
            atomicAdd(&Ah[A_HDEF], -ga);
        }
        unsigned long long* sc = reinterpret_cast<unsigned long long*>(A.sc);
        // arg-extremal fixtures: smallest index among those attaining the maximum
        // (stored as ~0 - i under atomicMax, so the zeroed word means "none")
        if (lh * la == M) atomicMax(&sc[SC_IDXP], ~0ull - (unsigned long long)i);
        if (lh == Lh) atomicMax(&sc[SC_IDXQ], ~0ull - (unsigned long long)i);
        if (la == La) atomicMax(&sc[SC_IDXR], ~0ull - (unsigned long long)i);
    }
    Ui = wave_sum(Ui);
    ui = wave_sum(ui);
    if ((threadIdx.x & 63) == 0) {
        atomicAdd(&A.sc[SC_U], Ui);
        atomicAdd(&A.sc[SC_GRHO], ui);
    }
}

// ---- epilogue: one workgroup
constexpr int EPI_THREADS = 1024;

__global__ __launch_bounds__(EPI_THREADS) void dyn_epilogue(DynArgs A) {
    const DynLayout& L = A.L;
    const int G = L.G, T = L.T, K = L.K;
    const int tid = threadIdx.x;
    const double* z = A.z;
    double* grad = A.grad;
    extern __shared__ double sh[];
    double* hyp = sh;                  // [6*G] exp(std): att, def, ha, aa, hd, ad
    double* gsum = hyp + 6 * G;        // [10*G] per-gameweek sums
    double* red = gsum + 10 * G;       // [32] scalar accumulators
    double* cpl = red + 32;            // [8*3] coupling entries {cell, which, value}
    double* covA = cpl + 24;           // [2*K]
    for (int i = tid; i < 6 * G; i += EPI_THREADS) hyp[i] = A.hyp[i];
    for (int i = tid; i < 10 * G + 32 + 24 + 2 * K; i += EPI_THREADS) gsum[i] = 0.0;
    __syncthreads();

    const unsigned long long* scu = reinterpret_cast<const unsigned long long*>(A.sc);
    const double M = __longlong_as_double((long long)scu[SC_MAXP]);
    const double Lh = __longlong_as_double((long long)scu[SC_MAXH]);
    const double La = __longlong_as_double((long long)scu[SC_MAXA]);
    const double zc = z[L.o_corr];
    double q, dq, sq;
    clipped_sig(zc, &q, &dq, &sq);
    const double UB = M > 1.0 ? 1.0 / M : 1.0;
    const double LB = -1.0 / fmax(Lh, La);
    const double rho = LB + q * (UB - LB);
    const double G_rho = A.sc[SC_GRHO];

    // ---- adjoint of the bounds: up to 3 fixtures get an extra d/d eta; recorded as
    // {cell, accumulator, value} entries that the team loop adds while it reads acc
    if (tid == 0) {
        int ne = 0;
        auto add = [&](long long idx1, bool home_rate, double v) {
            if (idx1 == 0) return;
            const long long i = idx1 - 1;
            const int g = A.gw[i], h = A.h[i], a = A.a[i], nv = A.nv[i];
            const int ch = g * T + h, ca = g * T + a;
            auto put = [&](int cell, int which, double val) {
                cpl[3 * ne] = (double)cell; cpl[3 * ne + 1] = (double)which; cpl[3 * ne + 2] = val;
                ++ne;
            };
            if (home_rate) {  // d/d eta_h
                put(ch, A_ATT, v); put(ca, A_DEF, -v);
                if (!nv) { put(ch, A_HATT, v); put(ca, A_ADEF, -v); }
            } else {
                put(ca, A_ATT, v); put(ch, A_DEF, -v);
                if (!nv) { put(ca, A_AATT, v); put(ch, A_HDEF, -v); }
            }
        };
        // at most 2 + 1 fixtures x 4 entries; the table holds 8: P uses both rates (8
        // entries) -> apply Q/R through a second small table region
        if (M > 1.0) {
            const double v = G_rho * q * (-UB);
            const long long ip = scu[SC_IDXP] ? (long long)(~0ull - scu[SC_IDXP]) + 1 : 0;
            add(ip, true, v);
            add(ip, false, v);
        }
        red[20] = (double)ne;  // entries so far live in cpl[0 .. 3*ne)
    }
    __syncthreads();
    // second coupling group (LB) kept in registers of every thread: one fixture, one rate
    long long lbi = 0;
    bool lb_home = false;
    const double lbv = G_rho * (1.0 - q) * (-LB);
    {
        const unsigned long long w = Lh >= La ? scu[SC_IDXQ] : scu[SC_IDXR];
        lbi = w ? (long long)(~0ull - w) + 1 : 0;
        lb_home = Lh >= La;
    }
    int lb_ch = -1, lb_ca = -1, lb_nv = 0;
    if (lbi > 0) {
        const long long i = lbi - 1;
        lb_ch = A.gw[i] * T + A.h[i];
        lb_ca = A.gw[i] * T + A.a[i];
        lb_nv = A.nv[i];
    }
    const int ne = (int)red[20];
    auto coupled = [&](int cell, int which, double base) {
        double v = base;
        for (int e = 0; e < ne; ++e)
            if ((int)cpl[3 * e] == cell && (int)cpl[3 * e + 1] == which) v += cpl[3 * e + 2];
        if (lbi > 0) {
            if (lb_home) {
                if (cell == lb_ch && which == A_ATT) v += lbv;
                if (cell == lb_ca && which == A_DEF) v -= lbv;
                if (!lb_nv && cell == lb_ch && which == A_HATT) v += lbv;
                if (!lb_nv && cell == lb_ca && which == A_ADEF) v -= lbv;
            } else {
                if (cell == lb_ca && which == A_ATT) v += lbv;
                if (cell == lb_ch && which == A_DEF) v -= lbv;
                if (!lb_nv && cell == lb_ca && which == A_AATT) v += lbv;
                if (!lb_nv && cell == lb_ch && which == A_HDEF) v -= lbv;
            }
        }
        return v;
    };

    // ---- per team: reverse cumulative sums over gameweeks, per-cell gradients
    double Lloc = 0.0;
    for (int t = tid; t < T; t += EPI_THREADS) {
        double RA = 0.0, RD = 0.0;
        for (int g = G - 1; g >= 0; --g) {
            const int c = g * T + t;
            const double* Ac = A.acc + (size_t)c * A_N;
            const double ga_ = A.random_walk ? coupled(c, A_ATT, Ac[A_ATT]) : 0.0;
            const double gd_ = A.random_walk ? coupled(c, A_DEF, Ac[A_DEF]) : 0.0;
            const double g_hat = coupled(c, A_HATT, Ac[A_HATT]);
            const double g_adf = coupled(c, A_ADEF, Ac[A_ADEF]);
            const double g_aat = coupled(c, A_AATT, Ac[A_AATT]);
            const double g_hdf = coupled(c, A_HDEF, Ac[A_HDEF]);
            RA += ga_;
            RD += gd_;
            const double s_att = hyp[g], s_def = hyp[G + g], s_ha = hyp[2 * G + g],
                         s_aa = hyp[3 * G + g], s_hd = hyp[4 * G + g], s_ad = hyp[5 * G + g];
            const double sa = z[L.o_sat + c], sd = z[L.o_sdt + c];
            const double zu = z[L.o_u + c];
            double u, du, su;
            clipped_sig(zu, &u, &du, &su);
            const double rp = 2.0 * u - 1.0, vv = 1.0 - rp * rp, e = sd - rp * sa;
            grad[L.o_sat + c] = -(s_att * RA - sa + rp * e / vv);
            grad[L.o_sdt + c] = -(s_def * RD - e / vv);
            const double dL_drp = e * sa / vv - rp * e * e / (vv * vv) + rp / vv;
            grad[L.o_u + c] = -((1.0 / u - 3.0 / (1.0 - u)) * du + 2.0 * dL_drp * du +
                                (1.0 - 2.0 * su));
            const double hat = z[L.o_hat + c], aat = z[L.o_aat + c], hdf = z[L.o_hdf + c],
                         adf = z[L.o_adf + c];
            grad[L.o_hat + c] = -(s_ha * g_hat - hat);
            grad[L.o_aat + c] = -(s_aa * g_aat - aat);
            grad[L.o_hdf + c] = -(s_hd * g_hdf - hdf);
            grad[L.o_adf + c] = -(s_ad * g_adf - adf);
            // per-gameweek sums: 0 sa*RA, 1 sd*RD, 2..5 sum G_x, 6..9 sum dec*G_x
            atomicAdd(&gsum[0 * G + g], sa * RA);
            atomicAdd(&gsum[1 * G + g], sd * RD);
            atomicAdd(&gsum[2 * G + g], g_hat);
            atomicAdd(&gsum[3 * G + g], g_aat);
            atomicAdd(&gsum[4 * G + g], g_hdf);
            atomicAdd(&gsum[5 * G + g], g_adf);
            atomicAdd(&gsum[6 * G + g], hat * g_hat);
            atomicAdd(&gsum[7 * G + g], aat * g_aat);
            atomicAdd(&gsum[8 * G + g], hdf * g_hdf);
            atomicAdd(&gsum[9 * G + g], adf * g_adf);
            // priors of the cell sites
            Lloc += log(u) + 3.0 * log1p(-u) + 2.995732273553991 - softplus(zu) - softplus(-zu);
            Lloc += -0.5 * sa * sa - HALF_LOG_2PI - 0.5 * e * e / vv - 0.5 * log(vv) - HALF_LOG_2PI;
            Lloc += -0.5 * (hat * hat + aat * aat + hdf * hdf + adf * adf) - 4.0 * HALF_LOG_2PI;
        }
        // RA, RD now hold the sums over all gameweeks: d/d(prior means of the walk)
        atomicAdd(&red[0], RD);  // d/d mean_defence
        for (int k = 0; k < K; ++k) {
            atomicAdd(&covA[k], A.xs[(size_t)t * K + k] * RA);
            atomicAdd(&covA[K + k], A.xs[(size_t)t * K + k] * RD);
        }
    }
    atomicAdd(&red[1], Lloc);
    __syncthreads();

    // ---- per gameweek: hyper-parameter gradients and priors
    double Lg = 0.0;
    for (int g = tid; g < G; g += EPI_THREADS) {
        const double s_att = hyp[g], s_def = hyp[G + g];
        grad[L.o_s_att + g] = -(s_att * gsum[0 * G + g] + 1.0 - s_att * s_att);
        grad[L.o_s_def + g] = -(s_def * gsum[1 * G + g] + 1.0 - s_def * s_def);
        Lg += -0.5 * s_att * s_att - HALF_LOG_2PI + LN2 + z[L.o_s_att + g];
        Lg += -0.5 * s_def * s_def - HALF_LOG_2PI + LN2 + z[L.o_s_def + g];
        const int o_mean[4] = {L.o_mha, L.o_maa, L.o_mhd, L.o_mad};
        const int o_std[4] = {L.o_s_ha, L.o_s_aa, L.o_s_hd, L.o_s_ad};
        const double mu[4] = {0.1, -0.1, 0.1, -0.1};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const double s = hyp[(2 + j) * G + g], mean = z[o_mean[j] + g];
            grad[o_mean[j] + g] = -(gsum[(2 + j) * G + g] - (mean - mu[j]) / 0.04);
            grad[o_std[j] + g] = -(s * gsum[(6 + j) * G + g] + 1.0 - s * s);
            const double r = (mean - mu[j]) / 0.2;
            Lg += -0.5 * r * r + 1.6094379124341003 - HALF_LOG_2PI;
            Lg += -0.5 * s * s - HALF_LOG_2PI + LN2 + z[o_std[j] + g];
        }
    }
    atomicAdd(&red[1], Lg);
    __syncthreads();
    if (tid < 2 * K) {
        const int o = tid < K ? L.o_bA + tid : L.o_bD + tid - K;
        grad[o] = -(covA[tid] - z[o]);
        atomicAdd(&red[1], -0.5 * z[o] * z[o] - HALF_LOG_2PI);
    }
    __syncthreads();
    if (tid == 0) {
        const double m = z[L.o_md];
        grad[L.o_md] = -(red[0] - m);
        grad[L.o_corr] = -(G_rho * (UB - LB) * dq + (1.0 - 2.0 * sq));
        double Ltot = red[1] + A.sc[SC_U] - A.lgsum;
        Ltot += -0.5 * m * m - HALF_LOG_2PI;
        Ltot += -softplus(zc) - softplus(-zc);  // Uniform(0,1): log_prob 0 + sigmoid Jacobian
        A.potential[0] = -Ltot;
        if (A.aux) {
            A.aux[0] = rho;
            A.aux[1] = LB;
            A.aux[2] = UB;
            A.aux[3] = q;
        }
    }
}

inline size_t epi_lds_bytes(int G, int K) { return (size_t)(16 * G + 32 + 24 + 2 * K) * 8; }

}  // namespace dcd
