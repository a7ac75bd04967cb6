// dc_kernels.hip.h -- gfx950 (CDNA4, wave64) kernels for the Dixon-Coles potential
// energy and its gradient.  No MFMA: the path is a stream + gather from an
// LDS-resident per-team table + reduction (HBM/latency bound).
//
//   dc_stream   : one pass over the fixture SoA (u16,u16,u8,u8[,f32] = 6 or 10 B per
//                 fixture, 16-B / 8-B vector loads, 8 fixtures per lane).  Per
//                 workgroup: per-team exp tables and the rho bounds are rebuilt in LDS
//                 from z, fixtures are streamed, per-(home,away) run sums are reduced
//                 in-lane -> across the wave (shuffles) -> into LDS per-team
//                 accumulators -> one slab of partial sums per workgroup.
//   dc_epilogue : one workgroup per chain; fixed-order reduction of the slabs, the
//                 rho-bounds adjoint, priors + Jacobians and the chain rule back to the
//                 unconstrained latent vector (float64).
//
// Mathematics: SURVEY.md Appendix A (restating bpl/dixon_coles.py:39-84,
// bpl/extended_dixon_coles.py:78-248, bpl/_util.py:17-93 under numpyro semantics).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "dc_layout.h"

namespace dc {

constexpr int LANE_FIX = 8;              // fixtures per lane per tile
constexpr int TILE = 64 * LANE_FIX;      // fixtures per wave-tile
constexpr int STREAM_BLOCK = 512;        // 8 waves per workgroup
constexpr int STREAM_WAVES = STREAM_BLOCK / 64;
constexpr int EPI_BLOCK = 256;
constexpr int N_SCAL = 4;                // SLAM, SLOG2, SU, CLIPC
constexpr int BOUNDS_WORDS = 8;          // u32 words per chain
constexpr uint32_t PAD_TEAM_SENTINEL = 0xFFFFu;

struct StreamArgs {
    const uint4* h;    // [n_tiles*64]  8 x u16 home index per lane
    const uint4* a;    // [n_tiles*64]  8 x u16 away index
    const uint2* x;    // [n_tiles*64]  8 x u8 home goals
    const uint2* y;    // [n_tiles*64]  8 x u8 away goals
    const float4* w;   // [n_tiles*128] 8 x f32 weights, or nullptr
    int n_tiles;
    int tiles_per_wave;
    const uint32_t* pairs;  // [P] unique (home | away<<16)
    int P;
    const double* xs;       // [T,K] standardised covariates (device) or nullptr
    double* slabs;          // [chains][n_wg][slab_stride]
    int slab_stride;        // 3T + N_SCAL
    uint32_t* bounds;       // [chains][BOUNDS_WORDS]
    Layout L;
};

// ------------------------------------------------------------------ wave helpers

__device__ __forceinline__ float wave_sum_f32(float v) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d, 64);
    return v;
}
__device__ __forceinline__ double wave_sum_f64(double v) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d, 64);
    return v;
}
__device__ __forceinline__ unsigned long long wave_max_u64(unsigned long long v) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        unsigned long long o = __shfl_xor(v, d, 64);
        v = o > v ? o : v;
    }
    return v;
}

// acc layout in LDS: att[T1] | def[T1] | ha[T1]   (T1 = T+1, slot T = padding sink)
__device__ __forceinline__ void flush_run(double* acc, int T1, uint32_t key, float sh,
                                          float sa) {
    const int h = key & 0xFFFFu, a = key >> 16;
    const double dh = (double)sh, da = (double)sa;
    atomicAdd(&acc[h], dh);            // d/d attack[home]   <- home-rate term
    atomicAdd(&acc[2 * T1 + h], dh);   // d/d home_adv[home]
    atomicAdd(&acc[T1 + a], dh);       // d/d defence[away]
    atomicAdd(&acc[a], da);            // d/d attack[away]   <- away-rate term
    atomicAdd(&acc[T1 + h], da);       // d/d defence[home]
}

// -------------------------------------------------------------------- dc_stream

template <bool WEIGHTED, bool CLIP>
__global__ __launch_bounds__(STREAM_BLOCK) void dc_stream(StreamArgs A,
                                                          const double* __restrict__ zs) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const Layout& L = A.L;
    const int T = L.T, T1 = T + 1;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int chain = blockIdx.y;
    const double* z = zs + (size_t)chain * L.D;

    // LDS carve (all offsets multiples of 16 B)
    float2* tabH = reinterpret_cast<float2*>(smem);            // {exp(att+ha), exp(-def)}
    float2* tabA = tabH + ((T1 + 1) & ~1);                     // {exp(att),    exp(-def)}
    double* acc = reinterpret_cast<double*>(tabA + ((T1 + 1) & ~1));  // [3*T1]
    double* red = acc + 3 * T1 + ((3 * T1) & 1);               // [STREAM_WAVES*4]
    unsigned long long* redm =
        reinterpret_cast<unsigned long long*>(red + STREAM_WAVES * N_SCAL);  // [W*3]
    double* shq = reinterpret_cast<double*>(redm + STREAM_WAVES * 3);        // [2]

    // ---- 0. issue the first tile's loads before anything else (latency overlap)
    const int gw = blockIdx.x * STREAM_WAVES + wave;
    int tile = gw * A.tiles_per_wave;
    const int tile_end = min(tile + A.tiles_per_wave, A.n_tiles);
    uint4 hv = make_uint4(0, 0, 0, 0), av = hv;
    uint2 xv = make_uint2(0, 0), yv = xv;
    float4 w0 = make_float4(0, 0, 0, 0), w1 = w0;
    if (tile < tile_end) {
        const size_t o = (size_t)tile * 64 + lane;
        hv = A.h[o];
        av = A.a[o];
        xv = A.x[o];
        yv = A.y[o];
        if (WEIGHTED) {
            w0 = A.w[2 * o];
            w1 = A.w[2 * o + 1];
        }
    }

    // ---- 1. per-team tables (float64 math, float32 storage) + zero accumulators
    for (int t = tid; t < T1; t += STREAM_BLOCK) {
        float2 vh = make_float2(0.f, 0.f), va = vh;
        if (t < T) {
            double att, def, ha;
            team_params(L, z, A.xs, t, &att, &def, &ha);
            const float edn = (float)exp(-def);
            vh = make_float2((float)exp(att + ha), edn);
            va = make_float2((float)exp(att), edn);
        }
        tabH[t] = vh;
        tabA[t] = va;
    }
    for (int i = tid; i < 3 * T1; i += STREAM_BLOCK) acc[i] = 0.0;
    if (tid == STREAM_BLOCK - 1) {
        double q, dq;
        clipped_sigmoid(z[L.o_corr], &q, &dq);
        shq[0] = q;
    }
    __syncthreads();

    // ---- 2. rho bounds over the unique-pair table (bpl/_util.py:23-30)
    unsigned long long kP = 0, kQ = 0, kR = 0;
    for (int p = tid; p < A.P; p += STREAM_BLOCK) {
        const uint32_t pr = A.pairs[p];
        const float2 th = tabH[pr & 0xFFFFu], ta = tabA[pr >> 16];
        float lh = th.x * ta.y, la = ta.x * th.y;
        if (CLIP) {
            lh = fminf(lh, (float)RATE_CLIP);
            la = fminf(la, (float)RATE_CLIP);
        }
        const unsigned long long idx = 0xFFFFFFFFu - (uint32_t)p;  // ties -> smallest p
        const unsigned long long cP = ((unsigned long long)__float_as_uint(lh * la) << 32) | idx;
        const unsigned long long cQ = ((unsigned long long)__float_as_uint(lh) << 32) | idx;
        const unsigned long long cR = ((unsigned long long)__float_as_uint(la) << 32) | idx;
        kP = cP > kP ? cP : kP;
        kQ = cQ > kQ ? cQ : kQ;
        kR = cR > kR ? cR : kR;
    }
    kP = wave_max_u64(kP);
    kQ = wave_max_u64(kQ);
    kR = wave_max_u64(kR);
    if (lane == 0) {
        redm[wave * 3 + 0] = kP;
        redm[wave * 3 + 1] = kQ;
        redm[wave * 3 + 2] = kR;
    }
    __syncthreads();
#pragma unroll
    for (int wv = 0; wv < STREAM_WAVES; ++wv) {
        const unsigned long long p0 = redm[wv * 3 + 0], p1 = redm[wv * 3 + 1],
                                 p2 = redm[wv * 3 + 2];
        kP = p0 > kP ? p0 : kP;
        kQ = p1 > kQ ? p1 : kQ;
        kR = p2 > kR ? p2 : kR;
    }
    const float Mf = __uint_as_float((uint32_t)(kP >> 32));
    const float Lhf = __uint_as_float((uint32_t)(kQ >> 32));
    const float Laf = __uint_as_float((uint32_t)(kR >> 32));
    const double q = shq[0];
    const double UB = Mf > 1.0f ? 1.0 / (double)Mf : 1.0;
    const double LB = -1.0 / (double)fmaxf(Lhf, Laf);
    const float rho = (float)(LB + q * (UB - LB));

    if (blockIdx.x == 0 && tid == 0) {
        uint32_t* b = A.bounds + (size_t)chain * BOUNDS_WORDS;
        const uint32_t iP = 0xFFFFFFFFu - (uint32_t)kP, iQ = 0xFFFFFFFFu - (uint32_t)kQ,
                       iR = 0xFFFFFFFFu - (uint32_t)kR;
        const uint32_t pP = A.P ? A.pairs[iP] : 0, pQ = A.P ? A.pairs[iQ] : 0,
                       pR = A.P ? A.pairs[iR] : 0;
        uint32_t flags = 0;
        if (CLIP) {
            const float c = (float)RATE_CLIP;
            if (tabH[pP & 0xFFFFu].x * tabA[pP >> 16].y > c) flags |= 1u;  // P home clipped
            if (tabA[pP >> 16].x * tabH[pP & 0xFFFFu].y > c) flags |= 2u;  // P away clipped
            if (tabH[pQ & 0xFFFFu].x * tabA[pQ >> 16].y > c) flags |= 4u;  // Q home clipped
            if (tabA[pR >> 16].x * tabH[pR & 0xFFFFu].y > c) flags |= 8u;  // R away clipped
        }
        b[0] = __float_as_uint(Mf);
        b[1] = __float_as_uint(Lhf);
        b[2] = __float_as_uint(Laf);
        b[3] = pP;
        b[4] = pQ;
        b[5] = pR;
        b[6] = flags;
        b[7] = 0;
    }

    // ---- 3. stream the fixtures
    double dSLAM = 0.0, dSLOG = 0.0, dSU = 0.0, dCLIP = 0.0;
    while (tile < tile_end) {
        // prefetch the next tile
        uint4 hn = hv, an = av;
        uint2 xn = xv, yn = yv;
        float4 w0n = w0, w1n = w1;
        if (tile + 1 < tile_end) {
            const size_t o = (size_t)(tile + 1) * 64 + lane;
            hn = A.h[o];
            an = A.a[o];
            xn = A.x[o];
            yn = A.y[o];
            if (WEIGHTED) {
                w0n = A.w[2 * o];
                w1n = A.w[2 * o + 1];
            }
        }
        const uint32_t hw[4] = {hv.x, hv.y, hv.z, hv.w};
        const uint32_t aw[4] = {av.x, av.y, av.z, av.w};
        const uint32_t xw[2] = {xv.x, xv.y};
        const uint32_t yw[2] = {yv.x, yv.y};
        const float wj[8] = {w0.x, w0.y, w0.z, w0.w, w1.x, w1.y, w1.z, w1.w};

        float shj[LANE_FIX], saj[LANE_FIX];
        uint32_t keyj[LANE_FIX];
        float slam = 0.f, slog = 0.f, su = 0.f, sclip = 0.f;
#pragma unroll
        for (int j = 0; j < LANE_FIX; ++j) {
            const uint32_t hj = (hw[j >> 1] >> (16 * (j & 1))) & 0xFFFFu;
            const uint32_t aj = (aw[j >> 1] >> (16 * (j & 1))) & 0xFFFFu;
            const uint32_t xj = (xw[j >> 2] >> (8 * (j & 3))) & 0xFFu;
            const uint32_t yj = (yw[j >> 2] >> (8 * (j & 3))) & 0xFFu;
            keyj[j] = hj | (aj << 16);
            const float2 th = tabH[hj], ta = tabA[aj];
            float lh = th.x * ta.y;  // exp(att[h] + ha[h]) * exp(-def[a])
            float la = ta.x * th.y;  // exp(att[a]) * exp(-def[h])
            bool ch = false, ca = false;
            float lh_raw = lh, la_raw = la;
            if (CLIP) {
                ch = lh > (float)RATE_CLIP;
                ca = la > (float)RATE_CLIP;
                lh = ch ? (float)RATE_CLIP : lh;
                la = ca ? (float)RATE_CLIP : la;
            }
            // tau (bpl/_util.py:58-91): arg = 1 + rho*c,
            //   c = -lh*la (0,0) | +la (1,0) | +lh (0,1) | -1 (1,1) | 0 otherwise
            const bool x0 = xj == 0, y0 = yj == 0;
            const bool low = (xj <= 1) & (yj <= 1);
            float c = x0 ? (y0 ? -lh * la : lh) : (y0 ? la : -1.0f);
            c = low ? c : 0.0f;
            const float t = fmaf(rho, c, 1.0f);
            const float l2 = __log2f(fmaxf(t, 0.0f));           // log(clip(.,0)): -inf at 0
            const float u = t > 0.0f ? c * __builtin_amdgcn_rcpf(t) : 0.0f;  // dlogtau/drho
            const float ru = rho * u;
            // -(dL/d eta) without the data-only goal counts (added in the epilogue):
            //   eta_h: lh - rho*u*[x==0]   (lh * dlogtau/dlh = rho*u for (0,0),(0,1))
            float sh = lh - (x0 ? ru : 0.0f);
            float sa = la - (y0 ? ru : 0.0f);
            float wv = 1.0f;
            if (WEIGHTED) wv = wj[j];
            if (CLIP) {
                // clipped rate: d/d eta = 0 -> cancel the goal count added later, and
                // correct k*eta -> k*log(15) in the value
                if (ch) {
                    sh = (float)xj;
                    sclip += wv * (float)xj * (__logf(lh_raw) - (float)LOG_RATE_CLIP);
                }
                if (ca) {
                    sa = (float)yj;
                    sclip += wv * (float)yj * (__logf(la_raw) - (float)LOG_RATE_CLIP);
                }
            }
            if (WEIGHTED) {
                sh *= wv;
                sa *= wv;
                slam += wv * (lh + la);
                slog += wv * l2;
                su += wv * u;
            } else {
                slam += lh + la;
                slog += l2;
                su += u;
            }
            shj[j] = sh;
            saj[j] = sa;
        }
        dSLAM += (double)slam;
        dSLOG += (double)slog;
        dSU += (double)su;
        if (CLIP) dCLIP += (double)sclip;

        // ---- per-(home,away) run sums: lane -> wave -> LDS per-team accumulators
        uint32_t diff = 0;
        float rsh = 0.f, rsa = 0.f;
#pragma unroll
        for (int j = 0; j < LANE_FIX; ++j) {
            diff |= keyj[j] ^ keyj[0];
            rsh += shj[j];
            rsa += saj[j];
        }
        uint32_t key = keyj[0];
        if (diff != 0) {  // rare: a pair boundary inside this lane's 8 fixtures
#pragma unroll
            for (int j = 0; j < LANE_FIX; ++j) flush_run(acc, T1, keyj[j], shj[j], saj[j]);
            rsh = 0.f;
            rsa = 0.f;
            key = keyj[LANE_FIX - 1];
        }
        const uint32_t k0 = __builtin_amdgcn_readfirstlane(key);
        if (__all(key == k0)) {  // whole wave-tile on one pair (the common case: sorted)
            rsh = wave_sum_f32(rsh);
            rsa = wave_sum_f32(rsa);
            if (lane == 0) flush_run(acc, T1, key, rsh, rsa);
        } else {  // segmented inclusive scan over runs of equal keys
            const uint32_t kprev = __shfl_up(key, 1, 64);
            int f = (lane == 0) | (kprev != key);
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                const float u0 = __shfl_up(rsh, d, 64), u1 = __shfl_up(rsa, d, 64);
                const int fu = __shfl_up(f, d, 64);
                if (lane >= d && !f) {
                    rsh += u0;
                    rsa += u1;
                    f |= fu;
                }
            }
            const uint32_t knext = __shfl_down(key, 1, 64);
            if (lane == 63 || knext != key) flush_run(acc, T1, key, rsh, rsa);
        }

        hv = hn;
        av = an;
        xv = xn;
        yv = yn;
        w0 = w0n;
        w1 = w1n;
        ++tile;
    }

    // ---- 4. workgroup reduction of the scalars, then the slab
    dSLAM = wave_sum_f64(dSLAM);
    dSLOG = wave_sum_f64(dSLOG);
    dSU = wave_sum_f64(dSU);
    if (CLIP) dCLIP = wave_sum_f64(dCLIP);
    if (lane == 0) {
        red[wave * N_SCAL + 0] = dSLAM;
        red[wave * N_SCAL + 1] = dSLOG;
        red[wave * N_SCAL + 2] = dSU;
        red[wave * N_SCAL + 3] = dCLIP;
    }
    __syncthreads();
    double* slab = A.slabs + ((size_t)chain * gridDim.x + blockIdx.x) * A.slab_stride;
    for (int i = tid; i < 3 * T; i += STREAM_BLOCK) {
        const int which = i / T, t = i - which * T;
        slab[i] = acc[which * T1 + t];
    }
    if (tid < N_SCAL) {
        double s = 0.0;
#pragma unroll
        for (int wv = 0; wv < STREAM_WAVES; ++wv) s += red[wv * N_SCAL + tid];
        slab[3 * T + tid] = s;
    }
}

// ------------------------------------------------------------------ dc_epilogue

struct EpiArgs {
    const double* slabs;
    int n_wg, slab_stride;
    const uint32_t* bounds;
    const double* xs;   // [T,K]
    const double* cA;   // [T] sum_{h=t} w x + sum_{a=t} w y   (coefficient of attack_t)
    const double* cD;   // [T] sum_{a=t} w x + sum_{h=t} w y   (coefficient of -defence_t)
    const double* cH;   // [T] sum_{h=t} w x                   (coefficient of home_adv_t)
    double lgsum;       // sum_i w_i (lgamma(x_i+1) + lgamma(y_i+1))
    Layout L;
};

template <int NV>
__device__ __forceinline__ void block_sum(double (&v)[NV], double* scratch, int tid) {
    const int lane = tid & 63, wave = tid >> 6;
#pragma unroll
    for (int i = 0; i < NV; ++i) v[i] = wave_sum_f64(v[i]);
    __syncthreads();
    if (lane == 0) {
#pragma unroll
        for (int i = 0; i < NV; ++i) scratch[wave * NV + i] = v[i];
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        double s = 0.0;
        for (int wv = 0; wv < EPI_BLOCK / 64; ++wv) s += scratch[wv * NV + i];
        v[i] = s;
    }
}

__global__ __launch_bounds__(EPI_BLOCK) void dc_epilogue(EpiArgs A,
                                                         const double* __restrict__ zs,
                                                         double* __restrict__ potential,
                                                         double* __restrict__ grads,
                                                         double* __restrict__ auxs) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const Layout& L = A.L;
    const int T = L.T, K = L.K;
    const int tid = threadIdx.x;
    const int chain = blockIdx.x;
    const double* z = zs + (size_t)chain * L.D;
    double* grad = grads + (size_t)chain * L.D;
    const int ncol = 3 * T + N_SCAL;

    double* col = reinterpret_cast<double*>(smem);  // [ncol] reduced slabs
    double* scratch = col + ncol;                   // [4*16]
    double* part = scratch + 64;                    // [RG*ncol]

    // ---- 1. fixed-order reduction of the workgroup slabs
    int RG = EPI_BLOCK / ncol;
    RG = RG < 1 ? 1 : (RG > 8 ? 8 : RG);
    const double* slabs = A.slabs + (size_t)chain * A.n_wg * A.slab_stride;
    for (int idx = tid; idx < ncol * RG; idx += EPI_BLOCK) {
        const int rg = idx / ncol, c = idx - rg * ncol;
        double s = 0.0;
        for (int wg = rg; wg < A.n_wg; wg += RG) s += slabs[(size_t)wg * A.slab_stride + c];
        part[rg * ncol + c] = s;
    }
    __syncthreads();
    for (int c = tid; c < ncol; c += EPI_BLOCK) {
        double s = 0.0;
        for (int rg = 0; rg < RG; ++rg) s += part[rg * ncol + c];
        col[c] = s;
    }
    __syncthreads();
    double* g_att = col;          // becomes dL/d attack_t
    double* g_def = col + T;      // dL/d defence_t
    double* g_ha = col + 2 * T;   // dL/d home_adv_t
    const double SLAM = col[3 * T + 0], SLOG = col[3 * T + 1], SU = col[3 * T + 2],
                 CLIPC = col[3 * T + 3];
    for (int t = tid; t < T; t += EPI_BLOCK) {
        g_att[t] = A.cA[t] - g_att[t];
        g_def[t] = -(A.cD[t] - g_def[t]);
        g_ha[t] = A.cH[t] - g_ha[t];
    }
    __syncthreads();

    // ---- 2. rho = LB + q (UB - LB) and the adjoint of the bounds (Appendix A.3)
    const uint32_t* b = A.bounds + (size_t)chain * BOUNDS_WORDS;
    const double M = (double)__uint_as_float(b[0]);
    const double Lh = (double)__uint_as_float(b[1]);
    const double La = (double)__uint_as_float(b[2]);
    double q, dq;
    clipped_sigmoid(z[L.o_corr], &q, &dq);
    const double UB = M > 1.0 ? 1.0 / M : 1.0;
    const double LB = -1.0 / fmax(Lh, La);
    const double rho = LB + q * (UB - LB);
    const double G_rho = SU;  // sum_i w_i dlogtau_i/drho
    if (tid == 0) {
        const uint32_t pP = b[3], pQ = b[4], pR = b[5], flags = b[6];
        if (M > 1.0) {  // UB = 1/M : d/d eta_h[P] = d/d eta_a[P] = -1/M
            const double v = G_rho * q * (-UB);
            const int h = pP & 0xFFFFu, a = pP >> 16;
            if (!(flags & 1u)) {
                g_att[h] += v;
                g_ha[h] += v;
                g_def[a] -= v;
            }
            if (!(flags & 2u)) {
                g_att[a] += v;
                g_def[h] -= v;
            }
        }
        const double v = G_rho * (1.0 - q) * (-LB);  // LB = -1/Lam : d/d eta = +1/Lam
        if (Lh >= La) {
            const int h = pQ & 0xFFFFu, a = pQ >> 16;
            if (!(flags & 4u)) {
                g_att[h] += v;
                g_ha[h] += v;
                g_def[a] -= v;
            }
        } else {
            const int h = pR & 0xFFFFu, a = pR >> 16;
            if (!(flags & 8u)) {
                g_att[a] += v;
                g_def[h] -= v;
            }
        }
    }
    __syncthreads();

    // ---- 3. priors, Jacobians, chain rule (float64).  L = log density; U = -L.
    const double zc = z[L.o_corr];
    const double sigc = sigmoid(zc);
    // Beta(2,2) on q + sigmoid Jacobian
    double Lsc = log(q) + log1p(-q) + 1.791759469228055 /*log 6*/ - softplus(zc) - softplus(-zc);
    const double g_corr = G_rho * (UB - LB) * dq + (1.0 / q - 1.0 / (1.0 - q)) * dq +
                          (1.0 - 2.0 * sigc);
    const double m = z[L.o_md];
    const double zsa = z[L.o_sa], zsd = z[L.o_sd];
    const double s_a = exp(zsa), s_d = exp(zsd);
    // HalfNormal(1) on exp(z) + Exp Jacobian: -s^2/2 - log sqrt(2 pi) + log 2 + z
    Lsc += -0.5 * s_a * s_a - HALF_LOG_2PI + LN2 + zsa;
    Lsc += -0.5 * s_d * s_d - HALF_LOG_2PI + LN2 + zsd;
    Lsc += -0.5 * m * m - HALF_LOG_2PI;  // mean_defence ~ N(0,1)

    if (L.model == MODEL_BASIC) {
        const double gam = z[L.o_ha];
        {
            const double r = (gam - 0.1) / 0.2;
            Lsc += -0.5 * r * r - log(0.2) - HALF_LOG_2PI;
        }
        // v: 0 sum g_def, 1 sum g_ha, 2 sum a~ g_att, 3 sum d~ g_def, 4 team part of L
        double v[5] = {0, 0, 0, 0, 0};
        for (int t = tid; t < T; t += EPI_BLOCK) {
            const double ad = z[L.o_adec + t], dd = z[L.o_ddec + t];
            const double ga = g_att[t], gd = g_def[t], gh = g_ha[t];
            grad[L.o_adec + t] = -(s_a * ga - ad);
            grad[L.o_ddec + t] = -(s_d * gd - dd);
            v[0] += gd;
            v[1] += gh;
            v[2] += ad * ga;
            v[3] += dd * gd;
            const double att = s_a * ad, def = m + s_d * dd;
            v[4] += -0.5 * ad * ad - 0.5 * dd * dd - 2.0 * HALF_LOG_2PI + att * A.cA[t] -
                    def * A.cD[t] + gam * A.cH[t];
        }
        block_sum<5>(v, scratch, tid);
        if (tid == 0) {
            grad[L.o_ha] = -(v[1] - (gam - 0.1) / 0.04);
            grad[L.o_md] = -(v[0] - m);
            grad[L.o_sa] = -(s_a * v[2] - s_a * s_a + 1.0);
            grad[L.o_sd] = -(s_d * v[3] - s_d * s_d + 1.0);
            grad[L.o_corr] = -g_corr;
            const double Ltot = Lsc + v[4] - SLAM - A.lgsum + LN2 * SLOG - CLIPC;
            potential[chain] = -Ltot;
        }
    } else {
        const double mha = z[L.o_mha], zsh = z[L.o_sh], s_h = exp(zsh);
        {
            const double r = (mha - 0.1) / 0.2;
            Lsc += -0.5 * r * r - log(0.2) - HALF_LOG_2PI;
        }
        Lsc += -0.5 * s_h * s_h - HALF_LOG_2PI + LN2 + zsh;
        const double zu = z[L.o_u];
        double u, du;
        clipped_sigmoid(zu, &u, &du);
        // Beta(2,4) on u + sigmoid Jacobian
        Lsc += log(u) + 3.0 * log1p(-u) + 2.995732273553991 /*log 20*/ - softplus(zu) -
               softplus(-zu);
        const double rp = 2.0 * u - 1.0, vv = 1.0 - rp * rp;
        // v: 0 sum g_def, 1 sum g_ha, 2 sum sa g_att, 3 sum sd g_def, 4 sum ha~ g_ha,
        //    5 dL/d rho_p, 6 team part of L
        double v[7] = {0, 0, 0, 0, 0, 0, 0};
        for (int t = tid; t < T; t += EPI_BLOCK) {
            const double sa = z[L.o_sat + t], sd = z[L.o_sdt + t], hd = z[L.o_hadec + t];
            const double ga = g_att[t], gd = g_def[t], gh = g_ha[t];
            const double e = sd - rp * sa;
            grad[L.o_sat + t] = -(s_a * ga - sa + rp * e / vv);
            grad[L.o_sdt + t] = -(s_d * gd - e / vv);
            grad[L.o_hadec + t] = -(s_h * gh - hd);
            v[0] += gd;
            v[1] += gh;
            v[2] += sa * ga;
            v[3] += sd * gd;
            v[4] += hd * gh;
            v[5] += e * sa / vv - rp * e * e / (vv * vv) + rp / vv;
            double att, def, ha;
            team_params(L, z, A.xs, t, &att, &def, &ha);
            v[6] += -0.5 * sa * sa - 0.5 * e * e / vv - 0.5 * log(vv) - 0.5 * hd * hd -
                    3.0 * HALF_LOG_2PI + att * A.cA[t] - def * A.cD[t] + ha * A.cH[t];
        }
        block_sum<7>(v, scratch, tid);
        // covariate coefficients: d/d beta_k = sum_t Xs[t,k] g_t - beta_k
        for (int k = tid; k < 2 * K; k += EPI_BLOCK) {
            const bool isd = k >= K;
            const int kk = isd ? k - K : k;
            const double* gt = isd ? g_def : g_att;
            double s = 0.0;
            for (int t = 0; t < T; ++t) s += A.xs[(size_t)t * K + kk] * gt[t];
            const int o = (isd ? L.o_bD : L.o_bA) + kk;
            grad[o] = -(s - z[o]);
        }
        if (tid == 0) {
            double Lcov = 0.0;
            for (int k = 0; k < K; ++k) {
                const double ba = z[L.o_bA + k], bd = z[L.o_bD + k];
                Lcov += -0.5 * ba * ba - 0.5 * bd * bd - 2.0 * HALF_LOG_2PI;
            }
            grad[L.o_mha] = -(v[1] - (mha - 0.1) / 0.04);
            grad[L.o_sh] = -(s_h * v[4] - s_h * s_h + 1.0);
            grad[L.o_md] = -(v[0] - m);
            grad[L.o_sa] = -(s_a * v[2] - s_a * s_a + 1.0);
            grad[L.o_sd] = -(s_d * v[3] - s_d * s_d + 1.0);
            grad[L.o_corr] = -g_corr;
            const double sigu = sigmoid(zu);
            grad[L.o_u] = -(2.0 * v[5] * du + (1.0 / u - 3.0 / (1.0 - u)) * du +
                            (1.0 - 2.0 * sigu));
            const double Ltot = Lsc + Lcov + v[6] - SLAM - A.lgsum + LN2 * SLOG - CLIPC;
            potential[chain] = -Ltot;
        }
    }
    if (tid == 0 && auxs != nullptr) {
        double* aux = auxs + (size_t)chain * 4;
        aux[0] = rho;
        aux[1] = LB;
        aux[2] = UB;
        aux[3] = q;
    }
}

}  // namespace dc
