// dc_vec.hip.h -- chain-vectorised evaluation (numpyro chain_method="vectorized",
// bplhip_logp_grad_batched): one streaming workgroup carries CB = 8 chains through its
// fixtures at once.
//
// Everything that depends on the fixtures only is done ONCE per lane and shared by the
// chains: the loads, the unpacking, the pair of the lane's run and its score-class counts
// (SWAR classification).  Per chain and lane what remains is two LDS gathers, two
// products, three tau terms and a handful of FMAs -- about 9 VALU operations per
// (fixture, chain) instead of ~45 for one chain alone, and the fixture bytes cross HBM/L2
// once per 8 chains.
//
//   dc_vec_stream  grid = (CB + n_wg) x ceil(chains / CB), 512 threads.
//     blocks 0..CB-1   prior workgroup of chain group*CB + blockIdx.x (prior_body, unchanged)
//     blocks CB..      streaming: wave b builds chain b's float32 tables and rho (no
//                      cross-wave reduction), then all 8 waves stream the workgroup's
//                      tiles for all 8 chains.  A lane whose 8 fixtures span several
//                      pairs is processed run by run (masked classification), so the
//                      per-chain arithmetic is always the run-level one.
//   dc_vec_tail    grid = chains: tail_body of dc_kernels.hip.h, one workgroup per chain
//                  (a kernel boundary replaces the ticket of the single-chain launch).
//
// Same hand-off record per chain as dc_eval ([zo | scal | compact]), same arithmetic per
// run; only the grouping of the float32 partial sums differs.
#pragma once
#include "dc_kernels.hip.h"

namespace dc {

constexpr int CB = WAVES;  // chains per streaming workgroup (wave b prepares chain b)

__host__ __device__ inline int vec_acc_len(int T) { return (3 * (T + 1) + 1) & ~1; }
__host__ __device__ inline size_t vec_stream_lds_bytes(int T) {
    size_t b = (size_t)CB * 2 * tab_len(T) * 8;   // tables (float2)
    b += (size_t)CB * vec_acc_len(T) * 8;         // accumulators (double)
    b += (size_t)WAVES * CB * 2 * 8;              // red: V, SU per wave and chain
    b += (size_t)CB * 4 + 32;                     // rho
    return b;
}

// what a lane knows about one run of equal (home, away) among its fixtures: shared by
// all chains
struct LaneClass {
    uint32_t key;                              // home | away << 16
    float n00, n10, n01, n11, nall, sx, sy;    // (weighted) score-class counts, goal sums
};

// Take the first not yet processed run of the lane (fixtures in `rem`), classify its
// fixtures, and remove them from `rem`.  A lane with rem == 0 returns an empty run on the
// sentinel team (zero table entries: contributes exactly 0).
template <bool WEIGHTED>
__device__ __forceinline__ LaneClass classify(const LaneData& Ld, uint32_t& rem, uint32_t sentinel) {
    LaneClass c;
    // all fixtures of a lane share one pair (runs are padded to the lane width): one pass
    c.key = (Ld.hw[0] & 0xFFFFu) | (Ld.aw[0] << 16);
    rem = 0;
    if (!WEIGHTED) {  // SWAR counts
        const uint32_t one = 0x01010101u;
        int c00 = 0, c10 = 0, c01 = 0, c11 = 0;
        uint32_t ax = 0, ay = 0;
#pragma unroll
        for (int q = 0; q < XWORDS; ++q) {
            const uint32_t x = Ld.xw[q], y = Ld.yw[q];
            c00 += __popc(zero_bytes(x | y));
            c10 += __popc(zero_bytes((x ^ one) | y));
            c01 += __popc(zero_bytes(x | (y ^ one)));
            c11 += __popc(zero_bytes((x ^ one) | (y ^ one)));
            ax = __builtin_amdgcn_sad_u8(x, 0u, ax);
            ay = __builtin_amdgcn_sad_u8(y, 0u, ay);
        }
        // null fixtures (goals (255, 255), never a low score) pad the end of a pair's run; the
        // lane's number of REAL fixtures rides above the home index
        const int nz = LANE_FIX - (int)(Ld.hw[0] >> 16);
        c.n00 = (float)c00; c.n10 = (float)c10; c.n01 = (float)c01; c.n11 = (float)c11;
        c.nall = (float)(LANE_FIX - nz);
        c.sx = (float)((int)ax - 255 * nz);
        c.sy = (float)((int)ay - 255 * nz);
        return c;
    }
    c.n00 = c.n10 = c.n01 = c.n11 = c.nall = c.sx = c.sy = 0.f;
#pragma unroll
    for (int j = 0; j < LANE_FIX; ++j) {
        const uint32_t xj = (Ld.xw[j >> 2] >> (8 * (j & 3))) & 0xFFu;
        const uint32_t yj = (Ld.yw[j >> 2] >> (8 * (j & 3))) & 0xFFu;
        const float wv = Ld.wj[j];  // (0 for null fixtures)
        c.n00 += (xj | yj) == 0 ? wv : 0.f;
        c.n10 += (xj == 1 && yj == 0) ? wv : 0.f;
        c.n01 += (xj == 0 && yj == 1) ? wv : 0.f;
        c.n11 += (xj == 1 && yj == 1) ? wv : 0.f;
        c.nall += wv;
        c.sx += wv * (float)xj;
        c.sy += wv * (float)yj;
    }
    (void)sentinel;
    return c;
}

struct ChainOut {
    float rsh, rsa;  // -(dL/d eta_h), -(dL/d eta_a) of the run, without the goal counts
    double slam;   // (float64: see lane_uniform of dc_kernels.hip.h)
    float slog, su, sclip;
};

// one chain's terms of one run (same arithmetic as lane_uniform of dc_kernels.hip.h)
template <bool CLIP>
__device__ __forceinline__ ChainOut chain_terms(const LaneClass& c, float rho, const float2* tabH,
                                                const float2* tabA, float l11, float u11) {
    const float2 th = tabH[c.key & 0xFFFFu], ta = tabA[c.key >> 16];
    float lh = th.x * ta.y;  // exp(att[h] + ha[h]) * exp(-def[a])
    float la = ta.x * th.y;  // exp(att[a]) * exp(-def[h])
    bool ch = false, ca = false;
    if (CLIP) {
        ch = lh > (float)RATE_CLIP;
        ca = la > (float)RATE_CLIP;
        lh = ch ? (float)RATE_CLIP : lh;
        la = ca ? (float)RATE_CLIP : la;
    }
    float l00, u00, l10, u10, l01, u01;
    class_terms(rho, -lh * la, &l00, &u00);
    class_terms(rho, la, &l10, &u10);
    class_terms(rho, lh, &l01, &u01);
    ChainOut o;
    o.slam = (double)c.nall * ((double)lh + (double)la);
    o.slog = (c.n00 != 0.f ? c.n00 * l00 : 0.f) + (c.n10 != 0.f ? c.n10 * l10 : 0.f) +
             (c.n01 != 0.f ? c.n01 * l01 : 0.f) + (c.n11 != 0.f ? c.n11 * l11 : 0.f);
    o.su = c.n00 * u00 + c.n10 * u10 + c.n01 * u01 + c.n11 * u11;
    o.rsh = c.nall * lh - rho * (c.n00 * u00 + c.n01 * u01);
    o.rsa = c.nall * la - rho * (c.n00 * u00 + c.n10 * u10);
    o.sclip = 0.f;
    if (CLIP) {  // (the value's k*eta -> k*log(15) correction: prior_body, ZO_PAIRC)
        if (ch) o.rsh = c.sx;
        if (ca) o.rsa = c.sy;
    }
    return o;
}

// value of chain `lane` out of CB wave-uniform values (lanes >= CB get the last one)
__device__ __forceinline__ float pick_chain(const float (&v)[CB], int lane) {
    float r = v[CB - 1];
#pragma unroll
    for (int b = CB - 2; b >= 0; --b) r = lane == b ? v[b] : r;
    return r;
}

template <bool WEIGHTED, bool CLIP, bool NUTS>
__global__ __launch_bounds__(BLOCK) void dc_vec_stream(EvalArgs A) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const Layout& L = A.L;
    const int T = L.T, T1 = T + 1, tl = tab_len(T), accn = vec_acc_len(T);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int chain0 = blockIdx.y * CB;

    if (blockIdx.x < CB) {  // prior workgroups of this group's chains
        const int chain = chain0 + blockIdx.x;
        if (chain >= A.chains) return;
        if (NUTS && nuts_of(A, chain)[nd::H_S_DONE] != 0.0) return;  // subtree already complete
        prior_body<CLIP>(A, chain, smem);
        return;
    }
    const int wgi = blockIdx.x - CB;
    if (NUTS) {  // lock-step chains: nothing to do once every chain of the group has finished
        bool all_done = true;
        for (int b = 0; b < CB && chain0 + b < A.chains; ++b)
            all_done = all_done && nuts_of(A, chain0 + b)[nd::H_S_DONE] != 0.0;
        if (all_done) return;
    }

    // ---- 0. loads in the order the data is wanted, unconditional with clamped indices (see the
    // prologue of dc_eval): this wave's chain's per-team entries of z, then the first tile
    const double* zb = z_of(A, min(chain0 + wave, A.chains - 1));
    const TeamZ tz0 = load_team_z<CLIP>(L, zb, min(lane, T - 1));
    const int gw = wgi * WAVES + wave;
    int tile = gw * A.tiles_per_wave;
    const int tile_end = min(tile + A.tiles_per_wave, A.n_tiles);
    LaneData cur = load_lane<WEIGHTED>(A, (size_t)min(tile, A.n_tiles - 1) * 64 + lane);
    const int o0 = A.wg_off[wgi], o1 = A.wg_off[wgi + 1];

    float2* tab = reinterpret_cast<float2*>(smem);                   // [CB][2][tl]
    double* acc = reinterpret_cast<double*>(tab + (size_t)CB * 2 * tl);  // [CB][accn]
    double* red = acc + (size_t)CB * accn;                           // [WAVES][CB][2]
    float* rhoL = reinterpret_cast<float*>(red + WAVES * CB * 2);    // [CB]

    // ---- 1. wave b: chain b's float32 tables, zeroed accumulators, rho
    {
        const int b = wave;
        const int chain = min(chain0 + b, A.chains - 1);  // (a padding chain repeats the last)
        const double* z = z_of(A, chain);
        const F32Scalars fs = f32_scalars<CLIP>(L, z);
        float2* tH = tab + (size_t)(2 * b) * tl;
        float2* tA = tH + tl;
        for (int t = lane; t <= T; t += 64) {
            float2 vh = make_float2(0.f, 0.f), va = vh;
            if (t < T)
                f32_table_entry<CLIP>(L, fs, z, A.xsf, t, t == lane ? tz0 : load_team_z<CLIP>(L, z, t), &vh, &va);
            tH[t] = vh;
            tA[t] = va;
        }
        for (int i = lane; i < accn; i += 64) acc[(size_t)b * accn + i] = 0.0;
        __syncthreads();
        float mP = 0.f, mQ = 0.f, mR = 0.f;  // bpl/_util.py:23-30 over the unique pairs
        // (eight pairs per round, index clamped: a repeat of the last pair changes no maximum; one
        // pair per round was a dependent L2 round trip each)
        for (int p0 = lane; p0 < A.P; p0 += 8 * 64) {
            uint32_t q[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) q[u] = A.pairs[min(p0 + 64 * u, A.P - 1)];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const uint32_t pr = q[u];
                const float2 th = tH[pr & 0xFFFFu], ta = tA[pr >> 16];
                float lh = th.x * ta.y, la = ta.x * th.y;
                if (CLIP) {
                    lh = fminf(lh, (float)RATE_CLIP);
                    la = fminf(la, (float)RATE_CLIP);
                }
                mP = fmaxf(mP, lh * la);
                mQ = fmaxf(mQ, lh);
                mR = fmaxf(mR, la);
            }
        }
        mP = wave_max_f32(mP);
        mQ = wave_max_f32(mQ);
        mR = wave_max_f32(mR);
        if (lane == 0) rhoL[b] = rho_f32(mP, mQ, mR, fs.q);
        __syncthreads();
    }
    // ---- 2. stream the fixtures.  Round 4: the chain-dependent arithmetic is done PER RUN, not per lane.
    // A lane's fixtures are one (home, away) pair (runs are padded to the lane width) and a tile of 64 lanes
    // holds one or two pairs at N = 1e6, so the rates and the three tau terms of a chain were worked out 64
    // times per tile with the same inputs -- 8 chains x (2 LDS gathers + 3 rcp + 3 log2 + ~35 other
    // operations) per lane, 4.6 us per tile and wave, 200+ registers and spills.  Everything a chain's
    // sums need from the fixtures is LINEAR in the lane's seven class sums (weights of the (0,0) / (1,0) /
    // (0,1) / (1,1) scorelines, of all fixtures, of the goals on either side), with coefficients that
    // depend on (run, chain) only.  So: classify per lane (shared by the chains, as before), add the seven
    // sums over the lanes of the run (DPP, chain independent), and let LANE b work chain b's terms out ONCE
    // from the run's totals -- the eight chains side by side on eight lanes.  The fixtures are still
    // streamed and classified on every evaluation; what changed is how often a transcendental is taken.
    const int myb = lane < CB ? lane : CB - 1;   // the chain this lane works for (lanes 8.. shadow chain 7)
    const float my_rho = rhoL[myb];
    float my_l11, my_u11;
    class_terms(my_rho, -1.0f, &my_l11, &my_u11);
    const float2* myH = tab + (size_t)(2 * myb) * tl;
    const float2* myA = myH + tl;
    double dV = 0.0, dSU = 0.0;   // lane b < CB: chain b's  ln2*SLOG - SLAM - CLIPC  and SU of this wave
    const uint32_t sentinel = (uint32_t)T;
    auto run_terms = [&](const LaneClass& tot) {   // one run's totals -> chain `myb`'s sums and accumulators
        const ChainOut o = chain_terms<CLIP>(tot, my_rho, myH, myA, my_l11, my_u11);
        if (lane < CB) {
            double v = fma((double)LN2, (double)o.slog, -o.slam);
            if (CLIP) v -= (double)o.sclip;
            dV += v;
            dSU += (double)o.su;
            flush_run(acc + (size_t)lane * accn, T1, tot.key, o.rsh, o.rsa);
        }
    };
    while (tile < tile_end) {
        // (unconditional prefetch: the last round re-requests its own tile and drops it)
        LaneData nxt = load_lane<WEIGHTED>(A, (size_t)min(tile + 1, tile_end - 1) * 64 + lane);
        uint32_t rem = 0;
        const LaneClass lc = classify<WEIGHTED>(cur, rem, sentinel);   // (one run per lane)
        const uint32_t kprev = prev_lane_u32(lc.key, ~lc.key);
        unsigned long long hd = __ballot(kprev != lc.key);  // lane 0 always a head
        while (hd) {   // the tile's runs, one after the other (one or two at N = 1e6; up to 64 in a tiny league)
            const int first = __ffsll((long long)hd) - 1;
            hd &= hd - 1;
            const int stop = hd ? __ffsll((long long)hd) - 1 : 64;
            const bool in = lane >= first && lane < stop;
            LaneClass tot;
            tot.key = (uint32_t)__builtin_amdgcn_readlane((int)lc.key, first);
            float a0 = in ? lc.n00 : 0.f, a1 = in ? lc.n10 : 0.f, a2 = in ? lc.n01 : 0.f, a3 = in ? lc.n11 : 0.f;
            float a4 = in ? lc.nall : 0.f, a5 = in ? lc.sx : 0.f, a6 = in ? lc.sy : 0.f;
            wave_sum2_f32(a0, a1);
            wave_sum2_f32(a2, a3);
            wave_sum2_f32(a4, a5);
            a6 = wave_sum_f32(a6);
            tot.n00 = a0; tot.n10 = a1; tot.n01 = a2; tot.n11 = a3; tot.nall = a4; tot.sx = a5; tot.sy = a6;
            run_terms(tot);
        }
        // (see dc_eval: keeps the compiler from consuming the prefetched words at issue time)
        asm volatile("" : "+v"(nxt.hw[0]), "+v"(nxt.aw[0]), "+v"(nxt.xw[0]), "+v"(nxt.yw[0]));
        cur = nxt;
        ++tile;
    }

    // ---- 3. workgroup reduction of the scalars, then the slabs of all chains
    if (lane < CB) {   // (lane b holds chain b's sums of this wave: nothing to reduce)
        red[(wave * CB + lane) * 2 + 0] = dV;
        red[(wave * CB + lane) * 2 + 1] = dSU;
    }
    __syncthreads();
    const int cnt = o1 - o0;
    const int nvalid = min(CB, A.chains - chain0);
    for (int i = tid; i < cnt * nvalid; i += BLOCK) {
        const int b = i / cnt, k = o0 + (i - b * cnt);
        const int slot = A.wg_slots[k];
        const int which = slot / T, t = slot - which * T;
        double* cmpw = A.hbuf + (size_t)(chain0 + b) * A.hb_stride + A.zo_stride + A.n_wg * N_SCAL;
        st_sc1(&cmpw[A.wg_dst[k]], acc[(size_t)b * accn + which * T1 + t]);
    }
    if (tid < CB * N_SCAL) {
        const int b = tid / N_SCAL, j = tid - b * N_SCAL;
        if (b < nvalid) {
            // the tail forms  -SLAM + ln2*SLOG - CLIPC: hand it V as -SLAM, and SU
            double s = 0.0;
            if (j == 0 || j == 2) {
#pragma unroll
                for (int wv = 0; wv < WAVES; ++wv) s += red[(wv * CB + b) * 2 + (j == 0 ? 0 : 1)];
                if (j == 0) s = -s;
            }
            st_sc1(&A.hbuf[(size_t)(chain0 + b) * A.hb_stride + A.zo_stride + wgi * N_SCAL + j], s);
        }
    }
}

template <bool STAGED, bool NUTS, bool EXT>
__global__ __launch_bounds__(BLOCK) void dc_vec_tail(EvalArgs A) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    if (NUTS && nuts_of(A, blockIdx.x)[nd::H_S_DONE] != 0.0) return;
    tail_body<STAGED, NUTS, EXT>(A, blockIdx.x, smem);
}

}  // namespace dc
