// nuts_dev.hip.h -- device-resident half of the NUTS transition (numpyro 0.13.2 iterative
// tree; host half and adaptation in nuts.hpp / bplhip.hip).
//
// Why: with a host-side tree every leapfrog costs a launch + a device->host read-back
// (~45 us at N = 1e6 against ~11 us of kernel).  Here the leaf bookkeeping (velocity-Verlet
// half steps, energy, uniform multinomial transition with threefry, checkpointed U-turn
// test) runs in the tail of dc_eval itself, on the wave that has just produced the
// gradient, and writes the NEXT leapfrog's position; the host enqueues the 2^j launches of
// a doubling (or of several doublings) without reading anything back and synchronises once
// per batch.  Launches that come after the subtree has finished return at once.
//
// All state lives in one device buffer ("NS"); one wave works on it, lanes stride over D.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "lean_math.hip.h"
#include "wave_reduce.hip.h"

namespace nd {

// The integrator arithmetic below must round exactly like the host driver's (nuts.hpp,
// compiled without FMA): no contraction, so both tree builders walk the same trajectory.
#pragma clang fp contract(off)

// ---- header words (doubles unless noted)
enum {
    H_EPS = 0,        // step size (positive)
    H_DIR,            // +1 / -1 direction of the current doubling
    H_E0,             // energy at the start of the transition
    H_MAXDE,          // divergence threshold
    // tree (whole transition)
    H_T_DEPTH, H_T_WEIGHT, H_T_TURN, H_T_DIV, H_T_SUMACC, H_T_NUM, H_T_PE, H_T_EPROP,
    // subtree of the current doubling
    H_S_MAX,          // leaves wanted = 2^depth
    H_S_NUM, H_S_WEIGHT, H_S_TURN, H_S_DIV, H_S_SUMACC, H_S_PE, H_S_EPROP,
    H_S_DONE,         // 1: remaining launches of this doubling return at once
    H_S_ACTIVE,       // 0: this doubling was enqueued speculatively and must not run
    H_KEY_HI, H_KEY_LO,  // threefry key of the subtree (as doubles holding u32)
    H_CUR_PE,         // potential of the current state (start of transition)
    H_STOP,           // 1: the tree is complete (turning / diverging / max depth)
    H_T_AUX0, H_T_AUX1, H_T_AUX2, H_T_AUX3,
    H_S_AUX0, H_S_AUX1, H_S_AUX2, H_S_AUX3,
    H_LEAF_AUX0, H_LEAF_AUX1, H_LEAF_AUX2, H_LEAF_AUX3,  // aux written by the evaluation
    H_LEAF_PE,        // potential written by the evaluation
    H_EVALS,          // evaluations actually performed (counter)
    H_N = 48
};

// ---- vectors of length D, in this order after the header
enum {
    V_INVM = 0,
    V_Z, V_G,                      // current state of the chain (start of the transition)
    V_ZN,                          // position of the NEXT evaluation (read by dc_eval)
    V_RH,                          // momentum after the first half step of that leapfrog
    V_GRAD,                        // gradient written by the evaluation
    V_TL_Z, V_TL_R, V_TL_G, V_TR_Z, V_TR_R, V_TR_G, V_TP_Z, V_TP_G, V_T_RSUM,   // tree
    V_SL_Z, V_SL_R, V_SL_G, V_SR_Z, V_SR_R, V_SR_G, V_SP_Z, V_SP_G, V_S_RSUM,   // subtree
    V_CKPT                         // 2 * max_depth vectors: r_ckpts | r_sum_ckpts
};

__host__ __device__ inline size_t ns_doubles(int D, int max_depth) {
    return (size_t)H_N + (size_t)(V_CKPT + 2 * max_depth) * D;
}
__host__ __device__ inline double* vec(double* ns, int D, int which) {
    return ns + H_N + (size_t)which * D;
}

// ---------------------------------------------------------------- threefry (device)
__device__ __forceinline__ uint32_t rotl32(uint32_t v, int r) { return (v << r) | (v >> (32 - r)); }
__device__ inline void tf_block(uint32_t k0, uint32_t k1, uint32_t c0, uint32_t c1, uint32_t* o0,
                                uint32_t* o1) {
    const int R0[4] = {13, 15, 26, 6}, R1[4] = {17, 29, 16, 24};
    const uint32_t ks[3] = {k0, k1, k0 ^ k1 ^ 0x1BD11BDAu};
    uint32_t x0 = c0 + ks[0], x1 = c1 + ks[1];
#pragma unroll
    for (int g = 0; g < 5; ++g) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            x0 += x1;
            x1 = rotl32(x1, (g & 1) ? R1[i] : R0[i]);
            x1 ^= x0;
        }
        x0 += ks[(g + 1) % 3];
        x1 += ks[(g + 2) % 3] + (uint32_t)(g + 1);
    }
    *o0 = x0;
    *o1 = x1;
}
// jax.random.split(key, 2): counts [0,1,2,3] -> blocks (0,2), (1,3)
__device__ inline void tf_split2(uint32_t khi, uint32_t klo, uint32_t* a_hi, uint32_t* a_lo,
                                 uint32_t* b_hi, uint32_t* b_lo) {
    uint32_t p0, q0, p1, q1;
    tf_block(khi, klo, 0u, 2u, &p0, &q0);
    tf_block(khi, klo, 1u, 3u, &p1, &q1);
    *a_hi = p0; *a_lo = p1;
    *b_hi = q0; *b_lo = q1;
}
// jax.random.bernoulli(key, p): uniform(key, ()) < p, float32 mantissa trick
__device__ inline float tf_uniform_f32(uint32_t khi, uint32_t klo) {
    uint32_t b0, b1;
    tf_block(khi, klo, 0u, 0u, &b0, &b1);
    return __uint_as_float((b0 >> 9) | 0x3F800000u) - 1.0f;
}
__device__ inline bool tf_bernoulli(uint32_t khi, uint32_t klo, double p) {
    return (double)tf_uniform_f32(khi, klo) < p;
}

// ---------------------------------------------------------------- wave helpers
// Wave-uniform sum over the 64 lanes by DPP (row shuffles + row broadcasts; no LDS crossbar
// round trips -- six dependent ds_bpermute pairs cost ~0.2 us on the chain's serial path).
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double nd_dpp_add(double v) {
    const long long x = __double_as_longlong(v);
    const int lo = __builtin_amdgcn_update_dpp(0, (int)x, CTRL, ROW_MASK, 0xF, false);
    const int hi = __builtin_amdgcn_update_dpp(0, (int)(x >> 32), CTRL, ROW_MASK, 0xF, false);
    return v + __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}
__device__ __forceinline__ double nd_readlane63(double v) {
    const long long x = __double_as_longlong(v);
    const int lo = __builtin_amdgcn_readlane((int)x, 63);
    const int hi = __builtin_amdgcn_readlane((int)(x >> 32), 63);
    return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}
// (the DPP steps written out -- wave_reduce.hip.h: no zeroed temporaries, two chains interleaved)
__device__ __forceinline__ double nd_wave_sum(double v) {
    wr::wave_reduce_sum1_f64_raw(v);
    return nd_readlane63(v);
}
__device__ __forceinline__ void nd_wave_sum2(double& a, double& b) {
    wr::wave_reduce_sum2_f64_raw(a, b);
    a = nd_readlane63(a);
    b = nd_readlane63(b);
}
// The tree bookkeeping below is written for a "team" of NT threads working on one chain's
// state: one wave (NT = 64, no barriers: the leaf's wave inside dc_eval, or a 64-thread launch)
// or a whole workgroup (NT > 64, for latent vectors of 10^4..10^5 entries: kw_leaf).  Thread
// `tid` owns elements tid, tid + NT, ... in every loop, so element-wise read-after-write
// between loops needs no synchronisation; header words are written by thread 0 only.
// scr: LDS scratch of NT/64 doubles (unused for NT = 64).
template <int NT>
__device__ __forceinline__ double team_sum(double v, int tid, double* scr) {
    v = nd_wave_sum(v);
    if (NT == 64) return v;
    __syncthreads();  // (scr may still be read from the previous sum)
    if ((tid & 63) == 0) scr[tid >> 6] = v;
    __syncthreads();
    double s = 0.0;
#pragma unroll
    for (int w = 0; w < NT / 64; ++w) s += scr[w];
    return s;
}
// stores of the team visible to its later loads (header words written by thread 0, vectors)
// (A workgroup team -- NT > 64 -- lives on ONE CU: its waves share the vector L1, so workgroup scope
// is all this needs.  Agent scope here was an L2 write-back of every dirty line of the chain's state
// and an invalidate that sent the next pass over the vectors back to memory -- with the dynamic
// model's 35 502-entry vectors ~200 us per chain advance, profiles/r03/dynamic_advance.txt.)
template <int NT>
__device__ __forceinline__ void team_sync() {
    if (NT > 64) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __syncthreads();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        return;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    __builtin_amdgcn_s_waitcnt(0);
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
}
// ---- teams.  The tree bookkeeping is written once, for a TEAM of threads that work on one chain's
// state; a team says which elements a thread owns (first(), stride()), who writes header words
// (leader()), how scalars of the state are read and written (ld / st), how it adds (sum, sum2) and
// how it synchronises (sync; readers_done in front of a rewrite of scalars every thread has read).
//   LocalTeam<64>   one wave (the leaf's wave inside dc_eval, the 64-thread launches): no barriers
//   LocalTeam<NT>   one workgroup
//   GridTeam        (below, with the wide leaf) every workgroup of a launch's grid row
template <int NT>
struct LocalTeam {
    static constexpr int UNROLL = NT > 64 ? 4 : 1;
    int tid;
    double* scr;
    __device__ __forceinline__ int first() const { return tid; }
    __device__ __forceinline__ int stride() const { return NT; }
    __device__ __forceinline__ bool leader() const { return tid == 0; }
    __device__ __forceinline__ double ld(const double* p) const { return *p; }
    __device__ __forceinline__ void st(double* p, double v) const { *p = v; }
    __device__ __forceinline__ double sum(double v) { return team_sum<NT>(v, tid, scr); }
    __device__ __forceinline__ void sum2(double& a, double& b) {
        a = team_sum<NT>(a, tid, scr);
        b = team_sum<NT>(b, tid, scr);
    }
    __device__ __forceinline__ void sync() { team_sync<NT>(); }
    __device__ __forceinline__ void readers_done() {
        if (NT > 64) __syncthreads();
    }
};
// numpyro _is_turning with a diagonal inverse mass matrix
template <int NT = 64>
__device__ inline bool is_turning(const double* invM, const double* r_left, const double* r_right,
                                  const double* r_sum, int D, int tid, double* scr = nullptr) {
    double dl = 0.0, dr = 0.0;
#pragma clang loop unroll_count(NT > 64 ? 4 : 1)
    for (int i = tid; i < D; i += NT) {
        const double rs = r_sum[i] - 0.5 * (r_left[i] + r_right[i]);
        dl += invM[i] * r_left[i] * rs;
        dr += invM[i] * r_right[i] * rs;
    }
    dl = team_sum<NT>(dl, tid, scr);
    dr = team_sum<NT>(dr, tid, scr);
    return (dl <= 0.0) | (dr <= 0.0);
}
// (restrict + unroll: the loads of a batch are in flight together; a loop of load -> store
// pairs on possibly aliasing pointers is one memory round trip per element)
template <int NT = 64>
__device__ __forceinline__ void vcopy(double* __restrict__ dst, const double* __restrict__ src, int D,
                                      int tid) {
#pragma clang loop unroll_count(NT > 64 ? 4 : 1)
    for (int i = tid; i < D; i += NT) dst[i] = src[i];
}
__device__ __forceinline__ double logaddexp(double a, double b) {
    if (a == b) return a + 0.6931471805599453;
    const double m = fmax(a, b);
    return m + log1p(exp(-fabs(a - b)));
}

// ---------------------------------------------------------------- leaf (one wave)
// Called by the wave that has just computed potential, aux and gradient of the position
// V_ZN (reached with half-stepped momentum V_RH); it hands them over in LDS:
// gL = grad[D] | potential | aux[4].
//
// The leaf sits on the serial path of the chain (the next evaluation needs the position it
// writes), so it is organised by memory round trips, not by statement order: (A) the whole
// header (one word per lane) and every vector it reads are requested at once; (B) all
// arithmetic runs from registers -- including the first checkpoint of an odd leaf, see
// LeafState; (C) only a leaf that closes several subtrees reads deeper checkpoints, one
// dependent round per further level; (D) stores.
// NE = vector elements per lane: 1 for D <= 64 (a third of the instructions), 4 for D <= 256
constexpr int LEAF_NE_MAX = 4;
__device__ __forceinline__ double hdr_word(double hv, int k) {  // k wave-uniform constant
    const long long x = __double_as_longlong(hv);
    const int lo = __builtin_amdgcn_readlane((int)x, k);
    const int hi = __builtin_amdgcn_readlane((int)(x >> 32), k);
    return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}
// everything the leaf reads from the state buffer, requested before the gradient exists.
// ONE round, nothing depends on the header: the first checkpoint an odd leaf compares against
// is the one the previous (even) leaf wrote -- that leaf's momentum (still in V_SL_R / V_SR_R,
// by direction) and the running sum it left in V_S_RSUM -- so it needs no indexed load.
template <int LEAF_NE>
struct LeafState {
    double hv;                      // header word `lane`
    double invM[LEAF_NE], zn[LEAF_NE], r[LEAF_NE], rs[LEAF_NE];
    double sl_r[LEAF_NE], sr_r[LEAF_NE];
    // filled by leaf_prepare (header-dependent, no memory access)
    double c_r[LEAF_NE], c_s[LEAF_NE];   // first checkpoint an odd leaf compares against
    int num, idx_max, idx_min;
    uint32_t nhi, nlo;              // the subtree's next rng key
    float u_take;                   // uniform of this leaf's transition bernoulli
};
template <int LEAF_NE>
__device__ __forceinline__ LeafState<LEAF_NE> leaf_prefetch(double* ns, int D, int max_depth, int lane) {
    LeafState<LEAF_NE> S;
    S.hv = lane < H_N ? ns[lane] : 0.0;
    const double* p_invM = vec(ns, D, V_INVM);
    const double* p_zn = vec(ns, D, V_ZN);
    const double* p_rh = vec(ns, D, V_RH);
    const double* p_rsum = vec(ns, D, V_S_RSUM);
    const double* p_slr = vec(ns, D, V_SL_R);
    const double* p_srr = vec(ns, D, V_SR_R);
#pragma unroll
    for (int e = 0; e < LEAF_NE; ++e) {
        const int i = lane + 64 * e;
        const bool ok = i < D;
        S.invM[e] = ok ? p_invM[i] : 0.0;
        S.zn[e] = ok ? p_zn[i] : 0.0;
        S.r[e] = ok ? p_rh[i] : 0.0;
        S.rs[e] = ok ? p_rsum[i] : 0.0;
        S.sl_r[e] = ok ? p_slr[i] : 0.0;
        S.sr_r[e] = ok ? p_srr[i] : 0.0;
    }
    return S;
}
// header-dependent preparation, registers only: leaf index -> checkpoint indices (numpyro
// _leaf_idx_to_ckpt_idxs), the first checkpoint, and the leaf's random numbers (which depend
// on the key alone).  The caller runs this wherever the leaf wave would otherwise idle.
__device__ __forceinline__ void leaf_rng_k(uint32_t khi, uint32_t klo, uint32_t* nhi, uint32_t* nlo,
                                           float* u_take) {
    uint32_t thi, tlo;
    tf_split2(khi, klo, nhi, nlo, &thi, &tlo);  // rng_key, transition_rng_key = split(rng_key)
    *u_take = tf_uniform_f32(thi, tlo);
}
__device__ __forceinline__ void leaf_rng(double hv, uint32_t* nhi, uint32_t* nlo, float* u_take) {
    const uint32_t khi = (uint32_t)hdr_word(hv, H_KEY_HI), klo = (uint32_t)hdr_word(hv, H_KEY_LO);
    uint32_t thi, tlo;
    tf_split2(khi, klo, nhi, nlo, &thi, &tlo);  // rng_key, transition_rng_key = split(rng_key)
    *u_take = tf_uniform_f32(thi, tlo);
}
template <bool DRAW, int LEAF_NE>
__device__ __forceinline__ void leaf_prepare(LeafState<LEAF_NE>& S) {
    S.num = (int)hdr_word(S.hv, H_S_NUM);  // leaves so far = index of this leaf
    S.idx_max = __popc((unsigned)S.num >> 1);
    const int trail = __ffs(~S.num) - 1;   // trailing one bits
    S.idx_min = S.idx_max - trail + 1;
    const bool right = hdr_word(S.hv, H_DIR) > 0.0;
#pragma unroll
    for (int e = 0; e < LEAF_NE; ++e) {
        S.c_r[e] = right ? S.sr_r[e] : S.sl_r[e];
        S.c_s[e] = S.rs[e];
    }
    if (DRAW) leaf_rng(S.hv, &S.nhi, &S.nlo, &S.u_take);  // (else: the caller fills them in)
}

// The leaf is instruction-issue bound on one wave (~1.3 us), so it is cut into two halves
// that need nothing from each other and run on two waves of the tail workgroup at once:
//   leaf_moves    integrator and tree geometry: full-step momentum, divergence test, running
//                 momentum sum, checkpointed U-turn test, `done`, the NEXT position, edges
//                 and checkpoints of the subtree
//   leaf_weights  the subtree's multinomial bookkeeping: exp/log1p/expit of the energy error,
//                 the transition bernoulli, the proposal, the rng key
// Both recompute the (cheap) kinetic energy with the same instruction sequence.  They write
// disjoint header words and vectors.  nuts_leaf runs them back to back on one wave.
struct LeafEnergy {
    double e_new, delta;
};
template <int LEAF_NE>
__device__ __forceinline__ LeafEnergy leaf_energy(const LeafState<LEAF_NE>& S, int D, int lane,
                                                  const double* gL, double eps, double (&r)[LEAF_NE],
                                                  double (&g)[LEAF_NE]) {
    double kin = 0.0;
#pragma unroll
    for (int e = 0; e < LEAF_NE; ++e) {
        const int i = lane + 64 * e;
        g[e] = i < D ? gL[i] : 0.0;
        r[e] = S.r[e] - 0.5 * eps * g[e];  // the leaf's full-step momentum
        kin += S.invM[e] * r[e] * r[e];
    }
    kin = 0.5 * nd_wave_sum(kin);
    LeafEnergy E;
    E.e_new = gL[D] + kin;
    E.delta = E.e_new - hdr_word(S.hv, H_E0);
    if (E.delta != E.delta) E.delta = __builtin_inf();
    return E;
}
// the scalar half of _combine_tree(current, leaf, biased_transition=False) for one leaf
struct LeafWeights {
    bool take;
    double w_sub, sum_acc;
};
// (w_cur, s_sumacc: the subtree's weight and accept sum so far; only read when num != 0)
template <class F>
__device__ __forceinline__ LeafWeights leaf_weigh_f(int num, double delta, float u_take, F hdr) {
    // (the short exp / log1p / reciprocal of lean_math.hip.h, 1-2 ulp: this chain sits on the leaf's serial
    // path, and the device library's versions are 100-250 dependent instructions each)
    const double w_leaf = -delta;
    const double acc_leaf = fmin(1.0, dc::lean::exp(-delta));
    LeafWeights W{true, w_leaf, acc_leaf};
    if (num != 0) {
        const double w_cur = hdr(H_S_WEIGHT);
        // expit(d) for the uniform transition and logaddexp(w_cur, w_leaf) share one exp
        const double d = w_leaf - w_cur;
        const double ex = dc::lean::exp(-fabs(d));
        const double prob = (d >= 0.0 ? 1.0 : ex) * dc::lean::rcp(1.0 + ex);
        W.take = (double)u_take < prob;
        W.w_sub = w_cur == w_leaf ? w_cur + 0.6931471805599453 : fmax(w_cur, w_leaf) + dc::lean::log1p_pos(ex);
        W.sum_acc = hdr(H_S_SUMACC) + acc_leaf;
    }
    return W;
}
__device__ __forceinline__ LeafWeights leaf_weigh(double hv, int num, double delta, float u_take) {
    return leaf_weigh_f(num, delta, u_take, [hv](int k) { return hdr_word(hv, k); });
}
__device__ __forceinline__ void leaf_weights_header(double* ns, const double* gL, int D, const LeafWeights& W,
                                                    double e_new, uint32_t nhi, uint32_t nlo) {
    ns[H_S_WEIGHT] = W.w_sub;
    ns[H_S_SUMACC] = W.sum_acc;
    ns[H_KEY_HI] = (double)nhi;
    ns[H_KEY_LO] = (double)nlo;
    if (W.take) {
        ns[H_S_PE] = gL[D];
        ns[H_S_EPROP] = e_new;
        ns[H_S_AUX0] = gL[D + 1]; ns[H_S_AUX1] = gL[D + 2];
        ns[H_S_AUX2] = gL[D + 3]; ns[H_S_AUX3] = gL[D + 4];
    }
}
__device__ __forceinline__ void leaf_moves_header(double* ns, double hv, int new_num, bool div_leaf,
                                                  bool turning, bool done) {
    ns[H_S_NUM] = (double)new_num;
    ns[H_S_DIV] = div_leaf ? 1.0 : 0.0;
    ns[H_S_TURN] = turning ? 1.0 : 0.0;
    ns[H_S_DONE] = done ? 1.0 : 0.0;
    ns[H_EVALS] = hdr_word(hv, H_EVALS) + 1.0;
}

// S needs hv, invM, zn, r (half-stepped) and the rng fields
struct NoSync {
    __device__ __forceinline__ void operator()(bool) const {}
};
// SYNC: called once between the decisions and the stores, with leaf_moves' verdict (subtree complete) -- the
// persistent kernel books a leaf on waves of their own across one of the workgroup's barriers
// (dc_kernels.hip.h leaf_window_*)
template <int LEAF_NE, class SYNC = NoSync>
__device__ inline LeafWeights leaf_weights(double* ns, int D, int lane, const double* gL,
                                           const LeafState<LEAF_NE>& S, const SYNC sync = SYNC()) {
    const double eps = hdr_word(S.hv, H_EPS) * hdr_word(S.hv, H_DIR);
    double r[LEAF_NE], g[LEAF_NE];
    const LeafEnergy E = leaf_energy(S, D, lane, gL, eps, r, g);
    const LeafWeights W = leaf_weigh(S.hv, (int)hdr_word(S.hv, H_S_NUM), E.delta, S.u_take);
    sync(false);
    if (W.take) {
        double* sp_z = vec(ns, D, V_SP_Z); double* sp_g = vec(ns, D, V_SP_G);
#pragma unroll
        for (int e = 0; e < LEAF_NE; ++e) {
            const int i = lane + 64 * e;
            if (i < D) { sp_z[i] = S.zn[e]; sp_g[i] = g[e]; }
        }
    }
    if (lane == 0) leaf_weights_header(ns, gL, D, W, E.e_new, S.nhi, S.nlo);
    return W;
}

// where the next leapfrog starts while the subtree goes on: the expressions of leaf_energy / leaf_moves
// (second half step of this leaf, first half step of the next, full position step), for the caller that
// publishes the position BEFORE the leaf is booked (dc_kernels.hip.h tail_waves, defer_out)
__device__ __forceinline__ double leaf_next_position(double zn, double invM, double r_half, double eps, double g) {
    const double r = r_half - 0.5 * eps * g;
    const double rn = r - 0.5 * eps * g;
    return zn + eps * invM * rn;
}
// S fully prefetched and prepared; returns (wave uniform) whether the subtree is complete
template <int LEAF_NE, class SYNC = NoSync>
__device__ inline bool leaf_moves(double* ns, int D, int max_depth, int lane, const double* gL,
                                  const LeafState<LEAF_NE>& S, double* zn_lds = nullptr, const SYNC sync = SYNC()) {
    const double hv = S.hv;
    const double h_eps = hdr_word(hv, H_EPS), h_dir = hdr_word(hv, H_DIR);
    const double eps = h_eps * h_dir;
    const bool going_right = h_dir > 0.0;
    const int num = S.num, idx_max = S.idx_max, idx_min = S.idx_min;
    double* ck_r = vec(ns, D, V_CKPT);
    double* ck_s = ck_r + (size_t)max_depth * D;

    // ---- (B) second half step, kinetic energy -> divergence
    double r[LEAF_NE], g[LEAF_NE], rs[LEAF_NE], c_r[LEAF_NE], c_s[LEAF_NE];
    const LeafEnergy E = leaf_energy(S, D, lane, gL, eps, r, g);
    const bool div_leaf = E.delta > hdr_word(hv, H_MAXDE);
#pragma unroll
    for (int e = 0; e < LEAF_NE; ++e) {
        rs[e] = num == 0 ? r[e] : S.rs[e] + r[e];  // _combine_tree: r_sum
        c_r[e] = S.c_r[e]; c_s[e] = S.c_s[e];
    }
    // checkpointed U-turn test (numpyro _is_iterative_turning)
    bool turning = false;
    for (int ci = idx_max; ci >= idx_min && !turning; --ci) {
        if (ci != idx_max) {  // (a leaf closing several subtrees: one more round per level)
#pragma unroll
            for (int e = 0; e < LEAF_NE; ++e) {
                const int i = lane + 64 * e;
                c_r[e] = i < D ? ck_r[(size_t)ci * D + i] : 0.0;
                c_s[e] = i < D ? ck_s[(size_t)ci * D + i] : 0.0;
            }
        }
        double dl = 0.0, dr = 0.0;
#pragma unroll
        for (int e = 0; e < LEAF_NE; ++e) {
            const double sub = rs[e] - c_s[e] + c_r[e];
            const double rsm = sub - 0.5 * (c_r[e] + r[e]);
            dl += S.invM[e] * c_r[e] * rsm;
            dr += S.invM[e] * r[e] * rsm;
        }
        nd_wave_sum2(dl, dr);
        turning = (dl <= 0.0) | (dr <= 0.0);
    }
    const int new_num = num + 1;
    const bool done = turning || div_leaf || new_num >= (int)hdr_word(hv, H_S_MAX);

    sync(done);
    // ---- (D) stores
    double* p_zn = vec(ns, D, V_ZN); double* p_rh = vec(ns, D, V_RH); double* p_rsum = vec(ns, D, V_S_RSUM);
    double* sl_z = vec(ns, D, V_SL_Z); double* sl_r = vec(ns, D, V_SL_R); double* sl_g = vec(ns, D, V_SL_G);
    double* sr_z = vec(ns, D, V_SR_Z); double* sr_r = vec(ns, D, V_SR_R); double* sr_g = vec(ns, D, V_SR_G);
    const bool wl = num == 0 || !going_right, wr = num == 0 || going_right;
    const bool wck = (num & 1) == 0;
#pragma unroll
    for (int e = 0; e < LEAF_NE; ++e) {
        const int i = lane + 64 * e;
        if (i < D) {
            // next leapfrog starts from this leaf (the subtree grows in one direction)
            const double zn = S.zn[e];
            const double rn = r[e] - 0.5 * eps * g[e];
            const double z_next = done ? zn : zn + eps * S.invM[e] * rn;
            p_zn[i] = z_next;
            if (zn_lds) zn_lds[i] = z_next;   // (the caller's LDS copy of the next position)
            p_rh[i] = done ? r[e] : rn;
            p_rsum[i] = rs[e];
            if (wl) { sl_z[i] = zn; sl_r[i] = r[e]; sl_g[i] = g[e]; }
            if (wr) { sr_z[i] = zn; sr_r[i] = r[e]; sr_g[i] = g[e]; }
            if (wck) {
                ck_r[(size_t)idx_max * D + i] = r[e];
                ck_s[(size_t)idx_max * D + i] = rs[e];
            }
        }
    }
    if (lane == 0) leaf_moves_header(ns, hv, new_num, div_leaf, turning, done);
    return done;
}

// ---- the NEXT leaf's state from this one's, in registers (persistent kernel: while a subtree goes on,
// what leaf_prefetch would load after leaf_moves / leaf_weights have stored is known already -- V_ZN,
// V_RH, V_S_RSUM, V_SL_R / V_SR_R and the header words each half reads).  Same expressions as the stores
// of leaf_moves (r, rn, z_next, rs) and of the two header writers, hence the same bits.  Each of the two
// leaf waves forwards ITS copy: the words the other half writes and this one never reads stay stale.
__device__ __forceinline__ double hdr_patch(double hv, int lane, int k, double v) { return lane == k ? v : hv; }
template <int LEAF_NE>
__device__ __forceinline__ void leaf_forward_moves(LeafState<LEAF_NE>& S, int D, int lane, const double* gL) {
    const double h_dir = hdr_word(S.hv, H_DIR);
    const double eps = hdr_word(S.hv, H_EPS) * h_dir;
    const bool going_right = h_dir > 0.0;
    const int num = (int)hdr_word(S.hv, H_S_NUM);
    const bool wl = num == 0 || !going_right, wr = num == 0 || going_right;
#pragma unroll
    for (int e = 0; e < LEAF_NE; ++e) {
        const int i = lane + 64 * e;
        const double g = i < D ? gL[i] : 0.0;
        const double r = S.r[e] - 0.5 * eps * g;
        const double rn = r - 0.5 * eps * g;
        S.zn[e] = S.zn[e] + eps * S.invM[e] * rn;
        S.rs[e] = num == 0 ? r : S.rs[e] + r;
        S.r[e] = rn;
        if (wl) S.sl_r[e] = r;
        if (wr) S.sr_r[e] = r;
    }
    S.hv = hdr_patch(hdr_patch(S.hv, lane, H_S_NUM, (double)(num + 1)), lane, H_EVALS, hdr_word(S.hv, H_EVALS) + 1.0);
}
template <int LEAF_NE>
__device__ __forceinline__ void leaf_forward_weights(LeafState<LEAF_NE>& S, int D, int lane, const double* gL,
                                                     const LeafWeights& W) {
    const double eps = hdr_word(S.hv, H_EPS) * hdr_word(S.hv, H_DIR);
    const int num = (int)hdr_word(S.hv, H_S_NUM);
#pragma unroll
    for (int e = 0; e < LEAF_NE; ++e) {
        const int i = lane + 64 * e;
        const double g = i < D ? gL[i] : 0.0;
        const double r = S.r[e] - 0.5 * eps * g;
        const double rn = r - 0.5 * eps * g;
        S.zn[e] = S.zn[e] + eps * S.invM[e] * rn;
        S.r[e] = rn;
    }
    double hv = hdr_patch(S.hv, lane, H_S_NUM, (double)(num + 1));
    hv = hdr_patch(hv, lane, H_S_WEIGHT, W.w_sub);
    hv = hdr_patch(hv, lane, H_S_SUMACC, W.sum_acc);
    hv = hdr_patch(hv, lane, H_KEY_HI, (double)S.nhi);
    S.hv = hdr_patch(hv, lane, H_KEY_LO, (double)S.nlo);
}

// both halves on one wave (the leaf as its own launch: kp_leaf)
template <int LEAF_NE>
__device__ inline bool nuts_leaf(double* ns, int D, int max_depth, int lane, const double* gL,
                                 const LeafState<LEAF_NE>& S) {
    const bool done = leaf_moves(ns, D, max_depth, lane, gL, S);
    leaf_weights(ns, D, lane, gL, S);
    return done;
}

// The same two halves for D > 64 inside dc_eval's tail, with the vectors staged in LDS
// instead of registers (a register-resident LeafState<4> pushes dc_eval past 128 VGPRs, i.e.
// down to one workgroup per CU -- the cliff that matters when several chains share a GPU):
//   stg = invM[D] | zn[D] | r_half[D] | r_sum[D] | sl_r[D] | sr_r[D] | r_full[D]
// (the first six copied by the tail's idle waves, the last written by leaf_moves_staged).
// Lane l owns elements l, l+64, ...: it alone reads and rewrites them between the passes of
// one half, so no barrier is needed; leaf_weights_staged only reads what the copy wrote.
// Per-lane summation order and arithmetic equal the LeafState<LEAF_NE_MAX> register version,
// so both walk the same trajectory bit for bit.
constexpr int LEAF_STAGE_LOADS = 6;   // vectors copied from the state buffer
constexpr int LEAF_STAGE_VECS = 7;    // + the full-step momentum
__device__ __forceinline__ LeafEnergy leaf_energy_staged(const double* stg, double hv, int D, int lane,
                                                         const double* gL, double eps, double* r_out) {
    const double* s_invM = stg;
    const double* s_rh = stg + 2 * D;
    double kin = 0.0;
    for (int i = lane; i < D; i += 64) {
        const double r = s_rh[i] - 0.5 * eps * gL[i];
        if (r_out) r_out[i] = r;
        kin += s_invM[i] * r * r;
    }
    kin = 0.5 * nd_wave_sum(kin);
    LeafEnergy E;
    E.e_new = gL[D] + kin;
    E.delta = E.e_new - hdr_word(hv, H_E0);
    if (E.delta != E.delta) E.delta = __builtin_inf();
    return E;
}
__device__ inline void leaf_weights_staged(double* ns, int D, int lane, const double* gL, const double* stg,
                                           double hv, uint32_t nhi, uint32_t nlo, float u_take) {
    const double eps = hdr_word(hv, H_EPS) * hdr_word(hv, H_DIR);
    const LeafEnergy E = leaf_energy_staged(stg, hv, D, lane, gL, eps, nullptr);
    const LeafWeights W = leaf_weigh(hv, (int)hdr_word(hv, H_S_NUM), E.delta, u_take);
    if (W.take) {
        const double* s_zn = stg + D;
        double* sp_z = vec(ns, D, V_SP_Z); double* sp_g = vec(ns, D, V_SP_G);
        for (int i = lane; i < D; i += 64) { sp_z[i] = s_zn[i]; sp_g[i] = gL[i]; }
    }
    if (lane == 0) leaf_weights_header(ns, gL, D, W, E.e_new, nhi, nlo);
}
__device__ inline bool leaf_moves_staged(double* ns, int D, int max_depth, int lane, const double* gL,
                                         double* stg, double hv, double* zn_lds = nullptr) {
    const double* s_invM = stg;
    const double* s_zn = stg + D;
    double* s_rs = stg + 3 * D;   // running sum: before / after this leaf
    const double* s_slr = stg + 4 * D;
    const double* s_srr = stg + 5 * D;
    double* s_r = stg + 6 * D;    // full-step momentum
    const double h_eps = hdr_word(hv, H_EPS), h_dir = hdr_word(hv, H_DIR);
    const double eps = h_eps * h_dir;
    const bool going_right = h_dir > 0.0;
    const int num = (int)hdr_word(hv, H_S_NUM);
    const int idx_max = __popc((unsigned)num >> 1);
    const int idx_min = idx_max - (__ffs(~num) - 1) + 1;
    double* ck_r = vec(ns, D, V_CKPT);
    double* ck_s = ck_r + (size_t)max_depth * D;

    // ---- pass 1: second half step, kinetic energy -> divergence
    const LeafEnergy E = leaf_energy_staged(stg, hv, D, lane, gL, eps, s_r);
    const bool div_leaf = E.delta > hdr_word(hv, H_MAXDE);

    // ---- pass 2: running sum; an odd leaf's first checkpoint is the previous leaf (LeafState)
    const bool cmp = idx_max >= idx_min;
    const double* c_r1 = going_right ? s_srr : s_slr;
    double dl = 0.0, dr = 0.0;
    for (int i = lane; i < D; i += 64) {
        const double r = s_r[i], rs_old = s_rs[i];
        const double rs = num == 0 ? r : rs_old + r;
        s_rs[i] = rs;
        if (cmp) {
            const double c_r = c_r1[i];
            const double sub = rs - rs_old + c_r;
            const double rsm = sub - 0.5 * (c_r + r);
            dl += s_invM[i] * c_r * rsm;
            dr += s_invM[i] * r * rsm;
        }
    }
    bool turning = false;
    if (cmp) {
        nd_wave_sum2(dl, dr);
        turning = (dl <= 0.0) | (dr <= 0.0);
    }
    for (int ci = idx_max - 1; ci >= idx_min && !turning; --ci) {  // deeper levels: from memory
        dl = 0.0; dr = 0.0;
        for (int i = lane; i < D; i += 64) {
            const double c_r = ck_r[(size_t)ci * D + i], c_s = ck_s[(size_t)ci * D + i];
            const double r = s_r[i];
            const double sub = s_rs[i] - c_s + c_r;
            const double rsm = sub - 0.5 * (c_r + r);
            dl += s_invM[i] * c_r * rsm;
            dr += s_invM[i] * r * rsm;
        }
        nd_wave_sum2(dl, dr);
        turning = (dl <= 0.0) | (dr <= 0.0);
    }
    const int new_num = num + 1;
    const bool done = turning || div_leaf || new_num >= (int)hdr_word(hv, H_S_MAX);

    // ---- pass 3: stores
    double* p_zn = vec(ns, D, V_ZN); double* p_rh = vec(ns, D, V_RH); double* p_rsum = vec(ns, D, V_S_RSUM);
    double* sl_z = vec(ns, D, V_SL_Z); double* sl_r = vec(ns, D, V_SL_R); double* sl_g = vec(ns, D, V_SL_G);
    double* sr_z = vec(ns, D, V_SR_Z); double* sr_r = vec(ns, D, V_SR_R); double* sr_g = vec(ns, D, V_SR_G);
    const bool wl = num == 0 || !going_right, wr = num == 0 || going_right;
    const bool wck = (num & 1) == 0;
    for (int i = lane; i < D; i += 64) {
        const double r = s_r[i], g = gL[i], zn = s_zn[i], rs = s_rs[i];
        const double rn = r - 0.5 * eps * g;
        const double z_next = done ? zn : zn + eps * s_invM[i] * rn;
        p_zn[i] = z_next;
        if (zn_lds) zn_lds[i] = z_next;   // (the caller's LDS copy of the next position)
        p_rh[i] = done ? r : rn;
        p_rsum[i] = rs;
        if (wl) { sl_z[i] = zn; sl_r[i] = r; sl_g[i] = g; }
        if (wr) { sr_z[i] = zn; sr_r[i] = r; sr_g[i] = g; }
        if (wck) {
            ck_r[(size_t)idx_max * D + i] = r;
            ck_s[(size_t)idx_max * D + i] = rs;
        }
    }
    if (lane == 0) leaf_moves_header(ns, hv, new_num, div_leaf, turning, done);
    return done;
}

// ---------------------------------------------------------------- small kernels (1 wave)

// start of a transition: tree = the current state with momentum r (uploaded to V_TL_R)
template <class Team>
__device__ inline void init_body_t(double* ns, int D, double eps, double max_de, Team& tm) {
    // one pass: every element is loaded once (four loads in flight) and fanned out -- a chain of
    // copy loops was one dependent memory round trip each on the chain's serial path
    const double* __restrict__ invM = vec(ns, D, V_INVM);
    const double* __restrict__ r = vec(ns, D, V_TL_R);
    const double* __restrict__ zc = vec(ns, D, V_Z);
    const double* __restrict__ gc = vec(ns, D, V_G);
    double* __restrict__ tl_z = vec(ns, D, V_TL_Z); double* __restrict__ tl_g = vec(ns, D, V_TL_G);
    double* __restrict__ tr_z = vec(ns, D, V_TR_Z); double* __restrict__ tr_r = vec(ns, D, V_TR_R);
    double* __restrict__ tr_g = vec(ns, D, V_TR_G); double* __restrict__ tp_z = vec(ns, D, V_TP_Z);
    double* __restrict__ tp_g = vec(ns, D, V_TP_G); double* __restrict__ t_rs = vec(ns, D, V_T_RSUM);
    double kin = 0.0;
#pragma clang loop unroll_count(Team::UNROLL)
    for (int i = tm.first(); i < D; i += tm.stride()) {
        const double ri = r[i], zi = zc[i], gi = gc[i];
        kin += invM[i] * ri * ri;
        tl_z[i] = zi; tl_g[i] = gi;
        tr_z[i] = zi; tr_r[i] = ri; tr_g[i] = gi;
        tp_z[i] = zi; tp_g[i] = gi;
        t_rs[i] = ri;
    }
    kin = 0.5 * tm.sum(kin);
    if (tm.leader()) {
        const double cur_pe = tm.ld(&ns[H_CUR_PE]);
        const double e0 = cur_pe + kin;
        tm.st(&ns[H_EPS], eps);
        tm.st(&ns[H_MAXDE], max_de);
        tm.st(&ns[H_E0], e0);
        tm.st(&ns[H_T_DEPTH], 0.0); tm.st(&ns[H_T_WEIGHT], 0.0); tm.st(&ns[H_T_TURN], 0.0); tm.st(&ns[H_T_DIV], 0.0);
        tm.st(&ns[H_T_SUMACC], 0.0); tm.st(&ns[H_T_NUM], 0.0); tm.st(&ns[H_T_PE], cur_pe); tm.st(&ns[H_T_EPROP], e0);
        tm.st(&ns[H_STOP], 0.0);
        tm.st(&ns[H_S_DONE], 1.0); tm.st(&ns[H_S_ACTIVE], 0.0);
    }
}
template <int NT = 64>
__device__ inline void init_body(double* ns, int D, double eps, double max_de, int lane,
                                 double* scr = nullptr) {
    LocalTeam<NT> tm{lane, scr};
    init_body_t(ns, D, eps, max_de, tm);
}

// start of doubling `j`: runs only if the tree is still at depth j and not finished
template <class Team>
__device__ inline void begin_body_t(double* ns, int D, int j, int going_right, uint32_t khi,
                                    uint32_t klo, Team& tm) {
    const bool active = tm.ld(&ns[H_STOP]) == 0.0 && (int)tm.ld(&ns[H_T_DEPTH]) == j;
    if (!active) {
        if (tm.leader()) { tm.st(&ns[H_S_ACTIVE], 0.0); tm.st(&ns[H_S_DONE], 1.0); }
        return;
    }
    const double dir = going_right ? 1.0 : -1.0;
    const double eps = tm.ld(&ns[H_EPS]) * dir;
    const double* __restrict__ invM = vec(ns, D, V_INVM);
    const double* __restrict__ ez = vec(ns, D, going_right ? V_TR_Z : V_TL_Z);
    const double* __restrict__ er = vec(ns, D, going_right ? V_TR_R : V_TL_R);
    const double* __restrict__ eg = vec(ns, D, going_right ? V_TR_G : V_TL_G);
    double* __restrict__ zn = vec(ns, D, V_ZN);
    double* __restrict__ rh = vec(ns, D, V_RH);
#pragma clang loop unroll_count(Team::UNROLL)
    for (int i = tm.first(); i < D; i += tm.stride()) {
        const double r = er[i] - 0.5 * eps * eg[i];
        rh[i] = r;
        zn[i] = ez[i] + eps * invM[i] * r;
    }
    if (tm.leader()) {
        tm.st(&ns[H_DIR], dir);
        tm.st(&ns[H_S_MAX], (double)(1 << j));
        tm.st(&ns[H_S_NUM], 0.0); tm.st(&ns[H_S_WEIGHT], 0.0); tm.st(&ns[H_S_TURN], 0.0); tm.st(&ns[H_S_DIV], 0.0);
        tm.st(&ns[H_S_SUMACC], 0.0);
        tm.st(&ns[H_S_DONE], 0.0); tm.st(&ns[H_S_ACTIVE], 1.0);
        tm.st(&ns[H_KEY_HI], (double)khi); tm.st(&ns[H_KEY_LO], (double)klo);
    }
}
template <int NT = 64>
__device__ inline void begin_body(double* ns, int D, int j, int going_right, uint32_t khi,
                                  uint32_t klo, int lane) {
    LocalTeam<NT> tm{lane, nullptr};
    begin_body_t(ns, D, j, going_right, khi, klo, tm);
}

// end of a doubling: _combine_tree(tree, subtree, biased_transition=True)
template <class Team>
__device__ inline void end_body_t(double* ns, int D, int max_depth, uint32_t thi, uint32_t tlo, Team& tm) {
    if (tm.ld(&ns[H_S_ACTIVE]) == 0.0) return;
    const bool going_right = tm.ld(&ns[H_DIR]) > 0.0;
    const bool s_turn = tm.ld(&ns[H_S_TURN]) != 0.0, s_div = tm.ld(&ns[H_S_DIV]) != 0.0;
    const double w_cur = tm.ld(&ns[H_T_WEIGHT]), w_new = tm.ld(&ns[H_S_WEIGHT]);
    double prob = exp(w_new - w_cur);
    if (s_turn || s_div) prob = 0.0;
    prob = fmin(prob, 1.0);
    const bool take = tf_bernoulli(thi, tlo, prob);
    // one pass over the vectors: the subtree's outer leaf becomes the tree's (edge vectors), the
    // momentum sums are added, the proposal is taken over when the biased transition says so, and
    // the U-turn test of the combined tree (numpyro _is_turning: left edge, right edge, r_sum)
    // accumulates -- same arithmetic per element as the separate loops this replaces
    const double* __restrict__ invM = vec(ns, D, V_INVM);
    const double* __restrict__ se_z = vec(ns, D, going_right ? V_SR_Z : V_SL_Z);
    const double* __restrict__ se_r = vec(ns, D, going_right ? V_SR_R : V_SL_R);
    const double* __restrict__ se_g = vec(ns, D, going_right ? V_SR_G : V_SL_G);
    const double* __restrict__ other_r = vec(ns, D, going_right ? V_TL_R : V_TR_R);  // the far edge stays
    const double* __restrict__ s_rsum = vec(ns, D, V_S_RSUM);
    const double* __restrict__ sp_z = vec(ns, D, V_SP_Z);
    const double* __restrict__ sp_g = vec(ns, D, V_SP_G);
    double* __restrict__ te_z = vec(ns, D, going_right ? V_TR_Z : V_TL_Z);
    double* __restrict__ te_r = vec(ns, D, going_right ? V_TR_R : V_TL_R);
    double* __restrict__ te_g = vec(ns, D, going_right ? V_TR_G : V_TL_G);
    double* __restrict__ t_rsum = vec(ns, D, V_T_RSUM);
    double* __restrict__ tp_z = vec(ns, D, V_TP_Z);
    double* __restrict__ tp_g = vec(ns, D, V_TP_G);
    double dl = 0.0, dr = 0.0;
#pragma clang loop unroll_count(Team::UNROLL)
    for (int i = tm.first(); i < D; i += tm.stride()) {
        const double ez = se_z[i], er = se_r[i], eg = se_g[i], orr = other_r[i];
        const double rsum = t_rsum[i] + s_rsum[i], im = invM[i];
        te_z[i] = ez; te_r[i] = er; te_g[i] = eg;
        t_rsum[i] = rsum;
        const double r_left = going_right ? orr : er, r_right = going_right ? er : orr;
        const double rs = rsum - 0.5 * (r_left + r_right);
        dl += im * r_left * rs;
        dr += im * r_right * rs;
    }
    // (the proposal in a loop of its own: with its four pointers in the loop above, the one-wave
    // instantiation -- a real call from dc_eval -- needed two registers more than dc_eval can give it)
    if (take) {
#pragma clang loop unroll_count(Team::UNROLL)
        for (int i = tm.first(); i < D; i += tm.stride()) {
            const double pz = sp_z[i], pg = sp_g[i];
            tp_z[i] = pz;
            tp_g[i] = pg;
        }
    }
    tm.sum2(dl, dr);
    const bool turning = s_turn | ((dl <= 0.0) | (dr <= 0.0));
    if (tm.leader()) {
        if (take) {
            tm.st(&ns[H_T_PE], tm.ld(&ns[H_S_PE]));
            tm.st(&ns[H_T_EPROP], tm.ld(&ns[H_S_EPROP]));
            tm.st(&ns[H_T_AUX0], tm.ld(&ns[H_S_AUX0])); tm.st(&ns[H_T_AUX1], tm.ld(&ns[H_S_AUX1]));
            tm.st(&ns[H_T_AUX2], tm.ld(&ns[H_S_AUX2])); tm.st(&ns[H_T_AUX3], tm.ld(&ns[H_S_AUX3]));
        }
        const double depth = tm.ld(&ns[H_T_DEPTH]) + 1.0;
        tm.st(&ns[H_T_DEPTH], depth);
        tm.st(&ns[H_T_WEIGHT], logaddexp(w_cur, w_new));
        tm.st(&ns[H_T_TURN], turning ? 1.0 : 0.0);
        tm.st(&ns[H_T_DIV], s_div ? 1.0 : 0.0);
        tm.st(&ns[H_T_SUMACC], tm.ld(&ns[H_T_SUMACC]) + tm.ld(&ns[H_S_SUMACC]));
        tm.st(&ns[H_T_NUM], tm.ld(&ns[H_T_NUM]) + tm.ld(&ns[H_S_NUM]));
        tm.st(&ns[H_STOP], (turning || s_div || (int)depth >= max_depth) ? 1.0 : 0.0);
        tm.st(&ns[H_S_ACTIVE], 0.0);
        tm.st(&ns[H_S_DONE], 1.0);
    }
}
template <int NT = 64>
__device__ inline void end_body(double* ns, int D, int max_depth, uint32_t thi, uint32_t tlo,
                                int lane, double* scr = nullptr) {
    LocalTeam<NT> tm{lane, scr};
    end_body_t(ns, D, max_depth, thi, tlo, tm);
}

// end of the transition: the proposal becomes the current state
template <class Team>
__device__ inline void finish_body_t(double* ns, int D, Team& tm) {
    const double* __restrict__ tp_z = vec(ns, D, V_TP_Z);
    const double* __restrict__ tp_g = vec(ns, D, V_TP_G);
    double* __restrict__ zc = vec(ns, D, V_Z);
    double* __restrict__ gc = vec(ns, D, V_G);
#pragma clang loop unroll_count(Team::UNROLL)
    for (int i = tm.first(); i < D; i += tm.stride()) {
        const double a = tp_z[i], b = tp_g[i];
        zc[i] = a;
        gc[i] = b;
    }
    if (tm.leader()) tm.st(&ns[H_CUR_PE], tm.ld(&ns[H_T_PE]));
}
template <int NT = 64>
__device__ inline void finish_body(double* ns, int D, int lane) {
    LocalTeam<NT> tm{lane, nullptr};
    finish_body_t(ns, D, tm);
}

// ---- one chain
__global__ __launch_bounds__(64) void k_init(double* ns, int D, double eps, double max_de) {
    init_body(ns, D, eps, max_de, threadIdx.x);
}
__global__ __launch_bounds__(64) void k_begin(double* ns, int D, int j, int going_right,
                                              uint32_t khi, uint32_t klo) {
    begin_body(ns, D, j, going_right, khi, klo, threadIdx.x);
}
__global__ __launch_bounds__(64) void k_end(double* ns, int D, int max_depth, uint32_t thi,
                                            uint32_t tlo) {
    end_body(ns, D, max_depth, thi, tlo, threadIdx.x);
}
__global__ __launch_bounds__(64) void k_finish(double* ns, int D) { finish_body(ns, D, threadIdx.x); }

// ---- lock-step chains: block c works on chain c's state (ns + c*stride); its step size and
// per-doubling directions / keys come from a parameter table uploaded once per transition:
//   par[c] = { eps, (going_right, sub_hi, sub_lo, tr_hi, tr_lo) x max_depth }   (doubles)
__host__ __device__ inline int par_doubles(int max_depth) { return 1 + 5 * max_depth; }
__global__ __launch_bounds__(64) void kv_init(double* ns, size_t stride, int D, const double* par,
                                              int par_stride, double max_de) {
    const int c = blockIdx.x;
    init_body(ns + c * stride, D, par[(size_t)c * par_stride], max_de, threadIdx.x);
}
__global__ __launch_bounds__(64) void kv_begin(double* ns, size_t stride, int D, int j,
                                               const double* par, int par_stride) {
    const int c = blockIdx.x;
    const double* p = par + (size_t)c * par_stride + 1 + 5 * j;
    begin_body(ns + c * stride, D, j, p[0] != 0.0, (uint32_t)p[1], (uint32_t)p[2], threadIdx.x);
}
__global__ __launch_bounds__(64) void kv_end(double* ns, size_t stride, int D, int j, int max_depth,
                                             const double* par, int par_stride) {
    const int c = blockIdx.x;
    const double* p = par + (size_t)c * par_stride + 1 + 5 * j;
    end_body(ns + c * stride, D, max_depth, (uint32_t)p[3], (uint32_t)p[4], threadIdx.x);
}
__global__ __launch_bounds__(64) void kv_finish(double* ns, size_t stride, int D) {
    finish_body(ns + blockIdx.x * stride, D, threadIdx.x);
}

// ---------------------------------------------------------------- persistent chains
// The whole chain on the device: when a doubling ends, the wave that booked the last leaf
// also combines the trees, and when the transition ends it adapts (dual averaging, Welford
// mass matrix, numpyro's windowed schedule), stores the draw and starts the next transition
// -- all random inputs of a chain are data independent (momentum normals, doubling
// directions and keys), so the host generates them up front.  The host only enqueues
// evaluations and looks at the "all done" flags once per chunk: no per-transition round trip
// and no lock step between chains (every chain always has a useful leapfrog to do).
//
// Per-chain "PD" block after the NS region: scalars, then W_MEAN | W_M2 | MSQRT (D each).
enum {
    P_ITER = 0, P_WARM, P_TOTAL, P_ALLDONE, P_STEP, P_DA_T, P_DA_XT, P_DA_XAVG, P_DA_GAVG,
    P_DA_PROX, P_WIN, P_NWIN, P_ADAPT_SS, P_ADAPT_MM, P_TARGET, P_W_N, P_MEAN_ACC, P_NDIV,
    P_THIN, P_START_IDX, P_MAXDE, P_N = 24
};
__host__ __device__ inline size_t pd_doubles(int D) { return (size_t)P_N + 3 * (size_t)D; }

struct Persist {             // device-resident descriptor, shared by all chains
    const double* normals;   // [C][n_iter][D]   momentum draws (unit normal)
    const double* par;       // [C][n_iter][max_depth][5]  going_right, sub_hi, sub_lo, tr_hi, tr_lo
    const int* win_end;      // [n_win] last iteration of each adaptation window
    double* draws;           // [C][kept][D]
    double* stats;           // [C][kept][6]  pe, accept_prob, step_size, num_steps, diverging, aux0
    int n_iter, kept, max_depth, n_win, D;
    size_t pd_off;           // doubles from a chain's NS base to its PD block
};

__device__ __forceinline__ void wave_mem_sync() {  // stores of this wave visible to its loads
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    __builtin_amdgcn_s_waitcnt(0);
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
}

// momentum of iteration `it` (r = mass_sqrt * unit normal), tree := current state, doubling 0
template <class Team>
__device__ inline void persist_start_transition_t(double* ns, const Persist& P, int chain, int it, Team& tm) {
    const int D = P.D;
    double* pd = ns + P.pd_off;
    const double* msq = pd + P_N + 2 * (size_t)D;
    const double* nrm = P.normals + ((size_t)chain * P.n_iter + it) * D;
    double* __restrict__ r = vec(ns, D, V_TL_R);
#pragma clang loop unroll_count(Team::UNROLL)
    for (int i = tm.first(); i < D; i += tm.stride()) r[i] = msq[i] * nrm[i];
    tm.sync();
    init_body_t(ns, D, tm.ld(&pd[P_STEP]), tm.ld(&pd[P_MAXDE]), tm);
    tm.sync();
    const double* q = P.par + ((size_t)chain * P.n_iter + it) * P.max_depth * 5;
    begin_body_t(ns, D, 0, q[0] != 0.0, (uint32_t)q[1], (uint32_t)q[2], tm);
}
template <int NT = 64>
__device__ inline void persist_start_transition(double* ns, const Persist& P, int chain, int it,
                                                int lane, double* scr = nullptr) {
    LocalTeam<NT> tm{lane, scr};
    persist_start_transition_t(ns, P, chain, it, tm);
}

// called by the team that booked the last leaf when the subtree of the current doubling is complete
template <class Team>
__device__ inline void persist_advance_t(double* ns, const Persist& P, int chain, Team& tm) {
    const int D = P.D, md = P.max_depth;
    double* pd = ns + P.pd_off;
    tm.sync();
    if (tm.ld(&pd[P_ALLDONE]) != 0.0) return;
    const int it = (int)tm.ld(&pd[P_ITER]);
    {   // end of the doubling: _combine_tree(tree, subtree, biased_transition=True)
        const int j = (int)tm.ld(&ns[H_T_DEPTH]);
        const double* q = P.par + (((size_t)chain * P.n_iter + it) * md + j) * 5;
        end_body_t(ns, D, md, (uint32_t)q[3], (uint32_t)q[4], tm);
        tm.sync();
    }
    if (tm.ld(&ns[H_STOP]) == 0.0) {  // next doubling of the same transition
        const int j = (int)tm.ld(&ns[H_T_DEPTH]);
        const double* q = P.par + (((size_t)chain * P.n_iter + it) * md + j) * 5;
        begin_body_t(ns, D, j, q[0] != 0.0, (uint32_t)q[1], (uint32_t)q[2], tm);
        return;
    }
    // ---- the transition is complete: proposal -> state, statistics, adaptation, next one
    const double num = tm.ld(&ns[H_T_NUM]);
    const double accept_prob = num > 0 ? tm.ld(&ns[H_T_SUMACC]) / num : 0.0;
    const bool diverging = tm.ld(&ns[H_T_DIV]) != 0.0;
    finish_body_t(ns, D, tm);
    tm.sync();
    const double* zc = vec(ns, D, V_Z);
    double* w_mean = pd + P_N;
    double* w_m2 = w_mean + D;
    double* msq = w_m2 + D;
    double* invM = vec(ns, D, V_INVM);
    const int warm = (int)tm.ld(&pd[P_WARM]), total = (int)tm.ld(&pd[P_TOTAL]);
    double step = tm.ld(&pd[P_STEP]);
    if (it < warm) {  // numpyro warmup_adapter.update_fn
        const bool adapt_ss = tm.ld(&pd[P_ADAPT_SS]) != 0.0, adapt_mm = tm.ld(&pd[P_ADAPT_MM]) != 0.0;
        double da_t = tm.ld(&pd[P_DA_T]), x_t = tm.ld(&pd[P_DA_XT]), x_avg = tm.ld(&pd[P_DA_XAVG]),
               g_avg = tm.ld(&pd[P_DA_GAVG]);
        if (adapt_ss) {  // dual_averaging(t0 = 10, kappa = 0.75, gamma = 0.05)
            const double g = tm.ld(&pd[P_TARGET]) - accept_prob;
            da_t += 1.0;
            g_avg = (1.0 - 1.0 / (da_t + 10.0)) * g_avg + g / (da_t + 10.0);
            x_t = tm.ld(&pd[P_DA_PROX]) - sqrt(da_t) / 0.05 * g_avg;
            const double weight_t = pow(da_t, -0.75);
            x_avg = (1.0 - weight_t) * x_avg + weight_t * x_t;
            const double sx = it == warm - 1 ? exp(x_avg) : exp(x_t);
            step = sx < 1.1754943508222875e-38 ? 1.1754943508222875e-38 : sx;
        }
        int win = (int)tm.ld(&pd[P_WIN]);
        const int nwin = (int)tm.ld(&pd[P_NWIN]);
        const bool is_middle = 0 < win && win < nwin - 1;
        double w_n = tm.ld(&pd[P_W_N]);
        if (adapt_mm && is_middle) {  // welford_covariance(diagonal=True)
            w_n += 1.0;
            for (int i = tm.first(); i < D; i += tm.stride()) {
                const double d_pre = zc[i] - w_mean[i];
                const double mnew = w_mean[i] + d_pre / w_n;
                w_mean[i] = mnew;
                w_m2[i] += d_pre * (zc[i] - mnew);
            }
        }
        const bool at_end = it == P.win_end[win];
        if (at_end) win += 1;
        double prox = tm.ld(&pd[P_DA_PROX]);
        if (at_end && is_middle) {
            if (adapt_mm) {
                for (int i = tm.first(); i < D; i += tm.stride()) {
                    double c = w_m2[i] / (w_n - 1.0);
                    c = (w_n / (w_n + 5.0)) * c + 1e-3 * (5.0 / (w_n + 5.0));
                    invM[i] = c;
                    msq[i] = 1.0 / sqrt(c);
                    w_mean[i] = 0.0;
                    w_m2[i] = 0.0;
                }
                w_n = 0.0;
            }
            if (adapt_ss) {
                da_t = 0.0; x_t = 0.0; x_avg = 0.0; g_avg = 0.0;
                prox = log(10.0 * step);
            }
        }
        tm.readers_done();  // every thread has read the scalars rewritten below
        if (tm.leader()) {
            tm.st(&pd[P_DA_T], da_t); tm.st(&pd[P_DA_XT], x_t); tm.st(&pd[P_DA_XAVG], x_avg); tm.st(&pd[P_DA_GAVG], g_avg);
            tm.st(&pd[P_DA_PROX], prox); tm.st(&pd[P_WIN], (double)win); tm.st(&pd[P_W_N], w_n); tm.st(&pd[P_STEP], step);
        }
    } else {
        const int n = it - warm + 1;
        const int thin = (int)tm.ld(&pd[P_THIN]), start_idx = (int)tm.ld(&pd[P_START_IDX]);
        if (it >= start_idx && (it - start_idx) % thin == thin - 1) {
            const int idx = (it - start_idx) / thin;
            double* __restrict__ dr = P.draws + ((size_t)chain * P.kept + idx) * D;
#pragma clang loop unroll_count(Team::UNROLL)
            for (int i = tm.first(); i < D; i += tm.stride()) dr[i] = zc[i];
            if (tm.leader()) {  // (the tree's header words stand until the next transition starts)
                double* st = P.stats + ((size_t)chain * P.kept + idx) * 6;
                st[0] = tm.ld(&ns[H_T_PE]); st[1] = accept_prob; st[2] = tm.ld(&ns[H_EPS]); st[3] = num;
                st[4] = diverging ? 1.0 : 0.0; st[5] = tm.ld(&ns[H_T_AUX0]);
            }
        }
        if (tm.leader()) {
            const double ma = tm.ld(&pd[P_MEAN_ACC]);
            tm.st(&pd[P_MEAN_ACC], ma + (accept_prob - ma) / n);
            if (diverging) tm.st(&pd[P_NDIV], tm.ld(&pd[P_NDIV]) + 1.0);
        }
    }
    if (it + 1 >= total) {  // chain finished: every later launch is a no-op for it
        if (tm.leader()) {
            tm.st(&pd[P_ITER], (double)(it + 1));
            tm.st(&pd[P_ALLDONE], 1.0);
            tm.st(&ns[H_S_DONE], 1.0);
            tm.st(&ns[H_S_ACTIVE], 0.0);
        }
        return;
    }
    if (tm.leader()) tm.st(&pd[P_ITER], (double)(it + 1));
    tm.sync();
    persist_start_transition_t(ns, P, chain, it + 1, tm);
}
// (dc_eval CALLS the one-wave instantiation -- the compiler does not inline it -- and a kernel is
// charged its callees' registers: this function's count decides whether the NUTS-aware dc_eval keeps
// two workgroups per CU, tests/test_kernel_resources.py)
template <int NT = 64>
__device__ inline void persist_advance(double* ns, const Persist& P, int chain, int lane,
                                       double* scr = nullptr) {
    LocalTeam<NT> tm{lane, scr};
    persist_advance_t(ns, P, chain, tm);
}

// Leaf bookkeeping as its own launch, for models whose evaluation is not dc_eval (the float64
// family: neutral-venue / World-Cup): the evaluation wrote potential, aux and gradient into
// the chain's state; one wave per chain books the leaf and advances the chain.
__global__ __launch_bounds__(64) void kp_leaf(double* ns_all, size_t stride, int D, int max_depth,
                                              Persist P) {
    extern __shared__ double gL[];  // grad[D] | potential | aux[4]
    double* ns = ns_all + blockIdx.x * stride;
    const int lane = threadIdx.x;
    if (ns[H_S_DONE] != 0.0) return;  // chain finished
    LeafState<LEAF_NE_MAX> leaf = leaf_prefetch<LEAF_NE_MAX>(ns, D, max_depth, lane);
    const double* gr = vec(ns, D, V_GRAD);
    for (int i = lane; i < D; i += 64) gL[i] = gr[i];
    if (lane == 0) {
        gL[D] = ns[H_LEAF_PE];
        gL[D + 1] = ns[H_LEAF_AUX0]; gL[D + 2] = ns[H_LEAF_AUX1];
        gL[D + 3] = ns[H_LEAF_AUX2]; gL[D + 4] = ns[H_LEAF_AUX3];
    }
    leaf_prepare<true>(leaf);
    __syncthreads();
    const bool sub_done = nuts_leaf(ns, D, max_depth, lane, gL, leaf);
    if (sub_done) persist_advance(ns, P, blockIdx.x, lane);
}

// first transition of every chain (after the host has set the initial state)
__global__ __launch_bounds__(64) void kp_start(double* ns, size_t stride, Persist P) {
    persist_start_transition(ns + blockIdx.x * stride, P, blockIdx.x, 0, threadIdx.x);
}


// ---------------------------------------------------------------- wide leaf (any D)
// The leaf for latent vectors too long for one wave (dynamic model: D ~ 10^4..10^5, leagues of
// more than 64 teams), as ONE launch of GW workgroups per chain after the evaluation (kw_leaf):
//   1  every workgroup adds its share of every reduction the leaf needs -- the kinetic energy and, for
//      each checkpoint level this leaf closes, the two U-turn dot products -- and writes the partials
//      (write-through) into part[chain][workgroup][KW_PW]
//   2  it stores what does not depend on the decisions, then meets the others at a ROW BARRIER
//   3  every workgroup adds the partials in a fixed order (same totals everywhere, deterministic),
//      takes the leaf's decisions and writes the rest of its slice; the row's leader rewrites the header
//   4  when the subtree is complete the whole row advances the chain (GridTeam below).
// Same arithmetic per element as leaf_moves / leaf_weights; only the order of the sums differs.
constexpr int KW_NTB = 1024;            // kw_leaf / kw_start
constexpr int KW_MAXL = 21;            // levels one leaf can close (max_tree_depth <= 20)
constexpr int KW_PW = 1 + 2 * KW_MAXL; // kin | (dl, dr) per level
constexpr int KW_MAX_WG = 64;
__host__ __device__ inline int kw_workgroups(int D, int nt) {
    const int g = (D + nt - 1) / nt;   // (one element per thread up to 64 workgroups)
    return g < 1 ? 1 : (g > KW_MAX_WG ? KW_MAX_WG : g);
}
struct WideLeaf {  // what both launches derive from the header (identical in every thread)
    double eps;
    bool going_right;
    int num, idx_max, idx_min, n_lev;
};
// hv: header word `lane` (one coalesced load per wave instead of a chain of scalar loads, each
// waited for before the next is issued)
__device__ __forceinline__ WideLeaf wide_leaf(double hv) {
    WideLeaf W;
    const double dir = hdr_word(hv, H_DIR);
    W.eps = hdr_word(hv, H_EPS) * dir;
    W.going_right = dir > 0.0;
    W.num = (int)hdr_word(hv, H_S_NUM);
    W.idx_max = __popc((unsigned)W.num >> 1);
    W.idx_min = W.idx_max - (__ffs(~W.num) - 1) + 1;
    W.n_lev = W.idx_max >= W.idx_min ? W.idx_max - W.idx_min + 1 : 0;
    return W;
}
// ---- the grid row of a wide launch as ONE team (see LocalTeam): the chain ADVANCE -- end of a
// doubling, end of a transition, adaptation, the next transition's start -- is a handful of passes over
// a dozen D-vectors, and on one workgroup that is one CU's load/store path: ~100 GB/s, 200 us per advance
// with the dynamic model's 35 502 entries (profiles/r03/dynamic_advance.txt), five times the leapfrog it
// follows.  Here every workgroup of kw_leaf's grid takes part:
//   vectors   thread t of the row owns elements t, t + row size, ... in EVERY pass, the same split as the
//             leaf's slice loop -- no element is ever read by a thread that did not write it, so the
//             vectors need no cross-workgroup visibility at all inside the launch
//   scalars   (header, adaptation block) are written by the row's leader with write-through stores and
//             read by everybody with L2-bypassing loads, a row barrier in between
//   sums      per-workgroup partials (write-through), a row barrier, then every wave adds the partials
//             in the same fixed order: identical totals, identical decisions everywhere
//   barrier   one agent-scope counter per chain, monotonic within the launch (barrier k is complete at
//             k x workgroups), polled by one lane, bounded; the last workgroup to leave the launch
//             (exit counter) zeroes both for the next one.
// The number of barriers a launch takes depends on the chain's state only, which every workgroup reads
// identically -- all workgroups walk the same sequence.
constexpr unsigned int FAULT_LEAF_BARRIER = 16u;   // (dc::Fault's next bit: a row barrier of the wide leaf timed out)
constexpr unsigned int ROW_SPIN_LIMIT = 1u << 22;
enum { RT_TICKET = 0, RT_BAR, RT_EXIT, RT_WORDS = 4 };   // per chain: words of the wide leaf's counters
__device__ __forceinline__ double nd_ld_sc1(const double* p) {
    return __longlong_as_double((long long)__hip_atomic_load(reinterpret_cast<const unsigned long long*>(p),
                                                             __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
}
__device__ __forceinline__ void nd_st_sc1(double* p, double v) {
    __hip_atomic_store(reinterpret_cast<unsigned long long*>(p), (unsigned long long)__double_as_longlong(v),
                       __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// a partial sum and the tag of the leaf it belongs to, 16 bytes written by ONE write-through store and
// read by ONE L2-bypassing load: the record is its own "ready" flag (no counter, no second round trip)
struct TaggedSum {
    double v;
    unsigned long long tag;
};
__device__ __forceinline__ void st_tagged(TaggedSum* p, double v, unsigned long long tag) {
    typedef unsigned long long u64x2 __attribute__((ext_vector_type(2)));
    u64x2 w = {(unsigned long long)__double_as_longlong(v), tag};
    // (the s_nop: a store of more than 64 bits reads its data registers a few cycles after issue)
    asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 2" ::"v"(p), "v"(w) : "memory");
}
__device__ __forceinline__ TaggedSum ld_tagged(const TaggedSum* p) {
    typedef unsigned long long u64x2 __attribute__((ext_vector_type(2)));
    u64x2 w;
    asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(w) : "v"(p) : "memory");
    TaggedSum r;
    r.v = __longlong_as_double((long long)w.x);
    r.tag = w.y;
    return r;
}
struct GridTeam {
    static constexpr int NT = KW_NTB;
    static constexpr int UNROLL = 2;
    int tid, wg, nwg;
    double* scr;            // LDS [2 * NT / 64]
    int* s_ok;              // LDS
    double* part;           // this chain's partials [2][KW_MAX_WG][2]
    unsigned int* ctr;      // this chain's counters (RT_*)
    unsigned int* fault;
    unsigned int arrived;   // arrivals that completed the last barrier
    int slot;
    bool dead;              // a bounded wait expired (the launch runs to its end, the host is told)
    __device__ __forceinline__ int first() const { return wg * NT + tid; }
    __device__ __forceinline__ int stride() const { return nwg * NT; }
    __device__ __forceinline__ bool leader() const { return wg == 0 && tid == 0; }
    __device__ __forceinline__ double ld(const double* p) const { return nd_ld_sc1(p); }
    __device__ __forceinline__ void st(double* p, double v) const { nd_st_sc1(p, v); }
    __device__ __forceinline__ void barrier() {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this thread's write-through stores have landed
        __syncthreads();
        arrived += (unsigned int)nwg;
        if (tid == 0) {
            (void)__hip_atomic_fetch_add(ctr + RT_BAR, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            bool ok = !dead;
            unsigned int spins = 0;
            while (ok && __hip_atomic_load(ctr + RT_BAR, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < arrived) {
                if (++spins >= ROW_SPIN_LIMIT) ok = false;
                __builtin_amdgcn_s_sleep(1);
            }
            *s_ok = ok;
        }
        __syncthreads();
        if (*s_ok == 0) dead = true;
    }
    __device__ __forceinline__ void sync() { barrier(); }
    __device__ __forceinline__ void readers_done() { barrier(); }
    __device__ __forceinline__ void sum2(double& a, double& b) {
        nd_wave_sum2(a, b);
        __syncthreads();   // (scr may still be read from the previous sum)
        if ((tid & 63) == 0) { scr[2 * (tid >> 6)] = a; scr[2 * (tid >> 6) + 1] = b; }
        __syncthreads();
        double* mine = part + ((size_t)slot * KW_MAX_WG + wg) * 2;
        if (tid == 0) {
            double sa = 0.0, sb = 0.0;
#pragma unroll
            for (int w = 0; w < NT / 64; ++w) { sa += scr[2 * w]; sb += scr[2 * w + 1]; }
            nd_st_sc1(mine, sa);
            nd_st_sc1(mine + 1, sb);
        }
        barrier();
        const int l = tid & 63;
        const double* theirs = part + ((size_t)slot * KW_MAX_WG + (l < nwg ? l : 0)) * 2;
        const double va = nd_ld_sc1(theirs), vb = nd_ld_sc1(theirs + 1);
        a = nd_wave_sum(l < nwg ? va : 0.0);
        b = nd_wave_sum(l < nwg ? vb : 0.0);
        slot ^= 1;   // (the next sum's partials must not land on words a slow workgroup still reads)
    }
    __device__ __forceinline__ double sum(double v) {
        double z = 0.0;
        sum2(v, z);
        return v;
    }
    // the launch is over for this workgroup: the last one to leave zeroes the counters
    __device__ __forceinline__ void leave() {
        __syncthreads();
        if (tid == 0) {
            if (dead && fault) (void)__hip_atomic_fetch_or(fault, FAULT_LEAF_BARRIER, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            const unsigned int k = __hip_atomic_fetch_add(ctr + RT_EXIT, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (k == (unsigned int)nwg - 1u) {
                __hip_atomic_store(ctr + RT_BAR, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(ctr + RT_EXIT, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    }
};
__host__ __device__ inline size_t row_part_doubles() { return 2 * (size_t)KW_MAX_WG * 2; }

// ONE launch per leapfrog (round 3; round 2: kw_leaf_a + kw_leaf_b, 5.0 + 7.8 us and a launch gap at
// D = 35 502).  Every thread keeps its (first two) elements in registers across the row barrier.
__global__ __launch_bounds__(KW_NTB) void kw_leaf(double* ns_all, size_t stride, int D, int max_depth,
                                                  TaggedSum* part, unsigned int* tickets, Persist P,
                                                  double* row_part, unsigned int* fault) {
    __shared__ double scr[2 * KW_NTB / 64];
    __shared__ double red[KW_NTB / 64][KW_PW];
    __shared__ double tot[KW_PW];
    __shared__ int s_ok;
    const int chain = blockIdx.y;
    double* ns = ns_all + chain * stride;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, GW = gridDim.x;
    const double hv = lane < H_N ? ns[lane] : 0.0;
    // this thread's first two elements, requested before anything that waits for the header
    const int el0 = blockIdx.x * KW_NTB + tid, estep = GW * KW_NTB;
    double pre[2][5];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int i = el0 + u * estep;
        const bool ok = i < D;
        pre[u][0] = ok ? vec(ns, D, V_GRAD)[i] : 0.0;
        pre[u][1] = ok ? vec(ns, D, V_ZN)[i] : 0.0;
        pre[u][2] = ok ? vec(ns, D, V_RH)[i] : 0.0;
        pre[u][3] = ok ? vec(ns, D, V_S_RSUM)[i] : 0.0;
        pre[u][4] = ok ? vec(ns, D, V_INVM)[i] : 0.0;
    }
    if (hdr_word(hv, H_S_DONE) != 0.0) return;  // chain finished (uniform over the grid row)
    const WideLeaf W = wide_leaf(hv);
    GridTeam tm;
    tm.tid = tid; tm.wg = blockIdx.x; tm.nwg = GW;
    tm.scr = scr; tm.s_ok = &s_ok;
    tm.part = row_part + (size_t)chain * row_part_doubles();
    tm.ctr = tickets + (size_t)chain * RT_WORDS; tm.fault = fault;
    tm.arrived = 0u; tm.slot = 0; tm.dead = false;

    const double* invM = vec(ns, D, V_INVM);
    const double* g = vec(ns, D, V_GRAD);
    double* p_zn = vec(ns, D, V_ZN); double* p_rh = vec(ns, D, V_RH); double* p_rsum = vec(ns, D, V_S_RSUM);
    double* sl_z = vec(ns, D, V_SL_Z); double* sl_r = vec(ns, D, V_SL_R); double* sl_g = vec(ns, D, V_SL_G);
    double* sr_z = vec(ns, D, V_SR_Z); double* sr_r = vec(ns, D, V_SR_R); double* sr_g = vec(ns, D, V_SR_G);
    double* sp_z = vec(ns, D, V_SP_Z); double* sp_g = vec(ns, D, V_SP_G);
    double* ck_r = vec(ns, D, V_CKPT);
    double* ck_s = ck_r + (size_t)max_depth * D;
    const double* c_r1 = W.going_right ? sr_r : sl_r;
    struct Elem { double gi, zn, rh, rs_old, im; };
    auto elem = [&](int it, int i) {
        Elem e;
        if (it < 2) {   // (selects, not an indexed register array)
            const bool f = it == 0;
            e.gi = f ? pre[0][0] : pre[1][0]; e.zn = f ? pre[0][1] : pre[1][1]; e.rh = f ? pre[0][2] : pre[1][2];
            e.rs_old = f ? pre[0][3] : pre[1][3]; e.im = f ? pre[0][4] : pre[1][4];
        } else {   // (longer vectors: from memory -- rs_old only before step 2 rewrites it)
            e.gi = g[i]; e.zn = p_zn[i]; e.rh = p_rh[i]; e.rs_old = p_rsum[i]; e.im = invM[i];
        }
        return e;
    };

    // ---- 1. this workgroup's share of every sum the decisions need: the kinetic energy and, per
    // checkpoint level this (odd) leaf closes, the two U-turn dot products
    const int nv = 1 + 2 * W.n_lev;
    {
        double kin = 0.0;
        int it = 0;
        for (int i = el0; i < D; i += estep, ++it) {
            const Elem e = elem(it, i);
            const double r = e.rh - 0.5 * W.eps * e.gi;
            kin += e.im * r * r;
        }
        kin = nd_wave_sum(kin);
        if (lane == 0) red[wave][0] = kin;
    }
    for (int l = 0; l < W.n_lev; ++l) {
        const int ci = W.idx_max - l;
        double dl = 0.0, dr = 0.0;
        int it = 0;
        for (int i = el0; i < D; i += estep, ++it) {
            const Elem e = elem(it, i);
            const double r = e.rh - 0.5 * W.eps * e.gi;
            const double rs = e.rs_old + r;
            // the first level is the previous leaf itself (see LeafState)
            const double c_r = l == 0 ? c_r1[i] : ck_r[(size_t)ci * D + i];
            const double c_s = l == 0 ? e.rs_old : ck_s[(size_t)ci * D + i];
            const double sub = rs - c_s + c_r;
            const double rsm = sub - 0.5 * (c_r + r);
            dl += e.im * c_r * rsm;
            dr += e.im * r * rsm;
        }
        nd_wave_sum2(dl, dr);
        if (lane == 0) { red[wave][1 + 2 * l] = dl; red[wave][2 + 2 * l] = dr; }
    }
    __syncthreads();
    // (tagged with this leaf's number: a record that carries it is this launch's -- the launches of a
    // stream do not overlap, and the counter moves with every leaf)
    const unsigned long long leaf_tag = (unsigned long long)hdr_word(hv, H_EVALS) + 1ull;
    if (tid < nv) {
        double sv = 0.0;
#pragma unroll
        for (int w = 0; w < KW_NTB / 64; ++w) sv += red[w][tid];
        st_tagged(&part[((size_t)chain * GW + blockIdx.x) * KW_PW + tid], sv, leaf_tag);
    }
    // ---- 2. what does not depend on the decisions goes out while the partials travel: the running
    // momentum sum, the subtree's edge on this side, the checkpoint of an even leaf
    const bool wl = W.num == 0 || !W.going_right, wr = W.num == 0 || W.going_right;
    const bool wck = (W.num & 1) == 0;
    {
        int it = 0;
        for (int i = el0; i < D; i += estep, ++it) {
            const Elem e = elem(it, i);
            const double r = e.rh - 0.5 * W.eps * e.gi;
            const double rs = W.num == 0 ? r : e.rs_old + r;
            p_rsum[i] = rs;
            if (wl) { sl_z[i] = e.zn; sl_r[i] = r; sl_g[i] = e.gi; }
            if (wr) { sr_z[i] = e.zn; sr_r[i] = r; sr_g[i] = e.gi; }
            if (wck) {
                ck_r[(size_t)W.idx_max * D + i] = r;
                ck_s[(size_t)W.idx_max * D + i] = rs;
            }
        }
    }
    // everything the decisions read from the header, before anyone rewrites it
    const double pe = hdr_word(hv, H_LEAF_PE), e0 = hdr_word(hv, H_E0), max_de = hdr_word(hv, H_MAXDE);
    const double aux0 = hdr_word(hv, H_LEAF_AUX0), aux1 = hdr_word(hv, H_LEAF_AUX1),
                 aux2 = hdr_word(hv, H_LEAF_AUX2), aux3 = hdr_word(hv, H_LEAF_AUX3);
    const double w_cur = hdr_word(hv, H_S_WEIGHT), s_sumacc = hdr_word(hv, H_S_SUMACC),
                 evals = hdr_word(hv, H_EVALS);
    const int s_max = (int)hdr_word(hv, H_S_MAX);
    uint32_t nhi, nlo;
    float u_take;
    leaf_rng(hv, &nhi, &nlo, &u_take);
    // ---- 3. the totals: one wave per value, lane b polls workgroup b's record until it carries this
    // leaf's tag (bounded) -- the load that finds it IS the hand-off; then a wave sum in fixed order.
    // A workgroup that has seen every record knows that every workgroup has read the header.
    static_assert(KW_MAX_WG <= 64, "one lane per partial");
    if (tid == 0) s_ok = 1;
    __syncthreads();
    for (int k = wave; k < nv; k += KW_NTB / 64) {
        const TaggedSum* rec = &part[((size_t)chain * GW + (lane < GW ? lane : 0)) * KW_PW + k];
        TaggedSum t;
        bool ok = false;
        for (unsigned int spin = 0; spin < ROW_SPIN_LIMIT; ++spin) {
            t = ld_tagged(rec);
            ok = __ballot(t.tag != leaf_tag) == 0ull;
            if (ok) break;
            __builtin_amdgcn_s_sleep(1);
        }
        const double sum = nd_wave_sum(lane < GW ? t.v : 0.0);
        if (lane == 0) {
            tot[k] = sum;
            if (!ok) s_ok = 0;
        }
    }
    __syncthreads();
    if (s_ok == 0) {   // a workgroup of the row never delivered: tell the host, leave the state alone
        if (tid == 0 && fault) (void)__hip_atomic_fetch_or(fault, FAULT_LEAF_BARRIER, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        return;
    }

    const double e_new = pe + 0.5 * tot[0];
    double delta = e_new - e0;
    if (delta != delta) delta = __builtin_inf();
    const bool div_leaf = delta > max_de;
    const LeafWeights LW = leaf_weigh_f(W.num, delta, u_take, [&](int k) {
        return k == H_S_WEIGHT ? w_cur : s_sumacc;
    });
    bool turning = false;
    for (int l = 0; l < W.n_lev; ++l) turning = turning || tot[1 + 2 * l] <= 0.0 || tot[2 + 2 * l] <= 0.0;
    const int new_num = W.num + 1;
    const bool done = turning || div_leaf || new_num >= s_max;

    // ---- 4. the rest of this workgroup's slice: the next position / half-stepped momentum, the proposal
    {
        int it = 0;
        for (int i = el0; i < D; i += estep, ++it) {
            const Elem e = elem(it, i);
            const double r = e.rh - 0.5 * W.eps * e.gi;
            const double rn = r - 0.5 * W.eps * e.gi;
            p_zn[i] = done ? e.zn : e.zn + W.eps * e.im * rn;
            p_rh[i] = done ? r : rn;
            if (LW.take) { sp_z[i] = e.zn; sp_g[i] = e.gi; }
        }
    }
    // ---- 5. the header (every workgroup read it before the barrier): the row's leader rewrites it
    auto write_header = [&](auto put) {
        put(&ns[H_S_NUM], (double)new_num);
        put(&ns[H_S_DIV], div_leaf ? 1.0 : 0.0);
        put(&ns[H_S_TURN], turning ? 1.0 : 0.0);
        put(&ns[H_S_DONE], done ? 1.0 : 0.0);
        put(&ns[H_EVALS], evals + 1.0);
        put(&ns[H_S_WEIGHT], LW.w_sub);
        put(&ns[H_S_SUMACC], LW.sum_acc);
        put(&ns[H_KEY_HI], (double)nhi);
        put(&ns[H_KEY_LO], (double)nlo);
        if (LW.take) {
            put(&ns[H_S_PE], pe);
            put(&ns[H_S_EPROP], e_new);
            put(&ns[H_S_AUX0], aux0); put(&ns[H_S_AUX1], aux1); put(&ns[H_S_AUX2], aux2); put(&ns[H_S_AUX3], aux3);
        }
    };
    if (tm.leader()) write_header([](double* p, double v) { nd_st_sc1(p, v); });
    // ---- 6. a complete subtree: the whole row advances the chain (the only launches that touch the
    // row's counters)
    if (done) {
        persist_advance_t(ns, P, chain, tm);   // (begins with a row barrier: the header is out)
        tm.leave();
    }
}
// first transition of every chain: the same row as kw_leaf
__global__ __launch_bounds__(KW_NTB) void kw_start(double* ns, size_t stride, Persist P, unsigned int* tickets,
                                                   double* row_part, unsigned int* fault) {
    __shared__ double scr[2 * KW_NTB / 64];
    __shared__ int s_ok;
    const int chain = blockIdx.y;
    GridTeam tm;
    tm.tid = threadIdx.x; tm.wg = blockIdx.x; tm.nwg = gridDim.x;
    tm.scr = scr; tm.s_ok = &s_ok;
    tm.part = row_part + (size_t)chain * row_part_doubles();
    tm.ctr = tickets + (size_t)chain * RT_WORDS; tm.fault = fault;
    tm.arrived = 0u; tm.slot = 0; tm.dead = false;
    persist_start_transition_t(ns + chain * stride, P, chain, 0, tm);
    tm.leave();
}

}  // namespace nd
