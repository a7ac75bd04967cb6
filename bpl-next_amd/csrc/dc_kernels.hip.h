// dc_kernels.hip.h -- gfx950 (CDNA4, wave64) kernels for the Dixon-Coles potential
// energy and its gradient.  No MFMA: the path is a stream + gather from an
// LDS-resident per-team table + reduction.  At the BASELINE sizes (6 MB / evaluation)
// it is latency bound, so the design minimises the dependent chain of one launch:
//
//   dc_eval   ONE launch per evaluation, grid = (1 + n_wg) x chains, 512 threads.
//     block 0  "prior" workgroup AND tail: everything that depends on z only, in float64 and in
//              parallel with the streaming: priors + Jacobians and their gradient, the
//              chain-rule scalars, the exact (float64) rates of every pair -> true rho,
//              arg-extremal pairs for the adjoint of the rho bounds, and the rounding
//              error of every float32 table entry (for a first-order correction of U).  Then it
//              polls the counted accumulator rows (below) and runs the epilogue: adjoint of the
//              bounds, first-order value corrections, chain rule -- adds and FMAs only -- and,
//              in the NUTS-aware instantiation, the leapfrog's bookkeeping (nuts_dev.hip.h).
//     blocks 1..n_wg  streaming workgroups:
//       - fixture loads of the first tile are issued before anything else;
//       - per-team tables {exp(att+ha), exp(-def)}, {exp(att), exp(-def)} rebuilt in LDS
//         from z in float32 (v_exp_f32), rho from three DPP max reductions over the
//         unique-pair table;
//       - fixtures: one lane = 32 consecutive fixtures of ONE (home, away) pair (runs are padded
//         to the lane width), goals as packed bytes, indices once per lane (6 or 10 B per fixture
//         in the caller's format, 2.25 / 6.25 B in the library's copy);
//         per lane: 2 LDS gathers, 2 products, 4 tau terms (v_log_f32 + v_rcp_f32 each);
//         per fixture: SWAR score-class test only;
//         per-(home,away) run sums: in lane -> across the wave by DPP -> float64 LDS
//         per-team accumulators (fixed point, units of 2^-30);
//       - hand-off: every touched per-team sum and the four scalars are ADDED into the chain's
//         COUNTED ACCUMULATOR ROWS (GA_ROW below: one integer atomic carries value and
//         contribution count) -- and the workgroup is done.
//   dc_eval_loop   the same roles inside ONE resident launch for up to 1024 leapfrogs of a
//              device-resident chain (tiles in registers, the next position as data-tagged
//              granules, the tail's copy of the position in LDS).
//
//   Wave reductions of several values at once (the ends of the pair walk, of a tile, of the epilogue's
//   waves) are written out in wave_reduce.hip.h (generated: tools/gen/wave_reduce_asm.py): at this time
//   scale the compiler's schedule of interleaved DPP chains -- one pair of temporaries for all chains,
//   canonicalising maxima -- was 0.3-0.6 us per reduction.
//
// Mathematics: SURVEY.md Appendix A (restating bpl/dixon_coles.py:39-84,
// bpl/extended_dixon_coles.py:78-248, bpl/_util.py:17-93 under numpyro semantics).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "dc_layout.h"
#include "lean_math.hip.h"
#include "nuts_dev.hip.h"
#include "wave_reduce.hip.h"

namespace dc {

#ifndef DC_LANE_FIX
#define DC_LANE_FIX 32
#endif
constexpr int LANE_FIX = DC_LANE_FIX;    // fixtures per lane per tile (4, 8, 16 or 32)
static_assert(LANE_FIX == 4 || LANE_FIX == 8 || LANE_FIX == 16 || LANE_FIX == 32,
              "LANE_FIX must be 4, 8, 16 or 32");
constexpr int HWORDS = LANE_FIX / 2;     // packed u16 pairs per lane
constexpr int XWORDS = LANE_FIX / 4;     // packed u8 quads per lane
constexpr int TILE = 64 * LANE_FIX;      // fixtures per wave-tile
constexpr int BLOCK = 512;               // 8 waves per workgroup
constexpr int WAVES = BLOCK / 64;
// device-resident NUTS: waves 4..7 of the tail workgroup idle during the per-team epilogue, so
// the leaf's preparation is spread over them and costs nothing on the serial path
constexpr int PAIR_BATCH = 4;            // pairs requested per round of the rho-bound loops
constexpr int LEAF_WAVE = 6;             // books the leaf (waves k and k + 4 share a SIMD: 6 and 7 sit
                                         // beside the two lightest epilogue groups, waves 2 and 3)
constexpr int RNG_WAVE = 7;              // draws the leaf's random numbers (threefry)
constexpr int COV_WAVE_A = 4, COV_WAVE_D = 5;  // covariate-coefficient gradients (any launch)
// (Copying the whole checkpoint area into LDS on waves 6,7 -- so that a leaf closing several
// subtrees needs no dependent load -- measured no gain for one chain and the larger LDS
// footprint cost 20 % at 16 chains: not done.)
constexpr int N_SCAL = 4;                // SLAM, SLOG2, SU, CLIPC
constexpr int MAX_RG = 32;               // row groups of the slab reduction
constexpr int RUN_LOOP_MAX = 4;          // runs per wave-tile handled by masked DPP sums
#ifndef DC_TK_GROUPS
#define DC_TK_GROUPS 16
#endif
constexpr int TK_GROUPS = DC_TK_GROUPS;            // two-level arrival tickets: group counters ...
constexpr int TK_STRIDE = 32;            // ... one per 128-byte line
constexpr int TK_WORDS = (1 + TK_GROUPS) * TK_STRIDE;  // u32 words per chain

// ---- dc_eval's cross-workgroup hand-off: COUNTED ACCUMULATOR ROWS in global memory.
// A streaming workgroup adds its per-team partial sums and its four scalars into one row per
// value with agent-scope integer atomics; the memory system does the reduction, and the tail reads
// 3T + 4*GA_SHARDS values instead of staging and column-summing one slab per workgroup (round 1:
// 2.0 of the 8.4 us of an evaluation).  Integer adds commute, so the result is bitwise reproducible
// whatever the arrival order.
// A double v (in units of 2^-30) is split as v = hi * 2^44 + lo, |lo| <= 2^43; hi is added only when
// non-zero, i.e. beyond 1.6e4.  Round 3: the row COUNTS ITS OWN CONTRIBUTIONS -- the lo word takes
//     lo + 2^45 (a bias that keeps every addend positive) + 2^56 (one contribution)
// in ONE atomic, so its top byte is the number of workgroups that have added (n_wg <= 255), the 54
// bits below hold sum(lo) + count * 2^45 (< 2^54 for 255 contributions), and bits 54 / 55 are two
// flag bits that no sum can reach.  A row's atomics hit one 128-byte line, i.e. one memory channel,
// in issue order: a workgroup adds hi (or ORs a flag bit in for a non-finite value) BEFORE the
// counted lo, so a row whose count is complete is complete -- and the tail reads {lo, hi} with ONE
// 16-byte load per row (an L2-bypassing load is one request per lane to the memory side, and a CU
// gets about one such request through per nanosecond: three 8-byte loads per row were 0.96 us of
// poll, one 16-byte load is the round trip alone).  That removes a whole stage from both
// sides of the hand-off: the streaming workgroup no longer drains its atomics, barriers and bumps an
// arrival counter (it is simply done), and the tail no longer polls counters and THEN loads the rows:
// it polls the rows, the load that finds a row complete IS the data (the expected count of every row
// is static, host-built).  Round 2's chain was three memory-side round trips between the last fixture
// and the epilogue (2.3 us of a 7 us evaluation, profiles/r02/stamps_timeline.txt); this is one.
// (Tried first this round and measured slower: a "tagged slab" of per-workgroup entries polled by
// the tail -- 2400 uncached 16-byte loads from one CU take 2 us, profiles/r03/stamps_tagged_slab.txt.)
// 9e-10 absolute resolution -- the addends are float32-born run sums, 1e-4 absolute -- and 1.5e20
// range (the init_to_uniform(radius=2) region reaches potentials of 1e15).  Values outside it,
// infinities and NaNs set a flag bit instead (and still count).
#ifndef DC_GA_ROW
#define DC_GA_ROW 16
#endif
constexpr int GA_ROW = DC_GA_ROW;        // int64 per row (one 128-byte line): [0] lo + flags + count, [1] hi
constexpr int GA_SHARDS = 16;            // the four scalars are added by every workgroup: sharded
constexpr double GA_HI_UNIT = 16384.0;             // 2^14: the hi word's unit (real units)
constexpr double GA_LO_SCALE = 1073741824.0;       // 2^30
constexpr double GA_LIMIT = 1.8446744073709552e19; // 2^64
constexpr long long GA_BIAS = 1ll << 45;
constexpr int GA_COUNT_SHIFT = 56, GA_FLAG_SHIFT = 54;
constexpr unsigned int GA_NEGINF = 1u, GA_BAD = 2u;
__host__ __device__ inline int ga_rows(int T) { return 3 * T + N_SCAL * GA_SHARDS; }

// z-only record written by the prior workgroup (doubles), per chain:
//   [0..ZO_HDR)         scalars, see enum
//   [ZO_HDR, +D)        gz[i]  = -dL_prior/dz_i  (everything not involving fixture sums)
//   [ZO_HDR+D, +3T)     eps: eAg[T] | eA[T] | eDn[T]   (log(true / float32 table entry))
enum {
    ZO_LZ = 0,    // z-only part of L: priors + Jacobians + sum_t(att cA - def cD + ha cH)
    ZO_SA, ZO_SD, ZO_SH,  // exp(z_std_*)
    ZO_Q, ZO_DQ,          // corr_coef_raw (clipped sigmoid) and its derivative
    ZO_UB, ZO_LB,         // true bounds (float64)
    ZO_RHO,               // true rho
    ZO_DRHO,              // rho_true - (double)rho_f32 used by the streaming workgroups
    ZO_M, ZO_LH, ZO_LA,   // true maxima
    ZO_PP, ZO_PQ, ZO_PR,  // arg-extremal pairs (home | away<<16, as double)
    ZO_FLAGS,             // bit0 P home clipped, bit1 P away clipped, bit2 Q, bit3 R
    ZO_PAIRC,             // per-pair value corrections (float32 rate-product rounding, clipped-rate log term)
    ZO_RHOF,              // (double)rho_f32: what every streaming workgroup used
    ZO_ILL,               // != 0: some (pair, class) has a float32 tau argument below TAU_ILL (ill_pass)
    ZO_HDR = 24
};

struct EvalArgs {
    // fixtures (library-owned, sorted by (home,away), padded to TILE with team T)
    const uint32_t* h;  // [n_tiles*64]  per lane: home index | real fixtures of the lane << 16
    const uint32_t* a;  // [n_tiles*64]  per lane: away index
    const uint32_t* x;  // [n_tiles*64*XWORDS]  LANE_FIX x u8 home goals
    const uint32_t* y;  // [n_tiles*64*XWORDS]  LANE_FIX x u8 away goals
    const float* w;     // [n_tiles*64*LANE_FIX] f32 weights, or nullptr
    int n_tiles;
    int tiles_per_wave;
    int active_waves;       // waves of a workgroup that own tiles (the first ones)
    const uint32_t* pairs;  // [P] unique (home | away<<16)
    const double* pairw;    // [P][4] per pair, data only: sum of weights | sum w x | sum w y | 0
    const double* pairc;    // [P][4] per pair, data only: weight of its (0,0) | (1,0) | (0,1) fixtures | 0  (ill_pass)
    double w11;             // weight of all (1,1) fixtures
    int dense_pairs;  // 1: every ordered pair h != a is present -- the maxima over pairs are separable (O(T))
    int P;
    const double* xs;       // [T,K] standardised covariates (float64) or nullptr
    const float* xsf;       // the same in float32 (streaming prologue)
    // data-only sums
    const double* cA;   // [T] sum_{h=t} w x + sum_{a=t} w y   (coefficient of attack_t)
    const double* cD;   // [T] sum_{a=t} w x + sum_{h=t} w y   (coefficient of -defence_t)
    const double* cH;   // [T] sum_{h=t} w x                   (coefficient of home_adv_t)
    double lgsum;       // sum_i w_i (lgamma(x_i+1) + lgamma(y_i+1))
    // static sparse-slab structure: a streaming workgroup only touches a few of the 3T
    // per-team slots (fixtures are sorted by pair), so it publishes just those
    const int* wg_off;      // [n_wg+1] offsets into wg_slots / the compact array
    const int* wg_slots;    // [total_c] slot (column) index in [0, 3T) of each compact entry
    const int* wg_dst;      // [total_c] where each entry goes in the column-major compact array
    const int* col_off;     // [3T+1]   column c owns compact[col_off[c] .. col_off[c+1]),
                            //          entries in workgroup order
    int total_c;
    // scratch: ONE contiguous hand-off record per chain, read by the tail in one batch:
    //   [ zo: zo_stride | scal: n_wg*N_SCAL | compact: total_c ]
    double* hbuf;           // [chains][hb_stride]
    int hb_stride;
    int n_wg;               // streaming workgroups (grid.x = n_wg + 1)
    int zo_stride;
    unsigned int* tickets;  // [chains][TK_WORDS] arrival counters (top + TK_GROUPS groups)
    long long* gacc;        // [chains][2][ga_rows(T)][GA_ROW] counted accumulator rows, two sets (dc_eval; dc_vec: unused)
    int chains;             // number of chains of this launch (dc_vec.hip.h)
    // in / out: chain c at z + c*z_stride, potential + c*p_stride, grad + c*g_stride,
    // aux + c*aux_stride (plain batches: D, 1, D, 4; device NUTS: all inside the state buffer)
    const double* z;        // [chains][D]
    double* potential;      // [chains]
    double* grad;           // [chains][D]
    double* aux;            // [chains][4] or nullptr
    int z_stride, p_stride, g_stride, aux_stride;
    unsigned long long* debug;  // diagnostic build only (DC_STAMPS): [n_wg+1][16]
    // device-resident NUTS (nuts_dev.hip.h): when set, z/potential/grad/aux point into this
    // state buffer, a finished subtree makes the launch return at once, and the tail runs
    // the leaf bookkeeping and writes the next leapfrog's position
    double* nuts;           // chain c at nuts + c*nuts_stride
    int nuts_stride;
    int nuts_max_depth;
    const nd::Persist* persist;  // persistent chains: doubling / transition advance in the tail
    // persistent EVALUATION kernel (dc_eval_loop: one chain): the launch stays resident for up to
    // `persist_steps` leapfrogs; the leaf's wave publishes the next position as data-tagged
    // granules {float32 z_i, tag}, tag = tag_base + 1 + step (the host hands out tag ranges that
    // never repeat, so a granule of an earlier launch can never look current)
    int persist_steps;
    int persist_spec;        // 1: the next position is published before the leaf is booked (tail_waves, defer_out)
    unsigned int tag_base;
    unsigned long long* zg;  // [chains][D] granules
    const int* ga_expect;    // [ga_rows(T)] contributions every accumulator row receives per evaluation (static)
    // the context's fault word (host memory mapped into the device, bplhip.hip): a bounded wait that
    // expires ORs its code in with a system-scope atomic -- nothing touches it otherwise -- and the
    // host turns it into BPLHIP_EHIP at its next entry point or synchronisation
    unsigned int* fault;
    Layout L;
};
enum : unsigned int {
    FAULT_EVAL_ARRIVALS = 1u,   // dc_eval: the streaming workgroups never arrived
    FAULT_LOOP_ARRIVALS = 2u,   // dc_eval_loop: the same inside the persistent kernel
    FAULT_LOOP_GRANULES = 4u,   // dc_eval_loop: the next position never arrived
    FAULT_DYN_BARRIER = 8u,     // dyn_fused: a grid barrier / a cell record timed out
};
// ---- memory ordering of the cross-workgroup hand-offs (round 4: the A/B VERDICT r03 asked for).
// Shipped: every arrival atomic and every polling load is RELAXED at agent scope; what orders a
// producer's data against its arrival is the hardware itself -- the data is either the atomic's own target
// (accumulator rows, data-tagged granules and records: nothing to order) or went out as write-through
// stores that the wave waits for (s_waitcnt vmcnt(0): acknowledged by the memory side) before it issues
// the arrival.  -DDC_STRICT_ORDER=1 builds the formal version instead: RELEASE on the arrivals, ACQUIRE on
// the pollers (on gfx950 a release at agent scope is buffer_wbl2 + s_waitcnt in front of the atomic, an
// acquire buffer_inv behind the load).  A/B of the two libraries on one box: profiles/r04/ab_strict_order.txt,
// DESIGN.md section 4 ("Memory ordering").
#ifndef DC_STRICT_ORDER
#define DC_STRICT_ORDER 0
#endif
#ifndef DC_ILL_CALL   // (diagnostic builds: 0 compiles the float64 pass of the ill-conditioned tau classes out)
#define DC_ILL_CALL 1
#endif
#define DC_ARRIVE_ORDER (DC_STRICT_ORDER ? __ATOMIC_RELEASE : __ATOMIC_RELAXED)
#define DC_POLL_ORDER (DC_STRICT_ORDER ? __ATOMIC_ACQUIRE : __ATOMIC_RELAXED)
#define DC_ARRIVE_RET_ORDER (DC_STRICT_ORDER ? __ATOMIC_ACQ_REL : __ATOMIC_RELAXED)   // the last arriver goes on to read
__device__ __forceinline__ void poll_acquired() {   // behind a polling load written in asm
    if (DC_STRICT_ORDER) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
}
__device__ __forceinline__ void raise_fault(unsigned int* fault, unsigned int code) {
    if (fault) (void)__hip_atomic_fetch_or(fault, code, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
__device__ __forceinline__ const double* z_of(const EvalArgs& A, int c) { return A.z + (size_t)c * A.z_stride; }
__device__ __forceinline__ double* grad_of(const EvalArgs& A, int c) { return A.grad + (size_t)c * A.g_stride; }
__device__ __forceinline__ double* pot_of(const EvalArgs& A, int c) { return A.potential + (size_t)c * A.p_stride; }
__device__ __forceinline__ double* aux_of(const EvalArgs& A, int c) { return A.aux + (size_t)c * A.aux_stride; }
__device__ __forceinline__ double* nuts_of(const EvalArgs& A, int c) { return A.nuts + (size_t)c * A.nuts_stride; }

// Diagnostic build (make stamps): thread 0 of every workgroup stores the 100 MHz
// s_memrealtime counter at phase boundaries into a buffer nothing else reads.
#ifdef DC_STAMPS
#define DC_STAMP(k)                                                                    \
    do {                                                                               \
        if (threadIdx.x == 0 && A.debug && blockIdx.y == 0)                            \
            A.debug[(size_t)blockIdx.x * 16 + (k)] = __builtin_amdgcn_s_memrealtime(); \
    } while (0)
#define DC_STAMP_LEAF(k) DC_STAMP_WAVE(LEAF_WAVE, k)
#define DC_STAMP_WAVE(w, k)                                                            \
    do {                                                                               \
        if (threadIdx.x == (w) * 64 && A.debug && blockIdx.y == 0)                     \
            A.debug[(size_t)blockIdx.x * 16 + (k)] = __builtin_amdgcn_s_memrealtime(); \
    } while (0)
// every wave of block 0, into the spare row behind the workgroups' records: which wave is the last
// one at a barrier of the prior part (slot0 + wave)
#define DC_STAMP_PW(slot0)                                                                                  \
    do {                                                                                                    \
        if ((threadIdx.x & 63) == 0 && blockIdx.x == 0 && A.debug && blockIdx.y == 0)                       \
            A.debug[(size_t)gridDim.x * 16 + (slot0) + (threadIdx.x >> 6)] = __builtin_amdgcn_s_memrealtime(); \
    } while (0)
#define DC_STAMP_SEQ(k)                                                                                     \
    do {                                                                                                    \
        if (threadIdx.x == 0 && blockIdx.x == 0 && A.debug && blockIdx.y == 0)                               \
            A.debug[(size_t)gridDim.x * 16 + (k)] = __builtin_amdgcn_s_memrealtime();                        \
    } while (0)
#else
#define DC_STAMP_SEQ(k) do { } while (0)
#define DC_STAMP_PW(slot0) do { } while (0)
#define DC_STAMP(k) do { } while (0)
#define DC_STAMP_LEAF(k) do { } while (0)
#define DC_STAMP_WAVE(w, k) do { } while (0)
#endif

// (lean float64 math: lean_math.hip.h)


// ------------------------------------------------------------------ wave helpers (DPP)

template <int CTRL, int ROW_MASK = 0xF>
__device__ __forceinline__ int dpp_i32(int old, int v) {
    return __builtin_amdgcn_update_dpp(old, v, CTRL, ROW_MASK, 0xF, false);
}
template <int CTRL, int ROW_MASK = 0xF>
__device__ __forceinline__ float dpp_f32(float old, float v) {
    return __int_as_float(dpp_i32<CTRL, ROW_MASK>(__float_as_int(old), __float_as_int(v)));
}
template <int CTRL, int ROW_MASK = 0xF>
__device__ __forceinline__ double dpp_f64(double old, double v) {
    const long long o = __double_as_longlong(old), x = __double_as_longlong(v);
    const int lo = dpp_i32<CTRL, ROW_MASK>((int)o, (int)x);
    const int hi = dpp_i32<CTRL, ROW_MASK>((int)(o >> 32), (int)(x >> 32));
    return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}
// Sum over the 64 lanes, returned wave-uniform.  quad_perm, quad_perm, row_ror:4,
// row_ror:8 leave every lane with its 16-lane row total; row_bcast:15 / row_bcast:31
// fold the rows so lane 63 holds the wave total.
__device__ __forceinline__ float wave_sum_f32(float v) {
    v += dpp_f32<0xB1>(0.f, v);        // quad_perm [1,0,3,2]
    v += dpp_f32<0x4E>(0.f, v);        // quad_perm [2,3,0,1]
    v += dpp_f32<0x124>(0.f, v);       // row_ror:4
    v += dpp_f32<0x128>(0.f, v);       // row_ror:8
    v += dpp_f32<0x142, 0xA>(0.f, v);  // row_bcast:15 into rows 1,3
    v += dpp_f32<0x143, 0xC>(0.f, v);  // row_bcast:31 into rows 2,3
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}
// two independent sums, interleaved step by step (twice the ILP of two calls)
__device__ __forceinline__ void wave_sum2_f32(float& a, float& b) {
    a += dpp_f32<0xB1>(0.f, a);        b += dpp_f32<0xB1>(0.f, b);
    a += dpp_f32<0x4E>(0.f, a);        b += dpp_f32<0x4E>(0.f, b);
    a += dpp_f32<0x124>(0.f, a);       b += dpp_f32<0x124>(0.f, b);
    a += dpp_f32<0x128>(0.f, a);       b += dpp_f32<0x128>(0.f, b);
    a += dpp_f32<0x142, 0xA>(0.f, a);  b += dpp_f32<0x142, 0xA>(0.f, b);
    a += dpp_f32<0x143, 0xC>(0.f, a);  b += dpp_f32<0x143, 0xC>(0.f, b);
    a = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(a), 63));
    b = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(b), 63));
}
// value of the previous lane (lane 0 gets `fill`): DPP wave_shr:1, no LDS crossbar
__device__ __forceinline__ uint32_t prev_lane_u32(uint32_t v, uint32_t fill) {
    return (uint32_t)dpp_i32<0x138>((int)fill, (int)v);
}
__device__ __forceinline__ float wave_max_f32(float v) {  // v >= 0  (written out: wave_reduce.hip.h)
    wr::wave_reduce_max1_f32_raw(v);
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}
// three maxima at once, interleaved step by step (a single wave issues a dependent DPP + max only
// every ~20 cycles)
// (written out, fused v_max_f32_dpp: wave_reduce.hip.h)
__device__ __forceinline__ void wave_max3_f32(float& a, float& b, float& c) {  // all >= 0
    wr::wave_reduce_max3_f32_raw(a, b, c);
    a = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(a), 63));
    b = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(b), 63));
    c = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(c), 63));
}
__device__ __forceinline__ double readlane63_f64(double v) {
    const long long x = __double_as_longlong(v);
    const int lo = __builtin_amdgcn_readlane((int)x, 63);
    const int hi = __builtin_amdgcn_readlane((int)(x >> 32), 63);
    return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}
__device__ __forceinline__ double wave_sum_f64(double v) {   // (written out: wave_reduce.hip.h)
    wr::wave_reduce_sum1_f64_raw(v);
    return readlane63_f64(v);
}
__device__ __forceinline__ void wave_sum2_f64(double& a, double& b) {
    wr::wave_reduce_sum2_f64_raw(a, b);
    a = readlane63_f64(a);
    b = readlane63_f64(b);
}
// four independent sums, interleaved step by step: a single wave issues a dependent DPP + add
// only every ~20 cycles, so one sum after another leaves the pipeline three quarters empty
// (round 3: written out in wave_reduce.hip.h -- the compiler's schedule of the interleaved C++
// serialised the four chains on one pair of temporaries)
__device__ __forceinline__ void wave_sum4_f64(double (&v)[4]) {
    wr::wave_reduce_sum4_f64_raw(v[0], v[1], v[2], v[3]);
#pragma unroll
    for (int j = 0; j < 4; ++j) v[j] = readlane63_f64(v[j]);
}
// three maxima at once, interleaved step by step
__device__ __forceinline__ void wave_max3_f64(double& a, double& b, double& c) {  // all >= 0  (wave_reduce.hip.h)
    wr::wave_reduce_max3_f64_raw(a, b, c);
    a = readlane63_f64(a); b = readlane63_f64(b); c = readlane63_f64(c);
}

// the prior part's pair walk ends in three float32 maxima, three float64 maxima and one float64 sum:
// seven chains of six dependent DPP steps -- issued one reduction after the other they were three
// times six latencies, interleaved step by step they are six
__device__ __forceinline__ void wave_bounds_reduce(float& a, float& b, float& c, double& p, double& q, double& r,
                                                   double& sum) {
    wr::wave_reduce_bounds_raw(a, b, c, p, q, r, sum);
    a = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(a), 63));
    b = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(b), 63));
    c = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(c), 63));
    p = readlane63_f64(p); q = readlane63_f64(q); r = readlane63_f64(r);
    sum = readlane63_f64(sum);
}

// Inclusive scans over the 64 lanes without the LDS crossbar.  Prefix (lane 0 first): four
// zero-filled row_shr steps give each 16-lane row its own prefix, row_bcast:15 / row_bcast:31 add
// the totals of the rows below.  Suffix (lane 63 first): four row_shl steps, then the totals of
// the rows above (lanes 16, 32, 48 hold them) by readlane -- there is no row_bcast downwards.
__device__ __forceinline__ double wave_prefix_dpp_f64(double v) {
    v += dpp_f64<0x111>(0.0, v);       // row_shr:1
    v += dpp_f64<0x112>(0.0, v);       // row_shr:2
    v += dpp_f64<0x114>(0.0, v);       // row_shr:4
    v += dpp_f64<0x118>(0.0, v);       // row_shr:8
    v += dpp_f64<0x142, 0xA>(0.0, v);  // row_bcast:15 into rows 1,3
    v += dpp_f64<0x143, 0xC>(0.0, v);  // row_bcast:31 into rows 2,3
    return v;
}
__device__ __forceinline__ double readlane_f64(double v, int l) {
    const long long x = __double_as_longlong(v);
    const int lo = __builtin_amdgcn_readlane((int)x, l);
    const int hi = __builtin_amdgcn_readlane((int)(x >> 32), l);
    return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}
// ... and the combine of the waves' records: eight values in lanes 0..7 (zeros elsewhere), three
// maxima and a sum in THREE DPP steps (two quad_perms leave every lane with its quad's result;
// row_ror:4 hands lane i the value of lane i - 4, so lanes 4..7 end with the result of all eight --
// lanes 0..3 receive the zeros of lanes 12..15), interleaved; results from lane 4
__device__ __forceinline__ void lanes8_max3_sum(double& p, double& q, double& r, double& sum, float& a, float& b,
                                                float& c) {
    wr::lanes8_reduce_bounds_raw(a, b, c, p, q, r, sum);
    p = readlane_f64(p, 4); q = readlane_f64(q, 4); r = readlane_f64(r, 4); sum = readlane_f64(sum, 4);
    a = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(a), 4));
    b = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(b), 4));
    c = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(c), 4));
}
__device__ __forceinline__ double wave_suffix_dpp_f64(double v, int lane) {
    v += dpp_f64<0x101>(0.0, v);       // row_shl:1
    v += dpp_f64<0x102>(0.0, v);       // row_shl:2
    v += dpp_f64<0x104>(0.0, v);       // row_shl:4
    v += dpp_f64<0x108>(0.0, v);       // row_shl:8
    const double t1 = readlane_f64(v, 16), t2 = readlane_f64(v, 32), t3 = readlane_f64(v, 48);
    const double hi = t2 + t3;
    return v + (lane < 16 ? t1 + hi : lane < 32 ? hi : lane < 48 ? t3 : 0.0);
}
// NV independent sums, interleaved step by step like wave_sum4_f64
template <int NV>
__device__ __forceinline__ void wave_sumN_f64(double (&v)[NV]) {
    if constexpr (NV == 2) {   // (written out: wave_reduce.hip.h)
        wr::wave_reduce_sum2_f64_raw(v[0], v[1]);
        v[0] = readlane63_f64(v[0]);
        v[1] = readlane63_f64(v[1]);
        return;
    }
#pragma unroll
    for (int j = 0; j < NV; ++j) v[j] += dpp_f64<0xB1>(0.0, v[j]);
#pragma unroll
    for (int j = 0; j < NV; ++j) v[j] += dpp_f64<0x4E>(0.0, v[j]);
#pragma unroll
    for (int j = 0; j < NV; ++j) v[j] += dpp_f64<0x124>(0.0, v[j]);
#pragma unroll
    for (int j = 0; j < NV; ++j) v[j] += dpp_f64<0x128>(0.0, v[j]);
#pragma unroll
    for (int j = 0; j < NV; ++j) v[j] += dpp_f64<0x142, 0xA>(0.0, v[j]);
#pragma unroll
    for (int j = 0; j < NV; ++j) v[j] += dpp_f64<0x143, 0xC>(0.0, v[j]);
#pragma unroll
    for (int j = 0; j < NV; ++j) v[j] = readlane63_f64(v[j]);
}
// NV independent sums over each 16-lane ROW (every lane of a row gets its row's totals):
// quad_perm, quad_perm, row_ror:4, row_ror:8 -- four steps, no cross-row traffic
template <int NV>
__device__ __forceinline__ void row_sum_f64(double (&v)[NV]) {
    if constexpr (NV == 6) {   // (written out: wave_reduce.hip.h)
        wr::row_reduce_sum6_f64_raw(v[0], v[1], v[2], v[3], v[4], v[5]);
        return;
    }
#pragma unroll
    for (int j = 0; j < NV; ++j) v[j] += dpp_f64<0xB1>(0.0, v[j]);
#pragma unroll
    for (int j = 0; j < NV; ++j) v[j] += dpp_f64<0x4E>(0.0, v[j]);
#pragma unroll
    for (int j = 0; j < NV; ++j) v[j] += dpp_f64<0x124>(0.0, v[j]);
#pragma unroll
    for (int j = 0; j < NV; ++j) v[j] += dpp_f64<0x128>(0.0, v[j]);
}
__device__ __forceinline__ double wave_max_f64(double v) {  // v >= 0  (written out: wave_reduce.hip.h)
    wr::wave_reduce_max1_f64_raw(v);
    return readlane63_f64(v);
}

// Workgroup sum of NV doubles (all threads get the totals).  scratch: WAVES*NV doubles.
// (The unrolled serial combine is one round of independent LDS reads; a lane-parallel read +
// second DPP reduction measured slower.)
// a barrier for LDS traffic alone: __syncthreads() also waits for the wave's outstanding global stores (a memory
// round trip when write-through stores have just been issued)
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
// LDS_ONLY: see lds_barrier (tail_general: the rows' re-arming stores and the gradient's are in flight)
template <int NV, bool LDS_ONLY = false>
__device__ __forceinline__ void block_sum(double (&v)[NV], double* scratch, int tid) {
    const int lane = tid & 63, wave = tid >> 6;
    if constexpr (NV >= 4) {   // (four chains of DPP steps at a time: wave_reduce.hip.h)
        double q4[4] = {v[0], v[1], v[2], v[3]};
        wave_sum4_f64(q4);
        v[0] = q4[0]; v[1] = q4[1]; v[2] = q4[2]; v[3] = q4[3];
        if constexpr (NV == 6) {
            wave_sum2_f64(v[4], v[5]);
        } else {
#pragma unroll
            for (int i = 4; i < NV; ++i) v[i] = wave_sum_f64(v[i]);
        }
    } else if constexpr (NV == 2) {
        wave_sum2_f64(v[0], v[1]);
    } else {
#pragma unroll
        for (int i = 0; i < NV; ++i) v[i] = wave_sum_f64(v[i]);
    }
    if (LDS_ONLY) lds_barrier(); else __syncthreads();
    if (lane == 0) {
#pragma unroll
        for (int i = 0; i < NV; ++i) scratch[wave * NV + i] = v[i];
    }
    if (LDS_ONLY) lds_barrier(); else __syncthreads();
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        double s = 0.0;
#pragma unroll
        for (int wv = 0; wv < WAVES; ++wv) s += scratch[wv * NV + i];
        v[i] = s;
    }
}

// acc layout in LDS: att[T1] | def[T1] | ha[T1]   (T1 = T+1, slot T = padding sink)
//   att[t] = sum_{h_i=t} sh_i + sum_{a_i=t} sa_i
//   def[t] = sum_{a_i=t} sh_i + sum_{h_i=t} sa_i
//   ha[t]  = sum_{h_i=t} sh_i
__device__ __forceinline__ void flush_run(double* acc, int T1, uint32_t key, float sh,
                                          float sa) {
    const int h = key & 0xFFFFu, a = key >> 16;
    const double dh = (double)sh, da = (double)sa;
    atomicAdd(&acc[h], dh);
    atomicAdd(&acc[2 * T1 + h], dh);
    atomicAdd(&acc[T1 + a], dh);
    atomicAdd(&acc[a], da);
    atomicAdd(&acc[T1 + h], da);
}

// dc_eval accumulates in FIXED POINT: every float32 addend is rounded to a multiple of 2^-30 when
// it is widened (it already is one from 2^-7 up) and kept in units of 2^-30, so every partial sum
// -- lane, wave, workgroup, accumulator row -- is an exact integer and the result does not depend
// on how the fixtures are partitioned over waves and workgroups (nor on the arrival order).
__device__ __forceinline__ double q30(float x) { return rint(ldexp((double)x, 30)); }
__device__ __forceinline__ void flush_run_q(double* acc, int T1, uint32_t key, float sh, float sa) {
    const int h = key & 0xFFFFu, a = key >> 16;
    const double dh = q30(sh), da = q30(sa);
    atomicAdd(&acc[h], dh);
    atomicAdd(&acc[2 * T1 + h], dh);
    atomicAdd(&acc[T1 + a], dh);
    atomicAdd(&acc[a], da);
    atomicAdd(&acc[T1 + h], da);
}

// write-through (sc1) store / L1-bypassing (sc1) load of one double: the in-launch
// hand-off between workgroups (cdna_hip_programming.md Guideline 16, split-K form)
__device__ __forceinline__ double ld_sc1(const double* p) {
    const unsigned long long u =
        __hip_atomic_load(reinterpret_cast<const unsigned long long*>(p), __ATOMIC_RELAXED,
                          __HIP_MEMORY_SCOPE_AGENT);
    return __longlong_as_double((long long)u);
}
__device__ __forceinline__ void st_sc1(double* p, double v) {
    __hip_atomic_store(reinterpret_cast<unsigned long long*>(p),
                       (unsigned long long)__double_as_longlong(v), __ATOMIC_RELAXED,
                       __HIP_MEMORY_SCOPE_AGENT);
}

// two doubles (16-byte aligned) in ONE write-through store.  The s_nop is the hazard the compiler
// would cover for its own stores and cannot see inside the asm: a store of more than 64 bits
// reads its data registers a few cycles after issue, and the next VALU write to them (the
// allocator reuses them at once) would change what is stored -- found as cell records that
// depended on how the stored expression was written.
typedef double double2_t __attribute__((ext_vector_type(2)));
// the matching L1-bypassing 16-byte load (the wait is part of it, as in ga_load)
__device__ __forceinline__ double2_t ld_sc1_x2(const double* p) {
    double2_t v;
    asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    return v;
}
// four of them in ONE round trip
__device__ __forceinline__ void ld_sc1_x2_4(const double* p0, const double* p1, const double* p2, const double* p3,
                                            double2_t (&v)[4]) {
    asm volatile("global_load_dwordx4 %0, %4, off sc1\n\tglobal_load_dwordx4 %1, %5, off sc1\n\t"
                 "global_load_dwordx4 %2, %6, off sc1\n\tglobal_load_dwordx4 %3, %7, off sc1\n\ts_waitcnt vmcnt(0)"
                 : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3]) : "v"(p0), "v"(p1), "v"(p2), "v"(p3) : "memory");
}
__device__ __forceinline__ void st_sc1_x2(double* p, double a, double b) {
    double2_t v = {a, b};
    asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 2" ::"v"(p), "v"(v) : "memory");
}

// ---- counted accumulator rows (see GA_ROW): add one double, read and re-arm one row
// an integer-valued double |x| < 2^51 as int64: the low mantissa bits of x + 1.5 * 2^52
__device__ __forceinline__ long long exact_i64(double x) {
    return __double_as_longlong(x + 6755399441055744.0) - 0x4338000000000000ll;
}
// one contribution of one workgroup to one row: flag / hi first, the COUNTED lo last (same line:
// the memory channel performs them in this order)
__device__ __forceinline__ void ga_add(long long* row, double v /* in units of 2^-30 */) {
    double h = 0.0, r = 0.0;
    if (!(fabs(v) < GA_LIMIT * GA_LO_SCALE)) {  // inf, nan, or beyond the hi word's range
        (void)__hip_atomic_fetch_or(reinterpret_cast<unsigned long long*>(row),
                                    (unsigned long long)((v == -__builtin_inf()) ? GA_NEGINF : GA_BAD) << GA_FLAG_SHIFT,
                                    __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else {
        h = rint(v * (1.0 / (GA_HI_UNIT * GA_LO_SCALE)));  // |h| < 2^50
        r = fma(-h, GA_HI_UNIT * GA_LO_SCALE, v);          // exact, |r| <= 2^43
        if (h != 0.0) {
            (void)__hip_atomic_fetch_add(row + 1, exact_i64(h), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            // (round 4) The reader takes a complete COUNT in the lo word as proof that every hi
            // contribution has landed.  What orders the two atomics is the hardware: both target ONE 16-byte
            // granule of one line, i.e. one memory channel, and a wave's requests to a channel are served in
            // issue order; the memory model does not promise it.  A release fence between them was tried
            // (thinking this branch rare -- it is not: at N = 1e6 a workgroup's per-team sums exceed the lo
            // word's 2^43 units, so most adds come through here): dc_eval 5.8 -> 7.3 us, the persistent chain
            // 7.3 -> 9.9 us per leapfrog (profiles/r04/ab_strict_order.txt).  Kept relaxed; the guard is
            // empirical: the wild-region parity tests and the bit-identical soak run through this branch.
            if (DC_STRICT_ORDER) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        }
    }
    (void)__hip_atomic_fetch_add(row, exact_i64(r) + GA_BIAS + (1ll << GA_COUNT_SHIFT), DC_ARRIVE_ORDER,
                                 __HIP_MEMORY_SCOPE_AGENT);
}
struct GaWords {
    long long lo, hi;
};
// ONE L2-bypassing 16-byte load (the wait is part of it: the compiler does not count loads it
// cannot see -- and vector loads return in order, so nothing younger is held up by waiting here)
__device__ __forceinline__ GaWords ga_load(const long long* row) {
    typedef long long i64x2 __attribute__((ext_vector_type(2)));
    i64x2 v;
    asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(row) : "memory");
    GaWords w;
    w.lo = v.x;
    w.hi = v.y;
    return w;
}
// the accumulator rows of a chain: TWO sets.  Plain launches use set 0 (the kernel boundary orders
// the tail's re-arm stores before the next launch's atomics); the persistent kernel alternates by
// step parity, and the tail, while it waits for this step's set to fill, also waits until the OTHER
// set -- the next step's, re-armed one step ago -- reads all zero, BEFORE it publishes the next
// position: no store of a re-arm can land on a row that is being added to, and a row that looks
// complete is never a stale one.
__host__ __device__ inline size_t ga_set_words(int T) { return (size_t)ga_rows(T) * GA_ROW; }
// two rows in ONE round trip (both loads issued, then one wait)
__device__ __forceinline__ void ga_load2(const long long* row_a, const long long* row_b, GaWords* a, GaWords* b) {
    typedef long long i64x2 __attribute__((ext_vector_type(2)));
    i64x2 va, vb;
    asm volatile("global_load_dwordx4 %0, %2, off sc1\n\tglobal_load_dwordx4 %1, %3, off sc1\n\ts_waitcnt vmcnt(0)"
                 : "=&v"(va), "=&v"(vb) : "v"(row_a), "v"(row_b) : "memory");
    a->lo = va.x; a->hi = va.y;
    b->lo = vb.x; b->hi = vb.y;
}
// four rows in ONE round trip (larger leagues: a thread of the tail takes several rows; one after the
// other they were a memory round trip each)
__device__ __forceinline__ void ga_load4(const long long* r0, const long long* r1, const long long* r2,
                                         const long long* r3, GaWords (&w)[4]) {
    typedef long long i64x2 __attribute__((ext_vector_type(2)));
    i64x2 v0, v1, v2, v3;
    asm volatile("global_load_dwordx4 %0, %4, off sc1\n\tglobal_load_dwordx4 %1, %5, off sc1\n\t"
                 "global_load_dwordx4 %2, %6, off sc1\n\tglobal_load_dwordx4 %3, %7, off sc1\n\ts_waitcnt vmcnt(0)"
                 : "=&v"(v0), "=&v"(v1), "=&v"(v2), "=&v"(v3) : "v"(r0), "v"(r1), "v"(r2), "v"(r3) : "memory");
    w[0].lo = v0.x; w[0].hi = v0.y; w[1].lo = v1.x; w[1].hi = v1.y;
    w[2].lo = v2.x; w[2].hi = v2.y; w[3].lo = v3.x; w[3].hi = v3.y;
}
__device__ __forceinline__ void ga_load3(const long long* r0, const long long* r1, const long long* r2,
                                         GaWords (&w)[4]) {
    typedef long long i64x2 __attribute__((ext_vector_type(2)));
    i64x2 v0, v1, v2;
    asm volatile("global_load_dwordx4 %0, %3, off sc1\n\tglobal_load_dwordx4 %1, %4, off sc1\n\t"
                 "global_load_dwordx4 %2, %5, off sc1\n\ts_waitcnt vmcnt(0)"
                 : "=&v"(v0), "=&v"(v1), "=&v"(v2) : "v"(r0), "v"(r1), "v"(r2) : "memory");
    w[0].lo = v0.x; w[0].hi = v0.y; w[1].lo = v1.x; w[1].hi = v1.y; w[2].lo = v2.x; w[2].hi = v2.y;
    w[3] = w[0];
}
__device__ __forceinline__ void ga_load2rows(const long long* r0, const long long* r1, GaWords (&w)[4]) {
    typedef long long i64x2 __attribute__((ext_vector_type(2)));
    i64x2 v0, v1;
    asm volatile("global_load_dwordx4 %0, %2, off sc1\n\tglobal_load_dwordx4 %1, %3, off sc1\n\ts_waitcnt vmcnt(0)"
                 : "=&v"(v0), "=&v"(v1) : "v"(r0), "v"(r1) : "memory");
    w[0].lo = v0.x; w[0].hi = v0.y; w[1].lo = v1.x; w[1].hi = v1.y;
    w[2] = w[0];
    w[3] = w[0];
}
__device__ __forceinline__ int ga_count(const GaWords& w) { return (int)((unsigned long long)w.lo >> GA_COUNT_SHIFT); }
__device__ __forceinline__ bool ga_is_zero(const GaWords& w) { return (w.lo | w.hi) == 0; }
__device__ __forceinline__ void ga_rearm(long long* row) {  // write-through zeros for the next evaluation
    __hip_atomic_store(row, 0ll, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(row + 1, 0ll, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// the value of a complete row (real units); a raised flag replaces it: -inf (a clipped tau term:
// log 0) or NaN (anything else)
__device__ __forceinline__ double ga_value(const GaWords& w) {
    const long long cnt = (long long)((unsigned long long)w.lo >> GA_COUNT_SHIFT);
    const unsigned int fl = (unsigned int)((unsigned long long)w.lo >> GA_FLAG_SHIFT) & 3u;
    const long long lo = (w.lo & ((1ll << GA_FLAG_SHIFT) - 1)) - cnt * GA_BIAS;
    double v = fma((double)w.hi, GA_HI_UNIT, (double)lo * (1.0 / GA_LO_SCALE));
    if (fl != 0u) v = (fl & GA_BAD) ? __builtin_nan("") : -__builtin_inf();
    return v;
}
constexpr int ARRIVE_SPIN_LIMIT = 1 << 18;   // bounded waits of the tail for the streaming workgroups (~0.1 s)

// ---------------------------------------------------------------- LDS footprints

__host__ __device__ inline int tab_len(int T) { return (T + 2) & ~1; }  // >= T+1, even
__host__ __device__ inline size_t stream_lds_bytes(int T) {
    const int T1 = T + 1;
    size_t b = 2 * (size_t)tab_len(T) * 8;                 // tabH, tabA (float2)
    b += (size_t)(3 * T1 + ((3 * T1) & 1)) * 8;            // acc
    b += (size_t)WAVES * N_SCAL * 8;                       // red
    b += (size_t)WAVES * 4 * 4;                            // redm (float)
    b += 16 * 4;                                           // flag
    return b;
}
// tail: everything it needs is staged into LDS in one round of loads
__host__ __device__ inline int xs_staged_k(int K) { return (K > 0 && K <= 16) ? K : 0; }
__host__ __device__ inline size_t tail_lds_bytes(int T, int D, int K, int zo_stride, int n_wg,
                                                 int total_c, bool staged) {
    size_t d = (size_t)zo_stride + 3 * (size_t)T + D + (3 * (size_t)T + N_SCAL + 4) +
               WAVES * 8 + (size_t)n_wg * N_SCAL + (size_t)T * xs_staged_k(K) /* xs, K <= 16 staged */ +
               (size_t)D + 8 /* hand-over to the NUTS leaf */ +
               (staged && D > 64 ? (size_t)nd::LEAF_STAGE_VECS * D : 0) /* its vectors, when not in registers */;
    size_t i = 3 * (size_t)T + 2;
    if (staged) d += (size_t)total_c;
    return d * 8 + ((i * 4 + 15) & ~(size_t)15) + 16;
}
__host__ __device__ inline size_t prior_lds_bytes(int T) {
    size_t b = 2 * (size_t)tab_len(T) * 8;                 // tabH, tabA (float2)
    b += 3 * (size_t)T * 8;                                // true tables (double)
    b += 3 * (size_t)T * 8;                                // att, def, ha (double)
    b += (40 + WAVES * 8 + WAVES * 8) * 8;                 // scalars, scratch, argmax
    return b;
}
// persistent kernel: the state of the leaf being booked one step behind its evaluation, parked in LDS by each
// of the two leaf waves (dc_eval_loop, leaf_window_*): header words [64] | invM, zn, r_half, r_sum, sl_r,
// sr_r [64 lanes each, per 64 elements] | the leaf's rng words [8]
constexpr int LEAF_PARK_VECS = 6;
__host__ __device__ inline size_t leaf_park_doubles(int D) {
    return 64 + (size_t)LEAF_PARK_VECS * 64 * ((D + 63) / 64) + 8;
}
__host__ __device__ inline size_t acc_tail_lds_bytes(int T, int D, int K, int zo_stride, bool stage);
__host__ __device__ inline size_t eval_lds_bytes(int T, int D, int K, int zo_stride, bool stage) {
    size_t b = stream_lds_bytes(T);
    // (the prior workgroup runs the tail: its scratch sits behind the tail's arrays)
    const size_t t = ((acc_tail_lds_bytes(T, D, K, zo_stride, stage) + 15) & ~(size_t)15) + prior_lds_bytes(T);
    return b > t ? b : t;
}

// ---------------------------------------------------------------- float32 tables
// The streaming workgroups and the prior workgroup run exactly this code, so the prior
// workgroup knows the table values bit for bit.

struct F32Scalars {
    float s_a, s_d, s_h, q, m, mha, gam;
};
__device__ __forceinline__ float exp_f32(float x) {
    return __builtin_amdgcn_exp2f(x * 1.44269504088896341f);
}
// SC1: z is read with L1-bypassing vector loads (the persistent evaluation kernel: the position
// was written by another CU of the SAME launch, and a uniform-address plain load would be a
// scalar load through the non-coherent scalar cache)
template <bool SC1>
__device__ __forceinline__ double zld(const double* p) {
    return SC1 ? ld_sc1(p) : *p;
}
template <bool EXT, bool SC1 = false>
__device__ __forceinline__ F32Scalars f32_scalars(const Layout& L, const double* z) {
    F32Scalars s;
    s.s_a = exp_f32((float)zld<SC1>(&z[L.o_sa]));
    s.s_d = exp_f32((float)zld<SC1>(&z[L.o_sd]));
    s.m = (float)zld<SC1>(&z[L.o_md]);
    const float zc = (float)zld<SC1>(&z[L.o_corr]);
    float q = __builtin_amdgcn_rcpf(1.0f + exp_f32(-zc));
    s.q = fminf(fmaxf(q, (float)SIG_LO), (float)SIG_HI);
    if (!EXT) {
        s.gam = (float)zld<SC1>(&z[L.o_ha]);
        s.s_h = 0.f;
        s.mha = 0.f;
    } else {
        s.gam = 0.f;
        s.s_h = exp_f32((float)zld<SC1>(&z[L.o_sh]));
        s.mha = (float)zld<SC1>(&z[L.o_mha]);
    }
    return s;
}
// the per-team entries of z a table entry needs (basic: attack, defence; extended: + home adv.)
struct TeamZ {
    double a, d, h;
};
template <bool EXT, bool SC1 = false>
__device__ __forceinline__ TeamZ load_team_z(const Layout& L, const double* z, int t) {
    TeamZ v;
    v.a = zld<SC1>(&z[(EXT ? L.o_sat : L.o_adec) + t]);
    v.d = zld<SC1>(&z[(EXT ? L.o_sdt : L.o_ddec) + t]);
    v.h = EXT ? zld<SC1>(&z[L.o_hadec + t]) : 0.0;
    return v;
}
// xs_at(t, k): the standardised covariate as float32 (a float copy in memory, or -- persistent
// kernel, tail workgroup -- the LDS-resident float64 copy narrowed: the same value)
struct XsFromF32 {
    const float* p;
    int K;
    __device__ __forceinline__ float operator()(int t, int k) const { return p[(size_t)t * K + k]; }
};
struct XsFromF64 {
    const double* p;
    int K;
    __device__ __forceinline__ float operator()(int t, int k) const { return (float)p[(size_t)t * K + k]; }
};
template <bool EXT, bool SC1 = false, class XS>
__device__ __forceinline__ void f32_table_entry_x(const Layout& L, const F32Scalars& s,
                                                  const double* z, XS xs_at, int t,
                                                  const TeamZ& tz, float2* vh, float2* va) {
    float att, def, ha;
    if (!EXT) {
        att = s.s_a * (float)tz.a;
        def = s.m + s.s_d * (float)tz.d;
        ha = s.gam;
    } else {
        float apm = 0.f, dpm = s.m;
        for (int k = 0; k < L.K; ++k) {
            const float xv = xs_at(t, k);
            apm += xv * (float)zld<SC1>(&z[L.o_bA + k]);
            dpm += xv * (float)zld<SC1>(&z[L.o_bD + k]);
        }
        att = apm + (float)tz.a * s.s_a;
        def = dpm + (float)tz.d * s.s_d;
        ha = s.mha + s.s_h * (float)tz.h;
    }
    const float edn = exp_f32(-def);
    *vh = make_float2(exp_f32(att + ha), edn);
    *va = make_float2(exp_f32(att), edn);
}
template <bool EXT, bool SC1 = false>
__device__ __forceinline__ void f32_table_entry(const Layout& L, const F32Scalars& s,
                                                const double* z, const float* xsf, int t,
                                                const TeamZ& tz, float2* vh, float2* va) {
    f32_table_entry_x<EXT, SC1>(L, s, z, XsFromF32{xsf, L.K}, t, tz, vh, va);
}
// `first`: team `tid`'s entries of z, loaded by the caller BEFORE its bulk loads -- vector loads
// return in order, and behind a tile's worth of fixture loads they would wait for HBM
// SMALLT (at most 64 teams, fewer than threads): a thread builds at most ONE entry, from `first` --
// no load inside the loop.  With the loop, the entry's z values are a phi of `first` and an in-loop
// load, and the compiler waits for EVERY outstanding load (the first tile's among them) before it
// uses them.
template <bool EXT, bool SC1 = false, bool SMALLT = false, class XS>
__device__ __forceinline__ void build_tables_f32_x(const Layout& L, const double* z,
                                                   XS xs_at, float2* tabH, float2* tabA,
                                                   int tid, const TeamZ& first, const F32Scalars& s) {
    if (SMALLT) {
        if (tid <= L.T) {
            float2 vh = make_float2(0.f, 0.f), va = vh;
            if (tid < L.T) f32_table_entry_x<EXT, SC1>(L, s, z, xs_at, tid, first, &vh, &va);
            tabH[tid] = vh;
            tabA[tid] = va;
        }
        return;
    }
    for (int t = tid; t <= L.T; t += BLOCK) {
        float2 vh = make_float2(0.f, 0.f), va = vh;
        if (t < L.T) f32_table_entry_x<EXT, SC1>(L, s, z, xs_at, t, t == tid ? first : load_team_z<EXT, SC1>(L, z, t), &vh, &va);
        tabH[t] = vh;
        tabA[t] = va;
    }
}
template <bool EXT, bool SC1 = false, bool SMALLT = false>
__device__ __forceinline__ void build_tables_f32(const Layout& L, const double* z,
                                                 const float* xsf, float2* tabH, float2* tabA,
                                                 int tid, const TeamZ& first, const F32Scalars& s) {
    build_tables_f32_x<EXT, SC1, SMALLT>(L, z, XsFromF32{xsf, L.K}, tabH, tabA, tid, first, s);
}
// tau argument of one score class in float32, and the threshold below which a class is left to the
// tail workgroup's float64 pass (class_terms, ill_pass)
constexpr float TAU_ILL = 1.0f / 64.0f;
__device__ __forceinline__ float class_arg(float rho, float c) { return fmaf(rho, c, 1.0f); }
// rho in float32 from the three float32 maxima (identical code in stream and prior)
__device__ __forceinline__ float rho_f32(float mP, float mQ, float mR, float q) {
    const float ub = mP > 1.0f ? __builtin_amdgcn_rcpf(mP) : 1.0f;
    const float lb = -__builtin_amdgcn_rcpf(fmaxf(mQ, mR));
    return fmaf(q, ub - lb, lb);
}
// ---- maxima over a COMPLETE pair table (every ordered pair h != a present: a league where
// everybody has hosted everybody): the rates are products of per-team table entries,
// lh(h,a) = AH_h * BD_a, la(h,a) = AA_a * BD_h, and rounding is monotone, so
//   max lh = the largest AH times the largest BD of a DIFFERENT team -- exact, from the top two of each;
//   max lh*la: the pair of the largest P_h = AH_h*BD_h and the largest Q_a = AA_a*BD_a of a different
//   team, evaluated with the per-pair expression (equal to the walk over all pairs unless two
//   candidates tie within an ulp -- and every workgroup computes the same value either way).
// O(T) instead of O(pairs) per workgroup: at T = 200 the walks were 6 us of every streaming workgroup
// and 13.5 us (float64, with arg-pairs) of the prior workgroup.  Taken from DENSE_MIN_PAIRS pairs on
// (more than 90 teams): below, a few pairs per thread are the cheaper walk (measured at 48 and 64 teams).
constexpr int DENSE_MIN_PAIRS = 4096;   // (round 4: from 8192 -- every complete table past 64 teams; 80 teams walked theirs in 10.4 us, 100 teams took 8.3)
// the largest and the second largest entry (of a different team) of ONE per-team array, with
// their teams: lanes stride over the teams, then two wave maxima.  get(t): team t's value (> 0).
// Teams: -1 when there is none.  Results wave-uniform.
template <class F>
struct Top2 {
    F m1, m2;
    int i1, i2;
};
__device__ __forceinline__ float wave_max_pos(float v) { return wave_max_f32(v); }
__device__ __forceinline__ double wave_max_pos(double v) { return wave_max_f64(v); }
template <class F, class Get>
__device__ __forceinline__ Top2<F> wave_top2(int T, int lane, Get get) {
    F a1 = (F)0, a2 = (F)0;
    int j1 = -1, j2 = -1;
    // (four teams per lane and round, all of them requested before the first is looked at -- index clamped, the
    // repeat does not take part: one team per trip was two dependent LDS round trips per trip, 0.17 us each
    // at 200 teams; the trip count is wave uniform)
    for (int b = 0; b < T; b += 256) {
        F v[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] = get(min(b + 64 * k + lane, T - 1));
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int t = b + 64 * k + lane;
            const bool first = t < T && v[k] > a1, second = t < T && !first && v[k] > a2;
            a2 = first ? a1 : second ? v[k] : a2;
            j2 = first ? j1 : second ? t : j2;
            a1 = first ? v[k] : a1;
            j1 = first ? t : j1;
        }
    }
    Top2<F> R;
    R.m1 = wave_max_pos(a1);
    const unsigned long long b1 = __ballot(a1 == R.m1 && j1 >= 0);
    const int win = b1 ? __ffsll((long long)b1) - 1 : 0;
    R.i1 = b1 ? __builtin_amdgcn_readlane(j1, win) : -1;
    const F c = lane == win ? a2 : a1;
    const int jc = lane == win ? j2 : j1;
    R.m2 = wave_max_pos(c);
    const unsigned long long b2 = __ballot(c == R.m2 && jc >= 0);
    R.i2 = b2 ? __builtin_amdgcn_readlane(jc, b2 ? __ffsll((long long)b2) - 1 : 0) : -1;
    R.m2 = b2 ? R.m2 : (F)0;
    return R;
}
// two float32 arrays at once (the prior part's waves that take two jobs: one pass over the table entries for
// both, the maxima of both reduced step by step together -- a wave issues a dependent DPP + max only every ~20
// cycles; two jobs one after the other were the longest pole of the bounds past 64 teams).  Results as wave_top2's.
template <class Get2>
__device__ __forceinline__ void wave_top2_pair_f32(int T, int lane, Get2 get2, Top2<float>* RA, Top2<float>* RB) {
    float a1[2] = {0.f, 0.f}, a2[2] = {0.f, 0.f};
    int j1[2] = {-1, -1}, j2[2] = {-1, -1};
    for (int b = 0; b < T; b += 256) {
        float v[4][2];
#pragma unroll
        for (int k = 0; k < 4; ++k) get2(min(b + 64 * k + lane, T - 1), &v[k][0], &v[k][1]);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int t = b + 64 * k + lane;
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const bool first = t < T && v[k][q] > a1[q], second = t < T && !first && v[k][q] > a2[q];
                a2[q] = first ? a1[q] : second ? v[k][q] : a2[q];
                j2[q] = first ? j1[q] : second ? t : j2[q];
                a1[q] = first ? v[k][q] : a1[q];
                j1[q] = first ? t : j1[q];
            }
        }
    }
    Top2<float> R[2];
    float m0 = a1[0], m1 = a1[1], d0 = 0.f;
    wave_max3_f32(m0, m1, d0);
    R[0].m1 = m0; R[1].m1 = m1;
    float c[2];
    int jc[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const unsigned long long b1 = __ballot(a1[q] == R[q].m1 && j1[q] >= 0);
        const int win = b1 ? __ffsll((long long)b1) - 1 : 0;
        R[q].i1 = b1 ? __builtin_amdgcn_readlane(j1[q], win) : -1;
        c[q] = lane == win ? a2[q] : a1[q];
        jc[q] = lane == win ? j2[q] : j1[q];
    }
    float n0 = c[0], n1 = c[1], d1 = 0.f;
    wave_max3_f32(n0, n1, d1);
    R[0].m2 = n0; R[1].m2 = n1;
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const unsigned long long b2 = __ballot(c[q] == R[q].m2 && jc[q] >= 0);
        R[q].i2 = b2 ? __builtin_amdgcn_readlane(jc[q], b2 ? __ffsll((long long)b2) - 1 : 0) : -1;
        R[q].m2 = b2 ? R[q].m2 : 0.f;
    }
    *RA = R[0];
    *RB = R[1];
}
// max over h != a of x_h * y_a from the top two of x and of y; *ph, *pa: the pair (or -1)
template <class F>
__device__ __forceinline__ F dense_pair_max(F x1, int ix1, F x2, int ix2, F y1, int iy1, F y2, int iy2,
                                            int* ph, int* pa) {
    if (ix1 != iy1) { *ph = ix1; *pa = iy1; return x1 * y1; }
    const F c12 = ix1 >= 0 && iy2 >= 0 ? x1 * y2 : (F)0, c21 = ix2 >= 0 && iy1 >= 0 ? x2 * y1 : (F)0;
    if (c12 >= c21) { *ph = ix1; *pa = iy2; return c12; }
    *ph = ix2; *pa = iy1;
    return c21;
}
// the five per-team arrays whose top two decide the maxima: 0 AH = tabH.x | 1 AA = tabA.x |
// 2 BD = tabH.y (= tabA.y) | 3 P = AH*BD | 4 Q = AA*BD
constexpr int DENSE_ARRAYS = 5;
// float32 maxima of a complete pair table: waves 0..4 take one array each (a wave's top two are two
// DPP maxima; all five on one wave, or on every wave, were as slow as walking 10 000 pairs), one
// LDS record each, one barrier, then every thread evaluates the three candidates.  false: a rate
// reaches the clip -- the clipped product is not separable, the caller walks the pairs.
// rec: 5 x 4 floats (m1, m2, i1, i2 as bit patterns).
// PRIOR (the prior part, whose waves 0..4 take the float64 arrays at the same time): the float32 arrays go
// to waves 5 (0, 1), 6 (2, 3) and 7 (4) -- one after the other on waves 0..4 the two precisions were 2.6 us
// of that workgroup at 200 teams.
struct NoHook {
    __device__ __forceinline__ void operator()() const {}
};
// AFTER: called by every wave right behind the barrier (the prior part past 64 teams: the rows' early poll)
template <bool CLIP, bool PRIOR = false, class AFTER = NoHook>
__device__ __forceinline__ bool dense_maxima_f32(int T, const float2* tabH, const float2* tabA, float* rec,
                                                 int tid, float* oP, float* oQ, float* oR,
                                                 unsigned long long* dbg = nullptr, AFTER after = AFTER()) {
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // (uniform branches below)
    auto job = [&](int j) {
        const Top2<float> R = wave_top2<float>(T, lane, [&](int t) {
            const float2 th = tabH[t], ta = tabA[t];
            return j == 0 ? th.x : j == 1 ? ta.x : j == 2 ? th.y : j == 3 ? th.x * th.y : ta.x * ta.y;
        });
        if (lane == 0) {
            rec[j * 4 + 0] = R.m1;
            rec[j * 4 + 1] = R.m2;
            rec[j * 4 + 2] = __int_as_float(R.i1);
            rec[j * 4 + 3] = __int_as_float(R.i2);
        }
    };
    if (!PRIOR) {
        if (wave < DENSE_ARRAYS) job(wave);
    } else if (wave >= DENSE_ARRAYS) {
        static_assert(WAVES == 8 && DENSE_ARRAYS == 5, "float32 arrays 0..4 on waves 5, 5, 6, 6, 7");
        if (wave < WAVES - 1) {   // arrays (0, 1) on wave 5, (2, 3) on wave 6: both in one pass
            const int j0 = 2 * (wave - DENSE_ARRAYS);
            Top2<float> Ra, Rb;
            wave_top2_pair_f32(T, lane, [&](int t, float* va, float* vb) {
                const float2 th = tabH[t], ta = tabA[t];
                *va = j0 == 0 ? th.x : th.y;
                *vb = j0 == 0 ? ta.x : th.x * th.y;
            }, &Ra, &Rb);
            if (lane == 0) {
                rec[j0 * 4 + 0] = Ra.m1; rec[j0 * 4 + 1] = Ra.m2;
                rec[j0 * 4 + 2] = __int_as_float(Ra.i1); rec[j0 * 4 + 3] = __int_as_float(Ra.i2);
                rec[j0 * 4 + 4] = Rb.m1; rec[j0 * 4 + 5] = Rb.m2;
                rec[j0 * 4 + 6] = __int_as_float(Rb.i1); rec[j0 * 4 + 7] = __int_as_float(Rb.i2);
            }
        } else {
            job(2 * (wave - DENSE_ARRAYS));   // (wave 7: array 4)
        }
    }
#ifdef DC_STAMPS
    if (PRIOR && lane == 0 && dbg) dbg[wave] = __builtin_amdgcn_s_memrealtime();
#endif
    __syncthreads();
    after();
    // (all twenty record words in ONE round of LDS reads: read where they are used -- inside the
    // branches of dense_pair_max -- each one was waited for on its own)
    float rv[DENSE_ARRAYS * 4];
#pragma unroll
    for (int k = 0; k < DENSE_ARRAYS * 4; ++k) rv[k] = rec[k];
    auto m1 = [&](int k) { return rv[k * 4]; };
    auto m2 = [&](int k) { return rv[k * 4 + 1]; };
    auto i1 = [&](int k) { return __float_as_int(rv[k * 4 + 2]); };
    auto i2 = [&](int k) { return __float_as_int(rv[k * 4 + 3]); };
    int h, a;
    const float mQ = dense_pair_max<float>(m1(0), i1(0), m2(0), i2(0), m1(2), i1(2), m2(2), i2(2), &h, &a);
    // la(h,a) = AA_a * BD_h: x runs over the away side
    const float mR = dense_pair_max<float>(m1(1), i1(1), m2(1), i2(1), m1(2), i1(2), m2(2), i2(2), &a, &h);
    // (with a margin: the prior workgroup's float64 maxima must be below the clip too)
    if (CLIP && (mQ > (float)RATE_CLIP * 0.99999f || mR > (float)RATE_CLIP * 0.99999f)) {
        __syncthreads();  // (the walk writes the records' words)
        return false;
    }
    // the product: candidates by P_h * Q_a, value by the per-pair expression
    dense_pair_max<float>(m1(3), i1(3), m2(3), i2(3), m1(4), i1(4), m2(4), i2(4), &h, &a);
    float mP = 0.f;
    if (h >= 0 && a >= 0) {
        const float2 th = tabH[h], ta = tabA[a];
        mP = (th.x * ta.y) * (ta.x * th.y);
    }
    *oP = mP;
    *oQ = mQ;
    *oR = mR;
    return true;
}

// workgroup maxima of the pair rates from the float32 tables
template <bool CLIP>
__device__ __forceinline__ void pair_maxima_f32(const EvalArgs& A, const float2* tabH,
                                                const float2* tabA, uint32_t pr0, float* redm,
                                                int tid, float* oP, float* oQ, float* oR) {
    const int lane = tid & 63, wave = tid >> 6;
    float mP = 0.f, mQ = 0.f, mR = 0.f;
    auto take = [&](uint32_t pr) {
        const float2 th = tabH[pr & 0xFFFFu], ta = tabA[pr >> 16];
        float lh = th.x * ta.y, la = ta.x * th.y;
        if (CLIP) {
            lh = fminf(lh, (float)RATE_CLIP);
            la = fminf(la, (float)RATE_CLIP);
        }
        mP = fmaxf(mP, lh * la);  // (a NaN rate reaches U through the rate sum anyway)
        mQ = fmaxf(mQ, lh);
        mR = fmaxf(mR, la);
    };
    if (tid < A.P) take(pr0);  // (requested in the prologue)
    // leagues with more pairs than threads: four unconditional loads per round, index clamped (a
    // repeat of the last pair changes no maximum) -- `p == tid ? pr0 : pairs[p]` in a rolled loop was
    // one dependent L2 round trip per pair
    for (int p0 = tid + BLOCK; p0 < A.P; p0 += PAIR_BATCH * BLOCK) {
        uint32_t q[PAIR_BATCH];
#pragma unroll
        for (int u = 0; u < PAIR_BATCH; ++u) q[u] = A.pairs[min(p0 + u * BLOCK, A.P - 1)];
#pragma unroll
        for (int u = 0; u < PAIR_BATCH; ++u) take(q[u]);
    }
    wave_max3_f32(mP, mQ, mR);
    if (lane == 0) {
        redm[wave * 4 + 0] = mP;
        redm[wave * 4 + 1] = mQ;
        redm[wave * 4 + 2] = mR;
    }
    __syncthreads();
#pragma unroll
    for (int wv = 0; wv < WAVES; ++wv) {
        mP = fmaxf(mP, redm[wv * 4 + 0]);
        mQ = fmaxf(mQ, redm[wv * 4 + 1]);
        mR = fmaxf(mR, redm[wv * 4 + 2]);
    }
    *oP = mP;
    *oQ = mQ;
    *oR = mR;
}

// ---- ill-conditioned tau classes in float64 (round 4; see class_terms).  Runs ONLY when the prior part
// raised ZO_ILL (rho within ~1/64 of one of its bounds, relative to the extremal rate:
// |corr_coef_raw| > 4.1 -- never during init_to_uniform(2), rarely on a trajectory); everybody else
// pays a uniform branch on one LDS word.  Not tuned: every thread walks pairs p = tid, tid + BLOCK, ...
// of the pair table, takes the two teams' float32 table entries (the prior part's LDS copies: the bits
// the streaming workgroups built, hence the same "t < TAU_ILL" decisions) and, for the classes they
// left out, adds from the exact float64 rates
//     value   W log(1 + rho c)                      -> red[0]   (-inf when 1 + rho c <= 0: tol = 0 -> red[3])
//     dL/drho W c / t                               -> red[2]   (G_rho: bounds' adjoint, corr_coef_raw)
//     -dL/d eta  = -rho W c / t  per side           -> slots[3T], as flush_run files a lane's sums
// The epilogue books first-order table / rho corrections (eps, ZO_DRHO) on everything that sits in its
// columns; the exact parts have no such error, so what it will book for them goes into red[1] with
// the opposite sign.  slots / red: LDS, zeroed by the caller, a barrier in between; ends with a barrier.
template <bool EXT>
__device__ void ill_core(const EvalArgs& A, const float2* tabH, const float2* tabA, const double* tru,
                         double rho, float rho_f, double* slots, double* red) {
    const int T = A.L.T;
    const int tid = threadIdx.x;
    double val = 0.0, cancel = 0.0, su = 0.0;
    bool neg_inf = false;
    auto eps_of = [&](double tv, float tabv) {   // log(true / table entry), as prior_body files it
        const double r = (tv - (double)tabv) / (double)tabv;
        return fabs(r) < 1e-4 ? r - 0.5 * r * r : 0.0;
    };
#pragma unroll 1
    for (int p = tid; p < A.P; p += BLOCK) {
        const uint32_t pr = A.pairs[p];
        const int h = pr & 0xFFFFu, a = pr >> 16;
        const float2 th = tabH[h], ta = tabA[a];
        float lhf = th.x * ta.y, laf = ta.x * th.y;   // (lane_uniform)
        bool ch = false, ca = false;
        if (EXT) {
            ch = lhf > (float)RATE_CLIP;
            ca = laf > (float)RATE_CLIP;
            lhf = ch ? (float)RATE_CLIP : lhf;
            laf = ca ? (float)RATE_CLIP : laf;
        }
        const unsigned int illm = (class_arg(rho_f, -lhf * laf) >= TAU_ILL ? 0u : 1u) |
                                  (class_arg(rho_f, laf) >= TAU_ILL ? 0u : 2u) |
                                  (class_arg(rho_f, lhf) >= TAU_ILL ? 0u : 4u);
        if (illm == 0u) continue;
        // exact rates (the clip DECISION is the streaming workgroups', as everywhere: the same float32 bits)
        const double lh = ch ? RATE_CLIP : tru[h] * tru[2 * T + a], la = ca ? RATE_CLIP : tru[T + a] * tru[2 * T + h];
        double dh = 0.0, da = 0.0;   // what the lanes would have put into rsh / rsa
#pragma unroll 1
        for (int cls = 0; cls < 3; ++cls) {   // (0,0) | (1,0) | (0,1)
            if (!((illm >> cls) & 1u)) continue;
            const double W = A.pairc[4 * (size_t)p + cls];
            if (!(W > 0.0)) continue;
            const double c = cls == 0 ? -lh * la : (cls == 1 ? la : lh);
            const double t = fma(rho, c, 1.0);
            if (!(t > 0.0)) {   // log(clip(t, 0)) = -inf, derivative 0 (bpl/_util.py:42); NaN propagates
                if (t <= 0.0) neg_inf = true;
                else val += t;
                continue;
            }
            const double u = c * lean::rcp(t);
            val += W * lean::log(t);
            su += W * u;
            if (cls != 1 && !ch) dh -= rho * W * u;   // d/d eta_home: (0,0) and (0,1)
            if (cls != 2 && !ca) da -= rho * W * u;   // d/d eta_away: (0,0) and (1,0)
        }
        if (dh != 0.0) {
            atomicAdd(&slots[h], dh);
            atomicAdd(&slots[2 * T + h], dh);
            atomicAdd(&slots[T + a], dh);
            cancel += dh * (eps_of(tru[h], th.x) + eps_of(tru[2 * T + a], ta.y));
        }
        if (da != 0.0) {
            atomicAdd(&slots[a], da);
            atomicAdd(&slots[T + h], da);
            cancel += da * (eps_of(tru[T + a], ta.x) + eps_of(tru[2 * T + h], th.y));
        }
    }
    if (tid == 0 && !(class_arg(rho_f, -1.0f) >= TAU_ILL) && A.w11 > 0.0) {   // (1,1): the same for every pair
        const double t = 1.0 - rho;
        if (t > 0.0) {
            val += A.w11 * lean::log(t);
            su -= A.w11 * lean::rcp(t);
        } else if (t <= 0.0) {
            neg_inf = true;
        } else {
            val += t;
        }
    }
    if (val != 0.0) atomicAdd(&red[0], val);
    if (cancel != 0.0) atomicAdd(&red[1], cancel);
    if (su != 0.0) atomicAdd(&red[2], su);
    if (neg_inf) red[3] = 1.0;
    __syncthreads();
}
// what the tail does with the result: value and rho-derivative into the record / the SU column
__device__ __forceinline__ void ill_apply(double* zoL, double* col, int ncol, double val, double cancel, double su,
                                          bool neg_inf) {
    const double add = val + cancel - su * zoL[ZO_DRHO];
    zoL[ZO_PAIRC] = neg_inf ? -__builtin_inf() : zoL[ZO_PAIRC] + add;
    col[ncol + 2] += su;
}
// dc_eval's tail workgroup: the prior part's tables are still in LDS (prior_smem), the columns take the
// slot sums directly
template <bool EXT>
__device__ void ill_pass_lds(const EvalArgs& A, double* zoL, double* col, double* scratch, char* prior_smem) {
    const int T = A.L.T;
    const float2* tabH = reinterpret_cast<const float2*>(prior_smem);
    const float2* tabA = tabH + tab_len(T);
    const double* tru = reinterpret_cast<const double*>(tabA + tab_len(T));
    double* red = scratch + 8;
    if (threadIdx.x < 4) red[threadIdx.x] = 0.0;
    __syncthreads();
    ill_core<EXT>(A, tabH, tabA, tru, zoL[ZO_RHO], (float)zoL[ZO_RHOF], col, red);
    if (threadIdx.x == 0) ill_apply(zoL, col, 3 * T, red[0], red[1], red[2], red[3] != 0.0);
    __syncthreads();
}
// dc_vec: the prior workgroups are not the tail (a launch of its own): they file the result behind the
// record -- [zo_ill_off, +3T) slot sums | value | cancel | su | -inf seen -- and the tail adds it
__host__ __device__ inline int zo_ill_off(int D, int T) { return (ZO_HDR + D + 3 * T + 1) & ~1; }
__host__ __device__ inline int zo_ill_len(int T) { return 3 * T + 4; }
__device__ __forceinline__ void ill_add_filed(const EvalArgs& A, double* zoL, double* col) {
    const int T = A.L.T, ncol = 3 * T;
    const double* ill = zoL + zo_ill_off(A.L.D, T);
    for (int i = threadIdx.x; i < ncol; i += BLOCK) col[i] += ill[i];
    __syncthreads();
    if (threadIdx.x == 0) ill_apply(zoL, col, ncol, ill[ncol], ill[ncol + 1], ill[ncol + 2], ill[ncol + 3] != 0.0);
    __syncthreads();
}

// ---------------------------------------------------------------- prior workgroup

// sigmoid site pieces from one exp + one log1p: value (clipped), derivative, and
// log p terms.  sp(x) = softplus(x).
struct SigSite {
    double v, dv, log_v, log_1mv, jac, sig;  // jac = -sp(z) - sp(-z); sig = unclipped
};
__device__ __forceinline__ SigSite sig_site(double zc) {
    // (lean:: -- the short float64 routines: this chain is on the critical path of every leapfrog
    // inside the persistent kernel, and of the extended model's plain evaluation)
    const double az = fabs(zc);
    const double e = lean::exp(-az);
    const double l = lean::log1p_pos(e);            // sp(-|z|)
    const double sp_pos = az + l;                   // sp(|z|)
    const double s_abs = lean::rcp(1.0 + e);        // sigmoid(|z|)
    const double s = zc >= 0 ? s_abs : 1.0 - s_abs;
    SigSite r;
    r.sig = s;
    r.jac = -(sp_pos + l);
    double lv = zc >= 0 ? -l : -sp_pos;       // log sigmoid(z)  = -sp(-z)
    double l1 = zc >= 0 ? -sp_pos : -l;       // log(1-sigmoid(z)) = -sp(z)
    if (s < SIG_LO) {
        r.v = SIG_LO;
        r.dv = 0.0;
        lv = log(SIG_LO);
        l1 = log1p(-SIG_LO);
    } else if (s > SIG_HI) {
        r.v = SIG_HI;
        r.dv = 0.0;
        lv = log(SIG_HI);
        l1 = log1p(-SIG_HI);
    } else {
        r.v = s;
        r.dv = s * (1.0 - s);
    }
    r.log_v = lv;
    r.log_1mv = l1;
    return r;
}

// TO_LDS: the record goes to `zo_lds` (dc_eval: the prior workgroup runs the tail itself and reads
// it from LDS); otherwise to the chain's global hand-off buffer with write-through stores (dc_vec:
// the tail is a separate launch).
// DENSE: the instantiation carries the separable bounds of complete pair tables (leagues of more
// than 64 teams only: the small-league kernels keep their code size)
// Z_LDS: the position is read from `z_lds` (the persistent kernel keeps it in LDS: the leaf wrote it
// there a moment ago, and a load from memory would be a 0.6 us round trip at the head of the step)
// ... and, with it, the static per-team sums (c_lds = cA | cD | cH) and, when they fit, the
// standardised covariates (xs_lds, null otherwise): the tail's LDS copies, filled once per launch --
// per step they were a round of global loads at the head of the critical path
struct NoEntryLoads {
    __device__ __forceinline__ void operator()() const {}
};
// ENTRY_LOADS: the caller's own data-only loads (dc_eval past 64 teams: tail_preload_static), issued BEHIND the
// position's -- vector loads return in order, and in front of them those loads of cold data held the tables back
// (the prior part's first barrier at 1.8 us with 200 teams, the streaming workgroups' at 1.05)
// EARLY: the caller's poll of the accumulator rows (dc_eval past 64 teams: tail_poll_early), called by every wave
// as soon as the bounds' top-two records are out -- the waves that do not go on to the record have nothing left
// to do but the float32 candidates nobody reads from them.  Returns whether it was called (complete pair tables).
template <bool CLIP, bool TO_LDS = false, bool DENSE = false, bool Z_LDS = false, class ENTRY_LOADS = NoEntryLoads,
          class EARLY = NoHook>
__device__ bool prior_body(const EvalArgs& A, int chain, char* smem, double* zo_lds = nullptr,
                           const double* z_lds = nullptr, const double* c_lds = nullptr,
                           const double* xs_lds = nullptr, ENTRY_LOADS entry_loads = ENTRY_LOADS(),
                           EARLY early = EARLY()) {
    const Layout& L = A.L;
    const int T = L.T, K = L.K, D = L.D;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const double* z = Z_LDS ? z_lds : z_of(A, chain);
    double* zo = TO_LDS ? zo_lds : A.hbuf + (size_t)chain * A.hb_stride;
    auto zput = [&](double* p, double v) {
        if (TO_LDS) *p = v;
        else st_sc1(p, v);
    };
    double* gz = zo + ZO_HDR;
    double* eps = gz + D;

    float2* tabH = reinterpret_cast<float2*>(smem);
    float2* tabA = tabH + tab_len(T);
    double* tru = reinterpret_cast<double*>(tabA + tab_len(T));  // EAg[T] | EA[T] | EDn[T]
    double* par = tru + 3 * T;                                    // att[T] | def[T] | ha[T]
    double* sc = par + 3 * T;                                     // [40] scalars
    double* scratch = sc + 40;                                    // [WAVES*8]
    double* amx = scratch + WAVES * 8;                            // [WAVES*8] argmax

    // ---- float32 tables and rho exactly as the streaming workgroups build them
    // This thread's pair(s) and their data-only sums (weights, w x, w y).  Up to BLOCK pairs sit TWO per
    // thread on the first four waves -- consecutive pairs, so lanes and waves stay in ascending pair
    // order (the tie rule below) -- one wave per SIMD: the walk ends in seven wave reductions, VALU
    // bound, and with a pair per thread on eight waves two of them shared every SIMD.
    // (Not the extended model: its walk is the heavier part -- clip branches, the clipped lanes'
    // corrections -- and two of those per lane cost more than the shared SIMDs: 9.0 against 8.8 us.)
    constexpr bool MAYBE_BIG = DENSE || !TO_LDS;
    constexpr bool Z_FIRST = DENSE && !Z_LDS;   // (dc_eval past 64 teams)
    TeamZ tz_first{};
    F32Scalars fs_first{};
    if (Z_FIRST) {
        tz_first = load_team_z<CLIP>(L, z, min(tid, T - 1));
        fs_first = f32_scalars<CLIP>(L, z);
        asm volatile("" ::: "memory");   // (keeps what follows behind them)
    }
    const bool two_each = !CLIP && A.P <= BLOCK;
    const int pfirst = two_each ? 2 * tid : tid;
    uint32_t pr0 = 0, pr1 = 0;
    double pw0[3] = {0.0, 0.0, 0.0}, pw1[3] = {0.0, 0.0, 0.0};
    // (a complete pair table past 64 teams takes the separable bounds: no pair is walked.  Should a rate reach the
    // clip the walk takes its first pair from the table like the others.)
    const bool no_walk = DENSE && A.dense_pairs;
    if (!no_walk && pfirst < A.P) {
        pr0 = A.pairs[pfirst];
        pw0[0] = A.pairw[4 * (size_t)pfirst]; pw0[1] = A.pairw[4 * (size_t)pfirst + 1]; pw0[2] = A.pairw[4 * (size_t)pfirst + 2];
    }
    if (two_each && pfirst + 1 < A.P) {
        pr1 = A.pairs[pfirst + 1];
        pw1[0] = A.pairw[4 * (size_t)pfirst + 4]; pw1[1] = A.pairw[4 * (size_t)pfirst + 5]; pw1[2] = A.pairw[4 * (size_t)pfirst + 6];
    }
    // the team-sum wave (see below) requests its inputs now: its loads queue behind nothing
    // (MAYBE_BIG: instantiations that can see more than 64 teams -- dc_eval's STAGED one and the persistent
    // kernel's cannot, and keep no trace of the other form)
    const bool sums_on_wave = !MAYBE_BIG || T <= 64;
    double pre[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
    if (MAYBE_BIG && !sums_on_wave && tid < T) {
        // larger leagues: the team sums follow the bounds' barrier, one team per thread up to BLOCK teams; their
        // inputs are requested here with everything else (behind that barrier they were a memory round trip
        // in front of the sums; requested just ahead of the bounds, the bounds' loops waited for them)
        pre[0] = A.cA[tid]; pre[1] = A.cD[tid]; pre[2] = A.cH[tid];
        pre[3] = z[CLIP ? L.o_sat + tid : L.o_adec + tid];
        pre[4] = z[CLIP ? L.o_sdt + tid : L.o_ddec + tid];
        pre[5] = CLIP ? z[L.o_hadec + tid] : 0.0;
    }
    if (sums_on_wave && wave == WAVES - 1 && lane < T) {
        if (Z_LDS) { pre[0] = c_lds[lane]; pre[1] = c_lds[T + lane]; pre[2] = c_lds[2 * T + lane]; }
        else { pre[0] = A.cA[lane]; pre[1] = A.cD[lane]; pre[2] = A.cH[lane]; }
        pre[3] = z[CLIP ? L.o_sat + lane : L.o_adec + lane];
        pre[4] = z[CLIP ? L.o_sdt + lane : L.o_ddec + lane];
        pre[5] = CLIP ? z[L.o_hadec + lane] : 0.0;
    }
    entry_loads();
    const F32Scalars fs = Z_FIRST ? fs_first : f32_scalars<CLIP>(L, z);
    if (Z_LDS && xs_lds != nullptr)
        build_tables_f32_x<CLIP>(L, z, XsFromF64{xs_lds, K}, tabH, tabA, tid, load_team_z<CLIP>(L, z, min(tid, T - 1)), fs);
    else
        build_tables_f32<CLIP>(L, z, A.xsf, tabH, tabA, tid, Z_FIRST ? tz_first : load_team_z<CLIP>(L, z, min(tid, T - 1)), fs);

    // DEFER (small leagues: wave 6 holds no pair and wave 7 is the team-sum wave): the scalar priors and
    // the cells' rounding errors -- which nothing in the cells reads -- sit behind the cells' barrier, on
    // those two waves, beside the pair walk.
    const bool defer = sums_on_wave && (two_each || 6 * 64 >= A.P);
    // (Round 4 tried to shorten the two phases in front of the pair walk, twice, and kept neither: (a) no
    // first phase at all, every cell thread working its exp(std) out itself -- three more exp in front of each
    // cell's own cost more than the barrier saved: cells done at 1.94 us instead of 1.68 in a plain launch;
    // (b) the sigmoid sites' chains moved beside the cells -- the first phase is no shorter without them
    // (0.84 against 0.80 us in the persistent kernel: it is argument reloads, the position's LDS reads and the
    // float32 tables), and the cells got 0.2 us slower next to the chains.  A/B on one box, tools/ab_libs.py.)
    // ---- z-only scalars, one transcendental chain per wave, in parallel
    //   0 s_a  1 s_d  2 s_h  3..8 corr site  9..14 u site
    // (past 64 teams the first waves build the float32 tables, one team per thread: the chains go to the last
    // waves -- up to 192 teams those build none)
    const bool chains_last = MAYBE_BIG && !sums_on_wave;
    const int ct0 = chains_last ? 3 * 64 : 0, ct1 = chains_last ? 4 * 64 : 64, ct2 = chains_last ? 5 * 64 : 128,
              ct3 = chains_last ? 6 * 64 : 192, ct4 = chains_last ? 7 * 64 : 256;
    if (tid == ct0) sc[0] = lean::exp(z[L.o_sa]);
    if (tid == ct1) sc[1] = lean::exp(z[L.o_sd]);
    if (tid == ct2) sc[2] = CLIP ? lean::exp(z[L.o_sh]) : 0.0;
    if (tid == ct3) {
        const SigSite s = sig_site(z[L.o_corr]);
        sc[3] = s.v; sc[4] = s.dv; sc[5] = s.log_v; sc[6] = s.log_1mv; sc[7] = s.jac;
        sc[8] = s.sig;
    }
    if (tid == ct4 && CLIP) {
        const SigSite s = sig_site(z[L.o_u]);
        sc[9] = s.v; sc[10] = s.dv; sc[11] = s.log_v; sc[12] = s.log_1mv; sc[13] = s.jac;
        sc[14] = s.sig;
    }
    __syncthreads();
    DC_STAMP(1);
    const double s_a = sc[0], s_d = sc[1], s_h = sc[2];
    const double q = sc[3], dq = sc[4];
    const double m = z[L.o_md];

    // ---- scalar priors + Jacobians (L = log density) and their gradient: one lane of wave 1,
    // while wave 0 builds the cells below (only the team sums v[] are still missing at the end).
    // (DEFER, above: behind the cells' barrier, so that it waits for the cells' exp alone -- it waited
    // 0.56 us for either of the two, the other waves 0.4 us for them)
    // (not deferred: the chain sits on a wave that builds no cells -- wave 6 up to 128 teams, wave 7 up to
    // 149; on wave 1 it ran in FRONT of that wave's cells, 0.5 us of the phase at 100 teams)
    // SLACK (leagues past 64 teams with a complete pair table, round 4): the bounds are ten top-two jobs on the
    // eight waves (below) -- waves 5 and 6 take two float32 arrays each, 1.5 us at 200 teams, everybody else one --
    // and the waves with one job spend their wait on what used to be phases of their own: the scalar priors (this
    // lane's serial chain ran on wave 1 in front of its cells: 0.5 us of that phase) go to wave 7 in front of its
    // job, the team sums (a phase of 0.6 us behind the bounds' barrier) to the waves that own the teams, behind
    // their float64 job, as wave sums the one thread that needs them adds up.
    const bool dense_in = DENSE && A.dense_pairs;
    const bool slack = MAYBE_BIG && !sums_on_wave && dense_in;
    const int lz_tid = slack ? 7 * 64 : defer || 3 * T <= 6 * 64 ? 6 * 64 : (3 * T <= 7 * 64 ? 7 * 64 : 64);
    double Lz = 0.0;
    auto scalar_priors = [&]() {
        const double zsa = z[L.o_sa], zsd = z[L.o_sd];
        Lz = sc[5] + sc[6] + 1.791759469228055 /*log 6*/ + sc[7];  // Beta(2,2) + Jac
        Lz += -0.5 * s_a * s_a - HALF_LOG_2PI + LN2 + zsa;  // HalfNormal(1) + Exp Jacobian
        Lz += -0.5 * s_d * s_d - HALF_LOG_2PI + LN2 + zsd;
        Lz += -0.5 * m * m - HALF_LOG_2PI;
        zput(&gz[L.o_corr], -((1.0 / q - 1.0 / (1.0 - q)) * dq + (1.0 - 2.0 * sc[8])));
        zput(&gz[L.o_md], m);
        zput(&gz[L.o_sa], s_a * s_a - 1.0);
        zput(&gz[L.o_sd], s_d * s_d - 1.0);
        if (!CLIP) {
            const double gam = z[L.o_ha], r = (gam - 0.1) / 0.2;
            Lz += -0.5 * r * r + 1.6094379124341003 /*-log 0.2*/ - HALF_LOG_2PI;
            zput(&gz[L.o_ha], (gam - 0.1) / 0.04);
        } else {
            const double mha = z[L.o_mha], zsh = z[L.o_sh];
            const double r = (mha - 0.1) / 0.2;
            Lz += -0.5 * r * r + 1.6094379124341003 - HALF_LOG_2PI;
            Lz += -0.5 * s_h * s_h - HALF_LOG_2PI + LN2 + zsh;
            Lz += sc[11] + 3.0 * sc[12] + 2.995732273553991 /*log 20*/ + sc[13];
            for (int k = 0; k < K; ++k) {
                const double ba = z[L.o_bA + k], bd = z[L.o_bD + k];
                Lz += -0.5 * ba * ba - 0.5 * bd * bd - 2.0 * HALF_LOG_2PI;
            }
            zput(&gz[L.o_mha], (mha - 0.1) / 0.04);
            zput(&gz[L.o_sh], s_h * s_h - 1.0);
        }
        zput(&zo[ZO_SA], s_a);
        zput(&zo[ZO_SD], s_d);
        zput(&zo[ZO_SH], s_h);
    };
    if (!defer && !slack && tid == lz_tid) scalar_priors();
    // eps = log(true/table) = log1p(r), r = (true - table)/table, |r| ~ 1e-7
    auto put_eps = [&](int j, int t, double tv) {
        const float tabv = j == 0 ? tabH[t].x : (j == 1 ? tabA[t].x : tabH[t].y);
        const double r = (tv - (double)tabv) / (double)tabv;
        double e = r - 0.5 * r * r;
        e = fabs(r) < 1e-4 ? e : 0.0;  // under/overflowed entry: no meaningful correction
        zput(&eps[j * T + t], e);
    };

    // ---- per team: constrained sites, true (float64) tables, rounding errors eps
    for (int i = tid; i < 3 * T; i += BLOCK) {
        const int j = i / T, t = i - j * T;  // j: 0 EAg, 1 EA, 2 EDn
        double att, def, ha;
        if (!CLIP) {
            att = s_a * z[L.o_adec + t];
            def = m + s_d * z[L.o_ddec + t];
            ha = z[L.o_ha];
        } else {
            double apm = 0.0, dpm = m;
            const double* xsp = Z_LDS && xs_lds != nullptr ? xs_lds : A.xs;
            for (int k = 0; k < K; ++k) {
                const double xv = xsp[(size_t)t * K + k];
                apm += xv * z[L.o_bA + k];
                dpm += xv * z[L.o_bD + k];
            }
            att = apm + z[L.o_sat + t] * s_a;
            def = dpm + z[L.o_sdt + t] * s_d;
            ha = z[L.o_mha] + s_h * z[L.o_hadec + t];
        }
        const double arg = j == 0 ? att + ha : (j == 1 ? att : -def);
        const double tv = lean::exp(arg);
        tru[j * T + t] = tv;
        if (!defer || CLIP) put_eps(j, t, tv);   // (extended model: the team-sum wave is the longer pole already)
        if (j == 0) {
            par[t] = att;
            par[T + t] = def;
            par[2 * T + t] = ha;
        }
    }
    __syncthreads();
    DC_STAMP(2);
    if (defer) {
        if (tid == lz_tid) scalar_priors();
        if (!CLIP && wave == WAVES - 1)
            for (int i = lane; i < 3 * T; i += 64) {
                const int j = i / T, t = i - j * T;
                put_eps(j, t, tru[i]);
            }
    }

    // ---- bounds: float32 maxima (-> rho_f32 of the streaming workgroups, same expressions as
    // pair_maxima_f32) and the true float64 maxima with their arg-pairs, in ONE pass over the
    // pairs and ONE barrier.  (Ties: the pair with the smallest index wins; a thread sees
    // its pairs in ascending order and keeps the first, lanes and waves are in ascending
    // pair order too.)
    float* redm = reinterpret_cast<float*>(sc + 16);  // [WAVES*4] (sc[16..32) is free; scratch is reused by block_sum)

    // ---- team sums of the z-only part of L and of its gradient (gz = -dL_prior/dz)
    // v: 0 team part of L (priors + att cA - def cD + ha cH); extended: 1 dL/d rho_p.
    // They need nothing from the bounds: for T <= 64 the last wave (idle in the pair loop up to
    // 448 pairs) computes them with lane = team while the others walk the pairs, instead of the
    // whole workgroup doing so afterwards (1.1 us of the extended model's longest pole).
    double v[2] = {0.0, 0.0};
    // (only the waves that evaluate team terms pay for the log and the division)
    double rp = 0.0, log_vv = 0.0, ivv = 1.0;
    if (CLIP && (!sums_on_wave || wave == WAVES - 1)) {
        rp = 2.0 * sc[9] - 1.0;
        const double vv = 1.0 - rp * rp;
        log_vv = lean::log(vv);
        ivv = lean::rcp(vv);  // one division instead of six per team
    }
    auto team_term = [&](int t, double cAt, double cDt, double cHt, double z0, double z1, double z2) {
        const double att = par[t], def = par[T + t], ha = par[2 * T + t];
        const double lin = att * cAt - def * cDt + ha * cHt;
        if (!CLIP) {
            const double ad = z0, dd = z1;
            v[0] += -0.5 * ad * ad - 0.5 * dd * dd - 2.0 * HALF_LOG_2PI + lin;
            zput(&gz[L.o_adec + t], ad);
            zput(&gz[L.o_ddec + t], dd);
        } else {
            const double sa = z0, sd = z1, hd = z2;
            const double e = sd - rp * sa;
            v[0] += -0.5 * sa * sa - 0.5 * e * e * ivv - 0.5 * log_vv - 0.5 * hd * hd -
                    3.0 * HALF_LOG_2PI + lin;
            v[1] += e * sa * ivv - rp * e * e * (ivv * ivv) + rp * ivv;
            zput(&gz[L.o_sat + t], sa - rp * e * ivv);
            zput(&gz[L.o_sdt + t], e * ivv);
            zput(&gz[L.o_hadec + t], hd);
        }
    };
    if (sums_on_wave && wave == WAVES - 1) {
        if (lane < T) team_term(lane, pre[0], pre[1], pre[2], pre[3], pre[4], pre[5]);
        double s0 = v[0], s1 = v[1];
        wave_sum2_f64(s0, s1);
        if (lane == 0) {
            sc[32] = s0;
            sc[33] = s1;
        }
    }
    float mPf = 0.f, mQf = 0.f, mRf = 0.f;
    double mP = 0.0, mQ = 0.0, mR = 0.0;
    uint32_t aP = 0, aQ = 0, aR = 0;  // arg-pairs (home | away << 16)
    // PER-PAIR VALUE CORRECTIONS (round 3), summed into ZO_PAIRC.  The streaming workgroups take a
    // pair's rate as the float32 product of two float32 table entries: rounded once per pair, the
    // same error for every fixture of the pair (the largest term of |U - U_float64| in round 2,
    // 5e-3 at N = 1e6).  The product of two float32 numbers is exact in float64, so this walk knows
    // that error exactly: + W_p (lh_f32 - th * ta).  And at a clipped rate (extended model) the
    // value needs  k eta -> k log 15:  - (sum_p w k) (eta - log 15) with the exact float64 eta
    // (round 2: v_log_f32 of the raw float32 rate in every lane, 20x the error of a plain point);
    // the clip DECISION is the streaming workgroups' own (the same float32 bits).
    double pairc = 0.0;
    auto take = [&](uint32_t pr, bool valid, double pW, double pSX, double pSY) {
        const int h = pr & 0xFFFFu, a = pr >> 16;
        {
            const float2 th = tabH[h], ta = tabA[a];
            float lhf = th.x * ta.y, laf = ta.x * th.y;
            const bool ch = CLIP && lhf > (float)RATE_CLIP, ca = CLIP && laf > (float)RATE_CLIP;
            if (valid) {
                const double ph = (double)th.x * (double)ta.y, pa = (double)ta.x * (double)th.y;  // exact
                // (DENSE = dc_eval past 64 teams: its lanes take the exact products themselves, lane_uniform<EXACT>)
                if (!DENSE) pairc += pW * ((ch ? 0.0 : (double)lhf - ph) + (ca ? 0.0 : (double)laf - pa));
                // (a clipped lane hands the tail its goal sum as "raw" accumulator -- that cancels the goal
                // count in the gradient -- and the tail's first-order table correction then books
                // - (sum w k) eps for it, which a clipped rate does not have: put it back)
                auto eps_of = [&](double tv, float tabv) {
                    const double r = (tv - (double)tabv) / (double)tabv;
                    return fabs(r) < 1e-4 ? r - 0.5 * r * r : 0.0;
                };
                if (ch)
                    pairc -= pSX * (par[h] + par[2 * T + h] - par[T + a] - LOG_RATE_CLIP -
                                    eps_of(tru[h], th.x) - eps_of(tru[2 * T + a], ta.y));
                if (ca)
                    pairc -= pSY * (par[a] - par[T + h] - LOG_RATE_CLIP - eps_of(tru[T + a], ta.x) -
                                    eps_of(tru[2 * T + h], th.y));
            }
            if (CLIP) {
                lhf = fminf(lhf, (float)RATE_CLIP);
                laf = fminf(laf, (float)RATE_CLIP);
            }
            mPf = fmaxf(mPf, lhf * laf);
            mQf = fmaxf(mQf, lhf);
            mRf = fmaxf(mRf, laf);
        }
        double lh = tru[h] * tru[2 * T + a], la = tru[T + a] * tru[2 * T + h];
        if (CLIP) {
            lh = fmin(lh, RATE_CLIP);
            la = fmin(la, RATE_CLIP);
        }
        // (a clamped repeat of the last pair must not take part in the tie rule)
        if (valid && lh * la > mP) { mP = lh * la; aP = pr; }
        if (valid && lh > mQ) { mQ = lh; aQ = pr; }
        if (valid && la > mR) { mR = la; aR = pr; }
    };
    // complete pair table: the maxima are separable (dense_maxima_f32).  The float32 ones exactly as
    // in the streaming workgroups (same value, same decision); for the float64 ones with their
    // arg-pairs waves 0..4 take one array each as well, and wave 0 evaluates the candidates; the
    // reductions below then see uniform values (zero from the other waves)
    float dPf = 0.f, dQf = 0.f, dRf = 0.f;
    double* rec64 = scratch;  // [5][4] (free until block_sum)
    double* wsum = scratch + 32;   // [WAVES][2] (SLACK: the waves' team sums)
    if (slack && tid == lz_tid) scalar_priors();
    if (dense_in && wave < DENSE_ARRAYS) {
        const int job = __builtin_amdgcn_readfirstlane(wave);
        // 0 AH | 1 AA | 2 BD | 3 AH * BD | 4 AA * BD  (x * 1.0 is exact)
        const double* first = tru + (job == 1 || job == 4 ? T : job == 2 ? 2 * T : 0);
        const Top2<double> R = wave_top2<double>(T, lane, [&](int t) {
            const double x = first[t], bd = tru[2 * T + t];
            return x * (job >= 3 ? bd : 1.0);
        });
        if (lane == 0) {
            rec64[wave * 4 + 0] = R.m1;
            rec64[wave * 4 + 1] = R.m2;
            rec64[wave * 4 + 2] = (double)R.i1;
            rec64[wave * 4 + 3] = (double)R.i2;
        }
    }
    if (slack) {
        if (wave * 64 < T || T > BLOCK) {   // (wave uniform)
            if (tid < T) team_term(tid, pre[0], pre[1], pre[2], pre[3], pre[4], pre[5]);
            for (int t = tid + BLOCK; t < T; t += BLOCK)
                team_term(t, A.cA[t], A.cD[t], A.cH[t], z[CLIP ? L.o_sat + t : L.o_adec + t],
                          z[CLIP ? L.o_sdt + t : L.o_ddec + t], CLIP ? z[L.o_hadec + t] : 0.0);
            double s0 = v[0], s1 = v[1];
            wave_sum2_f64(s0, s1);
            if (lane == 0) {
                wsum[2 * wave] = s0;
                wsum[2 * wave + 1] = s1;
            }
        } else if (lane == 0) {
            wsum[2 * wave] = 0.0;
            wsum[2 * wave + 1] = 0.0;
        }
    }
#ifdef DC_STAMPS
    unsigned long long* dense_dbg = A.debug && blockIdx.y == 0 ? A.debug + (size_t)gridDim.x * 16 : nullptr;
#else
    unsigned long long* dense_dbg = nullptr;
#endif
    const bool dense = dense_in && dense_maxima_f32<CLIP, true, EARLY>(T, tabH, tabA, redm, tid, &dPf, &dQf, &dRf, dense_dbg, early);  // (barrier inside)
    if (dense) {
        // (the values are wave-uniform already: no reductions, nothing written over the records; wave 0
        // files the float64 maxima and their arg-pairs for the combine below, zeros for the other waves)
        if (wave == 0) {
            double rv[DENSE_ARRAYS * 4];   // (one round of LDS reads, see dense_maxima_f32)
#pragma unroll
            for (int k = 0; k < DENSE_ARRAYS * 4; ++k) rv[k] = rec64[k];
            auto m1 = [&](int k) { return rv[k * 4]; };
            auto m2 = [&](int k) { return rv[k * 4 + 1]; };
            auto i1 = [&](int k) { return (int)rv[k * 4 + 2]; };
            auto i2 = [&](int k) { return (int)rv[k * 4 + 3]; };
            int h, a;
            mQ = dense_pair_max<double>(m1(0), i1(0), m2(0), i2(0), m1(2), i1(2), m2(2), i2(2), &h, &a);
            aQ = (uint32_t)h | ((uint32_t)a << 16);
            mR = dense_pair_max<double>(m1(1), i1(1), m2(1), i2(1), m1(2), i1(2), m2(2), i2(2), &a, &h);
            aR = (uint32_t)h | ((uint32_t)a << 16);
            dense_pair_max<double>(m1(3), i1(3), m2(3), i2(3), m1(4), i1(4), m2(4), i2(4), &h, &a);
            if (h >= 0 && a >= 0) {
                double lh = tru[h] * tru[2 * T + a], la = tru[T + a] * tru[2 * T + h];
                if (CLIP) {
                    lh = fmin(lh, RATE_CLIP);
                    la = fmin(la, RATE_CLIP);
                }
                mP = lh * la;
                aP = (uint32_t)h | ((uint32_t)a << 16);
            }
            // (wave uniform, and this wave is the one that combines: the maxima stay in its registers -- through
            // the waves' records and the eight-lane combine below they were 0.4 us of the record's serial chain)
        }
    } else {
        if (no_walk && pfirst < A.P) {   // (not requested at entry: see no_walk)
            pr0 = A.pairs[pfirst];
            pw0[0] = A.pairw[4 * (size_t)pfirst]; pw0[1] = A.pairw[4 * (size_t)pfirst + 1]; pw0[2] = A.pairw[4 * (size_t)pfirst + 2];
        }
        if (pfirst < A.P) take(pr0, true, pw0[0], pw0[1], pw0[2]);
        if (two_each && pfirst + 1 < A.P) take(pr1, true, pw1[0], pw1[1], pw1[2]);
        for (int p0 = tid + BLOCK; !two_each && p0 < A.P; p0 += PAIR_BATCH * BLOCK) {  // (see pair_maxima_f32)
            uint32_t q[PAIR_BATCH];
            double qw[PAIR_BATCH][3];
#pragma unroll
            for (int u = 0; u < PAIR_BATCH; ++u) {
                const size_t pi = (size_t)min(p0 + u * BLOCK, A.P - 1);
                q[u] = A.pairs[pi];
                qw[u][0] = A.pairw[4 * pi]; qw[u][1] = A.pairw[4 * pi + 1]; qw[u][2] = A.pairw[4 * pi + 2];
            }
#pragma unroll
            for (int u = 0; u < PAIR_BATCH; ++u) take(q[u], p0 + u * BLOCK < A.P, qw[u][0], qw[u][1], qw[u][2]);
        }
    }
    if (!dense) {
        // (a wave whose lanes hold no pair -- the team-sum wave among them -- files zeros: it was the
        // last one at the barrier below, with seven reductions of nothing behind its own work)
        const bool has_pairs = wave * (two_each ? 128 : 64) < A.P;
        double wP = mP, wQ = mQ, wR = mR, wC = pairc;
        uint32_t pP = 0, pQ = 0, pR = 0;
        if (has_pairs) {
            wave_bounds_reduce(mPf, mQf, mRf, wP, wQ, wR, wC);
            // first lane holding the wave maximum (no cross-lane min-reduction, no reload)
            const unsigned long long bP = __ballot(mP == wP), bQ = __ballot(mQ == wQ), bR = __ballot(mR == wR);
            pP = (uint32_t)__builtin_amdgcn_readlane((int)aP, bP ? __ffsll((long long)bP) - 1 : 0);
            pQ = (uint32_t)__builtin_amdgcn_readlane((int)aQ, bQ ? __ffsll((long long)bQ) - 1 : 0);
            pR = (uint32_t)__builtin_amdgcn_readlane((int)aR, bR ? __ffsll((long long)bR) - 1 : 0);
        }
        if (lane == 0) {
            redm[wave * 4 + 0] = has_pairs ? mPf : 0.f;
            redm[wave * 4 + 1] = has_pairs ? mQf : 0.f;
            redm[wave * 4 + 2] = has_pairs ? mRf : 0.f;
            amx[wave * 8 + 0] = has_pairs ? wP : 0.0; amx[wave * 8 + 1] = has_pairs ? wQ : 0.0;
            amx[wave * 8 + 2] = has_pairs ? wR : 0.0;
            amx[wave * 8 + 3] = (double)pP; amx[wave * 8 + 4] = (double)pQ;
            amx[wave * 8 + 5] = (double)pR;
            amx[wave * 8 + 6] = has_pairs ? wC : 0.0;
        }
    }
    DC_STAMP_PW(8);
    // (separable bounds: nothing crosses waves here any more -- wave 0 keeps its maxima, the team sums were filed
    // in front of dense_maxima_f32's barrier -- and the other waves go on to the rows' early poll)
    if (!(slack && dense)) __syncthreads();
    DC_STAMP(3);

    // ---- team sums: see above (one wave, beside the pair loop) or, for T > 64, here
    if (slack) {
        if (tid == lz_tid) {   // (the only reader)
            v[0] = v[1] = 0.0;
#pragma unroll
            for (int wv = 0; wv < WAVES; ++wv) {
                v[0] += wsum[2 * wv];
                v[1] += wsum[2 * wv + 1];
            }
        }
    } else if (!sums_on_wave) {
        if (tid < T) team_term(tid, pre[0], pre[1], pre[2], pre[3], pre[4], pre[5]);
        for (int t = tid + BLOCK; t < T; t += BLOCK)
            team_term(t, A.cA[t], A.cD[t], A.cH[t], z[CLIP ? L.o_sat + t : L.o_adec + t],
                      z[CLIP ? L.o_sdt + t : L.o_ddec + t], CLIP ? z[L.o_hadec + t] : 0.0);
        block_sum<2>(v, scratch, tid);
    } else {
        v[0] = sc[32];
        v[1] = sc[33];
    }
    DC_STAMP(12);

    if (tid < 2 * K) {  // covariate coefficients ~ N(0,1)
        const int o = (tid >= K ? L.o_bD + tid - K : L.o_bA + tid);
        zput(&gz[o], z[o]);
    }
    // The two serial pieces of the record run on different waves at the same time (a single lane
    // executing ~300 dependent float64 instructions is the longest pole of this workgroup).
    // ---- bounds: combine the waves' arg-maxima on lanes 0..WAVES-1 of wave 0 (a serial loop
    // of 48 dependent LDS reads cost 1.1 us here); ties: the lowest wave wins
    double M = 0.0, Lh = 0.0, La = 0.0, pairc_all = 0.0;
    float cPf = 0.f, cQf = 0.f, cRf = 0.f;   // the workgroup's float32 maxima (walk: from the waves' records)
    uint32_t pP = 0, pQ = 0, pR = 0;
    if (wave == 0 && dense) {
        M = mP; Lh = mQ; La = mR;
        pP = aP; pQ = aQ; pR = aR;
    } else if (wave == 0) {
        // (unconditional reads of a clamped record, then selects: behind `src ? ... : 0` every read sat in
        // its own exec-masked block with its own wait)
        const bool src = lane < WAVES;
        const double* rec = amx + (src ? lane : 0) * 8;
        const double r0 = rec[0], r1 = rec[1], r2 = rec[2], r3 = rec[3], r4 = rec[4], r5 = rec[5], r6 = rec[6];
        const double a0 = src ? r0 : 0.0, a1 = src ? r1 : 0.0, a2 = src ? r2 : 0.0;
        pairc_all = src ? r6 : 0.0;
        const uint32_t q0 = src ? (uint32_t)r3 : 0u, q1 = src ? (uint32_t)r4 : 0u, q2 = src ? (uint32_t)r5 : 0u;
        M = a0; Lh = a1; La = a2;
        static_assert(WAVES == 8, "lanes8_max3_sum: one record per wave in lanes 0..7");
        if (!dense) {
            const float* rf = redm + (src ? lane : 0) * 4;
            const float f0 = rf[0], f1 = rf[1], f2 = rf[2];
            cPf = src ? f0 : 0.f; cQf = src ? f1 : 0.f; cRf = src ? f2 : 0.f;
        }
        lanes8_max3_sum(M, Lh, La, pairc_all, cPf, cQf, cRf);
        const unsigned long long bP = __ballot(src && a0 == M), bQ = __ballot(src && a1 == Lh),
                                 bR = __ballot(src && a2 == La);
        pP = (uint32_t)__builtin_amdgcn_readlane((int)q0, bP ? __ffsll((long long)bP) - 1 : 0);
        pQ = (uint32_t)__builtin_amdgcn_readlane((int)q1, bQ ? __ffsll((long long)bQ) - 1 : 0);
        pR = (uint32_t)__builtin_amdgcn_readlane((int)q2, bR ? __ffsll((long long)bR) - 1 : 0);
    }
    double ill_flag = 0.0, rho_keep = 0.0;
    float rhof_keep = 0.f;
    if (tid == 0) {
        unsigned int flags = 0;
        if (CLIP) {
            if (tru[pP & 0xFFFFu] * tru[2 * T + (pP >> 16)] > RATE_CLIP) flags |= 1u;
            if (tru[T + (pP >> 16)] * tru[2 * T + (pP & 0xFFFFu)] > RATE_CLIP) flags |= 2u;
            if (tru[pQ & 0xFFFFu] * tru[2 * T + (pQ >> 16)] > RATE_CLIP) flags |= 4u;
            if (tru[T + (pR >> 16)] * tru[2 * T + (pR & 0xFFFFu)] > RATE_CLIP) flags |= 8u;
        }
        DC_STAMP(13);
        // (lean::rcp: estimate + two Newton steps, a third of a float64 division's dependent chain; the
        // division itself where the operand is not an ordinary number)
        auto inv = [](double x) { return x > 1e-300 && x < 1e300 ? lean::rcp(x) : 1.0 / x; };
        const double UB = M > 1.0 ? inv(M) : 1.0;
        const double LB = -inv(fmax(Lh, La));
        const double rho = LB + q * (UB - LB);
        // the float32 rho every streaming workgroup computed (bit for bit)
        float fP = dPf, fQ = dQf, fR = dRf;   // (separable bounds: already the workgroup's maxima)
        if (!dense) { fP = cPf; fQ = cQf; fR = cRf; }
        const float rho_f = rho_f32(fP, fQ, fR, fs.q);
        DC_STAMP(15);
        zput(&zo[ZO_Q], q);
        zput(&zo[ZO_DQ], dq);
        zput(&zo[ZO_UB], UB);
        zput(&zo[ZO_LB], LB);
        zput(&zo[ZO_RHO], rho);
        zput(&zo[ZO_DRHO], rho - (double)rho_f);
        zput(&zo[ZO_RHOF], (double)rho_f);
        {   // the smallest float32 tau argument of each class over ALL pairs (rounding is monotone: it sits
            // at the class's extremal c, i.e. at the float32 maxima): any below TAU_ILL?
            const bool fine = class_arg(rho_f, -fP) >= TAU_ILL && class_arg(rho_f, fQ) >= TAU_ILL &&
                              class_arg(rho_f, fR) >= TAU_ILL && class_arg(rho_f, -1.0f) >= TAU_ILL;
            zput(&zo[ZO_ILL], fine ? 0.0 : 1.0);
            ill_flag = fine ? 0.0 : 1.0;
            rho_keep = rho;
            rhof_keep = rho_f;
        }
        zput(&zo[ZO_M], M);
        zput(&zo[ZO_LH], Lh);
        zput(&zo[ZO_LA], La);
        zput(&zo[ZO_PP], (double)pP);
        zput(&zo[ZO_PQ], (double)pQ);
        zput(&zo[ZO_PR], (double)pR);
        zput(&zo[ZO_FLAGS], (double)flags);
        zput(&zo[ZO_PAIRC], pairc_all);
    }
    if (tid == lz_tid) {  // what was waiting for the team sums
        if (CLIP) {
            const double u = sc[9], du = sc[10];
            zput(&gz[L.o_u], -(2.0 * v[1] * du + (1.0 / u - 3.0 / (1.0 - u)) * du +
                                 (1.0 - 2.0 * sc[14])));
        }
        zput(&zo[ZO_LZ], Lz + v[0]);
    }
    if (!TO_LDS) {
        // dc_vec: this workgroup is not the tail (a launch of its own, without these tables): the
        // ill-conditioned tau classes (class_terms, ill_core) are worked out here and filed behind the
        // record.  One more barrier on a workgroup that is not the launch's critical path.
        if (tid == 0) {
            sc[34] = ill_flag;
            sc[35] = rho_keep;
            sc[36] = (double)rhof_keep;
        }
        __syncthreads();
        if (sc[34] != 0.0) {   // (uniform)
            double* slots = par;          // att | def | ha: nothing reads them any more
            double* red = scratch;        // (free: block_sum is done)
            for (int i = tid; i < 3 * T; i += BLOCK) slots[i] = 0.0;
            if (tid < 4) red[tid] = 0.0;
            __syncthreads();
            ill_core<CLIP>(A, tabH, tabA, tru, sc[35], (float)sc[36], slots, red);
            double* ill = zo + zo_ill_off(D, T);
            for (int i = tid; i < 3 * T; i += BLOCK) zput(&ill[i], slots[i]);
            if (tid < 4) zput(&ill[3 * T + tid], red[tid]);
        }
    }
    return dense_in;
}

// ---- the next position as data-tagged granules (persistent evaluation kernel).  One naturally
// aligned 8-byte {float32 value, tag} per latent entry, written by ONE write-through store: the
// reader needs no flag and no ordering (MI355X_MICROARCH.md, handoff-1to1).  float32 is what the
// streaming workgroups build their tables from anyway; the tail workgroup keeps the float64 state.
// (the "finished" tag is launch specific too -- tag_base + persist_steps + 1 -- or the FIN
// granules of one launch would end the next one before its tail had published anything)
__device__ __forceinline__ void st_granule(unsigned long long* g, float v, unsigned int tag) {
    __hip_atomic_store(g, ((unsigned long long)tag << 32) | (unsigned long long)__float_as_uint(v),
                       __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ unsigned long long ld_granule(const unsigned long long* g) {
    return __hip_atomic_load(g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// one wave: z (this workgroup's own float64 stores, plain loads) -> granules
__device__ __forceinline__ void publish_z(unsigned long long* zg, const double* z, int D, int lane,
                                          unsigned int tag) {
    for (int i = lane; i < D; i += 64) st_granule(&zg[i], (float)z[i], tag);
}
__device__ __forceinline__ void publish_fin(unsigned long long* zg, int D, int lane, unsigned int fin_tag) {
    for (int i = lane; i < D; i += 64) st_granule(&zg[i], 0.f, fin_tag);
}

// persistent chains: a subtree is complete (the doubling's last leaf, a U-turn, a divergence): the leaf
// wave combines the trees, adapts / stores the draw when the transition ends, and starts the next
// doubling or transition -- from a position only memory holds.  Called by the whole workgroup behind a
// barrier that follows the leaf's two halves; `pub_tag` != 0: publish that position (or FIN) for the
// streaming workgroups.  *fin_flag = 1 when the chain has finished.
__device__ __forceinline__ void subtree_complete(const EvalArgs& A, int chain, double* fin_flag, double* zn_lds,
                                                 unsigned int pub_tag) {
    const int D = A.L.D;
    const int t = threadIdx.x & 63, wave = threadIdx.x >> 6;
    double* ns = nuts_of(A, chain);
    if (wave == RNG_WAVE) nd::wave_mem_sync();   // (the other half's stores, before the leaf wave reads them)
    __syncthreads();
    if (wave == LEAF_WAVE) {
        __builtin_amdgcn_s_dcache_inv();   // (the advance reads chain state with uniform addresses)
        nd::persist_advance(ns, *A.persist, chain, t);
        if (zn_lds)   // (rare: the new doubling / transition starts from a position only memory holds)
            for (int i = t; i < D; i += 64) zn_lds[i] = nd::vec(ns, D, nd::V_ZN)[i];
        if (pub_tag != 0u) {  // a new doubling / transition starts somewhere else -- or nowhere
            const bool fin = (ns + A.persist->pd_off)[nd::P_ALLDONE] != 0.0;
            if (fin) {
                publish_fin(A.zg + (size_t)chain * D, D, t, A.tag_base + 1u + (unsigned int)A.persist_steps);
                if (t == 0) *fin_flag = 1.0;
            } else {
                publish_z(A.zg + (size_t)chain * D, nd::vec(ns, D, nd::V_ZN), D, t, pub_tag);
            }
        }
    }
}

// Per-team epilogue for T <= 64: lane t owns team t, sums are DPP wave reductions, adds and
// FMAs only.  A single wave issues a dependent instruction every ~5 cycles, so the work is
// split by OUTPUT GROUP over four waves that never need each other's results:
//   wave 0  attack sites      (+ attack coefficients)       wave 2  home-advantage sites, corr, u
//   wave 1  defence sites     (+ defence coefficients)      wave 3  the potential (value corrections)
// Same arithmetic and the same summation order on every run (deterministic).  With NUTS the
// workgroup then barriers and wave LEAF_WAVE (one of the four idle ones, so its header wait,
// index arithmetic and threefry draws cost nothing) books the leaf.
// LNE: vector elements per lane the leaf keeps in REGISTERS (D <= 64 LNE; deduced from leaf1).  1
// everywhere but in the persistent kernel, which takes 2 for 64 < D <= 128 (the extended model with
// 20 teams: D = 67 .. 77) instead of staging the leaf's vectors in LDS -- a single resident chain
// has the registers, and the staged leaf was 1.0 us slower per leapfrog.
// LOOP (the persistent kernel): outputs only -- its loop books the leaf itself (leaf_window_*, dc_eval_loop)
template <bool NUTS, bool EXT, int LNE, bool LOOP = false>
__device__ void tail_waves(const EvalArgs& A, int chain, const double* zoL, const double* cL,
                           const double* zL, const double* col, const double* xsL,
                           double* gradL, const nd::LeafState<LNE>& leaf1, double* stg,
                           unsigned int pub_tag = 0u /* tag of the NEXT step */,
                           double* zn_lds = nullptr /* the LDS copy of the position */) {
    const Layout& L = A.L;
    const int T = L.T, K = L.K, D = L.D;
    const int t = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const bool on = t < T;
    double* grad = grad_of(A, chain);
    constexpr bool nuts = NUTS;
    auto put = [&](int o, double v) {
        if (!LOOP) grad[o] = v;   // (the persistent kernel's leaf reads the LDS copy; nobody reads V_GRAD there)
        if (nuts) gradL[o] = v;
    };
    const double* gz = zoL + ZO_HDR;
    const double* eps = gz + D;
    const int ncol = 3 * T;
    const double q = zoL[ZO_Q], UB = zoL[ZO_UB], LB = zoL[ZO_LB];
    const double G_rho = col[ncol + 2];  // SU
    // adjoint of the bounds (Appendix A.3): what the arg-extremal pairs add to team t's
    // d/d attack (ja), d/d defence (jd), d/d home advantage (jh)
    double ja = 0.0, jd = 0.0, jh = 0.0;
    // (waves COV_WAVE_A / COV_WAVE_D, otherwise idle, take the covariate coefficients: they
    // rebuild the attack / defence adjoint themselves and run beside waves 0..3)
    const bool cov_a = K > 0 && wave == COV_WAVE_A, cov_d = K > 0 && wave == COV_WAVE_D;
    {   // (every record word read up front, then selects: with the reads inside the branches each one was
        // waited for on its own -- 0.5 us at the head of every epilogue wave)
        const double M = zoL[ZO_M], Lh = zoL[ZO_LH], La = zoL[ZO_LA];
        const double fP = zoL[ZO_PP], fQ = zoL[ZO_PQ], fR = zoL[ZO_PR], fF = zoL[ZO_FLAGS];
        const bool use = A.P > 0 && (wave < 3 || cov_a || cov_d);
        const uint32_t pP = (uint32_t)fP, pQ = (uint32_t)fQ, pR = (uint32_t)fR;
        const unsigned int flags = (unsigned int)fF;
        {
            const double v = use && M > 1.0 ? G_rho * q * (-UB) : 0.0;
            const int h = pP & 0xFFFFu, a = pP >> 16;
            const double v1 = (flags & 1u) ? 0.0 : v, v2 = (flags & 2u) ? 0.0 : v;
            ja += (t == h ? v1 : 0.0) + (t == a ? v2 : 0.0);
            jh += t == h ? v1 : 0.0;
            jd -= (t == a ? v1 : 0.0) + (t == h ? v2 : 0.0);
        }
        {
            const double v = use ? G_rho * (1.0 - q) * (-LB) : 0.0;
            const bool lb_home = Lh >= La;
            const uint32_t pL = lb_home ? pQ : pR;
            const int h = pL & 0xFFFFu, a = pL >> 16;
            const double vv = (flags & (lb_home ? 4u : 8u)) ? 0.0 : v;
            // home rate: attack + home advantage of h, defence of a; away rate: attack of a, defence of h
            ja += lb_home ? (t == h ? vv : 0.0) : (t == a ? vv : 0.0);
            jh += lb_home && t == h ? vv : 0.0;
            jd -= lb_home ? (t == a ? vv : 0.0) : (t == h ? vv : 0.0);
        }
    }
    const bool basic = !EXT;
    // covariate coefficients: sum_t Xs[t,k] g_t for every k, four sums in flight; lane j of a
    // batch stores its coefficient
    auto cov_grad = [&](int o, double gt) {
        if (K <= 8 && xsL != nullptr) {   // (wave uniform)
            // up to eight coefficients in ONE pass: lane 8 k + j takes coefficient k over teams j, j + 8, ...
            // (g_t from lane t by a wave shuffle, the covariates from the LDS copy), then three DPP steps fold
            // the eight lanes of a coefficient.  Four wave-wide float64 sums per round of four coefficients were
            // 0.6 us of the extended model's 8.2 with five covariates (a what-if build without them: 7.6).
            const int k = t >> 3, j = t & 7;
            double sk = 0.0;
            for (int tb = 0; tb < T; tb += 8) {   // (wave-uniform trip count)
                const int tt = tb + j;
                const double g = __shfl(gt, tt < 64 ? tt : 63);
                const double xv = k < K && tt < T ? xsL[(size_t)tt * K + k] : 0.0;
                sk += xv * g;
            }
            sk += dpp_f64<0x141>(0.0, sk);   // row_half_mirror: lane j <-> 7 - j
            sk += dpp_f64<0xB1>(0.0, sk);    // quad_perm [1,0,3,2]
            sk += dpp_f64<0x4E>(0.0, sk);    // quad_perm [2,3,0,1]
            if (j == 0 && k < K) put(o + k, gz[o + k] - sk);
            return;
        }
        for (int k0 = 0; k0 < K; k0 += 4) {
            double v[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int k = k0 + j;
                const double xv =
                    on && k < K ? (xsL ? xsL[(size_t)t * K + k] : A.xs[(size_t)t * K + k]) : 0.0;
                v[j] = xv * gt;
            }
            wave_sum4_f64(v);
            const double mine = t == 0 ? v[0] : t == 1 ? v[1] : t == 2 ? v[2] : v[3];
            if (t < 4 && k0 + t < K) put(o + k0 + t, gz[o + k0 + t] - mine);
        }
    };
    if (wave == 0) {  // ---- attack
        const double s_a = zoL[ZO_SA];
        const double ga = on ? cL[t] - col[t] + ja : 0.0;
        const int o = basic ? L.o_adec : L.o_sat;
        const double dec = on ? zL[o + t] : 0.0;
        if (on) put(o + t, gz[o + t] - s_a * ga);
        const double dot_a = wave_sum_f64(dec * ga);
        if (t == 0) put(L.o_sa, gz[L.o_sa] - s_a * dot_a);
    } else if (cov_a) {  // ---- attack coefficients, d/d beta_k: sum_t Xs[t,k] g_t
        cov_grad(L.o_bA, on ? cL[t] - col[t] + ja : 0.0);
    } else if (cov_d) {  // ---- defence coefficients
        cov_grad(L.o_bD, on ? -(cL[T + t] - col[T + t]) + jd : 0.0);
    } else if (wave == 1) {  // ---- defence
        const double s_d = zoL[ZO_SD];
        const double gd = on ? -(cL[T + t] - col[T + t]) + jd : 0.0;
        const int o = basic ? L.o_ddec : L.o_sdt;
        const double dec = on ? zL[o + t] : 0.0;
        if (on) put(o + t, gz[o + t] - s_d * gd);
        double sum_gd = gd, dot_d = dec * gd;
        wave_sum2_f64(sum_gd, dot_d);
        if (t == 0) {
            put(L.o_md, gz[L.o_md] - sum_gd);
            put(L.o_sd, gz[L.o_sd] - s_d * dot_d);
        }
    } else if (wave == 2) {  // ---- home advantage, corr_coef_raw, u
        const double gh = on ? cL[2 * T + t] - col[2 * T + t] + jh : 0.0;
        const double sum_gh = wave_sum_f64(gh);
        if (basic) {
            if (t == 0) put(L.o_ha, gz[L.o_ha] - sum_gh);
        } else {
            const double s_h = zoL[ZO_SH];
            const double hd = on ? zL[L.o_hadec + t] : 0.0;
            if (on) put(L.o_hadec + t, gz[L.o_hadec + t] - s_h * gh);
            const double dot_h = wave_sum_f64(hd * gh);
            if (t == 0) {
                put(L.o_mha, gz[L.o_mha] - sum_gh);
                put(L.o_sh, gz[L.o_sh] - s_h * dot_h);
                put(L.o_u, gz[L.o_u]);
            }
        }
        if (t == 0) put(L.o_corr, gz[L.o_corr] - G_rho * (UB - LB) * zoL[ZO_DQ]);
    } else if (wave == 3) {  // ---- value: first-order corrections for the float32 table rounding
        const double ra = on ? col[t] : 0.0, rd = on ? col[T + t] : 0.0, rh = on ? col[2 * T + t] : 0.0;
        double corr = on ? -(rh * eps[t] + (ra - rh) * eps[T + t] + rd * eps[2 * T + t]) : 0.0;
        corr = wave_sum_f64(corr);
        const double Ltot = zoL[ZO_LZ] + corr + zoL[ZO_PAIRC] + G_rho * zoL[ZO_DRHO] - col[ncol + 0] - A.lgsum +
                            LN2 * col[ncol + 1] - col[ncol + 3];
        if (t == 0) {
            *pot_of(A, chain) = -Ltot;
            if (A.aux != nullptr) {
                double* aux = aux_of(A, chain);
                aux[0] = zoL[ZO_RHO];
                aux[1] = LB;
                aux[2] = UB;
                aux[3] = q;
            }
            if (nuts) {
                gradL[D] = -Ltot;
                gradL[D + 1] = zoL[ZO_RHO];
                gradL[D + 2] = LB;
                gradL[D + 3] = UB;
                gradL[D + 4] = q;
            }
        }
    }
    DC_STAMP(14);
    if (nuts && !LOOP) {  // device-resident NUTS: the leaf wave finishes the leapfrog and books the leaf
        const bool small = D <= 64 * LNE;  // LNE vector elements per lane: registers; else LDS (stg)
        nd::LeafState<LNE> lf1 = leaf1;
        double* ns = nuts_of(A, chain);
        // preparation that needs the header only, while waves 0..3 write the outputs
        if (wave == LEAF_WAVE && small) nd::leaf_prepare<false>(lf1);
        if (wave == RNG_WAVE) nd::leaf_rng(lf1.hv, &lf1.nhi, &lf1.nlo, &lf1.u_take);
        DC_STAMP_LEAF(13);
        __syncthreads();
        DC_STAMP_LEAF(12);
        // the two halves of the leaf (nuts_dev.hip.h), one wave each; the other waves only join
        // the barriers (every wave reaches the end of this function: the persistent evaluation
        // kernel goes on to the next leapfrog from here)
        if (wave == LEAF_WAVE) {
            // (zn_lds: the next position also goes to the tail's LDS copy -- every reader of this
            // step's copy is behind the barrier above)
            const bool sub_done =
                small ? nd::leaf_moves(ns, D, A.nuts_max_depth, t, gradL, lf1, zn_lds)
                      : nd::leaf_moves_staged(ns, D, A.nuts_max_depth, t, gradL, stg, lf1.hv, zn_lds);
            DC_STAMP_LEAF(11);
            if (t == 0) {
                gradL[D + 5] = sub_done ? 1.0 : 0.0;
                gradL[D + 6] = 0.0;  // (chain finished: set below)
            }
            // persistent kernel: the next position goes out NOW (the subtree goes on: V_ZN is final;
            // this wave stored it, so its own loads see it) -- the streaming workgroups start the
            // next leapfrog while the other half of the leaf is still being booked
            if (pub_tag != 0u && !sub_done)
                publish_z(A.zg + (size_t)chain * D, zn_lds ? zn_lds : nd::vec(ns, D, nd::V_ZN), D, t, pub_tag);
        } else if (wave == RNG_WAVE) {
            if (small) nd::leaf_weights(ns, D, t, gradL, lf1);
            else nd::leaf_weights_staged(ns, D, t, gradL, stg, lf1.hv, lf1.nhi, lf1.nlo, lf1.u_take);
            DC_STAMP_WAVE(RNG_WAVE, 15);
        }
        if (A.persist == nullptr) return;
        // persistent chains: a finished subtree (rare) is combined by the leaf wave, which must
        // then see the other half's stores -- release + barrier, acquire in persist_advance
        __syncthreads();
        if (gradL[D + 5] == 0.0) return;  // (uniform)
        subtree_complete(A, chain, gradL + D + 6, zn_lds, pub_tag);
    }
}

// General epilogue (any number of teams): value corrections, adjoint of the bounds and the chain
// rule over LDS-resident sums, whole workgroup.  zoL / cL / zL / col as staged by the caller.
template <bool EXT>
__device__ void tail_general(const EvalArgs& A, int chain, const double* zoL, const double* cL,
                             const double* zL, double* col, double* scratch) {
    const Layout& L = A.L;
    const int T = L.T, K = L.K, D = L.D;
    const int tid = threadIdx.x;
    double* grad = grad_of(A, chain);
    const int ncol = 3 * T;
    const double* gz = zoL + ZO_HDR;
    const double* eps = gz + D;
    const double s_a = zoL[ZO_SA], s_d = zoL[ZO_SD], s_h = zoL[ZO_SH];
    const double q = zoL[ZO_Q], dq = zoL[ZO_DQ], UB = zoL[ZO_UB], LB = zoL[ZO_LB];
    const double rho = zoL[ZO_RHO], drho = zoL[ZO_DRHO];
    const double M = zoL[ZO_M], Lh = zoL[ZO_LH], La = zoL[ZO_LA], Lz = zoL[ZO_LZ];
    double* g_att = col;          // raw accumulators, then dL/d attack_t
    double* g_def = col + T;      // dL/d defence_t
    double* g_ha = col + 2 * T;   // dL/d home_adv_t
    const double SLAM = col[ncol + 0], SLOG = col[ncol + 1], SU = col[ncol + 2],
                 CLIPC = col[ncol + 3];
    const double G_rho = SU;  // sum_i w_i dlogtau_i/drho

    // ---- 3. per team, in ONE pass and in registers (round 4; three passes over LDS with a barrier each and the
    // bounds' adjoint on a single thread between them were 2.6 us at 200 teams):
    //   first-order value corrections for the float32 rounding of the tables
    //     dL = - sum_t [ ha_raw eAg_t + (att_raw - ha_raw) eA_t + def_raw eDn_t ]
    //   (and of rho:  dL = G_rho (rho_true - rho_f32), below);  raw sums -> dL/d(team sites);
    //   the adjoint of the bounds (Appendix A.3) where the team is one of the (at most four) arg-extremal ones
    //   -- every thread tests its own team, in the order the single thread added them;
    //   the chain rule to z (adds and FMAs only); grad = gz - (fixture-sum terms)
    const uint32_t pP = (uint32_t)zoL[ZO_PP], pQ = (uint32_t)zoL[ZO_PQ], pR = (uint32_t)zoL[ZO_PR];
    const unsigned int flags = (unsigned int)zoL[ZO_FLAGS];
    const bool any = A.P > 0, useP = any && M > 1.0, lhs = Lh >= La;
    const double vP = G_rho * q * (-UB);          // UB = 1/M : d/d eta_h[P] = d/d eta_a[P] = -1/M
    const double vL = G_rho * (1.0 - q) * (-LB);  // LB = -1/Lam : d/d eta = +1/Lam
    const int hP = pP & 0xFFFFu, aP = pP >> 16;
    const int hL = (lhs ? pQ : pR) & 0xFFFFu, aL = (lhs ? pQ : pR) >> 16;
    const bool p1 = useP && !(flags & 1u), p2 = useP && !(flags & 2u);
    const bool l1 = any && lhs && !(flags & 4u), l2 = any && !lhs && !(flags & 8u);
    double corr = 0.0;
    DC_STAMP_SEQ(12);
    auto team = [&](int t, double* ga_, double* gd_, double* gh_) {
        const double ra = g_att[t], rd = g_def[t], rh = g_ha[t];
        corr -= rh * eps[t] + (ra - rh) * eps[T + t] + rd * eps[2 * T + t];
        double ga = cL[t] - ra, gd = -(cL[T + t] - rd), gh = cL[2 * T + t] - rh;
        if (p1 && t == hP) { ga += vP; gh += vP; }
        if (p1 && t == aP) gd -= vP;
        if (p2 && t == aP) ga += vP;
        if (p2 && t == hP) gd -= vP;
        if (l1 && t == hL) { ga += vL; gh += vL; }
        if (l1 && t == aL) gd -= vL;
        if (l2 && t == aL) ga += vL;
        if (l2 && t == hL) gd -= vL;
        *ga_ = ga; *gd_ = gd; *gh_ = gh;
    };
    if (!EXT) {
        // v: 0 sum g_def, 1 sum g_ha, 2 sum a~ g_att, 3 sum d~ g_def, 4 correction
        double v[5] = {0, 0, 0, 0, 0};
        for (int t = tid; t < T; t += BLOCK) {
            double ga, gd, gh;
            team(t, &ga, &gd, &gh);
            const double ad = zL[L.o_adec + t], dd = zL[L.o_ddec + t];
            grad[L.o_adec + t] = gz[L.o_adec + t] - s_a * ga;
            grad[L.o_ddec + t] = gz[L.o_ddec + t] - s_d * gd;
            v[0] += gd;
            v[1] += gh;
            v[2] += ad * ga;
            v[3] += dd * gd;
        }
        v[4] = corr;
        DC_STAMP_SEQ(13);
        block_sum<5, true>(v, scratch, tid);
        DC_STAMP_SEQ(14);
        if (tid == 0) {
            grad[L.o_ha] = gz[L.o_ha] - v[1];
            grad[L.o_md] = gz[L.o_md] - v[0];
            grad[L.o_sa] = gz[L.o_sa] - s_a * v[2];
            grad[L.o_sd] = gz[L.o_sd] - s_d * v[3];
            grad[L.o_corr] = gz[L.o_corr] - G_rho * (UB - LB) * dq;
            const double Ltot =
                Lz + v[4] + zoL[ZO_PAIRC] + G_rho * drho - SLAM - A.lgsum + LN2 * SLOG - CLIPC;
            *pot_of(A, chain) = -Ltot;
        }
    } else {
        // v: 0 sum g_def, 1 sum g_ha, 2 sum sa g_att, 3 sum sd g_def, 4 sum ha~ g_ha, 5 corr
        double v[6] = {0, 0, 0, 0, 0, 0};
        for (int t = tid; t < T; t += BLOCK) {
            double ga, gd, gh;
            team(t, &ga, &gd, &gh);
            // (the coefficient sums below read every team's: this thread's own entries, behind its own reads)
            g_att[t] = ga;
            g_def[t] = gd;
            const double sa = zL[L.o_sat + t], sd = zL[L.o_sdt + t], hd = zL[L.o_hadec + t];
            grad[L.o_sat + t] = gz[L.o_sat + t] - s_a * ga;
            grad[L.o_sdt + t] = gz[L.o_sdt + t] - s_d * gd;
            grad[L.o_hadec + t] = gz[L.o_hadec + t] - s_h * gh;
            v[0] += gd;
            v[1] += gh;
            v[2] += sa * ga;
            v[3] += sd * gd;
            v[4] += hd * gh;
        }
        v[5] = corr;
        block_sum<6, true>(v, scratch, tid);   // (its barriers also publish g_att / g_def)
        for (int k = tid; k < 2 * K; k += BLOCK) {  // d/d beta_k: sum_t Xs[t,k] g_t
            const bool isd = k >= K;
            const int kk = isd ? k - K : k;
            const double* gt = isd ? g_def : g_att;
            double s = 0.0;
            for (int t = 0; t < T; ++t) s += A.xs[(size_t)t * K + kk] * gt[t];
            const int o = (isd ? L.o_bD : L.o_bA) + kk;
            grad[o] = gz[o] - s;
        }
        if (tid == 0) {
            grad[L.o_mha] = gz[L.o_mha] - v[1];
            grad[L.o_sh] = gz[L.o_sh] - s_h * v[4];
            grad[L.o_md] = gz[L.o_md] - v[0];
            grad[L.o_sa] = gz[L.o_sa] - s_a * v[2];
            grad[L.o_sd] = gz[L.o_sd] - s_d * v[3];
            grad[L.o_corr] = gz[L.o_corr] - G_rho * (UB - LB) * dq;
            grad[L.o_u] = gz[L.o_u];
            const double Ltot =
                Lz + v[5] + zoL[ZO_PAIRC] + G_rho * drho - SLAM - A.lgsum + LN2 * SLOG - CLIPC;
            *pot_of(A, chain) = -Ltot;
        }
    }
    if (tid == 0 && A.aux != nullptr) {
        double* aux = aux_of(A, chain);
        aux[0] = rho;
        aux[1] = LB;
        aux[2] = UB;
        aux[3] = q;
    }
}

template <bool STAGED, bool NUTS, bool EXT>
__device__ void tail_body(const EvalArgs& A, int chain, char* smem) {
    const Layout& L = A.L;
    const int T = L.T, K = L.K, D = L.D;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int ncol = 3 * T;
    const int nsc = A.n_wg * N_SCAL;

    // LDS carve: the hand-off record keeps its global order [zo | scal | compact]
    double* hL = reinterpret_cast<double*>(smem);
    double* zoL = hL;                                // [zo_stride]  prior workgroup record
    double* scl = zoL + A.zo_stride;                 // [n_wg*N_SCAL]
    double* cmp = scl + nsc;                         // [total_c] (STAGED)
    double* cL = cmp + (STAGED ? A.total_c : 0);     // [3T] cA | cD | cH
    double* zL = cL + 3 * T;                         // [D]
    double* col = zL + D;                            // [3T + N_SCAL + 4] reduced sums
    double* scratch = col + ncol + N_SCAL + 4;       // [WAVES*8]
    double* xsL = scratch + WAVES * 8;               // [T*K] when K <= 16
    double* gradL = xsL + (size_t)T * xs_staged_k(K);  // [D+8] grad | U | aux (NUTS hand-over)
    double* stg = gradL + D + 8;                        // [6*D] NUTS leaf vectors when D > 64
    int* coff = reinterpret_cast<int*>(stg + (STAGED && D > 64 ? nd::LEAF_STAGE_VECS * D : 0));  // [3T+1]
    DC_STAMP(7);

    // device-resident NUTS: wave LEAF_WAVE (idle during the per-team epilogue) will book the
    // leaf; its state is requested right after its hand-off loads below -- loads return in
    // order, so the (colder) state lines must not sit in front of them
    nd::LeafState<1> leaf1{};            // D <= 64: the leaf's vectors in registers (waves 4, 5)
    double bigv[nd::LEAF_STAGE_LOADS];   // D > 64: waves 4..7 stage one 64-element slice each
    const bool small = D <= 64;

    // ---- 1. ONE round of loads: every global value the tail needs is requested before
    // the first one is used (a rolled load -> LDS-store loop would serialise them)
    const double* hb = A.hbuf + (size_t)chain * A.hb_stride;
    const double* compact = hb + A.zo_stride + nsc;
    const double* z = z_of(A, chain);
    const int nstage = A.zo_stride + nsc + (STAGED ? A.total_c : 0);
    const bool xs_staged = K > 0 && K <= 16;
    {
        // static / caller data first (plain loads), one element per thread per pass
        const int i = tid;
        const double c0 = i < ncol ? (i < T ? A.cA[i] : (i < 2 * T ? A.cD[i - T] : A.cH[i - 2 * T])) : 0.0;
        const double z0 = i < D ? z[i] : 0.0;
        const int o0 = i <= ncol ? A.col_off[i] : 0;
        const double x0 = (xs_staged && i < T * K) ? A.xs[i] : 0.0;
        double v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int j = u * BLOCK + tid;
            v[u] = ld_sc1(&hb[j < nstage ? j : nstage - 1]);
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int j = u * BLOCK + tid;
            if (j < nstage) hL[j] = v[u];
        }
        if (i < ncol) cL[i] = c0;
        if (i < D) zL[i] = z0;
        if (i <= ncol) coff[i] = o0;
        if (xs_staged && i < T * K) xsL[i] = x0;
        // (the leaf's state is requested after these stores: behind the branch on the wave the compiler
        // would make the stores wait for the -- colder -- state loads as well.  They still land long
        // before the leaf starts.)
        if (NUTS && STAGED) {
            double* ns = nuts_of(A, chain);
            if (small) {
                if (wave == LEAF_WAVE || wave == RNG_WAVE)
                    leaf1 = nd::leaf_prefetch<1>(ns, D, A.nuts_max_depth, lane);
            } else if (wave >= 4) {
                static_assert(WAVES - 4 == nd::LEAF_NE_MAX, "one idle wave per 64-element slice");
                const int i = tid - 4 * 64;
                const int which[nd::LEAF_STAGE_LOADS] = {nd::V_INVM, nd::V_ZN, nd::V_RH, nd::V_S_RSUM,
                                                         nd::V_SL_R, nd::V_SR_R};
#pragma unroll
                for (int k = 0; k < nd::LEAF_STAGE_LOADS; ++k)
                    bigv[k] = i < D ? nd::vec(ns, D, which[k])[i] : 0.0;
                if (wave == LEAF_WAVE || wave == RNG_WAVE) leaf1.hv = lane < nd::H_N ? ns[lane] : 0.0;
            }
        }
    }
#pragma unroll 1
    for (int i0 = 8 * BLOCK; i0 < nstage; i0 += 8 * BLOCK) {  // larger records: more batches
        double v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int j = i0 + u * BLOCK + tid;
            v[u] = ld_sc1(&hb[j < nstage ? j : nstage - 1]);
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int j = i0 + u * BLOCK + tid;
            if (j < nstage) hL[j] = v[u];
        }
    }
    for (int i = tid + BLOCK; i < ncol; i += BLOCK)
        cL[i] = i < T ? A.cA[i] : (i < 2 * T ? A.cD[i - T] : A.cH[i - 2 * T]);
    for (int i = tid + BLOCK; i < D; i += BLOCK) zL[i] = z[i];
    for (int i = tid + BLOCK; i <= ncol; i += BLOCK) coff[i] = A.col_off[i];
    if (xs_staged)
        for (int i = tid + BLOCK; i < T * K; i += BLOCK) xsL[i] = A.xs[i];
    __syncthreads();
    DC_STAMP(8);

    // ---- 2. fixed-order (deterministic) column sums of the sparse slabs: the compact
    // array is stored column-major, 8 lanes share a column, half-row DPP folds them
    for (int c0 = 0; c0 < ncol; c0 += BLOCK / 8) {
        const int c = c0 + (tid >> 3), j = tid & 7;
        double s = 0.0;
        if (c < ncol) {
            const int k1 = coff[c + 1];
            if (STAGED) {
                for (int k = coff[c] + j; k < k1; k += 8) s += cmp[k];
            } else {
                // (the compact array does not fit LDS here: straight from memory, eight loads in
                // flight per lane -- one at a time was a ~1 us round trip per 8 workgroups)
                for (int k = coff[c] + j; k < k1; k += 64) {
                    double v[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) v[u] = ld_sc1(&compact[min(k + 8 * u, k1 - 1)]);
#pragma unroll
                    for (int u = 0; u < 8; ++u) s += k + 8 * u < k1 ? v[u] : 0.0;
                }
            }
        }
        s += dpp_f64<0xB1>(0.0, s);   // quad_perm [1,0,3,2]
        s += dpp_f64<0x4E>(0.0, s);   // quad_perm [2,3,0,1]
        s += dpp_f64<0x141>(0.0, s);  // row_half_mirror: the other quad of the 8
        if (c < ncol && j == 0) col[c] = s;
    }
    if (wave < N_SCAL) {  // scalar `wave`: lanes stride the workgroups, then a DPP sum
        double s = 0.0;
        for (int r = lane; r < A.n_wg; r += 64) s += scl[r * N_SCAL + wave];
        s = wave_sum_f64(s);
        if (lane == 0) col[ncol + wave] = s;
    }
    __syncthreads();
    DC_STAMP(9);
    if (DC_ILL_CALL && zoL[ZO_ILL] != 0.0) ill_add_filed(A, zoL, col);   // (uniform; rare: see class_terms, prior_body)
    // STAGED <=> T <= 64 (host): the two epilogues never meet in one instantiation (code size
    // matters at ~9 us per launch)
    if (STAGED) {  // lane = team: four waves, one output group each, no LDS traffic
        if (NUTS) {  // the idle waves' shares of the leaf preparation (see LEAF_WAVE)
            if (!small && wave >= 4) {
                const int i = tid - 4 * 64;
#pragma unroll
                for (int k = 0; k < nd::LEAF_STAGE_LOADS; ++k)
                    if (i < D) stg[k * D + i] = bigv[k];
            }
        }
        tail_waves<NUTS, EXT>(A, chain, zoL, cL, zL, col, xs_staged ? xsL : nullptr, gradL, leaf1, stg);
        DC_STAMP(10);
        return;
    }
    tail_general<EXT>(A, chain, zoL, cL, zL, col, scratch);
    DC_STAMP(10);
}

// dc_eval's tail over the accumulator rows (GA_ROW), run by the prior workgroup once every
// streaming workgroup has arrived: ONE round of loads -- the 3T + 4*GA_SHARDS reduced values, z
// and the static per-team sums (the prior record is already in LDS) -- then one barrier and the
// epilogue.  The rows are re-armed (zeroed, write-through) for the next launch of this chain.  LDS: [zo | cL | zL | col | scratch | xs | gradL |
// leaf stage].  Waves 0..6 take the team rows, wave 7 the scalar rows: lane = 16*scalar + shard,
// so the shard sums are 16-lane DPP row sums.
__host__ __device__ inline size_t acc_tail_lds_bytes(int T, int D, int K, int zo_stride, bool stage) {
    size_t d = (size_t)zo_stride + 3 * (size_t)T + D + (3 * (size_t)T + N_SCAL + 4) + WAVES * 8 +
               (size_t)T * xs_staged_k(K) + 2 * ((size_t)D + 8) +
               (stage && D > 64 ? (size_t)nd::LEAF_STAGE_VECS * D : 0) + (stage ? 2 * leaf_park_doubles(D) + 8 : 0);
    return d * 8 + 16;
}
// Everything the tail needs that does NOT depend on the streaming workgroups is requested by
// tail_preload before the prior workgroup starts waiting for them (z, the static per-team sums,
// the covariates, the NUTS leaf's state); after the wait only the accumulator rows are loaded.
struct TailPre {
    double c0, z0, x0;
    int expect;   // contributions this thread's (first) accumulator row receives per evaluation
    // larger leagues (BIG): the second round of the static sums and of the position, and the expected counts of
    // the thread's four batched rows (tail_acc) -- requested behind the rows' poll they were two more memory
    // round trips between the prior part and the epilogue (1.6 us from the last row add to the epilogue at 200 teams)
    double c1, z1;
    int ex[4];
};
constexpr int ROW_THREADS = (WAVES - 1) * 64;   // waves 0..6 take team rows, wave 7 the scalar rows
// larger leagues: row k of a thread's batch of up to four (tail_acc).  Loads that bypass the L1 are served one
// after the other PER LINE at the memory side, so a lane without a k-th row must not invent a request: it repeats
// the address of the first lane of its own wave instruction (same instruction, same line: one request), and a
// wave whose whole k-th instruction is past the end repeats its first row's.  (Measured at 200 teams: clamped to
// the last row -- 300 threads on one line -- 0.6 us per evaluation; every lane its own first row again -- four
// requests per line -- 1.7 us.)  The batch holds as many loads as the league needs (batched_loads).
template <int RT = ROW_THREADS>
__device__ __forceinline__ int batched_row(int b0, int k, int lane, int ncol) {
    const int i0 = b0 + k * RT;
    return i0 + lane < ncol ? i0 + lane : (i0 < ncol ? i0 : min(b0, ncol - 1));
}
template <int RT = ROW_THREADS>
__device__ __forceinline__ int batched_loads(int b0_first, int ncol) {   // of the batch that starts at wave 0's b0
    return min(4, (ncol - b0_first + RT - 1) / RT);
}
// EARLY (dc_eval past 64 teams, round 4): the rows are polled by waves 1..6 (team rows) and 7 (scalar rows) as
// soon as each is through with the prior part -- while wave 0's first lane still works the bounds' record out
// (0.85 us) -- instead of behind the barrier that ends it: at 200 teams the streaming workgroups are done at 4 us
// and the prior part at 7, so the first poll finds every row complete, and its round trip (1.1 us with two rows
// per thread) and what led up to it (0.4 us) leave the critical path.
constexpr int EARLY_THREADS = (WAVES - 2) * 64;
// the (first) accumulator row a thread of the tail takes: waves 0..6 the team rows, wave 7 the scalar
// rows, lane = 16*scalar + shard
__device__ __forceinline__ int tail_row_of(int tid, int ncol) {
    const int lane = tid & 63, wave = tid >> 6;
    return wave < WAVES - 1 ? min(tid, ncol - 1) : ncol + (lane & 15) * N_SCAL + (lane >> 4);
}
// ZL: the position already sits in the tail's LDS copy zL (persistent kernel): not loaded, not staged
// the data-only part (z, static sums, covariates, expected counts): a plain launch requests it at
// kernel entry, IN FRONT of the prior part -- requested behind it, the barrier that follows waited a
// memory round trip (0.4 us) for it with the prior record already done
template <bool ZL = false, bool BIG = false>
__device__ __forceinline__ void tail_preload_static(const EvalArgs& A, int chain, TailPre& P) {
    const Layout& L = A.L;
    const int T = L.T, K = L.K, D = L.D;
    const int tid = threadIdx.x;
    const int ncol = 3 * T, i = tid;
    const double* z = z_of(A, chain);
    const bool xs_staged = K > 0 && K <= 16;
    // (ZL: the persistent kernel filled cL / zL / xsL once; nothing to load per step)
    P.c0 = !ZL && i < ncol ? (i < T ? A.cA[i] : (i < 2 * T ? A.cD[i - T] : A.cH[i - 2 * T])) : 0.0;
    P.z0 = !ZL && i < D ? z[i] : 0.0;
    P.x0 = (!ZL && xs_staged && i < T * K) ? A.xs[i] : 0.0;
    P.expect = A.ga_expect[tail_row_of(tid, ncol)];
    if (BIG) {
        const int i1 = tid + BLOCK;
        P.c1 = !ZL && i1 < ncol ? (i1 < T ? A.cA[i1] : (i1 < 2 * T ? A.cD[i1 - T] : A.cH[i1 - 2 * T])) : 0.0;
        P.z1 = !ZL && i1 < D ? z[i1] : 0.0;
#pragma unroll
        for (int k = 0; k < 4; ++k)
            P.ex[k] = A.ga_expect[batched_row<EARLY_THREADS>(max((tid & ~63) - 64, 0), k, tid & 63, ncol)];
    }
}
template <bool SMALLT, bool NUTS, bool ZL = false, int LNE>
__device__ __forceinline__ void tail_preload(const EvalArgs& A, int chain, TailPre& P,
                                             nd::LeafState<LNE>& leaf1,
                                             double (&bigv)[nd::LEAF_STAGE_LOADS], bool static_done = false,
                                             bool want_leaf = true) {
    const Layout& L = A.L;
    const int D = L.D;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (!static_done) tail_preload_static<ZL, !SMALLT>(A, chain, P);
    if (NUTS && SMALLT && want_leaf) {
        double* ns = nuts_of(A, chain);
        if (D <= 64 * LNE) {
            if (wave == LEAF_WAVE || wave == RNG_WAVE)
                leaf1 = nd::leaf_prefetch<LNE>(ns, D, A.nuts_max_depth, lane);
        } else if (wave >= 4) {
            static_assert(WAVES - 4 == nd::LEAF_NE_MAX, "one idle wave per 64-element slice");
            const int i4 = tid - 4 * 64;
            const int which[nd::LEAF_STAGE_LOADS] = {nd::V_INVM, nd::V_ZN, nd::V_RH, nd::V_S_RSUM,
                                                     nd::V_SL_R, nd::V_SR_R};
#pragma unroll
            for (int k = 0; k < nd::LEAF_STAGE_LOADS; ++k)
                bigv[k] = i4 < D ? nd::vec(ns, D, which[k])[i4] : 0.0;
            if (wave == LEAF_WAVE || wave == RNG_WAVE) leaf1.hv = lane < nd::H_N ? ns[lane] : 0.0;
        }
    }
}
// one poll of a row (wave-uniform result)
__device__ __forceinline__ bool ga_poll_row(const long long* row, int expect, GaWords* out, const long long* other) {
    bool mine = true;
    if (other != nullptr) {
        GaWords o;
        ga_load2(row, other, out, &o);
        mine = ga_is_zero(o);
    } else {
        *out = ga_load(row);
    }
    mine = mine && ga_count(*out) == expect;
    return __ballot(!mine) == 0ull;
}
// (*okflag must hold 1 and a barrier must lie between that store and this call.)  false: the bounded
// wait for the rows expired (the caller poisons the outputs).
// one accumulator row until its count is complete (wave-uniform exit; bounded like every wait here)
// `other` (persistent kernel): the same row of the chain's other set must read all zero as well (the
// next step's rows, re-armed one step ago: see ga_set_words) -- requested in the SAME round of loads
__device__ __forceinline__ bool ga_take_row(const long long* row, int expect, GaWords* out,
                                            const long long* other = nullptr) {
    bool ok = false;
    for (int spin = 0; spin < ARRIVE_SPIN_LIMIT; ++spin) {
        ok = ga_poll_row(row, expect, out, other);
        if (ok) break;
        __builtin_amdgcn_s_sleep(1);
    }
    poll_acquired();
    return ok;
}
// (Tried and measured slower, twice -- profiles/r03/stamps_early_poll.txt: polling the rows EARLY,
// from waves that idle while one lane each of waves 0 and 1 finishes the prior record.  With one
// dependent load per poll the first came back too early and the second too late (8.5 us per
// leapfrog against 7.8); with four polls in flight the record's serial tail itself slowed down by
// 1.2 us next to the polling waves (9.0).  The rows are polled when the prior part is done.
// A third variant -- ONE poll issued in front of the barrier that ends the prior part, so that its
// round trip covers tail_preload's loads and that barrier -- looked 0.6 us better in the stamped
// build and is 7 % slower in the shipped one (profiles/r03/ab_first_poll.txt, same box: 119.7k against
// 128-130k leapfrogs/s; the plain launch 7.05 against 6.77 us): there it comes back just before the
// last rows are complete, and the poll that follows starts a full round trip later than it would have.)
// `set`: which of the chain's two row sets this evaluation used; CHECK_OTHER (persistent kernel): the
// wait also covers the other set reading all zero (see ga_set_words)
// the early poll itself: every wave but the first, once it has left prior_body (no barrier in between: okflag
// was raised in front of the prior part).  Results into the tail's `col`, as tail_acc files them.
__device__ __forceinline__ void tail_poll_early(const EvalArgs& A, int chain, char* smem, const TailPre& P, int* okflag) {
    const Layout& L = A.L;
    const int T = L.T, D = L.D;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ncol = 3 * T;
    if (wave == 0) return;
    double* col = reinterpret_cast<double*>(smem) + A.zo_stride + ncol + D;
    long long* ga = A.gacc + (size_t)chain * 2 * ga_set_words(T);
    bool ok = true;
    DC_STAMP_WAVE(1, 11);
    if (wave < WAVES - 1) {
        for (int b0 = (wave - 1) * 64; b0 < ncol; b0 += 4 * EARLY_THREADS) {
            int r[4], ex[4];
            const int nload = max(2, batched_loads<EARLY_THREADS>(b0 - (wave - 1) * 64, ncol));
            const bool pre_ex = b0 == (wave - 1) * 64;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                r[k] = batched_row<EARLY_THREADS>(b0, k, lane, ncol);
                ex[k] = pre_ex ? P.ex[k] : A.ga_expect[r[k]];
            }
            GaWords w[4];
            bool got = false;
            for (int spin = 0; spin < ARRIVE_SPIN_LIMIT; ++spin) {
                if (nload == 2)   // (wave uniform)
                    ga_load2rows(ga + (size_t)r[0] * GA_ROW, ga + (size_t)r[1] * GA_ROW, w);
                else if (nload == 3)
                    ga_load3(ga + (size_t)r[0] * GA_ROW, ga + (size_t)r[1] * GA_ROW, ga + (size_t)r[2] * GA_ROW, w);
                else
                    ga_load4(ga + (size_t)r[0] * GA_ROW, ga + (size_t)r[1] * GA_ROW, ga + (size_t)r[2] * GA_ROW,
                             ga + (size_t)r[3] * GA_ROW, w);
                const bool mine = ga_count(w[0]) == ex[0] && ga_count(w[1]) == ex[1] &&
                                  (nload < 3 || ga_count(w[2]) == ex[2]) && (nload < 4 || ga_count(w[3]) == ex[3]);
                got = __ballot(!mine) == 0ull;
                if (got) break;
                __builtin_amdgcn_s_sleep(1);
            }
            ok = got && ok;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int i2 = b0 + k * EARLY_THREADS + lane;
                if (k < nload && i2 < ncol) col[i2] = ga_value(w[k]);
            }
        }
    } else {  // scalar rows: sum the shards (as tail_acc)
        GaWords w0;
        ok = ga_take_row(ga + (size_t)tail_row_of(tid, ncol) * GA_ROW, P.expect, &w0);
        double v = ga_value(w0);
        v += dpp_f64<0xB1>(0.0, v);   // quad_perm [1,0,3,2]
        v += dpp_f64<0x4E>(0.0, v);   // quad_perm [2,3,0,1]
        v += dpp_f64<0x124>(0.0, v);  // row_ror:4
        v += dpp_f64<0x128>(0.0, v);  // row_ror:8
        if ((lane & 15) == 0) col[ncol + (lane >> 4)] = v;
    }
    poll_acquired();
    DC_STAMP_WAVE(1, 6);
    if (!ok && lane == 0) *okflag = 0;
}
// POLLED: tail_poll_early has taken the rows
template <bool SMALLT, bool NUTS, bool EXT, bool ZL = false, bool POLLED = false, int LNE>
__device__ __forceinline__ bool tail_acc(const EvalArgs& A, int chain, char* smem, const TailPre& P,
                                         const nd::LeafState<LNE>& leaf1,
                                         const double (&bigv)[nd::LEAF_STAGE_LOADS], int* okflag,
                                         unsigned int pub_tag = 0u, int set = 0, bool check_other = false) {
    const Layout& L = A.L;
    const int T = L.T, K = L.K, D = L.D;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int ncol = 3 * T;
    static_assert(N_SCAL * GA_SHARDS == 64 && WAVES == 8, "wave 7 = the scalar rows");

    double* zoL = reinterpret_cast<double*>(smem);      // [zo_stride]  prior workgroup record
    double* cL = zoL + A.zo_stride;                     // [3T] cA | cD | cH
    double* zL = cL + ncol;                             // [D]
    double* col = zL + D;                               // [3T + N_SCAL + 4] reduced sums
    double* scratch = col + ncol + N_SCAL + 4;          // [WAVES*8]
    double* xsL = scratch + WAVES * 8;                  // [T*K] when K <= 16
    double* gradL = xsL + (size_t)T * xs_staged_k(K);   // [2][D+8] grad | U | aux (NUTS hand-over; the persistent kernel alternates)
    double* stg = gradL + 2 * (D + 8);                  // [7*D] NUTS leaf vectors when D > 64
    DC_STAMP(7);

    const bool small = D <= 64 * LNE;
    const double* z = z_of(A, chain);
    long long* ga = A.gacc + ((size_t)chain * 2 + set) * ga_set_words(T);
    const bool xs_staged = K > 0 && K <= 16;
    {
        // ---- 1. poll this thread's row until every workgroup that feeds it has added: the load that
        // finds the count complete IS the hand-off (no arrival counter, no second round of loads)
        const int i = tid;
        GaWords w0;
        const size_t ro = (size_t)tail_row_of(tid, ncol) * GA_ROW;
        // (persistent kernel: the next step's rows, re-armed one step ago, must read all zero too)
        // (larger leagues: a team-row thread takes its first row with the others, in one round trip)
        const bool batched = !POLLED && !SMALLT && !check_other && wave < WAVES - 1 && ncol > ROW_THREADS;
        bool ok = batched || POLLED ||
                  ga_take_row(ga + ro, P.expect, &w0,
                              check_other ? A.gacc + ((size_t)chain * 2 + (set ^ 1)) * ga_set_words(T) + ro : nullptr);
        if (!ZL && i < ncol) cL[i] = P.c0;
        if (!ZL && i < D) zL[i] = P.z0;
        if (!ZL && xs_staged && i < T * K) xsL[i] = P.x0;
        if (POLLED) {
            // (nothing: col holds the rows)
        } else if (wave < WAVES - 1) {
            if (!batched && i < ncol) col[i] = ga_value(w0);
            // larger models: the remaining team rows, up to FOUR per thread and poll in flight together
            // (wave-uniform trip count: the polls ballot; lanes past the last row: batched_row)
            // (SMALLT: at most 192 rows, all of them taken above -- and no trace of this loop in those kernels)
            for (int b0 = (tid & ~63) + (batched ? 0 : ROW_THREADS); !SMALLT && b0 < ncol; b0 += 4 * ROW_THREADS) {
                int r[4], ex[4];
                const int nload = max(2, batched_loads(b0 - (tid & ~63), ncol));
                // (the first batch's expected counts came with the prologue's loads: TailPre)
                const bool pre_ex = false;   // (TailPre's counts follow the early poll's row mapping)
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    r[k] = batched_row(b0, k, lane, ncol);
                    ex[k] = pre_ex ? P.ex[k] : A.ga_expect[r[k]];
                }
                GaWords w[4];
                bool got = false;
                DC_STAMP(11);
                for (int spin = 0; spin < ARRIVE_SPIN_LIMIT; ++spin) {
                    if (nload == 2)   // (wave uniform)
                        ga_load2rows(ga + (size_t)r[0] * GA_ROW, ga + (size_t)r[1] * GA_ROW, w);
                    else if (nload == 3)
                        ga_load3(ga + (size_t)r[0] * GA_ROW, ga + (size_t)r[1] * GA_ROW, ga + (size_t)r[2] * GA_ROW, w);
                    else
                        ga_load4(ga + (size_t)r[0] * GA_ROW, ga + (size_t)r[1] * GA_ROW, ga + (size_t)r[2] * GA_ROW,
                                 ga + (size_t)r[3] * GA_ROW, w);
                    const bool mine = ga_count(w[0]) == ex[0] && ga_count(w[1]) == ex[1] &&
                                      (nload < 3 || ga_count(w[2]) == ex[2]) && (nload < 4 || ga_count(w[3]) == ex[3]);
                    got = __ballot(!mine) == 0ull;
                    if (got) break;
                    __builtin_amdgcn_s_sleep(1);
                }
                ok = got && ok;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int i2 = b0 + k * ROW_THREADS + lane;
                    if (i2 < ncol) col[i2] = ga_value(w[k]);
                }
            }
        } else {  // scalar rows: sum the shards (16-lane rows; a flagged shard is -inf / NaN and stays so)
            double v = ga_value(w0);
            v += dpp_f64<0xB1>(0.0, v);   // quad_perm [1,0,3,2]
            v += dpp_f64<0x4E>(0.0, v);   // quad_perm [2,3,0,1]
            v += dpp_f64<0x124>(0.0, v);  // row_ror:4
            v += dpp_f64<0x128>(0.0, v);  // row_ror:8
            if ((lane & 15) == 0) col[ncol + (lane >> 4)] = v;
        }
        if (!ok && lane == 0) *okflag = 0;
    }
    DC_STAMP(14);
    if (!ZL) {
        if (!SMALLT) {
            if (tid + BLOCK < ncol) cL[tid + BLOCK] = P.c1;
            if (tid + BLOCK < D) zL[tid + BLOCK] = P.z1;
        }
        for (int i = tid + 2 * BLOCK; !SMALLT && i < ncol; i += BLOCK)   // (SMALLT: 3T <= 192)
            cL[i] = i < T ? A.cA[i] : (i < 2 * T ? A.cD[i - T] : A.cH[i - 2 * T]);
        for (int i = tid + (SMALLT ? 1 : 2) * BLOCK; i < D; i += BLOCK) zL[i] = z[i];
        if (xs_staged)
            for (int i = tid + BLOCK; i < T * K; i += BLOCK) xsL[i] = A.xs[i];
    }
    __syncthreads();
    DC_STAMP(8);
    if (*okflag == 0) return false;
    // re-arm the rows for this chain's next launch: write-through zeros, issued only now -- a
    // barrier waits for the wave's outstanding stores, and these take a memory round trip
    if (wave < WAVES - 1) {
        for (int i = tid; i < ncol; i += ROW_THREADS) ga_rearm(ga + (size_t)i * GA_ROW);
    } else {
        ga_rearm(ga + (size_t)(ncol + (lane & 15) * N_SCAL + (lane >> 4)) * GA_ROW);
    }
    DC_STAMP(9);
    if (DC_ILL_CALL && zoL[ZO_ILL] != 0.0)   // (uniform; rare: see class_terms.  The prior part's LDS sits behind the tail's arrays)
        ill_pass_lds<EXT>(A, zoL, col, scratch,
                          smem + ((acc_tail_lds_bytes(T, D, K, A.zo_stride, SMALLT && NUTS) + 15) & ~(size_t)15));
    if (SMALLT) {  // lane = team: four waves, one output group each, no LDS traffic
        if (NUTS) {  // the idle waves' shares of the leaf preparation (see LEAF_WAVE)
            if (!small && wave >= 4) {
                const int i = tid - 4 * 64;
#pragma unroll
                for (int k = 0; k < nd::LEAF_STAGE_LOADS; ++k)
                    if (i < D) stg[k * D + i] = bigv[k];
            }
        }
        tail_waves<NUTS, EXT>(A, chain, zoL, cL, zL, col, xs_staged ? xsL : nullptr, gradL, leaf1, stg, pub_tag,
                              ZL ? zL : nullptr);
        DC_STAMP(10);
        return true;
    }
    tail_general<EXT>(A, chain, zoL, cL, zL, col, scratch);
    DC_STAMP(10);
    return true;
}
// a flag word for the tail's wait, inside the tail's own arrays (its `scratch`, which nothing uses
// before the epilogue): it can be set BEFORE the barrier that ends the prior part
__device__ __forceinline__ int* acc_tail_flag(const EvalArgs& A, char* smem) {
    const Layout& L = A.L;
    return reinterpret_cast<int*>(reinterpret_cast<double*>(smem) + A.zo_stride + 3 * L.T + L.D + (3 * L.T + N_SCAL + 4));
}
// the persistent kernel's parked leaf states and flags (behind the leaf's staging vectors)
__device__ __forceinline__ double* acc_tail_park(const EvalArgs& A, char* smem, bool stage) {
    const Layout& L = A.L;
    return reinterpret_cast<double*>(smem) + A.zo_stride + 3 * L.T + L.D + (3 * L.T + N_SCAL + 4) + WAVES * 8 +
           (size_t)L.T * xs_staged_k(L.K) + 2 * ((size_t)L.D + 8) +
           (stage && L.D > 64 ? (size_t)nd::LEAF_STAGE_VECS * L.D : 0);
}
// where tail_acc keeps its NUTS hand-over block gradL (grad | U | aux | sub_done | finished)
__device__ __forceinline__ double* acc_tail_gradL(const EvalArgs& A, char* smem) {
    const Layout& L = A.L;
    return reinterpret_cast<double*>(smem) + A.zo_stride + 3 * L.T + L.D + (3 * L.T + N_SCAL + 4) + WAVES * 8 +
           (size_t)L.T * xs_staged_k(L.K);
}

// ------------------------------------------------------------ persistent kernel: the tail of one step
// Round 4: THE LEAF IS BOOKED ONE STEP BEHIND ITS EVALUATION.  While a subtree goes on, the next leapfrog
// starts at zn + eps M^-1 (r_half - eps g): one fma chain per element once the gradient exists, whereas the
// leaf's decisions (energy, U-turn test, multinomial transition) only matter when they END the subtree.
// Whether the subtree is complete after a leaf is known from the header (num + 1 == S_MAX: that leaf is
// booked at once, the sequential path); otherwise only a U-turn or a divergence ends it -- about once per
// transition.  So the leaf wave publishes the next position right behind the epilogue's barrier, and the
// leaf itself is booked during the NEXT step's wait for the accumulator rows and its epilogue, on the two
// waves that idle there (LEAF_WAVE: leaf_window_moves, across the barrier that ends the wait; RNG_WAVE:
// leaf_window_weights, beside the epilogue).  Its state is not re-read from memory: each leaf wave
// forwards its registers (nd::leaf_forward_*) and parks them in LDS across the prior part.  A step was
//   prior 3.1 + wait 0.9 + sums 0.25 + epilogue 1.1 + leaf 0.8-1.0 + publish 0.25 us
// with the streaming workgroups done at 2.9 (profiles/r03/stamps_timeline.txt); the leaf and its
// publish leave that chain.  When the leaf of step s - 1 turns out to have ended its subtree, evaluation s
// -- consumed by then: no drain -- is simply dropped and the chain moves on from where it really is.
// (First tried with the leaf inside the next PRIOR part's first phase: that phase is 0.8 us, the leaf
// 1.3 us + 0.6 us of store drain in front of the phase's barrier -- 8.8 us per step instead of 7.5.)
enum { LF_SUBDONE = 0, LF_FIN = 1, LF_SPEC = 2, LF_N = 8 };
template <int LNE>
__device__ __forceinline__ void leaf_park(double* pk, const nd::LeafState<LNE>& S, int t, bool with_rng) {
    pk[t] = S.hv;
#pragma unroll
    for (int e = 0; e < LNE; ++e) {
        double* v = pk + 64 + (size_t)e * LEAF_PARK_VECS * 64 + t;
        v[0] = S.invM[e]; v[64] = S.zn[e]; v[128] = S.r[e]; v[192] = S.rs[e]; v[256] = S.sl_r[e]; v[320] = S.sr_r[e];
    }
    if (with_rng && t == 0) {
        double* w = pk + 64 + (size_t)LNE * LEAF_PARK_VECS * 64;
        w[0] = (double)S.nhi; w[1] = (double)S.nlo; w[2] = (double)S.u_take;
    }
}
template <int LNE>
__device__ __forceinline__ nd::LeafState<LNE> leaf_unpark(const double* pk, int t, bool with_rng) {
    nd::LeafState<LNE> S;
    S.hv = pk[t];
#pragma unroll
    for (int e = 0; e < LNE; ++e) {
        const double* v = pk + 64 + (size_t)e * LEAF_PARK_VECS * 64 + t;
        S.invM[e] = v[0]; S.zn[e] = v[64]; S.r[e] = v[128]; S.rs[e] = v[192]; S.sl_r[e] = v[256]; S.sr_r[e] = v[320];
    }
    if (with_rng) {
        const double* w = pk + 64 + (size_t)LNE * LEAF_PARK_VECS * 64;
        S.nhi = (uint32_t)w[0]; S.nlo = (uint32_t)w[1]; S.u_take = (float)w[2];
    }
    return S;
}
struct WorkgroupSync {
    __device__ __forceinline__ void operator()(bool) const { __syncthreads(); }
};
// LEAF_WAVE, in place of polling a row it does not own: the previous step's leaf (gradient and potential in
// gL_prev), the workgroup's barrier in the middle (after the decisions, before the stores)
template <int LNE>
__device__ __forceinline__ void leaf_window_moves(const EvalArgs& A, int chain, const double* gL_prev, double* pk,
                                                  double* flags, int t) {
    const int D = A.L.D;
    nd::LeafState<LNE> S = leaf_unpark<LNE>(pk, t, false);
    nd::leaf_prepare<false>(S);
    const bool done = nd::leaf_moves(nuts_of(A, chain), D, A.nuts_max_depth, t, gL_prev, S, (double*)nullptr,
                                     WorkgroupSync());
    if (t == 0) flags[LF_SUBDONE] = done ? 1.0 : 0.0;
    if (!done) {   // the subtree goes on: this wave's copy of the next leaf's state
        nd::leaf_forward_moves(S, D, t, gL_prev);
        leaf_park<LNE>(pk, S, t, false);
    }
}
// RNG_WAVE, likewise (dc_eval's tail polls the scalar rows on this wave; here another idle wave does):
// the other half of that leaf, then the next leaf's random numbers
template <int LNE>
__device__ __forceinline__ void leaf_window_weights(const EvalArgs& A, int chain, const double* gL_prev, double* pk,
                                                    int t) {
    const int D = A.L.D;
    nd::LeafState<LNE> S = leaf_unpark<LNE>(pk, t, true);
    const nd::LeafWeights W = nd::leaf_weights(nuts_of(A, chain), D, t, gL_prev, S, WorkgroupSync());
    nd::leaf_forward_weights(S, D, t, gL_prev, W);
    nd::leaf_rng(S.hv, &S.nhi, &S.nlo, &S.u_take);
    leaf_park<LNE>(pk, S, t, true);
}
// One step's wait + epilogue in the persistent kernel (T <= 64, D <= 64 LNE).  `prev`: the previous step's
// leaf is still to be booked (its state parked, its gradient in the other gradL buffer); otherwise leaf1 is
// this step's leaf as prefetched from memory, and is parked here.  false: the bounded wait expired.
template <bool EXT, int LNE>
__device__ __forceinline__ bool loop_tail(const EvalArgs& A, int chain, char* smem,
                                          const nd::LeafState<LNE>& leaf1, int* okflag, int set, bool prev) {
    const Layout& L = A.L;
    const int T = L.T, K = L.K, D = L.D;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int ncol = 3 * T;
    double* zoL = reinterpret_cast<double*>(smem);
    double* cL = zoL + A.zo_stride;
    double* zL = cL + ncol;
    double* col = zL + D;
    double* scratch = col + ncol + N_SCAL + 4;
    double* xsL = scratch + WAVES * 8;
    double* gradL0 = xsL + (size_t)T * xs_staged_k(K);
    double* gL_cur = gradL0 + (size_t)set * (D + 8);
    const double* gL_prev = gradL0 + (size_t)(set ^ 1) * (D + 8);
    double* pkL = acc_tail_park(A, smem, true);
    double* pkR = pkL + leaf_park_doubles(D);
    double* flags = pkR + leaf_park_doubles(D);
    const bool xs_staged = K > 0 && K <= 16;
    long long* ga = A.gacc + ((size_t)chain * 2 + set) * ga_set_words(T);
    // (T <= 64: the team rows are waves 0..2's; the scalar rows go to wave SCAL_WAVE here -- in dc_eval's tail
    // they are the last wave's, which is RNG_WAVE)
    constexpr int SCAL_WAVE = 5;
    static_assert(SCAL_WAVE != LEAF_WAVE && SCAL_WAVE != RNG_WAVE && SCAL_WAVE >= 3, "an idle wave");
    if (wave == LEAF_WAVE && prev) {
        leaf_window_moves<LNE>(A, chain, gL_prev, pkL, flags, lane);   // (the barrier below is inside)
    } else if (wave == RNG_WAVE && prev) {
        leaf_window_weights<LNE>(A, chain, gL_prev, pkR, lane);        // (likewise)
    } else {
        // poll this thread's row until every workgroup that feeds it has added (tail_acc); the other set --
        // the next step's, re-armed one step ago -- must read all zero in the same round of loads
        GaWords w0;
        const int row = wave == SCAL_WAVE ? ncol + (lane & 15) * N_SCAL + (lane >> 4) : min(tid, ncol - 1);
        const size_t ro = (size_t)row * GA_ROW;
        const bool ok = ga_take_row(ga + ro, A.ga_expect[row], &w0,
                                    A.gacc + ((size_t)chain * 2 + (set ^ 1)) * ga_set_words(T) + ro);
        if (wave != SCAL_WAVE) {
            if (tid < ncol) col[tid] = ga_value(w0);
        } else {  // scalar rows: sum the shards
            double v = ga_value(w0);
            v += dpp_f64<0xB1>(0.0, v);
            v += dpp_f64<0x4E>(0.0, v);
            v += dpp_f64<0x124>(0.0, v);
            v += dpp_f64<0x128>(0.0, v);
            if ((lane & 15) == 0) col[ncol + (lane >> 4)] = v;
        }
        if (!ok && lane == 0) *okflag = 0;
        if (wave == LEAF_WAVE) {   // (!prev: this step's leaf came from memory)
            leaf_park<LNE>(pkL, leaf1, lane, false);
            if (lane == 0) flags[LF_SUBDONE] = 0.0;
        }
        __syncthreads();
    }
    DC_STAMP(8);
    if (*okflag == 0) return false;
    if (wave == SCAL_WAVE) {
        ga_rearm(ga + (size_t)(ncol + (lane & 15) * N_SCAL + (lane >> 4)) * GA_ROW);
    } else if (tid < ncol) {
        ga_rearm(ga + (size_t)tid * GA_ROW);
    }
    DC_STAMP(9);
    if (DC_ILL_CALL && zoL[ZO_ILL] != 0.0)   // (uniform; rare: see class_terms)
        ill_pass_lds<EXT>(A, zoL, col, scratch, smem + ((acc_tail_lds_bytes(T, D, K, A.zo_stride, true) + 15) & ~(size_t)15));
    tail_waves<true, EXT, LNE, true>(A, chain, zoL, cL, zL, col, xs_staged ? xsL : nullptr, gL_cur, leaf1, nullptr);
    if (wave == RNG_WAVE && !prev) {   // this step's leaf came from memory: its random numbers, parked
        nd::LeafState<LNE> S = leaf1;
        nd::leaf_rng(S.hv, &S.nhi, &S.nlo, &S.u_take);
        leaf_park<LNE>(pkR, S, lane, true);
    }
    DC_STAMP(10);
    return true;
}

// ------------------------------------------------------------ per-lane fixture math

// the lane's LANE_FIX consecutive fixtures, still packed as loaded.  All of them share one
// (home, away) pair, so the indices are stored once per lane (run-length encoded):
//   hw[0] = home | (number of real fixtures of the lane) << 16,   aw[0] = away
struct LaneData {
    uint32_t hw[1], aw[1], xw[XWORDS], yw[XWORDS];
    float wj[LANE_FIX];
};
template <int NW>
__device__ __forceinline__ void load_words(const uint32_t* p, uint32_t (&d)[NW]) {
    if (NW == 8) {
        const uint4 v = *reinterpret_cast<const uint4*>(p), u = *reinterpret_cast<const uint4*>(p + 4);
        d[0] = v.x; d[1] = v.y; d[NW > 2 ? 2 : 0] = v.z; d[NW > 3 ? 3 : 0] = v.w;
        d[NW > 4 ? 4 : 0] = u.x; d[NW > 5 ? 5 : 0] = u.y; d[NW > 6 ? 6 : 0] = u.z; d[NW > 7 ? 7 : 0] = u.w;
    } else if (NW == 4) {
        const uint4 v = *reinterpret_cast<const uint4*>(p);
        d[0] = v.x; d[1] = v.y; d[NW > 2 ? 2 : 0] = v.z; d[NW > 3 ? 3 : 0] = v.w;
    } else if (NW == 2) {
        const uint2 v = *reinterpret_cast<const uint2*>(p);
        d[0] = v.x; d[1] = v.y;
    } else {
        d[0] = *p;
    }
}
// the lane's weights (time-weighted likelihood): 4 B per fixture, the bulk of a weighted tile
template <bool WEIGHTED>
__device__ __forceinline__ void load_lane_weights(const EvalArgs& A, size_t o, LaneData& L) {
    if (WEIGHTED) {
#pragma unroll
        for (int q = 0; q < LANE_FIX / 4; ++q) {
            const float4 v = *reinterpret_cast<const float4*>(A.w + o * LANE_FIX + 4 * q);
            L.wj[4 * q] = v.x; L.wj[4 * q + 1] = v.y; L.wj[4 * q + 2] = v.z; L.wj[4 * q + 3] = v.w;
        }
    } else {
#pragma unroll
        for (int j = 0; j < LANE_FIX; ++j) L.wj[j] = 1.0f;
    }
}
// indices and goals from bare pointers (dc_eval's first tile: pointers that arrive in preloaded SGPRs)
__device__ __forceinline__ LaneData load_lane_p(const uint32_t* ph, const uint32_t* pa, const uint32_t* px,
                                                const uint32_t* py, size_t o) {
    LaneData L;
    L.hw[0] = ph[o];
    L.aw[0] = pa[o];
    load_words<XWORDS>(px + o * XWORDS, L.xw);
    load_words<XWORDS>(py + o * XWORDS, L.yw);
    return L;
}
// indices and goals only (WITH_WEIGHTS = false: the caller requests the weights later)
template <bool WEIGHTED, bool WITH_WEIGHTS = true>
__device__ __forceinline__ LaneData load_lane(const EvalArgs& A, size_t o /* tile*64 + lane */) {
    LaneData L;
    L.hw[0] = A.h[o];
    L.aw[0] = A.a[o];
    load_words<XWORDS>(A.x + o * XWORDS, L.xw);
    load_words<XWORDS>(A.y + o * XWORDS, L.yw);
    if (WITH_WEIGHTS) load_lane_weights<WEIGHTED>(A, o, L);
    return L;
}

struct LaneOut {
    uint32_t key;           // (home | away<<16) the lane's pending run sums belong to
    float rsh, rsa;         // pending run sums: -(dL/d eta_h), -(dL/d eta_a) w/o goal counts
    double slam;            // float64: see lane_uniform
    float slog, su, sclip;
};

__device__ __forceinline__ float clamp0(float t) {  // max(t, 0) in one instruction
    return __builtin_amdgcn_fmed3f(t, 0.0f, __builtin_inff());
}
// 0x80 in every byte of v that is zero (exact, no cross-byte carries)
__device__ __forceinline__ uint32_t zero_bytes(uint32_t v) {
    return ~(((v & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | v | 0x7F7F7F7Fu);
}
// Score classes of four fixtures at once (bpl/_util.py:58-91 tests goals == 0 / == 1): bit 7 of
// byte j of low7 is set when fixture j is a low score (both goals <= 1); x7 / y7 when in addition
// the home / away side scored exactly one.
struct ScoreMasks {
    uint32_t low7, x7, y7;
};
__device__ __forceinline__ ScoreMasks score_masks(uint32_t x, uint32_t y) {
    const uint32_t u = ((x | y) >> 1) & 0x7F7F7F7Fu;          // byte >> 1: zero <=> both goals <= 1
    ScoreMasks m;
    m.low7 = ~(u + 0x7F7F7F7Fu) & 0x80808080u;                // (no carry across bytes: u <= 0x7F)
    m.x7 = (x << 7) & m.low7;                                 // bit 0 of byte j -> bit 7 of byte j
    m.y7 = (y << 7) & m.low7;
    return m;
}
// tau argument 1 + rho*c of one score class: log2 of its clip at 0, and dlogtau/drho
// (round 3) t = 1 + rho c is rounded to float32 -- 6e-8 absolute on a number near 1, and the same
// error for every fixture of a (pair, class): at N = 1e6 that was the LARGEST term of
// |U - U_float64| left (7e-3; tools: the emulation in profiles/r03/parity_errors.txt).  The rounding
// is known exactly: 1 - t is exact near 1, so one more fma gives r = (1 + rho c) - t, and
// log2(1 + rho c) = log2 t + r / t / ln 2 to first order -- three instructions per class and lane.
// (round 4) ILL-CONDITIONED CLASSES.  Near a bound of rho one of the arguments goes to 0 (tol = 0,
// bpl/_util.py:42,58-85): log t and c / t amplify the float32 roundings of rho and of the rates by
// 1 / t and 1 / t^2 -- 1e-6 from a bound the value was off by O(1) and the gradient, a difference of
// two terms of 1e6 (the direct one and the bounds' adjoint), by more than its size.  A class whose
// float32 argument is below TAU_ILL contributes NOTHING here; the tail workgroup adds it in float64 from
// the exact rates (ill_pass: a data-only table holds every pair's class weights).  The decision is taken
// on the same float32 bits there (same tables, same rho_f32, this same expression) and, being
// monotone in c, is known for ALL pairs from the three float32 maxima: ZO_ILL.  Also covers t <= 0
// (the float64 side returns the -inf of the reference's log(clip(., 0))).
__device__ __forceinline__ void class_terms(float rho, float c, float* l2, float* u) {
    const float t = class_arg(rho, c);
    const float it = __builtin_amdgcn_rcpf(t);
    const float r = fmaf(rho, c, 1.0f - t);
    const bool ok = t >= TAU_ILL;   // (false for NaN as well: the Poisson part carries it)
    *l2 = ok ? fmaf(r * it, 1.44269504088896341f, __log2f(t)) : 0.0f;
    *u = ok ? c * it : 0.0f;
}

// All fixtures of a lane are one pair (h, a): the library pads every pair's run to a multiple
// of LANE_FIX with NULL fixtures -- same pair, goals (255, 255), weight 0 -- so a lane never
// straddles a pair boundary and the rates and tau terms are computed once per lane.
// EXACT (dc_eval past 64 teams, round 4): the rate sum takes the EXACT products of the float32 table entries
// (exact in float64) instead of their float32 roundings.  Up to 64 teams the prior part's pair walk books that
// rounding per pair (prior_body, ZO_PAIRC); a complete pair table past 64 teams is not walked at all, and the
// correction was simply missing there (1.2e-2 of U = 9e6 at 91 teams, uniform(-2, 2): as large as the whole
// tolerance).  Four more float64 instructions per lane and tile.
template <bool WEIGHTED, bool CLIP, bool EXACT = false>
__device__ __forceinline__ LaneOut lane_uniform(const LaneData& Ld, float rho,
                                                const float2* tabH, const float2* tabA) {
    const uint32_t h = Ld.hw[0] & 0xFFFFu, a = Ld.aw[0] & 0xFFFFu;
    const float2 th = tabH[h], ta = tabA[a];
    float lh = th.x * ta.y;  // exp(att[h] + ha[h]) * exp(-def[a])
    float la = ta.x * th.y;  // exp(att[a]) * exp(-def[h])
    bool ch = false, ca = false;
    if (CLIP) {
        ch = lh > (float)RATE_CLIP;
        ca = la > (float)RATE_CLIP;
        lh = ch ? (float)RATE_CLIP : lh;
        la = ca ? (float)RATE_CLIP : la;
    }
    // tau (bpl/_util.py:58-91): c = -lh*la (0,0) | +la (1,0) | +lh (0,1) | -1 (1,1)
    float l00, u00, l10, u10, l01, u01, l11, u11;
    class_terms(rho, -lh * la, &l00, &u00);
    class_terms(rho, la, &l10, &u10);
    class_terms(rho, lh, &l01, &u01);
    class_terms(rho, -1.0f, &l11, &u11);
    float n00, n10, n01, n11, nall, sx, sy;  // (weighted) class counts, total, goal sums
    if (!WEIGHTED) {
        int c_low = 0, c_x1 = 0, c_y1 = 0, c11 = 0, nz = 0;
        uint32_t ax = 0, ay = 0;
#pragma unroll
        for (int q = 0; q < XWORDS; ++q) {
            const uint32_t x = Ld.xw[q], y = Ld.yw[q];
            const ScoreMasks m = score_masks(x, y);
            c_low += __popc(m.low7);
            c_x1 += __popc(m.x7);
            c_y1 += __popc(m.y7);
            c11 += __popc(m.x7 & m.y7);
            if (CLIP) {
                ax = __builtin_amdgcn_sad_u8(x, 0u, ax);
                ay = __builtin_amdgcn_sad_u8(y, 0u, ay);
            }
        }
        // null fixtures (goals (255, 255), never a low score) pad the end of a pair's run; the
        // lane's number of REAL fixtures rides in the second home-index halfword, which nothing
        // else reads (all fixtures of a lane share one pair)
        nz = LANE_FIX - (int)(Ld.hw[0] >> 16);
        n00 = (float)(c_low - c_x1 - c_y1 + c11);
        n10 = (float)(c_x1 - c11);
        n01 = (float)(c_y1 - c11);
        n11 = (float)c11;
        nall = (float)(LANE_FIX - nz);
        sx = (float)((int)ax - 255 * nz);
        sy = (float)((int)ay - 255 * nz);
    } else {
        // weights are streamed (4 B per fixture); the class of every fixture comes from the same
        // SWAR masks, turned into 0 / 128 factors by v_cvt_f32_ubyteN: two instructions per
        // (fixture, class) instead of a compare-select chain per fixture
        n00 = n10 = n01 = n11 = nall = sx = sy = 0.f;
#pragma unroll
        for (int q = 0; q < XWORDS; ++q) {
            const uint32_t x = Ld.xw[q], y = Ld.yw[q];
            const ScoreMasks m = score_masks(x, y);
            const uint32_t m11 = m.x7 & m.y7, m10 = m.x7 & ~m.y7, m01 = m.y7 & ~m.x7,
                           m00 = m.low7 & ~(m.x7 | m.y7);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float wv = Ld.wj[4 * q + j];
                n00 = fmaf(wv, (float)((m00 >> (8 * j)) & 0xFFu), n00);
                n10 = fmaf(wv, (float)((m10 >> (8 * j)) & 0xFFu), n10);
                n01 = fmaf(wv, (float)((m01 >> (8 * j)) & 0xFFu), n01);
                n11 = fmaf(wv, (float)((m11 >> (8 * j)) & 0xFFu), n11);
                nall += wv;
                if (CLIP) {
                    sx = fmaf(wv, (float)((x >> (8 * j)) & 0xFFu), sx);
                    sy = fmaf(wv, (float)((y >> (8 * j)) & 0xFFu), sy);
                }
            }
        }
        n00 *= 1.0f / 128.0f;  // (exact: the factors are 0 or 2^7)
        n10 *= 1.0f / 128.0f;
        n01 *= 1.0f / 128.0f;
        n11 *= 1.0f / 128.0f;
    }
    LaneOut o;
    o.key = h | (a << 16);
    // (round 3) float64: nall (lh + la) in float32 is rounded the same way in EVERY lane of a pair
    // (82 lanes per pair at N = 1e6) -- with the two corrections of prior_body / class_terms in, the
    // largest term of |U - U_float64| left (6e-3 at N = 1e6).  Five float64 instructions per lane.
    if (EXACT)
        o.slam = (double)nall * ((ch ? RATE_CLIP : (double)th.x * (double)ta.y) + (ca ? RATE_CLIP : (double)ta.x * (double)th.y));
    else
        o.slam = (double)nall * ((double)lh + (double)la);
    // a class with no fixture must not contribute (its log may be -inf): 0 * -inf = NaN
    o.slog = (n00 != 0.f ? n00 * l00 : 0.f) + (n10 != 0.f ? n10 * l10 : 0.f) +
             (n01 != 0.f ? n01 * l01 : 0.f) + (n11 != 0.f ? n11 * l11 : 0.f);
    o.su = n00 * u00 + n10 * u10 + n01 * u01 + n11 * u11;
    // -(dL/d eta) without the data-only goal counts (added in the tail):
    //   eta_h: lh - rho*u*[x==0]   (lh * dlogtau/dlh = rho*u for (0,0),(0,1))
    o.rsh = nall * lh - rho * (n00 * u00 + n01 * u01);
    o.rsa = nall * la - rho * (n00 * u00 + n10 * u10);
    o.sclip = 0.f;
    if (CLIP) {
        // clipped rate: d/d eta = 0 -> cancel the goal count added later.  (The value's correction
        // k*eta -> k*log(15) is the prior workgroup's, in float64 and per pair: prior_body, ZO_PAIRC.)
        if (ch) o.rsh = sx;
        if (ca) o.rsa = sy;
    }
    return o;
}

// ------------------------------------------------------------------------- dc_eval

template <class T>
__device__ __forceinline__ T* as_global(T* p) {
    return (T*)(__attribute__((address_space(1))) T*)p;
}
// A fresh copy of the kernel's (single, by-value) argument from the kernarg segment.  The
// pointer is made opaque so the scalar loads stay where the copy is taken.
// OFF: bytes of kernel arguments in front of the EvalArgs block (dc_eval: the preloaded ones, EVAL_PRE_BYTES)
template <int OFF = 0>
__device__ __forceinline__ EvalArgs reload_args() {
    typedef const __attribute__((address_space(4))) uint32_t* kptr;
    kptr src = (kptr)__builtin_amdgcn_kernarg_segment_ptr() + OFF / 4;
    asm volatile("" : "+s"(src));
    constexpr int NW = (int)(sizeof(EvalArgs) / 4);
    static_assert(sizeof(EvalArgs) % 4 == 0, "EvalArgs is copied word by word");
    uint32_t tmp[NW];
#pragma unroll
    for (int i = 0; i < NW; ++i) tmp[i] = src[i];
    EvalArgs B;
    __builtin_memcpy(&B, tmp, sizeof B);
    // the copied pointers have lost their address space (flat loads): they are global memory
#define DC_GLOBAL(f) B.f = as_global(B.f)
    DC_GLOBAL(pairs); DC_GLOBAL(pairw); DC_GLOBAL(xs); DC_GLOBAL(xsf); DC_GLOBAL(cA); DC_GLOBAL(cD); DC_GLOBAL(cH);
    DC_GLOBAL(hbuf); DC_GLOBAL(tickets); DC_GLOBAL(gacc); DC_GLOBAL(z); DC_GLOBAL(potential);
    DC_GLOBAL(grad); DC_GLOBAL(aux); DC_GLOBAL(nuts); DC_GLOBAL(debug);
#undef DC_GLOBAL
    return B;
}

#ifdef DC_MIN_WAVES  // waves per SIMD the register allocation must allow (2 workgroups per CU = 4)
#define DC_LAUNCH_BOUNDS __launch_bounds__(BLOCK, DC_MIN_WAVES)
#else
#define DC_LAUNCH_BOUNDS __launch_bounds__(BLOCK)
#endif
// Round 4: KERNEL-ARGUMENT PRELOAD.  Everything the first loads of a workgroup need -- the position, the
// four fixture arrays, the sizes that locate its tile and the latent layout (a function of the team and
// covariate counts) -- comes as 14 dwords of plain leading arguments, which gfx950 delivers in SGPRs at wave
// launch (-mllvm -amdgpu-kernarg-preload-count=14, bpl-next_amd/csrc/Makefile): the position's and the tile's
// loads issue at once instead of behind the scalar loads of the 400-byte EvalArgs block (a miss every launch:
// the kernarg segment is fresh), i.e. one dependent memory latency less at the head of EVERY workgroup.
// (The block itself still follows; reload_args<EVAL_PRE_BYTES>() finds it behind these.)
constexpr int EVAL_PRE_BYTES = 56;   // 5 pointers + 4 ints
template <bool WEIGHTED, bool CLIP, bool STAGED, bool NUTS>
__global__ DC_LAUNCH_BOUNDS void dc_eval(const double* pz, const uint32_t* ph, const uint32_t* pa, const uint32_t* px,
                                         const uint32_t* py, int tk /* T | K << 16 */, int p_tiles,
                                         int tw_aw /* tiles per wave | active waves << 16 */, int p_zstride, EvalArgs A) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const Layout L = make_layout(CLIP ? MODEL_EXTENDED : MODEL_BASIC, tk & 0xFFFF, tk >> 16);
    const int T = L.T, T1 = T + 1;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int chain = blockIdx.y;
    const double* z = pz + (size_t)chain * p_zstride;
#ifdef DC_STAMPS
    const unsigned long long t_entry = __builtin_amdgcn_s_memrealtime();
    const unsigned long long c_entry = __builtin_amdgcn_s_memtime();
#endif
    // device-resident NUTS: the subtree this launch belonged to may already be complete
    const double nuts_done = NUTS ? nuts_of(A, chain)[nd::H_S_DONE] : 0.0;
#ifdef DC_STAMPS  // (launches that return at once leave the last real launch's record alone)
    if (!(NUTS && nuts_done != 0.0) && threadIdx.x == 0 && A.debug && blockIdx.y == 0)
        A.debug[(size_t)blockIdx.x * 16] = t_entry;
#endif

    // LDS carve of the streaming part (all offsets multiples of 16 B)
    float2* tabH = reinterpret_cast<float2*>(smem);            // {exp(att+ha), exp(-def)}
    float2* tabA = tabH + tab_len(T);                          // {exp(att),    exp(-def)}
    double* acc = reinterpret_cast<double*>(tabA + tab_len(T));  // [3*T1]
    double* red = acc + 3 * T1 + ((3 * T1) & 1);               // [WAVES*4]
    float* redm = reinterpret_cast<float*>(red + WAVES * N_SCAL);  // [WAVES*4]

    if (blockIdx.x == 0) {
        // ---- the prior workgroup: z-only work beside the streaming, then it WAITS for the
        // streaming workgroups' arrivals and runs the tail.  (It used to take a ticket like the
        // others and hand its record over through memory; the last arriver -- two dependent
        // returning atomics later -- ran the tail: ~0.5 us more on the critical path, and this
        // workgroup's own drain + ticket, 1.1 us, made it the last arriver.)
        if (NUTS && nuts_done != 0.0) return;
        const size_t tail_bytes = (acc_tail_lds_bytes(T, L.D, L.K, A.zo_stride, STAGED && NUTS) + 15) & ~(size_t)15;
        TailPre pre;
        // (the position and the layout as the preloaded arguments give them: the prior part's first loads do
        // not wait for the argument block either)
        EvalArgs Ap = A;
        Ap.L = L;
        Ap.z = pz;
        Ap.z_stride = p_zstride;
        if (STAGED) {
            tail_preload_static<false, false>(Ap, chain, pre);
            prior_body<CLIP, true, false>(Ap, chain, smem + tail_bytes, reinterpret_cast<double*>(smem));
        } else {
            if (tid == 0) *acc_tail_flag(Ap, smem) = 1;   // (the prior part's barriers lie in between)
            auto entry_loads = [&]() { tail_preload_static<false, true>(Ap, chain, pre); };
            prior_body<CLIP, true, true, false, decltype(entry_loads)>(
                Ap, chain, smem + tail_bytes, reinterpret_cast<double*>(smem), nullptr, nullptr, nullptr, entry_loads);
            // (Polled from inside the prior part instead -- behind the barrier of the bounds' top-two records, 0.7 us
            // earlier -- the first poll comes back before the last row adds are visible and the second one a round
            // trip later than this one: 10.8 us instead of 9.7 at 200 teams.  profiles/r04/teams_rework.txt)
            tail_poll_early(Ap, chain, smem, pre, acc_tail_flag(Ap, smem));
        }
        DC_STAMP(4);
        // The tail reads its arguments from the kernarg segment again (scalar loads behind an
        // opaque pointer): kept live in SGPRs from the kernel entry they were spilled.
        // (Twice: once for the preload, once more after the wait -- the second copy comes from
        // the scalar cache, and nothing is held in SGPRs across the polling loop.)
        nd::LeafState<1> leaf1{};            // D <= 64: the leaf's vectors in registers (waves 4, 5)
        double bigv[nd::LEAF_STAGE_LOADS];   // D > 64: waves 4..7 stage one 64-element slice each
        tail_preload<STAGED, NUTS>(reload_args<EVAL_PRE_BYTES>(), chain, pre, leaf1, bigv, true);
        if (STAGED && tid == 0) *acc_tail_flag(A, smem) = 1;
        __syncthreads();
        DC_STAMP(5);
        const EvalArgs B = reload_args<EVAL_PRE_BYTES>();
        // (tail_acc waits for the rows itself: every thread polls the row it will read; larger leagues:
        // tail_poll_early has)
        if (!tail_acc<STAGED, NUTS, CLIP, false, !STAGED>(B, chain, smem, pre, leaf1, bigv, acc_tail_flag(B, smem))) {
            // the streaming workgroups never arrived (bounded wait): poison the outputs
            double* grad = grad_of(A, chain);
            for (int i = tid; i < L.D; i += BLOCK) grad[i] = __builtin_nan("");
            if (tid == 0) {
                *pot_of(A, chain) = __builtin_nan("");
                raise_fault(A.fault, FAULT_EVAL_ARRIVALS);
            }
        }
        return;
    }
    {
        const int wgi = blockIdx.x - 1;
        // ---- 0. every load the prologue needs, in the order the data is wanted: this thread's pair
        // and its entries of z (L2 hits) first, then the first tile (HBM), then the slab slots.
        // Vector loads return in order, so the tables must not queue behind the tile; and the loads
        // are unconditional (indices clamped) wherever possible: behind a branch the compiler can
        // no longer count how many younger loads may stay outstanding and waits for all of them,
        // and a load that a branch also zeroes is waited for on the spot.
        const TeamZ tz0 = load_team_z<CLIP>(L, z, min(tid, T - 1));
        const F32Scalars fs = f32_scalars<CLIP>(L, z);
        // (basic model: every read of z is above, so a compiler barrier can pin them in front of
        // the tile loads -- without it the per-team loads are sunk to their use, behind the tile.
        // With covariates z is read again in the table loop, and behind a barrier those reads
        // would stop being scalar loads.)
        if (!CLIP) asm volatile("" ::: "memory");
        const int p_tpw = tw_aw & 0xFFFF, p_aw = tw_aw >> 16;
        const int gw = wgi * p_aw + wave;
        int tile = wave < p_aw ? gw * p_tpw : p_tiles;
        const int tile_end = min(tile + p_tpw, p_tiles);
        // (a wave without tiles loads the last one and never uses it)
        // (the weights of a time-weighted tile -- 8 KB per wave -- are requested AFTER the table
        // loads: vector loads return in order, and behind them the covariate rows of the tables
        // waited 1.7 us for HBM)
        const size_t lane0 = (size_t)min(tile, p_tiles - 1) * 64 + lane;
        LaneData cur = load_lane_p(ph, pa, px, py, lane0);
        // (what follows reads the argument block: the loads above do not wait for it)
        uint32_t pr0 = 0;
        if (tid < A.P) pr0 = A.pairs[tid];
        const int o0 = A.wg_off[wgi], o1 = A.wg_off[wgi + 1];  // static sparse-slab slots
        // (unconditional, index clamped: a load into a register that a branch also zeroes made the
        // compiler wait for it -- and, in order, for the whole tile -- right here)
        const int kq = min(o0 + tid, A.total_c - 1);
        const int slot0 = A.wg_slots[kq];
        // (uniform over the whole grid.  Checking this after the table loads are in flight
        // measured no gain: the flag's latency is not what delays z, z itself is cold.)
        if (NUTS && nuts_done != 0.0) return;

        // ---- 1. per-team tables (float32) + zero accumulators
        build_tables_f32<CLIP, false, STAGED>(L, z, A.xsf, tabH, tabA, tid, tz0, fs);
        for (int i = tid; i < 3 * T1; i += BLOCK) acc[i] = 0.0;
        __syncthreads();
        DC_STAMP(1);

        // ---- 2. rho bounds over the unique-pair table (bpl/_util.py:23-30): values only
        float mP, mQ, mR;
        if (!(!STAGED && A.dense_pairs && dense_maxima_f32<CLIP>(T, tabH, tabA, redm, tid, &mP, &mQ, &mR)))
            pair_maxima_f32<CLIP>(A, tabH, tabA, pr0, redm, tid, &mP, &mQ, &mR);
        const float rho = rho_f32(mP, mQ, mR, fs.q);
        DC_STAMP(2);
        // (the weights are requested only here: while they were in flight across the two barriers
        // above, the tables and the bounds were ready 1.4 us later)
        if (WEIGHTED) asm volatile("" ::: "memory");
        load_lane_weights<WEIGHTED>(A, lane0, cur);

#ifdef DC_STAMPS
        float rho_dbg = 0.f;
#endif
        // ---- 3. stream the fixtures
        double dSLAM = 0.0, dSLOG = 0.0, dSU = 0.0, dCLIP = 0.0;  // per lane
        // One lane = LANE_FIX consecutive fixtures of ONE (home, away) pair (runs are padded with
        // null fixtures): the two rates and the four score-class tau terms are computed once for
        // the lane and each fixture is only classified.
        auto process = [&](const LaneData& ld) {
#ifdef DC_STAMPS
            if (ld.hw[0] == 0xFFFFFFFFu) rho_dbg += 1.0f;  // forces the loads to have landed
            DC_STAMP(12);
#endif
            const LaneOut lo = lane_uniform<WEIGHTED, CLIP, !STAGED>(ld, rho, tabH, tabA);
            // scalars: float32 over the lane's fixtures only, fixed point (q30) from there on
            // (a float32 sum over the whole wave-tile would cost ~1e-4 absolute in U)
            dSLAM += rint(ldexp(lo.slam, 30));
            dSLOG += q30(lo.slog);
            dSU += q30(lo.su);
            if (CLIP) dCLIP += q30(lo.sclip);
            float rsh = lo.rsh, rsa = lo.rsa;
            const uint32_t key = lo.key;
#ifdef DC_STAMPS
            if (rsh == 12345.678f) rho_dbg += 1.0f;
            DC_STAMP(13);
#endif

            // ---- per-(home,away) run sums: lane -> wave -> LDS per-team accumulators
            const uint32_t kprev = prev_lane_u32(key, ~key);
            const unsigned long long heads = __ballot(kprev != key);  // lane 0 always a head
            const int nruns = __popcll(heads);
            if (nruns == 1) {  // whole wave-tile on one pair (the common case: sorted)
                wave_sum2_f32(rsh, rsa);
                if (lane == 0) flush_run_q(acc, T1, key, rsh, rsa);
            } else if (nruns <= RUN_LOOP_MAX) {  // a few runs: one masked DPP sum per run
                unsigned long long hd = heads;
                while (hd) {
                    const int first = __ffsll((long long)hd) - 1;
                    hd &= hd - 1;
                    const int stop = hd ? __ffsll((long long)hd) - 1 : 64;
                    const bool in = lane >= first && lane < stop;
                    const uint32_t kk = (uint32_t)__builtin_amdgcn_readlane((int)key, first);
                    float s0 = in ? rsh : 0.f, s1 = in ? rsa : 0.f;
                    wave_sum2_f32(s0, s1);
                    if (lane == 0) flush_run_q(acc, T1, kk, s0, s1);
                }
            } else {  // many short runs: every lane adds its own sums (LDS atomics)
                flush_run_q(acc, T1, key, rsh, rsa);
            }
        };
        if (tile < tile_end) {  // (wave uniform; a wave without tiles skips the lot)
            // every tile but the last is processed with the NEXT one in flight ...
            while (tile + 1 < tile_end) {
                LaneData nxt = load_lane<WEIGHTED>(A, (size_t)(tile + 1) * 64 + lane);
                process(cur);
                // The prefetched words must not be touched before this point: without the opaque
                // redefinition the compiler hoists the next tile's first use (index masking) up to
                // the loads, which turns the prefetch into a stall (measured: +20 % at N = 1e8).
                asm volatile("" : "+v"(nxt.hw[0]), "+v"(nxt.aw[0]), "+v"(nxt.xw[0]), "+v"(nxt.yw[0]));
                cur = nxt;
                ++tile;
            }
            // ... and the last one (the ONLY one up to ~1e6 fixtures) with nothing in flight: the
            // unconditional prefetch this replaces made a single-tile wave wait for a reload of its
            // own tile, a whole memory round trip at the end of its critical path
            process(cur);
        }
        DC_STAMP(3);
#ifdef DC_STAMPS
        if (rho_dbg == 77.f) dSLAM += 1.0;
#endif

        // ---- 4. workgroup reduction of the scalars, then the slab (write-through)
        {   // (four chains in one written-out reduction: one after the other they were 0.3 us)
            double sv[4] = {dSLAM, dSLOG, dSU, CLIP ? dCLIP : 0.0};
            wave_sum4_f64(sv);
            dSLAM = sv[0]; dSLOG = sv[1]; dSU = sv[2]; dCLIP = sv[3];
        }
        if (lane == 0) {
            red[wave * N_SCAL + 0] = dSLAM;
            red[wave * N_SCAL + 1] = dSLOG;
            red[wave * N_SCAL + 2] = dSU;
            red[wave * N_SCAL + 3] = dCLIP;
        }
        DC_STAMP(15);
        __syncthreads();
        DC_STAMP(11);
        {   // add the slots this workgroup's fixtures touch (static list) and its four scalars
            // into the chain's counted accumulator rows (integer atomics at agent scope, no return):
            // the count travels with the value, so nothing follows -- no drain, no barrier, no
            // arrival counter; the workgroup is done
            long long* ga = A.gacc + (size_t)chain * 2 * ga_set_words(T);
            for (int k = o0 + tid; k < o1; k += BLOCK) {
                const int slot = k == o0 + tid ? slot0 : A.wg_slots[k];
                const int which = (slot >= T) + (slot >= 2 * T);   // (0 <= slot < 3T; acc is [3][T + 1]: no division)
                ga_add(ga + (size_t)slot * GA_ROW, acc[slot + which]);
            }
            if (tid >= BLOCK - N_SCAL) {  // (the last wave: the slot lanes are the first ones)
                const int k = tid - (BLOCK - N_SCAL);
                double sv = 0.0;
#pragma unroll
                for (int wv = 0; wv < WAVES; ++wv) sv += red[wv * N_SCAL + k];
                ga_add(ga + (size_t)(3 * T + (wgi % GA_SHARDS) * N_SCAL + k) * GA_ROW, sv);
            }
        }
        DC_STAMP(4);
        DC_STAMP(5);
        DC_STAMP(6);
    }
#ifdef DC_STAMPS  // shader clock of this workgroup's life: cycles (s_memtime) per 10 ns tick
    if (threadIdx.x == 0 && A.debug && blockIdx.y == 0)
        A.debug[(size_t)blockIdx.x * 16 + 9] = __builtin_amdgcn_s_memtime() - c_entry;
#endif
}

// ------------------------------------------------------------------- dc_eval_loop
// PERSISTENT evaluation kernel of a device-resident chain (one chain, <= 64 teams, the NUTS
// leaf in the tail): the launch stays resident for up to `persist_steps` leapfrogs instead of
// one launch per leapfrog.  What a dependent launch pays every time -- the dispatch ramp of the
// grid (0.7-1.2 us first to last workgroup), the kernel boundary, the kernel-argument and
// position loads from cold caches, the first tile from HBM/MALL (1.6 us after entry) -- is paid
// once: a wave's tile of fixtures stays in its REGISTERS for the whole launch, and the next
// position travels as data-tagged granules {float32 z_i, tag} that the leaf's wave stores the
// moment it knows it (no flag, no drain, no barrier on the way):
//   tail workgroup (block 0)   prior -> wait for the arrivals -> tail + leaf (+ chain advance);
//        the leaf wave publishes the granules of the next step
//   streaming workgroups       every thread polls the granules it needs (L1-bypassing loads,
//        BOUNDED), tables -> rho -> the resident tile -> accumulator rows -> arrive
// Every workgroup invalidates its scalar cache at the top of a step (the chain state is
// rewritten by vector stores between steps; uniform-address loads go through that cache).
// The grid must be co-resident (host: one workgroup per CU); a finished chain ends the launch
// (FIN granules); every spin is bounded and a timeout ends the launch the same way.
constexpr int GRANULE_SPIN_LIMIT = 1 << 21;

// this thread's granules of step `want`: a lane's team entries + the scalar sites
struct ZPolled {
    float a, d, h;            // team entries (decentered / standardised sites)
    float sa, sd, md, corr;   // std_attack, std_defence, mean_defence, corr_coef_raw
    float e0, e1;             // basic: home_advantage, -; extended: std_home_advantage, mean_home_advantage
    float coef;               // extended with covariates: lane k < K holds attack coefficient k, lane K + k defence k
    int state;                // 1 ok, 0 the chain finished, -1 timed out
};
// covariate coefficients that travel with the lanes' own granules (one per lane, fetched in the SAME
// round of loads and handed round by readlane): polled one after the other inside the table build they
// were 2K dependent round trips per leapfrog
constexpr int LOOP_COEF_MAX_K = 32;
template <bool EXT>
__device__ __forceinline__ ZPolled poll_z(const unsigned long long* zg, const Layout& L, int t,
                                          unsigned int want, unsigned int fin_tag, int lane) {
    const int K = EXT ? L.K : 0;
    const int o_coef = K > 0 && K <= LOOP_COEF_MAX_K ? (lane < K ? L.o_bA + lane : lane < 2 * K ? L.o_bD + lane - K : L.o_bA)
                                                    : L.o_sa;
    constexpr int NG = EXT ? 10 : 9;   // (the basic model has no coefficients: its batch stays at nine)
    const int idx[10] = {(EXT ? L.o_sat : L.o_adec) + t, (EXT ? L.o_sdt : L.o_ddec) + t,
                         EXT ? L.o_hadec + t : L.o_sa, L.o_sa, L.o_sd, L.o_md, L.o_corr,
                         EXT ? L.o_sh : L.o_ha, EXT ? L.o_mha : L.o_sa, o_coef};
    unsigned long long v[10] = {};
    ZPolled r;
    r.state = -1;
    for (int spin = 0; spin < GRANULE_SPIN_LIMIT; ++spin) {
#pragma unroll
        for (int k = 0; k < NG; ++k) v[k] = ld_granule(&zg[idx[k]]);
        bool ok = true, fin = false;
#pragma unroll
        for (int k = 0; k < NG; ++k) {
            const unsigned int tg = (unsigned int)(v[k] >> 32);
            ok = ok && tg == want;
            fin = fin || tg == fin_tag;
        }
        if (__ballot(fin) != 0ull) { r.state = 0; break; }
        if (__ballot(!ok) == 0ull) { r.state = 1; break; }
        __builtin_amdgcn_s_sleep(1);
    }
    auto f = [&](int k) { return __uint_as_float((unsigned int)v[k]); };
    r.a = f(0); r.d = f(1); r.h = f(2); r.sa = f(3); r.sd = f(4); r.md = f(5); r.corr = f(6);
    r.e0 = f(7); r.e1 = f(8);
    r.coef = f(9);
    return r;
}
// one more granule (covariate coefficients), same protocol
__device__ __forceinline__ float poll_one(const unsigned long long* g, unsigned int want, unsigned int fin_tag,
                                          int* state) {
    unsigned long long v = 0;
    for (int spin = 0; spin < GRANULE_SPIN_LIMIT; ++spin) {
        v = ld_granule(g);
        const unsigned int tg = (unsigned int)(v >> 32);
        if (tg == want) break;
        if (tg == fin_tag) { *state = 0; break; }
        if (spin + 1 == GRANULE_SPIN_LIMIT) *state = -1;
        __builtin_amdgcn_s_sleep(1);
    }
    return __uint_as_float((unsigned int)v);
}

// LNE: see tail_waves (the host takes 2 when 64 < D <= 128)
template <bool WEIGHTED, bool CLIP, int LNE = 1>
__global__ DC_LAUNCH_BOUNDS void dc_eval_loop(EvalArgs A) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr bool STAGED = true, NUTS = true;
    const Layout& L = A.L;
    const int T = L.T, T1 = T + 1;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int chain = blockIdx.y;
    if (nuts_of(A, chain)[nd::H_S_DONE] != 0.0) return;  // the chain had finished before this launch
    const int steps = A.persist_steps;
    unsigned long long* zg = A.zg + (size_t)chain * L.D;
    const unsigned int fin_tag = A.tag_base + 1u + (unsigned int)steps;

    if (blockIdx.x == 0) {
        const size_t tail_bytes = (acc_tail_lds_bytes(T, L.D, L.K, A.zo_stride, true) + 15) & ~(size_t)15;
        // step 0's position is where the previous launch (or the chain's start) left it
        if (wave == LEAF_WAVE) publish_z(zg, nd::vec(nuts_of(A, chain), L.D, nd::V_ZN), L.D, lane, A.tag_base + 1u);
        // ... and from here on the tail keeps the position in LDS (zL of tail_acc's layout): the leaf
        // writes the next one there, the prior part and the epilogue read it from there
        // ... together with what never changes during the launch: the static per-team sums and (when
        // they fit) the covariates
        double* cL = reinterpret_cast<double*>(smem) + A.zo_stride;
        double* zL = cL + 3 * T;
        const bool xs_staged = L.K > 0 && L.K <= 16;
        double* xsL = zL + L.D + (3 * T + N_SCAL + 4) + WAVES * 8;   // (tail_acc's layout)
        for (int i = tid; i < L.D; i += BLOCK) zL[i] = nd::vec(nuts_of(A, chain), L.D, nd::V_ZN)[i];
        for (int i = tid; i < 3 * T; i += BLOCK) cL[i] = i < T ? A.cA[i] : (i < 2 * T ? A.cD[i - T] : A.cH[i - 2 * T]);
        if (xs_staged)
            for (int i = tid; i < T * L.K; i += BLOCK) xsL[i] = A.xs[i];
        __syncthreads();
        // (round 4: the leaf is booked one step behind its evaluation -- see loop_tail)
        bool prev = false;   // the previous step's leaf is still to be booked
        double* gradL0 = acc_tail_gradL(A, smem);
        double* pkL = acc_tail_park(A, smem, true);
        double* pkR = pkL + leaf_park_doubles(L.D);
        double* flags = pkR + leaf_park_doubles(L.D);
        if (tid < LF_N) flags[tid] = 0.0;
        for (int s = 0; s < steps; ++s) {
            DC_STAMP(0);
            // (no scalar-cache invalidate here any more: what changes between steps is read through
            // vector loads -- the position from LDS, the leaf's state lane-indexed, the rows with sc1 --
            // and an invalidate made the kernel-argument reloads below miss, 0.3 us per step; the one
            // path that reads state with uniform addresses, the chain advance, invalidates for itself)
            if (tid == 0) *acc_tail_flag(A, smem) = 1;   // (in front of the prior part's first barrier)
            const bool last = s + 1 == steps;
            const unsigned int next_tag = last ? 0u : A.tag_base + 2u + (unsigned int)s;   // evaluation s + 1's
            double* gL_cur = gradL0 + (size_t)(s & 1) * (L.D + 8);
            prior_body<CLIP, true, false, true>(reload_args(), chain, smem + tail_bytes, reinterpret_cast<double*>(smem),
                                                zL, cL, xs_staged ? xsL : nullptr);
            DC_STAMP(4);
            TailPre pre;
            nd::LeafState<LNE> leaf1{};
            double bigv[nd::LEAF_STAGE_LOADS];
            // (a leaf that follows a booked-behind one takes its state from LDS, not from memory)
            tail_preload<STAGED, NUTS, true>(reload_args(), chain, pre, leaf1, bigv, false, !prev);
            __syncthreads();
            DC_STAMP(6);
            bool ok;
            {
                const EvalArgs B = reload_args();
                ok = loop_tail<CLIP, LNE>(B, chain, smem, leaf1, acc_tail_flag(B, smem), s & 1, prev);
            }
            __syncthreads();   // the outputs of evaluation s are in gL_cur; the previous leaf is booked
            if (!ok || *acc_tail_flag(A, smem) == 0) {  // a bounded wait expired: end the launch for everybody, poison the outputs
                if (wave == LEAF_WAVE) publish_fin(zg, L.D, lane, fin_tag);
                if (tid == 0) {
                    *pot_of(A, chain) = __builtin_nan("");
                    raise_fault(A.fault, FAULT_LOOP_ARRIVALS);
                }
                return;
            }
            DC_STAMP(5);
            bool complete = prev && flags[LF_SUBDONE] != 0.0;   // the previous leaf ended its subtree: evaluation s is void
            prev = false;
            if (!complete) {
                // ---- this step's leaf (parked by both leaf waves): does the subtree go on after it?
                if (wave == LEAF_WAVE) {
                    const double hv = pkL[lane];
                    const int num = (int)nd::hdr_word(hv, nd::H_S_NUM);
                    const bool spec = A.persist_spec && next_tag != 0u && num + 1 < (int)nd::hdr_word(hv, nd::H_S_MAX);
                    if (spec) {   // yes unless it turns or diverges: the next position goes out NOW
                        const double eps = nd::hdr_word(hv, nd::H_EPS) * nd::hdr_word(hv, nd::H_DIR);
#pragma unroll
                        for (int e = 0; e < LNE; ++e) {
                            const int i = lane + 64 * e;
                            const double* v = pkL + 64 + (size_t)e * LEAF_PARK_VECS * 64 + lane;
                            if (i < L.D) zL[i] = nd::leaf_next_position(v[64], v[0], v[128], eps, gL_cur[i]);
                        }
                    }
                    if (lane == 0) flags[LF_SPEC] = spec ? 1.0 : 0.0;
                    // (the granules go out behind the LDS stores, and this wave joins the barrier below WITHOUT
                    // waiting for them: __syncthreads() would hold the whole workgroup for the write-through
                    // stores' round trip, 0.7 us -- 6.88 -> 7.64 in the stamped step.  Its next full barrier is
                    // the prior part's first, by when they have landed.)
                    if (spec) publish_z(zg, zL, L.D, lane, next_tag);
                    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
                } else {
                    __syncthreads();   // (the next position and the decision are in LDS)
                }
                if (flags[LF_SPEC] != 0.0) {
                    prev = true;   // booked during the next step's wait and epilogue
                    continue;
                }
                // ---- the sequential path: this leaf completes its subtree by count (or the launch ends here)
                if (wave == LEAF_WAVE) {
                    nd::LeafState<LNE> S = leaf_unpark<LNE>(pkL, lane, false);
                    nd::leaf_prepare<false>(S);
                    const bool sub_done = nd::leaf_moves(nuts_of(A, chain), L.D, A.nuts_max_depth, lane, gL_cur, S, zL);
                    if (lane == 0) flags[LF_SUBDONE] = sub_done ? 1.0 : 0.0;
                    if (next_tag != 0u && !sub_done) publish_z(zg, zL, L.D, lane, next_tag);
                } else if (wave == RNG_WAVE) {
                    const nd::LeafState<LNE> S = leaf_unpark<LNE>(pkR, lane, true);
                    (void)nd::leaf_weights(nuts_of(A, chain), L.D, lane, gL_cur, S);
                }
                __syncthreads();
                complete = flags[LF_SUBDONE] != 0.0;
            }
            if (complete) {
                subtree_complete(reload_args(), chain, flags + LF_FIN, zL, next_tag);
                __syncthreads();
                if (flags[LF_FIN] != 0.0) return;  // the chain finished (FIN is out)
            }
            if (last && ld_sc1(&(nuts_of(A, chain) + A.persist->pd_off)[nd::P_ALLDONE]) != 0.0) return;
        }
        return;
    }

    // ---- streaming workgroups
    float2* tabH = reinterpret_cast<float2*>(smem);
    float2* tabA = tabH + tab_len(T);
    double* acc = reinterpret_cast<double*>(tabA + tab_len(T));
    double* red = acc + 3 * T1 + ((3 * T1) & 1);
    float* redm = reinterpret_cast<float*>(red + WAVES * N_SCAL);
    int* flag = reinterpret_cast<int*>(redm + WAVES * 4);
    const int wgi = blockIdx.x - 1;
    uint32_t pr0 = 0;
    if (tid < A.P) pr0 = A.pairs[tid];
    const int gw = wgi * A.active_waves + wave;
    const int tile0 = wave < A.active_waves ? gw * A.tiles_per_wave : A.n_tiles;
    const int tile_end = min(tile0 + A.tiles_per_wave, A.n_tiles);
    // the wave's (first) tile: loaded ONCE, resident in registers for every leapfrog of the launch
    const LaneData res = load_lane<WEIGHTED>(A, (size_t)min(tile0, A.n_tiles - 1) * 64 + lane);
    const int o0 = A.wg_off[wgi], o1 = A.wg_off[wgi + 1];
    const int kq = min(o0 + tid, A.total_c - 1);
    const int slot0 = A.wg_slots[kq];
    long long* ga0 = A.gacc + (size_t)chain * 2 * ga_set_words(T);
    if (tid == 0) *flag = 1;
    // this thread's row of the standardised covariates: data, resident for the whole launch
    constexpr int KREG = 8;
    const bool coef_fast = CLIP && L.K > 0 && L.K <= KREG;
    float xs_reg[KREG];
#pragma unroll
    for (int k = 0; k < KREG; ++k)
        xs_reg[k] = coef_fast && k < L.K ? A.xsf[(size_t)min(tid, T - 1) * L.K + k] : 0.f;

    for (int s = 0; s < steps; ++s) {
        const unsigned int want = A.tag_base + 1u + (unsigned int)s;
        // ---- this thread's granules: one round trip brings the data and the "go"
        ZPolled zp = poll_z<CLIP>(zg, L, min(tid, T - 1), want, fin_tag, lane);
        DC_STAMP(0);
        F32Scalars fs;
        fs.s_a = exp_f32(zp.sa);
        fs.s_d = exp_f32(zp.sd);
        fs.m = zp.md;
        {
            const float q = __builtin_amdgcn_rcpf(1.0f + exp_f32(-zp.corr));
            fs.q = fminf(fmaxf(q, (float)SIG_LO), (float)SIG_HI);
        }
        fs.gam = CLIP ? 0.f : zp.e0;
        fs.s_h = CLIP ? exp_f32(zp.e0) : 0.f;
        fs.mha = CLIP ? zp.e1 : 0.f;
        int state = zp.state;
        for (int t = tid; t <= T; t += BLOCK) {  // the float32 tables, exactly as f32_table_entry builds them
            float2 vh = make_float2(0.f, 0.f), va = vh;
            if (t < T) {
                float za = zp.a, zd = zp.d, zh = zp.h;
                if (t != tid) {  // (more than 512 teams never get here: <= 64)
                    za = poll_one(&zg[(CLIP ? L.o_sat : L.o_adec) + t], want, fin_tag, &state);
                    zd = poll_one(&zg[(CLIP ? L.o_sdt : L.o_ddec) + t], want, fin_tag, &state);
                    zh = CLIP ? poll_one(&zg[L.o_hadec + t], want, fin_tag, &state) : 0.f;
                }
                float att, def, ha;
                if (!CLIP) {
                    att = fs.s_a * za;
                    def = fs.m + fs.s_d * zd;
                    ha = fs.gam;
                } else {
                    float apm = 0.f, dpm = fs.m;
                    if (coef_fast && t == tid) {   // (the usual case: coefficients came with zp, the row is resident)
#pragma unroll
                        for (int k = 0; k < KREG; ++k) {
                            const float bA = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(zp.coef), k < L.K ? k : 0));
                            const float bD = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(zp.coef), k < L.K ? L.K + k : 0));
                            apm += k < L.K ? xs_reg[k] * bA : 0.f;
                            dpm += k < L.K ? xs_reg[k] * bD : 0.f;
                        }
                    } else
                    for (int k = 0; k < L.K; ++k) {
                        const float xv = A.xsf[(size_t)t * L.K + k];
                        apm += xv * poll_one(&zg[L.o_bA + k], want, fin_tag, &state);
                        dpm += xv * poll_one(&zg[L.o_bD + k], want, fin_tag, &state);
                    }
                    att = apm + za * fs.s_a;
                    def = dpm + zd * fs.s_d;
                    ha = fs.mha + fs.s_h * zh;
                }
                const float edn = exp_f32(-def);
                vh = make_float2(exp_f32(att + ha), edn);
                va = make_float2(exp_f32(att), edn);
            }
            tabH[t] = vh;
            tabA[t] = va;
        }
        for (int i = tid; i < 3 * T1; i += BLOCK) acc[i] = 0.0;
        if (state != 1) *flag = state;  // (any thread: the chain finished, or a bounded wait expired)
        __syncthreads();
        if (*flag != 1) {
            if (*flag < 0 && tid == 0) raise_fault(A.fault, FAULT_LOOP_GRANULES);
            return;
        }
        DC_STAMP(1);
        float mP, mQ, mR;
        pair_maxima_f32<CLIP>(A, tabH, tabA, pr0, redm, tid, &mP, &mQ, &mR);
        const float rho = rho_f32(mP, mQ, mR, fs.q);
        DC_STAMP(2);

        double dSLAM = 0.0, dSLOG = 0.0, dSU = 0.0, dCLIP = 0.0;
        auto process = [&](const LaneData& ld) {
            const LaneOut lo = lane_uniform<WEIGHTED, CLIP>(ld, rho, tabH, tabA);
            dSLAM += rint(ldexp(lo.slam, 30));
            dSLOG += q30(lo.slog);
            dSU += q30(lo.su);
            if (CLIP) dCLIP += q30(lo.sclip);
            float rsh = lo.rsh, rsa = lo.rsa;
            const uint32_t key = lo.key;
            const uint32_t kprev = prev_lane_u32(key, ~key);
            const unsigned long long heads = __ballot(kprev != key);
            const int nruns = __popcll(heads);
            if (nruns == 1) {
                wave_sum2_f32(rsh, rsa);
                if (lane == 0) flush_run_q(acc, T1, key, rsh, rsa);
            } else if (nruns <= RUN_LOOP_MAX) {
                unsigned long long hd = heads;
                while (hd) {
                    const int first = __ffsll((long long)hd) - 1;
                    hd &= hd - 1;
                    const int stop = hd ? __ffsll((long long)hd) - 1 : 64;
                    const bool in = lane >= first && lane < stop;
                    const uint32_t kk = (uint32_t)__builtin_amdgcn_readlane((int)key, first);
                    float s0 = in ? rsh : 0.f, s1 = in ? rsa : 0.f;
                    wave_sum2_f32(s0, s1);
                    if (lane == 0) flush_run_q(acc, T1, kk, s0, s1);
                }
            } else {
                flush_run_q(acc, T1, key, rsh, rsa);
            }
        };
        if (tile0 < tile_end) {
            process(res);
            for (int tile = tile0 + 1; tile < tile_end; ++tile)  // longer streams: the rest from L2 / HBM
                process(load_lane<WEIGHTED>(A, (size_t)tile * 64 + lane));
        }
        DC_STAMP(3);
        {   // (four chains in one written-out reduction: one after the other they were 0.3 us)
            double sv[4] = {dSLAM, dSLOG, dSU, CLIP ? dCLIP : 0.0};
            wave_sum4_f64(sv);
            dSLAM = sv[0]; dSLOG = sv[1]; dSU = sv[2]; dCLIP = sv[3];
        }
        if (lane == 0) {
            red[wave * N_SCAL + 0] = dSLAM;
            red[wave * N_SCAL + 1] = dSLOG;
            red[wave * N_SCAL + 2] = dSU;
            red[wave * N_SCAL + 3] = dCLIP;
        }
        __syncthreads();
        {   // the counted rows: fire and forget, then straight on to the next position (which is
            // published only after the tail has seen every row complete, i.e. after every thread
            // here has read acc / red -- no barrier needed before the next step overwrites them)
            long long* ga = ga0 + (size_t)(s & 1) * ga_set_words(T);   // (the rows alternate by step parity)
            for (int k = o0 + tid; k < o1; k += BLOCK) {
                const int slot = k == o0 + tid ? slot0 : A.wg_slots[k];
                const int which = (slot >= T) + (slot >= 2 * T);   // (0 <= slot < 3T; acc is [3][T + 1]: no division)
                ga_add(ga + (size_t)slot * GA_ROW, acc[slot + which]);
            }
            if (tid >= BLOCK - N_SCAL) {
                const int k = tid - (BLOCK - N_SCAL);
                double sv = 0.0;
#pragma unroll
                for (int wv = 0; wv < WAVES; ++wv) sv += red[wv * N_SCAL + k];
                ga_add(ga + (size_t)(3 * T + (wgi % GA_SHARDS) * N_SCAL + k) * GA_ROW, sv);
            }
        }
        DC_STAMP(4);
        DC_STAMP(5);
        DC_STAMP(6);
    }
}

}  // namespace dc
