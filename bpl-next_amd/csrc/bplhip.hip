// bplhip.hip -- C-ABI (include/bplhip.h) over the gfx950 kernels in dc_kernels.hip.h.
// Host side: context, fixture re-layout (sort by (home,away) pair, pad to the wave
// tile, data-only sums), launch sequences, hipGraph replay, and the NUTS driver glue.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <thread>
#include <tuple>
#include <vector>

#include "../../include/bplhip.h"
#include "dc_dynamic.hip.h"
#include "dc_kernels.hip.h"
#include "dc_neutral.hip.h"
#include "dc_predict.hip.h"
#include "dc_vec.hip.h"
#include "nuts.hpp"
#include "threefry.hpp"

namespace {

thread_local std::string g_create_error;

struct DevBuf {
    void* p = nullptr;
    size_t bytes = 0;
    ~DevBuf() { release(); }
    void release() {
        if (p) (void)hipFree(p);
        p = nullptr;
        bytes = 0;
    }
    hipError_t ensure(size_t n) {
        if (n <= bytes) return hipSuccess;
        release();
        hipError_t e = hipMalloc(&p, n);
        if (e == hipSuccess) bytes = n;
        return e;
    }
    template <class T>
    T* as() const {
        return static_cast<T*>(p);
    }
};

struct GraphKey {
    int count, n_z;
    const void *z, *pot, *grad;
    bool operator<(const GraphKey& o) const {
        return std::tie(count, n_z, z, pot, grad) <
               std::tie(o.count, o.n_z, o.z, o.pot, o.grad);
    }
};

}  // namespace

struct bplhip_ctx {
    int device = 0;
    std::string err;
    bool bound = false;
    dc::Layout L{};
    int64_t n = 0;
    int n_tiles = 0;
    // Launch geometry of dc_eval ("partition"): which waves of a workgroup own tiles, tiles per
    // wave, workgroups, and the sparse-slab structure that follows from it.  Two of them when the
    // stream is short: with idle CUs a chain runs on twice the workgroups with 4 of 8 waves owning
    // tiles (two waves share a SIMD and a tile's lane arithmetic is issue bound: -3 % at N = 1e6,
    // -13 % at 1e5); several chains per launch fill the chip with the 8-wave partition (the
    // 4-wave one cost them 15-25 %).  ep = the one selected for the current launch.
    struct EvalPart {
        int aw = dc::WAVES, tpw = 1, n_wg = 1, total_c = 0;
        bool staged = true;   // the tail stages the compact array in LDS
        DevBuf d_wg_off, d_wg_slots, d_col_off, d_wg_dst;
        DevBuf d_ga_expect;   // contributions every accumulator row receives per evaluation (dc::ga_add)
    } parts[2];
    int n_parts = 1;
    EvalPart* ep = &parts[0];
    int n_cu = 256;
    bool weighted = false;
    int P = 0;
    double lgsum = 0.0;
    // device buffers (library owned)
    DevBuf d_h, d_a, d_x, d_y, d_w, d_pairs, d_xs, d_cA, d_cD, d_cH, d_tickets, d_debug, d_xsf, d_hbuf;
    DevBuf d_gacc;  // accumulator rows of dc_eval's hand-off (dc::GA_ROW)
    DevBuf d_pairw; // per unique pair: sum of weights, of w x, of w y (data only; dc::EvalArgs::pairw)
    DevBuf d_pairc; // per unique pair: weight of its (0,0), (1,0), (0,1) fixtures (dc::ill_pass)
    double w11 = 0.0;  // weight of all (1,1) fixtures
    // fault word: host memory mapped into the device.  A kernel whose bounded wait expires ORs its
    // code in (dc::raise_fault); every entry point looks at it on the way in and on the way out
    // (consume_fault), so a timed-out hand-off becomes BPLHIP_EHIP + a message instead of NaN outputs
    // that a sampler would book as divergences.
    unsigned int* h_fault = nullptr;
    unsigned int* d_fault = nullptr;
    int opt_chunk_graph = 1;
    bool dyn_fused_ok = true;   // cleared when the single-launch dynamic kernel timed out: four launches from then on
    int dyn_fused_blocks_per_cu = -1;  // occupancy of dyn_fused (queried once)
    long long dyn_max_gw = 0;          // fixtures of the largest gameweek (dyn_fused<true> sizes its LDS by it)
    int opt_dyn_big_wgs = 0;           // dyn_fused<true>: workgroups aimed for (0: one per CU)
    int opt_dyn_gather = 1;            // dyn_fused<false>: adjoint records + gather (1) or float64 atomics into the cells (0)
    size_t dyn_big_lds = 0;            // ... the LDS size its cached occupancy belongs to
    int dyn_big_blocks_per_cu = 0;
    bool dyn_big_attr_set = false, neu_big_attr_set = false;
    DevBuf d_zb;    // persistent evaluation kernel: the next position as tagged granules
    unsigned int loop_tag = 0;  // tags handed out so far (a launch of k steps takes k + 1 of them)
    int opt_persistent_kernel = 1;  // 1: a single chain's leapfrogs run inside one resident launch
    int opt_persist_spec = 1;       // ... and the next position is published before the leaf is booked (dc::tail_waves)
    int opt_pair_order = -1;        // fixture layout: 0 (home, away) order, 1 Z-order over (home, away), -1: Z-order past 64 teams
    int opt_dense_pairs = 1;        // 1: complete pair tables take the separable (O(teams)) bounds
    bool pairs_complete = false;
    int opt_fused_small = 1;        // 1: neutral / dynamic evaluations that fit one CU's LDS run as one launch
    bool neu_attr_set = false;
    int slab_chains = 0;
    // tuning options (bplhip_set_option)
    int opt_device_nuts = 1;  // 1: tree builder on the device (nuts_dev.hip.h) when supported
    int opt_max_wg = 255;  // streaming workgroups (+1 prior workgroup = one per CU)
    int opt_active_waves = 0;  // waves per workgroup that own tiles (0 = automatic, see set_fixtures)
    int opt_persistent_nuts = 1;  // bplhip_nuts_run_chains: whole chains on the device (0: lock step)
    int opt_vec_min_chains = 12;  // batched calls with at least this many chains use dc_vec (0: never);
                                  // fewer run as grid.y copies of the single-chain launch (62 workgroups per
                                  // chain at N = 1e6).  Round 4: 12 (was 32) -- with the chain arithmetic done per
                                  // run dc_vec takes 10.5 us for 8 or 16 chains, the copies 9.2 for 8 and 13.8 for 16
    int opt_gridy_max_chains = 8;   // persistent chains: grid.y copies of the NUTS-aware launch up to here (was 32:
                                    // 16 chains 698k -> 921k leapfrogs/s, 32 chains 986k -> 1.22M through dc_vec;
                                    // 8 chains stay with the copies, 582k against 474k: profiles/r04/lockstep_gridy.txt)
    int opt_vec_tpw = 0;         // > 0: force this many tiles per wave for every chain count
    // chain-vectorised partitions (dc_vec.hip.h): fewer, fatter workgroups the more chains
    // share a launch (the per-workgroup prologue builds 8 chains' tables); each has its
    // own sparse-slab structure.  vps[0..2]: 1x / 2x / 3x the single-chain tiles per wave,
    // used for <= 8 / <= 23 / more chains; vp = the one selected for the current launch.
    struct VecPart {
        bool ok = false, staged = true;
        int tpw = 1, n_wg = 1, total_c = 0, slab_chains = 0;
        DevBuf d_wg_off, d_wg_slots, d_col_off, d_wg_dst, d_hbuf;
    } vps[3];
    VecPart* vp = &vps[0];
    bool lds_attr_set = false;
    // NUTS scratch (device): z, potential, grad, aux + pinned host mirror
    DevBuf d_nuts, d_ns;
    // posterior draws for the device predict path (dc_predict.hip.h)
    // (tables 0..7: attack, defence, home_advantage | home_attack, away_attack, home_defence,
    // away_defence, confederation_strength; float64 [S, cols] for the pointwise kernel and float32
    // team-major [cols, S] for the grid kernel)
    DevBuf dp_tab[8], dp_tab32[8], dp_corr, dp_corr32, dp_q;
    int pred_S = 0, pred_T = 0, pred_C = 0, pred_ha_stride = 0;
    bool pred_venue = false;
    double* h_pinned = nullptr;
    size_t h_pinned_bytes = 0;
    // host copies needed by bplhip_constrain (rho bounds over the unique pairs)
    std::vector<uint32_t> h_pairs;
    std::vector<double> h_xs;
    // dynamic (time-varying) model: separate float64 path (dc_dynamic.hip.h)
    bool dynamic = false;
    dcd::DynLayout DL{};
    int dyn_random_walk = 1;
    bool dyn_scratch_clean = false;  // the single-launch kernel finds (and leaves) its scratch zeroed
    // neutral-venue model (dc_neutral.hip.h): the dynamic model's fixture passes + own z side
    bool neutral = false;
    dcn::NeuLayout NL{};
    DevBuf dd_gw, dd_nv, dd_cells, dd_acc, dd_hyp, dd_hc, dd_ac;
    DevBuf dd_fpack, dd_sched, dd_slot_off;  // single-launch neutral kernel (dcn::neu_fused)
    std::vector<unsigned long long> neu_keys;   // neu_big: the sorted fixtures' run keys (host copy: run counts per wave)
    int neu_runs_nb = -1;                       // ... the grid size the cached answer belongs to
    bool neu_runs_ok = false;
    int opt_neu_runs = 1;                       // neu_big: per-run arithmetic (1) / per-fixture (0)
    bool neu_fusable = false;
    int neu_slots = 0;
    int opt_debug_stop = 0;         // diagnostic build only
    DevBuf dd_gwoff, dd_tick;  // dynamic model: first fixture of each gameweek; arrival counters
    DevBuf dd_fx8;             // dynamic model: one packed word per fixture (dcd::pack_fixture), dyn_fused<true>
    DevBuf dd_fadj, dd_inc_off, dd_inc;   // dyn_fused<false>: per-fixture adjoint records + the cells' incidence lists
    bool dyn_gather = false;   // every cell's list is short enough for the gathering phase 4
    bool dyn_attr_set = false;
    std::map<GraphKey, hipGraphExec_t> graphs;
    hipStream_t cap_stream = nullptr;
    bool capturing = false;   // between hipStreamBeginCapture and hipStreamEndCapture on cap_stream: nothing
                              // may allocate, free or synchronise (capture_launches)

    ~bplhip_ctx() {
        for (auto& kv : graphs) (void)hipGraphExecDestroy(kv.second);
        if (cap_stream) (void)hipStreamDestroy(cap_stream);
        if (h_pinned) (void)hipHostFree(h_pinned);
    }
};

namespace {

int fail(bplhip_ctx* c, int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (c) c->err = buf;
    else g_create_error = buf;
    return code;
}

#define HIP_TRY(c, expr)                                                            \
    do {                                                                            \
        hipError_t e_ = (expr);                                                     \
        if (e_ != hipSuccess)                                                       \
            return fail((c), BPLHIP_EHIP, "%s failed: %s (%s:%d)", #expr,           \
                        hipGetErrorString(e_), __FILE__, __LINE__);                 \
    } while (0)

constexpr size_t LDS_LIMIT = 160 * 1024;

// the prior workgroup's record: header | gz[D] | eps[3T] | (dc_vec) the ill-conditioned classes' sums [3T + 4]
int zo_stride_of(const dc::Layout& L) { return (dc::zo_ill_off(L.D, L.T) + dc::zo_ill_len(L.T) + 1) & ~1; }

// dc_eval's hand-off buffer holds the prior workgroup's record only (the fixture sums travel
// through the accumulator rows, d_gacc)
int hb_stride_of(const bplhip_ctx* c) { return zo_stride_of(c->L); }
size_t gacc_bytes_of(const bplhip_ctx* c, int chains) {
    return (size_t)chains * 2 * dc::ga_set_words(c->L.T) * sizeof(long long);   // (two sets per chain)
}
// the partition a launch of `chains` chains uses: the short-stream one (part 0 of two) while its
// workgroups still find a CU each
void select_part(bplhip_ctx* c, int chains) {
    int pi = 0;
    if (c->n_parts > 1 && (long long)chains * (c->parts[0].n_wg + 1) > c->n_cu) pi = 1;
    c->ep = &c->parts[pi];
}

void drop_graphs(bplhip_ctx* c);

// Zero the arrival tickets and the accumulator rows.  Every completed launch leaves them zero
// (the last arriver re-arms both), so this matters after a growth and after a launch that did not
// complete (a fault, an aborted run): entry points that start a chain call it stream-ordered.
int reset_handoff(bplhip_ctx* c, hipStream_t s, bool sync) {
    if (c->slab_chains <= 0) return BPLHIP_OK;
    HIP_TRY(c, hipMemsetAsync(c->d_tickets.p, 0, (size_t)c->slab_chains * dc::TK_WORDS * sizeof(unsigned int), s));
    HIP_TRY(c, hipMemsetAsync(c->d_gacc.p, 0, gacc_bytes_of(c, c->slab_chains), s));
    if (sync) HIP_TRY(c, hipDeviceSynchronize());
    return BPLHIP_OK;
}

// Hand-off slabs and tickets for `chains` chains per launch.  Growing them frees the old buffers
// (hipFree waits for the device), so every cached hipGraphExec -- their kernel arguments hold the
// old pointers -- is dropped with them, and the fresh tickets are zeroed before anything can
// launch on any stream.
int ensure_slabs(bplhip_ctx* c, int chains) {
    select_part(c, chains);
    if (chains <= c->slab_chains) return BPLHIP_OK;
    drop_graphs(c);
    HIP_TRY(c, c->d_hbuf.ensure((size_t)chains * hb_stride_of(c) * sizeof(double)));
    HIP_TRY(c, c->d_tickets.ensure((size_t)chains * dc::TK_WORDS * sizeof(unsigned int)));
    HIP_TRY(c, c->d_gacc.ensure(gacc_bytes_of(c, chains)));
    c->slab_chains = chains;
    return reset_handoff(c, nullptr, true);
}

// the same at the start of a chain: also zero the tickets / rows a launch that never completed
// may have left armed
int ensure_slabs_fresh(bplhip_ctx* c, int chains, hipStream_t s) {
    const int rc = ensure_slabs(c, chains);
    return rc != BPLHIP_OK ? rc : reset_handoff(c, s, false);
}

size_t ctx_lds_bytes(const bplhip_ctx* c, bool staged) {
    return dc::eval_lds_bytes(c->L.T, c->L.D, c->L.K, zo_stride_of(c->L), staged);
}

template <bool W, bool C, bool S, bool N>
int launch_eval_n(bplhip_ctx* c, const dc::EvalArgs& A, int chains, hipStream_t s) {
    const dim3 grid(c->ep->n_wg + 1, chains), block(dc::BLOCK);
    const size_t lds = ctx_lds_bytes(c, S);
    if (lds > 48 * 1024) {  // (idempotent; only for large team counts)
        HIP_TRY(c, hipFuncSetAttribute(reinterpret_cast<const void*>(dc::dc_eval<W, C, S, N>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    }
    // (the leading arguments are the ones gfx950 preloads into SGPRs: dc_eval)
    hipLaunchKernelGGL((dc::dc_eval<W, C, S, N>), grid, block, lds, s, A.z, A.h, A.a, A.x, A.y,
                       (int)(A.L.T | (A.L.K << 16)), A.n_tiles, (int)(A.tiles_per_wave | (A.active_waves << 16)),
                       A.z_stride, A);
    HIP_TRY(c, hipGetLastError());
    return BPLHIP_OK;
}
template <bool W, bool C, bool S>
int launch_eval_s(bplhip_ctx* c, const dc::EvalArgs& A, int chains, hipStream_t s) {
    // the NUTS-aware instantiation is a separate kernel: the plain evaluation carries none
    // of its code (instruction-cache footprint matters at ~11 us per launch)
    return A.nuts ? launch_eval_n<W, C, S, true>(c, A, chains, s)
                  : launch_eval_n<W, C, S, false>(c, A, chains, s);
}
template <bool W, bool C>
int launch_eval_t(bplhip_ctx* c, const dc::EvalArgs& A, int chains, hipStream_t s) {
    return c->ep->staged ? launch_eval_s<W, C, true>(c, A, chains, s)
                     : launch_eval_s<W, C, false>(c, A, chains, s);
}

// Enqueue one (batched) evaluation: ONE launch.  No host sync.
int launch_eval_dynamic(bplhip_ctx* c, int chains, const double* z, double* pot, double* grad,
                        double* aux, hipStream_t s) {
    const dcd::DynLayout& L = c->DL;
    const size_t GT = (size_t)L.G * L.T;
    for (int ch = 0; ch < chains; ++ch) {  // chains run back to back
        dcd::DynArgs A{};
        A.h = c->d_h.as<const uint16_t>();
        A.a = c->d_a.as<const uint16_t>();
        A.x = c->d_x.as<const uint8_t>();
        A.y = c->d_y.as<const uint8_t>();
        A.gw = c->dd_gw.as<const uint16_t>();
        A.nv = c->dd_nv.as<const uint8_t>();
        A.n = c->n;
        A.xs = L.K ? c->d_xs.as<const double>() : nullptr;
        A.lgsum = c->lgsum;
        A.cells = c->dd_cells.as<double>();
        A.acc = c->dd_acc.as<double>();
        A.sc = A.acc + GT * dcd::A_N;
        A.gsum = A.sc + dcd::SC_N;
        A.red = A.gsum + 10 * (size_t)L.G;
        A.cov = A.red + dcd::R_N;
        A.grp = A.cov + 2 * (size_t)L.K;
        A.hyp = c->dd_hyp.as<double>();
        A.z = z + (size_t)ch * L.D;
        A.potential = pot + ch;
        A.grad = grad + (size_t)ch * L.D;
        A.aux = aux ? aux + (size_t)ch * 4 : nullptr;
        A.random_walk = c->dyn_random_walk;
        A.L = L;
        // fixture passes: at most 1024 workgroups of 16 waves (every workgroup ends with same-address
        // global atomics: maxima, U, its private accumulators); pass 2 takes contiguous chunks
        const int fb = c->n >= (1 << 18) ? dcd::FIX_BLOCK : 256;  // threads per workgroup
        const long long nb_all = (c->n + fb - 1) / fb;
        const int nb = (int)std::min<long long>(nb_all, 1024);
        A.chunk = ((c->n + nb - 1) / nb + fb - 1) / fb * fb;
        const int nb2 = (int)((c->n + A.chunk - 1) / A.chunk);
        const int cell_blocks = (L.T + dcd::CELL_BLOCK / 64 - 1) / (dcd::CELL_BLOCK / 64);
        A.scratch_n = dcd::scratch_doubles(L.G, L.T, L.K);
        A.gw_off = c->dd_gwoff.as<const int>();
        A.tickets = c->dd_tick.as<unsigned int>();
        A.fault = c->d_fault;
        A.gather = c->dyn_gather && c->opt_dyn_gather;
        A.fadj = c->dd_fadj.as<double>();
        A.inc_off = c->dd_inc_off.as<const int>();
        A.inc = c->dd_inc.as<const unsigned int>();
        const int team_blocks = (L.T + dcd::BACK_BLOCK / 64 - 1) / (dcd::BACK_BLOCK / 64);
        // one launch when every workgroup is resident and a wave spans all gameweeks
        // (and a workgroup's share of the fixtures is a few rounds of its threads)
        // ... and its workgroups wait for each other inside the launch (flagged cells, two grid barriers):
        // every one of them must be resident at once, on THIS device in its current compute partition
        if (c->dyn_fused_blocks_per_cu < 0) {
            int per_cu = 0;
            if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, dcd::dyn_fused<false>, dcd::FUSED_DYN_BLOCK, 0) != hipSuccess)
                per_cu = 0;
            c->dyn_fused_blocks_per_cu = per_cu;
        }
        const bool co_resident = (long long)c->dyn_fused_blocks_per_cu * c->n_cu >= team_blocks;
        const bool fused_shape = c->opt_fused_small && c->dyn_fused_ok && L.G <= dcd::FUSED_DYN_MAX_G &&
                                 L.T <= dcd::FUSED_DYN_MAX_T;
        auto arm_scratch = [&]() -> int {
            if (!c->dyn_scratch_clean) {  // (the four-launch path leaves its scratch and its cells as it ends)
                HIP_TRY(c, hipMemsetAsync(A.acc, 0, A.scratch_n * 8, s));
                HIP_TRY(c, hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(A.cells), (int)dcd::CELL_EMPTY_WORD,
                                             GT * dcd::P_N * 2, s));
                c->dyn_scratch_clean = true;
            }
            return BPLHIP_OK;
        };
        if (fused_shape && co_resident && c->opt_dyn_big_wgs <= 0 &&
            c->n <= (long long)team_blocks * dcd::FUSED_DYN_BLOCK * 4) {
            if (int rc = arm_scratch()) return rc;
            hipLaunchKernelGGL(dcd::dyn_fused<false>, dim3(team_blocks), dim3(dcd::FUSED_DYN_BLOCK), 0, s, A);
            HIP_TRY(c, hipGetLastError());
            continue;
        }
        if (fused_shape) {
            // more fixtures: the same single launch with `wpg` workgroups per gameweek, a gameweek's slice
            // of the fixtures each (its cell records, accumulators and rates in LDS) -- when the slices fit
            // and every workgroup is resident
            constexpr size_t STATIC_LDS = 24 * 1024;   // dyn_fused's own arrays (20.7 KB), rounded up
            const int want = c->opt_dyn_big_wgs > 0 ? c->opt_dyn_big_wgs : c->n_cu;
            bool launched = false;
            for (int mult = 1; mult <= 4 && !launched; ++mult) {
                const int wpg = std::max(1, want * mult / L.G);
                const int nbig = std::max(team_blocks, wpg * L.G);
                const long long cap = (c->dyn_max_gw + wpg - 1) / wpg;
                if (cap > (1 << 24)) continue;
                // (the slice's fixture words in LDS too when they fit beside its rates)
                const bool stage = dcd::fused_big_lds_bytes(L.T, (int)cap, true) + STATIC_LDS <= LDS_LIMIT;
                const size_t lds = dcd::fused_big_lds_bytes(L.T, (int)cap, stage);
                if (lds + STATIC_LDS > LDS_LIMIT) continue;
                if (!c->dyn_big_attr_set) {
                    HIP_TRY(c, hipFuncSetAttribute(reinterpret_cast<const void*>(dcd::dyn_fused<true>),
                                                   hipFuncAttributeMaxDynamicSharedMemorySize,
                                                   (int)(LDS_LIMIT - STATIC_LDS)));
                    c->dyn_big_attr_set = true;
                }
                if (c->dyn_big_lds != lds) {
                    int per_cu = 0;
                    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, dcd::dyn_fused<true>,
                                                                     dcd::FUSED_DYN_BLOCK, lds) != hipSuccess)
                        per_cu = 0;
                    c->dyn_big_lds = lds;
                    c->dyn_big_blocks_per_cu = per_cu;
                }
                if ((long long)c->dyn_big_blocks_per_cu * c->n_cu < nbig) continue;
                if (int rc = arm_scratch()) return rc;
                A.wpg = wpg;
                A.rate_cap = (int)cap;
                A.fx8 = c->dd_fx8.as<const unsigned long long>();
                A.stage_fx = stage ? 1 : 0;
                hipLaunchKernelGGL(dcd::dyn_fused<true>, dim3(nbig), dim3(dcd::FUSED_DYN_BLOCK), lds, s, A);
                HIP_TRY(c, hipGetLastError());
                launched = true;
            }
            if (launched) continue;
        }
        if (!c->dyn_attr_set) {
            HIP_TRY(c, hipFuncSetAttribute(reinterpret_cast<const void*>(dcd::dyn_pass2),
                                           hipFuncAttributeMaxDynamicSharedMemorySize,
                                           dcd::PASS2_LDS_CELLS * dcd::A_N * 8));
            HIP_TRY(c, hipFuncSetAttribute(reinterpret_cast<const void*>(dcd::dyn_back),
                                           hipFuncAttributeMaxDynamicSharedMemorySize,
                                           (int)dcd::back_lds_bytes(dcd::BACK_LDS_G)));
            c->dyn_attr_set = true;
            c->lds_attr_set = true;
        }
        c->dyn_scratch_clean = false;
        hipLaunchKernelGGL(dcd::dyn_cells, dim3(cell_blocks), dim3(dcd::CELL_BLOCK), 0, s, A);
        hipLaunchKernelGGL(dcd::dyn_pass1, dim3(nb), dim3(fb), 0, s, A);
        hipLaunchKernelGGL(dcd::dyn_pass2, dim3(nb2), dim3(fb),
                           (size_t)dcd::PASS2_LDS_CELLS * dcd::A_N * 8, s, A);
        hipLaunchKernelGGL(dcd::dyn_back, dim3(team_blocks), dim3(dcd::BACK_BLOCK),
                           dcd::back_lds_bytes(L.G), s, A);
        HIP_TRY(c, hipGetLastError());
    }
    return BPLHIP_OK;
}

// the neutral model's single-launch kernel can take the NUTS leaf with it (nuts_state != nullptr)
bool neutral_leaf_fusable(const bplhip_ctx* c) {
    return c->neutral && c->opt_fused_small && c->neu_fusable && c->NL.D <= 64 * nd::LEAF_NE_MAX &&
           (dcn::fused_lds_doubles(c->NL, c->n, c->neu_slots) + dcn::fused_leaf_doubles(c->NL)) * 8 <= LDS_LIMIT;
}
int launch_eval_neutral(bplhip_ctx* c, int chains, const double* z, double* pot, double* grad,
                        double* aux, hipStream_t s, double* nuts_state = nullptr, size_t nuts_stride = 0,
                        int nuts_depth = 0, const nd::Persist* persist = nullptr) {
    const dcn::NeuLayout& L = c->NL;
    if (c->opt_fused_small && c->neu_fusable) {  // one workgroup per chain, one launch for all of them
        dcn::FusedArgs A{};
        A.L = L;
        A.n = (int)c->n;
        A.fx = c->dd_fpack.as<const dcn::FusedFixture>();
        A.n_slots = c->neu_slots;
        A.sched = c->dd_sched.as<const uint32_t>();
        A.slot_off = c->dd_slot_off.as<const int>();
        A.xs = L.K ? c->d_xs.as<const double>() : nullptr;
        A.lgsum = c->lgsum;
        A.z = z;
        A.potential = pot;
        A.grad = grad;
        A.aux = aux;
        A.stop_after = c->opt_debug_stop;
        if (!c->neu_attr_set) {
            HIP_TRY(c, hipFuncSetAttribute(reinterpret_cast<const void*>(dcn::neu_fused<false>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_LIMIT));
            HIP_TRY(c, hipFuncSetAttribute(reinterpret_cast<const void*>(dcn::neu_fused<true>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_LIMIT));
            c->neu_attr_set = true;
        }
        if (nuts_state) {  // persistent chains: one launch evaluates AND books the leapfrog of every chain
            A.nuts = nuts_state;
            A.nuts_stride = nuts_stride;
            A.max_depth = nuts_depth;
            A.persist = *persist;
            hipLaunchKernelGGL(dcn::neu_fused<true>, dim3(chains), dim3(dcn::FUSED_BLOCK),
                               (dcn::fused_lds_doubles(L, c->n, c->neu_slots) + dcn::fused_leaf_doubles(L)) * 8, s, A);
            HIP_TRY(c, hipGetLastError());
            return BPLHIP_OK;
        }
        hipLaunchKernelGGL(dcn::neu_fused<false>, dim3(chains), dim3(dcn::FUSED_BLOCK),
                           dcn::fused_lds_doubles(L, c->n, c->neu_slots) * 8, s, A);
        HIP_TRY(c, hipGetLastError());
        return BPLHIP_OK;
    }
    for (int ch = 0; ch < chains; ++ch) {  // multi-launch path: chains run back to back
        dcn::NeuArgs A{};
        dcd::DynArgs& F = A.F;
        F.h = c->d_h.as<const uint16_t>();
        F.a = c->d_a.as<const uint16_t>();
        F.x = c->d_x.as<const uint8_t>();
        F.y = c->d_y.as<const uint8_t>();
        F.gw = nullptr;
        F.nv = c->dd_nv.as<const uint8_t>();
        F.w = c->weighted ? c->d_w.as<const float>() : nullptr;
        F.n = c->n;
        F.xs = L.K ? c->d_xs.as<const double>() : nullptr;
        F.lgsum = c->lgsum;
        F.cells = c->dd_cells.as<double>();
        F.acc = c->dd_acc.as<double>();
        F.sc = F.acc + (size_t)L.T * dcd::A_N;
        F.cacc = F.sc + dcd::SC_N;
        F.n_conf = L.C;
        F.hc = L.C ? c->dd_hc.as<const uint8_t>() : nullptr;
        F.ac = L.C ? c->dd_ac.as<const uint8_t>() : nullptr;
        F.cs = z + (size_t)ch * L.D + L.o_conf;
        F.z = z + (size_t)ch * L.D;
        F.potential = pot + ch;
        F.grad = grad + (size_t)ch * L.D;
        F.aux = aux ? aux + (size_t)ch * 4 : nullptr;
        F.L.G = 1;
        F.L.T = L.T;
        F.L.K = L.K;
        F.L.o_corr = L.o_corr;
        A.L = L;
        F.scratch_n = (size_t)L.T * dcd::A_N + dcd::SC_N + L.C;
        F.tickets = c->dd_tick.as<unsigned int>();
        F.fault = c->d_fault;
        if (c->opt_fused_small && c->dyn_fused_ok && F.tickets) {
            // one launch (dcn::neu_big): a slice of the fixtures per workgroup, one workgroup per CU, when the
            // slice's rates fit the LDS and every workgroup is resident
            constexpr size_t STATIC_LDS = 2 * 1024;
            const int want = c->opt_dyn_big_wgs > 0 ? c->opt_dyn_big_wgs : c->n_cu;
            const int nbig = (int)std::max<long long>(1, std::min<long long>(want, (c->n + dcn::NEU_BIG_BLOCK - 1) / dcn::NEU_BIG_BLOCK));
            const long long cap = (c->n + nbig - 1) / nbig;
            const size_t lds = dcn::big_lds_bytes(L, cap);
            if (lds + STATIC_LDS <= LDS_LIMIT) {
                if (!c->neu_big_attr_set) {
                    HIP_TRY(c, hipFuncSetAttribute(reinterpret_cast<const void*>(dcn::neu_big),
                                                   hipFuncAttributeMaxDynamicSharedMemorySize,
                                                   (int)(LDS_LIMIT - STATIC_LDS)));
                    c->neu_big_attr_set = true;
                }
                if (c->dyn_big_lds != lds) {
                    int per_cu = 0;
                    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, dcn::neu_big, dcn::NEU_BIG_BLOCK, lds) != hipSuccess)
                        per_cu = 0;
                    c->dyn_big_lds = lds;
                    c->dyn_big_blocks_per_cu = per_cu;
                }
                if ((long long)c->dyn_big_blocks_per_cu * c->n_cu >= nbig) {
                    if (!c->dyn_scratch_clean) {  // (the multi-launch path leaves its scratch as it ends)
                        HIP_TRY(c, hipMemsetAsync(F.acc, 0, F.scratch_n * 8 * dcn::NEU_BIG_GROUPS, s));
                        c->dyn_scratch_clean = true;
                    }
                    F.rate_cap = (int)cap;
                    A.fxp = c->dd_fpack.as<const dcn::FusedFixture>();
                    if (c->neu_runs_nb != nbig) {   // per-run arithmetic: every wave's part must fit its run table
                        constexpr int WV = dcn::NEU_BIG_BLOCK / 64;
                        bool ok = (long long)c->neu_keys.size() == c->n &&
                                  (size_t)WV * dcn::NEU_RUNS_MAX * dcn::NEU_RUN_W <= 2 * (size_t)cap;
                        for (int b = 0; ok && b < nbig; ++b) {
                            const long long i_lo = std::min<long long>((long long)b * cap, c->n), i_hi = std::min<long long>(i_lo + cap, c->n);
                            const long long n_mine = i_hi - i_lo, per_wave = (n_mine + WV - 1) / WV;
                            for (int w = 0; ok && w < WV; ++w) {
                                const long long w0 = std::min<long long>(w * per_wave, n_mine), w1 = std::min<long long>(w0 + per_wave, n_mine);
                                int runs = 0;
                                // (a wave counts a run per step-aligned piece as well: one per key change, plus one where a
                                // 64-fixture step that straddles runs ends inside a run -- bounded by twice the key changes + 1)
                                for (long long i = w0; i < w1; ++i)
                                    runs += i == w0 || c->neu_keys[i_lo + i] != c->neu_keys[i_lo + i - 1];
                                ok = 2 * runs + 1 <= dcn::NEU_RUNS_MAX;
                            }
                        }
                        c->neu_runs_ok = ok;
                        c->neu_runs_nb = nbig;
                    }
                    A.runs = c->neu_runs_ok && c->opt_neu_runs;
                    hipLaunchKernelGGL(dcn::neu_big, dim3(nbig), dim3(dcn::NEU_BIG_BLOCK), lds, s, A);
                    HIP_TRY(c, hipGetLastError());
                    continue;
                }
            }
        }
        c->dyn_scratch_clean = false;
        const int fb = c->n >= (1 << 18) ? dcd::FIX_BLOCK : 256;  // threads per workgroup
        const long long nb_all = (c->n + fb - 1) / fb;
        const int nb = (int)std::min<long long>(nb_all, 1024);
        F.chunk = ((c->n + nb - 1) / nb + fb - 1) / fb * fb;
        const int nb2 = (int)((c->n + F.chunk - 1) / F.chunk);
        hipLaunchKernelGGL(dcn::neu_cells, dim3((L.T + 255) / 256), dim3(256), 0, s, A);
        if (!c->lds_attr_set) {
            HIP_TRY(c, hipFuncSetAttribute(reinterpret_cast<const void*>(dcd::dyn_pass2),
                                           hipFuncAttributeMaxDynamicSharedMemorySize,
                                           dcd::PASS2_LDS_CELLS * dcd::A_N * 8));
            c->lds_attr_set = true;
        }
        hipLaunchKernelGGL(dcd::dyn_pass1, dim3(nb), dim3(fb), 0, s, F);
        hipLaunchKernelGGL(dcd::dyn_pass2, dim3(nb2), dim3(fb),
                           (size_t)dcd::PASS2_LDS_CELLS * dcd::A_N * 8, s, F);
        hipLaunchKernelGGL(dcn::neu_epilogue, dim3(1), dim3(dcn::NEU_EPI),
                           (size_t)(dcn::NEU_SUMS + 2 * L.K) * 8, s, A);
        HIP_TRY(c, hipGetLastError());
    }
    return BPLHIP_OK;
}

dc::EvalArgs eval_args(bplhip_ctx* c, int chains, const double* z, double* pot, double* grad,
                       double* aux) {
    dc::EvalArgs A{};
    A.h = c->d_h.as<const uint32_t>();
    A.a = c->d_a.as<const uint32_t>();
    A.x = c->d_x.as<const uint32_t>();
    A.y = c->d_y.as<const uint32_t>();
    A.w = c->weighted ? c->d_w.as<const float>() : nullptr;
    A.n_tiles = c->n_tiles;
    A.active_waves = c->ep->aw;
    A.tiles_per_wave = c->ep->tpw;
    A.pairs = c->d_pairs.as<const uint32_t>();
    A.pairw = c->d_pairw.as<const double>();
    A.pairc = c->d_pairc.as<const double>();
    A.w11 = c->w11;
    A.P = c->P;
    A.dense_pairs = c->opt_dense_pairs && c->pairs_complete && c->P >= dc::DENSE_MIN_PAIRS;
    A.xs = c->L.K ? c->d_xs.as<const double>() : nullptr;
    A.xsf = c->L.K ? c->d_xsf.as<const float>() : nullptr;
    A.cA = c->d_cA.as<const double>();
    A.cD = c->d_cD.as<const double>();
    A.cH = c->d_cH.as<const double>();
    A.lgsum = c->lgsum;
    A.wg_off = c->ep->d_wg_off.as<const int>();
    A.wg_slots = c->ep->d_wg_slots.as<const int>();
    A.col_off = c->ep->d_col_off.as<const int>();
    A.wg_dst = c->ep->d_wg_dst.as<const int>();
    A.total_c = c->ep->total_c;
    A.hbuf = c->d_hbuf.as<double>();
    A.hb_stride = hb_stride_of(c);
    A.n_wg = c->ep->n_wg;
    A.zo_stride = zo_stride_of(c->L);
    A.tickets = c->d_tickets.as<unsigned int>();
    A.gacc = c->d_gacc.as<long long>();
    A.ga_expect = c->ep->d_ga_expect.as<const int>();
    A.chains = chains;
    A.z = z;
    A.potential = pot;
    A.grad = grad;
    A.aux = aux;
    A.z_stride = c->L.D;
    A.g_stride = c->L.D;
    A.p_stride = 1;
    A.aux_stride = 4;
    A.debug = c->d_debug.as<unsigned long long>();
    A.fault = c->d_fault;
    A.L = c->L;
    return A;
}

int launch_eval(bplhip_ctx* c, int chains, const double* z, double* pot, double* grad,
                double* aux, hipStream_t s, double* nuts_state = nullptr, int nuts_depth = 0,
                const nd::Persist* persist = nullptr, int nuts_stride = 0) {
    if (c->neutral) return launch_eval_neutral(c, chains, z, pot, grad, aux, s);
    if (c->dynamic) return launch_eval_dynamic(c, chains, z, pot, grad, aux, s);
    select_part(c, chains);
    dc::EvalArgs A = eval_args(c, chains, z, pot, grad, aux);
    A.nuts = nuts_state;
    A.nuts_max_depth = nuts_depth;
    A.persist = persist;
    if (nuts_state && nuts_stride) {  // several chains (grid.y): everything lives in the state buffers
        A.nuts_stride = nuts_stride;
        A.z_stride = A.g_stride = A.p_stride = A.aux_stride = nuts_stride;
    }
    const bool clip = c->L.model == dc::MODEL_EXTENDED;
    if (c->weighted) return clip ? launch_eval_t<true, true>(c, A, chains, s)
                                 : launch_eval_t<true, false>(c, A, chains, s);
    return clip ? launch_eval_t<false, true>(c, A, chains, s)
                : launch_eval_t<false, false>(c, A, chains, s);
}

// ---- persistent evaluation kernel (dc_eval_loop): up to `steps` leapfrogs of ONE device-resident
// chain in one launch.  Needs the leaf in the tail (basic / extended, <= 64 teams) and a grid that
// is co-resident (one workgroup per CU).
bool loop_ok(const bplhip_ctx* c) {
    if (c->dynamic || c->neutral || !c->opt_persistent_kernel) return false;
    const bplhip_ctx::EvalPart& ep = c->parts[0];
    // (the leaf's vectors live in registers / LDS lanes there: two elements per lane at most)
    return ep.staged && ep.n_wg + 1 <= c->n_cu && ctx_lds_bytes(c, true) <= 64 * 1024 && c->L.D <= 128;
}
template <bool W, bool C, int LNE>
int launch_loop_l(bplhip_ctx* c, const dc::EvalArgs& A, hipStream_t s) {
    const size_t lds = ctx_lds_bytes(c, true);
    if (lds > 48 * 1024)
        HIP_TRY(c, hipFuncSetAttribute(reinterpret_cast<const void*>(dc::dc_eval_loop<W, C, LNE>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL((dc::dc_eval_loop<W, C, LNE>), dim3(c->ep->n_wg + 1, 1), dim3(dc::BLOCK), lds, s, A);
    HIP_TRY(c, hipGetLastError());
    return BPLHIP_OK;
}
template <bool W, bool C>
int launch_loop_t(bplhip_ctx* c, const dc::EvalArgs& A, hipStream_t s) {
    // the leaf's vectors in registers: one element per lane up to 64 latent entries, two up to 128
    // (the extended model: 3T + 2K + 7); beyond, the LDS-staged leaf of the one-element kernel
    if (c->L.D > 64) return launch_loop_l<W, C, 2>(c, A, s);
    return launch_loop_l<W, C, 1>(c, A, s);
}
int launch_eval_loop(bplhip_ctx* c, double* ns, int nuts_depth, const nd::Persist* persist, int steps,
                     hipStream_t s) {
    const int D = c->L.D;
    select_part(c, 1);
    if (c->d_zb.bytes < (size_t)D * 8 || c->loop_tag > 0xF0000000u) {  // fresh granules: tag 0 everywhere
        HIP_TRY(c, c->d_zb.ensure((size_t)D * 8));
        HIP_TRY(c, hipMemsetAsync(c->d_zb.p, 0, (size_t)D * 8, s));
        c->loop_tag = 0;
    }
    dc::EvalArgs A = eval_args(c, 1, nd::vec(ns, D, nd::V_ZN), ns + nd::H_LEAF_PE, nd::vec(ns, D, nd::V_GRAD),
                               ns + nd::H_LEAF_AUX0);
    A.tag_base = c->loop_tag;
    c->loop_tag += (unsigned int)steps + 2u;  // steps + the launch's own "finished" tag
    A.nuts = ns;
    A.nuts_max_depth = nuts_depth;
    A.persist = persist;
    A.persist_steps = steps;
    A.persist_spec = c->opt_persist_spec;
    A.zg = c->d_zb.as<unsigned long long>();
    const bool clip = c->L.model == dc::MODEL_EXTENDED;
    if (c->weighted) return clip ? launch_loop_t<true, true>(c, A, s) : launch_loop_t<true, false>(c, A, s);
    return clip ? launch_loop_t<false, true>(c, A, s) : launch_loop_t<false, false>(c, A, s);
}

// ---- chain-vectorised evaluation (dc_vec.hip.h): stream launch + tail launch
int vec_hb_stride(const bplhip_ctx* c) {
    return (zo_stride_of(c->L) + c->vp->n_wg * dc::N_SCAL + c->vp->total_c + 1) & ~1;
}
size_t vec_tail_lds(const bplhip_ctx* c, bool staged) {
    return dc::tail_lds_bytes(c->L.T, c->L.D, c->L.K, zo_stride_of(c->L), c->vp->n_wg, c->vp->total_c, staged);
}
template <bool S, bool N, bool E>
int launch_vec_tail(bplhip_ctx* c, const dc::EvalArgs& A, int chains, hipStream_t s) {
    const size_t tl = vec_tail_lds(c, S);
    if (tl > 48 * 1024)
        HIP_TRY(c, hipFuncSetAttribute(reinterpret_cast<const void*>(dc::dc_vec_tail<S, N, E>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)tl));
    hipLaunchKernelGGL((dc::dc_vec_tail<S, N, E>), dim3(chains), dim3(dc::BLOCK), tl, s, A);
    return BPLHIP_OK;
}
template <bool W, bool C, bool N>
int launch_vec_n(bplhip_ctx* c, const dc::EvalArgs& A, int chains, hipStream_t s) {
    const int groups = (chains + dc::CB - 1) / dc::CB;
    const size_t lds = std::max(dc::vec_stream_lds_bytes(c->L.T), dc::prior_lds_bytes(c->L.T));
    if (lds > 48 * 1024)
        HIP_TRY(c, hipFuncSetAttribute(reinterpret_cast<const void*>(dc::dc_vec_stream<W, C, N>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL((dc::dc_vec_stream<W, C, N>), dim3(dc::CB + c->vp->n_wg, groups),
                       dim3(dc::BLOCK), lds, s, A);
    const int rc = c->vp->staged ? launch_vec_tail<true, N, C>(c, A, chains, s)
                                : launch_vec_tail<false, N, C>(c, A, chains, s);
    if (rc != BPLHIP_OK) return rc;
    HIP_TRY(c, hipGetLastError());
    return BPLHIP_OK;
}
template <bool W, bool C>
int launch_vec_t(bplhip_ctx* c, const dc::EvalArgs& A, int chains, hipStream_t s) {
    return A.nuts ? launch_vec_n<W, C, true>(c, A, chains, s)
                  : launch_vec_n<W, C, false>(c, A, chains, s);
}
// nuts != nullptr: lock-step device NUTS; chain c's state at nuts + c*nuts_stride, and
// z / pot / grad / aux point into chain 0's state (strides = nuts_stride)
// select the partition for `chains` chains and size its hand-off buffer.  Growing it is a hipMalloc
// (+ hipFree): not allowed while a stream of this thread is capturing -- whoever captures launches of
// the vectorised kernel sizes it first (round 3 did not: every sampler run with more chains than
// gridy_max_chains on a fresh context failed inside the chunk graph's capture).
int prepare_vec(bplhip_ctx* c, int chains) {
    // The thinnest partition whose grid still fits the chip in (about) one round -- groups x workgroups + one
    // prior workgroup per chain <= CUs: tiles per wave set the launch's length as long as nobody queues behind
    // another workgroup (measured, N = 1e6, us per launch at 1x / 2x / 3x / 4x tiles per wave: 8 chains 10.4 /
    // 12.9 / 14.6 / 15.8, 32 chains 12.7 / 13.2 / 14.5 / 16.1, 64 chains 17.0 / 16.3 / 14.9 / 16.3, 128 chains
    // 21.7 / 22.3 / 19.0 / 21.7, 256 chains 32.8 / 30.1 / 30.0 / 31.6: profiles/r04/vec_tiles_per_wave.txt.
    // Round 3 took 1x / 2x / 4x by chain count alone: 4x at 32 and at 64 chains).
    {
        const int groups = (chains + dc::CB - 1) / dc::CB;
        int pi = 2;
        for (int k = 0; k < 3; ++k)
            if ((long long)groups * c->vps[k].n_wg + chains <= c->n_cu + c->n_cu / 8) {   // (a few workgroups over: still one round)
                pi = k;
                break;
            }
        c->vp = &c->vps[pi];
    }
    if (chains > c->vp->slab_chains) {
        if (c->capturing)
            return fail(c, BPLHIP_ESTATE, "dc_vec: hand-off buffer for %d chains not sized before the capture", chains);
        HIP_TRY(c, c->vp->d_hbuf.ensure((size_t)chains * vec_hb_stride(c) * sizeof(double)));
        c->vp->slab_chains = chains;
    }
    return BPLHIP_OK;
}
int launch_eval_vec(bplhip_ctx* c, int chains, const double* z, double* pot, double* grad,
                    double* aux, hipStream_t s, double* nuts = nullptr, int nuts_stride = 0,
                    int nuts_depth = 0, const nd::Persist* persist = nullptr) {
    {
        const int prc = prepare_vec(c, chains);
        if (prc != BPLHIP_OK) return prc;
    }
    dc::EvalArgs A = eval_args(c, chains, z, pot, grad, aux);
    if (nuts) {
        A.nuts = nuts;
        A.nuts_stride = nuts_stride;
        A.nuts_max_depth = nuts_depth;
        A.persist = persist;
        A.z_stride = A.g_stride = A.p_stride = A.aux_stride = nuts_stride;
    }
    A.tiles_per_wave = c->vp->tpw;
    A.wg_off = c->vp->d_wg_off.as<const int>();
    A.wg_slots = c->vp->d_wg_slots.as<const int>();
    A.col_off = c->vp->d_col_off.as<const int>();
    A.wg_dst = c->vp->d_wg_dst.as<const int>();
    A.total_c = c->vp->total_c;
    A.hbuf = c->vp->d_hbuf.as<double>();
    A.hb_stride = vec_hb_stride(c);
    A.n_wg = c->vp->n_wg;
    A.tickets = nullptr;
    const bool clip = c->L.model == dc::MODEL_EXTENDED;
    if (c->weighted) return clip ? launch_vec_t<true, true>(c, A, chains, s)
                                 : launch_vec_t<true, false>(c, A, chains, s);
    return clip ? launch_vec_t<false, true>(c, A, chains, s)
                : launch_vec_t<false, false>(c, A, chains, s);
}

bool vec_ok(const bplhip_ctx* c) {
    return !c->dynamic && !c->neutral && c->vps[0].ok && c->vps[1].ok && c->vps[2].ok;
}
bool use_vec(const bplhip_ctx* c, int chains) {
    return vec_ok(c) && c->opt_vec_min_chains > 0 && chains >= c->opt_vec_min_chains;
}

// Static sparse-slab structure: which of the 3T per-team slots each streaming workgroup's
// fixtures touch (att[h], ha[h], def[a], att[a], def[h]), and the transposed (per column,
// workgroup order) position lists for the tail's reduction.
struct SparseSlabs {
    std::vector<int> wg_off, wg_slots, col_off, wg_dst;
};
// tiles_per_wg: contiguous tiles of one workgroup
SparseSlabs build_sparse_slabs(const std::vector<uint16_t>& hs, const std::vector<uint16_t>& as,
                               int64_t n, int T, int tiles_per_wg, int n_wg) {
    SparseSlabs o;
    o.wg_off.assign(n_wg + 1, 0);
    o.col_off.assign(3 * T + 1, 0);
    std::vector<char> touched(3 * (size_t)T);
    const int64_t per_wg = (int64_t)tiles_per_wg * dc::TILE;
    for (int w = 0; w < n_wg; ++w) {
        std::fill(touched.begin(), touched.end(), 0);
        const int64_t r0 = (int64_t)w * per_wg, r1 = std::min<int64_t>(n, r0 + per_wg);
        for (int64_t r = r0; r < r1; ++r) {
            const int hh = hs[r], aa = as[r];
            touched[hh] = touched[2 * T + hh] = touched[T + aa] = 1;
            touched[aa] = touched[T + hh] = 1;
        }
        for (int sidx = 0; sidx < 3 * T; ++sidx)
            if (touched[sidx]) o.wg_slots.push_back(sidx);
        o.wg_off[w + 1] = (int)o.wg_slots.size();
    }
    const int total = (int)o.wg_slots.size();
    for (int k = 0; k < total; ++k) o.col_off[o.wg_slots[k] + 1] += 1;
    for (int cidx = 0; cidx < 3 * T; ++cidx) o.col_off[cidx + 1] += o.col_off[cidx];
    o.wg_dst.resize(std::max(total, 1));
    std::vector<int> fill(o.col_off.begin(), o.col_off.end() - 1);
    for (int k = 0; k < total; ++k) o.wg_dst[k] = fill[o.wg_slots[k]]++;  // k ascending = wg order
    if (o.wg_slots.empty()) o.wg_slots.push_back(0);
    return o;
}

void drop_graphs(bplhip_ctx* c) {  // (declared above ensure_slabs)
    for (auto& kv : c->graphs) (void)hipGraphExecDestroy(kv.second);
    c->graphs.clear();
}

// Capture what `enqueue(stream)` launches into an executable graph.  A capture is an optimisation,
// never a requirement: whatever goes wrong inside it (a launch path that wanted to allocate, an API
// the runtime refuses while capturing, an instantiation failure) ends the capture, clears the
// runtime's error state and returns nullptr with BPLHIP_OK -- the caller then enqueues the same
// launches one by one.  Callers run the launches ONCE un-captured first, so that everything a launch
// path sizes or sets lazily (buffers, function attributes, occupancy queries) is settled.
template <class F>
int capture_launches(bplhip_ctx* c, F&& enqueue, hipGraphExec_t* out) {
    *out = nullptr;
    if (!c->cap_stream) HIP_TRY(c, hipStreamCreateWithFlags(&c->cap_stream, hipStreamNonBlocking));
    if (hipStreamBeginCapture(c->cap_stream, hipStreamCaptureModeThreadLocal) != hipSuccess) {
        (void)hipGetLastError();
        return BPLHIP_OK;
    }
    c->capturing = true;
    const bool dyn_clean = c->dyn_scratch_clean;
    const int crc = enqueue(c->cap_stream);
    c->capturing = false;
    hipGraph_t g = nullptr;
    const hipError_t e = hipStreamEndCapture(c->cap_stream, &g);
    hipGraphExec_t ge = nullptr;
    if (crc == BPLHIP_OK && e == hipSuccess && g && hipGraphInstantiate(&ge, g, nullptr, nullptr, 0) == hipSuccess) {
        (void)hipGraphDestroy(g);
        *out = ge;
        return BPLHIP_OK;
    }
    if (g) (void)hipGraphDestroy(g);
    (void)hipGetLastError();
    c->dyn_scratch_clean = dyn_clean;   // (a captured memset never ran)
    return BPLHIP_OK;
}

}  // namespace

// The C-ABI never lets a C++ exception escape (std::bad_alloc from a host buffer, std::system_error
// from a host thread, ...): the entry points that allocate run inside this guard.

// A raised fault word: name it, put the hand-off state back (counters, accumulator rows, the dynamic
// model's scratch) and report.  The device is idle or running work that no longer matters when this
// is called from an entry point; from inside a sampler it is called right after a stream sync.
static int consume_fault(bplhip_ctx* c, const char* where) {
    if (!c || !c->h_fault) return BPLHIP_OK;
    const unsigned int f = __atomic_load_n(c->h_fault, __ATOMIC_RELAXED);
    if (f == 0u) return BPLHIP_OK;
    (void)hipSetDevice(c->device);
    (void)hipDeviceSynchronize();
    __atomic_store_n(c->h_fault, 0u, __ATOMIC_RELAXED);
    (void)reset_handoff(c, nullptr, true);
    if (f & dc::FAULT_DYN_BARRIER) {
        c->dyn_fused_ok = false;
        c->dyn_scratch_clean = false;
        if (c->dd_tick.p) (void)hipMemset(c->dd_tick.p, 0, c->dd_tick.bytes);
    }
    return fail(c, BPLHIP_EHIP,
                "%s: a device-side hand-off timed out (fault word 0x%x:%s%s%s%s); outputs of the affected evaluations are "
                "NaN, the hand-off state has been reset%s",
                where, f, (f & dc::FAULT_EVAL_ARRIVALS) ? " dc_eval arrivals" : "",
                (f & dc::FAULT_LOOP_ARRIVALS) ? " dc_eval_loop arrivals" : "",
                (f & dc::FAULT_LOOP_GRANULES) ? " dc_eval_loop granules" : "",
                (f & dc::FAULT_DYN_BARRIER) ? " dyn_fused barrier" : (f & nd::FAULT_LEAF_BARRIER) ? " wide leaf row barrier" : "",
                (f & dc::FAULT_DYN_BARRIER) ? ", the dynamic model uses its four-launch path from now on" : "");
}

template <class F>
static int guarded(bplhip_ctx* c, const char* where, F&& body) {
    try {
        int rc = consume_fault(c, where);   // raised by asynchronous work of an earlier call
        if (rc != BPLHIP_OK) return rc;
        rc = body();
        return rc != BPLHIP_OK ? rc : consume_fault(c, where);
    } catch (const std::bad_alloc&) {
        return c ? fail(c, BPLHIP_ENOMEM, "%s: out of host memory", where) : BPLHIP_ENOMEM;
    } catch (const std::exception& e) {
        return c ? fail(c, BPLHIP_EHIP, "%s: %s", where, e.what()) : BPLHIP_EHIP;
    } catch (...) {
        return c ? fail(c, BPLHIP_EHIP, "%s: unknown C++ exception", where) : BPLHIP_EHIP;
    }
}

extern "C" {

int bplhip_abi_version(void) { return BPLHIP_ABI_VERSION; }

static int bplhip_create_impl(bplhip_ctx** out, int device_id) {
    if (!out) return fail(nullptr, BPLHIP_EINVAL, "bplhip_create: out is NULL");
    *out = nullptr;
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev <= 0)
        return fail(nullptr, BPLHIP_EHIP, "bplhip_create: no HIP device (%s)",
                    e == hipSuccess ? "count=0" : hipGetErrorString(e));
    if (device_id < 0 || device_id >= ndev)
        return fail(nullptr, BPLHIP_EINVAL, "bplhip_create: device %d out of range [0,%d)",
                    device_id, ndev);
    e = hipSetDevice(device_id);
    if (e != hipSuccess)
        return fail(nullptr, BPLHIP_EHIP, "hipSetDevice(%d): %s", device_id,
                    hipGetErrorString(e));
    bplhip_ctx* c = new (std::nothrow) bplhip_ctx();
    if (!c) return fail(nullptr, BPLHIP_ENOMEM, "bplhip_create: out of host memory");
    c->device = device_id;
    int cus = 0;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device_id) == hipSuccess && cus > 0)
        c->n_cu = cus;
    void* hf = nullptr;
    void* df = nullptr;
    if (hipHostMalloc(&hf, 64, hipHostMallocMapped) != hipSuccess || hipHostGetDevicePointer(&df, hf, 0) != hipSuccess) {
        if (hf) (void)hipHostFree(hf);
        delete c;
        return fail(nullptr, BPLHIP_EHIP, "bplhip_create: cannot map the fault word");
    }
    std::memset(hf, 0, 64);
    c->h_fault = static_cast<unsigned int*>(hf);
    c->d_fault = static_cast<unsigned int*>(df);
    *out = c;
    return BPLHIP_OK;
}

void bplhip_destroy(bplhip_ctx* ctx) {
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    (void)hipDeviceSynchronize();  // (nothing may write the fault word after it is unmapped)
    if (ctx->h_fault) (void)hipHostFree(ctx->h_fault);
    delete ctx;
}

const char* bplhip_last_error(const bplhip_ctx* ctx) {
    return ctx ? ctx->err.c_str() : g_create_error.c_str();
}

static int bplhip_set_fixtures_impl(bplhip_ctx* c, int model_kind, int64_t n, int32_t n_teams,
                        const uint16_t* home_idx, const uint16_t* away_idx,
                        const uint8_t* home_goals, const uint8_t* away_goals,
                        const float* weights, const double* covariates, int32_t k,
                        void* stream) {
    if (!c) return BPLHIP_EINVAL;
    c->bound = false;
    c->dynamic = false;
    c->neutral = false;
    if (model_kind != BPLHIP_MODEL_BASIC && model_kind != BPLHIP_MODEL_EXTENDED)
        return fail(c, BPLHIP_EINVAL, "set_fixtures: unknown model_kind %d", model_kind);
    if (n < 1 || n > (int64_t)1 << 40)
        return fail(c, BPLHIP_EINVAL, "set_fixtures: n=%lld must be >= 1", (long long)n);
    if (n_teams < 1 || n_teams > 65534)
        return fail(c, BPLHIP_EINVAL, "set_fixtures: n_teams=%d out of range [1,65534]", n_teams);
    if (!home_idx || !away_idx || !home_goals || !away_goals)
        return fail(c, BPLHIP_EINVAL, "set_fixtures: null fixture array");
    if (k < 0 || (k > 0 && !covariates) || (k == 0 && covariates))
        return fail(c, BPLHIP_EINVAL, "set_fixtures: covariates/k mismatch (k=%d)", k);
    if (model_kind == BPLHIP_MODEL_BASIC && (k != 0 || weights))
        return fail(c, BPLHIP_EINVAL,
                    "set_fixtures: the basic model takes neither covariates nor weights");
    if (dc::stream_lds_bytes(n_teams) > LDS_LIMIT || dc::prior_lds_bytes(n_teams) > LDS_LIMIT)
        return fail(c, BPLHIP_EUNSUPPORTED,
                    "set_fixtures: n_teams=%d exceeds the LDS-resident table limit", n_teams);
    HIP_TRY(c, hipSetDevice(c->device));
    hipStream_t s = static_cast<hipStream_t>(stream);

    // ---- read the caller's device arrays once
    std::vector<uint16_t> h(n), a(n);
    std::vector<uint8_t> x(n), y(n);
    std::vector<float> w;
    HIP_TRY(c, hipMemcpyAsync(h.data(), home_idx, n * 2, hipMemcpyDeviceToHost, s));
    HIP_TRY(c, hipMemcpyAsync(a.data(), away_idx, n * 2, hipMemcpyDeviceToHost, s));
    HIP_TRY(c, hipMemcpyAsync(x.data(), home_goals, n, hipMemcpyDeviceToHost, s));
    HIP_TRY(c, hipMemcpyAsync(y.data(), away_goals, n, hipMemcpyDeviceToHost, s));
    if (weights) {
        w.resize(n);
        HIP_TRY(c, hipMemcpyAsync(w.data(), weights, n * 4, hipMemcpyDeviceToHost, s));
    }
    HIP_TRY(c, hipStreamSynchronize(s));
    for (int64_t i = 0; i < n; ++i)
        if (h[i] >= n_teams || a[i] >= n_teams)
            return fail(c, BPLHIP_EINVAL, "set_fixtures: team index out of range at fixture %lld",
                        (long long)i);

    // ---- sort by (home, away): runs of equal pairs become contiguous, so a wave-tile
    // reduces to one (or a few) per-pair sums.  Stable, so equal pairs keep input order.
    std::vector<uint64_t> order(n);
    for (int64_t i = 0; i < n; ++i)
        order[i] = ((uint64_t)h[i] << 48) | ((uint64_t)a[i] << 32) | (uint64_t)(uint32_t)i;
    if (n > (int64_t)0xFFFFFFFF) return fail(c, BPLHIP_EUNSUPPORTED, "set_fixtures: n too large");
    std::sort(order.begin(), order.end());

    const int T = n_teams;
    // Leagues of more than 64 teams: the FIXTURES are laid out along the Z-order curve over (home, away)
    // instead (the pair table below keeps the (home, away) order: its index is the tie rule of the bounds).
    // A workgroup's contiguous slice then covers ~sqrt(its pairs) home teams x as many away teams, and adds
    // into that many accumulator rows; in (home, away) order a slice is one home team against everybody --
    // ~3T row adds per workgroup, and every workgroup of the grid adding into the same away-side rows, one
    // after the other at the memory side.  Runs of one pair stay contiguous either way.
    const bool z_order = c->opt_pair_order == 1 || (c->opt_pair_order < 0 && T > 64);
    std::vector<uint64_t> forder;
    if (z_order) {
        auto spread = [](uint32_t v) {  // 16 bits -> every other bit of 32
            v = (v | (v << 8)) & 0x00FF00FFu;
            v = (v | (v << 4)) & 0x0F0F0F0Fu;
            v = (v | (v << 2)) & 0x33333333u;
            v = (v | (v << 1)) & 0x55555555u;
            return v;
        };
        forder.resize(n);
        for (int64_t i = 0; i < n; ++i)
            forder[i] = ((uint64_t)((spread(h[i]) << 1) | spread(a[i])) << 32) | (uint64_t)(uint32_t)i;
        std::sort(forder.begin(), forder.end());
    }
    // Every pair's run is padded to a multiple of LANE_FIX with NULL fixtures (same pair,
    // goals (255, 255), weight 0): a lane never straddles a pair boundary.  Then the whole
    // array is padded to the tile size with the sentinel team T (zero table entries).
    std::vector<uint16_t> hs, as;
    std::vector<uint8_t> xs8, ys8;
    std::vector<uint8_t> is_null;  // padding fixtures (any scoreline is a legal REAL one, 255-255 included)
    std::vector<float> ws;
    hs.reserve(n + n / 8 + dc::TILE); as.reserve(hs.capacity());
    xs8.reserve(hs.capacity()); ys8.reserve(hs.capacity());
    if (weights) ws.reserve(hs.capacity());
    std::vector<uint32_t> pairs;
    std::vector<double> pairw;  // [P][4]: sum w | sum w x | sum w y | 0  (for the prior workgroup's corrections)
    std::vector<double> pairc;  // [P][4]: weight of the pair's (0,0) | (1,0) | (0,1) fixtures | 0  (dc::ill_pass)
    double w11 = 0.0;
    std::vector<double> cA(T, 0.0), cD(T, 0.0), cH(T, 0.0);
    double lgsum = 0.0;
    auto pad_run = [&]() {
        while (hs.size() % dc::LANE_FIX) {
            hs.push_back(hs.back());
            as.push_back(as.back());
            xs8.push_back(255);
            ys8.push_back(255);
            is_null.push_back(1);
            if (weights) ws.push_back(0.0f);
        }
    };
    {   // the fixture arrays, in layout order
        const std::vector<uint64_t>& lay = z_order ? forder : order;
        uint32_t last_pk = 0;
        for (int64_t r = 0; r < n; ++r) {
            const uint32_t i = (uint32_t)lay[r];
            const uint32_t pk = (uint32_t)h[i] | ((uint32_t)a[i] << 16);
            if (r > 0 && pk != last_pk) pad_run();
            last_pk = pk;
            hs.push_back(h[i]);
            as.push_back(a[i]);
            xs8.push_back(x[i]);
            ys8.push_back(y[i]);
            is_null.push_back(0);
            if (weights) ws.push_back(w[i]);
        }
    }
    // the pair table and the host-side sums, in (home, away) order
    for (int64_t r = 0; r < n; ++r) {
        const uint32_t i = (uint32_t)order[r];
        const uint32_t pk = (uint32_t)h[i] | ((uint32_t)a[i] << 16);
        if (pairs.empty() || pairs.back() != pk) {
            pairs.push_back(pk);
            pairw.insert(pairw.end(), {0.0, 0.0, 0.0, 0.0});
            pairc.insert(pairc.end(), {0.0, 0.0, 0.0, 0.0});
        }
        const double wi = weights ? (double)w[i] : 1.0;
        cA[h[i]] += wi * x[i];
        cA[a[i]] += wi * y[i];
        cD[a[i]] += wi * x[i];
        cD[h[i]] += wi * y[i];
        cH[h[i]] += wi * x[i];
        {
            double* pw = &pairw[pairw.size() - 4];
            pw[0] += wi;
            pw[1] += wi * x[i];
            pw[2] += wi * y[i];
            if (x[i] <= 1 && y[i] <= 1) {   // score classes of bpl/_util.py:58-91
                if (x[i] == 1 && y[i] == 1) w11 += wi;
                else pairc[pairc.size() - 4 + (x[i] == 1 ? 1 : (y[i] == 1 ? 2 : 0))] += wi;
            }
        }
        lgsum += wi * (std::lgamma((double)x[i] + 1.0) + std::lgamma((double)y[i] + 1.0));
    }
    pad_run();
    const int64_t n_lanefix = (int64_t)hs.size();  // real + null fixtures
    const int n_tiles = (int)((n_lanefix + dc::TILE - 1) / dc::TILE);
    const int64_t n_pad = (int64_t)n_tiles * dc::TILE;
    hs.resize(n_pad, (uint16_t)T);
    as.resize(n_pad, (uint16_t)T);
    xs8.resize(n_pad, 2);
    ys8.resize(n_pad, 2);
    if (weights) ws.resize(n_pad, 0.0f);

    // ---- launch geometry: 8 waves per workgroup, contiguous tiles per wave
    const int max_wg = c->opt_max_wg;  // one workgroup per CU by default
    // partitions (see EvalPart): part 0 for one chain (or few: while the chip has idle CUs), with
    // 4 of 8 waves owning tiles when that costs no extra tiles per wave; part 1 = all 8 waves
    int aw0 = c->opt_active_waves;
    if (aw0 <= 0 || aw0 > dc::WAVES) aw0 = n_tiles <= max_wg * (dc::WAVES / 2) ? dc::WAVES / 2 : dc::WAVES;
    c->n_parts = aw0 < dc::WAVES && c->opt_active_waves <= 0 ? 2 : 1;
    int tpw = 1;  // of the 8-wave geometry (the chain-vectorised partitions scale it)
    {
        tpw = (n_tiles + max_wg * dc::WAVES - 1) / (max_wg * dc::WAVES);
        if (tpw < 1) tpw = 1;
    }
    // device copies of the indices, run-length encoded: ONE word per lane (all fixtures of a
    // lane share one pair): home | (number of real fixtures of the lane) << 16, and away
    const int64_t n_lanes = n_pad / dc::LANE_FIX;
    std::vector<uint32_t> h_lane(n_lanes), a_lane(n_lanes);
    for (int64_t l = 0; l < n_lanes; ++l) {
        uint32_t cnt = 0;
        for (int j = 0; j < dc::LANE_FIX; ++j) {
            const int64_t r = l * dc::LANE_FIX + j;
            cnt += r < n_lanefix && !is_null[r];
        }
        h_lane[l] = (uint32_t)hs[l * dc::LANE_FIX] | (cnt << 16);
        a_lane[l] = (uint32_t)as[l * dc::LANE_FIX];
    }

    // ---- upload
    for (int pi = 0; pi < c->n_parts; ++pi) {
        bplhip_ctx::EvalPart& ep = c->parts[pi];
        ep.aw = pi == 0 ? aw0 : dc::WAVES;
        ep.tpw = (n_tiles + max_wg * ep.aw - 1) / (max_wg * ep.aw);
        if (ep.tpw < 1) ep.tpw = 1;
        const int waves = (n_tiles + ep.tpw - 1) / ep.tpw;
        ep.n_wg = (waves + ep.aw - 1) / ep.aw;
        const SparseSlabs sp = build_sparse_slabs(hs, as, n_lanefix, T, ep.tpw * ep.aw, ep.n_wg);
        ep.total_c = sp.wg_off[ep.n_wg];
        HIP_TRY(c, ep.d_wg_off.ensure(sp.wg_off.size() * 4));
        HIP_TRY(c, ep.d_wg_slots.ensure(sp.wg_slots.size() * 4));
        HIP_TRY(c, ep.d_col_off.ensure(sp.col_off.size() * 4));
        HIP_TRY(c, ep.d_wg_dst.ensure(sp.wg_dst.size() * 4));
        // (synchronous copies: the host vectors die with this iteration)
        HIP_TRY(c, hipMemcpy(ep.d_wg_off.p, sp.wg_off.data(), sp.wg_off.size() * 4, hipMemcpyHostToDevice));
        HIP_TRY(c, hipMemcpy(ep.d_wg_slots.p, sp.wg_slots.data(), sp.wg_slots.size() * 4, hipMemcpyHostToDevice));
        HIP_TRY(c, hipMemcpy(ep.d_col_off.p, sp.col_off.data(), sp.col_off.size() * 4, hipMemcpyHostToDevice));
        HIP_TRY(c, hipMemcpy(ep.d_wg_dst.p, sp.wg_dst.data(), sp.wg_dst.size() * 4, hipMemcpyHostToDevice));
        {   // how many workgroups add to each accumulator row per evaluation (the rows count their
            // contributions: dc::ga_add): team row = the workgroups that touch the slot, scalar row
            // (shard sh, scalar j) = the workgroups with index = sh (mod GA_SHARDS)
            const int ncol = 3 * T;
            std::vector<int> expect(dc::ga_rows(T), 0);
            for (int cidx = 0; cidx < ncol; ++cidx) expect[cidx] = sp.col_off[cidx + 1] - sp.col_off[cidx];
            for (int w = 0; w < ep.n_wg; ++w)
                for (int j = 0; j < dc::N_SCAL; ++j) expect[ncol + (w % dc::GA_SHARDS) * dc::N_SCAL + j] += 1;
            for (int e : expect)
                if (e > 255)
                    return fail(c, BPLHIP_EUNSUPPORTED,
                                "set_fixtures: %d workgroups feed one accumulator row (the row's counter holds 255)", e);
            HIP_TRY(c, ep.d_ga_expect.ensure(expect.size() * 4));
            HIP_TRY(c, hipMemcpy(ep.d_ga_expect.p, expect.data(), expect.size() * 4, hipMemcpyHostToDevice));
        }
    }
    HIP_TRY(c, c->d_h.ensure(n_lanes * 4));
    HIP_TRY(c, c->d_a.ensure(n_lanes * 4));
    HIP_TRY(c, c->d_x.ensure(n_pad));
    HIP_TRY(c, c->d_y.ensure(n_pad));
    HIP_TRY(c, hipMemcpyAsync(c->d_h.p, h_lane.data(), n_lanes * 4, hipMemcpyHostToDevice, s));
    HIP_TRY(c, hipMemcpyAsync(c->d_a.p, a_lane.data(), n_lanes * 4, hipMemcpyHostToDevice, s));
    HIP_TRY(c, hipMemcpyAsync(c->d_x.p, xs8.data(), n_pad, hipMemcpyHostToDevice, s));
    HIP_TRY(c, hipMemcpyAsync(c->d_y.p, ys8.data(), n_pad, hipMemcpyHostToDevice, s));
    if (weights) {
        HIP_TRY(c, c->d_w.ensure(n_pad * 4));
        HIP_TRY(c, hipMemcpyAsync(c->d_w.p, ws.data(), n_pad * 4, hipMemcpyHostToDevice, s));
    }
    HIP_TRY(c, c->d_pairw.ensure(std::max<size_t>(pairw.size(), 4) * 8));
    HIP_TRY(c, hipMemcpy(c->d_pairw.p, pairw.data(), pairw.size() * 8, hipMemcpyHostToDevice));
    HIP_TRY(c, c->d_pairc.ensure(std::max<size_t>(pairc.size(), 4) * 8));
    HIP_TRY(c, hipMemcpy(c->d_pairc.p, pairc.data(), pairc.size() * 8, hipMemcpyHostToDevice));
    c->w11 = w11;
    HIP_TRY(c, c->d_pairs.ensure(pairs.size() * 4));
    HIP_TRY(c, hipMemcpyAsync(c->d_pairs.p, pairs.data(), pairs.size() * 4,
                              hipMemcpyHostToDevice, s));
    HIP_TRY(c, c->d_cA.ensure(T * 8));
    HIP_TRY(c, c->d_cD.ensure(T * 8));
    HIP_TRY(c, c->d_cH.ensure(T * 8));
    HIP_TRY(c, hipMemcpyAsync(c->d_cA.p, cA.data(), T * 8, hipMemcpyHostToDevice, s));
    HIP_TRY(c, hipMemcpyAsync(c->d_cD.p, cD.data(), T * 8, hipMemcpyHostToDevice, s));
    HIP_TRY(c, hipMemcpyAsync(c->d_cH.p, cH.data(), T * 8, hipMemcpyHostToDevice, s));
    c->h_xs.clear();
    if (k > 0) {
        c->h_xs.assign(covariates, covariates + (size_t)T * k);
        HIP_TRY(c, c->d_xs.ensure((size_t)T * k * 8));
        HIP_TRY(c, hipMemcpyAsync(c->d_xs.p, c->h_xs.data(), (size_t)T * k * 8,
                                  hipMemcpyHostToDevice, s));
        std::vector<float> xsf(c->h_xs.begin(), c->h_xs.end());
        HIP_TRY(c, c->d_xsf.ensure((size_t)T * k * 4));
        HIP_TRY(c, hipMemcpy(c->d_xsf.p, xsf.data(), (size_t)T * k * 4, hipMemcpyHostToDevice));
    }
    HIP_TRY(c, hipStreamSynchronize(s));

    drop_graphs(c);
    c->L = dc::make_layout(model_kind, T, k);
    c->lds_attr_set = false;
    c->n = n;
    c->n_tiles = n_tiles;
    c->weighted = weights != nullptr;
    c->P = (int)pairs.size();
    c->lgsum = lgsum;
    // every ordered pair h != a present (and no team against itself): the bounds are separable
    c->pairs_complete = T >= 2 && (long long)pairs.size() == (long long)T * (T - 1) &&
                        std::none_of(pairs.begin(), pairs.end(), [](uint32_t pr) { return (pr & 0xFFFFu) == (pr >> 16); });
    c->h_pairs = std::move(pairs);
    c->slab_chains = 0;
    // staged <=> T <= 64: the tail's one-lane-per-team epilogue (and the device-resident NUTS)
    for (int pi = 0; pi < c->n_parts; ++pi) {
        c->ep = &c->parts[pi];
        c->ep->staged = T <= 64 && ctx_lds_bytes(c, true) <= 96 * 1024;
        if (ctx_lds_bytes(c, c->ep->staged) > LDS_LIMIT)
            return fail(c, BPLHIP_EUNSUPPORTED, "set_fixtures: tail LDS footprint too large");
    }
    select_part(c, 1);
    int rc = ensure_slabs(c, 1);
    if (rc != BPLHIP_OK) return rc;

    // ---- chain-vectorised partition: fewer, fatter workgroups (the per-workgroup prologue
    // of 8 chains' tables is amortised over more tiles)
    for (int pi = 0; pi < 3; ++pi) {
        auto& vp = c->vps[pi];
        c->vp = &vp;  // (the LDS helpers below read the current partition)
        vp.ok = false;
        vp.slab_chains = 0;
        vp.tpw = c->opt_vec_tpw > 0 ? std::max(tpw, c->opt_vec_tpw) : tpw * (pi + 1);   // (1x / 2x / 3x: prepare_vec)
        const int vwaves = (n_tiles + vp.tpw - 1) / vp.tpw;
        vp.n_wg = (vwaves + dc::WAVES - 1) / dc::WAVES;
        const SparseSlabs vs = build_sparse_slabs(hs, as, n_lanefix, T, vp.tpw * dc::WAVES, vp.n_wg);
        vp.total_c = vs.wg_off[vp.n_wg];
        HIP_TRY(c, vp.d_wg_off.ensure(vs.wg_off.size() * 4));
        HIP_TRY(c, vp.d_wg_slots.ensure(vs.wg_slots.size() * 4));
        HIP_TRY(c, vp.d_col_off.ensure(vs.col_off.size() * 4));
        HIP_TRY(c, vp.d_wg_dst.ensure(vs.wg_dst.size() * 4));
        HIP_TRY(c, hipMemcpy(vp.d_wg_off.p, vs.wg_off.data(), vs.wg_off.size() * 4, hipMemcpyHostToDevice));
        HIP_TRY(c, hipMemcpy(vp.d_wg_slots.p, vs.wg_slots.data(), vs.wg_slots.size() * 4, hipMemcpyHostToDevice));
        HIP_TRY(c, hipMemcpy(vp.d_col_off.p, vs.col_off.data(), vs.col_off.size() * 4, hipMemcpyHostToDevice));
        HIP_TRY(c, hipMemcpy(vp.d_wg_dst.p, vs.wg_dst.data(), vs.wg_dst.size() * 4, hipMemcpyHostToDevice));
        vp.staged = T <= 64 && vec_tail_lds(c, true) <= 96 * 1024;
        vp.ok = dc::vec_stream_lds_bytes(T) <= 64 * 1024 && vec_tail_lds(c, vp.staged) <= LDS_LIMIT;
    }
    c->vp = &c->vps[0];
    c->bound = true;
    return BPLHIP_OK;
}

int bplhip_set_option(bplhip_ctx* c, const char* name, int value) {
    if (!c || !name) return BPLHIP_EINVAL;
    const std::string n(name);
    if (n == "device_nuts") {
        c->opt_device_nuts = value != 0;
        return BPLHIP_OK;
    }
    if (n == "dyn_big_wgs") {  // dynamic model, single launch: > 0 takes the sliced form (dyn_fused<true>) whatever
                               // N is, aiming for that many workgroups; 0: sliced form past 1024 fixtures per
                               // team workgroup, one workgroup per CU
        if (value < 0 || value > 65535) return fail(c, BPLHIP_EINVAL, "set_option: dyn_big_wgs out of range");
        c->opt_dyn_big_wgs = value;
        drop_graphs(c);
        return BPLHIP_OK;
    }
    if (n == "persistent_nuts") {
        c->opt_persistent_nuts = value != 0;
        return BPLHIP_OK;
    }
    if (n == "persistent_kernel") {  // 0: one launch per leapfrog even for a single resident chain
        c->opt_persistent_kernel = value != 0;
        return BPLHIP_OK;
    }
#ifdef DC_STAMPS
    if (n == "debug_stop") {
        c->opt_debug_stop = value;
        drop_graphs(c);
        return BPLHIP_OK;
    }
#endif
    if (n == "debug_raise_fault") {  // test hook: raise the fault word as a timed-out kernel would
        if (c->h_fault) __atomic_fetch_or(c->h_fault, (unsigned int)value, __ATOMIC_RELAXED);
        return BPLHIP_OK;
    }
    if (n == "pair_order") {   // takes effect at the next set_fixtures
        if (value < -1 || value > 1) return fail(c, BPLHIP_EINVAL, "set_option: pair_order is -1, 0 or 1");
        c->opt_pair_order = value;
        return BPLHIP_OK;
    }
    if (n == "dense_pairs") {  // 0: the rho bounds always walk the pair table
        c->opt_dense_pairs = value != 0;
        drop_graphs(c);
        return BPLHIP_OK;
    }
    if (n == "fused_small") {  // 0: neutral / dynamic evaluations always take the multi-launch path
        c->opt_fused_small = value != 0;
        drop_graphs(c);  // (captured launch sequences belong to the other path)
        return BPLHIP_OK;
    }
    if (n == "gridy_max_chains") {
        if (value < 1 || value > 4096) return fail(c, BPLHIP_EINVAL, "gridy_max_chains out of range");
        c->opt_gridy_max_chains = value;
        return BPLHIP_OK;
    }
    if (n == "vec_min_chains") {  // batched calls with >= value chains use the vectorised kernel
        if (value < 0) return fail(c, BPLHIP_EINVAL, "vec_min_chains must be >= 0");
        c->opt_vec_min_chains = value;
        return BPLHIP_OK;
    }
    if (n == "vec_tiles_per_wave") {  // 0 = by chain count; takes effect at the next bplhip_set_fixtures
        if (value < 0 || value > 4096) return fail(c, BPLHIP_EINVAL, "vec_tiles_per_wave out of range");
        c->opt_vec_tpw = value;
        return BPLHIP_OK;
    }
    if (n == "neu_runs") {  // neutral model's sliced single launch: per-run arithmetic (1) / per fixture (0)
        c->opt_neu_runs = value != 0;
        drop_graphs(c);
        return BPLHIP_OK;
    }
    if (n == "dyn_gather") {  // dynamic model's small single launch: per-fixture adjoint records gathered by the cells (1) / atomics (0)
        c->opt_dyn_gather = value != 0;
        drop_graphs(c);
        return BPLHIP_OK;
    }
    if (n == "persist_spec") {  // persistent kernel: the next position before the leaf (1) or after it (0)
        c->opt_persist_spec = value != 0;
        return BPLHIP_OK;
    }
    if (n == "chunk_graph") {  // persistent chains: a chunk of leapfrogs as one replayed hipGraph (1) or launch by launch (0)
        c->opt_chunk_graph = value != 0;
        return BPLHIP_OK;
    }
    if (n == "max_wg") {  // takes effect at the next bplhip_set_fixtures
        // (an accumulator row counts its contributions in 8 bits, dc::GA_COUNT_SHIFT: with few teams every
        // workgroup adds to every row)
        if (value < 1 || value > 255) return fail(c, BPLHIP_EINVAL, "max_wg out of range [1,255]");
        c->opt_max_wg = value;
        return BPLHIP_OK;
    }
    if (n == "active_waves") {  // waves per workgroup that own tiles; 0 = automatic; next set_fixtures
        if (value < 0 || value > dc::WAVES) return fail(c, BPLHIP_EINVAL, "active_waves: 0 (automatic) .. 8");
        c->opt_active_waves = value;
        return BPLHIP_OK;
    }
    return fail(c, BPLHIP_EINVAL, "unknown option '%s'", name);
}

int bplhip_latent_dim(const bplhip_ctx* c) {
    if (!c) return BPLHIP_EINVAL;
    if (!c->bound) return BPLHIP_ESTATE;
    return c->neutral ? c->NL.D : (c->dynamic ? c->DL.D : c->L.D);
}

static int bplhip_set_fixtures_neutral_impl(bplhip_ctx* c, int64_t n, int32_t n_teams, const uint16_t* home_idx,
                                const uint16_t* away_idx, const uint8_t* home_goals,
                                const uint8_t* away_goals, const uint8_t* neutral_venue,
                                const uint8_t* home_conf, const uint8_t* away_conf, int32_t n_conf,
                                const float* weights, const double* covariates, int32_t k,
                                void* stream) {
    if (!c) return BPLHIP_EINVAL;
    c->bound = false;
    if (n < 1 || n > (int64_t)0xFFFFFFFF || n_teams < 1 || n_teams > 65534)
        return fail(c, BPLHIP_EINVAL, "set_fixtures_neutral: bad sizes");
    if (n_conf < 0 || n_conf > 255 || (n_conf > 0 && (!home_conf || !away_conf)) ||
        (n_conf == 0 && (home_conf || away_conf)))
        return fail(c, BPLHIP_EINVAL, "set_fixtures_neutral: confederation arrays / n_conf mismatch");
    if (!home_idx || !away_idx || !home_goals || !away_goals || !neutral_venue)
        return fail(c, BPLHIP_EINVAL, "set_fixtures_neutral: null fixture array");
    if (k < 0 || (k > 0 && !covariates) || (k == 0 && covariates))
        return fail(c, BPLHIP_EINVAL, "set_fixtures_neutral: covariates/k mismatch");
    HIP_TRY(c, hipSetDevice(c->device));
    hipStream_t s = static_cast<hipStream_t>(stream);
    std::vector<uint16_t> h(n), a(n);
    std::vector<uint8_t> x(n), y(n), nv(n);
    std::vector<float> w;
    HIP_TRY(c, hipMemcpyAsync(h.data(), home_idx, n * 2, hipMemcpyDeviceToHost, s));
    HIP_TRY(c, hipMemcpyAsync(a.data(), away_idx, n * 2, hipMemcpyDeviceToHost, s));
    HIP_TRY(c, hipMemcpyAsync(x.data(), home_goals, n, hipMemcpyDeviceToHost, s));
    HIP_TRY(c, hipMemcpyAsync(y.data(), away_goals, n, hipMemcpyDeviceToHost, s));
    HIP_TRY(c, hipMemcpyAsync(nv.data(), neutral_venue, n, hipMemcpyDeviceToHost, s));
    if (weights) {
        w.resize(n);
        HIP_TRY(c, hipMemcpyAsync(w.data(), weights, n * 4, hipMemcpyDeviceToHost, s));
    }
    HIP_TRY(c, hipStreamSynchronize(s));
    double lgsum = 0.0;
    for (int64_t i = 0; i < n; ++i) {
        if (h[i] >= n_teams || a[i] >= n_teams || nv[i] > 1)
            return fail(c, BPLHIP_EINVAL, "set_fixtures_neutral: value out of range at fixture %lld",
                        (long long)i);
        const double wi = weights ? (double)w[i] : 1.0;
        lgsum += wi * (std::lgamma((double)x[i] + 1.0) + std::lgamma((double)y[i] + 1.0));
    }
    // sort by (venue, home, away): a wave of dyn_pass2 then usually sits on one pair and adds
    // its 64 contributions with one set of atomics.  (Confederations follow the teams.)
    std::vector<uint32_t> order(n);
    {
        for (int64_t i = 0; i < n; ++i) order[i] = (uint32_t)i;
        std::stable_sort(order.begin(), order.end(), [&](uint32_t p, uint32_t q) {
            // (last key: the fixtures with a tau term -- both sides <= 1 goal -- first within a run, so that a
            // wave's 64 consecutive fixtures mostly agree on having one: dcn::neu_big)
            const uint64_t kp = ((uint64_t)nv[p] << 33) | ((uint64_t)h[p] << 17) | ((uint64_t)a[p] << 1) |
                                (uint64_t)!(x[p] <= 1 && y[p] <= 1);
            const uint64_t kq = ((uint64_t)nv[q] << 33) | ((uint64_t)h[q] << 17) | ((uint64_t)a[q] << 1) |
                                (uint64_t)!(x[q] <= 1 && y[q] <= 1);
            return kp < kq;
        });
        auto permute = [&](auto& v) {
            auto t = v;
            for (int64_t i = 0; i < n; ++i) t[i] = v[order[i]];
            v.swap(t);
        };
        permute(h); permute(a); permute(x); permute(y); permute(nv);
        if (weights) permute(w);
    }
    HIP_TRY(c, c->d_h.ensure(n * 2));
    HIP_TRY(c, c->d_a.ensure(n * 2));
    HIP_TRY(c, c->d_x.ensure(n));
    HIP_TRY(c, c->d_y.ensure(n));
    HIP_TRY(c, c->dd_nv.ensure(n));
    HIP_TRY(c, hipMemcpyAsync(c->d_h.p, h.data(), n * 2, hipMemcpyHostToDevice, s));
    HIP_TRY(c, hipMemcpyAsync(c->d_a.p, a.data(), n * 2, hipMemcpyHostToDevice, s));
    HIP_TRY(c, hipMemcpyAsync(c->d_x.p, x.data(), n, hipMemcpyHostToDevice, s));
    HIP_TRY(c, hipMemcpyAsync(c->d_y.p, y.data(), n, hipMemcpyHostToDevice, s));
    HIP_TRY(c, hipMemcpyAsync(c->dd_nv.p, nv.data(), n, hipMemcpyHostToDevice, s));
    if (weights) {
        HIP_TRY(c, c->d_w.ensure(n * 4));
        HIP_TRY(c, hipMemcpyAsync(c->d_w.p, w.data(), n * 4, hipMemcpyHostToDevice, s));
    }
    std::vector<uint8_t> hcv, acv;
    if (n_conf > 0) {  // (device -> device: the library keeps its own copy)
        hcv.resize(n); acv.resize(n);
        HIP_TRY(c, hipMemcpy(hcv.data(), home_conf, n, hipMemcpyDeviceToHost));
        HIP_TRY(c, hipMemcpy(acv.data(), away_conf, n, hipMemcpyDeviceToHost));
        for (int64_t i = 0; i < n; ++i)
            if (hcv[i] >= n_conf || acv[i] >= n_conf)
                return fail(c, BPLHIP_EINVAL, "set_fixtures_neutral: confederation out of range at fixture %lld",
                            (long long)i);
        {   // same order as the fixtures
            auto t1 = hcv, t2 = acv;
            for (int64_t i = 0; i < n; ++i) { t1[i] = hcv[order[i]]; t2[i] = acv[order[i]]; }
            hcv.swap(t1); acv.swap(t2);
        }
        HIP_TRY(c, c->dd_hc.ensure(n));
        HIP_TRY(c, c->dd_ac.ensure(n));
        HIP_TRY(c, hipMemcpy(c->dd_hc.p, hcv.data(), n, hipMemcpyHostToDevice));
        HIP_TRY(c, hipMemcpy(c->dd_ac.p, acv.data(), n, hipMemcpyHostToDevice));
    }
    // single-launch kernel (dcn::neu_fused): packed fixtures and the schedule of its gather step
    const dcn::NeuLayout NL = dcn::make_neu_layout(n_teams, k, n_conf);
    c->neu_fusable = n <= dcn::FUSED_MAX_N && n_teams <= dcn::FUSED_MAX_T && n_conf <= dcn::FUSED_MAX_C;
    if (c->neu_fusable) {
        std::vector<dcn::FusedFixture> pack(n);
        std::vector<int> off(n_teams + 1, 0);
        for (int64_t i = 0; i < n; ++i) {
            dcn::FusedFixture f{};
            f.h = h[i]; f.a = a[i]; f.x = x[i]; f.y = y[i]; f.nv = nv[i];
            f.hc = n_conf ? hcv[i] : 0;
            f.ac = n_conf ? acv[i] : 0;
            f.w = weights ? w[i] : 1.0f;
            pack[i] = f;
            ++off[h[i] + 1];
            ++off[a[i] + 1];
        }
        for (int t = 0; t < n_teams; ++t) off[t + 1] += off[t];
        // team-major incidence entries (fixture order within a team: the sums have a fixed order)
        std::vector<uint32_t> inc(2 * n);
        {
            std::vector<int> fill(off.begin(), off.end() - 1);
            for (int64_t i = 0; i < n; ++i) {
                inc[fill[h[i]]++] = ((uint32_t)i << 2) | ((uint32_t)nv[i] << 1);
                inc[fill[a[i]]++] = ((uint32_t)i << 2) | ((uint32_t)nv[i] << 1) | 1u;
            }
        }
        // slots: <= 64 consecutive entries of one team, taken by one 16-lane row (lane `sub`
        // holds entries sub, sub + 16, ...); row r of the workgroup takes slots r, r + 64, ...
        constexpr int SLOT = 16 * dcn::FUSED_PRE;
        std::vector<int> slot_off(n_teams + 1, 0);
        for (int t = 0; t < n_teams; ++t) slot_off[t + 1] = slot_off[t] + (off[t + 1] - off[t] + SLOT - 1) / SLOT;
        const int n_slots = slot_off[n_teams];
        const int rounds = std::max(1, (n_slots + dcn::FUSED_ROWS - 1) / dcn::FUSED_ROWS);
        std::vector<uint32_t> sched((size_t)rounds * dcn::FUSED_PRE * dcn::FUSED_BLOCK, dcn::INC_NONE);
        for (int t = 0; t < n_teams; ++t)
            for (int sl = slot_off[t]; sl < slot_off[t + 1]; ++sl) {
                const int b0 = off[t] + (sl - slot_off[t]) * SLOT, b1 = std::min(b0 + SLOT, off[t + 1]);
                const int round = sl / dcn::FUSED_ROWS, row = sl % dcn::FUSED_ROWS;
                for (int e = b0; e < b1; ++e) {
                    const int sub = (e - b0) % 16, r = (e - b0) / 16;
                    sched[((size_t)round * dcn::FUSED_PRE + r) * dcn::FUSED_BLOCK + row * 16 + sub] = inc[e];
                }
            }
        c->neu_slots = n_slots;
        c->neu_fusable = dcn::fused_lds_doubles(NL, n, n_slots) * 8 <= LDS_LIMIT;
        if (c->neu_fusable) {
            HIP_TRY(c, c->dd_fpack.ensure(pack.size() * sizeof(dcn::FusedFixture)));
            HIP_TRY(c, c->dd_sched.ensure(sched.size() * 4));
            HIP_TRY(c, c->dd_slot_off.ensure(slot_off.size() * 4));
            HIP_TRY(c, hipMemcpy(c->dd_fpack.p, pack.data(), pack.size() * sizeof(dcn::FusedFixture), hipMemcpyHostToDevice));
            HIP_TRY(c, hipMemcpy(c->dd_sched.p, sched.data(), sched.size() * 4, hipMemcpyHostToDevice));
            HIP_TRY(c, hipMemcpy(c->dd_slot_off.p, slot_off.data(), slot_off.size() * 4, hipMemcpyHostToDevice));
        }
    }
    c->neu_keys.clear();
    c->neu_runs_nb = -1;
    if (!c->neu_fusable) {  // dcn::neu_big streams the same 16-byte records
        std::vector<dcn::FusedFixture> pack(n);
        c->neu_keys.resize(n);
        for (int64_t i = 0; i < n; ++i) {
            dcn::FusedFixture f{};
            f.h = h[i]; f.a = a[i]; f.x = x[i]; f.y = y[i]; f.nv = nv[i];
            f.hc = n_conf ? hcv[i] : 0;
            f.ac = n_conf ? acv[i] : 0;
            f.w = weights ? w[i] : 1.0f;
            pack[i] = f;
            c->neu_keys[i] = ((unsigned long long)f.h << 33) | ((unsigned long long)f.a << 17) |
                             ((unsigned long long)(f.nv != 0) << 16) | ((unsigned long long)f.hc << 8) | (unsigned long long)f.ac;
        }
        HIP_TRY(c, c->dd_fpack.ensure(pack.size() * sizeof(dcn::FusedFixture)));
        HIP_TRY(c, hipMemcpy(c->dd_fpack.p, pack.data(), pack.size() * sizeof(dcn::FusedFixture), hipMemcpyHostToDevice));
    }
    HIP_TRY(c, c->dd_cells.ensure((size_t)n_teams * dcd::P_N * 8));
    // (dcn::neu_big adds into NEU_BIG_GROUPS copies of the scratch; the multi-launch path uses the first)
    const size_t neu_scratch_bytes = ((size_t)n_teams * dcd::A_N + dcd::SC_N + n_conf) * 8 * dcn::NEU_BIG_GROUPS;
    HIP_TRY(c, c->dd_acc.ensure(neu_scratch_bytes));
    // the single-launch kernel for large N (dcn::neu_big) finds its scratch and its counters zeroed, and leaves them so
    HIP_TRY(c, hipMemset(c->dd_acc.p, 0, neu_scratch_bytes));
    HIP_TRY(c, c->dd_tick.ensure(dcd::TICKET_BYTES));
    HIP_TRY(c, hipMemset(c->dd_tick.p, 0, dcd::TICKET_BYTES));
    c->dyn_scratch_clean = true;
    c->dyn_big_lds = 0;   // (the cached occupancy belongs to one kernel and one LDS size)
    c->h_xs.clear();
    if (k > 0) {
        c->h_xs.assign(covariates, covariates + (size_t)n_teams * k);
        HIP_TRY(c, c->d_xs.ensure((size_t)n_teams * k * 8));
        HIP_TRY(c, hipMemcpyAsync(c->d_xs.p, c->h_xs.data(), (size_t)n_teams * k * 8,
                                  hipMemcpyHostToDevice, s));
    }
    HIP_TRY(c, hipStreamSynchronize(s));
    drop_graphs(c);
    c->NL = dcn::make_neu_layout(n_teams, k, n_conf);
    c->n = n;
    c->lgsum = lgsum;
    c->weighted = weights != nullptr;
    c->lds_attr_set = false;
    c->dynamic = false;
    c->neutral = true;
    c->bound = true;
    return BPLHIP_OK;
}

static int bplhip_set_fixtures_dynamic_impl(bplhip_ctx* c, int64_t n, int32_t n_teams, int32_t n_gameweeks,
                                const uint16_t* home_idx, const uint16_t* away_idx,
                                const uint8_t* home_goals, const uint8_t* away_goals,
                                const uint16_t* gameweek, const uint8_t* neutral_venue,
                                const double* covariates, int32_t k, int32_t random_walk,
                                void* stream) {
    if (!c) return BPLHIP_EINVAL;
    c->bound = false;
    c->neutral = false;
    if (n < 1 || n_teams < 1 || n_teams > 65534 || n_gameweeks < 1 || n_gameweeks > 65535)
        return fail(c, BPLHIP_EINVAL, "set_fixtures_dynamic: bad sizes");
    if (!home_idx || !away_idx || !home_goals || !away_goals || !gameweek || !neutral_venue)
        return fail(c, BPLHIP_EINVAL, "set_fixtures_dynamic: null fixture array");
    if (k < 0 || (k > 0 && !covariates) || (k == 0 && covariates))
        return fail(c, BPLHIP_EINVAL, "set_fixtures_dynamic: covariates/k mismatch");
    if ((int64_t)n_teams * n_gameweeks > (int64_t)1 << 26)
        return fail(c, BPLHIP_EUNSUPPORTED, "set_fixtures_dynamic: too many (gameweek, team) cells");
    HIP_TRY(c, hipSetDevice(c->device));
    hipStream_t s = static_cast<hipStream_t>(stream);
    std::vector<uint16_t> h(n), a(n), g(n);
    std::vector<uint8_t> x(n), y(n), nv(n);
    HIP_TRY(c, hipMemcpyAsync(h.data(), home_idx, n * 2, hipMemcpyDeviceToHost, s));
    HIP_TRY(c, hipMemcpyAsync(a.data(), away_idx, n * 2, hipMemcpyDeviceToHost, s));
    HIP_TRY(c, hipMemcpyAsync(g.data(), gameweek, n * 2, hipMemcpyDeviceToHost, s));
    HIP_TRY(c, hipMemcpyAsync(x.data(), home_goals, n, hipMemcpyDeviceToHost, s));
    HIP_TRY(c, hipMemcpyAsync(y.data(), away_goals, n, hipMemcpyDeviceToHost, s));
    HIP_TRY(c, hipMemcpyAsync(nv.data(), neutral_venue, n, hipMemcpyDeviceToHost, s));
    HIP_TRY(c, hipStreamSynchronize(s));
    double lgsum = 0.0;
    for (int64_t i = 0; i < n; ++i) {
        if (h[i] >= n_teams || a[i] >= n_teams || g[i] >= n_gameweeks || nv[i] > 1)
            return fail(c, BPLHIP_EINVAL, "set_fixtures_dynamic: index out of range at fixture %lld",
                        (long long)i);
        lgsum += std::lgamma((double)x[i] + 1.0) + std::lgamma((double)y[i] + 1.0);
    }
    {   // stable sort by gameweek: a workgroup of dyn_pass2 then spans only a few gameweeks
        std::vector<uint32_t> order(n);
        for (int64_t i = 0; i < n; ++i) order[i] = (uint32_t)i;
        if (n > (int64_t)0xFFFFFFFF) return fail(c, BPLHIP_EUNSUPPORTED, "set_fixtures_dynamic: n too large");
        std::stable_sort(order.begin(), order.end(), [&](uint32_t p, uint32_t q) { return g[p] < g[q]; });
        auto permute = [&](auto& v) {
            auto w = v;
            for (int64_t i = 0; i < n; ++i) w[i] = v[order[i]];
            v.swap(w);
        };
        // ... and inside a gameweek, in blocks of 1024, the fixtures with a tau term (both sides <= 1 goal)
        // first: a wave's 64 consecutive fixtures then mostly agree on whether they have one, and the sliced
        // single launch (dyn_fused<true>) runs its log / reciprocal for a third of the waves, not for all
        for (int64_t b0 = 0; b0 < n;) {
            int64_t b1 = b0;
            while (b1 < n && b1 - b0 < 1024 && g[order[b1]] == g[order[b0]]) ++b1;
            std::stable_partition(order.begin() + b0, order.begin() + b1,
                                  [&](uint32_t p) { return x[p] <= 1 && y[p] <= 1; });
            b0 = b1;
        }
        permute(h); permute(a); permute(x); permute(y); permute(nv); permute(g);
    }
    const size_t GT = (size_t)n_teams * n_gameweeks;
    HIP_TRY(c, c->d_h.ensure(n * 2));
    HIP_TRY(c, c->d_a.ensure(n * 2));
    HIP_TRY(c, c->d_x.ensure(n));
    HIP_TRY(c, c->d_y.ensure(n));
    HIP_TRY(c, c->dd_gw.ensure(n * 2));
    HIP_TRY(c, c->dd_nv.ensure(n));
    HIP_TRY(c, hipMemcpyAsync(c->d_h.p, h.data(), n * 2, hipMemcpyHostToDevice, s));
    HIP_TRY(c, hipMemcpyAsync(c->d_a.p, a.data(), n * 2, hipMemcpyHostToDevice, s));
    HIP_TRY(c, hipMemcpyAsync(c->d_x.p, x.data(), n, hipMemcpyHostToDevice, s));
    HIP_TRY(c, hipMemcpyAsync(c->d_y.p, y.data(), n, hipMemcpyHostToDevice, s));
    HIP_TRY(c, hipMemcpyAsync(c->dd_gw.p, g.data(), n * 2, hipMemcpyHostToDevice, s));
    HIP_TRY(c, hipMemcpyAsync(c->dd_nv.p, nv.data(), n, hipMemcpyHostToDevice, s));
    std::vector<unsigned long long> fx8(n);   // (kept alive until the stream is synchronised below)
    for (int64_t i = 0; i < n; ++i) fx8[i] = dcd::pack_fixture(h[i], a[i], x[i], y[i], nv[i]);
    HIP_TRY(c, c->dd_fx8.ensure(n * 8));
    HIP_TRY(c, hipMemcpyAsync(c->dd_fx8.p, fx8.data(), n * 8, hipMemcpyHostToDevice, s));
    {   // incidence lists of the cells (gameweek, team) over the sorted fixtures: dyn_fused<false>'s gather
        std::vector<int> off(GT + 1, 0);
        for (int64_t i = 0; i < n; ++i) {
            off[(size_t)g[i] * n_teams + h[i] + 1] += 1;
            off[(size_t)g[i] * n_teams + a[i] + 1] += 1;
        }
        int longest = 0;
        for (size_t cidx = 0; cidx < GT; ++cidx) {
            longest = std::max(longest, off[cidx + 1]);
            off[cidx + 1] += off[cidx];
        }
        c->dyn_gather = longest <= dcd::GATHER_MAX_INCIDENT && n < (1ll << 30);
        if (c->dyn_gather) {
            std::vector<unsigned int> inc(std::max<size_t>(2 * (size_t)n, 1));
            std::vector<int> fill(off.begin(), off.end() - 1);
            for (int64_t i = 0; i < n; ++i) {   // (fixture order within a list: ascending, the summation order)
                const unsigned int venue = nv[i] ? 1u << 30 : 0u;
                inc[fill[(size_t)g[i] * n_teams + h[i]]++] = (unsigned int)i | venue;
                inc[fill[(size_t)g[i] * n_teams + a[i]]++] = (unsigned int)i | venue | (1u << 31);
            }
            HIP_TRY(c, c->dd_inc_off.ensure(off.size() * 4));
            HIP_TRY(c, c->dd_inc.ensure(inc.size() * 4));
            HIP_TRY(c, c->dd_fadj.ensure((size_t)n * 16));
            HIP_TRY(c, hipMemcpy(c->dd_inc_off.p, off.data(), off.size() * 4, hipMemcpyHostToDevice));
            HIP_TRY(c, hipMemcpy(c->dd_inc.p, inc.data(), inc.size() * 4, hipMemcpyHostToDevice));
        }
    }
    HIP_TRY(c, c->dd_cells.ensure(GT * dcd::P_N * 8));
    HIP_TRY(c, c->dd_acc.ensure(dcd::scratch_doubles(n_gameweeks, n_teams, k) * 8));
    // the single-launch kernel finds its scratch zeroed and its cell records armed (and leaves them so):
    // done here, so that a launch sequence captured right after this call holds no fills
    HIP_TRY(c, hipMemsetAsync(c->dd_acc.p, 0, dcd::scratch_doubles(n_gameweeks, n_teams, k) * 8, s));
    HIP_TRY(c, hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(c->dd_cells.p), (int)dcd::CELL_EMPTY_WORD,
                                 GT * dcd::P_N * 2, s));
    c->dyn_scratch_clean = true;
    c->dyn_big_lds = 0;   // (the cached occupancy belongs to one kernel and one LDS size)
    HIP_TRY(c, c->dd_hyp.ensure((size_t)6 * n_gameweeks * 8));
    {   // first fixture of every gameweek (sorted), and the two arrival counters
        std::vector<int> gw_off(n_gameweeks + 1, 0);
        for (int64_t i = 0; i < n; ++i) gw_off[g[i] + 1] += 1;
        c->dyn_max_gw = 0;
        for (int j = 0; j < n_gameweeks; ++j) c->dyn_max_gw = std::max<long long>(c->dyn_max_gw, gw_off[j + 1]);
        for (int j = 0; j < n_gameweeks; ++j) gw_off[j + 1] += gw_off[j];
        HIP_TRY(c, c->dd_gwoff.ensure(gw_off.size() * 4));
        HIP_TRY(c, hipMemcpy(c->dd_gwoff.p, gw_off.data(), gw_off.size() * 4, hipMemcpyHostToDevice));
        HIP_TRY(c, c->dd_tick.ensure(dcd::TICKET_BYTES));
        HIP_TRY(c, hipMemset(c->dd_tick.p, 0, dcd::TICKET_BYTES));
        HIP_TRY(c, hipMemset(c->dd_acc.p, 0, dcd::scratch_doubles(n_gameweeks, n_teams, k) * 8));
    }
    c->h_xs.clear();
    if (k > 0) {
        c->h_xs.assign(covariates, covariates + (size_t)n_teams * k);
        HIP_TRY(c, c->d_xs.ensure((size_t)n_teams * k * 8));
        HIP_TRY(c, hipMemcpyAsync(c->d_xs.p, c->h_xs.data(), (size_t)n_teams * k * 8,
                                  hipMemcpyHostToDevice, s));
    }
    HIP_TRY(c, hipStreamSynchronize(s));
    drop_graphs(c);
    c->DL = dcd::make_dyn_layout(n_gameweeks, n_teams, k);
    c->dyn_random_walk = random_walk != 0;
    c->n = n;
    c->lgsum = lgsum;
    c->dynamic = true;
    c->bound = true;
    return BPLHIP_OK;
}

static int bplhip_logp_grad_batched_impl(bplhip_ctx* c, int32_t n_chains, const double* z,
                             double* potential, double* grad, double* aux, void* stream) {
    if (!c) return BPLHIP_EINVAL;
    if (!c->bound) return fail(c, BPLHIP_ESTATE, "logp_grad: no fixtures bound");
    if (!z || !potential || !grad) return fail(c, BPLHIP_EINVAL, "logp_grad: null pointer");
    if (n_chains < 1 || n_chains > 65535)
        return fail(c, BPLHIP_EINVAL, "logp_grad: n_chains=%d out of range", n_chains);
    HIP_TRY(c, hipSetDevice(c->device));
    if (use_vec(c, n_chains))
        return launch_eval_vec(c, n_chains, z, potential, grad, aux, static_cast<hipStream_t>(stream));
    if (!c->dynamic && !c->neutral) {
        int rc = ensure_slabs(c, n_chains);
        if (rc != BPLHIP_OK) return rc;
    }
    return launch_eval(c, n_chains, z, potential, grad, aux, static_cast<hipStream_t>(stream));
}

int bplhip_logp_grad(bplhip_ctx* c, const double* z, double* potential, double* grad,
                     double* aux, void* stream) {
    return bplhip_logp_grad_batched(c, 1, z, potential, grad, aux, stream);
}

static int bplhip_logp_grad_graph_impl(bplhip_ctx* c, int32_t count, int32_t n_z, const double* z,
                           double* potential, double* grad, int32_t replays, void* stream) {
    if (!c) return BPLHIP_EINVAL;
    if (!c->bound) return fail(c, BPLHIP_ESTATE, "logp_grad_graph: no fixtures bound");
    if (!z || !potential || !grad || count < 1 || n_z < 1 || replays < 0)
        return fail(c, BPLHIP_EINVAL, "logp_grad_graph: bad argument");
    HIP_TRY(c, hipSetDevice(c->device));
    const GraphKey key{count, n_z, z, potential, grad};
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int D = bplhip_latent_dim(c);
    auto enqueue = [&](hipStream_t st) -> int {
        int rc = BPLHIP_OK;
        for (int i = 0; i < count && rc == BPLHIP_OK; ++i) {
            const int j = i % n_z;
            rc = launch_eval(c, 1, z + (size_t)j * D, potential + j, grad + (size_t)j * D, nullptr, st);
        }
        return rc;
    };
    auto it = c->graphs.find(key);
    if (it == c->graphs.end()) {
        // one un-captured evaluation first: whatever the launch path sets up lazily happens outside the capture
        int rc = launch_eval(c, 1, z, potential, grad, nullptr, s);
        if (rc != BPLHIP_OK) return rc;
        hipGraphExec_t ge = nullptr;
        rc = capture_launches(c, enqueue, &ge);
        if (rc != BPLHIP_OK) return rc;
        if (!ge) {  // no graph on this runtime: the same launches, one by one
            for (int r = 0; r < replays; ++r) {
                rc = enqueue(s);
                if (rc != BPLHIP_OK) return rc;
            }
            return BPLHIP_OK;
        }
        it = c->graphs.emplace(key, ge).first;
    }
    for (int r = 0; r < replays; ++r) HIP_TRY(c, hipGraphLaunch(it->second, s));
    return BPLHIP_OK;
}

#ifdef DC_STAMPS
// diagnostic build only: allocate / read the per-workgroup timestamp buffer [n_wg][16]
int bplhip_debug_stamps(bplhip_ctx* c, unsigned long long* out, int n_words) {
    if (!c || !c->bound) return BPLHIP_ESTATE;
    const size_t bytes = (size_t)(c->ep->n_wg + 2) * 16 * 8;
    if (!c->d_debug.p) {
        HIP_TRY(c, c->d_debug.ensure(bytes));
        HIP_TRY(c, hipMemset(c->d_debug.p, 0, bytes));
        return c->ep->n_wg + 1;
    }
    HIP_TRY(c, hipDeviceSynchronize());
    const size_t want = std::min(bytes, (size_t)n_words * 8);
    HIP_TRY(c, hipMemcpy(out, c->d_debug.p, want, hipMemcpyDeviceToHost));
    return c->ep->n_wg + 1;
}
#endif

void bplhip_threefry_split(uint32_t key_hi, uint32_t key_lo, int32_t n, uint32_t* out) {
    if (n <= 0 || !out) return;
    std::vector<tf::Key> ks(n);
    tf::split({key_hi, key_lo}, n, ks.data());
    for (int i = 0; i < n; ++i) {
        out[2 * i] = ks[i].hi;
        out[2 * i + 1] = ks[i].lo;
    }
}

void bplhip_threefry_bits(uint32_t key_hi, uint32_t key_lo, int32_t n, uint32_t* out) {
    if (n <= 0 || !out) return;
    tf::random_bits({key_hi, key_lo}, n, out);
}

void bplhip_nuts_default_cfg(bplhip_nuts_cfg* cfg) {
    if (!cfg) return;
    cfg->num_warmup = 500;
    cfg->num_samples = 1000;
    cfg->max_tree_depth = 10;
    cfg->adapt_step_size = 1;
    cfg->adapt_mass_matrix = 1;
    cfg->thinning = 1;
    cfg->step_size = 1.0;
    cfg->target_accept_prob = 0.8;
    cfg->init_radius = 2.0;
    cfg->max_delta_energy = 1000.0;
}

}  // extern "C"

// ------------------------------------------------------------------ NUTS glue + constrain

namespace {

// Potential functor handed to the templated NUTS driver: one launch sequence + one
// pinned-host read-back per evaluation.
struct HipPotential {
    bplhip_ctx* c;
    hipStream_t s;
    int D;
    double *d_z, *d_pot, *d_grad, *d_aux;  // device
    double* hp;                            // pinned: [D grad | pot | aux4]
    int rc = BPLHIP_OK;
    int dim() const { return D; }
    // returns false on a HIP failure (rc holds the code)
    bool operator()(const double* z, double* U, double* grad, double* aux) {
        if (hipMemcpyAsync(d_z, z, (size_t)D * 8, hipMemcpyHostToDevice, s) != hipSuccess) {
            rc = fail(c, BPLHIP_EHIP, "nuts: H2D of z failed");
            return false;
        }
        rc = launch_eval(c, 1, d_z, d_pot, d_grad, d_aux, s);
        if (rc != BPLHIP_OK) return false;
        // d_grad, d_pot, d_aux are contiguous: one D2H
        if (hipMemcpyAsync(hp, d_grad, (size_t)(D + 1 + 4) * 8, hipMemcpyDeviceToHost, s) !=
                hipSuccess ||
            hipStreamSynchronize(s) != hipSuccess) {
            rc = fail(c, BPLHIP_EHIP, "nuts: D2H / sync failed: %s",
                      hipGetErrorString(hipGetLastError()));
            return false;
        }
        std::memcpy(grad, hp, (size_t)D * 8);
        *U = hp[D];
        if (aux) std::memcpy(aux, hp + D + 1, 4 * 8);
        return true;
    }
};

}  // namespace

namespace {

// Tree builder on the device: per doubling the host enqueues k_begin, 2^j dc_eval launches
// (leaf bookkeeping in their tails) and k_end, and synchronises once per batch of
// doublings; the first batch of a transition speculates on the previous tree's depth.
struct DeviceEngine {
    bplhip_ctx* c;
    hipStream_t s;
    const nuts::Config& cfg;
    int D, max_depth;
    double* ns;        // device state (nuts_dev.hip.h)
    double* hp;        // pinned: [r: D | hdr: H_N | z: D]
    int64_t evals = 0;
    int last_depth = 1;
    int rc = BPLHIP_OK;
    int dim() const { return D; }
    int64_t leapfrogs() const { return evals; }
    bool hip_ok(hipError_t e, const char* what) {
        if (e == hipSuccess) return true;
        rc = fail(c, BPLHIP_EHIP, "device nuts: %s: %s", what, hipGetErrorString(e));
        return false;
    }
    bool set_state(const double* z, double* pe_out, bool* finite) {
        std::memcpy(hp, z, (size_t)D * 8);
        double* zn = nd::vec(ns, D, nd::V_ZN);
        double* gr = nd::vec(ns, D, nd::V_GRAD);
        if (!hip_ok(hipMemcpyAsync(zn, hp, (size_t)D * 8, hipMemcpyHostToDevice, s), "H2D z")) return false;
        rc = launch_eval(c, 1, zn, ns + nd::H_LEAF_PE, gr, ns + nd::H_LEAF_AUX0, s);
        if (rc != BPLHIP_OK) return false;
        // adopt as the current state of the chain
        (void)hipMemcpyAsync(nd::vec(ns, D, nd::V_Z), zn, (size_t)D * 8, hipMemcpyDeviceToDevice, s);
        (void)hipMemcpyAsync(nd::vec(ns, D, nd::V_G), gr, (size_t)D * 8, hipMemcpyDeviceToDevice, s);
        (void)hipMemcpyAsync(ns + nd::H_CUR_PE, ns + nd::H_LEAF_PE, 8, hipMemcpyDeviceToDevice, s);
        (void)hipMemcpyAsync(ns + nd::H_T_AUX0, ns + nd::H_LEAF_AUX0, 32, hipMemcpyDeviceToDevice, s);
        double* h_hdr = hp + D;
        (void)hipMemcpyAsync(h_hdr, ns, (size_t)nd::H_N * 8, hipMemcpyDeviceToHost, s);
        (void)hipMemcpyAsync(hp + D + nd::H_N, gr, (size_t)D * 8, hipMemcpyDeviceToHost, s);
        if (!hip_ok(hipStreamSynchronize(s), "sync(set_state)")) return false;
        const double pe = h_hdr[nd::H_LEAF_PE];
        bool ok = std::isfinite(pe);
        for (int i = 0; i < D && ok; ++i) ok = std::isfinite(hp[D + nd::H_N + i]);
        *finite = ok;
        *pe_out = pe;
        evals += 1;
        return true;
    }
    bool transition(const double* r, tf::Key key, double step_size, const nuts::vec& inv_mass,
                    bool mass_changed, double* z_out, nuts::TransitionOut* out) {
        double* h_hdr = hp + D;
        if (mass_changed) {
            // (pinned staging reused: the copy is ordered before the momentum copy)
            std::memcpy(hp + D + nd::H_N, inv_mass.data(), (size_t)D * 8);
            if (!hip_ok(hipMemcpyAsync(nd::vec(ns, D, nd::V_INVM), hp + D + nd::H_N, (size_t)D * 8,
                                       hipMemcpyHostToDevice, s), "H2D inv_mass")) return false;
        }
        std::memcpy(hp, r, (size_t)D * 8);
        if (!hip_ok(hipMemcpyAsync(nd::vec(ns, D, nd::V_TL_R), hp, (size_t)D * 8,
                                   hipMemcpyHostToDevice, s), "H2D r")) return false;
        hipLaunchKernelGGL(nd::k_init, dim3(1), dim3(64), 0, s, ns, D, step_size, cfg.max_delta_energy);
        double* zn = nd::vec(ns, D, nd::V_ZN);
        double* gr = nd::vec(ns, D, nd::V_GRAD);
        int depth = 0;
        bool stop = false;
        bool first_batch = true;
        while (!stop && depth < max_depth) {
            int nb = first_batch ? std::max(1, std::min(last_depth, max_depth)) : 1;
            nb = std::min(nb, max_depth - depth);
            for (int j = depth; j < depth + nb; ++j) {
                // numpyro build_tree: key, direction_key, doubling_key = split(key, 3);
                // _double_tree: key, transition_key = split(doubling_key)
                tf::Key k_next, k_dir, k_dbl, k_sub, k_tr;
                tf::split3(key, &k_next, &k_dir, &k_dbl);
                key = k_next;
                const int going_right = tf::bernoulli(k_dir, 0.5) ? 1 : 0;
                tf::split2(k_dbl, &k_sub, &k_tr);
                hipLaunchKernelGGL(nd::k_begin, dim3(1), dim3(64), 0, s, ns, D, j, going_right,
                                   k_sub.hi, k_sub.lo);
                const int leaves = 1 << j;
                for (int l = 0; l < leaves; ++l) {
                    rc = launch_eval(c, 1, zn, ns + nd::H_LEAF_PE, gr, ns + nd::H_LEAF_AUX0, s, ns,
                                     max_depth);
                    if (rc != BPLHIP_OK) return false;
                }
                hipLaunchKernelGGL(nd::k_end, dim3(1), dim3(64), 0, s, ns, D, max_depth, k_tr.hi,
                                   k_tr.lo);
            }
            // one read-back per batch: header + the tree's current proposal (final if STOP)
            (void)hipMemcpyAsync(h_hdr, ns, (size_t)nd::H_N * 8, hipMemcpyDeviceToHost, s);
            (void)hipMemcpyAsync(hp + D + nd::H_N, nd::vec(ns, D, nd::V_TP_Z), (size_t)D * 8,
                                 hipMemcpyDeviceToHost, s);
            if (!hip_ok(hipStreamSynchronize(s), "sync(doubling)")) return false;
            first_batch = false;
            depth = (int)h_hdr[nd::H_T_DEPTH];
            stop = h_hdr[nd::H_STOP] != 0.0;
        }
        // the proposal becomes the chain's current state (no host wait needed)
        hipLaunchKernelGGL(nd::k_finish, dim3(1), dim3(64), 0, s, ns, D);
        if (!hip_ok(hipGetLastError(), "launch")) return false;
        const double num = h_hdr[nd::H_T_NUM];
        out->accept_prob = num > 0 ? h_hdr[nd::H_T_SUMACC] / num : 0.0;
        out->num_steps = (int)num;
        out->diverging = h_hdr[nd::H_T_DIV] != 0.0;
        out->pe = h_hdr[nd::H_T_PE];
        out->aux[0] = h_hdr[nd::H_T_AUX0];
        out->aux[1] = h_hdr[nd::H_T_AUX1];
        out->aux[2] = h_hdr[nd::H_T_AUX2];
        out->aux[3] = h_hdr[nd::H_T_AUX3];
        std::memcpy(z_out, hp + D + nd::H_N, (size_t)D * 8);
        evals += (int64_t)num;
        last_depth = std::max(1, depth);
        return true;
    }
};


// Lock-step chains (numpyro chain_method="vectorized"): all chains' trees are built
// together; every leapfrog of the batch is ONE chain-vectorised evaluation (dc_vec.hip.h)
// whose tail books each chain's leaf.  Chains whose subtree / tree is complete idle.
struct VecDeviceEngine {
    bplhip_ctx* c;
    hipStream_t s;
    const nuts::Config& cfg;
    int D, max_depth, C;
    double* ns;       // device [C][stride]
    size_t stride;
    double* par;      // device [C][par_stride]
    int par_stride;
    // pinned staging: r [C][D] | inv_mass [C][D] | hdr [C][H_N] | prop [C][D] | par [C][par_stride]
    double *h_r, *h_m, *h_hdr, *h_prop, *h_par;
    int last_depth = 1;
    int rc = BPLHIP_OK;
    int dim() const { return D; }
    bool hip_ok(hipError_t e, const char* what) {
        if (e == hipSuccess) return true;
        rc = fail(c, BPLHIP_EHIP, "lock-step nuts: %s: %s", what, hipGetErrorString(e));
        return false;
    }
    bool set_state(int ch, const double* z, double* pe_out, bool* finite) {
        double* nsc = ns + (size_t)ch * stride;
        std::memcpy(h_r, z, (size_t)D * 8);
        double* zn = nd::vec(nsc, D, nd::V_ZN);
        double* gr = nd::vec(nsc, D, nd::V_GRAD);
        if (!hip_ok(hipMemcpyAsync(zn, h_r, (size_t)D * 8, hipMemcpyHostToDevice, s), "H2D z")) return false;
        rc = launch_eval(c, 1, zn, nsc + nd::H_LEAF_PE, gr, nsc + nd::H_LEAF_AUX0, s);
        if (rc != BPLHIP_OK) return false;
        (void)hipMemcpyAsync(nd::vec(nsc, D, nd::V_Z), zn, (size_t)D * 8, hipMemcpyDeviceToDevice, s);
        (void)hipMemcpyAsync(nd::vec(nsc, D, nd::V_G), gr, (size_t)D * 8, hipMemcpyDeviceToDevice, s);
        (void)hipMemcpyAsync(nsc + nd::H_CUR_PE, nsc + nd::H_LEAF_PE, 8, hipMemcpyDeviceToDevice, s);
        (void)hipMemcpyAsync(nsc + nd::H_T_AUX0, nsc + nd::H_LEAF_AUX0, 32, hipMemcpyDeviceToDevice, s);
        (void)hipMemcpyAsync(h_hdr, nsc, (size_t)nd::H_N * 8, hipMemcpyDeviceToHost, s);
        (void)hipMemcpyAsync(h_prop, gr, (size_t)D * 8, hipMemcpyDeviceToHost, s);
        if (!hip_ok(hipStreamSynchronize(s), "sync(set_state)")) return false;
        const double pe = h_hdr[nd::H_LEAF_PE];
        bool ok = std::isfinite(pe);
        for (int i = 0; i < D && ok; ++i) ok = std::isfinite(h_prop[i]);
        *finite = ok;
        *pe_out = pe;
        return true;
    }
    bool transition_all(std::vector<nuts::ChainDriver>& cds, std::vector<nuts::TransitionOut>* outs) {
        bool any_mass = false;
        for (int ch = 0; ch < C; ++ch) {
            nuts::ChainDriver& cd = cds[ch];
            std::memcpy(h_r + (size_t)ch * D, cd.r.data(), (size_t)D * 8);
            std::memcpy(h_m + (size_t)ch * D, cd.inv_mass.data(), (size_t)D * 8);
            any_mass = any_mass || cd.mass_changed;
            // numpyro build_tree: key, direction_key, doubling_key = split(key, 3);
            // _double_tree: key, transition_key = split(doubling_key) -- data independent,
            // so all doublings' directions and keys are known up front
            double* p = h_par + (size_t)ch * par_stride;
            p[0] = cd.step_size;
            tf::Key key = cd.k_tr;
            for (int j = 0; j < max_depth; ++j) {
                tf::Key k_next, k_dir, k_dbl, k_sub, k_tr;
                tf::split3(key, &k_next, &k_dir, &k_dbl);
                key = k_next;
                tf::split2(k_dbl, &k_sub, &k_tr);
                double* q = p + 1 + 5 * j;
                q[0] = tf::bernoulli(k_dir, 0.5) ? 1.0 : 0.0;
                q[1] = (double)k_sub.hi; q[2] = (double)k_sub.lo;
                q[3] = (double)k_tr.hi;  q[4] = (double)k_tr.lo;
            }
        }
        const size_t pitch = stride * 8, row = (size_t)D * 8;
        if (any_mass &&
            !hip_ok(hipMemcpy2DAsync(nd::vec(ns, D, nd::V_INVM), pitch, h_m, row, row, C,
                                     hipMemcpyHostToDevice, s), "H2D inv_mass")) return false;
        if (!hip_ok(hipMemcpy2DAsync(nd::vec(ns, D, nd::V_TL_R), pitch, h_r, row, row, C,
                                     hipMemcpyHostToDevice, s), "H2D r")) return false;
        if (!hip_ok(hipMemcpyAsync(par, h_par, (size_t)C * par_stride * 8, hipMemcpyHostToDevice, s),
                    "H2D par")) return false;
        hipLaunchKernelGGL(nd::kv_init, dim3(C), dim3(64), 0, s, ns, stride, D, par, par_stride,
                           cfg.max_delta_energy);
        int depth = 0;
        bool stop = false, first_batch = true;
        while (!stop && depth < max_depth) {
            int nb = first_batch ? std::max(1, std::min(last_depth, max_depth)) : 1;
            nb = std::min(nb, max_depth - depth);
            for (int j = depth; j < depth + nb; ++j) {
                hipLaunchKernelGGL(nd::kv_begin, dim3(C), dim3(64), 0, s, ns, stride, D, j, par,
                                   par_stride);
                const int leaves = 1 << j;
                for (int l = 0; l < leaves; ++l) {
                    rc = launch_eval_vec(c, C, nd::vec(ns, D, nd::V_ZN), ns + nd::H_LEAF_PE,
                                         nd::vec(ns, D, nd::V_GRAD), ns + nd::H_LEAF_AUX0, s, ns,
                                         (int)stride, max_depth);
                    if (rc != BPLHIP_OK) return false;
                }
                hipLaunchKernelGGL(nd::kv_end, dim3(C), dim3(64), 0, s, ns, stride, D, j, max_depth,
                                   par, par_stride);
            }
            // one read-back per batch: every chain's header + current proposal
            (void)hipMemcpy2DAsync(h_hdr, (size_t)nd::H_N * 8, ns, pitch, (size_t)nd::H_N * 8, C,
                                   hipMemcpyDeviceToHost, s);
            (void)hipMemcpy2DAsync(h_prop, row, nd::vec(ns, D, nd::V_TP_Z), pitch, row, C,
                                   hipMemcpyDeviceToHost, s);
            if (!hip_ok(hipStreamSynchronize(s), "sync(doubling)")) return false;
            first_batch = false;
            depth += nb;
            stop = true;
            for (int ch = 0; ch < C; ++ch) stop = stop && h_hdr[(size_t)ch * nd::H_N + nd::H_STOP] != 0.0;
        }
        hipLaunchKernelGGL(nd::kv_finish, dim3(C), dim3(64), 0, s, ns, stride, D);
        if (!hip_ok(hipGetLastError(), "launch")) return false;
        int dmax = 1;
        for (int ch = 0; ch < C; ++ch) {
            const double* hd = h_hdr + (size_t)ch * nd::H_N;
            nuts::TransitionOut& o = (*outs)[ch];
            const double num = hd[nd::H_T_NUM];
            o.accept_prob = num > 0 ? hd[nd::H_T_SUMACC] / num : 0.0;
            o.num_steps = (int)num;
            o.diverging = hd[nd::H_T_DIV] != 0.0;
            o.pe = hd[nd::H_T_PE];
            for (int i = 0; i < 4; ++i) o.aux[i] = hd[nd::H_T_AUX0 + i];
            std::memcpy(cds[ch].z.data(), h_prop + (size_t)ch * D, (size_t)D * 8);
            dmax = std::max(dmax, (int)hd[nd::H_T_DEPTH]);
        }
        last_depth = dmax;
        return true;
    }
};

// Persistent chains (nuts_dev.hip.h): the host finds the initial states, generates every
// chain's random inputs (they are data independent), uploads them, and from then on only
// enqueues evaluations -- C == 1: the single-chain kernel, else the chain-vectorised one --
// checking the chains' "all done" flags once per chunk of launches.

// the leaf can run in dc_eval's tail (nuts_dev.hip.h): the basic / extended models, <= 64 teams
static bool leaf_in_tail(const bplhip_ctx* c) {
    bool staged = true;
    for (int pi = 0; pi < c->n_parts; ++pi) staged = staged && c->parts[pi].staged;
    return !c->neutral && !c->dynamic && c->L.T <= 64 && staged && c->L.D <= 64 * nd::LEAF_NE_MAX;
}
// persistent chains keep all momentum draws of a run on the device: [C][n_iter][D] doubles
// ... and the wide leaf's grid row (kw_leaf: its workgroups poll each other's records and, when a
// subtree is complete, meet at row barriers) must fit the device at once.  Workgroups are dispatched in
// order, so a row never waits for a later one -- but a row that does not fit would wait for itself.
static bool persistent_fits(const bplhip_ctx* c, const bplhip_nuts_cfg* cfg, int C) {
    const double bytes = (double)C * (cfg->num_warmup + cfg->num_samples) * bplhip_latent_dim(c) * 8.0;
    if (nd::kw_workgroups(bplhip_latent_dim(c), nd::KW_NTB) > c->n_cu) return false;
    return bytes <= 16.0 * 1024 * 1024 * 1024;
}

int run_chains_persistent(bplhip_ctx* c, hipStream_t s, const nuts::Config& nc, int C, const double* z0,
                          const tf::Key* keys, double* draws_out, std::vector<nuts::Result>* res) {
    const int D = bplhip_latent_dim(c), md = nc.max_tree_depth;
    // the leaf as its own launch(es) after a plain evaluation: one wave per chain (kp_leaf) up
    // to 256 latent entries, GW workgroups per chain (kw_leaf) beyond
    const bool generic = !leaf_in_tail(c);
    const bool wide = generic && D > 64 * nd::LEAF_NE_MAX;
    const int GWB = nd::kw_workgroups(D, nd::KW_NTB);
    const int n_iter = nc.num_warmup + nc.num_samples;
    const int kept = nc.num_samples / nc.thinning;
    const size_t nsd = (nd::ns_doubles(D, md) + 1) & ~(size_t)1;
    const size_t stride = (nsd + nd::pd_doubles(D) + 1) & ~(size_t)1;
    const std::vector<nuts::Window> sched = nuts::build_adaptation_schedule(nc.num_warmup);
    std::vector<int> win_end;
    for (const auto& w : sched) win_end.push_back(w.end);
    if (win_end.empty()) win_end.push_back(-1);

    DevBuf d_ns, d_norm, d_par, d_win, d_draws, d_stats, d_desc, d_part, d_tick, d_rowpart;
    if (wide) {
        HIP_TRY(c, d_part.ensure((size_t)C * GWB * nd::KW_PW * sizeof(nd::TaggedSum)));
        HIP_TRY(c, hipMemsetAsync(d_part.p, 0, (size_t)C * GWB * nd::KW_PW * sizeof(nd::TaggedSum), s));   // (tag 0: nobody's)
        HIP_TRY(c, d_tick.ensure((size_t)C * nd::RT_WORDS * 4));   // per chain: ticket, row barrier, exit counter
        HIP_TRY(c, hipMemsetAsync(d_tick.p, 0, (size_t)C * nd::RT_WORDS * 4, s));
        HIP_TRY(c, d_rowpart.ensure((size_t)C * nd::row_part_doubles() * 8));
    }
    HIP_TRY(c, d_ns.ensure((size_t)C * stride * 8));
    HIP_TRY(c, hipMemsetAsync(d_ns.p, 0, (size_t)C * stride * 8, s));
    HIP_TRY(c, d_norm.ensure((size_t)C * n_iter * D * 8));
    HIP_TRY(c, d_par.ensure((size_t)C * n_iter * md * 5 * 8));
    HIP_TRY(c, d_win.ensure(win_end.size() * 4));
    HIP_TRY(c, d_draws.ensure((size_t)C * kept * D * 8));
    HIP_TRY(c, d_stats.ensure((size_t)C * kept * 6 * 8));
    HIP_TRY(c, d_desc.ensure(sizeof(nd::Persist)));
    double* ns = d_ns.as<double>();

    // ---- initial states (ChainDriver::init: key plumbing + init_to_uniform with retries)
    res->assign(C, nuts::Result{});
    std::vector<nuts::ChainDriver> cds;
    cds.reserve(C);
    for (int ch = 0; ch < C; ++ch) cds.emplace_back(nc, D, draws_out + (size_t)ch * kept * D, &(*res)[ch]);
    std::vector<double> stage(std::max<size_t>((size_t)D + nd::H_N, 64));
    int rc = BPLHIP_OK;
    for (int ch = 0; ch < C; ++ch) {
        double* nsc = ns + (size_t)ch * stride;
        auto set_state = [&](const double* zz, double* pe, bool* fin) {
            double* zn = nd::vec(nsc, D, nd::V_ZN);
            double* gr = nd::vec(nsc, D, nd::V_GRAD);
            if (hipMemcpyAsync(zn, zz, (size_t)D * 8, hipMemcpyHostToDevice, s) != hipSuccess) return false;
            rc = launch_eval(c, 1, zn, nsc + nd::H_LEAF_PE, gr, nsc + nd::H_LEAF_AUX0, s);
            if (rc != BPLHIP_OK) return false;
            (void)hipMemcpyAsync(nd::vec(nsc, D, nd::V_Z), zn, (size_t)D * 8, hipMemcpyDeviceToDevice, s);
            (void)hipMemcpyAsync(nd::vec(nsc, D, nd::V_G), gr, (size_t)D * 8, hipMemcpyDeviceToDevice, s);
            (void)hipMemcpyAsync(nsc + nd::H_CUR_PE, nsc + nd::H_LEAF_PE, 8, hipMemcpyDeviceToDevice, s);
            (void)hipMemcpyAsync(nsc + nd::H_T_AUX0, nsc + nd::H_LEAF_AUX0, 32, hipMemcpyDeviceToDevice, s);
            (void)hipMemcpyAsync(stage.data(), nsc + nd::H_LEAF_PE, 8, hipMemcpyDeviceToHost, s);
            (void)hipMemcpyAsync(stage.data() + 1, gr, (size_t)D * 8, hipMemcpyDeviceToHost, s);
            if (hipStreamSynchronize(s) != hipSuccess) return false;
            bool ok = std::isfinite(stage[0]);
            for (int i = 0; i < D && ok; ++i) ok = std::isfinite(stage[1 + i]);
            *fin = ok;
            *pe = stage[0];
            return true;
        };
        const int st = cds[ch].init(set_state, z0 ? z0 + (size_t)ch * D : nullptr, keys[ch]);
        if (st == nuts::ST_EVAL_FAILED) return rc != BPLHIP_OK ? rc : fail(c, BPLHIP_EHIP, "persistent nuts: init failed");
        if (st == nuts::ST_NO_FINITE_INIT) return BPLHIP_ENUMERIC;
    }

    // ---- all random inputs of every chain (sample_kernel / build_tree / _double_tree splits)
    {
        std::vector<double> nrm((size_t)C * n_iter * D), par((size_t)C * n_iter * md * 5);
        std::vector<tf::Key> k_moms((size_t)C * n_iter);
        for (int ch = 0; ch < C; ++ch) {
            tf::Key key_hmc = cds[ch].key_hmc;
            for (int it = 0; it < n_iter; ++it) {
                tf::Key k_mom, k_tr;
                tf::split3(key_hmc, &key_hmc, &k_mom, &k_tr);
                k_moms[(size_t)ch * n_iter + it] = k_mom;
                tf::Key key = k_tr;
                for (int j = 0; j < md; ++j) {
                    tf::Key k_next, k_dir, k_dbl, k_sub, k_t2;
                    tf::split3(key, &k_next, &k_dir, &k_dbl);
                    key = k_next;
                    tf::split2(k_dbl, &k_sub, &k_t2);
                    double* q = par.data() + (((size_t)ch * n_iter + it) * md + j) * 5;
                    q[0] = tf::bernoulli(k_dir, 0.5) ? 1.0 : 0.0;
                    q[1] = (double)k_sub.hi; q[2] = (double)k_sub.lo;
                    q[3] = (double)k_t2.hi;  q[4] = (double)k_t2.lo;
                }
            }
        }
        {   // the momentum draws themselves (C * n_iter * D normals): host threads for long vectors
            const size_t jobs = (size_t)C * n_iter;
            unsigned nthr = nrm.size() > (size_t)1 << 20 ? std::thread::hardware_concurrency() : 1;
            nthr = std::max(1u, std::min(nthr, 32u));
            auto work = [&](unsigned t) {
                for (size_t j = t; j < jobs; j += nthr) tf::normal(k_moms[j], D, nrm.data() + j * D);
            };
            if (nthr == 1) {
                work(0);
            } else {
                std::vector<std::thread> pool;
                for (unsigned t = 0; t < nthr; ++t) pool.emplace_back(work, t);
                for (auto& th : pool) th.join();
            }
        }
        HIP_TRY(c, hipMemcpy(d_norm.p, nrm.data(), nrm.size() * 8, hipMemcpyHostToDevice));
        HIP_TRY(c, hipMemcpy(d_par.p, par.data(), par.size() * 8, hipMemcpyHostToDevice));
        HIP_TRY(c, hipMemcpy(d_win.p, win_end.data(), win_end.size() * 4, hipMemcpyHostToDevice));
    }
    nd::Persist P{};
    P.normals = d_norm.as<const double>();
    P.par = d_par.as<const double>();
    P.win_end = d_win.as<const int>();
    P.draws = d_draws.as<double>();
    P.stats = d_stats.as<double>();
    P.n_iter = n_iter; P.kept = kept; P.max_depth = md; P.n_win = (int)win_end.size(); P.D = D;
    P.pd_off = nsd;
    HIP_TRY(c, hipMemcpy(d_desc.p, &P, sizeof P, hipMemcpyHostToDevice));
    {   // PD blocks + identity mass matrices
        std::vector<double> pd(nd::pd_doubles(D), 0.0), ones(D, 1.0);
        pd[nd::P_WARM] = nc.num_warmup; pd[nd::P_TOTAL] = n_iter; pd[nd::P_STEP] = nc.step_size;
        pd[nd::P_DA_PROX] = std::log(10.0 * nc.step_size);
        pd[nd::P_NWIN] = (double)sched.size();
        pd[nd::P_ADAPT_SS] = nc.adapt_step_size; pd[nd::P_ADAPT_MM] = nc.adapt_mass_matrix;
        pd[nd::P_TARGET] = nc.target_accept_prob; pd[nd::P_THIN] = nc.thinning;
        pd[nd::P_START_IDX] = nc.num_warmup + nc.num_samples % nc.thinning;
        pd[nd::P_MAXDE] = nc.max_delta_energy;
        for (int i = 0; i < D; ++i) pd[nd::P_N + 2 * (size_t)D + i] = 1.0;  // mass_sqrt
        for (int ch = 0; ch < C; ++ch) {
            double* nsc = ns + (size_t)ch * stride;
            HIP_TRY(c, hipMemcpy(nsc + nsd, pd.data(), pd.size() * 8, hipMemcpyHostToDevice));
            HIP_TRY(c, hipMemcpy(nd::vec(nsc, D, nd::V_INVM), ones.data(), (size_t)D * 8, hipMemcpyHostToDevice));
        }
    }
    std::vector<double> evals0(C);
    HIP_TRY(c, hipMemcpy2D(evals0.data(), 8, ns + nd::H_EVALS, stride * 8, 8, C, hipMemcpyDeviceToHost));

    // ---- run: first transitions, then blind chunks of evaluations
    if (wide) hipLaunchKernelGGL(nd::kw_start, dim3(GWB, C), dim3(nd::KW_NTB), 0, s, ns, stride, P,
                                 d_tick.as<unsigned int>(), d_rowpart.as<double>(), c->d_fault);
    else hipLaunchKernelGGL(nd::kp_start, dim3(C), dim3(64), 0, s, ns, stride, P);
    const nd::Persist* dP = d_desc.as<const nd::Persist>();
    std::vector<double> flags(C);
    const bool looped = C == 1 && !generic && loop_ok(c);  // one resident launch per chunk
    const int chunk = looped ? 1024 : 256;
    bool all_done = false;
    // every launch advances every unfinished chain by one leapfrog: an upper bound exists
    const double max_steps = (double)n_iter * (double)((1u << md) - 1) + 2.0 * chunk;
    double steps_done = 0.0;
    // one leapfrog of every unfinished chain: the launches a chunk repeats `chunk` times (same arguments
    // every time: everything lives in the chains' state buffers)
    auto enqueue_leapfrogs = [&](hipStream_t st, int count) -> int {
        int erc = BPLHIP_OK;
        for (int k = 0; k < count; ++k) {
            if (generic && neutral_leaf_fusable(c)) {  // evaluation + leaf of every chain in ONE launch
                erc = launch_eval_neutral(c, C, nullptr, nullptr, nullptr, nullptr, st, ns, stride, md, &P);
                if (erc != BPLHIP_OK) return erc;
                continue;
            }
            if (generic) {
                for (int ch = 0; ch < C && erc == BPLHIP_OK; ++ch) {
                    double* nsc = ns + (size_t)ch * stride;
                    erc = launch_eval(c, 1, nd::vec(nsc, D, nd::V_ZN), nsc + nd::H_LEAF_PE,
                                      nd::vec(nsc, D, nd::V_GRAD), nsc + nd::H_LEAF_AUX0, st);
                }
                if (erc != BPLHIP_OK) return erc;
                if (wide) {
                    hipLaunchKernelGGL(nd::kw_leaf, dim3(GWB, C), dim3(nd::KW_NTB), 0, st, ns, stride, D, md,
                                       d_part.as<nd::TaggedSum>(), d_tick.as<unsigned int>(), P, d_rowpart.as<double>(),
                                       c->d_fault);
                } else {
                    hipLaunchKernelGGL(nd::kp_leaf, dim3(C), dim3(64), (size_t)(D + 8) * 8, st, ns, stride, D,
                                       md, P);
                }
                continue;
            }
            // up to `gridy_max_chains` chains run as grid.y copies of the single-chain launch
            // (62 workgroups each at N = 1e6; measured faster than sharing up to ~32 chains);
            // more chains share the chain-vectorised kernel
            erc = C <= c->opt_gridy_max_chains
                      ? launch_eval(c, C, nd::vec(ns, D, nd::V_ZN), ns + nd::H_LEAF_PE, nd::vec(ns, D, nd::V_GRAD),
                                    ns + nd::H_LEAF_AUX0, st, ns, md, dP, C > 1 ? (int)stride : 0)
                      : launch_eval_vec(c, C, nd::vec(ns, D, nd::V_ZN), ns + nd::H_LEAF_PE,
                                        nd::vec(ns, D, nd::V_GRAD), ns + nd::H_LEAF_AUX0, st, ns, (int)stride,
                                        md, dP);
            if (erc != BPLHIP_OK) return erc;
        }
        return BPLHIP_OK;
    };
    // The chunk as ONE hipGraph, captured once per run and replayed (option chunk_graph, default on): the
    // launches of a chunk are identical, and enqueued one by one each of them paid the host's launch
    // path and the XCDs' staggered start of a stand-alone launch (profiles/r03/dispatch_ramp.txt) --
    // dependent launches of a graph pay neither.
    struct ExecGuard {
        hipGraphExec_t e = nullptr;
        ~ExecGuard() { if (e) (void)hipGraphExecDestroy(e); }
    } chunk_graph;
    if (!looped && c->opt_chunk_graph) {
        // one un-captured leapfrog first: the launch path sizes its buffers and sets its function
        // attributes outside the capture (it advances the chains like any other leapfrog)
        rc = enqueue_leapfrogs(s, 1);
        if (rc != BPLHIP_OK) return rc;
        steps_done += 1.0;
        rc = capture_launches(c, [&](hipStream_t st) { return enqueue_leapfrogs(st, chunk); }, &chunk_graph.e);
        if (rc != BPLHIP_OK) return rc;   // (no graph is not an error: the loop below enqueues launch by launch)
    }
    while (!all_done) {
        if (steps_done > max_steps)
            return fail(c, BPLHIP_EHIP, "persistent nuts: chains did not finish within the leapfrog bound");
        steps_done += chunk;
#ifdef DC_STAMPS  // diagnostic build: stop mid-chain so the stamp record is an ordinary leapfrog's
        if (const char* cap = getenv("BPLHIP_DEBUG_MAX_STEPS"))
            if (steps_done > atof(cap)) return fail(c, BPLHIP_EHIP, "debug: step cap reached");
#endif
        if (looped) {  // the whole chunk inside one resident launch
            rc = launch_eval_loop(c, ns, md, dP, chunk, s);
            if (rc != BPLHIP_OK) return rc;
        } else if (chunk_graph.e) {
            HIP_TRY(c, hipGraphLaunch(chunk_graph.e, s));
        } else {
            rc = enqueue_leapfrogs(s, chunk);
            if (rc != BPLHIP_OK) return rc;
        }
        HIP_TRY(c, hipMemcpy2DAsync(flags.data(), 8, ns + nsd + nd::P_ALLDONE, stride * 8, 8, C,
                                    hipMemcpyDeviceToHost, s));
        HIP_TRY(c, hipStreamSynchronize(s));
        {   // a hand-off that timed out inside the chunk: stop here, not at the leapfrog bound
            const int frc = consume_fault(c, "persistent nuts");
            if (frc != BPLHIP_OK) return frc;
        }
        all_done = true;
        for (int ch = 0; ch < C; ++ch) all_done = all_done && flags[ch] != 0.0;
    }

    // ---- results
    std::vector<double> stats((size_t)C * kept * 6), pd(nd::pd_doubles(D)), hdr(nd::H_N);
    HIP_TRY(c, hipMemcpy(draws_out, d_draws.p, (size_t)C * kept * D * 8, hipMemcpyDeviceToHost));
    HIP_TRY(c, hipMemcpy(stats.data(), d_stats.p, stats.size() * 8, hipMemcpyDeviceToHost));
    for (int ch = 0; ch < C; ++ch) {
        nuts::Result& r = (*res)[ch];
        const double* nsc = ns + (size_t)ch * stride;
        HIP_TRY(c, hipMemcpy(pd.data(), nsc + nsd, pd.size() * 8, hipMemcpyDeviceToHost));
        HIP_TRY(c, hipMemcpy(hdr.data(), nsc, (size_t)nd::H_N * 8, hipMemcpyDeviceToHost));
        r.potential_energy.resize(kept); r.accept_prob.resize(kept); r.step_size.resize(kept);
        r.aux0.resize(kept); r.num_steps.resize(kept); r.diverging.resize(kept);
        for (int i = 0; i < kept; ++i) {
            const double* st = stats.data() + ((size_t)ch * kept + i) * 6;
            r.potential_energy[i] = st[0]; r.accept_prob[i] = st[1]; r.step_size[i] = st[2];
            r.num_steps[i] = (int)st[3]; r.diverging[i] = (int)st[4]; r.aux0[i] = st[5];
        }
        r.final_step_size = pd[nd::P_STEP];
        r.mean_accept_prob = pd[nd::P_MEAN_ACC];
        r.total_divergences = (int64_t)pd[nd::P_NDIV];
        r.total_leapfrogs = (int64_t)(hdr[nd::H_EVALS] - evals0[ch]);
        r.inverse_mass_matrix.resize(D);
        HIP_TRY(c, hipMemcpy(r.inverse_mass_matrix.data(), nd::vec(const_cast<double*>(nsc), D, nd::V_INVM),
                             (size_t)D * 8, hipMemcpyDeviceToHost));
    }
    return BPLHIP_OK;
}

// numpyro's configuration + the latent sites in model execution order (init key order)
nuts::Config make_nuts_config(const bplhip_ctx* c, const bplhip_nuts_cfg* cfg) {
    nuts::Config nc;
    nc.num_warmup = cfg->num_warmup;
    nc.num_samples = cfg->num_samples;
    nc.max_tree_depth = cfg->max_tree_depth;
    nc.adapt_step_size = cfg->adapt_step_size != 0;
    nc.adapt_mass_matrix = cfg->adapt_mass_matrix != 0;
    nc.thinning = cfg->thinning;
    nc.step_size = cfg->step_size;
    nc.target_accept_prob = cfg->target_accept_prob;
    nc.init_radius = cfg->init_radius;
    nc.max_delta_energy = cfg->max_delta_energy;

    if (c->neutral) {  // bpl/neutral_dixon_coles.py:138-261, model execution order
        const dcn::NeuLayout& L = c->NL;
        const int T = L.T;
        nc.sites = {{L.o_md, 1}, {L.o_s_att, 1}, {L.o_s_def, 1}, {L.o_mha, 1}, {L.o_maa, 1},
                    {L.o_mhd, 1}, {L.o_mad, 1}, {L.o_s_ha, 1}, {L.o_s_aa, 1}, {L.o_s_hd, 1},
                    {L.o_s_ad, 1}};
        if (L.C) nc.sites.push_back({L.o_u, 1});  // bpl/neutral_dixon_coles_WC.py:116 (u first)
        if (L.K) {
            nc.sites.push_back({L.o_bA, L.K});
            nc.sites.push_back({L.o_bD, L.K});
        }
        if (!L.C) nc.sites.push_back({L.o_u, 1});
        for (int o : {L.o_sat, L.o_sdt, L.o_hat, L.o_aat, L.o_hdf, L.o_adf}) nc.sites.push_back({o, T});
        if (L.C) nc.sites.push_back({L.o_conf, L.C});
        nc.sites.push_back({L.o_corr, 1});
    } else if (c->dynamic) {  // bpl/dynamic_dixon_coles.py:74-241, model execution order
        const dcd::DynLayout& L = c->DL;
        const int G = L.G, GT = L.G * L.T;
        nc.sites = {{L.o_mha, G}, {L.o_maa, G}, {L.o_mhd, G}, {L.o_mad, G}, {L.o_s_ha, G},
                    {L.o_s_aa, G}, {L.o_s_hd, G}, {L.o_s_ad, G}, {L.o_s_att, G}, {L.o_s_def, G},
                    {L.o_md, 1}};
        if (L.K) {
            nc.sites.push_back({L.o_bA, L.K});
            nc.sites.push_back({L.o_bD, L.K});
        }
        for (int o : {L.o_u, L.o_sat, L.o_sdt, L.o_hat, L.o_aat, L.o_hdf, L.o_adf})
            nc.sites.push_back({o, GT});
        nc.sites.push_back({L.o_corr, 1});
    } else {  // latent sites in MODEL EXECUTION order (seed-handler key order, Appendix B.5)
        const dc::Layout& L = c->L;
        const int T = L.T, K = L.K;
        if (L.model == dc::MODEL_BASIC) {
            // bpl/dixon_coles.py:46-78
            nc.sites = {{L.o_ha, 1}, {L.o_md, 1}, {L.o_sa, 1}, {L.o_sd, 1},
                        {L.o_adec, T}, {L.o_ddec, T}, {L.o_corr, 1}};
        } else {
            // bpl/extended_dixon_coles.py:112-235
            nc.sites = {{L.o_mha, 1}, {L.o_sh, 1}, {L.o_md, 1}, {L.o_sa, 1}, {L.o_sd, 1}};
            if (K) {
                nc.sites.push_back({L.o_bA, K});
                nc.sites.push_back({L.o_bD, K});
            }
            nc.sites.push_back({L.o_u, 1});
            nc.sites.push_back({L.o_sat, T});
            nc.sites.push_back({L.o_sdt, T});
            nc.sites.push_back({L.o_hadec, T});
            nc.sites.push_back({L.o_corr, 1});
        }
    }

    return nc;
}

void fill_stats(bplhip_nuts_stats* stats, const nuts::Result& res, double wall, int D) {
    if (!stats) return;
    const size_t kept = res.potential_energy.size();
    auto cp = [&](double* dst, const std::vector<double>& v) {
        if (dst) std::memcpy(dst, v.data(), kept * 8);
    };
    cp(stats->potential_energy, res.potential_energy);
    cp(stats->accept_prob, res.accept_prob);
    cp(stats->step_size, res.step_size);
    cp(stats->corr_coef, res.aux0);
    if (stats->num_steps) std::memcpy(stats->num_steps, res.num_steps.data(), kept * 4);
    if (stats->diverging) std::memcpy(stats->diverging, res.diverging.data(), kept * 4);
    stats->final_step_size = res.final_step_size;
    stats->mean_accept_prob = res.mean_accept_prob;
    stats->total_leapfrogs = res.total_leapfrogs;
    stats->total_divergences = res.total_divergences;
    stats->wall_seconds = wall;
    if (stats->inverse_mass_matrix)
        std::memcpy(stats->inverse_mass_matrix, res.inverse_mass_matrix.data(), (size_t)D * 8);
}

}  // namespace

static int bplhip_nuts_run_impl(bplhip_ctx* c, const bplhip_nuts_cfg* cfg, const double* z0,
                               uint32_t seed_hi, uint32_t seed_lo, double* draws_out,
                               bplhip_nuts_stats* stats, void* stream) {
    if (!c) return BPLHIP_EINVAL;
    if (!c->bound) return fail(c, BPLHIP_ESTATE, "nuts_run: no fixtures bound");
    if (!cfg || !draws_out) return fail(c, BPLHIP_EINVAL, "nuts_run: null cfg/draws_out");
    if (cfg->num_warmup < 0 || cfg->num_samples < 1 || cfg->max_tree_depth < 1 ||
        cfg->max_tree_depth > 20 || cfg->thinning < 1 || !(cfg->step_size > 0))
        return fail(c, BPLHIP_EINVAL, "nuts_run: bad configuration");
    HIP_TRY(c, hipSetDevice(c->device));
    const int D = bplhip_latent_dim(c);
    const size_t nd = (size_t)2 * D + 1 + 4;
    HIP_TRY(c, c->d_nuts.ensure(nd * 8));
    if (c->h_pinned_bytes < nd * 8) {
        if (c->h_pinned) (void)hipHostFree(c->h_pinned);
        c->h_pinned = nullptr;
        HIP_TRY(c, hipHostMalloc((void**)&c->h_pinned, nd * 8, hipHostMallocDefault));
        c->h_pinned_bytes = nd * 8;
    }
    HipPotential pot{c, static_cast<hipStream_t>(stream), D};
    pot.d_z = c->d_nuts.as<double>();
    pot.d_grad = pot.d_z + D;
    pot.d_pot = pot.d_grad + D;
    pot.d_aux = pot.d_pot + 1;
    pot.hp = c->h_pinned;

    nuts::Config nc = make_nuts_config(c, cfg);

    nuts::Result res;
    const auto t0 = std::chrono::steady_clock::now();
    int st;
    int dev_rc = BPLHIP_OK;
    const bool device_tree = c->opt_device_nuts && leaf_in_tail(c);
    const bool generic_persist = c->opt_device_nuts && c->opt_persistent_nuts && !leaf_in_tail(c) &&
                                 persistent_fits(c, cfg, 1);
    if ((device_tree || generic_persist) && c->opt_persistent_nuts) {
        // the whole chain on the device (nuts_dev.hip.h, persistent chains)
        if (!generic_persist) {
            int rc1 = ensure_slabs_fresh(c, 1, static_cast<hipStream_t>(stream));
            if (rc1 != BPLHIP_OK) return rc1;
        }
        const tf::Key k1{seed_hi, seed_lo};
        std::vector<nuts::Result> pres;
        const int prc = run_chains_persistent(c, static_cast<hipStream_t>(stream), nc, 1, z0, &k1,
                                              draws_out, &pres);
        if (prc == BPLHIP_ENUMERIC)
            return fail(c, BPLHIP_ENUMERIC, "nuts_run: no finite initial point after 100 tries");
        if (prc != BPLHIP_OK) return prc;
        res = pres[0];
        st = nuts::ST_OK;
    } else if (device_tree) {
        const size_t nsd = nd::ns_doubles(D, nc.max_tree_depth);
        HIP_TRY(c, c->d_ns.ensure(nsd * 8));
        HIP_TRY(c, hipMemsetAsync(c->d_ns.p, 0, nsd * 8, static_cast<hipStream_t>(stream)));
        const size_t need = ((size_t)2 * D + nd::H_N) * 8;
        if (c->h_pinned_bytes < need) {
            if (c->h_pinned) (void)hipHostFree(c->h_pinned);
            c->h_pinned = nullptr;
            HIP_TRY(c, hipHostMalloc((void**)&c->h_pinned, need, hipHostMallocDefault));
            c->h_pinned_bytes = need;
        }
        int rc1 = ensure_slabs_fresh(c, 1, static_cast<hipStream_t>(stream));
        if (rc1 != BPLHIP_OK) return rc1;
        DeviceEngine E{c, static_cast<hipStream_t>(stream), nc, D, nc.max_tree_depth,
                       c->d_ns.as<double>(), c->h_pinned};
        // identity mass matrix until adapted
        std::vector<double> ones(D, 1.0);
        HIP_TRY(c, hipMemcpy(nd::vec(E.ns, D, nd::V_INVM), ones.data(), (size_t)D * 8, hipMemcpyHostToDevice));
        st = nuts::run_chain_engine(E, nc, z0, tf::Key{seed_hi, seed_lo}, draws_out, &res);
        dev_rc = E.rc;
    } else {
        st = nuts::run_chain(pot, nc, z0, tf::Key{seed_hi, seed_lo}, draws_out, &res);
        dev_rc = pot.rc;
    }
    const double wall =
        std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    if (st == nuts::ST_EVAL_FAILED) return dev_rc != BPLHIP_OK ? dev_rc : BPLHIP_EHIP;
    if (st == nuts::ST_NO_FINITE_INIT)
        return fail(c, BPLHIP_ENUMERIC, "nuts_run: no finite initial point after 100 tries");
    fill_stats(stats, res, wall, D);
    return BPLHIP_OK;
}

static int bplhip_nuts_run_chains_impl(bplhip_ctx* c, const bplhip_nuts_cfg* cfg, int32_t n_chains,
                                      const double* z0, const uint32_t* seeds, double* draws_out,
                                      bplhip_nuts_stats* stats, void* stream) {
    if (!c) return BPLHIP_EINVAL;
    if (!c->bound) return fail(c, BPLHIP_ESTATE, "nuts_run_chains: no fixtures bound");
    if (!cfg || !draws_out || !seeds) return fail(c, BPLHIP_EINVAL, "nuts_run_chains: null argument");
    if (n_chains < 1 || n_chains > 4096) return fail(c, BPLHIP_EINVAL, "nuts_run_chains: bad n_chains");
    if (cfg->num_warmup < 0 || cfg->num_samples < 1 || cfg->max_tree_depth < 1 ||
        cfg->max_tree_depth > 20 || cfg->thinning < 1 || !(cfg->step_size > 0))
        return fail(c, BPLHIP_EINVAL, "nuts_run_chains: bad configuration");
    // every model whose leaf does not fit dc_eval's tail: persistent chains with the leaf as its
    // own launch(es) (kp_leaf / kw_leaf_*)
    const bool generic_ok = !leaf_in_tail(c) && c->opt_persistent_nuts && persistent_fits(c, cfg, n_chains);
    if (!generic_ok &&
        (!leaf_in_tail(c) || !vec_ok(c) || !(c->vps[0].staged && c->vps[1].staged && c->vps[2].staged)))
        return fail(c, BPLHIP_EUNSUPPORTED,
                    "nuts_run_chains: these chains do not fit on the device (momentum draws > 16 GiB, or "
                    "persistent_nuts=0 with a model outside dc_eval's tail)");
    HIP_TRY(c, hipSetDevice(c->device));
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int D = bplhip_latent_dim(c), C = n_chains;
    const nuts::Config nc = make_nuts_config(c, cfg);
    if (c->opt_persistent_nuts) {
        if (!generic_ok) {
            int rc0 = ensure_slabs_fresh(c, std::max(1, std::min(C, c->opt_gridy_max_chains)), static_cast<hipStream_t>(stream));
            if (rc0 != BPLHIP_OK) return rc0;
        }
        std::vector<tf::Key> pkeys(C);
        for (int ch = 0; ch < C; ++ch) pkeys[ch] = tf::Key{seeds[2 * ch], seeds[2 * ch + 1]};
        std::vector<nuts::Result> pres;
        const auto pt0 = std::chrono::steady_clock::now();
        const int prc = run_chains_persistent(c, s, nc, C, z0, pkeys.data(), draws_out, &pres);
        const double pwall =
            std::chrono::duration<double>(std::chrono::steady_clock::now() - pt0).count();
        if (prc == BPLHIP_ENUMERIC)
            return fail(c, BPLHIP_ENUMERIC, "nuts_run_chains: no finite initial point after 100 tries");
        if (prc != BPLHIP_OK) return prc;
        if (stats)
            for (int ch = 0; ch < C; ++ch) fill_stats(&stats[ch], pres[ch], pwall, D);
        return BPLHIP_OK;
    }
    const size_t stride = (nd::ns_doubles(D, nc.max_tree_depth) + 1) & ~(size_t)1;
    const int par_stride = nd::par_doubles(nc.max_tree_depth);
    DevBuf d_ns, d_par;
    HIP_TRY(c, d_ns.ensure((size_t)C * stride * 8));
    HIP_TRY(c, d_par.ensure((size_t)C * par_stride * 8));
    HIP_TRY(c, hipMemsetAsync(d_ns.p, 0, (size_t)C * stride * 8, s));
    const size_t pinned = ((size_t)C * (3 * D + nd::H_N + par_stride)) * 8;
    double* hp = nullptr;
    HIP_TRY(c, hipHostMalloc((void**)&hp, pinned, hipHostMallocDefault));
    int rc1 = ensure_slabs_fresh(c, 1, static_cast<hipStream_t>(stream));
    if (rc1 != BPLHIP_OK) {
        (void)hipHostFree(hp);
        return rc1;
    }
    VecDeviceEngine E{c, s, nc, D, nc.max_tree_depth, C, d_ns.as<double>(), stride,
                      d_par.as<double>(), par_stride};
    E.h_r = hp;
    E.h_m = E.h_r + (size_t)C * D;
    E.h_hdr = E.h_m + (size_t)C * D;
    E.h_prop = E.h_hdr + (size_t)C * nd::H_N;
    E.h_par = E.h_prop + (size_t)C * D;
    {   // identity mass matrices until adapted
        std::vector<double> ones((size_t)C * D, 1.0);
        hipError_t e = hipMemcpy2D(nd::vec(E.ns, D, nd::V_INVM), stride * 8, ones.data(), (size_t)D * 8,
                                   (size_t)D * 8, C, hipMemcpyHostToDevice);
        if (e != hipSuccess) {
            (void)hipHostFree(hp);
            return fail(c, BPLHIP_EHIP, "nuts_run_chains: H2D mass: %s", hipGetErrorString(e));
        }
    }
    std::vector<tf::Key> keys(C);
    for (int ch = 0; ch < C; ++ch) keys[ch] = tf::Key{seeds[2 * ch], seeds[2 * ch + 1]};
    std::vector<nuts::Result> res;
    const auto t0 = std::chrono::steady_clock::now();
    const int st = nuts::run_chains_lockstep(E, nc, C, z0, keys.data(), draws_out, &res);
    (void)hipStreamSynchronize(s);
    const double wall =
        std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    (void)hipHostFree(hp);
    if (st == nuts::ST_EVAL_FAILED) return E.rc != BPLHIP_OK ? E.rc : BPLHIP_EHIP;
    if (st == nuts::ST_NO_FINITE_INIT)
        return fail(c, BPLHIP_ENUMERIC, "nuts_run_chains: no finite initial point after 100 tries");
    if (stats)
        for (int ch = 0; ch < C; ++ch) fill_stats(&stats[ch], res[ch], wall, D);
    return BPLHIP_OK;
}

static int bplhip_constrain_impl(bplhip_ctx* c, const double* z_draws, int64_t s,
                                double* attack, double* defence, double* home_advantage,
                                double* corr_coef) {
    if (!c) return BPLHIP_EINVAL;
    if (!c->bound) return fail(c, BPLHIP_ESTATE, "constrain: no fixtures bound");
    if (c->dynamic) return fail(c, BPLHIP_ESTATE, "constrain: use bplhip_constrain_dynamic");
    if (c->neutral) return fail(c, BPLHIP_ESTATE, "constrain: the neutral model's sites are mapped by the caller");
    if (!z_draws || s < 0) return fail(c, BPLHIP_EINVAL, "constrain: bad argument");
    const dc::Layout& L = c->L;
    const int T = L.T;
    const double* xs = c->h_xs.empty() ? nullptr : c->h_xs.data();
    const bool clip = L.model == dc::MODEL_EXTENDED;
    std::vector<double> att(T), def(T), ha(T);
    for (int64_t i = 0; i < s; ++i) {
        const double* z = z_draws + (size_t)i * L.D;
        for (int t = 0; t < T; ++t) dc::team_params(L, z, xs, t, &att[t], &def[t], &ha[t]);
        if (attack) std::memcpy(attack + (size_t)i * T, att.data(), (size_t)T * 8);
        if (defence) std::memcpy(defence + (size_t)i * T, def.data(), (size_t)T * 8);
        if (home_advantage) {
            if (L.model == dc::MODEL_BASIC) home_advantage[i] = ha[0];
            else std::memcpy(home_advantage + (size_t)i * T, ha.data(), (size_t)T * 8);
        }
        if (corr_coef) {
            // compute_corr_coef_bounds (bpl/_util.py:23-30) over the unique pairs
            double M = 0.0, Lh = 0.0, La = 0.0;
            for (uint32_t pk : c->h_pairs) {
                const int h = pk & 0xFFFFu, a = pk >> 16;
                double lh = std::exp(att[h] - def[a] + ha[h]);
                double la = std::exp(att[a] - def[h]);
                if (clip) {
                    lh = std::fmin(lh, dc::RATE_CLIP);
                    la = std::fmin(la, dc::RATE_CLIP);
                }
                M = std::fmax(M, lh * la);
                Lh = std::fmax(Lh, lh);
                La = std::fmax(La, la);
            }
            const double UB = M > 1.0 ? 1.0 / M : 1.0;
            const double LB = -1.0 / std::fmax(Lh, La);
            double q, dq;
            dc::clipped_sigmoid(z[L.o_corr], &q, &dq);
            corr_coef[i] = LB + q * (UB - LB);
        }
    }
    return BPLHIP_OK;
}

// Dynamic model: constrained / deterministic sites per draw.  HOST in, HOST out; outputs
// [s, G, T] each (any may be NULL): attack, defence (the walk), home_attack, away_attack,
// home_defence, away_defence.  corr_coef comes from the sampler statistics.
static int bplhip_constrain_dynamic_impl(bplhip_ctx* c, const double* z_draws, int64_t s,
                                        double* attack, double* defence, double* home_attack,
                                        double* away_attack, double* home_defence,
                                        double* away_defence) {
    if (!c) return BPLHIP_EINVAL;
    if (!c->bound || !c->dynamic) return fail(c, BPLHIP_ESTATE, "constrain_dynamic: no dynamic model bound");
    if (!z_draws || s < 0) return fail(c, BPLHIP_EINVAL, "constrain_dynamic: bad argument");
    const dcd::DynLayout& L = c->DL;
    const int G = L.G, T = L.T, K = L.K;
    const size_t GT = (size_t)G * T;
    for (int64_t i = 0; i < s; ++i) {
        const double* z = z_draws + (size_t)i * L.D;
        for (int t = 0; t < T; ++t) {
            double att = 0.0, def = z[L.o_md];
            for (int k = 0; k < K; ++k) {
                att += c->h_xs[(size_t)t * K + k] * z[L.o_bA + k];
                def += c->h_xs[(size_t)t * K + k] * z[L.o_bD + k];
            }
            for (int g = 0; g < G; ++g) {
                const size_t cidx = (size_t)g * T + t, o = (size_t)i * GT + cidx;
                double a_ = 0.0, d_ = 0.0;
                if (c->dyn_random_walk) {
                    att += z[L.o_sat + cidx] * std::exp(z[L.o_s_att + g]);
                    def += z[L.o_sdt + cidx] * std::exp(z[L.o_s_def + g]);
                    a_ = att;
                    d_ = def;
                }
                if (attack) attack[o] = a_;
                if (defence) defence[o] = d_;
                if (home_attack) home_attack[o] = z[L.o_mha + g] + std::exp(z[L.o_s_ha + g]) * z[L.o_hat + cidx];
                if (away_attack) away_attack[o] = z[L.o_maa + g] + std::exp(z[L.o_s_aa + g]) * z[L.o_aat + cidx];
                if (home_defence) home_defence[o] = z[L.o_mhd + g] + std::exp(z[L.o_s_hd + g]) * z[L.o_hdf + cidx];
                if (away_defence) away_defence[o] = z[L.o_mad + g] + std::exp(z[L.o_s_ad + g]) * z[L.o_adf + cidx];
            }
        }
    }
    return BPLHIP_OK;
}

// ---- predict path on the device (rows f-2, f-4)
enum { PT_ATT = 0, PT_DEF, PT_HA, PT_HAT, PT_AAT, PT_HDF, PT_ADF, PT_CONF };
// one posterior table: float64 as given + float32 transposed (a column's draws contiguous)
static int predict_upload(bplhip_ctx* c, int which, const double* src, size_t rows, size_t cols) {
    HIP_TRY(c, c->dp_tab[which].ensure(rows * cols * 8));
    HIP_TRY(c, hipMemcpy(c->dp_tab[which].p, src, rows * cols * 8, hipMemcpyHostToDevice));
    std::vector<float> tmp(rows * cols);
    for (size_t r = 0; r < rows; ++r)
        for (size_t q = 0; q < cols; ++q) tmp[q * rows + r] = (float)src[r * cols + q];
    HIP_TRY(c, c->dp_tab32[which].ensure(tmp.size() * 4));
    HIP_TRY(c, hipMemcpy(c->dp_tab32[which].p, tmp.data(), tmp.size() * 4, hipMemcpyHostToDevice));
    return BPLHIP_OK;
}
static int predict_upload_corr(bplhip_ctx* c, const double* corr_coef, size_t s) {
    HIP_TRY(c, c->dp_corr.ensure(s * 8));
    HIP_TRY(c, hipMemcpy(c->dp_corr.p, corr_coef, s * 8, hipMemcpyHostToDevice));
    std::vector<float> tmp(s);
    for (size_t r = 0; r < s; ++r) tmp[r] = (float)corr_coef[r];
    HIP_TRY(c, c->dp_corr32.ensure(s * 4));
    HIP_TRY(c, hipMemcpy(c->dp_corr32.p, tmp.data(), s * 4, hipMemcpyHostToDevice));
    return BPLHIP_OK;
}

static int bplhip_predict_set_posterior_impl(bplhip_ctx* c, int32_t s, int32_t t,
                                            const double* attack, const double* defence,
                                            const double* home_advantage,
                                            int32_t home_advantage_per_team,
                                            const double* corr_coef) {
    if (!c) return BPLHIP_EINVAL;
    if (s < 1 || t < 1 || !attack || !defence || !home_advantage || !corr_coef)
        return fail(c, BPLHIP_EINVAL, "predict_set_posterior: bad argument");
    HIP_TRY(c, hipSetDevice(c->device));
    c->pred_S = 0;
    int rc = predict_upload(c, PT_ATT, attack, (size_t)s, (size_t)t);
    if (rc == BPLHIP_OK) rc = predict_upload(c, PT_DEF, defence, (size_t)s, (size_t)t);
    if (rc == BPLHIP_OK) rc = predict_upload(c, PT_HA, home_advantage, (size_t)s, home_advantage_per_team ? (size_t)t : 1);
    if (rc == BPLHIP_OK) rc = predict_upload_corr(c, corr_coef, (size_t)s);
    if (rc != BPLHIP_OK) return rc;
    c->pred_S = s;
    c->pred_T = t;
    c->pred_C = 0;
    c->pred_ha_stride = home_advantage_per_team ? t : 0;
    c->pred_venue = false;
    return BPLHIP_OK;
}

static int bplhip_predict_set_posterior_venue_impl(bplhip_ctx* c, int32_t s, int32_t t, const double* attack,
                                                  const double* defence, const double* home_attack,
                                                  const double* away_attack, const double* home_defence,
                                                  const double* away_defence, int32_t n_conf,
                                                  const double* confederation_strength,
                                                  const double* corr_coef) {
    if (!c) return BPLHIP_EINVAL;
    if (s < 1 || t < 1 || !attack || !defence || !home_attack || !away_attack || !home_defence ||
        !away_defence || !corr_coef || n_conf < 0 || (n_conf > 0) != (confederation_strength != nullptr))
        return fail(c, BPLHIP_EINVAL, "predict_set_posterior_venue: bad argument");
    HIP_TRY(c, hipSetDevice(c->device));
    c->pred_S = 0;
    const double* src[6] = {attack, defence, home_attack, away_attack, home_defence, away_defence};
    const int slot[6] = {PT_ATT, PT_DEF, PT_HAT, PT_AAT, PT_HDF, PT_ADF};
    for (int i = 0; i < 6; ++i) {
        const int rc = predict_upload(c, slot[i], src[i], (size_t)s, (size_t)t);
        if (rc != BPLHIP_OK) return rc;
    }
    if (n_conf) {
        const int rc = predict_upload(c, PT_CONF, confederation_strength, (size_t)s, (size_t)n_conf);
        if (rc != BPLHIP_OK) return rc;
    }
    const int rc = predict_upload_corr(c, corr_coef, (size_t)s);
    if (rc != BPLHIP_OK) return rc;
    c->pred_S = s;
    c->pred_T = t;
    c->pred_C = n_conf;
    c->pred_ha_stride = 0;
    c->pred_venue = true;
    return BPLHIP_OK;
}

// argument checks shared by the four query entry points; venue = the caller is a *_venue entry
static int predict_check_query(bplhip_ctx* c, const char* what, bool venue, int64_t m, const uint16_t* home_idx,
                               const uint16_t* away_idx, const uint8_t* neutral, const uint16_t* home_conf,
                               const uint16_t* away_conf) {
    if (c->pred_S == 0) return fail(c, BPLHIP_ESTATE, "%s: no posterior set", what);
    if (venue != c->pred_venue)
        return fail(c, BPLHIP_ESTATE, "%s: the posterior was set with predict_set_posterior%s", what,
                    c->pred_venue ? "_venue" : "");
    if (m < 0 || m > 0x7FFFFFFF || (m > 0 && (!home_idx || !away_idx)))
        return fail(c, BPLHIP_EINVAL, "%s: bad argument", what);
    if (venue && m > 0 && (!neutral || (c->pred_C > 0) != (home_conf != nullptr) || (home_conf != nullptr) != (away_conf != nullptr)))
        return fail(c, BPLHIP_EINVAL, "%s: neutral_venue is required, confederations exactly when the posterior has them", what);
    for (int64_t i = 0; i < m; ++i) {
        if (home_idx[i] >= c->pred_T || away_idx[i] >= c->pred_T)
            return fail(c, BPLHIP_EINVAL, "%s: team index out of range at %lld", what, (long long)i);
        if (venue && home_conf && (home_conf[i] >= c->pred_C || away_conf[i] >= c->pred_C))
            return fail(c, BPLHIP_EINVAL, "%s: confederation index out of range at %lld", what, (long long)i);
    }
    return BPLHIP_OK;
}

static int predict_score_grid_any(bplhip_ctx* c, const char* what, bool venue, int64_t m, const uint16_t* home_idx,
                                  const uint16_t* away_idx, const uint8_t* neutral, const uint16_t* home_conf,
                                  const uint16_t* away_conf, int32_t max_goals, void* out, bool f32, void* stream) {
    if (!c) return BPLHIP_EINVAL;
    if (max_goals < 0 || max_goals > dcp::GRID_MAX_GOALS)
        return fail(c, BPLHIP_EINVAL, "%s: max_goals=%d out of range [0,%d]", what, max_goals, dcp::GRID_MAX_GOALS);
    int rc = predict_check_query(c, what, venue, m, home_idx, away_idx, neutral, home_conf, away_conf);
    if (rc != BPLHIP_OK) return rc;
    if (m > 0 && !out) return fail(c, BPLHIP_EINVAL, "%s: bad argument", what);
    if (m == 0) return BPLHIP_OK;
    HIP_TRY(c, hipSetDevice(c->device));
    hipStream_t s = static_cast<hipStream_t>(stream);
    const size_t cells = (size_t)m * (max_goals + 1) * (max_goals + 1);
    // u16 h, a, hc, ac then u8 neutral, rounded up to 8 bytes
    const size_t idx_bytes = ((size_t)m * 9 + 7) & ~(size_t)7;
    const size_t cell_bytes = f32 ? 4 : 8;
    HIP_TRY(c, c->dp_q.ensure(idx_bytes + cells * cell_bytes));
    uint16_t* q = c->dp_q.as<uint16_t>();
    uint8_t* qn = reinterpret_cast<uint8_t*>(q + 4 * m);
    double* d_out = reinterpret_cast<double*>(c->dp_q.as<char>() + idx_bytes);
    HIP_TRY(c, hipMemcpyAsync(q, home_idx, (size_t)m * 2, hipMemcpyHostToDevice, s));
    HIP_TRY(c, hipMemcpyAsync(q + m, away_idx, (size_t)m * 2, hipMemcpyHostToDevice, s));
    if (venue) {
        HIP_TRY(c, hipMemcpyAsync(qn, neutral, (size_t)m, hipMemcpyHostToDevice, s));
        if (home_conf) {
            HIP_TRY(c, hipMemcpyAsync(q + 2 * m, home_conf, (size_t)m * 2, hipMemcpyHostToDevice, s));
            HIP_TRY(c, hipMemcpyAsync(q + 3 * m, away_conf, (size_t)m * 2, hipMemcpyHostToDevice, s));
        }
    }
    dcp::GridArgs A{};
    for (int k = 0; k < 64; ++k) A.rk[k] = (float)(1.0 / ((double)k + 1.0));  // the constants of the pmf recurrence
    {
        double fct = 1.0;
        for (int k = 0; k < 16; ++k) {
            A.rfact[k] = 1.0 / fct;
            fct *= (double)(k + 1);
        }
    }
    A.S = c->pred_S;
    A.T = c->pred_T;
    A.attack = c->dp_tab32[PT_ATT].as<const float>();
    A.defence = c->dp_tab32[PT_DEF].as<const float>();
    A.home_adv = c->dp_tab32[PT_HA].as<const float>();
    A.ha_stride = c->pred_ha_stride;
    A.home_attack = c->dp_tab32[PT_HAT].as<const float>();
    A.away_attack = c->dp_tab32[PT_AAT].as<const float>();
    A.home_defence = c->dp_tab32[PT_HDF].as<const float>();
    A.away_defence = c->dp_tab32[PT_ADF].as<const float>();
    A.conf = c->pred_C ? c->dp_tab32[PT_CONF].as<const float>() : nullptr;
    A.corr = c->dp_corr32.as<const float>();
    A.M = (int)m;
    A.G = max_goals;
    A.h = q;
    A.a = q + m;
    A.hc = q + 2 * m;
    A.ac = q + 3 * m;
    A.neutral = qn;
    A.out = f32 ? nullptr : d_out;
    A.out32 = f32 ? reinterpret_cast<float*>(d_out) : nullptr;
    const dim3 grid((unsigned)((m + dcp::GRID_WAVES - 1) / dcp::GRID_WAVES)), block(64 * dcp::GRID_WAVES);
    if (venue) hipLaunchKernelGGL(dcp::predict_score_grid<true>, grid, block, 0, s, A);
    else hipLaunchKernelGGL(dcp::predict_score_grid<false>, grid, block, 0, s, A);
    HIP_TRY(c, hipGetLastError());
    HIP_TRY(c, hipMemcpyAsync(out, d_out, cells * cell_bytes, hipMemcpyDeviceToHost, s));
    HIP_TRY(c, hipStreamSynchronize(s));
    return BPLHIP_OK;
}

static int predict_score_proba_any(bplhip_ctx* c, const char* what, bool venue, int64_t m, const uint16_t* home_idx,
                                   const uint16_t* away_idx, const uint16_t* home_goals, const uint16_t* away_goals,
                                   const uint8_t* neutral, const uint16_t* home_conf, const uint16_t* away_conf,
                                   double* out, void* stream) {
    if (!c) return BPLHIP_EINVAL;
    int rc = predict_check_query(c, what, venue, m, home_idx, away_idx, neutral, home_conf, away_conf);
    if (rc != BPLHIP_OK) return rc;
    if (m > 0 && (!home_goals || !away_goals || !out)) return fail(c, BPLHIP_EINVAL, "%s: bad argument", what);
    if (m == 0) return BPLHIP_OK;
    HIP_TRY(c, hipSetDevice(c->device));
    hipStream_t s = static_cast<hipStream_t>(stream);
    // u16 h, a, x, y, hc, ac then u8 neutral, rounded up to 8 bytes; then the f64 results
    const size_t idx_bytes = ((size_t)m * 13 + 7) & ~(size_t)7;
    HIP_TRY(c, c->dp_q.ensure(idx_bytes + (size_t)m * 8));
    uint16_t* q = c->dp_q.as<uint16_t>();
    uint8_t* qn = reinterpret_cast<uint8_t*>(q + 6 * m);
    double* d_out = reinterpret_cast<double*>(c->dp_q.as<char>() + idx_bytes);
    HIP_TRY(c, hipMemcpyAsync(q, home_idx, (size_t)m * 2, hipMemcpyHostToDevice, s));
    HIP_TRY(c, hipMemcpyAsync(q + m, away_idx, (size_t)m * 2, hipMemcpyHostToDevice, s));
    HIP_TRY(c, hipMemcpyAsync(q + 2 * m, home_goals, (size_t)m * 2, hipMemcpyHostToDevice, s));
    HIP_TRY(c, hipMemcpyAsync(q + 3 * m, away_goals, (size_t)m * 2, hipMemcpyHostToDevice, s));
    if (venue) {
        HIP_TRY(c, hipMemcpyAsync(qn, neutral, (size_t)m, hipMemcpyHostToDevice, s));
        if (home_conf) {
            HIP_TRY(c, hipMemcpyAsync(q + 4 * m, home_conf, (size_t)m * 2, hipMemcpyHostToDevice, s));
            HIP_TRY(c, hipMemcpyAsync(q + 5 * m, away_conf, (size_t)m * 2, hipMemcpyHostToDevice, s));
        }
    }
    dcp::PredictArgs A{};
    A.S = c->pred_S;
    A.T = c->pred_T;
    A.C = c->pred_C;
    A.attack = c->dp_tab[PT_ATT].as<const double>();
    A.defence = c->dp_tab[PT_DEF].as<const double>();
    A.home_adv = c->dp_tab[PT_HA].as<const double>();
    A.ha_stride = c->pred_ha_stride;
    A.home_attack = c->dp_tab[PT_HAT].as<const double>();
    A.away_attack = c->dp_tab[PT_AAT].as<const double>();
    A.home_defence = c->dp_tab[PT_HDF].as<const double>();
    A.away_defence = c->dp_tab[PT_ADF].as<const double>();
    A.conf = c->pred_C ? c->dp_tab[PT_CONF].as<const double>() : nullptr;
    A.corr = c->dp_corr.as<const double>();
    A.M = m;
    A.h = q;
    A.a = q + m;
    A.x = q + 2 * m;
    A.y = q + 3 * m;
    A.hc = q + 4 * m;
    A.ac = q + 5 * m;
    A.neutral = qn;
    A.out = d_out;
    const dim3 grid((unsigned)((m + 255) / 256)), block(256);
    if (venue) hipLaunchKernelGGL(dcp::predict_score_proba<true>, grid, block, 0, s, A);
    else hipLaunchKernelGGL(dcp::predict_score_proba<false>, grid, block, 0, s, A);
    HIP_TRY(c, hipGetLastError());
    HIP_TRY(c, hipMemcpyAsync(out, d_out, (size_t)m * 8, hipMemcpyDeviceToHost, s));
    HIP_TRY(c, hipStreamSynchronize(s));
    return BPLHIP_OK;
}

// ---- guarded C-ABI entry points (see `guarded`)
extern "C" int bplhip_create(bplhip_ctx** out, int device_id) {
    return guarded(nullptr, "bplhip_create", [&] { return bplhip_create_impl(out, device_id); });
}
extern "C" int bplhip_set_fixtures(bplhip_ctx* c, int model_kind, int64_t n, int32_t n_teams,
                        const uint16_t* home_idx, const uint16_t* away_idx,
                        const uint8_t* home_goals, const uint8_t* away_goals,
                        const float* weights, const double* covariates, int32_t k,
                        void* stream) {
    return guarded(c, "bplhip_set_fixtures", [&] { return bplhip_set_fixtures_impl(c, model_kind, n, n_teams, home_idx, away_idx, home_goals, away_goals, weights, covariates, k, stream); });
}
extern "C" int bplhip_set_fixtures_neutral(bplhip_ctx* c, int64_t n, int32_t n_teams, const uint16_t* home_idx,
                                const uint16_t* away_idx, const uint8_t* home_goals,
                                const uint8_t* away_goals, const uint8_t* neutral_venue,
                                const uint8_t* home_conf, const uint8_t* away_conf, int32_t n_conf,
                                const float* weights, const double* covariates, int32_t k,
                                void* stream) {
    return guarded(c, "bplhip_set_fixtures_neutral", [&] { return bplhip_set_fixtures_neutral_impl(c, n, n_teams, home_idx, away_idx, home_goals, away_goals, neutral_venue, home_conf, away_conf, n_conf, weights, covariates, k, stream); });
}
extern "C" int bplhip_set_fixtures_dynamic(bplhip_ctx* c, int64_t n, int32_t n_teams, int32_t n_gameweeks,
                                const uint16_t* home_idx, const uint16_t* away_idx,
                                const uint8_t* home_goals, const uint8_t* away_goals,
                                const uint16_t* gameweek, const uint8_t* neutral_venue,
                                const double* covariates, int32_t k, int32_t random_walk,
                                void* stream) {
    return guarded(c, "bplhip_set_fixtures_dynamic", [&] { return bplhip_set_fixtures_dynamic_impl(c, n, n_teams, n_gameweeks, home_idx, away_idx, home_goals, away_goals, gameweek, neutral_venue, covariates, k, random_walk, stream); });
}
extern "C" int bplhip_logp_grad_batched(bplhip_ctx* c, int32_t n_chains, const double* z,
                             double* potential, double* grad, double* aux, void* stream) {
    return guarded(c, "bplhip_logp_grad_batched", [&] { return bplhip_logp_grad_batched_impl(c, n_chains, z, potential, grad, aux, stream); });
}
extern "C" int bplhip_logp_grad_graph(bplhip_ctx* c, int32_t count, int32_t n_z, const double* z,
                           double* potential, double* grad, int32_t replays, void* stream) {
    return guarded(c, "bplhip_logp_grad_graph", [&] { return bplhip_logp_grad_graph_impl(c, count, n_z, z, potential, grad, replays, stream); });
}
extern "C" int bplhip_nuts_run(bplhip_ctx* c, const bplhip_nuts_cfg* cfg, const double* z0,
                               uint32_t seed_hi, uint32_t seed_lo, double* draws_out,
                               bplhip_nuts_stats* stats, void* stream) {
    return guarded(c, "bplhip_nuts_run", [&] { return bplhip_nuts_run_impl(c, cfg, z0, seed_hi, seed_lo, draws_out, stats, stream); });
}
extern "C" int bplhip_nuts_run_chains(bplhip_ctx* c, const bplhip_nuts_cfg* cfg, int32_t n_chains,
                                      const double* z0, const uint32_t* seeds, double* draws_out,
                                      bplhip_nuts_stats* stats, void* stream) {
    return guarded(c, "bplhip_nuts_run_chains", [&] { return bplhip_nuts_run_chains_impl(c, cfg, n_chains, z0, seeds, draws_out, stats, stream); });
}
extern "C" int bplhip_constrain(bplhip_ctx* c, const double* z_draws, int64_t s,
                                double* attack, double* defence, double* home_advantage,
                                double* corr_coef) {
    return guarded(c, "bplhip_constrain", [&] { return bplhip_constrain_impl(c, z_draws, s, attack, defence, home_advantage, corr_coef); });
}
extern "C" int bplhip_constrain_dynamic(bplhip_ctx* c, const double* z_draws, int64_t s,
                                        double* attack, double* defence, double* home_attack,
                                        double* away_attack, double* home_defence,
                                        double* away_defence) {
    return guarded(c, "bplhip_constrain_dynamic", [&] { return bplhip_constrain_dynamic_impl(c, z_draws, s, attack, defence, home_attack, away_attack, home_defence, away_defence); });
}
extern "C" int bplhip_predict_set_posterior(bplhip_ctx* c, int32_t s, int32_t t,
                                            const double* attack, const double* defence,
                                            const double* home_advantage,
                                            int32_t home_advantage_per_team,
                                            const double* corr_coef) {
    return guarded(c, "bplhip_predict_set_posterior", [&] { return bplhip_predict_set_posterior_impl(c, s, t, attack, defence, home_advantage, home_advantage_per_team, corr_coef); });
}
extern "C" int bplhip_predict_score_grid(bplhip_ctx* c, int64_t m, const uint16_t* home_idx,
                                         const uint16_t* away_idx, int32_t max_goals, double* out,
                                         void* stream) {
    return guarded(c, "bplhip_predict_score_grid", [&] {
        return predict_score_grid_any(c, "predict_score_grid", false, m, home_idx, away_idx, nullptr, nullptr, nullptr, max_goals, out, false, stream);
    });
}
extern "C" int bplhip_predict_score_grid_f32(bplhip_ctx* c, int64_t m, const uint16_t* home_idx,
                                             const uint16_t* away_idx, int32_t max_goals, float* out,
                                             void* stream) {
    return guarded(c, "bplhip_predict_score_grid_f32", [&] {
        return predict_score_grid_any(c, "predict_score_grid_f32", false, m, home_idx, away_idx, nullptr, nullptr, nullptr, max_goals, out, true, stream);
    });
}
extern "C" int bplhip_predict_set_posterior_venue(bplhip_ctx* c, int32_t s, int32_t t, const double* attack,
                                                  const double* defence, const double* home_attack,
                                                  const double* away_attack, const double* home_defence,
                                                  const double* away_defence, int32_t n_conf,
                                                  const double* confederation_strength,
                                                  const double* corr_coef) {
    return guarded(c, "bplhip_predict_set_posterior_venue", [&] {
        return bplhip_predict_set_posterior_venue_impl(c, s, t, attack, defence, home_attack, away_attack, home_defence,
                                                       away_defence, n_conf, confederation_strength, corr_coef);
    });
}
extern "C" int bplhip_predict_score_grid_venue(bplhip_ctx* c, int64_t m, const uint16_t* home_idx,
                                               const uint16_t* away_idx, const uint8_t* neutral_venue,
                                               const uint16_t* home_conf, const uint16_t* away_conf,
                                               int32_t max_goals, double* out, void* stream) {
    return guarded(c, "bplhip_predict_score_grid_venue", [&] {
        return predict_score_grid_any(c, "predict_score_grid_venue", true, m, home_idx, away_idx, neutral_venue, home_conf, away_conf, max_goals, out, false, stream);
    });
}
extern "C" int bplhip_predict_score_grid_venue_f32(bplhip_ctx* c, int64_t m, const uint16_t* home_idx,
                                                   const uint16_t* away_idx, const uint8_t* neutral_venue,
                                                   const uint16_t* home_conf, const uint16_t* away_conf,
                                                   int32_t max_goals, float* out, void* stream) {
    return guarded(c, "bplhip_predict_score_grid_venue_f32", [&] {
        return predict_score_grid_any(c, "predict_score_grid_venue_f32", true, m, home_idx, away_idx, neutral_venue, home_conf, away_conf, max_goals, out, true, stream);
    });
}
extern "C" int bplhip_predict_score_proba_venue(bplhip_ctx* c, int64_t m, const uint16_t* home_idx,
                                                const uint16_t* away_idx, const uint16_t* home_goals,
                                                const uint16_t* away_goals, const uint8_t* neutral_venue,
                                                const uint16_t* home_conf, const uint16_t* away_conf,
                                                double* out, void* stream) {
    return guarded(c, "bplhip_predict_score_proba_venue", [&] {
        return predict_score_proba_any(c, "predict_score_proba_venue", true, m, home_idx, away_idx, home_goals, away_goals, neutral_venue, home_conf, away_conf, out, stream);
    });
}
__global__ void selftest_math_kernel(int which, long long n, const double* in, double* out) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double x = in[i];
    out[i] = which == 0 ? dc::lean::exp(x) : which == 1 ? dc::lean::log(x)
           : which == 2 ? dc::lean::log1p_pos(x) : dc::lean::rcp(x);
}
extern "C" int bplhip_selftest_math(bplhip_ctx* c, int32_t which, int64_t n, const double* in, double* out) {
    return guarded(c, "bplhip_selftest_math", [&] {
        if (!c || which < 0 || which > 3 || n < 1 || !in || !out) return c ? fail(c, BPLHIP_EINVAL, "selftest_math: bad arguments") : BPLHIP_EINVAL;
        HIP_TRY(c, hipSetDevice(c->device));
        DevBuf din, dout;
        HIP_TRY(c, din.ensure((size_t)n * 8));
        HIP_TRY(c, dout.ensure((size_t)n * 8));
        HIP_TRY(c, hipMemcpy(din.p, in, (size_t)n * 8, hipMemcpyHostToDevice));
        hipLaunchKernelGGL(selftest_math_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, which,
                           (long long)n, din.as<const double>(), dout.as<double>());
        HIP_TRY(c, hipGetLastError());
        HIP_TRY(c, hipMemcpy(out, dout.p, (size_t)n * 8, hipMemcpyDeviceToHost));
        return (int)BPLHIP_OK;
    });
}
extern "C" int bplhip_predict_score_proba(bplhip_ctx* c, int64_t m, const uint16_t* home_idx,
                                          const uint16_t* away_idx, const uint16_t* home_goals,
                                          const uint16_t* away_goals, double* out, void* stream) {
    return guarded(c, "bplhip_predict_score_proba", [&] { return predict_score_proba_any(c, "predict_score_proba", false, m, home_idx, away_idx, home_goals, away_goals, nullptr, nullptr, nullptr, out, stream); });
}
