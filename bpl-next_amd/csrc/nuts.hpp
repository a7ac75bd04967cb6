// nuts.hpp -- host-side NUTS driver (header-only, float64), templated on the potential
// functor so the same code drives the HIP path (product) and a CPU potential in the
// host-logic tests.  It restates numpyro 0.13.2's iterative NUTS as reached from
// `NUTS(self._model)` + `MCMC(...)` at bpl/dixon_coles.py:100-116 with every default:
// velocity-Verlet leapfrog, iterative tree doubling with checkpointed U-turn checks,
// uniform (inside a subtree) / biased (across doublings) multinomial transitions,
// divergence at dE > 1000, max depth 10, windowed warm-up with dual-averaging step size
// and regularised Welford diagonal mass matrix, init_to_uniform(radius=2).
// (numpyro/jax are not in the reference tree: SURVEY.md Appendix B is the blueprint.)
//
// Potential concept:
//   int  dim() const;
//   bool operator()(const double* z, double* U, double* grad, double* aux4);  // false = backend failure
#pragma once
#include <cmath>
#include <cstdint>
#include <limits>
#include <utility>
#include <vector>

#include "threefry.hpp"

namespace nuts {

using vec = std::vector<double>;

enum { ST_OK = 0, ST_EVAL_FAILED = 1, ST_NO_FINITE_INIT = 2 };

struct Site {
    int offset, size;  // a latent site in MODEL EXECUTION order (for init_to_uniform keys)
};

struct Config {
    int num_warmup = 500, num_samples = 1000, max_tree_depth = 10, thinning = 1;
    bool adapt_step_size = true, adapt_mass_matrix = true;
    double step_size = 1.0, target_accept_prob = 0.8, init_radius = 2.0,
           max_delta_energy = 1000.0;
    std::vector<Site> sites;  // empty -> one site covering the whole vector
};

struct Result {
    vec potential_energy, accept_prob, step_size, aux0;
    std::vector<int32_t> num_steps, diverging;
    vec inverse_mass_matrix;
    double final_step_size = 0, mean_accept_prob = 0;
    int64_t total_leapfrogs = 0, total_divergences = 0;
};

// ---------------------------------------------------------------- adaptation pieces

struct Window {
    int start, end;
};

// numpyro.infer.hmc_util.build_adaptation_schedule (Stan's windowed schedule)
inline std::vector<Window> build_adaptation_schedule(int num_steps) {
    std::vector<Window> sch;
    if (num_steps < 20) {
        sch.push_back({0, num_steps - 1});
        return sch;
    }
    int start_buffer = 75, end_buffer = 50, init_window = 25;
    if (start_buffer + end_buffer + init_window > num_steps) {
        start_buffer = (int)(0.15 * num_steps);
        end_buffer = (int)(0.1 * num_steps);
        init_window = num_steps - start_buffer - end_buffer;
    }
    sch.push_back({0, start_buffer - 1});
    const int end_window_start = num_steps - end_buffer;
    int next_size = init_window, next_start = start_buffer;
    while (next_start < end_window_start) {
        const int cur_start = next_start;
        int cur_size = next_size;
        if (3 * cur_size <= end_window_start - cur_start) next_size = 2 * cur_size;
        else cur_size = end_window_start - cur_start;
        next_start = cur_start + cur_size;
        sch.push_back({cur_start, next_start - 1});
    }
    sch.push_back({end_window_start, num_steps - 1});
    return sch;
}

// numpyro.infer.hmc_util.dual_averaging(t0=10, kappa=0.75, gamma=0.05)
struct DualAveraging {
    double x_t = 0, x_avg = 0, g_avg = 0, prox_center = 0;
    int t = 0;
    void init(double prox) {
        x_t = x_avg = g_avg = 0;
        t = 0;
        prox_center = prox;
    }
    void update(double g) {
        constexpr double t0 = 10, kappa = 0.75, gamma = 0.05;
        t += 1;
        g_avg = (1.0 - 1.0 / (t + t0)) * g_avg + g / (t + t0);
        x_t = prox_center - std::sqrt((double)t) / gamma * g_avg;
        const double weight_t = std::pow((double)t, -kappa);
        x_avg = (1.0 - weight_t) * x_avg + weight_t * x_t;
    }
};

// numpyro.infer.hmc_util.welford_covariance(diagonal=True)
struct Welford {
    vec mean, m2;
    int n = 0;
    void init(int D) {
        mean.assign(D, 0.0);
        m2.assign(D, 0.0);
        n = 0;
    }
    void update(const vec& x) {
        n += 1;
        for (size_t i = 0; i < x.size(); ++i) {
            const double d_pre = x[i] - mean[i];
            mean[i] += d_pre / n;
            m2[i] += d_pre * (x[i] - mean[i]);
        }
    }
    void final_regularized(vec* cov) const {
        cov->resize(mean.size());
        for (size_t i = 0; i < mean.size(); ++i) {
            double c = m2[i] / (n - 1);
            c = (double(n) / (n + 5.0)) * c + 1e-3 * (5.0 / (n + 5.0));
            (*cov)[i] = c;
        }
    }
};

inline void leaf_idx_to_ckpt_idxs(int n, int* idx_min, int* idx_max) {
    int c = 0;
    for (int v = n >> 1; v > 0; v >>= 1) c += v & 1;
    int s = 0;
    for (int v = n; v & 1; v >>= 1) s += 1;
    *idx_max = c;
    *idx_min = c - s + 1;
}

// ------------------------------------------------------------------------- the tree

struct Phase {  // one phase-space point
    vec z, r, g;
};

struct Tree {
    Phase left, right;
    vec z_prop, g_prop;
    double pe_prop = 0, energy_prop = 0, aux_prop[4] = {0, 0, 0, 0};
    int depth = 0;
    double weight = 0;
    vec r_sum;
    bool turning = false, diverging = false;
    double sum_accept = 0;
    int num_proposals = 0;
};

template <class Pot>
struct Sampler {
    Pot& pot;
    const Config& cfg;
    int D;
    vec inv_mass, mass_sqrt;
    double step_size;
    int64_t leapfrogs = 0;
    bool eval_failed = false;
    std::vector<vec> r_ckpts, r_sum_ckpts;

    Sampler(Pot& p, const Config& c) : pot(p), cfg(c), D(p.dim()) {
        inv_mass.assign(D, 1.0);
        mass_sqrt.assign(D, 1.0);
        step_size = c.step_size;
        r_ckpts.assign(c.max_tree_depth, vec(D, 0.0));
        r_sum_ckpts.assign(c.max_tree_depth, vec(D, 0.0));
    }

    double kinetic(const vec& r) const {
        double k = 0;
        for (int i = 0; i < D; ++i) k += inv_mass[i] * r[i] * r[i];
        return 0.5 * k;
    }

    bool is_turning(const vec& r_left, const vec& r_right, const vec& r_sum) const {
        double dl = 0, dr = 0;
        for (int i = 0; i < D; ++i) {
            const double rs = r_sum[i] - 0.5 * (r_left[i] + r_right[i]);
            dl += inv_mass[i] * r_left[i] * rs;
            dr += inv_mass[i] * r_right[i] * rs;
        }
        return (dl <= 0) | (dr <= 0);
    }

    // one velocity-Verlet step (numpyro velocity_verlet.update_fn) + leaf bookkeeping
    Tree build_basetree(const Phase& from, bool going_right, double energy_current) {
        const double eps = going_right ? step_size : -step_size;
        Tree t;
        Phase p;
        p.z.resize(D);
        p.r.resize(D);
        p.g.resize(D);
        for (int i = 0; i < D; ++i) {
            p.r[i] = from.r[i] - 0.5 * eps * from.g[i];
            p.z[i] = from.z[i] + eps * inv_mass[i] * p.r[i];
        }
        double pe = 0, aux[4] = {0, 0, 0, 0};
        if (!pot(p.z.data(), &pe, p.g.data(), aux)) eval_failed = true;
        leapfrogs += 1;
        for (int i = 0; i < D; ++i) p.r[i] -= 0.5 * eps * p.g[i];
        const double energy_new = pe + kinetic(p.r);
        double delta = energy_new - energy_current;
        if (std::isnan(delta)) delta = std::numeric_limits<double>::infinity();
        t.left = p;
        t.right = p;
        t.z_prop = p.z;
        t.g_prop = p.g;
        t.pe_prop = pe;
        t.energy_prop = energy_new;
        for (int i = 0; i < 4; ++i) t.aux_prop[i] = aux[i];
        t.depth = 0;
        t.weight = -delta;
        t.r_sum = p.r;
        t.turning = false;
        t.diverging = delta > cfg.max_delta_energy;
        t.sum_accept = std::fmin(1.0, std::exp(-delta));
        t.num_proposals = 1;
        return t;
    }

    static double logaddexp(double a, double b) {
        if (a == b) return a + 0.6931471805599453;  // also covers (-inf, -inf)
        const double m = std::fmax(a, b);
        return m + std::log1p(std::exp(-std::fabs(a - b)));
    }

    Tree combine(const Tree& cur, const Tree& nw, bool going_right, tf::Key key, bool biased) {
        Tree out;
        if (going_right) {
            out.left = cur.left;
            out.right = nw.right;
        } else {
            out.left = nw.left;
            out.right = cur.right;
        }
        out.r_sum.resize(D);
        for (int i = 0; i < D; ++i) out.r_sum[i] = cur.r_sum[i] + nw.r_sum[i];
        double prob;
        if (biased) {
            prob = std::exp(nw.weight - cur.weight);
            if (nw.turning || nw.diverging) prob = 0.0;
            prob = std::fmin(prob, 1.0);
            out.turning = nw.turning | is_turning(out.left.r, out.right.r, out.r_sum);
        } else {
            prob = 1.0 / (1.0 + std::exp(-(nw.weight - cur.weight)));  // expit
            out.turning = cur.turning;
        }
        const bool take_new = tf::bernoulli(key, prob);
        const Tree& src = take_new ? nw : cur;
        out.z_prop = src.z_prop;
        out.g_prop = src.g_prop;
        out.pe_prop = src.pe_prop;
        out.energy_prop = src.energy_prop;
        for (int i = 0; i < 4; ++i) out.aux_prop[i] = src.aux_prop[i];
        out.depth = cur.depth + 1;
        out.weight = logaddexp(cur.weight, nw.weight);
        out.diverging = nw.diverging;
        out.sum_accept = cur.sum_accept + nw.sum_accept;
        out.num_proposals = cur.num_proposals + nw.num_proposals;
        return out;
    }

    bool is_iterative_turning(const vec& r, const vec& r_sum, int idx_min, int idx_max) const {
        vec sub(D);
        for (int i = idx_max; i >= idx_min; --i) {
            for (int k = 0; k < D; ++k) sub[k] = r_sum[k] - r_sum_ckpts[i][k] + r_ckpts[i][k];
            if (is_turning(r_ckpts[i], r, sub)) return true;
        }
        return false;
    }

    Tree iterative_build_subtree(const Tree& proto, bool going_right, tf::Key key,
                                 double energy_current) {
        const int max_num = 1 << proto.depth;
        Tree cur = proto;
        cur.num_proposals = 0;
        bool turning = false;
        while (cur.num_proposals < max_num && !turning && !cur.diverging && !eval_failed) {
            tf::Key k_next, k_tr;
            tf::split2(key, &k_next, &k_tr);
            key = k_next;
            const Phase& edge = going_right ? cur.right : cur.left;
            Tree leaf = build_basetree(edge, going_right, energy_current);
            const int leaf_idx = cur.num_proposals;
            Tree nt = leaf_idx == 0 ? leaf : combine(cur, leaf, going_right, k_tr, false);
            int imin, imax;
            leaf_idx_to_ckpt_idxs(leaf_idx, &imin, &imax);
            if ((leaf_idx & 1) == 0) {
                r_ckpts[imax] = leaf.right.r;
                r_sum_ckpts[imax] = nt.r_sum;
            }
            turning = is_iterative_turning(leaf.right.r, nt.r_sum, imin, imax);
            cur = std::move(nt);
        }
        cur.depth = proto.depth;
        cur.turning = turning;
        return cur;
    }

    // numpyro build_tree: returns the final tree of one NUTS transition
    Tree build_tree(const vec& z, const vec& r, double pe, const vec& g, const double* aux,
                    tf::Key key) {
        const double energy_current = pe + kinetic(r);
        Tree tree;
        tree.left = {z, r, g};
        tree.right = tree.left;
        tree.z_prop = z;
        tree.g_prop = g;
        tree.pe_prop = pe;
        tree.energy_prop = energy_current;
        for (int i = 0; i < 4; ++i) tree.aux_prop[i] = aux[i];
        tree.r_sum = r;
        while (tree.depth < cfg.max_tree_depth && !tree.turning && !tree.diverging &&
               !eval_failed) {
            tf::Key k_next, k_dir, k_dbl;
            tf::split3(key, &k_next, &k_dir, &k_dbl);
            key = k_next;
            const bool going_right = tf::bernoulli(k_dir, 0.5);
            tf::Key k_sub, k_tr;
            tf::split2(k_dbl, &k_sub, &k_tr);
            Tree nw = iterative_build_subtree(tree, going_right, k_sub, energy_current);
            tree = combine(tree, nw, going_right, k_tr, true);
        }
        return tree;
    }
};

// One attempt of init_to_uniform(radius) as numpyro's find_valid_initial_params runs it
// (numpyro 0.13.2 infer/util.py: NUTS's default strategy with prototype parameters at hand takes
// the branch that "doesn't require tracing the model"):
//   key, subkey = split(key)
//   for each latent site in model trace order:
//       z_site = uniform(subkey, shape, -r, r);  key, subkey = split(key)
// `key` is carried into the next attempt.  (Round 1 had restated the seed-handler branch -- one
// key per site from handlers.seed, split again inside init_to_uniform -- which numpyro only takes
// for strategies other than init_to_uniform; found by the independent restatement the tests hold.)
inline void draw_init(tf::Key* key, const std::vector<Site>& sites, double radius, vec* z) {
    tf::Key sub;
    tf::split2(*key, key, &sub);
    for (const Site& s : sites) {
        tf::uniform(sub, s.size, (float)-radius, (float)radius, z->data() + s.offset);
        tf::split2(*key, key, &sub);
    }
}

// What one NUTS transition reports back to the adaptation / collection loop.
struct TransitionOut {
    double accept_prob = 0, pe = 0, aux[4] = {0, 0, 0, 0};
    int num_steps = 0;
    bool diverging = false;
};

// Engine concept (the tree builder behind run_chain_engine):
//   int  dim() const;
//   bool set_state(const double* z, double* pe, bool* finite);   // evaluate + adopt z
//   bool transition(const double* r, tf::Key key, double step_size, const vec& inv_mass,
//                   bool mass_changed, double* z_out, TransitionOut* out);
//   int64_t leapfrogs() const;
// `false` = backend failure.

// Host tree builder over a potential functor (one evaluation + read-back per leapfrog).
template <class Pot>
struct HostEngine {
    Pot& pot;
    const Config& cfg;
    Sampler<Pot> S;
    vec z, g;
    double pe = 0, aux[4] = {0, 0, 0, 0};
    HostEngine(Pot& p, const Config& c) : pot(p), cfg(c), S(p, c), z(p.dim()), g(p.dim()) {}
    int dim() const { return S.D; }
    int64_t leapfrogs() const { return S.leapfrogs; }
    bool set_state(const double* z0, double* pe_out, bool* finite) {
        for (int i = 0; i < S.D; ++i) z[i] = z0[i];
        if (!pot(z.data(), &pe, g.data(), aux)) return false;
        bool ok = std::isfinite(pe);
        for (int i = 0; i < S.D && ok; ++i) ok = std::isfinite(g[i]);
        *finite = ok;
        *pe_out = pe;
        return true;
    }
    bool transition(const double* r, tf::Key key, double step_size, const vec& inv_mass,
                    bool /*mass_changed*/, double* z_out, TransitionOut* out) {
        S.step_size = step_size;
        S.inv_mass = inv_mass;
        vec rv(r, r + S.D);
        Tree tree = S.build_tree(z, rv, pe, g, aux, key);
        if (S.eval_failed) return false;
        z = tree.z_prop;
        g = tree.g_prop;
        pe = tree.pe_prop;
        for (int i = 0; i < 4; ++i) aux[i] = tree.aux_prop[i];
        out->accept_prob = tree.sum_accept / tree.num_proposals;
        out->num_steps = tree.num_proposals;
        out->diverging = tree.diverging;
        out->pe = pe;
        for (int i = 0; i < 4; ++i) out->aux[i] = aux[i];
        for (int i = 0; i < S.D; ++i) z_out[i] = z[i];
        return true;
    }
};

// One chain's loop state around the tree builder: key plumbing, momentum draws, warm-up
// adaptation and draw collection (numpyro hmc.py init_kernel / sample_kernel,
// hmc_util.py warmup_adapter).  A tree engine runs between begin() and end(); several
// ChainDrivers can share one lock-step engine.
struct ChainDriver {
    const Config& cfg;
    int D;
    double* draws_out;
    Result* res;
    vec z, r, eps_n, inv_mass, mass_sqrt;
    tf::Key key_hmc{}, k_tr{};
    std::vector<Window> sched;
    int num_windows = 0, window_idx = 0, total = 0, kept = 0, start_idx = 0;
    double step_size = 1.0, used_step = 1.0, mean_accept = 0.0;
    bool mass_changed = true;
    DualAveraging ss;
    Welford mm;

    ChainDriver(const Config& c, int dim, double* draws, Result* r_)
        : cfg(c), D(dim), draws_out(draws), res(r_), z(dim), r(dim), eps_n(dim),
          inv_mass(dim, 1.0), mass_sqrt(dim, 1.0) {}

    // initial state (z0 or init_to_uniform with retries) through set_state(z, &pe, &finite)
    template <class SetState>
    int init(SetState&& set_state, const double* z0, tf::Key key) {
        std::vector<Site> sites = cfg.sites;
        if (sites.empty()) sites.push_back({0, D});
        // NUTS.init: key, key_init_model = split(key)
        tf::Key key_init;
        tf::split2(key, &key, &key_init);
        double pe = 0;
        bool finite = false;
        if (z0) {
            for (int i = 0; i < D; ++i) z[i] = z0[i];
            if (!set_state(z.data(), &pe, &finite)) return ST_EVAL_FAILED;
        } else {
            tf::Key k = key_init;
            for (int attempt = 0; attempt < 100 && !finite; ++attempt) {
                draw_init(&k, sites, cfg.init_radius, &z);
                if (!set_state(z.data(), &pe, &finite)) return ST_EVAL_FAILED;
            }
            if (!finite) return ST_NO_FINITE_INIT;
        }
        // init_kernel: key_hmc, key_wa, key_momentum = split(key, 3)
        tf::Key key_wa, key_mom0;
        tf::split3(key, &key_hmc, &key_wa, &key_mom0);
        sched = build_adaptation_schedule(cfg.num_warmup);
        num_windows = (int)sched.size();
        step_size = cfg.step_size;
        ss.init(std::log(10.0 * step_size));
        mm.init(D);
        total = cfg.num_warmup + cfg.num_samples;
        kept = cfg.num_samples / cfg.thinning;
        start_idx = cfg.num_warmup + cfg.num_samples % cfg.thinning;
        res->potential_energy.assign(kept, 0.0);
        res->accept_prob.assign(kept, 0.0);
        res->step_size.assign(kept, 0.0);
        res->aux0.assign(kept, 0.0);
        res->num_steps.assign(kept, 0);
        res->diverging.assign(kept, 0);
        return ST_OK;
    }

    // sample_kernel: key, key_momentum, key_transition = split(key, 3); momentum r
    void begin() {
        tf::Key k_mom;
        tf::split3(key_hmc, &key_hmc, &k_mom, &k_tr);
        tf::normal(k_mom, D, eps_n.data());
        for (int i = 0; i < D; ++i) r[i] = mass_sqrt[i] * eps_n[i];
        used_step = step_size;
    }

    // after the transition (z already holds the new state)
    void end(int it, const TransitionOut& out) {
        mass_changed = false;
        const double accept_prob = out.accept_prob;
        if (it < cfg.num_warmup) {  // warmup_adapter.update_fn
            const int t = it;
            if (cfg.adapt_step_size) {
                ss.update(cfg.target_accept_prob - accept_prob);
                double s = (t == cfg.num_warmup - 1) ? std::exp(ss.x_avg) : std::exp(ss.x_t);
                const double tiny = 1.1754943508222875e-38;
                step_size = s < tiny ? tiny : s;
            }
            const bool is_middle = (0 < window_idx) && (window_idx < num_windows - 1);
            if (cfg.adapt_mass_matrix && is_middle) mm.update(z);
            const bool at_end = t == sched[window_idx].end;
            if (at_end) window_idx += 1;
            if (at_end && is_middle) {
                if (cfg.adapt_mass_matrix) {
                    mm.final_regularized(&inv_mass);
                    for (int i = 0; i < D; ++i) mass_sqrt[i] = 1.0 / std::sqrt(inv_mass[i]);
                    mm.init(D);
                    mass_changed = true;
                }
                if (cfg.adapt_step_size) ss.init(std::log(10.0 * step_size));
            }
        } else {
            const int n = it - cfg.num_warmup + 1;
            mean_accept += (accept_prob - mean_accept) / n;
            if (out.diverging) res->total_divergences += 1;
            if (it >= start_idx && (it - start_idx) % cfg.thinning == cfg.thinning - 1) {
                const int idx = (it - start_idx) / cfg.thinning;
                for (int i = 0; i < D; ++i) draws_out[(size_t)idx * D + i] = z[i];
                res->potential_energy[idx] = out.pe;
                res->accept_prob[idx] = accept_prob;
                res->step_size[idx] = used_step;
                res->aux0[idx] = out.aux[0];
                res->num_steps[idx] = out.num_steps;
                res->diverging[idx] = out.diverging ? 1 : 0;
            }
        }
    }

    void finish(int64_t leapfrogs) {
        res->final_step_size = step_size;
        res->mean_accept_prob = mean_accept;
        res->total_leapfrogs = leapfrogs;
        res->inverse_mass_matrix = inv_mass;
    }
};

template <class Engine>
int run_chain_engine(Engine& E, const Config& cfg, const double* z0, tf::Key key,
                     double* draws_out, Result* res) {
    ChainDriver cd(cfg, E.dim(), draws_out, res);
    const int st = cd.init(
        [&](const double* zz, double* pe, bool* fin) { return E.set_state(zz, pe, fin); }, z0, key);
    if (st != ST_OK) return st;
    const int64_t leap0 = E.leapfrogs();
    for (int it = 0; it < cd.total; ++it) {
        cd.begin();
        TransitionOut out;
        if (!E.transition(cd.r.data(), cd.k_tr, cd.step_size, cd.inv_mass, cd.mass_changed,
                          cd.z.data(), &out))
            return ST_EVAL_FAILED;
        cd.end(it, out);
    }
    cd.finish(E.leapfrogs() - leap0);
    return ST_OK;
}

// Lock-step chains on one engine: every iteration all chains draw their momenta, the
// engine builds all trees together (VecEngine::transition_all), then every chain adapts.
template <class VecEngine>
int run_chains_lockstep(VecEngine& E, const Config& cfg, int n_chains, const double* z0,
                        const tf::Key* keys, double* draws_out, std::vector<Result>* res) {
    const int D = E.dim();
    const int kept = cfg.num_samples / cfg.thinning;
    res->assign(n_chains, Result{});
    std::vector<ChainDriver> cds;
    cds.reserve(n_chains);
    for (int c = 0; c < n_chains; ++c)
        cds.emplace_back(cfg, D, draws_out + (size_t)c * kept * D, &(*res)[c]);
    for (int c = 0; c < n_chains; ++c) {
        const int st = cds[c].init(
            [&](const double* zz, double* pe, bool* fin) { return E.set_state(c, zz, pe, fin); },
            z0 ? z0 + (size_t)c * D : nullptr, keys[c]);
        if (st != ST_OK) return st;
    }
    std::vector<TransitionOut> outs(n_chains);
    std::vector<int64_t> leaps(n_chains, 0);
    for (int it = 0; it < cds[0].total; ++it) {
        for (auto& cd : cds) cd.begin();
        if (!E.transition_all(cds, &outs)) return ST_EVAL_FAILED;
        for (int c = 0; c < n_chains; ++c) {
            cds[c].end(it, outs[c]);
            leaps[c] += outs[c].num_steps;
        }
    }
    for (int c = 0; c < n_chains; ++c) cds[c].finish(leaps[c]);
    return ST_OK;
}

template <class Pot>
int run_chain(Pot& pot, const Config& cfg, const double* z0, tf::Key key, double* draws_out,
              Result* res) {
    HostEngine<Pot> E(pot, cfg);
    return run_chain_engine(E, cfg, z0, key, draws_out, res);
}

}  // namespace nuts
