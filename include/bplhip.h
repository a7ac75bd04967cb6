/* bplhip.h -- C-ABI of libbplhip.so: the MI355X (gfx950) Dixon-Coles log-density +
 * gradient path and the NUTS driver around it.
 *
 * The reference (anguswilliams91/bpl-next) is pure Python over numpyro/JAX and has NO
 * FFI layer of its own; the seam this library replaces is numpyro's
 * `value_and_grad(potential_fn)(z)` call made once per leapfrog step by
 * `NUTS(self._model)` / `MCMC(...).run(...)`:
 *     bpl/dixon_coles.py:100-116            (basic model driver)
 *     bpl/extended_dixon_coles.py:293-316   (extended model driver)
 * Each entry point below cites the reference code whose work it takes over.
 *
 * Conventions
 *   - plain C linkage, plain pointers and sizes, no C++/torch types;
 *   - every function returns 0 (BPLHIP_OK) or a negative BPLHIP_E* code; the message
 *     is kept per context (bplhip_last_error); no exception or abort crosses the ABI;
 *   - "device" pointers are HIP device pointers owned by the caller (e.g.
 *     torch.Tensor.data_ptr()); "host" pointers are ordinary host memory;
 *   - `stream` is a hipStream_t passed as void* (NULL = the null stream).  All device
 *     work is enqueued on it; functions documented "asynchronous" do not synchronise;
 *   - a context is bound to one device and is not thread-safe; distinct contexts may be
 *     used concurrently from distinct host threads;
 *   - a non-finite potential is NOT an error (NUTS treats it as a divergence): it is
 *     returned as +inf / nan.
 *
 * Latent vector layout (flat, numpyro's sorted-site-name order; all float64):
 *   basic    attack_decentered[T], corr_coef_raw, defence_decentered[T], home_advantage,
 *            mean_defence, std_attack, std_defence                         D = 2T+5
 *   extended attack_coefficients[K], corr_coef_raw, defence_coefficients[K],
 *            home_advantage_decentered[T], mean_defence, mean_home_advantage,
 *            standardised_attack[T], standardised_defence[T], std_attack, std_defence,
 *            std_home_advantage, u                                         D = 3T+2K+7
 */
#ifndef BPLHIP_H
#define BPLHIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define BPLHIP_ABI_VERSION 1

enum {
    BPLHIP_OK = 0,
    BPLHIP_EINVAL = -1,   /* bad argument (null pointer, size, index out of range) */
    BPLHIP_ESTATE = -2,   /* call out of order (e.g. logp_grad before set_fixtures) */
    BPLHIP_EHIP = -3,     /* a HIP runtime call failed, or a device-side hand-off timed out
                           * (the kernels' waits for each other are bounded; one that expires raises a
                           * host-visible fault word which the NEXT entry point -- or the running
                           * sampler at its next synchronisation -- reports and clears; the affected
                           * evaluations' outputs are NaN); see bplhip_last_error  */
    BPLHIP_ENOMEM = -4,
    BPLHIP_EUNSUPPORTED = -5,
    BPLHIP_ENUMERIC = -6  /* NUTS could not find a finite initial point            */
};

enum {
    BPLHIP_MODEL_BASIC = 0,    /* bpl/dixon_coles.py:39-84           */
    BPLHIP_MODEL_EXTENDED = 1, /* bpl/extended_dixon_coles.py:78-248 */
    BPLHIP_MODEL_DYNAMIC = 2,  /* bpl/dynamic_dixon_coles.py:63-247 (bound through
                                  bplhip_set_fixtures_dynamic)       */
    BPLHIP_MODEL_NEUTRAL = 3   /* bpl/neutral_dixon_coles.py:102-283 (bound through
                                  bplhip_set_fixtures_neutral)       */
};

typedef struct bplhip_ctx bplhip_ctx;

/* ABI version of the loaded library (== BPLHIP_ABI_VERSION of the header it was built
 * against). */
int bplhip_abi_version(void);

/* Create / destroy a context on HIP device `device_id`. */
int bplhip_create(bplhip_ctx** out, int device_id);
void bplhip_destroy(bplhip_ctx* ctx);

/* Last error message of this context ("" if none); `ctx` may be NULL for the message of
 * a failed bplhip_create. The pointer stays valid until the next call on the context. */
const char* bplhip_last_error(const bplhip_ctx* ctx);

/* Bind the model arguments.  Replaces the concrete (non-traced) arguments the reference
 * hands to `mcmc.run(...)`: bpl/dixon_coles.py:108-116 (home_ind, away_ind, num_teams,
 * home_goals, away_goals) and bpl/extended_dixon_coles.py:303-316 (+ team_covariates,
 * weights = exp(-epsilon*time_diff), optionally rescaled: :202-205).
 *
 *   home_idx/away_idx  device u16[n]  team indices in [0, n_teams) -- the reference's
 *                                     storage dtype, bpl/base.py:16-22, parse_teams
 *                                     bpl/_util.py:115-135
 *   home_goals/away_goals device u8[n]
 *   weights            device f32[n] or NULL (unweighted)
 *   covariates         HOST f64[n_teams*k] row-major, ALREADY standardised
 *                      (bpl/extended_dixon_coles.py:124-127), or NULL with k = 0;
 *                      must be NULL for the basic model
 *
 * The arrays are read once (synchronously on `stream`) and re-laid-out into a library
 * owned SoA copy (sorted by (home,away) pair, every pair's run padded to the lane width,
 * the whole padded to the tile size); the caller's buffers are not referenced after the
 * call returns.  (The padding fixtures carry goals 255-255 and weight 0; a real 255-255 fixture is
 * legal -- which fixtures of a lane are real is recorded separately.) */
int bplhip_set_fixtures(bplhip_ctx* ctx, int model_kind, int64_t n, int32_t n_teams,
                        const uint16_t* home_idx, const uint16_t* away_idx,
                        const uint8_t* home_goals, const uint8_t* away_goals,
                        const float* weights, const double* covariates, int32_t k,
                        void* stream);

/* Bind the arguments of the dynamic (time-varying, neutral-venue) model,
 * bpl/dynamic_dixon_coles.py:63-247 as run at :286-300: per-gameweek hyper-parameters,
 * [G,T] tables, `gameweek` (0-based, device u16[n]) and `neutral_venue` (device u8[n], 0/1).
 * Latent layout (sorted site names, D = 7GT + 10G + 2 + 2K): attack_coefficients[K],
 * away_attack_decentered[G,T], away_defence_decentered[G,T], corr_coef_raw,
 * defence_coefficients[K], home_attack_decentered[G,T], home_defence_decentered[G,T],
 * mean_away_attack[G], mean_away_defence[G], mean_defence, mean_home_attack[G],
 * mean_home_defence[G], standardised_attack[G,T], standardised_defence[G,T], std_attack[G],
 * std_away_attack[G], std_away_defence[G], std_defence[G], std_home_attack[G],
 * std_home_defence[G], u[G,T].  `random_walk` 1 = the INTENDED model, attack[g] = attack[g-1] +
 * standardised_attack[g]*std_attack[g] (:192-218 discard that update);
 * 0 = the code as written (attack = defence = 0).  After this call logp_grad / nuts_run
 * work on the dynamic model; draws are mapped by bplhip_constrain_dynamic. */
int bplhip_set_fixtures_dynamic(bplhip_ctx* ctx, int64_t n, int32_t n_teams,
                                int32_t n_gameweeks, const uint16_t* home_idx,
                                const uint16_t* away_idx, const uint8_t* home_goals,
                                const uint8_t* away_goals, const uint16_t* gameweek,
                                const uint8_t* neutral_venue, const double* covariates,
                                int32_t k, int32_t random_walk, void* stream);

/* Dynamic model: HOST draws f64[s, D] -> constrained/deterministic sites f64[s, G, T]
 * (any output may be NULL): attack, defence (`attack_j`/`defence_j` of :195-218),
 * home_attack, away_attack, home_defence, away_defence (:142-189). */
int bplhip_constrain_dynamic(bplhip_ctx* ctx, const double* z_draws, int64_t s,
                             double* attack, double* defence, double* home_attack,
                             double* away_attack, double* home_defence,
                             double* away_defence);

/* Bind the arguments of the neutral-venue model, bpl/neutral_dixon_coles.py:102-283 as run
 * at :342-356: `neutral_venue` device u8[n] (0/1), `weights` device f32[n] = the final
 * per-fixture weights (time decay x game weights, :251-257; NULL = all ones), covariates as
 * for the extended model.  Latent layout (sorted site names, D = 6T + 2K + 13):
 * attack_coefficients[K], away_attack_decentered[T], away_defence_decentered[T],
 * corr_coef_raw, defence_coefficients[K], home_attack_decentered[T],
 * home_defence_decentered[T], mean_away_attack, mean_away_defence, mean_defence,
 * mean_home_attack, mean_home_defence, standardised_attack[T], standardised_defence[T],
 * std_attack, std_away_attack, std_away_defence, std_defence, std_home_attack,
 * std_home_defence, u.  World-Cup variant (bpl/neutral_dixon_coles_WC.py:83-232): `home_conf`,
 * `away_conf` device u8[n] confederation indices and n_conf > 0 add the site
 * confederation_strength_decentered[n_conf] (after away_defence_decentered, D += n_conf);
 * NULL, NULL, 0 for the plain neutral model.  After this call logp_grad / nuts_run work on
 * the neutral model. */
int bplhip_set_fixtures_neutral(bplhip_ctx* ctx, int64_t n, int32_t n_teams,
                                const uint16_t* home_idx, const uint16_t* away_idx,
                                const uint8_t* home_goals, const uint8_t* away_goals,
                                const uint8_t* neutral_venue, const uint8_t* home_conf,
                                const uint8_t* away_conf, int32_t n_conf, const float* weights,
                                const double* covariates, int32_t k, void* stream);

/* Tuning knobs (no reference counterpart; defaults are the measured best):
 *   "device_nuts" 1 (default) = NUTS tree builder on the device for every model: leaf
 *            bookkeeping in the tail of the evaluation kernel (basic/extended models,
 *            n_teams <= 64) or in leaf launches after it (everything else);
 *            0 = host tree builder (one read-back per leapfrog; the cross-check engine).
 *   "persistent_nuts" 1 (default) = the whole chain on the device (adaptation included, the
 *            host only enqueues evaluations); 0 = device trees with host-side adaptation
 *            (models of the evaluation kernel's tail only).
 *   "persistent_kernel" 1 (default) = one resident chain of the basic / extended model runs its
 *            leapfrogs INSIDE one launch (the streaming workgroups poll the next position);
 *            0 = one launch per leapfrog.
 *   "persist_spec" 1 (default) = inside that resident launch the next position goes out as soon as
 *            the gradient exists and the leaf is booked beside the next step's prior part (a U-turn or
 *            divergence -- once per transition -- discards one evaluation); 0 = leaf first, then publish
 *   "fused_small" 1 (default) = neutral / dynamic evaluations run as ONE launch: small ones on
 *            one workgroup per chain with everything in LDS (neutral) / one workgroup per four
 *            teams with phases behind grid barriers (dynamic), larger ones sliced over all CUs
 *            (a slice of the fixtures per workgroup, tree barriers) while the slices fit the LDS;
 *            0 = always the multi-launch path.
 *   "dyn_gather" 1 (default) = the dynamic model's small single launch hands the adjoints over as one
 *            16-byte record per fixture and gathers them per cell (host-built incidence lists; taken when no
 *            cell takes part in more than 16 fixtures); 0 = float64 atomics into the cells' accumulators
 *   "neu_runs" 1 (default) = the neutral model's sliced single launch works rates, tau terms and adjoints
 *            out once per (venue, home, away) run from seven sums over the run's fixtures (taken when every
 *            wave's part of the slice holds at most 15 runs); 0 = per fixture
 *   "dyn_big_wgs" 0 (default) = the sliced single launch uses one workgroup per CU, and the
 *            dynamic model takes it past 1024 fixtures per team workgroup; > 0 = that many
 *            workgroups, and the dynamic model takes the sliced form whatever its size.
 *   "dense_pairs" 1 (default) = a complete pair table (every ordered pair h != a present) of 4096
 *            pairs or more -- every complete table past 64 teams -- takes the rho bounds from the
 *            top two table entries per role (O(teams)); 0 = always walk the pair table.
 *   "pair_order" -1 (default) = past 64 teams the fixtures are laid out along the Z-order curve over
 *            (home, away), so that a workgroup's slice touches ~sqrt of its pairs' teams; up to 64
 *            teams in (home, away) order.  0 / 1 = (home, away) / Z-order whatever the league's
 *            size.  Applies at the next bplhip_set_fixtures; results do not depend on it beyond
 *            the order of the float32 run sums.
 *   "max_wg" streaming workgroups per evaluation, 1..255 (default 255: with the prior workgroup
 *            one per CU; an accumulator row counts its contributors in 8 bits); applies at the next
 *            bplhip_set_fixtures
 *   "gridy_max_chains" 8 (default) = bplhip_nuts_run_chains keeps up to this many chains as grid.y
 *            copies of the single-chain NUTS-aware launch; more share the chain-vectorised kernel
 *   "vec_min_chains" 12 (default) = bplhip_logp_grad_batched takes the chain-vectorised kernel
 *            (dc_vec: the fixtures are read once per 8 chains) from this many chains on; 0 = never
 *   "vec_tiles_per_wave" 0 (default) = the chain-vectorised partitions scale with the chain count
 *            (1x / 2x / 3x the single-chain tiles per wave, the thinnest whose grid fits the chip in one round); > 0 = this many; next bplhip_set_fixtures
 *   "chunk_graph" 1 (default) = a persistent sampler replays its chunk of 256 leapfrog launches as
 *            ONE hipGraph (falls back to plain launches by itself if a capture fails); 0 = always
 *            launch by launch
 *   "debug_raise_fault" test hook: ORs `value` into the context's fault word, as a kernel whose
 *            bounded wait expired would (the next entry point returns BPLHIP_EHIP)
 *   "active_waves" waves per workgroup that own tiles: 0 (default) = automatic -- short
 *            streams get a second partition with 4 of 8 waves owning tiles, used while the
 *            launch's workgroups still find a CU each; 1..8 = one fixed partition; applies
 *            at the next bplhip_set_fixtures */
int bplhip_set_option(bplhip_ctx* ctx, const char* name, int value);

/* D of the bound model (negative error code if no fixtures are bound). */
int bplhip_latent_dim(const bplhip_ctx* ctx);

/* THE HOT PATH.  U(z) = -log p(z, data) in unconstrained space and dU/dz -- what
 * numpyro's `value_and_grad(potential_fn)(z)` computes once per leapfrog from the model
 * declared at bpl/dixon_coles.py:39-84 / bpl/extended_dixon_coles.py:78-248, including
 * compute_corr_coef_bounds (bpl/_util.py:17-31) and dixon_coles_correlation_term
 * (bpl/_util.py:35-93).  Asynchronous, stream ordered.
 *   z          device f64[D]
 *   potential  device f64[1]
 *   grad       device f64[D]
 *   aux        device f64[4] or NULL: {corr_coef (the `deterministic` site of
 *              bpl/dixon_coles.py:80), LB, UB, raw} */
int bplhip_logp_grad(bplhip_ctx* ctx, const double* z, double* potential, double* grad,
                     double* aux, void* stream);

/* The same for `n_chains` independent latent vectors in one launch sequence (numpyro
 * chain_method="vectorized", reachable through mcmc_kwargs at bpl/dixon_coles.py:105).
 *   z [n_chains, D], potential [n_chains], grad [n_chains, D], aux [n_chains, 4] or NULL */
int bplhip_logp_grad_batched(bplhip_ctx* ctx, int32_t n_chains, const double* z,
                             double* potential, double* grad, double* aux, void* stream);

/* Pre-record `count` back-to-back evaluations z[i % n_z] -> (potential[i % n_z],
 * grad[i % n_z]) as one hipGraph and replay it `replays` times on `stream`
 * (asynchronous).  This is how the NUTS driver issues the 2^depth leapfrogs of a tree
 * doubling without a host round trip per launch; it is exposed for the op-level
 * benchmark. */
int bplhip_logp_grad_graph(bplhip_ctx* ctx, int32_t count, int32_t n_z, const double* z,
                           double* potential, double* grad, int32_t replays, void* stream);

/* ---- NUTS driver: numpyro.infer.{NUTS,MCMC} as configured at
 * bpl/dixon_coles.py:100-116 (all NUTS defaults; num_warmup / num_samples forwarded). */
typedef struct bplhip_nuts_cfg {
    int32_t num_warmup;         /* bpl/dixon_coles.py:90 default 500                */
    int32_t num_samples;        /* bpl/dixon_coles.py:91 default 1000               */
    int32_t max_tree_depth;     /* numpyro default 10                               */
    int32_t adapt_step_size;    /* numpyro default 1                                */
    int32_t adapt_mass_matrix;  /* numpyro default 1 (diagonal, regularised)        */
    int32_t thinning;           /* numpyro MCMC default 1                           */
    double step_size;           /* numpyro default 1.0                              */
    double target_accept_prob;  /* numpyro default 0.8                              */
    double init_radius;         /* init_to_uniform(radius=2)                        */
    double max_delta_energy;    /* numpyro default 1000                             */
} bplhip_nuts_cfg;

/* Fill `cfg` with numpyro's defaults as reached from bpl/dixon_coles.py:100-106. */
void bplhip_nuts_default_cfg(bplhip_nuts_cfg* cfg);

typedef struct bplhip_nuts_stats {
    /* per kept draw, HOST arrays of length num_samples/thinning, each may be NULL */
    double* potential_energy;
    double* accept_prob;
    double* step_size;
    int32_t* num_steps;
    int32_t* diverging;
    double* corr_coef;          /* deterministic site `corr_coef` of each draw      */
    /* scalars, filled by the call */
    double final_step_size;
    double mean_accept_prob;
    int64_t total_leapfrogs;    /* potential+gradient evaluations, warm-up included */
    int64_t total_divergences;  /* post warm-up                                     */
    double wall_seconds;
    double* inverse_mass_matrix; /* HOST f64[D] or NULL: adapted diagonal           */
} bplhip_nuts_stats;

/* Run one chain.  `seed_hi:seed_lo` is the 2x32 threefry key (jax.random.PRNGKey(s) ==
 * {0, s} for a 32-bit s; chain c of a multi-chain run uses split(key, num_chains)[c]).
 *   z0         HOST f64[D] or NULL (NULL = init_to_uniform(radius) + retry until finite,
 *              numpyro find_valid_initial_params; non-NULL = run_kwargs init_params,
 *              bpl/dixon_coles.py:115)
 *   draws_out  HOST f64[num_samples/thinning, D] unconstrained draws (post warm-up)
 * Synchronous (returns when the chain has finished). */
int bplhip_nuts_run(bplhip_ctx* ctx, const bplhip_nuts_cfg* cfg, const double* z0,
                    uint32_t seed_hi, uint32_t seed_lo, double* draws_out,
                    bplhip_nuts_stats* stats, void* stream);

/* Run `n_chains` chains together on this GPU (numpyro chain_method="vectorized",
 * MCMC(num_chains=...) at bpl/dixon_coles.py:101-106): persistent chains, every launch
 * advances every unfinished chain by one leapfrog.  Every model; BPLHIP_EUNSUPPORTED only
 * when the run's momentum draws (n_chains * iterations * D doubles) exceed 16 GiB or with
 * "persistent_nuts" 0 outside the basic / extended models with n_teams <= 64 (then run the
 * chains one after another with bplhip_nuts_run).
 *   z0        HOST f64[n_chains, D] or NULL       seeds  HOST u32[n_chains, 2] (hi, lo)
 *   draws_out HOST f64[n_chains, num_samples/thinning, D]
 *   stats     n_chains statistics records, or NULL; wall_seconds is the whole run's.
 * Chain c follows the same key sequence as bplhip_nuts_run with seeds[c]. */
int bplhip_nuts_run_chains(bplhip_ctx* ctx, const bplhip_nuts_cfg* cfg, int32_t n_chains,
                           const double* z0, const uint32_t* seeds, double* draws_out,
                           bplhip_nuts_stats* stats, void* stream);

/* Map unconstrained draws to the constrained / deterministic sites the reference reads
 * from `mcmc.get_samples()` (bpl/dixon_coles.py:118-122,
 * bpl/extended_dixon_coles.py:319-331).  HOST in, HOST out; any output may be NULL.
 *   z_draws f64[s, D]
 *   attack, defence f64[s, T]; home_advantage f64[s] (basic) or f64[s, T] (extended);
 *   corr_coef f64[s]  (needs the rho bounds over all bound fixtures) */
int bplhip_constrain(bplhip_ctx* ctx, const double* z_draws, int64_t s, double* attack,
                     double* defence, double* home_advantage, double* corr_coef);

/* ---- predict path on the device (post-fit).
 * bplhip_predict_set_posterior uploads the posterior draws the reference keeps as
 * attributes after fit (bpl/dixon_coles.py:118-122): attack/defence f64[s,t],
 * home_advantage f64[s] (basic) or f64[s,t] (extended, home_advantage_per_team = 1),
 * corr_coef f64[s] -- HOST pointers.  bplhip_predict_score_proba evaluates
 * `predict_score_proba` (bpl/dixon_coles.py:139-163, bpl/extended_dixon_coles.py:360-399):
 * out[i] = mean over draws of exp(tau term) * Poisson(x_i; home rate) * Poisson(y_i; away
 * rate), HOST u16[m] in, HOST f64[m] out, synchronous.  No fixtures need to be bound. */
int bplhip_predict_set_posterior(bplhip_ctx* ctx, int32_t s, int32_t t, const double* attack,
                                 const double* defence, const double* home_advantage,
                                 int32_t home_advantage_per_team, const double* corr_coef);
int bplhip_predict_score_proba(bplhip_ctx* ctx, int64_t m, const uint16_t* home_idx,
                               const uint16_t* away_idx, const uint16_t* home_goals,
                               const uint16_t* away_goals, double* out, void* stream);
/* `predict_score_grid_proba` (bpl/base.py:74-111): for each of the m fixtures the whole
 * (max_goals+1) x (max_goals+1) grid of scoreline probabilities, out[i, x, y] = mean over draws of
 * exp(tau term) * Poisson(x; home rate) * Poisson(y; away rate) -- the primitive the reference's
 * predict_outcome_proba (:113-148), predict_score_n_proba / predict_concede_n_proba (:248-348)
 * and sample_score / sample_outcome (:150-246) are reductions of.  One wave per fixture on the
 * matrix cores (float32 pmf outer products, float64 accumulation across 64-draw blocks).
 * HOST u16[m] in, HOST f64[m, max_goals+1, max_goals+1] out, max_goals <= 63, synchronous. */
int bplhip_predict_score_grid(bplhip_ctx* ctx, int64_t m, const uint16_t* home_idx,
                              const uint16_t* away_idx, int32_t max_goals, double* out,
                              void* stream);
/* ... the same grids as HOST f32[m, max_goals+1, max_goals+1] -- the dtype the reference's
 * predict_score_grid_proba returns (jax float32, bpl/base.py:74-111): the float64 accumulators
 * are rounded once at the store, and half as many bytes come back over PCIe (the copy back is
 * most of a large query: 97 280 grids of 16 x 16 are 199 MB as float64). */
int bplhip_predict_score_grid_f32(bplhip_ctx* ctx, int64_t m, const uint16_t* home_idx,
                                  const uint16_t* away_idx, int32_t max_goals, float* out,
                                  void* stream);

/* The same three entry points for the venue-aware rate form of the neutral-venue family:
 * `_calculate_expected_goals` of bpl/neutral_dixon_coles.py:399-423 (four per-team offsets that
 * are switched off at neutral venues), bpl/neutral_dixon_coles_WC.py:385-424 (plus the difference
 * of the two sides' confederation strengths) and, for the dynamic class, the rates of its MODEL,
 * bpl/dynamic_dixon_coles.py:220-231, on the tables of one gameweek:
 *   on = 1 - neutral_venue,  dc = confederation_strength[home_conf] - confederation_strength[away_conf]
 *   log home rate = attack[h] - defence[a] + on (home_attack[h] - away_defence[a]) + dc
 *   log away rate = attack[a] - defence[h] + on (away_attack[a] - home_defence[h]) - dc
 * (A DELIBERATE DEVIATION for the dynamic class: upstream's own predict-time
 * `_calculate_expected_goals`, bpl/dynamic_dixon_coles.py:336-361, differs from the model it was fitted
 * with -- "+ on away_defence[a]" in the home rate, "- on away_attack[a] - on home_defence[h]" in the away
 * rate -- and indexes no gameweek (the class is unfinished upstream).  Predictions
 * here use the rates the likelihood used.)
 * set_posterior_venue: six HOST f64[s,t] tables, confederation_strength HOST f64[s,n_conf] or NULL
 * with n_conf = 0, corr_coef f64[s].  Queries: neutral_venue HOST u8[m] (required), home_conf /
 * away_conf HOST u16[m] exactly when the posterior has confederations (else NULL).  A context holds
 * ONE posterior: the plain and the venue entry points cannot be mixed (BPLHIP_ESTATE). */
int bplhip_predict_set_posterior_venue(bplhip_ctx* ctx, int32_t s, int32_t t, const double* attack,
                                       const double* defence, const double* home_attack,
                                       const double* away_attack, const double* home_defence,
                                       const double* away_defence, int32_t n_conf,
                                       const double* confederation_strength,
                                       const double* corr_coef);
int bplhip_predict_score_proba_venue(bplhip_ctx* ctx, int64_t m, const uint16_t* home_idx,
                                     const uint16_t* away_idx, const uint16_t* home_goals,
                                     const uint16_t* away_goals, const uint8_t* neutral_venue,
                                     const uint16_t* home_conf, const uint16_t* away_conf,
                                     double* out, void* stream);
int bplhip_predict_score_grid_venue(bplhip_ctx* ctx, int64_t m, const uint16_t* home_idx,
                                    const uint16_t* away_idx, const uint8_t* neutral_venue,
                                    const uint16_t* home_conf, const uint16_t* away_conf,
                                    int32_t max_goals, double* out, void* stream);
int bplhip_predict_score_grid_venue_f32(bplhip_ctx* ctx, int64_t m, const uint16_t* home_idx,
                                        const uint16_t* away_idx, const uint8_t* neutral_venue,
                                        const uint16_t* home_conf, const uint16_t* away_conf,
                                        int32_t max_goals, float* out, void* stream);

/* Self-test of the library's own float64 device math (csrc/dc_kernels.hip.h, namespace
 * dc::lean -- the short exp / log / log1p / reciprocal the float64 kernels use on their critical
 * paths; no reference counterpart).  which: 0 exp(x), 1 log(x), 2 log(1 + x) for x >= 0, 3 1/x for
 * normal x.  HOST f64[n] in and out, synchronous. */
int bplhip_selftest_math(bplhip_ctx* ctx, int32_t which, int64_t n, const double* in, double* out);

/* threefry2x32 helpers with jax.random semantics (jax 0.4.24, non-partitionable
 * threefry): used by the Python host for key plumbing (random.split for multi-chain
 * runs, bpl/dixon_coles.py:107).  out has 2*n words: n keys (hi, lo). */
void bplhip_threefry_split(uint32_t key_hi, uint32_t key_lo, int32_t n, uint32_t* out);
/* n raw 32-bit draws, jax.random.bits(key, (n,), uint32). */
void bplhip_threefry_bits(uint32_t key_hi, uint32_t key_lo, int32_t n, uint32_t* out);

#ifdef __cplusplus
}
#endif
#endif /* BPLHIP_H */
