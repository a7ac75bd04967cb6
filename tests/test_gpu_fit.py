"""GPU end-to-end tests through the public API: the reference's own property tests
(tests/test_dixon_coles.py:7-14, tests/test_base_models.py:15-96,
tests/test_extended_dixon_coles.py:4-47) with the HIP path underneath, plus the
sampler-level parity ladder of SURVEY.md §7.3-5 (L3: posterior moments of the HIP sampler
vs the same driver on the CPU oracle potential, within Monte-Carlo error)."""
import numpy as np
import pytest

import cases
import dc_oracle as O
import dc_oracle_c as OC

pytestmark = pytest.mark.gpu

from bpl import DixonColesMatchPredictor, ExtendedDixonColesMatchPredictor  # noqa: E402
from bpl.base import MAX_GOALS  # noqa: E402

MODELS = [DixonColesMatchPredictor, ExtendedDixonColesMatchPredictor]


@pytest.fixture(scope="module")
def fitted(hip_ctx):
    """One 100+100 fit per model, shared by the property tests (the reference refits for
    every test; the assertions are the same)."""
    td = O.dummy_data_recipe()
    return {cls: cls().fit(td, num_samples=100, num_warmup=100) for cls in MODELS}, td


def test_fit_default(hip_ctx, dummy_data):  # tests/test_dixon_coles.py:7-14
    model = DixonColesMatchPredictor().fit(dummy_data)
    assert model.attack is not None and model.defence is not None
    assert model.home_advantage is not None and model.teams is not None
    assert model.corr_coef is not None
    assert model.attack.shape == (1000, 20) and model.home_advantage.shape == (1000,)
    info = model.mcmc_info_
    assert info["divergences"] == 0
    assert 0.6 < info["accept_prob"].mean() < 0.97
    # goals are iid Poisson(2.1) / Poisson(1.7): exp(home_advantage) ~ 2.1 / 1.7
    assert abs(model.home_advantage.mean() - np.log(2.1 / 1.7)) < 0.08
    # corr_coef recorded on the device during sampling == host-side constrain()
    assert np.allclose(info["corr_coef"], model.corr_coef, atol=1e-6)


@pytest.mark.parametrize("model_cls", MODELS)
def test_predict_score_proba(fitted, model_cls):
    models, dd = fitted
    model = models[model_cls]
    probs = model.predict_score_proba(dd["home_team"], dd["away_team"], dd["home_goals"], dd["away_goals"])
    assert np.all((probs >= 0) & (probs <= 1))
    assert 0 <= model.predict_score_proba("0", "1", 1, 0)[0] <= 1


@pytest.mark.parametrize("model_cls", MODELS)
def test_predict_outcome_proba(fitted, model_cls):
    models, dd = fitted
    model = models[model_cls]
    probs = model.predict_outcome_proba(dd["home_team"], dd["away_team"])
    assert np.allclose(probs["home_win"] + probs["away_win"] + probs["draw"], 1.0, atol=1e-5)
    p = model.predict_outcome_proba("0", "1")
    assert p["home_win"] + p["away_win"] + p["draw"] == pytest.approx(1.0, abs=1e-5)


@pytest.mark.parametrize("model_cls", MODELS)
def test_predict_score_n_and_concede_n_proba(fitted, model_cls):
    models, _ = fitted
    model = models[model_cls]
    n = np.arange(MAX_GOALS + 1)
    ph, pa = model.predict_score_n_proba(n, "0", "1"), model.predict_score_n_proba(n, "0", "1", home=False)
    assert np.all((ph >= 0) & (ph <= 1)) and sum(ph) == pytest.approx(1.0, abs=1e-5)
    assert np.all((pa >= 0) & (pa <= 1)) and sum(pa) == pytest.approx(1.0, abs=1e-5)
    assert sum(ph * n) > sum(pa * n)
    ch, ca = model.predict_concede_n_proba(n, "0", "1"), model.predict_concede_n_proba(n, "0", "1", home=False)
    assert sum(ch) == pytest.approx(1.0, abs=1e-5) and sum(ch * n) < sum(ca * n)
    a = model.predict_concede_n_proba(1, "0", "1")
    b = model.predict_score_n_proba(1, "1", "0", home=False)
    assert a.tolist() == pytest.approx(b.tolist(), abs=1e-5)


def test_time_weighted_vs_not(hip_ctx, timed_dummy_data):  # tests/test_extended_dixon_coles.py:4-25
    m0 = ExtendedDixonColesMatchPredictor().fit(timed_dummy_data)
    a0, d0 = m0.attack.mean(axis=0), m0.defence.mean(axis=0)
    assert abs(a0[1] - a0[0]) < 0.05 and abs(d0[1] - d0[0]) < 0.05
    m1 = ExtendedDixonColesMatchPredictor().fit(timed_dummy_data, epsilon=1)
    a1, d1 = m1.attack.mean(axis=0), m1.defence.mean(axis=0)
    assert (a1[1] - a1[0]) > 0.75 and abs(d1[1] - d1[0]) > 0.75


def test_epsilon(hip_ctx, timed_dummy_data):  # tests/test_extended_dixon_coles.py:28-47
    """Increasing epsilon increases the impact of time weighting.  The reference asserts
    `delta_attack_2 > 1.5 * delta_attack_1` on ONE 500 + 1000 numpyro chain (seed 42; its "defence"
    deltas re-read .attack, :36,:42, so that is its only assertion).  On this two-team recipe the
    attack gap and the defence gap trade off against each other (their SUM is what the data pin),
    so a 1000-draw mean of the attack gap alone has a Monte-Carlo s.d. of 0.08-0.10 and the ratio
    of two of them one of 0.13: profiles/r03/epsilon_ratio.txt (tools/epsilon_ratio.py, 48 seeds at
    the reference's 500 + 1000) has the ratio at 1.39 +- 0.13, range 1.11 .. 1.67, above 1.5 for 11
    of 48 seeds, and the exact posterior gaps (8 x 5000 draws) at 0.925 / 1.280, ratio 1.383;
    profiles/r03/epsilon_ratio_cpu.txt gets the same gaps from an independent NUTS on torch
    autograd of the literal model transcription.  So the reference's 1.5 is a value its seed
    happened to reach, inside the sampled range, not a property of the posterior.  Asserted here,
    over 24 seeds: the reference's assertion verbatim for every seed whose chains reach it (at
    least one must), 1.5 inside the sampled range, and the seed-averaged gaps at their exact
    values."""
    gaps = []
    for seed in [42] + list(range(23)):
        m1 = ExtendedDixonColesMatchPredictor().fit(timed_dummy_data, epsilon=1, random_state=seed)
        m2 = ExtendedDixonColesMatchPredictor().fit(timed_dummy_data, epsilon=2, random_state=seed)
        attack_epsilon1, attack_epsilon2 = m1.attack.mean(axis=0), m2.attack.mean(axis=0)
        delta_attack_1 = abs(attack_epsilon1[1] - attack_epsilon1[0])
        delta_attack_2 = abs(attack_epsilon2[1] - attack_epsilon2[0])
        d1, d2 = m1.defence.mean(axis=0), m2.defence.mean(axis=0)
        gaps.append((delta_attack_1, delta_attack_2, abs(d1[1] - d1[0]), abs(d2[1] - d2[0])))
        # every single chain: the well-determined quantity, attack + defence gap together, widens by
        # 1.36 .. 1.47 (48 seeds, profiles/r03/epsilon_ratio.txt)
        assert 1.3 < (delta_attack_2 + gaps[-1][3]) / (delta_attack_1 + gaps[-1][2]) < 1.55
    gaps = np.array(gaps)
    ratio = gaps[:, 1] / gaps[:, 0]
    print(f"ratio: mean {ratio.mean():.3f} min {ratio.min():.3f} max {ratio.max():.3f}; "
          f"gaps {gaps.mean(axis=0).round(3)}")
    reached = ratio > 1.5
    assert reached.any() and not reached.all()          # 1.5 lies inside the sampled range
    for delta_attack_1, delta_attack_2, _, _ in gaps[reached]:
        assert delta_attack_2 > 1.5 * delta_attack_1    # the reference's assertion, verbatim
    # seed-averaged gaps against the exact posterior values (s.d. of a 24-seed mean: 0.02)
    assert abs(gaps[:, 0].mean() - 0.925) < 0.07 and abs(gaps[:, 1].mean() - 1.280) < 0.08
    assert abs(gaps[:, 2].mean() - 0.896) < 0.07 and abs(gaps[:, 3].mean() - 1.291) < 0.08
    assert 1.25 < ratio.mean() < 1.52


def test_covariates_rescale_and_multichain(hip_ctx, dummy_data):
    """Paths the reference tests never reach: team_covariates,
    rescale_weights, num_chains > 1 (sequential chains on one GPU), thinning."""
    td = dict(dummy_data)
    rs = np.random.RandomState(0)
    td["team_covariates"] = {str(i): list(rs.normal(size=3)) for i in range(20)}
    td["time_diff"] = np.linspace(3, 0, 380)
    m = ExtendedDixonColesMatchPredictor().fit(
        td, num_warmup=60, num_samples=40, epsilon=0.5, rescale_weights=True,
        mcmc_kwargs={"num_chains": 2, "thinning": 2, "progress_bar": False})
    assert m.attack.shape == (40, 20)  # 2 chains x 40/2 draws
    assert m.attack_coefficients.shape == (40, 3) and m.defence_coefficients.shape == (40, 3)
    assert m.rho.shape == (40,) and np.all(np.abs(m.rho) < 1)
    assert np.all(m.std_attack > 0) and np.isfinite(m.home_advantage).all()
    # the two chains are keyed by split(PRNGKey(42), 2): different draws
    assert not np.allclose(m.attack[:20], m.attack[20:])
    with pytest.raises(TypeError):
        DixonColesMatchPredictor().fit(dummy_data, num_warmup=1, num_samples=1, mcmc_kwargs={"bogus": 1})


def test_init_params_and_reproducibility(hip_ctx, dummy_data):
    z0 = np.random.RandomState(1).uniform(-0.1, 0.1, 45)
    kw = dict(num_warmup=30, num_samples=20, run_kwargs={"init_params": z0})
    a = DixonColesMatchPredictor().fit(dummy_data, random_state=7, **kw)
    b = DixonColesMatchPredictor().fit(dummy_data, random_state=7, **kw)
    c = DixonColesMatchPredictor().fit(dummy_data, random_state=8, **kw)
    assert np.array_equal(a.attack, b.attack)  # deterministic reduction order -> bitwise
    assert not np.array_equal(a.attack, c.attack)


def test_hip_sampler_vs_cpu_potential_sampler(hip_ctx, dummy_data):
    """Ladder L3: same driver, same key; the HIP potential and the float64 CPU oracle
    potential give posteriors that agree within Monte-Carlo error (trajectories themselves
    decorrelate after a few transitions: float32 tables vs float64)."""
    m = DixonColesMatchPredictor().fit(dummy_data, random_state=3, num_warmup=300, num_samples=600)
    fx = cases.fixtures("dummy")
    rc, draws, stats, summ = OC.nuts_dc(OC.CFixtures(O.MODEL_BASIC, fx), 300, 600, (0, 3))
    assert rc == 0
    sl = O.site_slices(O.MODEL_BASIC, 20)
    zh = m.mcmc_info_["unconstrained"]
    for name in ("home_advantage", "mean_defence", "std_attack", "std_defence", "corr_coef_raw"):
        a, b = zh[:, sl[name]].ravel(), draws[:, sl[name]].ravel()
        se = np.sqrt(a.var() / 100 + b.var() / 100)  # ESS >= ~100 each, conservatively
        assert abs(a.mean() - b.mean()) < 4 * se + 1e-3, name
    a, b = zh[:, sl["attack_decentered"]], draws[:, sl["attack_decentered"]]
    assert np.abs(a.mean(0) - b.mean(0)).max() < 0.45  # sd ~ 1 each, 20 comparisons
    # (mean accept prob depends on each chain's own adapted step size: only a sanity band)
    assert 0.6 < m.mcmc_info_["accept_prob"].mean() < 0.99 and 0.6 < stats[:, 1].mean() < 0.99


def test_first_transitions_match_cpu_potential(hip_ctx):
    """Ladder L1: with adaptation off, a fixed step size and a fixed start the HIP-driven
    and the CPU-oracle-driven chains take the same tree decisions for the first
    transitions: states agree to float32-table accuracy."""
    from bpl._ffi import MODEL_BASIC, default_nuts_cfg

    fx = cases.fixtures("dummy")
    hip_ctx.set_fixtures(MODEL_BASIC, fx.home_idx.astype(np.uint16), fx.away_idx.astype(np.uint16),
                         fx.home_goals.astype(np.uint8), fx.away_goals.astype(np.uint8), 20)
    z0 = np.random.RandomState(2).uniform(-0.2, 0.2, 45)
    cfg = default_nuts_cfg()
    cfg.num_warmup, cfg.num_samples, cfg.step_size = 0, 4, 0.02
    draws, st = hip_ctx.nuts_run(cfg, (0, 11), z0)
    rc, ref, stats, _ = OC.nuts_dc(OC.CFixtures(O.MODEL_BASIC, fx), 0, 4, (0, 11), z0=z0, step_size=0.02)
    assert rc == 0
    assert st["num_steps"].tolist() == stats[:, 2].astype(int).tolist()
    assert np.abs(draws - ref).max() < 1e-4
    assert np.abs(st["potential_energy"] - stats[:, 0]).max() < 1e-3


def test_device_tree_matches_host_tree(hip_ctx):
    """The device-resident tree builder (leaf bookkeeping in the kernel tail) and the host
    tree builder run the same algorithm on the same threefry streams.  With adaptation off
    (fixed step size) the two produce the same trees and the same draws to ~1e-14; with
    step-size adaptation on, 1e-16 differences in the energy sums are fed back through dual
    averaging and amplified by the (chaotic) trajectories, so only tree sizes and statistics
    are compared there."""
    from bpl._ffi import MODEL_BASIC, MODEL_EXTENDED, default_nuts_cfg

    for model, name in ((MODEL_BASIC, "dummy"), (MODEL_EXTENDED, "dummy_cov")):
        fx = cases.fixtures(name)
        cov = None if fx.covariates is None or model == MODEL_BASIC else O.standardise_covariates(fx.covariates)
        hip_ctx.set_fixtures(model, fx.home_idx.astype(np.uint16), fx.away_idx.astype(np.uint16),
                             fx.home_goals.astype(np.uint8), fx.away_goals.astype(np.uint8), 20,
                             covariates_std=cov)
        z0 = np.random.RandomState(2).uniform(-0.2, 0.2, hip_ctx.dim)
        cfg = default_nuts_cfg()
        cfg.num_warmup, cfg.num_samples, cfg.step_size = 0, 8, 0.02
        # three engines: whole chain on the device (persistent), device tree with host-side
        # adaptation, host tree
        engines = {"persistent": (1, 1), "device": (1, 0), "host": (0, 0)}

        def run(name, key, z_init=None):
            dn, pn = engines[name]
            hip_ctx.set_option("device_nuts", dn)
            hip_ctx.set_option("persistent_nuts", pn)
            try:
                return hip_ctx.nuts_run(cfg, key, z_init)
            finally:
                hip_ctx.set_option("device_nuts", 1)
                hip_ctx.set_option("persistent_nuts", 1)

        d0, s0 = run("host", (0, 11), z0)
        for name in ("persistent", "device"):
            d1, s1 = run(name, (0, 11), z0)
            assert s1["total_leapfrogs"] == s0["total_leapfrogs"] > 100
            assert s1["num_steps"].tolist() == s0["num_steps"].tolist()
            # rounding-level differences grow ~30x per transition (trajectories of 100+ steps)
            assert np.abs(d1[:4] - d0[:4]).max() < 1e-10 and np.abs(d1 - d0).max() < 1e-4
            assert np.abs(s1["potential_energy"][:4] - s0["potential_energy"][:4]).max() < 1e-8
            assert np.abs(s1["accept_prob"][:4] - s0["accept_prob"][:4]).max() < 1e-9
            assert np.allclose(s1["corr_coef"][:4], s0["corr_coef"][:4], atol=1e-12)
            assert np.allclose(s1["step_size"], 0.02)
        # adaptation on: healthy statistics in all three
        cfg.num_warmup, cfg.num_samples, cfg.step_size = 60, 40, 1.0
        for name in engines:
            d, st = run(name, (0, 5))
            # (60 warm-up iterations from init_to_uniform(radius=2) leave the step size barely
            # adapted: a stray post-warm-up divergence among 40 draws is not a defect)
            assert np.isfinite(d).all() and st["total_divergences"] <= 3, (name, st["total_divergences"])
            assert 0.55 < st["mean_accept_prob"] <= 1.0
            assert st["inverse_mass_matrix"].shape == (hip_ctx.dim,) and (st["inverse_mass_matrix"] > 0).all()


@pytest.mark.parametrize("persistent", [1, 0])
def test_lockstep_chains_match_single_chains(hip_ctx, persistent):
    """Several chains on one GPU (bplhip_nuts_run_chains, numpyro chain_method="vectorized"):
    `persistent` 1 = whole chains on the device (adaptation, draw collection and the next
    transition's start in the kernel tail; the host only enqueues evaluations), 0 = lock-step
    trees with host-side adaptation.  Both follow the same key sequences and the
    same algorithm as bplhip_nuts_run chain by chain.  The vectorised kernel groups its
    float32 partial sums differently (1e-10 relative in U), so with a fixed step size the
    trees coincide and the first draws agree closely; the difference then grows with the
    chaotic trajectories, as between the host and device tree builders."""
    from bpl._ffi import MODEL_BASIC, MODEL_EXTENDED, default_nuts_cfg

    hip_ctx.set_option("persistent_nuts", persistent)
    for model, name, nch in ((MODEL_BASIC, "dummy", 5), (MODEL_EXTENDED, "dummy_cov", 9)):
        fx = cases.fixtures(name)
        cov = None if fx.covariates is None or model == MODEL_BASIC else O.standardise_covariates(fx.covariates)
        hip_ctx.set_fixtures(model, fx.home_idx.astype(np.uint16), fx.away_idx.astype(np.uint16),
                             fx.home_goals.astype(np.uint8), fx.away_goals.astype(np.uint8), 20,
                             covariates_std=cov)
        D = hip_ctx.dim
        z0 = np.random.RandomState(2).uniform(-0.2, 0.2, (nch, D))
        keys = [(0, 11 + c) for c in range(nch)]
        cfg = default_nuts_cfg()
        cfg.num_warmup, cfg.num_samples, cfg.step_size = 0, 6, 0.02
        multi = hip_ctx.nuts_run_chains(cfg, keys, z0)
        for c in range(nch):
            d1, s1 = hip_ctx.nuts_run(cfg, keys[c], z0[c])
            dm, sm = multi[c]
            assert sm["num_steps"][:3].tolist() == s1["num_steps"][:3].tolist()
            assert np.abs(dm[:3] - d1[:3]).max() < 1e-5
            assert np.abs(sm["potential_energy"][:3] - s1["potential_energy"][:3]).max() < 1e-4
            assert sm["total_leapfrogs"] > 50
        # adaptation on, init_to_uniform: healthy statistics for every chain
        cfg.num_warmup, cfg.num_samples, cfg.step_size = 150, 40, 1.0
        multi = hip_ctx.nuts_run_chains(cfg, keys)
        for d, st in multi:
            assert np.isfinite(d).all() and st["total_divergences"] <= 2
            assert 0.55 < st["mean_accept_prob"] <= 1.0
        # distinct keys -> distinct chains
        assert np.abs(multi[0][0] - multi[1][0]).max() > 1e-3
        # one chain through the same entry point (persistent: the single-chain kernel)
        one = hip_ctx.nuts_run_chains(cfg, keys[:1])
        assert np.isfinite(one[0][0]).all() and 0.55 < one[0][1]["mean_accept_prob"] <= 1.0
    hip_ctx.set_option("persistent_nuts", 1)


@pytest.mark.parametrize("nch", [9, 33, 64])
def test_many_chains_on_a_fresh_context(nch):
    """More chains than `gridy_max_chains` (8; 32 until round 4) share the chain-vectorised kernel, whose launches a
    chunk captures into a hipGraph.  On a FRESH context nothing has sized that kernel's hand-off
    buffer yet (round 3: the capture failed with "operation not permitted when stream is
    capturing" and `fit(num_chains=64)` raised).  Same keys through bplhip_nuts_run, chain by chain:
    same trees, first draws to 1e-5 (reference: mcmc_kwargs={"num_chains": ...},
    bpl/dixon_coles.py:101-106)."""
    from bpl._ffi import MODEL_BASIC, HipContext, default_nuts_cfg

    fx = cases.fixtures("dummy")
    cfg = default_nuts_cfg()
    cfg.num_warmup, cfg.num_samples, cfg.step_size = 0, 4, 0.02
    cfg.adapt_step_size = cfg.adapt_mass_matrix = 0
    keys = [(0, 101 + c) for c in range(nch)]
    z0 = np.random.RandomState(5).uniform(-0.2, 0.2, (nch, 45))
    runs = {}
    for graph in (1, 0):   # the chunk as a replayed graph, and launch by launch: the same chains
        ctx = HipContext(0)
        try:
            ctx.set_option("chunk_graph", graph)
            ctx.set_fixtures(MODEL_BASIC, fx.home_idx.astype(np.uint16), fx.away_idx.astype(np.uint16),
                             fx.home_goals.astype(np.uint8), fx.away_goals.astype(np.uint8), 20)
            runs[graph] = ctx.nuts_run_chains(cfg, keys, z0)     # first call on this context
            if graph:
                singles = [ctx.nuts_run(cfg, keys[c], z0[c]) for c in range(nch)]
        finally:
            ctx.close()
    for c in range(nch):
        dm, sm = runs[1][c]
        d1, s1 = singles[c]
        assert sm["num_steps"][:3].tolist() == s1["num_steps"][:3].tolist(), c
        assert np.abs(dm[:3] - d1[:3]).max() < 1e-5, c
        assert np.abs(sm["potential_energy"][:3] - s1["potential_energy"][:3]).max() < 1e-4
        assert np.array_equal(dm, runs[0][c][0]), c   # graph replay == launch by launch, bit for bit
    assert np.abs(runs[1][0][0] - runs[1][nch - 1][0]).max() > 1e-3


def test_fit_forty_chains(dummy_data):
    """fit(mcmc_kwargs={"num_chains": 40}) on a fresh context (every fit makes its own): adapted
    chains through the chain-vectorised sampler; shapes, chain-major order, healthy statistics."""
    m = DixonColesMatchPredictor().fit(dummy_data, random_state=11, num_warmup=150, num_samples=25,
                                       mcmc_kwargs={"num_chains": 40})
    assert m.attack.shape == (1000, 20) and m.corr_coef.shape == (1000,)
    assert np.isfinite(m.attack).all() and np.isfinite(m.corr_coef).all()
    assert m.mcmc_info_["num_chains"] == 40 and m.mcmc_info_["divergences"] <= 10
    acc = m.mcmc_info_["accept_prob"].reshape(40, 25).mean(1)
    assert (acc > 0.5).all()
    # distinct chains, one posterior: chain means scatter by their Monte-Carlo error only
    cm = m.home_advantage.reshape(40, 25).mean(1)
    assert cm.std() > 1e-4 and cm.std() < 0.1


def test_fit_num_chains_lockstep_vs_sequential(dummy_data):
    """fit(mcmc_kwargs={"num_chains": 4}): the lock-step default and chain_method="sequential"
    sample the same posterior (shapes, chain-major order, means within Monte-Carlo error)."""
    from bpl import DixonColesMatchPredictor

    fits = {}
    for method in ("vectorized", "sequential"):
        m = DixonColesMatchPredictor().fit(dummy_data, random_state=3, num_warmup=200, num_samples=200,
                                           mcmc_kwargs={"num_chains": 4, "chain_method": method})
        assert m.attack.shape == (800, 20) and m.corr_coef.shape == (800,)
        assert m.mcmc_info_["divergences"] == 0
        fits[method] = m
    a, b = fits["vectorized"], fits["sequential"]
    sd = np.sqrt(a.attack.var(0) + b.attack.var(0))
    assert (np.abs(a.attack.mean(0) - b.attack.mean(0)) < 0.35 * sd).all()
    assert abs(a.home_advantage.mean() - b.home_advantage.mean()) < 0.05
    # chains are distinct and chain-major
    assert np.abs(a.attack[:200].mean(0) - a.attack[200:400].mean(0)).max() > 1e-6


@pytest.mark.parametrize("model_cls", MODELS)
def test_device_predict_matches_float64_restatement(fitted, model_cls):
    """Row f-2: the predict kernels (csrc/dc_predict.hip.h) against a float64 numpy restatement
    of the reference's predict_score_proba (tests/fake_ctx.py) on a fitted posterior.
    Tolerances: the pointwise kernel is float64 throughout (1e-12); the grid kernel computes the
    Poisson pmfs in float32 (v_exp_f32 of an argument of magnitude <= ~40: 2e-6 relative) on
    float32 posterior draws, accumulated on the matrix cores in blocks of 64 draws and in float64
    across blocks: 3e-6 relative + 1e-12."""
    from fake_ctx import FakePredictCtx

    models, dd = fitted
    model = models[model_cls]
    ref = FakePredictCtx()
    ref.predict_set_posterior(model.attack, model.defence, model.home_advantage, model.corr_coef)
    h, a = model._team_indices(dd["home_team"], dd["away_team"])
    # pointwise
    got = model.predict_score_proba(dd["home_team"], dd["away_team"], dd["home_goals"], dd["away_goals"])
    want = ref.predict_score_proba(h, a, dd["home_goals"], dd["away_goals"])
    assert np.abs(got - want).max() < 1e-12
    assert abs(model.predict_score_proba("0", "1", 1, 0)[0] - ref.predict_score_proba([0], [1], [1], [0])[0]) < 1e-12
    # the grid, default depth and a multi-tile one (max_goals = 20: 2 x 2 tiles of 16)
    for depth in (MAX_GOALS, 20, 3):
        grid, xs, ys = model.predict_score_grid_proba(dd["home_team"][:60], dd["away_team"][:60], max_goals=depth)
        want = ref.predict_score_grid(h[:60], a[:60], depth)
        assert grid.shape == (60, depth + 1, depth + 1) and xs.shape == (depth + 1, depth + 1)
        assert xs[3, 0] == 3 and ys[0, 3] == 3
        err = np.abs(grid - want)
        assert (err <= 3e-6 * want + 1e-12).all(), (depth, err.max())
        # float32 output (bplhip_predict_score_grid_f32; the reference's own dtype): the float64 accumulators
        # rounded once, at the store
        g32 = model._device().predict_score_grid(h[:60], a[:60], depth, dtype=np.float32)
        assert g32.dtype == np.float32 and np.array_equal(g32, grid.astype(np.float32))
    # the reductions of the grid
    out = model.predict_outcome_proba(dd["home_team"][:40], dd["away_team"][:40])
    g = ref.predict_score_grid(h[:40], a[:40], MAX_GOALS)
    xs, ys = np.meshgrid(np.arange(MAX_GOALS + 1), np.arange(MAX_GOALS + 1), indexing="ij")
    for key, mask in (("home_win", xs > ys), ("draw", xs == ys), ("away_win", xs < ys)):
        assert np.abs(out[key] - g[:, mask].sum(axis=1)).max() < 3e-6
    n = np.arange(MAX_GOALS + 1)
    g01 = ref.predict_score_grid([0], [1], MAX_GOALS)[0]
    g10 = ref.predict_score_grid([1], [0], MAX_GOALS)[0]
    assert np.abs(model.predict_score_n_proba(n, "0", "1") - g01.sum(axis=1)).max() < 3e-6
    assert np.abs(model.predict_score_n_proba(n, "0", "1", home=False) - g10.sum(axis=0)).max() < 3e-6
    assert np.abs(model.predict_concede_n_proba(n, "0", "1") - g01.sum(axis=0)).max() < 3e-6
    assert np.abs(model.predict_concede_n_proba(n, "0", "1", home=False) - g10.sum(axis=1)).max() < 3e-6
    # n beyond max_goals: the other side is still summed over 0..max_goals only
    deep = ref.predict_score_grid([0], [1], 18)[0]
    assert abs(model.predict_score_n_proba(18, "0", "1")[0] - deep[18, :MAX_GOALS + 1].sum()) < 1e-9
