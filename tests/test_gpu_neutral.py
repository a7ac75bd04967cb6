"""GPU: the neutral-venue model (row f-4) through the C-ABI vs its float64 oracle, and the
reference's own property tests (tests/test_neutral_dixon_coles.py) on a fitted model.

Tolerances (float64 path end to end; the only float32 data are the weights, which the oracle
receives rounded to float32 as well):  |dU| <= 1e-11 |U|,  |dgrad|_inf <= 1e-10 |grad|_inf.
"""
import numpy as np
import pytest

import dc_neutral_oracle as NO

pytestmark = pytest.mark.gpu
TOL = 1e-02
MAX_GOALS = 15


def _bind(ctx, fx):
    cov = None if fx.covariates is None else NO.standardise_covariates(fx.covariates)
    ctx.set_fixtures_neutral(fx.home_idx, fx.away_idx, fx.home_goals, fx.away_goals, fx.neutral,
                             fx.n_teams, weights=fx.weights.astype(np.float32), covariates_std=cov,
                             home_conf=fx.home_conf, away_conf=fx.away_conf, n_conf=fx.n_conf)
    assert ctx.dim == NO.latent_dim(fx.n_teams, fx.k, fx.n_conf)


def _cases():
    dd = NO.neutral_dummy_recipe()
    yield "dummy", NO.fixtures_from_data(dd)
    yield "dummy_eps", NO.fixtures_from_data(dd, epsilon=0.3, rescale_weights=True)
    yield "dummy_cov", NO.fixtures_from_data(dd, epsilon=0.1,
                                             covariates=np.random.RandomState(0).normal(size=(20, 3)))
    yield "synthetic_6e3", NO.synthetic_neutral(6_000, 150, k=3)   # several incidence rounds per wave
    yield "synthetic_3e4", NO.synthetic_neutral(30_000, 30, k=2)
    yield "synthetic_1e6", NO.synthetic_neutral(1_000_000, 20)
    # World-Cup variant: confederation strengths
    yield "wc_dummy", NO.fixtures_from_data_wc(dd, epsilon=0.2, rescale_weights=True)
    yield "wc_synthetic_3e4", NO.synthetic_neutral(30_000, 30, k=2, n_conf=5)
    yield "wc_synthetic_1e6", NO.synthetic_neutral(1_000_000, 40, n_conf=6)


@pytest.mark.parametrize("name,fx", list(_cases()))
def test_logp_grad_matches_oracle(hip_ctx, name, fx):
    import torch

    fx.weights = fx.weights.astype(np.float32).astype(np.float64)  # what the device holds
    _bind(hip_ctx, fx)
    D = hip_ctx.dim
    # both evaluation paths: a single launch (neu_fused: one workgroup, small leagues; neu_big: a slice of
    # the fixtures per workgroup, any N) and the four-launch one -- and the single launch again after it
    # (it must find the scratch the four launches left behind cleared)
    for fused in (1, 0, 1):
        hip_ctx.set_option("fused_small", fused)
        for seed, scale in ((1, 0.2), (2, 0.5), (3, 1.0)):
            z = np.random.RandomState(seed).uniform(-scale, scale, D)
            Uo, go, auxo = NO.potential_and_grad(fx, z)
            U, g, aux = hip_ctx.logp_grad(torch.tensor(z, dtype=torch.float64, device=hip_ctx.device))
            U, g, aux = float(U.cpu()[0]), g.cpu().numpy(), aux.cpu().numpy()[0]
            print(f"{name:14s} fused={fused} N={fx.n:8d} U={Uo:.6f} dU={U - Uo:+.2e} "
                  f"dg={np.abs(g - go).max():.2e} |g|={np.abs(go).max():.2e}")
            assert abs(U - Uo) <= 1e-11 * abs(Uo)
            assert np.abs(g - go).max() <= 1e-10 * np.abs(go).max()
            assert abs(aux[0] - auxo["rho"]) <= 1e-12
            assert abs(aux[1] - auxo["LB"]) <= 1e-12 and abs(aux[2] - auxo["UB"]) <= 1e-12
    hip_ctx.set_option("fused_small", 1)


@pytest.mark.parametrize("n_conf", [0, 4])
def test_batched_chains_match_single(hip_ctx, n_conf):
    """Several chains in one call (one workgroup per chain in the single-launch kernel) give the
    results of the chains evaluated one by one: bit for bit for the plain model (its sums have a
    fixed order), to rounding with confederations (their adjoint is added with LDS atomics)."""
    import torch

    fx = NO.synthetic_neutral(5_000, 24, k=2, n_conf=n_conf)
    fx.weights = fx.weights.astype(np.float32).astype(np.float64)
    _bind(hip_ctx, fx)
    z = torch.tensor(np.random.RandomState(5).uniform(-0.4, 0.4, (5, hip_ctx.dim)), dtype=torch.float64,
                     device=hip_ctx.device)
    Ub, gb, auxb = hip_ctx.logp_grad(z)
    for c in range(z.shape[0]):
        U, g, aux = hip_ctx.logp_grad(z[c].contiguous())
        if n_conf == 0:
            assert torch.equal(U[0], Ub[c]) and torch.equal(g.reshape(-1), gb[c])
        else:
            assert abs(float(U[0] - Ub[c])) <= 1e-13 * abs(float(U[0]))
            assert float((g.reshape(-1) - gb[c]).abs().max()) <= 1e-12 * float(g.abs().max())
        assert torch.equal(aux[0], auxb[c])
        Uo, go, _ = NO.potential_and_grad(fx, z[c].cpu().numpy())
        assert abs(float(U[0]) - Uo) <= 1e-11 * abs(Uo)
        assert np.abs(g.cpu().numpy().reshape(-1) - go).max() <= 1e-10 * np.abs(go).max()


@pytest.fixture(scope="module")
def model():
    from bpl import NeutralDixonColesMatchPredictor

    dd = NO.neutral_dummy_recipe()
    m = NeutralDixonColesMatchPredictor().fit(dd, num_warmup=300, num_samples=300)
    return m, dd


def test_fit(model):
    m, _ = model
    for nm in ("attack", "defence", "home_attack", "home_defence", "away_attack", "away_defence",
               "teams", "corr_coef"):
        assert getattr(m, nm) is not None
    assert m.attack.shape == (300, 20) and m.corr_coef.shape == (300,)
    assert m.mcmc_info_["divergences"] <= 3
    # the data were drawn with a home mean of 2.1 against 1.7 away, 1.9 / 1.9 on neutral ground
    assert m.mean_home_attack.mean() - m.mean_away_attack.mean() > 0.0
    # deterministic corr_coef site obeys its bounds at every draw
    h = np.array([m._teams_dict[t] for t in _["home_team"]])
    a = np.array([m._teams_dict[t] for t in _["away_team"]])
    lh, la = m._calculate_expected_goals(h, a, _["neutral_venue"])
    UB = np.minimum(1.0 / (lh * la).max(axis=1), 1.0)
    LB = -1.0 / np.maximum(lh.max(axis=1), la.max(axis=1))
    assert np.all(m.corr_coef <= UB + 1e-9) and np.all(m.corr_coef >= LB - 1e-9)


def test_predict_score_proba(model):
    m, dd = model
    probs = m.predict_score_proba(dd["home_team"], dd["away_team"], dd["home_goals"],
                                  dd["away_goals"], dd["neutral_venue"])
    assert np.all((probs >= 0) & (probs <= 1))
    assert 0 <= m.predict_score_proba("0", "1", 1, 0, 0)[0] <= 1


def test_predict_outcome_proba(model):
    m, dd = model
    probs = m.predict_outcome_proba(dd["home_team"], dd["away_team"], dd["neutral_venue"])
    assert np.allclose(probs["home_win"] + probs["away_win"] + probs["draw"], 1.0, atol=TOL)
    single = m.predict_outcome_proba("0", "1", 0)
    assert single["home_win"] + single["away_win"] + single["draw"] == pytest.approx(1.0, abs=TOL)
    ko = m.predict_outcome_proba("0", "1", 1, knockout=True)
    assert set(ko) == {"home_win", "away_win"} and ko["home_win"] + ko["away_win"] == pytest.approx(1.0)


def test_predict_n_proba(model):
    m, _ = model
    n = np.arange(MAX_GOALS + 1)
    ph = m.predict_score_n_proba(n, "0", "1")
    pa = m.predict_score_n_proba(n, "0", "1", home=False)
    assert len(ph) == len(n) and np.all((ph >= 0) & (ph <= 1)) and sum(ph) == pytest.approx(1.0, abs=TOL)
    assert sum(pa) == pytest.approx(1.0, abs=TOL)
    assert sum(ph * n) > sum(pa * n)  # score more at home
    ch = m.predict_concede_n_proba(n, "0", "1")
    ca = m.predict_concede_n_proba(n, "0", "1", home=False)
    assert sum(ch) == pytest.approx(1.0, abs=TOL) and sum(ch * n) < sum(ca * n)  # concede more away
    # scoring at home == the opponent conceding away
    assert np.allclose(ph, m.predict_concede_n_proba(n, "1", "0", home=False), atol=1e-12)
    assert len(m.predict_score_n_proba(1, "0", "1")) == 1


def test_sampling_and_new_team(model):
    m, _ = model
    s = m.sample_score(["0", "2"], ["1", "3"], [0, 1], num_samples=50, random_state=1)
    assert s["home_score"].shape == (2, 50) and s["away_score"].shape == (2, 50)
    w = m.sample_outcome(["0", "2"], ["1", "3"], [0, 1], num_samples=50, random_state=1)
    assert w.shape == (2, 50) and set(np.unique(w[0])) <= {"0", "1", "Draw"}
    wk = m.sample_outcome(["0"], ["1"], [1], knockout=True, num_samples=50, random_state=1)
    assert set(np.unique(wk)) <= {"0", "1"}
    with pytest.raises(ValueError):
        m.add_new_team("0")
    np.random.seed(0)
    m.add_new_team("new")
    assert m.attack.shape[1] == 21 and "new" in m.teams
    assert 0 <= m.predict_score_proba("new", "1", 1, 0, 0)[0] <= 1


def test_fit_two_chains():
    from bpl import NeutralDixonColesMatchPredictor

    dd = NO.neutral_dummy_recipe()
    m = NeutralDixonColesMatchPredictor().fit(dd, num_warmup=150, num_samples=100,
                                              mcmc_kwargs={"num_chains": 2})
    assert m.attack.shape == (200, 20) and np.isfinite(m.attack).all()
    assert np.abs(m.attack[:100].mean(0) - m.attack[100:].mean(0)).max() > 1e-6  # distinct chains
    assert np.abs(m.attack[:100].mean(0) - m.attack[100:].mean(0)).max() < 0.6   # same posterior


def test_errors():
    from bpl import NeutralDixonColesMatchPredictor

    dd = NO.neutral_dummy_recipe()
    bad = {k: v for k, v in dd.items() if k != "time_diff"}
    with pytest.raises(ValueError):
        NeutralDixonColesMatchPredictor().fit(bad, epsilon=1.0, num_warmup=5, num_samples=5)
    bad = {k: v for k, v in dd.items() if k != "game_weights"}
    with pytest.raises(TypeError):
        NeutralDixonColesMatchPredictor().fit(bad, num_warmup=5, num_samples=5)


# ---- World-Cup variant: the reference's tests/test_neutral_dixon_coles_WC.py on a fitted model
@pytest.fixture(scope="module")
def model_wc():
    from bpl import NeutralDixonColesMatchPredictorWC

    dd = NO.neutral_dummy_recipe()
    return NeutralDixonColesMatchPredictorWC().fit(dd, num_warmup=300, num_samples=300), dd


def test_wc_fit_and_predict(model_wc):
    m, dd = model_wc
    for nm in ("confederation_strength", "attack", "defence", "home_attack", "home_defence",
               "away_attack", "away_defence", "teams", "conferences", "corr_coef"):
        assert getattr(m, nm) is not None
    assert m.confederation_strength.shape == (300, 5) and list(m.conferences) == ["0", "1", "2", "3", "4"]
    probs = m.predict_score_proba(dd["home_team"], dd["away_team"], dd["home_conf"], dd["away_conf"],
                                  dd["home_goals"], dd["away_goals"], dd["neutral_venue"])
    assert np.all((probs >= 0) & (probs <= 1))
    assert 0 <= m.predict_score_proba("0", "1", "0", "1", 1, 0, 0)[0] <= 1
    out = m.predict_outcome_proba(dd["home_team"], dd["away_team"], dd["home_conf"], dd["away_conf"],
                                  dd["neutral_venue"])
    assert np.allclose(out["home_win"] + out["away_win"] + out["draw"], 1.0, atol=5e-2)
    n = np.arange(MAX_GOALS + 1)
    ph = m.predict_score_n_proba(n, "0", "1", "0", "0")
    pa = m.predict_score_n_proba(n, "0", "1", "0", "0", home=False)
    assert sum(ph) == pytest.approx(1.0, abs=5e-2) and sum(ph * n) > sum(pa * n)
    assert np.allclose(ph, m.predict_concede_n_proba(n, "1", "0", "0", "0", home=False), atol=1e-12)
    s = m.sample_score(["0"], ["5"], ["0"], ["1"], [1], num_samples=20, random_state=3)
    assert s["home_score"].shape == (1, 20)
    w = m.sample_outcome(["0"], ["5"], ["0"], ["1"], [1], knockout=True, num_samples=20, random_state=3)
    assert set(np.unique(w)) <= {"0", "5"}


# ---- device predict path of the venue-aware family (csrc/dc_predict.hip.h, VENUE = 1)
def _venue_check(m, h, a, nv, conf, x, y):
    """Device kernels against tests/fake_ctx.FakePredictCtx (float64 numpy restatement of the
    reference's predict_score_proba).  Tolerances as for the league models (tests/test_gpu_fit.py):
    pointwise kernel float64 throughout -> 1e-12; grid kernel float32 pmfs on float32 draws, float64
    across 64-draw blocks -> 3e-6 relative + 1e-12."""
    from fake_ctx import FakePredictCtx

    ref = FakePredictCtx()
    ref.predict_set_posterior_venue(m.attack, m.defence, m.home_attack, m.away_attack, m.home_defence,
                                    m.away_defence, m.corr_coef, confederation_strength=m.confederation_strength)
    got = m._score_proba(h, a, x, y, nv, conf)
    want = ref.predict_score_proba(h, a, x, y, nv, conf)
    assert np.abs(got - want).max() < 1e-12
    for depth in (MAX_GOALS, 20, 3):
        grid = m._grid_probs(h[:40], a[:40], nv[:40], None if conf is None else (conf[0][:40], conf[1][:40]), depth)
        want = ref.predict_score_grid(h[:40], a[:40], depth, nv[:40],
                                      None if conf is None else (conf[0][:40], conf[1][:40]))
        err = np.abs(grid - want)
        assert (err <= 3e-6 * want + 1e-12).all(), (depth, err.max(), (err / (want + 1e-300)).max())
        g32 = m._device().predict_score_grid(h[:40], a[:40], depth, neutral=nv[:40],
                                             conf=None if conf is None else (conf[0][:40], conf[1][:40]), dtype=np.float32)
        assert g32.dtype == np.float32 and np.array_equal(g32, np.asarray(grid).astype(np.float32))   # (bplhip_predict_score_grid_venue_f32)
    return ref


def test_device_predict_matches_float64_restatement(model):
    m, dd = model
    h, a, nv = m._parse_fixture_args(dd["home_team"], dd["away_team"], dd["neutral_venue"])
    assert nv.min() == 0 and nv.max() == 1
    ref = _venue_check(m, h, a, nv, None, np.asarray(dd["home_goals"]), np.asarray(dd["away_goals"]))
    # the reductions of the grid, through the public API
    out = m.predict_outcome_proba(dd["home_team"][:30], dd["away_team"][:30], dd["neutral_venue"][:30])
    g = ref.predict_score_grid(h[:30], a[:30], MAX_GOALS, nv[:30])
    xs, ys = np.meshgrid(np.arange(MAX_GOALS + 1), np.arange(MAX_GOALS + 1), indexing="ij")
    for key, mask in (("home_win", xs > ys), ("draw", xs == ys), ("away_win", xs < ys)):
        assert np.abs(out[key] - g[:, mask].sum(axis=1)).max() < 3e-6
    n = np.arange(MAX_GOALS + 1)
    t0, t1 = m._teams_dict["0"], m._teams_dict["1"]
    for venue in (0, 1):
        g01 = ref.predict_score_grid([t0], [t1], MAX_GOALS, [venue])[0]
        g10 = ref.predict_score_grid([t1], [t0], MAX_GOALS, [venue])[0]
        assert np.abs(m.predict_score_n_proba(n, "0", "1", neutral_venue=venue) - g01.sum(axis=1)).max() < 3e-6
        assert np.abs(m.predict_score_n_proba(n, "0", "1", home=False, neutral_venue=venue) - g10.sum(axis=0)).max() < 3e-6
        assert np.abs(m.predict_concede_n_proba(n, "0", "1", neutral_venue=venue) - g01.sum(axis=0)).max() < 3e-6
    # a fitted model pickles after predicting (the device context is not part of its state)
    import pickle

    clone = pickle.loads(pickle.dumps(m))
    assert clone._predict_ctx is None
    assert np.array_equal(clone.predict_score_proba("0", "1", 1, 0, 0), m.predict_score_proba("0", "1", 1, 0, 0))


def test_wc_device_predict_matches_float64_restatement(model_wc):
    m, dd = model_wc
    h, a, hc, ac, nv = m._parse_fixture_args(dd["home_team"], dd["away_team"], dd["home_conf"], dd["away_conf"],
                                             dd["neutral_venue"])
    assert len(set(hc.tolist())) > 1
    _venue_check(m, h, a, nv, (hc, ac), np.asarray(dd["home_goals"]), np.asarray(dd["away_goals"]))
    # the plain and the venue entry points cannot be mixed on one context
    from bpl._ffi import BPLHIP_ESTATE, BplHipError

    with pytest.raises(BplHipError) as e:
        m._device().predict_score_grid(h[:2], a[:2], 5)
    assert e.value.code == BPLHIP_ESTATE
    with pytest.raises(BplHipError):
        m._device().predict_score_grid(h[:2], a[:2], 5, neutral=nv[:2])  # confederations missing
