"""Host logic of the neutral-venue / World-Cup predictors (no GPU): the predict_* family on a
hand-made posterior, mirroring the property asserts of the reference's
tests/test_neutral_dixon_coles.py and tests/test_neutral_dixon_coles_WC.py."""
import numpy as np
import pytest

from bpl import NeutralDixonColesMatchPredictor, NeutralDixonColesMatchPredictorWC
from bpl.neutral_dixon_coles import latent_sites, make_weights
from fake_ctx import FakePredictCtx

MAX_GOALS = 15
TOL = 1e-2


def _posterior(cls, S=200, T=6, C=0, seed=0):
    rs = np.random.RandomState(seed)
    m = cls()
    m.teams = np.array([str(i) for i in range(T)])
    m._teams_dict = {t: i for i, t in enumerate(m.teams)}
    m.attack = rs.normal(0.0, 0.2, (S, T))
    m.defence = rs.normal(0.0, 0.2, (S, T))
    m.home_attack = rs.normal(0.15, 0.05, (S, T))
    m.away_attack = rs.normal(-0.1, 0.05, (S, T))
    m.home_defence = rs.normal(0.1, 0.05, (S, T))
    m.away_defence = rs.normal(-0.1, 0.05, (S, T))
    m.corr_coef = rs.uniform(-0.05, 0.03, S)
    m.rho = rs.uniform(-0.2, 0.6, S)
    m.mean_defence = rs.normal(0, 0.1, S)
    m.std_attack = np.abs(rs.normal(0.3, 0.05, S))
    m.std_defence = np.abs(rs.normal(0.3, 0.05, S))
    for nm, mu in (("home_attack", 0.1), ("away_attack", -0.1), ("home_defence", 0.1), ("away_defence", -0.1)):
        setattr(m, "mean_" + nm, rs.normal(mu, 0.02, S))
        setattr(m, "std_" + nm, np.abs(rs.normal(0.1, 0.02, S)))
    if C:
        m.conferences = np.array([str(i) for i in range(C)])
        m._conferences_dict = {c: i for i, c in enumerate(m.conferences)}
        m.confederation_strength = rs.normal(0, 0.2, (S, C))
    m._predict_ctx = FakePredictCtx()  # (no GPU here: the numpy restatement stands in for the kernels)
    return m


def test_neutral_predict_family():
    m = _posterior(NeutralDixonColesMatchPredictor)
    home, away = ["0", "1", "2"], ["3", "4", "5"]
    p = m.predict_score_proba(home, away, [1, 0, 2], [0, 0, 1], [0, 1, 0])
    assert p.shape == (3,) and np.all((p >= 0) & (p <= 1))
    assert 0 <= m.predict_score_proba("0", "1", 1, 0, 0)[0] <= 1
    out = m.predict_outcome_proba(home, away, [0, 1, 0])
    assert np.allclose(out["home_win"] + out["draw"] + out["away_win"], 1.0, atol=TOL)
    ko = m.predict_outcome_proba(home, away, [0, 1, 0], knockout=True)
    assert set(ko) == {"home_win", "away_win"} and np.allclose(ko["home_win"] + ko["away_win"], 1.0)
    grid, hg, ag = m.predict_score_grid_proba(home, away, [0, 0, 1], max_goals=7)
    assert grid.shape == (3, 8, 8) and hg.shape == (8, 8)
    n = np.arange(MAX_GOALS + 1)
    ph, pa = m.predict_score_n_proba(n, "0", "1"), m.predict_score_n_proba(n, "0", "1", home=False)
    assert sum(ph) == pytest.approx(1.0, abs=TOL) and sum(ph * n) > sum(pa * n)  # score more at home
    assert np.allclose(ph, m.predict_concede_n_proba(n, "1", "0", home=False), atol=1e-12)
    # at a neutral venue the home/away offsets vanish: swapping the sides mirrors the rates
    lh, la = m._calculate_expected_goals(["0"], ["1"], [1])
    lh2, la2 = m._calculate_expected_goals(["1"], ["0"], [1])
    assert np.allclose(lh, la2) and np.allclose(la, lh2)
    s = m.sample_score(home, away, [0, 1, 0], num_samples=30, random_state=7)
    assert s["home_score"].shape == (3, 30) and s["home_score"].max() <= MAX_GOALS
    s2 = m.sample_score(home, away, [0, 1, 0], num_samples=30, random_state=7)
    assert np.array_equal(s["home_score"], s2["home_score"])  # threefry: reproducible
    w = m.sample_outcome(home, away, [0, 1, 0], num_samples=30, random_state=7)
    assert w.shape == (3, 30) and set(np.unique(w[0])) <= {"0", "3", "Draw"}
    wk = m.sample_outcome(home, away, [0, 1, 0], knockout=True, num_samples=30, random_state=7)
    assert "Draw" not in set(np.unique(wk))
    with pytest.raises(ValueError):
        m.add_new_team("0")
    np.random.seed(1)
    m.add_new_team("new")
    assert m.attack.shape[1] == 7 and m.home_defence.shape[1] == 7
    assert 0 <= m.predict_score_proba("new", "0", 0, 0, 1)[0] <= 1


def test_world_cup_predict_family():
    m = _posterior(NeutralDixonColesMatchPredictorWC, C=3)
    home, away, hc, ac = ["0", "1"], ["2", "3"], ["0", "1"], ["2", "0"]
    p = m.predict_score_proba(home, away, hc, ac, [1, 0], [0, 0], [0, 1])
    assert p.shape == (2,) and np.all((p >= 0) & (p <= 1))
    out = m.predict_outcome_proba(home, away, hc, ac, [0, 1])
    assert np.allclose(out["home_win"] + out["draw"] + out["away_win"], 1.0, atol=5e-2)
    # a stronger confederation raises the scoring rate
    m.confederation_strength[:, 0] = 1.0
    m.confederation_strength[:, 2] = -1.0
    lh_strong, _ = m._calculate_expected_goals(["0"], ["2"], ["0"], ["2"], [1])
    lh_weak, _ = m._calculate_expected_goals(["0"], ["2"], ["2"], ["0"], [1])
    assert np.all(lh_strong > lh_weak)
    n = np.arange(MAX_GOALS + 1)
    ph = m.predict_score_n_proba(n, "0", "1", "0", "1")
    assert sum(ph) == pytest.approx(1.0, abs=5e-2)
    assert np.allclose(ph, m.predict_concede_n_proba(n, "1", "0", "1", "0", home=False), atol=1e-12)
    s = m.sample_score(home, away, hc, ac, [0, 1], num_samples=10, random_state=3)
    assert s["away_score"].shape == (2, 10)
    w = m.sample_outcome(home, away, hc, ac, [1, 1], knockout=True, num_samples=10, random_state=3)
    assert set(np.unique(w)) <= {"0", "1", "2", "3"}


def test_layout_and_weights():
    assert sum(n for _, n in latent_sites(20, 0)) == 6 * 20 + 13
    assert sum(n for _, n in latent_sites(20, 5, 4)) == 6 * 20 + 10 + 4 + 13
    names = [n for n, _ in latent_sites(5, 2, 3)]
    assert names == sorted(names) and "confederation_strength_decentered" in names
    w = make_weights(3, [0.0, 1.0, 2.0], 0.5, [1.0, 2.0, 0.5], True)
    base = np.exp(-0.5 * np.arange(3.0))
    assert np.allclose(w, 3 * base / base.sum() * np.array([1.0, 2.0, 0.5]))
    with pytest.raises(TypeError):
        make_weights(3, None, None, None, False)
