"""CPU tests of the dynamic-model oracle (test infrastructure): hand-derived adjoint vs
torch autograd of a literal transcription of bpl/dynamic_dixon_coles.py:63-247, for the
intended random walk and for the reference-as-written semantics."""
import numpy as np
import pytest

import dc_dynamic_oracle as DO


@pytest.mark.parametrize("random_walk", [True, False])
@pytest.mark.parametrize("k", [0, 3])
def test_adjoint_vs_autograd(random_walk, k):
    fx = DO.small_recipe(k=k)
    D = DO.latent_dim(fx.n_gameweeks, fx.n_teams, fx.k)
    sl = DO.site_slices(fx.n_gameweeks, fx.n_teams, fx.k)
    for seed in (1, 2):
        z = np.random.RandomState(seed).uniform(-0.5, 0.5, D)
        if seed == 2:
            z[sl["mean_home_attack"]] = 1.2  # forces M > 1 (UB = 1/M branch)
        U, g, aux = DO.potential_and_grad(fx, z, random_walk)
        Ut, gt, corr = DO.torch_potential_and_grad(fx, z, random_walk)
        assert U == pytest.approx(Ut, rel=1e-12)
        assert np.abs(g - gt).max() <= 1e-11 * np.abs(gt).max()
        assert aux["rho"] == pytest.approx(corr, abs=1e-13)
        if seed == 2:
            assert aux["UB"] < 1.0


def test_layout_and_as_written_semantics():
    G, T, K = 5, 7, 2
    assert DO.latent_dim(G, T, K) == 7 * G * T + 10 * G + 2 + 2 * K
    names = [n for n, _ in DO.site_list(G, T, K)]
    assert names == sorted(names)
    assert DO.latent_dim(50, 100) == 35502  # BASELINE config 4
    # as written, the walk sites do not reach the likelihood: gradient = prior only
    fx = DO.small_recipe()
    D = DO.latent_dim(fx.n_gameweeks, fx.n_teams)
    sl = DO.site_slices(fx.n_gameweeks, fx.n_teams)
    z = np.random.RandomState(3).uniform(-0.5, 0.5, D)
    _, g, _ = DO.potential_and_grad(fx, z, random_walk=False)
    assert g[sl["standardised_attack"]] != pytest.approx(0)  # prior still acts
    _, g2, _ = DO.potential_and_grad(fx, z + 0.0, random_walk=True)
    assert not np.allclose(g[sl["standardised_attack"]], g2[sl["standardised_attack"]])
    assert g[sl["std_attack"]] == pytest.approx(-(1.0 - np.exp(2 * z[sl["std_attack"]])))


def test_config4_recipe_shapes():
    fx = DO.config4_recipe()
    assert fx.n == 2500 and fx.n_teams == 100 and fx.n_gameweeks == 50
    for w in range(50):  # one round of 50 disjoint pairings per gameweek
        m = fx.gameweek == w
        assert m.sum() == 50
        assert len(set(fx.home_idx[m]) | set(fx.away_idx[m])) == 100


def test_golden_vectors():
    """tests/golden/m2_small_cov.npz (oracle/make_golden.py) is reproduced by the oracle as it stands."""
    import os

    d = np.load(os.path.join(os.path.dirname(__file__), "golden", "m2_small_cov.npz"))
    fx = DO.DynFixtures(d["home_idx"], d["away_idx"], d["home_goals"], d["away_goals"], d["gameweek"],
                        d["neutral"], int(d["n_teams"]), int(d["n_gameweeks"]), covariates=d["covariates"])
    for rw in (1, 0):
        for i in range(d["z"].shape[0]):
            U, g, aux = DO.potential_and_grad(fx, d["z"][i], bool(rw))
            assert abs(U - d[f"U_rw{rw}"][i]) <= 1e-12 * abs(U)
            assert np.abs(g - d[f"grad_rw{rw}"][i]).max() <= 1e-11 * np.abs(g).max()
            assert abs(aux["rho"] - d[f"rho_rw{rw}"][i]) <= 1e-13
