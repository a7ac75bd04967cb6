"""The neutral-venue model's oracle (row f-4): numpy hand-derived adjoint vs torch autograd of a
literal transcription vs central finite differences."""
import numpy as np
import pytest

import dc_neutral_oracle as NO


def _cases():
    dd = NO.neutral_dummy_recipe()
    yield "dummy", NO.fixtures_from_data(dd)
    yield "dummy_eps", NO.fixtures_from_data(dd, epsilon=0.3, rescale_weights=True)
    cov = np.random.RandomState(0).normal(size=(20, 3))
    yield "dummy_cov", NO.fixtures_from_data(dd, epsilon=0.1, covariates=cov)
    yield "synthetic", NO.synthetic_neutral(3000, 7, k=2)
    yield "wc_dummy", NO.fixtures_from_data_wc(dd, epsilon=0.2, rescale_weights=True)
    yield "wc_synthetic", NO.synthetic_neutral(3000, 9, k=1, n_conf=3)


@pytest.mark.parametrize("name,fx", list(_cases()))
def test_numpy_matches_torch_autograd(name, fx):
    D = NO.latent_dim(fx.n_teams, fx.k, fx.n_conf)
    for seed, scale in ((0, 0.1), (1, 0.3), (2, 0.6), (3, 1.0)):  # (z = 0 ties every rate: the max adjoint is a convention there)
        z = np.random.RandomState(seed).uniform(-scale, scale, D)
        U, g, aux = NO.potential_and_grad(fx, z)
        Ut, gt, auxt = NO.torch_potential_and_grad(fx, z)
        assert abs(U - Ut) <= 1e-11 * abs(Ut)
        assert np.abs(g - gt).max() <= 1e-10 * np.abs(gt).max()
        assert abs(aux["rho"] - auxt["rho"]) < 1e-13
        assert abs(aux["LB"] - auxt["LB"]) < 1e-13 and abs(aux["UB"] - auxt["UB"]) < 1e-13


def test_finite_differences():
    fx = NO.synthetic_neutral(500, 6, k=1, n_conf=2)
    D = NO.latent_dim(6, 1, 2)
    z = np.random.RandomState(5).uniform(-0.4, 0.4, D)
    U, g, _ = NO.potential_and_grad(fx, z)
    hstep = 1e-6
    for i in range(D):
        zp, zm = z.copy(), z.copy()
        zp[i] += hstep
        zm[i] -= hstep
        fd = (NO.potential_and_grad(fx, zp)[0] - NO.potential_and_grad(fx, zm)[0]) / (2 * hstep)
        assert abs(fd - g[i]) <= 2e-5 * max(1.0, abs(g[i])), (i, fd, g[i])


def test_layout_and_weights():
    assert NO.latent_dim(20) == 6 * 20 + 13 and NO.latent_dim(20, 5) == 6 * 20 + 10 + 13
    names = [n for n, _ in NO.site_list(4, 2)]
    assert names == sorted(names)
    w = NO.make_weights(4, [0.0, 1.0, 2.0, 3.0], 0.5, [1.0, 2.0, 1.0, 0.5], rescale_weights=True)
    base = np.exp(-0.5 * np.arange(4.0))
    assert np.allclose(w, 4 * base / base.sum() * np.array([1.0, 2.0, 1.0, 0.5]))


@pytest.mark.parametrize("name", ["dummy", "dummy_eps_cov"])
def test_golden_vectors(name):
    """tests/golden/m3_*.npz (oracle/make_golden.py): the committed vectors are reproduced by
    both restatements -- a regression pin for the oracle itself and the data of the GPU test."""
    import os

    g = np.load(os.path.join(os.path.dirname(__file__), "golden", f"m3_{name}.npz"))
    cov = g["covariates"] if g["covariates"].size else None
    fx = NO.NeutralFixtures(g["home_idx"], g["away_idx"], g["home_goals"], g["away_goals"],
                            g["neutral"], g["weights"], int(g["n_teams"]), covariates=cov)
    for i in range(g["z"].shape[0]):
        U, gr, aux = NO.potential_and_grad(fx, g["z"][i])
        assert abs(U - g["U"][i]) <= 1e-12 * abs(U) and np.abs(gr - g["grad"][i]).max() <= 1e-9
        Ut, gt, _ = NO.torch_potential_and_grad(fx, g["z"][i])
        assert abs(Ut - g["U"][i]) <= 1e-11 * abs(U)
        assert np.abs(gt - g["grad"][i]).max() <= 1e-10 * np.abs(gt).max()
        assert abs(aux["rho"] - g["rho"][i]) < 1e-13
