"""CPU tests of the oracle (test infrastructure): the float64 restatement against
(i) SURVEY.md Appendix C known answers, (ii) a literal torch-autograd transcription of
the reference model, (iii) central finite differences, (iv) the C twin, (v) the committed
golden vectors.  PARITY UNPINNED at the reference level (no reference goldens exist)."""
import glob
import os

import numpy as np
import pytest

import cases
import dc_oracle as O
import dc_oracle_c as OC

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def test_reference_fixture_recipe_matches_survey():
    """tests/conftest.py:7-29 of the reference regenerated: legacy seed-42 streams and
    string-sorted team indices (team "2" has index 12)."""
    td = O.dummy_data_recipe()
    assert list(td["home_goals"][:12]) == [4, 1, 3, 3, 1, 2, 1, 1, 2, 2, 1, 0]
    assert td["home_goals"].sum() == 782 and td["away_goals"].sum() == 645
    assert list(td["away_goals"][:12]) == [2, 3, 1, 2, 3, 3, 2, 2, 2, 2, 0, 4]
    fx, teams = O.fixtures_from_training_data(td)
    assert list(teams[:4]) == ["0", "1", "10", "11"] and teams[12] == "2"
    assert list(fx.away_idx[:5]) == [1, 12, 13, 14, 15]
    x, y = fx.home_goals, fx.away_goals
    counts = [int(((x == a) & (y == b)).sum()) for a, b in ((0, 0), (1, 0), (0, 1), (1, 1))]
    assert counts == [7, 21, 16, 18]


def test_appendix_c_known_answers():
    fx = cases.fixtures("dummy")
    U0, g0, _ = O.potential_and_grad(O.MODEL_BASIC, fx, np.zeros(45))
    assert U0 == pytest.approx(1547.9659390103925, rel=1e-13)
    assert g0[20] == pytest.approx(-6.0, abs=1e-10)  # corr_coef_raw (tie-free entry)
    z = np.random.RandomState(7).uniform(-0.5, 0.5, 45)
    U, g, aux = O.potential_and_grad(O.MODEL_BASIC, fx, z)
    assert U == pytest.approx(1791.4661986284964, rel=1e-13)
    assert aux["rho"] == pytest.approx(-0.05495986387168672, rel=1e-12)
    assert np.linalg.norm(g) == pytest.approx(977.3286321683876, rel=1e-12)
    assert g[:3] == pytest.approx([-54.47257541369639, -26.44913340834539, -24.635457821029256], rel=1e-12)
    assert g[20] == pytest.approx(-2.5820316021936187, rel=1e-12)
    assert g[41] == pytest.approx(-497.53544123337326, rel=1e-12)
    z = np.random.RandomState(7).uniform(-0.5, 0.5, 67)
    U, g, aux = O.potential_and_grad(O.MODEL_EXTENDED, fx, z)
    assert U == pytest.approx(2084.060561351137, rel=1e-13)
    assert aux["rho"] == pytest.approx(-0.113455610805, rel=1e-9)
    assert np.linalg.norm(g) == pytest.approx(1173.063544122, rel=1e-9)
    fxc = cases.fixtures("dummy_cov")
    z = np.random.RandomState(7).uniform(-0.5, 0.5, 77)
    U, g, aux = O.potential_and_grad(O.MODEL_EXTENDED, fxc, z)
    assert U == pytest.approx(2244.148616868948, rel=1e-13)
    assert np.linalg.norm(g) == pytest.approx(1412.524758719, rel=1e-9)
    fxw, _ = O.fixtures_from_training_data(O.dummy_data_recipe())
    fxw.weights = O.time_weights(np.linspace(5, 0, 380), 1.0)
    z = np.random.RandomState(7).uniform(-0.5, 0.5, 67)
    U, g, _ = O.potential_and_grad(O.MODEL_EXTENDED, fxw, z)
    assert U == pytest.approx(492.82799116789965, rel=1e-13)
    assert np.linalg.norm(g) == pytest.approx(264.504106783, rel=1e-9)


ALL = [(O.MODEL_BASIC, n) for n in ("dummy", "timed", "ragged_1", "ragged_777")] + [
    (O.MODEL_EXTENDED, n) for n in ("dummy", "dummy_cov", "dummy_w", "timed_w", "ragged_777")
]


@pytest.mark.parametrize("model,name", ALL)
def test_adjoint_vs_autograd_and_fd(model, name):
    """Hand-derived adjoint == torch autograd of the literal transcription of the reference
    model (1e-12), and ~= central differences (1e-6 relative)."""
    import dc_torch_ref as R

    fx = cases.fixtures(name)
    for pname, z in cases.z_points(model, fx):
        U, g, aux = O.potential_and_grad(model, fx, z)
        Ut, gt, corr = R.potential_and_grad(model, fx, z)
        assert U == pytest.approx(Ut, rel=1e-12), pname
        assert np.abs(g - gt).max() <= 1e-11 * np.abs(gt).max(), pname
        assert aux["rho"] == pytest.approx(corr, abs=1e-13)
        if fx.n <= 400:
            fd = O.finite_difference_grad(model, fx, z)
            assert np.abs(g - fd).max() <= 2e-6 * np.abs(g).max(), pname


@pytest.mark.parametrize("model,name", ALL)
def test_c_twin_matches_numpy(model, name):
    fx = cases.fixtures(name)
    cf = OC.CFixtures(model, fx)
    for pname, z in cases.z_points(model, fx):
        U, g, aux = O.potential_and_grad(model, fx, z)
        for nt in (1, 3):
            Uc, gc, auxc = OC.potential_and_grad(cf, z, nt)
            assert Uc == pytest.approx(U, rel=1e-12), pname
            assert np.abs(g - gc).max() <= 1e-11 * np.abs(g).max(), pname
            assert auxc[0] == pytest.approx(aux["rho"], abs=1e-13)


def test_golden_vectors():
    files = sorted(glob.glob(os.path.join(GOLD, "m[01]_*.npz")))  # (m3_*: tests/test_oracle_neutral.py)
    assert len(files) >= 7
    for f in files:
        d = np.load(f)
        model = int(d["model"])
        fx = O.Fixtures(d["home_idx"], d["away_idx"], d["home_goals"], d["away_goals"],
                        int(d["n_teams"]),
                        weights=None if d["weights"].size == 0 else d["weights"],
                        covariates=None if d["covariates"].size == 0 else d["covariates"])
        cf = OC.CFixtures(model, fx)
        for i in range(d["z"].shape[0]):
            U, g, aux = O.potential_and_grad(model, fx, d["z"][i])
            assert U == pytest.approx(float(d["U"][i]), rel=1e-12), f
            assert np.abs(g - d["grad"][i]).max() <= 1e-11 * np.abs(g).max(), f
            assert aux["rho"] == pytest.approx(float(d["rho"][i]), abs=1e-13)
            assert aux["cond_val"] == pytest.approx(float(d["cond_val"][i]), rel=1e-9)
            if not d["tied"][i]:   # the two tie rules differ at ties only (fixtures of one pair: to rounding)
                assert np.abs(d["grad"][i] - d["grad_first_pair"][i]).max() <= \
                    1e-13 * np.abs(g).max() + 64 * 2.0 ** -53 * aux["cond_grad_raw"]
            # the C restatement: another float64 program (libm's exp instead of numpy's, another summation
            # order).  At the points 1e-4 / 1e-6 from a bound of rho the tau term amplifies the last bits of
            # the rates by 1 / t in the value and 1 / t^2 in the gradient: 64 ulp of the uncapped conditioning
            eps64 = 64 * 2.0 ** -53
            Uc, gc, _ = OC.potential_and_grad(cf, d["z"][i])
            assert abs(Uc - float(d["U"][i])) <= 1e-12 * abs(U) + eps64 * aux["cond_val_raw"], (f, i)
            if not d["tied"][i]:   # (the C restatement keeps the first tied FIXTURE: a third element of the subdifferential)
                assert np.abs(gc - d["grad"][i]).max() <= 1e-11 * np.abs(g).max() + eps64 * aux["cond_grad_raw"], (f, i)
            gf = O.potential_and_grad(model, fx, d["z"][i], ties="first_pair")[1]
            assert np.abs(gf - d["grad_first_pair"][i]).max() <= 1e-11 * np.abs(g).max()


def test_tie_rule_at_zero_matches_autograd():
    """z = 0: every rate is exactly 1, so every fixture attains every extremum of bpl/_util.py:23-30 and both
    two-entry min / max tie as well.  jnp.min / jnp.max split the derivative evenly over tied entries;
    torch.amin / amax / minimum / maximum (oracle/dc_torch_ref.py, the literal transcription) do the same, and
    the hand-derived adjoint restates it (ties="split", the default).  ties="first_pair" -- what the product
    computes -- is a different element of the subdifferential there and the same gradient everywhere else."""
    import dc_torch_ref as TR

    for model, name in [(O.MODEL_BASIC, "dummy"), (O.MODEL_BASIC, "ragged_777"), (O.MODEL_EXTENDED, "dummy_cov"),
                        (O.MODEL_EXTENDED, "timed_w")]:
        fx = cases.fixtures(name)
        D = O.latent_dim(model, fx.n_teams, fx.k if model == O.MODEL_EXTENDED else 0)
        z = np.zeros(D)
        U, g, aux = O.potential_and_grad(model, fx, z)
        Ut, gt = TR.potential_and_grad(model, fx, z)[:2]
        assert aux["tied"]
        assert U == pytest.approx(Ut, rel=1e-14)
        assert np.abs(g - gt).max() <= 1e-12 * np.abs(g).max()
        U1, g1, _ = O.potential_and_grad(model, fx, z, ties="first_pair")
        assert U1 == U and np.abs(g1 - g).max() > 0.5
        z = np.random.RandomState(3).uniform(-0.5, 0.5, D)
        # (fixtures of ONE pair always tie: an even split over them sums to the one-hot, up to rounding)
        ga, gb = O.potential_and_grad(model, fx, z)[1], O.potential_and_grad(model, fx, z, ties="first_pair")[1]
        assert np.abs(ga - gb).max() <= 1e-13 * np.abs(ga).max()


def test_tau_clip_returns_inf_and_zero_adjoint():
    """tol=0 (bpl/_util.py:42): log(clip(arg, 0)) = -inf when rho sits on the bound; the
    clipped branch has zero gradient."""
    fx = cases.fixtures("dummy")
    z = np.random.RandomState(7).uniform(-0.5, 0.5, 45)
    z[41] = 1.0
    z[20] = 60.0  # q -> 1 - eps(float32): rho = UB(1 - eps) + ... stays finite
    U, g, _ = O.potential_and_grad(O.MODEL_BASIC, fx, z)
    assert np.isfinite(U) and np.isfinite(g).all()


def test_edge_cases_single_fixture_and_unused_teams():
    fx = cases.fixtures("ragged_1")
    z = np.random.RandomState(1).uniform(-0.5, 0.5, O.latent_dim(O.MODEL_BASIC, fx.n_teams))
    U, g, _ = O.potential_and_grad(O.MODEL_BASIC, fx, z)
    sl = O.site_slices(O.MODEL_BASIC, fx.n_teams)
    # teams that play no fixture only see their N(0,1) prior: dU/dz = z
    unused = [t for t in range(fx.n_teams) if t not in (fx.home_idx[0], fx.away_idx[0])]
    assert g[sl["attack_decentered"]][unused] == pytest.approx(z[sl["attack_decentered"]][unused])
    assert np.isfinite(U)


def test_cpu_port_matches_the_float64_oracle():
    """oracle/dc_cpu_port.c (the CPU comparator bench.py times beside the GPU: the HIP kernel's
    algorithm on host cores, float32 per-fixture arithmetic) against the float64 oracle, every
    case family and latent point, one and several threads: float32-table tolerances."""
    import cases
    import dc_oracle as O
    import dc_oracle_c as OC

    for model, name in [(O.MODEL_BASIC, "dummy"), (O.MODEL_BASIC, "ragged_777"), (O.MODEL_BASIC, "league_1e5"),
                        (O.MODEL_EXTENDED, "dummy_cov"), (O.MODEL_EXTENDED, "dummy_w"),
                        (O.MODEL_EXTENDED, "timed_w"), (O.MODEL_EXTENDED, "ragged_5000")]:
        fx = cases.fixtures(name)
        cf = OC.CFixtures(model, fx)
        for nt in (1, 3):
            port = OC.CpuPort(cf, nt)
            for pname, z in cases.z_points(model, fx):
                Uo, go, auxo = O.potential_and_grad(model, fx, z)
                U, g, aux = port.eval(z)
                assert abs(U - Uo) <= 2e-6 * abs(Uo), (name, pname, nt)
                assert np.abs(g - go).max() <= 5e-6 * np.abs(go).max(), (name, pname, nt)
                assert abs(aux[0] - auxo["rho"]) <= 1e-6
            port.close()
