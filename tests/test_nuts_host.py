"""Host logic of the NUTS driver (bpl-next_amd/csrc/nuts.hpp) without a GPU: the product's
header-only driver is linked against CPU potentials by the TEST harness
oracle/nuts_harness.cpp (the C oracle's Dixon-Coles potential, or a Gaussian).
numpyro is not in the reference tree; these are the sampler-invariant checks of
SURVEY.md §7.3-5 (ladder L2/L3) and the schedule facts of Appendix B."""
import ctypes as C

import numpy as np
import pytest

import cases
import dc_oracle as O
import dc_oracle_c as OC


def test_adaptation_schedule_matches_appendix_b():
    h = OC.harness()
    buf = (C.c_int * 64)()

    def sched(n):
        k = h.harness_schedule(n, buf, 32)
        return [(buf[2 * i], buf[2 * i + 1]) for i in range(k)]

    assert sched(500) == [(0, 74), (75, 99), (100, 149), (150, 249), (250, 449), (450, 499)]
    assert sched(100) == [(0, 14), (15, 89), (90, 99)]
    assert sched(10) == [(0, 9)]
    assert sched(1000)[0] == (0, 74) and sched(1000)[-1] == (950, 999)


def test_checkpoint_indices():
    # numpyro _leaf_idx_to_ckpt_idxs: idx_max = popcount(n >> 1), idx_min = idx_max - trailing_ones(n) + 1
    h = OC.harness()
    a, b = C.c_int(), C.c_int()
    expect = {0: (1, 0), 1: (0, 0), 2: (2, 1), 3: (0, 1), 5: (1, 1), 6: (3, 2), 7: (0, 2), 13: (2, 2)}
    for n, (imin, imax) in expect.items():
        h.harness_ckpt_idxs(n, C.byref(a), C.byref(b))
        assert (a.value, b.value) == (imin, imax), n


def test_gaussian_target_invariants():
    """NUTS on N(0, diag(sd^2)), D = 10: sample moments, adapted mass matrix ~ variances,
    mean accept prob ~ target 0.8, no divergences, seed-determinism."""
    sd = np.array([0.1, 0.3, 1, 1, 2, 3, 0.5, 5, 1, 0.2])
    rc, draws, stats, summ = OC.nuts_gauss(sd, 500, 2000, (0, 123))
    assert rc == 0
    assert np.all(np.abs(draws.mean(axis=0)) < 0.15 * sd)
    assert np.all(np.abs(draws.std(axis=0) / sd - 1) < 0.12)
    assert 0.7 < summ[1] < 0.92          # mean accept prob after warm-up
    assert summ[3] == 0                  # divergences
    inv_mass = summ[4:]
    assert np.all(np.abs(np.log(inv_mass / sd**2)) < 0.6)
    rc2, draws2, _, _ = OC.nuts_gauss(sd, 500, 2000, (0, 123))
    assert np.array_equal(draws, draws2)
    rc3, draws3, _, _ = OC.nuts_gauss(sd, 500, 2000, (0, 124))
    assert not np.array_equal(draws, draws3)
    # energy of a standard normal target: E[U] = D/2
    assert abs(stats[:, 0].mean() - 5.0) < 0.4
    assert stats[:, 2].max() <= 1023 and stats[:, 2].min() >= 1


def test_thinning_and_init_params():
    sd = np.ones(4)
    rc, d1, _, _ = OC.nuts_gauss(sd, 50, 40, (0, 1), thin=4)
    assert rc == 0 and d1.shape == (10, 4)
    rc, d2, _, _ = OC.nuts_gauss(sd, 0, 5, (0, 1), z0=np.full(4, 0.25))
    assert rc == 0 and np.isfinite(d2).all()


def test_max_tree_depth_bounds_steps():
    sd = np.full(3, 1e-3)  # step size 1 is far too large at first -> many halvings
    rc, _, stats, _ = OC.nuts_gauss(sd, 100, 50, (0, 5), depth=3)
    assert rc == 0 and stats[:, 2].max() <= 7


@pytest.mark.parametrize("model", [O.MODEL_BASIC, O.MODEL_EXTENDED])
def test_dixon_coles_posterior_on_cpu_potential(model):
    """The driver on the oracle's Dixon-Coles potential (reference dummy_data): finite
    draws, healthy acceptance, home advantage and mean rates recovered (the data are iid
    Poisson(2.1)/Poisson(1.7): exp(home_adv) ~ 2.1/1.7)."""
    fx = cases.fixtures("dummy")
    cf = OC.CFixtures(model, fx)
    rc, draws, stats, summ = OC.nuts_dc(cf, 150, 150, (0, 42))
    assert rc == 0 and np.isfinite(draws).all()
    assert 0.6 < summ[1] <= 1.0
    sl = O.site_slices(model, 20)
    if model == O.MODEL_BASIC:
        ha = draws[:, sl["home_advantage"]].mean()
    else:
        ha = draws[:, sl["mean_home_advantage"]].mean()
    assert abs(ha - np.log(2.1 / 1.7)) < 0.12
    # potential energy along the chain is consistent with the oracle at the same points
    U, _, _ = O.potential_and_grad(model, fx, draws[-1])
    assert U == pytest.approx(stats[-1, 0], rel=1e-10)
