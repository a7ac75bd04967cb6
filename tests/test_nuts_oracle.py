"""Row a11 (numpyro's NUTS / warm-up / MCMC, bpl/dixon_coles.py:100-116) checked against
something that is NOT the product's own driver: oracle/nuts_oracle.py, an independent numpy
restatement of numpyro 0.13.2's sampler, and the trajectories it wrote to tests/golden/nuts_*.npz
(oracle/make_nuts_golden.py).  No GPU: the product's host driver (bpl-next_amd/csrc/nuts.hpp) runs
here on CPU potentials through the test harness; tests/test_gpu_nuts_golden.py runs the
device-resident chains against the same files."""
import os

import numpy as np
import pytest

import cases
import dc_oracle as O
import dc_oracle_c as OC
import nuts_oracle as NO

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def gold(name):
    return np.load(os.path.join(GOLD, name + ".npz"))


# ---------------------------------------------------------------- the oracle's own pins


def test_oracle_prng_is_pinned_from_outside():
    # Threefry-2x32-20 known answers (Random123 kat_vectors / jax tests/random_test.py)
    assert NO.threefry_block(0, 0, 0, 0) == (0x6B200159, 0x99BA4EFE)
    assert NO.threefry_block(0xFFFFFFFF, 0xFFFFFFFF, 0xFFFFFFFF, 0xFFFFFFFF) == (0x1CB996FC, 0xBB002BE7)
    assert NO.threefry_block(0x13198A2E, 0x03707344, 0x243F6A88, 0x85A308D3) == (0xC4923A9C, 0x483DF7A0)
    # values jax publishes (tests / docs, jax 0.4.x, non-partitionable threefry)
    assert NO.random_bits(NO.prng_key(1701), 3) == [56197195, 4200222568, 961309823]
    assert NO.split(NO.prng_key(42), 2) == [(2465931498, 3679230171), (255383827, 267815257)]
    assert NO.split(NO.prng_key(0), 2) == [(4146024105, 967050713), (2718843009, 1272950319)]
    ref = np.array([0.18784384, -1.2833426, -0.2710917, 1.2490594, 0.24447003], dtype=np.float32)
    assert np.abs(NO.normal(NO.prng_key(0), 5) - ref).max() < 2e-7          # float64 inverse CDF
    got32 = NO.normal(NO.prng_key(0), 5, float32_result=True)
    assert np.abs(got32.view(np.int32) - ref.view(np.int32)).max() <= 1     # jax: last float32 bit


def test_oracle_schedule_and_checkpoint_tables():
    # SURVEY.md Appendix B.4 (Stan's windows as numpyro builds them)
    assert NO.build_adaptation_schedule(500) == [(0, 74), (75, 99), (100, 149), (150, 249), (250, 449), (450, 499)]
    assert NO.build_adaptation_schedule(100) == [(0, 14), (15, 89), (90, 99)]
    assert NO.build_adaptation_schedule(19) == [(0, 18)]
    # numpyro's comments in _leaf_idx_to_ckpt_idxs: idx_max 6 -> 2, 7 -> 2, 13 -> 2;
    # trailing ones 6 -> 0, 7 -> 3, 13 -> 1
    assert NO.leaf_idx_to_ckpt_idxs(6) == (3, 2)
    assert NO.leaf_idx_to_ckpt_idxs(7) == (0, 2)
    assert NO.leaf_idx_to_ckpt_idxs(13) == (2, 2)
    assert NO.leaf_idx_to_ckpt_idxs(0) == (1, 0)


def test_oracle_reproduces_its_goldens():
    g = gold("nuts_gauss_fixed")
    sd = g["sd"]
    o = NO.run_chain(lambda z: (0.5 * float(np.sum(z * z / sd ** 2)), z / sd ** 2), tuple(int(k) for k in g["key"]),
                     int(g["num_warmup"]), int(g["num_samples"]), z0=g["z0"], step_size=float(g["step_size0"]))
    assert o["num_steps"].tolist() == g["num_steps"].tolist()
    assert np.array_equal(o["draws"], g["draws"])


def test_oracle_gaussian_statistics():
    """Sampler invariants of the restatement itself: moments of a diagonal Gaussian, adapted
    step size giving ~0.8 acceptance, adapted inverse mass matrix ~ the variances."""
    sd = np.array([1.0, 3.0, 0.3])
    o = NO.run_chain(lambda z: (0.5 * float(np.sum(z * z / sd ** 2)), z / sd ** 2), (0, 3), 300, 700,
                     z0=np.zeros(3))
    d = o["draws"]
    assert np.abs(d.mean(0) / sd).max() < 0.2
    assert np.abs(d.std(0) / sd - 1).max() < 0.15
    assert 0.6 < o["accept_prob"][300:].mean() < 0.95
    assert np.abs(np.log(o["inverse_mass_matrix"] / sd ** 2)).max() < 0.7
    assert not o["diverging"][300:].any()


# ---------------------------------------------------------------- the product's host driver


def _harness(g, pot_kind, cf=None):
    key = tuple(int(k) for k in g["key"])
    warm, samp = int(g["num_warmup"]), int(g["num_samples"])
    z0 = g["z0"] if bool(g["z0_given"]) else None
    depth = int(g["max_tree_depth"])
    if pot_kind == "gauss":
        return OC.nuts_gauss(g["sd"], warm, samp, key, depth=depth, z0=z0, step_size=float(g["step_size0"]))
    return OC.nuts_dc(cf, warm, samp, key, depth=depth, z0=z0, step_size=float(g["step_size0"]))


@pytest.mark.parametrize("name,kind,model,fix", [
    ("nuts_gauss_fixed", "gauss", None, None),
    ("nuts_gauss_adapt", "gauss", None, None),
    ("nuts_dummy_basic_fixed", "dc", O.MODEL_BASIC, "dummy"),
    ("nuts_dummy_basic_adapt", "dc", O.MODEL_BASIC, "dummy"),
    ("nuts_dummy_basic_adapt_shallow", "dc", O.MODEL_BASIC, "dummy"),
    ("nuts_dummy_ext_fixed", "dc", O.MODEL_EXTENDED, "dummy_cov"),
])
def test_host_driver_follows_the_oracle(name, kind, model, fix):
    """Same key, same start, same potential: the product's tree builder, transitions, dual
    averaging, Welford windows and key-split order must reproduce the independent restatement's
    trajectory -- tree sizes exactly, draws to rounding (amplified ~30x per transition by the
    chaotic trajectories, hence the graded tolerances)."""
    g = gold(name)
    cf = OC.CFixtures(model, cases.fixtures(fix)) if kind == "dc" else None
    rc, draws, stats, summ = _harness(g, kind, cf)
    assert rc == 0
    w = int(g["num_warmup"])
    assert stats[:, 2].astype(int).tolist() == g["num_steps"][w:].tolist()
    assert stats[:, 3].astype(bool).tolist() == g["diverging"][w:].tolist()
    assert int(summ[2]) == int(g["num_steps"].sum())          # every leapfrog, warm-up included
    # fixed step: rounding only.  After w adapted transitions the 1e-16 differences have been
    # amplified by every trajectory they went through (deep trees on the Dixon-Coles posterior)
    tol = 1e-10 if w == 0 else (1e-5 if kind == "gauss" else (1e-9 if "shallow" in name else 5e-2))
    assert np.abs(draws[:4] - g["draws"][:4]).max() < tol
    assert np.abs(draws - g["draws"]).max() < max(1e-3, 4 * tol)
    assert np.abs(stats[:4, 1] - g["accept_prob"][w:w + 4]).max() < max(tol, 1e-9)
    assert abs(summ[0] / float(g["final_step_size"]) - 1) < (1e-12 if w == 0 else tol)
    if w:
        assert np.abs(summ[4:] / g["inverse_mass_matrix"] - 1).max() < tol


@pytest.mark.parametrize("name,model,fix", [("nuts_dummy_basic_init", O.MODEL_BASIC, "dummy"),
                                            ("nuts_dummy_ext_init", O.MODEL_EXTENDED, "dummy_cov")])
def test_host_driver_initial_point_follows_the_oracle(name, model, fix):
    """init_to_uniform(radius=2) through find_valid_initial_params (numpyro/infer/util.py, the
    branch that does not trace the model): one uniform(-2, 2) block per latent site in model trace
    order, keys split as numpyro splits them, retried until U and grad U are finite."""
    g = gold(name)
    cf = OC.CFixtures(model, cases.fixtures(fix))
    key = tuple(int(k) for k in g["key"])
    rc, draws, stats, summ = OC.nuts_dc(cf, 0, 1, key, z0=None, step_size=1e-12)
    assert rc == 0
    assert np.abs(draws[0] - g["z0"]).max() < 1e-7  # (one leapfrog of 1e-12 away)
    rc, draws, stats, summ = _harness(g, "dc", cf)
    w = int(g["num_warmup"])
    assert stats[:, 2].astype(int).tolist() == g["num_steps"][w:].tolist()
    assert np.abs(draws - g["draws"]).max() < 5e-3  # (20 adapted transitions from a wild start)
