"""Shared seeded test cases: (model, Fixtures, z) triples used by the CPU oracle tests,
the golden-vector generator and the GPU parity tests."""
import numpy as np

import dc_oracle as O


def float32_weights(td, eps, rescale=False):
    """Weights as the product path stores them: float64 formula, float32 storage."""
    w = O.time_weights(td, eps, rescale)
    return w.astype(np.float32).astype(np.float64)


def fixtures(name):
    if name == "dummy":
        fx, _ = O.fixtures_from_training_data(O.dummy_data_recipe())
        return fx
    if name == "dummy_cov":
        fx, _ = O.fixtures_from_training_data(O.dummy_data_recipe())
        fx.covariates = np.random.RandomState(0).normal(size=(20, 5))
        return fx
    if name == "dummy_w":
        fx, _ = O.fixtures_from_training_data(O.dummy_data_recipe())
        fx.weights = float32_weights(np.linspace(5, 0, 380), 1.0)
        return fx
    if name == "timed":
        td = O.timed_dummy_data_recipe()
        fx, _ = O.fixtures_from_training_data(
            {k: td[k] for k in ("home_team", "away_team", "home_goals", "away_goals")}
        )
        return fx
    if name == "timed_w":
        fx = fixtures("timed")
        fx.weights = float32_weights(O.timed_dummy_data_recipe()["time_diff"], 1.0, True)
        return fx
    if name.startswith("leaguew"):  # long runs per pair AND time weights (uniform-lane path)
        n = int(float(name.split("_")[1]))
        h, a, x, y = O.synthetic_league(n)
        fx = O.Fixtures(h, a, x, y, 20)
        fx.weights = float32_weights(np.linspace(5, 0, n), 1.0)
        return fx
    if name.startswith("league"):
        n = int(float(name.split("_")[1]))
        h, a, x, y = O.synthetic_league(n)
        return O.Fixtures(h, a, x, y, 20)
    if name.startswith("wide"):  # many teams (> 64: the general tail epilogue): wide_N_T
        _, n, T = name.split("_")
        n, T = int(n), int(T)
        rs = np.random.RandomState(n + T)
        h = rs.randint(0, T, n)
        a = (h + 1 + rs.randint(0, T - 1, n)) % T
        return O.Fixtures(h, a, rs.poisson(1.4, n), rs.poisson(1.1, n), T)
    if name.startswith("ragged"):
        # random (not tiled) pairs incl. teams that never play at home, odd N (tail tile)
        n = int(name.split("_")[1])
        rs = np.random.RandomState(n)
        T = 37
        h = rs.randint(0, T - 3, n)
        a = (h + 1 + rs.randint(0, T - 1, n)) % T
        return O.Fixtures(h, a, rs.poisson(1.4, n), rs.poisson(1.1, n), T)
    raise KeyError(name)


def z_points(model, fx, n_random=3):
    """A few latent points: uniform(-.5,.5) seeds, one forcing M>1 (UB branch), and for
    the extended model one forcing the rate clip at 15."""
    D = O.latent_dim(model, fx.n_teams, fx.k if model == O.MODEL_EXTENDED else 0)
    sl = O.site_slices(model, fx.n_teams, fx.k if model == O.MODEL_EXTENDED else 0)
    pts = []
    for s in (7, 11, 123)[:n_random]:
        pts.append((f"u{s}", np.random.RandomState(s).uniform(-0.5, 0.5, D)))
    z = np.random.RandomState(5).uniform(-0.5, 0.5, D)
    if model == O.MODEL_BASIC:
        z[sl["home_advantage"]] = 1.0
    else:
        z[sl["mean_home_advantage"]] = 1.0
    pts.append(("ub", z))
    if model == O.MODEL_EXTENDED:
        z = np.random.RandomState(9).uniform(-0.5, 0.5, D)
        z[sl["std_attack"]] = 1.5
        z[sl["standardised_attack"]] *= 4
        pts.append(("clip", z))
    return pts


def u_tolerance(n_fixtures, U):
    """The stated bound on |U_hip - U_float64| for the float32-table kernels (models 0 / 1):
    TWICE the tolerance SURVEY.md section 8c asks for,  2 (1e-6 sqrt(N) + 1e-9 |U|)  (+ 1e-9): 9e-3 at
    N = 1e6, U = 3.6e6.  Measured (profiles/r03/parity_errors.txt, every case of tests/test_gpu_parity.py
    incl. the rate-clip points and BASELINE config 3 at full size): at most 0.61 of it.
    Round 2's gate was 3e-7 (|U| + 4N) / sqrt(P), x20 at the clip points: 0.12 and 2.4 there.  What
    made the difference ("Numerics"): the float32 rate product, the clipped-rate
    log term and the tail's spurious table correction of clipped lanes are now corrected exactly, per
    pair, in float64 by the prior workgroup; the tau argument 1 + rho c carries its own rounding
    error; the per-lane rate sums are float64."""
    return 2.0 * (1e-6 * (float(n_fixtures) ** 0.5) + 1e-9 * abs(U)) + 1e-9
