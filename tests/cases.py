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
    if name.startswith("dummy_covk_"):   # the reference's recipe with K covariates: dummy_covk_K
        k = int(name.split("_")[2])
        fx, _ = O.fixtures_from_training_data(O.dummy_data_recipe())
        fx.covariates = np.random.RandomState(100 + k).normal(size=(20, k))
        return fx
    if name.startswith("ragged64cov_"):   # 64 teams (the widest one-lane-per-team epilogue), K covariates
        k = int(name.split("_")[1])
        rs = np.random.RandomState(640 + k)
        n, T = 9000, 64
        h = rs.randint(0, T, n)
        a = (h + 1 + rs.randint(0, T - 1, n)) % T
        fx = O.Fixtures(h, a, rs.poisson(1.4, n), rs.poisson(1.1, n), T)
        fx.covariates = rs.normal(size=(T, k))
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
    if name.startswith("widewc"):  # many teams, time weights AND covariates: widewc_N_T
        _, n, T = name.split("_")
        fx = fixtures(f"wide_{n}_{T}")
        fx.weights = float32_weights(np.linspace(4, 0, int(n)), 1.0)
        fx.covariates = np.random.RandomState(int(T)).normal(size=(int(T), 3))
        return fx
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


def _corr_slot(model, fx):
    K = fx.k if model == O.MODEL_EXTENDED else 0
    return O.site_slices(model, fx.n_teams, K)["corr_coef_raw"].start


def near_bound_points(model, fx, distances=(1e-2, 1e-4, 1e-6)):
    """Latent points whose correlation coefficient sits a given ABSOLUTE distance from one of its
    bounds (bpl/_util.py:17-31: rho = LB + sigmoid(corr_coef_raw) (UB - LB)): there one of the tau
    arguments of bpl/_util.py:58-85 -- 1 - rho lh la at the arg-max pair (upper bound, UB = 1/M),
    1 - rho (upper bound, UB = 1), 1 + rho lh / 1 + rho la at the largest rate (lower bound) -- is the
    distance times O(1), and its log and reciprocal amplify every rounding of rho and of the rates
    (tol = 0, bpl/_util.py:42).  SURVEY.md section 8c asks for such a point ("rho within 1e-6 of a bound").
    The bounds do not depend on corr_coef_raw, so it is solved for: q = (rho - LB) / (UB - LB).
    Three bases: M > 1 (UB = 1/M binds), every rate < 1 (UB = 1: the (1,1) term), and the lower bound."""
    D = O.latent_dim(model, fx.n_teams, fx.k if model == O.MODEL_EXTENDED else 0)
    sl = O.site_slices(model, fx.n_teams, fx.k if model == O.MODEL_EXTENDED else 0)
    ic = _corr_slot(model, fx)
    ha = "home_advantage" if model == O.MODEL_BASIC else "mean_home_advantage"
    pts = []
    for base, side in (("M", "ub"), ("1", "ub"), ("L", "lb")):
        z = np.random.RandomState(31).uniform(-0.5, 0.5, D)
        if base == "M":
            z[sl[ha]] = 1.0                 # M = max lh la > 1
        elif base == "1":
            z[sl["mean_defence"]] = 1.6     # every rate < 1: UB = 1
            z[sl[ha]] = 0.0
            for site in ("attack_coefficients", "defence_coefficients"):
                if site in sl:
                    z[sl[site]] *= 0.1
        _, _, aux = O.potential_and_grad(model, fx, z)
        LB, UB = aux["LB"], aux["UB"]
        assert base == "L" or (UB < 1.0) == (base == "M"), (base, UB)
        for d in distances:
            q = 1.0 - d / (UB - LB) if side == "ub" else d / (UB - LB)
            zz = z.copy()
            zz[ic] = np.log(q) - np.log1p(-q)
            pts.append((f"{side}{base}-{d:.0e}", zz))
    return pts


def golden_points(model, fx):
    """SURVEY.md section 8c's list: z = 0, RandomState(7).uniform(-.5, .5), 8 further seeded random points,
    the UB-branch point, (extended) a rate-clip point, and rho 1e-2 / 1e-4 / 1e-6 from each bound."""
    D = O.latent_dim(model, fx.n_teams, fx.k if model == O.MODEL_EXTENDED else 0)
    pts = [("zero", np.zeros(D))] + z_points(model, fx)
    pts += [(f"r{s}", np.random.RandomState(1000 + s).uniform(-0.7, 0.7, D)) for s in range(8)]
    return pts + near_bound_points(model, fx)


EPS32 = 2.0 ** -24   # unit roundoff of float32


def u_tolerance_cond(n_fixtures, U, aux):
    """u_tolerance + what the CONDITIONING of the tau term allows the float32 part of the kernel.
    Its per-fixture inputs are float32: rho, and the rates as products of two float32 table entries.  The
    first-order effect of the table and rho roundings on U is removed exactly (DESIGN.md section 4
    "Numerics"), the rounding of 1 + rho c itself is carried (class_terms); what is left per low-score
    fixture is the three roundings of the products lh = t t', la = t t', c = lh la -- a relative error
    <= 3 EPS32 of rho c, i.e. 3 EPS32 |rho c| / t in log t.  Classes with t < 1/64 never see float32 (the
    tail workgroup's float64 pass, ill_core), so the amplification 1 / t is capped at 64:
        |dU| <= u_tolerance(N, U) + 4 EPS32 cond_val,   cond_val = sum_i w_i |rho c_i| min(1 / t_i, 64)
    (oracle/dc_oracle.py).  Away from the bounds the second term is < 2 % of the first; 1e-6 from a bound it
    is < 30 % of it -- round 3's kernel was off by O(1) there and was gated at 2e-2 |U|."""
    return u_tolerance(n_fixtures, U) + 4.0 * EPS32 * aux["cond_val"]


def g_tolerance_cond(g, aux):
    """|dgrad|_inf gate: 5e-7 |grad|_inf + 1e-7 (float32 tables) + the conditioning of the tau term's
    derivative rho c / t, which a relative input error d moves by |rho c| / t^2 d:
    8 EPS32 cond_grad, cond_grad = the largest per-team sum of w |rho c| min(1 / t, 64)^2 (capped as above;
    the chain rule to z multiplies by std = O(1) factors, which the 8 covers)."""
    return 5e-7 * np.abs(g).max() + 1e-7 + 8.0 * EPS32 * aux["cond_grad"]


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
