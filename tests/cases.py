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


EPS32 = 2.0 ** -24
LANE_FIXTURES, TILE_LANES = 32, 64   # csrc/dc_layout.h: fixtures per lane, lanes per wave tile


def u_tolerance(home_idx, away_idx, home_goals, away_goals, weights, attack, defence, home_advantage,
                clip, U):
    """The stated bound on |U_hip - U_float64| for the float32-table kernels (models 0 / 1), from the
    kernel's own error sources ("Numerics"); eps = 2^-24:

      3 eps Q         the float32 product of the two table entries of a pair is rounded ONCE per pair and
                      multiplies all its fixtures: Q = sqrt(sum_pairs (sum_fixtures w (lh + la))^2)
                      (the table entries' own rounding is removed to first order by the kernel);
      3 eps R / sqrt(tiles)   the per-lane sums w (lh + la) are added across the 64 lanes of a wave tile
                      in float32 before they reach the float64 accumulators: R = sum_fixtures w (lh + la),
                      tiles = ceil(lanes / 64), lanes = sum_pairs ceil(n_pair / 32);
      64 eps QC       extended model, pairs whose rate is clipped at 15: k log(raw rate) goes through
                      v_log_f32 (1 ulp of a base-2 logarithm of magnitude <= 16, i.e. <= 32 eps, twice),
                      common to the fixtures of a pair: QC = sqrt(sum_pairs (sum_clipped w k)^2);
      1e-12 |U| + 1e-9  float64 bookkeeping.
    Measured errors (profiles/r03/parity_errors.txt) are 0.04 .. 0.4 of this bound."""
    h, a = np.asarray(home_idx, np.int64), np.asarray(away_idx, np.int64)
    att, dfn = np.asarray(attack, np.float64), np.asarray(defence, np.float64)
    ha = np.broadcast_to(np.asarray(home_advantage, np.float64), att.shape)
    w = np.ones(h.size) if weights is None or np.size(weights) == 0 else np.asarray(weights, np.float64)
    lh, la = np.exp(att[h] - dfn[a] + ha[h]), np.exp(att[a] - dfn[h])
    clipped = np.zeros(h.size)
    if clip:
        clipped = np.asarray(home_goals) * (lh > 15.0) + np.asarray(away_goals) * (la > 15.0)
        lh, la = np.minimum(lh, 15.0), np.minimum(la, 15.0)
    _, pair, count = np.unique(h * 65536 + a, return_inverse=True, return_counts=True)
    Q = np.sqrt((np.bincount(pair, weights=w * (lh + la)) ** 2).sum())
    QC = np.sqrt((np.bincount(pair, weights=w * clipped) ** 2).sum())
    R = float((w * (lh + la)).sum())
    lanes = int(np.ceil(count / LANE_FIXTURES).sum())
    tiles = max(1, -(-lanes // TILE_LANES))
    return 3 * EPS32 * (Q + R / np.sqrt(tiles)) + 64 * EPS32 * QC + 1e-12 * abs(U) + 1e-9
