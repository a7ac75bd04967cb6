"""Generates tests/golden/*.npz -- golden input/output vectors of the hot path.

PARITY UNPINNED (see oracle/dc_oracle.py): the reference's tests hold no numbers for
this path and numpyro/jax are not installed, so these vectors come from the float64
restatement, after it has been checked three ways (tests/test_oracle.py): against a
literal torch-autograd transcription of the reference model, against central finite
differences, and against the known answers of SURVEY.md Appendix C.  Inputs are the
reference's own fixture recipes (tests/conftest.py:7-62) plus seeded synthetic leagues.

Run:  python oracle/make_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path[:0] = [HERE, os.path.join(os.path.dirname(HERE), "tests")]

import cases  # noqa: E402
import dc_oracle as O  # noqa: E402

OUT = os.path.join(os.path.dirname(HERE), "tests", "golden")

GOLDEN_CASES = [
    (O.MODEL_BASIC, "dummy"),
    (O.MODEL_BASIC, "timed"),
    (O.MODEL_BASIC, "ragged_777"),
    (O.MODEL_EXTENDED, "dummy"),
    (O.MODEL_EXTENDED, "dummy_cov"),
    (O.MODEL_EXTENDED, "dummy_w"),
    (O.MODEL_EXTENDED, "timed_w"),
    # "the N=1e5 synthetic of section 8d" (the fixtures are data too: 0.2 MB each)
    (O.MODEL_BASIC, "league_1e5"),
    (O.MODEL_EXTENDED, "league_1e5"),
]


def main():
    os.makedirs(OUT, exist_ok=True)
    for model, name in GOLDEN_CASES:
        fx = cases.fixtures(name)
        # SURVEY.md section 8c's list: z = 0, RandomState(7).uniform(-.5, .5), 8 further random points, the
        # UB-branch point, the rate-clip point (extended), rho 1e-2 / 1e-4 / 1e-6 from each of its bounds
        pts = cases.golden_points(model, fx)
        zs, Us, gs, rhos, lbs, ubs, atts, dfns, has, cvs, cgs = [], [], [], [], [], [], [], [], [], [], []
        gfp, tied = [], []
        for _, z in pts:
            U, g, aux = O.potential_and_grad(model, fx, z)   # the reference's tie rule (even split)
            zs.append(z)
            Us.append(U)
            gs.append(g)
            # the product's element of the subdifferential where extremal rates tie (z = 0; clipped rates):
            # one arg-extremal pair, the smallest (home, away) key -- see dc_oracle.likelihood_and_adjoint
            gfp.append(O.potential_and_grad(model, fx, z, ties="first_pair")[1])
            tied.append(aux["tied"])
            rhos.append(aux["rho"])
            lbs.append(aux["LB"])
            ubs.append(aux["UB"])
            cvs.append(aux["cond_val"])
            cgs.append(aux["cond_grad"])
            atts.append(aux["attack"])
            dfns.append(aux["defence"])
            has.append(np.broadcast_to(aux["home_advantage"], (fx.n_teams,)))
        np.savez_compressed(
            os.path.join(OUT, f"m{model}_{name}.npz"),
            model=model,
            home_idx=fx.home_idx.astype(np.uint16),
            away_idx=fx.away_idx.astype(np.uint16),
            home_goals=fx.home_goals.astype(np.uint8),
            away_goals=fx.away_goals.astype(np.uint8),
            n_teams=fx.n_teams,
            weights=np.zeros(0) if fx.weights is None else fx.weights,
            covariates=np.zeros((0, 0)) if fx.covariates is None else fx.covariates,
            point_names=np.array([p[0] for p in pts]),
            z=np.stack(zs),
            U=np.array(Us),
            grad=np.stack(gs),
            grad_first_pair=np.stack(gfp),
            tied=np.array(tied),
            rho=np.array(rhos),
            LB=np.array(lbs),
            UB=np.array(ubs),
            # conditioning of the tau term at each point (cases.u_tolerance_cond / g_tolerance_cond)
            cond_val=np.array(cvs),
            cond_grad=np.array(cgs),
            # deterministic sites at each point
            attack=np.stack(atts),
            defence=np.stack(dfns),
            home_advantage=np.stack(has),
        )
        print(f"wrote m{model}_{name}.npz  ({len(pts)} points, N={fx.n})")
    neutral_golden()
    dynamic_golden()


def dynamic_golden():
    """Dynamic (time-varying) model, BASELINE config 4's model at a small size (7 teams x 5 gameweeks,
    300 fixtures, 3 covariates; oracle/dc_dynamic_oracle.py small_recipe): both walk settings."""
    import dc_dynamic_oracle as DO

    fx = DO.small_recipe(k=3)
    D = DO.latent_dim(fx.n_gameweeks, fx.n_teams, fx.k)
    sl = DO.site_slices(fx.n_gameweeks, fx.n_teams, fx.k)
    # (no z = 0: every rate ties there, and which arg-extremal fixture carries the bounds' adjoint is a
    # convention of this float64 path, not of a reference -- the class is unfinished upstream)
    zs = [np.random.RandomState(s).uniform(-sc, sc, D) for s, sc in ((7, 0.3), (2, 0.3), (3, 0.3), (4, 1.0))]
    zs[1][sl["mean_home_attack"]] = 1.2
    out = {}
    for rw in (1, 0):
        res = [DO.potential_and_grad(fx, z, bool(rw)) for z in zs]
        out[f"U_rw{rw}"] = np.array([r[0] for r in res])
        out[f"grad_rw{rw}"] = np.stack([r[1] for r in res])
        out[f"rho_rw{rw}"] = np.array([r[2]["rho"] for r in res])
    np.savez_compressed(
        os.path.join(OUT, "m2_small_cov.npz"), model=2,
        home_idx=fx.home_idx.astype(np.uint16), away_idx=fx.away_idx.astype(np.uint16),
        home_goals=fx.home_goals.astype(np.uint8), away_goals=fx.away_goals.astype(np.uint8),
        gameweek=fx.gameweek.astype(np.uint16), neutral=fx.neutral.astype(np.uint8),
        n_teams=fx.n_teams, n_gameweeks=fx.n_gameweeks, covariates=fx.covariates, z=np.stack(zs), **out)
    print(f"wrote m2_small_cov.npz  ({len(zs)} points x 2 walk settings, N={fx.n})")


def neutral_golden():
    """Neutral-venue model (row f-4): the reference's `neutral_dummy_data` recipe
    (tests/conftest.py:66-117) with time decay + rescaling and 3 covariates."""
    import dc_neutral_oracle as NO

    dd = NO.neutral_dummy_recipe()
    cov = np.random.RandomState(0).normal(size=(20, 3))
    for name, fx in (("dummy", NO.fixtures_from_data(dd)),
                     ("dummy_eps_cov", NO.fixtures_from_data(dd, epsilon=0.3, rescale_weights=True,
                                                             covariates=cov))):
        fx.weights = fx.weights.astype(np.float32).astype(np.float64)  # as the device holds them
        D = NO.latent_dim(fx.n_teams, fx.k)
        zs = [np.random.RandomState(s).uniform(-sc, sc, D) for s, sc in ((1, 0.2), (2, 0.5), (3, 1.0))]
        res = [NO.potential_and_grad(fx, z) for z in zs]
        np.savez_compressed(
            os.path.join(OUT, f"m3_{name}.npz"), model=3,
            home_idx=fx.home_idx.astype(np.uint16), away_idx=fx.away_idx.astype(np.uint16),
            home_goals=fx.home_goals.astype(np.uint8), away_goals=fx.away_goals.astype(np.uint8),
            neutral=fx.neutral.astype(np.uint8), weights=fx.weights, n_teams=fx.n_teams,
            covariates=np.zeros((0, 0)) if fx.covariates is None else fx.covariates,
            z=np.stack(zs), U=np.array([r[0] for r in res]), grad=np.stack([r[1] for r in res]),
            rho=np.array([r[2]["rho"] for r in res]), LB=np.array([r[2]["LB"] for r in res]),
            UB=np.array([r[2]["UB"] for r in res]))
        print(f"wrote m3_{name}.npz  (3 points, N={fx.n})")


if __name__ == "__main__":
    main()
