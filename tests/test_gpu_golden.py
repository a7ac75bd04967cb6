"""GPU: the HIP path (through the C-ABI) against the committed golden vectors alone
(tests/golden/*.npz, made by oracle/make_golden.py) -- no oracle code runs here.

Tolerances as in tests/test_gpu_parity.py for the float32-table kernels (models 0/1):
  |dU| <= cases.u_tolerance(N, U) = 2 (1e-6 sqrt(N) + 1e-9 |U|)  (twice SURVEY.md section 8c's)
          + 4 EPS32 cond_val   (the tau term's conditioning, stored with every point: cases.u_tolerance_cond;
                                < 2 % of the first term away from rho's bounds),
  |dgrad|_inf <= 5e-7 |grad|_inf + 1e-7 + 8 EPS32 cond_grad;
float64 paths: model 3 (neutral venue) |dU| <= 1e-11 |U|, |dgrad|_inf <= 1e-10 |grad|_inf; model 2 (dynamic,
atomics in arbitrary order) 1e-9 relative.
The points (SURVEY.md section 8c's list): z = 0, RandomState(7).uniform(-.5, .5), 8 further random points, the
UB-branch and rate-clip points, rho 1e-2 / 1e-4 / 1e-6 from each of its bounds; fixtures: the reference's
recipes, a ragged league and the N = 1e5 synthetic of section 8d.
"""
import glob
import os

import numpy as np
import pytest

import cases

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")


def _standardise(cov):
    return (cov - cov.mean(axis=0)) / cov.std(axis=0)


@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(GOLD, "m*.npz"))),
                         ids=lambda p: os.path.basename(p)[:-4])
def test_golden(hip_ctx, path):
    import torch

    d = np.load(path)
    model, T = int(d["model"]), int(d["n_teams"])
    if model == 2:
        _dynamic(hip_ctx, d)
        return
    cov = _standardise(d["covariates"]) if d["covariates"].size else None
    w = d["weights"].astype(np.float32) if d["weights"].size else None
    if model == 3:
        hip_ctx.set_fixtures_neutral(d["home_idx"], d["away_idx"], d["home_goals"], d["away_goals"],
                                     d["neutral"], T, weights=w, covariates_std=cov)
    else:
        hip_ctx.set_fixtures(model, d["home_idx"], d["away_idx"], d["home_goals"], d["away_goals"], T,
                             weights=w, covariates_std=cov if model == 1 else None)
    worst = [0.0, 0.0]
    for i in range(d["z"].shape[0]):
        U, g, aux = hip_ctx.logp_grad(torch.tensor(d["z"][i], dtype=torch.float64, device=hip_ctx.device))
        U, g, aux = float(U.cpu()[0]), g.cpu().numpy(), aux.cpu().numpy()[0]
        # (where extremal rates tie -- z = 0, rates clipped at 15 -- the reference's jnp.min / jnp.max split the
        # bounds' adjoint evenly over the tied fixtures, d["grad"]; the product keeps one arg-extremal pair,
        # another element of the same subdifferential: d["grad_first_pair"], identical wherever nothing ties.
        # U, rho and the bounds do not depend on the rule.)
        Uo, go = float(d["U"][i]), d["grad_first_pair"][i] if model != 3 else d["grad"][i]
        if not np.isfinite(Uo):
            assert not np.isfinite(U) or U > 1e300
            continue
        if model == 3:
            tolU, tolg = 1e-11 * abs(Uo), 1e-10 * np.abs(go).max()
        else:
            auxo = {"cond_val": float(d["cond_val"][i]), "cond_grad": float(d["cond_grad"][i])}
            tolU = cases.u_tolerance_cond(d["home_idx"].size, Uo, auxo)
            tolg = cases.g_tolerance_cond(go, auxo)
        worst = [max(worst[0], abs(U - Uo) / tolU), max(worst[1], np.abs(g - go).max() / tolg)]
        assert abs(U - Uo) <= tolU, (i, U, Uo)
        assert np.abs(g - go).max() <= tolg, i
        assert abs(aux[0] - d["rho"][i]) <= 1e-6
        assert abs(aux[1] - d["LB"][i]) <= 1e-6 and abs(aux[2] - d["UB"][i]) <= 1e-6
    print(f"{os.path.basename(path)}: {d['z'].shape[0]} points, worst |dU| / gate {worst[0]:.2f}, "
          f"|dgrad| / gate {worst[1]:.2f}")


def _dynamic(ctx, d):
    import torch

    cov = _standardise(d["covariates"])
    for rw in (1, 0):
        ctx.set_fixtures_dynamic(d["home_idx"], d["away_idx"], d["home_goals"], d["away_goals"], d["gameweek"],
                                 d["neutral"], int(d["n_teams"]), int(d["n_gameweeks"]), covariates_std=cov,
                                 random_walk=bool(rw))
        for i in range(d["z"].shape[0]):
            U, g, aux = ctx.logp_grad(torch.tensor(d["z"][i], dtype=torch.float64, device=ctx.device))
            U, g, aux = float(U.cpu()[0]), g.cpu().numpy(), aux.cpu().numpy()[0]
            Uo, go = float(d[f"U_rw{rw}"][i]), d[f"grad_rw{rw}"][i]
            assert abs(U - Uo) <= 1e-9 * abs(Uo), (rw, i, U, Uo)
            assert np.abs(g - go).max() <= 1e-9 * np.abs(go).max()
            assert abs(aux[0] - d[f"rho_rw{rw}"][i]) <= 1e-12
