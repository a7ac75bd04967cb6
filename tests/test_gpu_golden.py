"""GPU: the HIP path (through the C-ABI) against the committed golden vectors alone
(tests/golden/*.npz, made by oracle/make_golden.py) -- no oracle code runs here.

Tolerances as in tests/test_gpu_parity.py for the float32-table kernels (models 0/1):
  |dU| <= cases.u_tolerance(N, U) = 2 (1e-6 sqrt(N) + 1e-9 |U|)  (twice SURVEY.md section 8c's),
  |dgrad|_inf <= 5e-7 |grad|_inf + 1e-7;
float64 path (model 3, neutral venue): |dU| <= 1e-11 |U|, |dgrad|_inf <= 1e-10 |grad|_inf.
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
    cov = _standardise(d["covariates"]) if d["covariates"].size else None
    w = d["weights"].astype(np.float32) if d["weights"].size else None
    if model == 3:
        hip_ctx.set_fixtures_neutral(d["home_idx"], d["away_idx"], d["home_goals"], d["away_goals"],
                                     d["neutral"], T, weights=w, covariates_std=cov)
    else:
        hip_ctx.set_fixtures(model, d["home_idx"], d["away_idx"], d["home_goals"], d["away_goals"], T,
                             weights=w, covariates_std=cov if model == 1 else None)
    for i in range(d["z"].shape[0]):
        U, g, aux = hip_ctx.logp_grad(torch.tensor(d["z"][i], dtype=torch.float64, device=hip_ctx.device))
        U, g, aux = float(U.cpu()[0]), g.cpu().numpy(), aux.cpu().numpy()[0]
        Uo, go = float(d["U"][i]), d["grad"][i]
        if not np.isfinite(Uo):
            assert not np.isfinite(U) or U > 1e300
            continue
        if model == 3:
            tolU, tolg = 1e-11 * abs(Uo), 1e-10 * np.abs(go).max()
        else:
            tolU = cases.u_tolerance(d["home_idx"].size, Uo)
            tolg = 5e-7 * np.abs(go).max() + 1e-7
        assert abs(U - Uo) <= tolU, (i, U, Uo)
        assert np.abs(g - go).max() <= tolg
        assert abs(aux[0] - d["rho"][i]) <= 1e-6
        assert abs(aux[1] - d["LB"][i]) <= 1e-6 and abs(aux[2] - d["UB"][i]) <= 1e-6
