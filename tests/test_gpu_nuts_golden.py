"""Device-resident chains (persistent chains: tree builder, transitions, adaptation in the
evaluation kernel's tail, nuts_dev.hip.h) against the trajectories of the INDEPENDENT NUTS
restatement (oracle/nuts_oracle.py -> tests/golden/nuts_*.npz).  The HIP potential differs from
the float64 one the goldens were made with by ~1e-9 relative (float32 tables), and every
transition amplifies a difference ~30x, so tree sizes are compared over the first transitions
and draws with graded tolerances."""
import os

import numpy as np
import pytest

import cases
import dc_oracle as O

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

ENGINES = {"persistent": (1, 1), "device": (1, 0), "host": (0, 0)}


def _bind(ctx, model, name):
    fx = cases.fixtures(name)
    cov = None if fx.covariates is None or model == O.MODEL_BASIC else O.standardise_covariates(fx.covariates)
    ctx.set_fixtures(model, fx.home_idx.astype(np.uint16), fx.away_idx.astype(np.uint16),
                     fx.home_goals.astype(np.uint8), fx.away_goals.astype(np.uint8), fx.n_teams,
                     covariates_std=cov)


def _run(ctx, engine, cfg, key, z0):
    dn, pn = ENGINES[engine]
    ctx.set_option("device_nuts", dn)
    ctx.set_option("persistent_nuts", pn)
    try:
        return ctx.nuts_run(cfg, key, z0)
    finally:
        ctx.set_option("device_nuts", 1)
        ctx.set_option("persistent_nuts", 1)


@pytest.mark.parametrize("gname,model,fix", [("nuts_dummy_basic_fixed", O.MODEL_BASIC, "dummy"),
                                             ("nuts_dummy_ext_fixed", O.MODEL_EXTENDED, "dummy_cov")])
@pytest.mark.parametrize("engine", list(ENGINES))
def test_fixed_step_chain_follows_the_oracle(hip_ctx, gname, model, fix, engine):
    from bpl._ffi import default_nuts_cfg

    g = np.load(os.path.join(GOLD, gname + ".npz"))
    _bind(hip_ctx, model, fix)
    cfg = default_nuts_cfg()
    cfg.num_warmup, cfg.num_samples, cfg.step_size = 0, int(g["num_samples"]), float(g["step_size0"])
    d, st = _run(hip_ctx, engine, cfg, tuple(int(k) for k in g["key"]), g["z0"])
    print(engine, "tree sizes", st["num_steps"].tolist(), "oracle", g["num_steps"].tolist(),
          "max |d draw| per transition", np.abs(d - g["draws"]).max(axis=1))
    assert st["num_steps"][:5].tolist() == g["num_steps"][:5].tolist()
    assert np.abs(d[:3] - g["draws"][:3]).max() < 1e-5
    assert np.abs(st["accept_prob"][:3] - g["accept_prob"][:3]).max() < 1e-5
    assert np.abs(st["potential_energy"][:3] - g["potential_energy"][:3]).max() < 1e-3
    assert np.allclose(st["step_size"], float(g["step_size0"]))


@pytest.mark.parametrize("gname,model,fix", [("nuts_dummy_basic_init", O.MODEL_BASIC, "dummy"),
                                             ("nuts_dummy_ext_init", O.MODEL_EXTENDED, "dummy_cov")])
def test_initial_point_follows_the_oracle(hip_ctx, gname, model, fix):
    """init_to_uniform(radius=2) + the retry loop on the device potential: the same point as the
    independent restatement's (the draws are float32-valued, so equality is exact up to the one
    1e-12 leapfrog used to read the point back)."""
    from bpl._ffi import default_nuts_cfg

    g = np.load(os.path.join(GOLD, gname + ".npz"))
    _bind(hip_ctx, model, fix)
    cfg = default_nuts_cfg()
    cfg.num_warmup, cfg.num_samples, cfg.step_size = 0, 1, 1e-12
    d, st = hip_ctx.nuts_run(cfg, tuple(int(k) for k in g["key"]), None)
    assert np.abs(d[0] - g["z0"]).max() < 1e-7


def test_adapted_chain_follows_the_oracle(hip_ctx):
    """Warm-up adaptation on the device -- dual averaging, the Welford window with its mass matrix
    update and dual-averaging restart (t = 19 of 22), the final averaging of the step size --
    against the independent restatement.  Trees are capped at 3 leapfrogs (max_tree_depth = 2) so
    that the float32-table potential tracks the float64 one through all 22 adapted transitions."""
    from bpl._ffi import default_nuts_cfg

    g = np.load(os.path.join(GOLD, "nuts_dummy_basic_adapt_shallow.npz"))
    _bind(hip_ctx, O.MODEL_BASIC, "dummy")
    cfg = default_nuts_cfg()
    w = int(g["num_warmup"])
    cfg.num_warmup, cfg.num_samples, cfg.step_size = w, int(g["num_samples"]), 1.0
    cfg.max_tree_depth = int(g["max_tree_depth"])
    for engine in ENGINES:
        d, st = _run(hip_ctx, engine, cfg, tuple(int(k) for k in g["key"]), g["z0"])
        print(engine, st["num_steps"].tolist(), g["num_steps"][w:].tolist(), st["final_step_size"],
              float(g["final_step_size"]), st["total_leapfrogs"], int(g["num_steps"].sum()),
              np.abs(d - g["draws"]).max())
        assert st["total_leapfrogs"] == int(g["num_steps"].sum())
        assert st["num_steps"].tolist() == g["num_steps"][w:].tolist()
        assert abs(st["final_step_size"] / float(g["final_step_size"]) - 1) < 1e-4
        assert np.abs(st["inverse_mass_matrix"] / g["inverse_mass_matrix"] - 1).max() < 1e-4
        assert np.abs(d - g["draws"]).max() < 1e-4
