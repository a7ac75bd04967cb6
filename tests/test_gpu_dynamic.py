"""GPU parity of the dynamic (time-varying) model path: HIP (float64 kernels,
dc_dynamic.hip.h) vs the float64 oracle, BASELINE config 4 (T=100, G=50, N=2500, D=35502)
and small ragged cases; plus a short NUTS fit through the Python class.
Tolerance (float64 arithmetic, atomics in arbitrary order): 1e-9 relative."""
import numpy as np
import pytest

import dc_dynamic_oracle as DO

pytestmark = pytest.mark.gpu


def _eval(ctx, fx, z, random_walk=True, fused=1, big_wgs=0):
    import torch

    ctx.set_option("fused_small", fused)  # 1: one launch with grid barriers (small leagues); 0: four launches
    ctx.set_option("dyn_big_wgs", big_wgs)  # > 0: the single launch in its sliced form (dyn_fused<true>)
    cov = None if fx.covariates is None else DO.standardise_covariates(fx.covariates)
    ctx.set_fixtures_dynamic(fx.home_idx, fx.away_idx, fx.home_goals, fx.away_goals, fx.gameweek,
                             fx.neutral, fx.n_teams, fx.n_gameweeks, covariates_std=cov,
                             random_walk=random_walk)
    assert ctx.dim == DO.latent_dim(fx.n_gameweeks, fx.n_teams, fx.k)
    zt = torch.tensor(z, dtype=torch.float64, device=ctx.device)
    U, g, aux = ctx.logp_grad(zt)
    U2, g2, _ = ctx.logp_grad(zt)
    assert abs(float(U2[0]) - float(U[0])) <= 1e-12 * abs(float(U[0]))
    ctx.set_option("fused_small", 1)
    ctx.set_option("dyn_big_wgs", 0)
    return float(U.cpu()[0]), g.cpu().numpy(), aux.cpu().numpy()[0]


@pytest.mark.parametrize("fused", [1, 0])
@pytest.mark.parametrize("random_walk", [True, False])
@pytest.mark.parametrize("case", ["small", "small_cov", "config4"])
def test_dynamic_logp_grad_matches_oracle(hip_ctx, case, random_walk, fused):
    fx = {"small": lambda: DO.small_recipe(), "small_cov": lambda: DO.small_recipe(k=3),
          "config4": DO.config4_recipe}[case]()
    D = DO.latent_dim(fx.n_gameweeks, fx.n_teams, fx.k)
    sl = DO.site_slices(fx.n_gameweeks, fx.n_teams, fx.k)
    for seed in (7, 2):
        z = np.random.RandomState(seed).uniform(-0.3, 0.3, D)
        if seed == 2:
            z[sl["mean_home_attack"]] = 1.2
        Uo, go, auxo = DO.potential_and_grad(fx, z, random_walk)
        U, g, aux = _eval(hip_ctx, fx, z, random_walk, fused)
        print(f"{case} rw={random_walk} fused={fused} seed={seed} D={D} U={Uo:.6f} dU={U - Uo:+.2e} "
              f"dg={np.abs(g - go).max():.2e} / {np.abs(go).max():.2e}")
        assert abs(U - Uo) <= 1e-9 * abs(Uo)
        assert np.abs(g - go).max() <= 1e-9 * np.abs(go).max()
        assert abs(aux[0] - auxo["rho"]) <= 1e-12


@pytest.mark.parametrize("big_wgs", [1, 37, 300])
@pytest.mark.parametrize("case", ["small_cov", "config4", "ragged"])
def test_dynamic_sliced_single_launch_matches_oracle(hip_ctx, case, big_wgs):
    """dyn_fused<true> (a gameweek's slice of the fixtures per workgroup, cells / accumulators / rates in
    LDS; taken by default past 1024 fixtures per team workgroup) forced on small leagues: fewer
    workgroups than team workgroups, workgroups without fixtures, workgroups without teams; `ragged`:
    empty gameweeks, neutral venues, one gameweek with most of the fixtures."""
    if case == "ragged":
        rs = np.random.RandomState(12)
        T, G, n = 23, 9, 5000
        h = rs.randint(0, T, n)
        a = (h + 1 + rs.randint(0, T - 1, n)) % T
        gw = np.where(rs.rand(n) < 0.6, 4, rs.choice([0, 1, 3, 4, 6, 8], n))   # gameweeks 2, 5, 7 are empty
        fx = DO.DynFixtures(h, a, rs.poisson(1.5, n), rs.poisson(1.2, n), gw, (rs.rand(n) < 0.3).astype(int), T, G)
    else:
        fx = {"small_cov": lambda: DO.small_recipe(k=3), "config4": DO.config4_recipe}[case]()
    D = DO.latent_dim(fx.n_gameweeks, fx.n_teams, fx.k)
    for random_walk in (True, False):
        z = np.random.RandomState(7).uniform(-0.3, 0.3, D)
        Uo, go, auxo = DO.potential_and_grad(fx, z, random_walk)
        U, g, aux = _eval(hip_ctx, fx, z, random_walk, 1, big_wgs)
        assert abs(U - Uo) <= 1e-9 * abs(Uo)
        assert np.abs(g - go).max() <= 1e-9 * np.abs(go).max()
        assert abs(aux[0] - auxo["rho"]) <= 1e-12
        # and the four-launch path right after it on the same context (scratch handed over clean)
        U4, g4, _ = _eval(hip_ctx, fx, z, random_walk, 0)
        assert abs(U4 - Uo) <= 1e-9 * abs(Uo) and np.abs(g4 - go).max() <= 1e-9 * np.abs(go).max()


def test_dynamic_throughput_variant_1e6(hip_ctx):
    """SURVEY.md §8(d) throughput variant: uniformly random (gameweek, h != a), N = 1e6."""
    rs = np.random.RandomState(4)
    n, T, G = 1_000_000, 100, 50
    h = rs.randint(0, T, n)
    a = (h + 1 + rs.randint(0, T - 1, n)) % T
    fx = DO.DynFixtures(h, a, rs.poisson(1.5, n), rs.poisson(1.2, n), rs.randint(0, G, n),
                        np.zeros(n, int), T, G)
    z = np.random.RandomState(7).uniform(-0.2, 0.2, DO.latent_dim(G, T))
    Uo, go, _ = DO.potential_and_grad(fx, z)
    U, g, _ = _eval(hip_ctx, fx, z)
    assert abs(U - Uo) <= 1e-9 * abs(Uo)
    assert np.abs(g - go).max() <= 1e-9 * np.abs(go).max()


def test_dynamic_fit_smoke(hip_ctx):
    from bpl.dynamic_dixon_coles import DynamicNeutralDixonColesMatchPredictor

    rs = np.random.RandomState(0)
    T, G, per = 6, 4, 30
    teams = [f"t{i}" for i in range(T)]
    td = {"home_team": [], "away_team": [], "home_goals": [], "away_goals": [], "gameweek": [],
          "neutral_venue": []}
    for g in range(G):
        for _ in range(per):
            i, j = rs.choice(T, 2, replace=False)
            td["home_team"].append(teams[i])
            td["away_team"].append(teams[j])
            td["home_goals"].append(rs.poisson(1.6))
            td["away_goals"].append(rs.poisson(1.2))
            td["gameweek"].append(g)
            td["neutral_venue"].append(int(rs.rand() < 0.2))
    z0 = np.zeros(7 * G * T + 10 * G + 2)
    m = DynamicNeutralDixonColesMatchPredictor().fit(td, num_warmup=60, num_samples=40,
                                                     run_kwargs={"init_params": z0})
    assert m.attack.shape == (40, G, T) and m.home_attack.shape == (40, G, T)
    assert m.corr_coef.shape == (40,) and np.isfinite(m.attack).all()
    p = m.predict_outcome_proba(["t0", "t1"], ["t2", "t3"], [0, 1])
    assert np.allclose(p["home_win"] + p["draw"] + p["away_win"], 1.0, atol=1e-5)
    assert m.predict_score_proba("t0", "t1", 1, 0, 0).shape == (1,)
    # the predict methods run on the device (venue-aware kernels on ONE gameweek's tables):
    # against the float64 restatement, for the default (last) gameweek and an earlier one
    from fake_ctx import FakePredictCtx

    for week in (None, 1):
        g = G - 1 if week is None else week
        ref = FakePredictCtx()
        ref.predict_set_posterior_venue(*(getattr(m, nm)[:, g, :] for nm in m._VENUE_TABLES), m.corr_coef)
        h, a, nv = np.array([0, 1, 4]), np.array([2, 3, 5]), np.array([0, 1, 0])
        got = m.predict_score_proba(["t0", "t1", "t4"], ["t2", "t3", "t5"], [1, 0, 2], [0, 0, 1], nv, gameweek=week)
        assert np.abs(got - ref.predict_score_proba(h, a, [1, 0, 2], [0, 0, 1], nv)).max() < 1e-12
        out = m.predict_outcome_proba(["t0", "t1", "t4"], ["t2", "t3", "t5"], nv, gameweek=week)
        grid = ref.predict_score_grid(h, a, 15, nv)
        assert np.abs(out["draw"] - np.trace(grid, axis1=1, axis2=2)).max() < 3e-6
        lh, la = m._calculate_expected_goals(["t0"], ["t2"], [0], gameweek=week)
        assert lh.shape == (40, 1) and np.all(lh > 0)


def test_dynamic_chain_on_device_matches_host_tree(hip_ctx):
    """The dynamic model's latent vector (here D = 700+) is booked by the wide leaf launches
    (nuts_dev.hip.h kw_leaf) with the whole chain on the device; with a fixed step size it
    builds the same trees as the host tree engine on the same threefry streams."""
    from bpl._ffi import default_nuts_cfg

    T, G, n = 10, 8, 600
    rs = np.random.RandomState(3)
    h = rs.randint(0, T, n)
    a = (h + 1 + rs.randint(0, T - 1, n)) % T
    fx = DO.DynFixtures(h, a, rs.poisson(1.5, n), rs.poisson(1.2, n), np.sort(rs.randint(0, G, n)),
                        (rs.rand(n) < 0.2).astype(int), T, G)
    hip_ctx.set_fixtures_dynamic(fx.home_idx, fx.away_idx, fx.home_goals, fx.away_goals, fx.gameweek,
                                 fx.neutral, T, G)
    assert hip_ctx.dim == DO.latent_dim(G, T) > 256
    cfg = default_nuts_cfg()
    cfg.num_warmup, cfg.num_samples, cfg.step_size, cfg.max_tree_depth = 0, 6, 0.02, 6
    z0 = np.random.RandomState(5).uniform(-0.1, 0.1, hip_ctx.dim)
    hip_ctx.set_option("device_nuts", 0)
    try:
        d0, s0 = hip_ctx.nuts_run(cfg, (0, 9), z0)
    finally:
        hip_ctx.set_option("device_nuts", 1)
    d1, s1 = hip_ctx.nuts_run(cfg, (0, 9), z0)
    assert s0["total_leapfrogs"] > 30
    assert s1["num_steps"].tolist() == s0["num_steps"].tolist()
    assert np.abs(d1[:3] - d0[:3]).max() < 1e-9 and np.abs(d1 - d0).max() < 1e-4
    assert np.abs(s1["potential_energy"][:3] - s0["potential_energy"][:3]).max() < 1e-7
    # adaptation on the device (step size + Welford mass matrix with the workgroup as the team)
    cfg.num_warmup, cfg.num_samples, cfg.step_size = 120, 30, 1.0
    d, st = hip_ctx.nuts_run(cfg, (0, 9))
    assert np.isfinite(d).all() and st["total_divergences"] <= 2 and 0.5 < st["mean_accept_prob"] <= 1.0


def test_dynamic_wide_leaf_row_of_workgroups(hip_ctx):
    """A latent vector long enough for SEVERAL workgroups per chain in the wide leaf (kw_leaf: tagged
    partial records polled across the grid row; the whole row advances the chain behind row barriers):
    same trees and draws as the host tree engine, two chains side by side, adaptation included."""
    from bpl._ffi import default_nuts_cfg

    T, G, n = 40, 16, 1500
    rs = np.random.RandomState(13)
    h = rs.randint(0, T, n)
    a = (h + 1 + rs.randint(0, T - 1, n)) % T
    fx = DO.DynFixtures(h, a, rs.poisson(1.5, n), rs.poisson(1.2, n), np.sort(rs.randint(0, G, n)),
                        (rs.rand(n) < 0.1).astype(int), T, G)
    hip_ctx.set_fixtures_dynamic(fx.home_idx, fx.away_idx, fx.home_goals, fx.away_goals, fx.gameweek,
                                 fx.neutral, T, G)
    assert hip_ctx.dim > 2 * 2048   # (at least three workgroups of 1024 threads, two elements each)
    cfg = default_nuts_cfg()
    cfg.num_warmup, cfg.num_samples, cfg.step_size, cfg.max_tree_depth = 0, 5, 0.01, 5
    z0 = np.random.RandomState(5).uniform(-0.05, 0.05, hip_ctx.dim)
    hip_ctx.set_option("device_nuts", 0)
    try:
        d0, s0 = hip_ctx.nuts_run(cfg, (0, 11), z0)
    finally:
        hip_ctx.set_option("device_nuts", 1)
    d1, s1 = hip_ctx.nuts_run(cfg, (0, 11), z0)
    assert s0["total_leapfrogs"] > 30
    assert s1["num_steps"].tolist() == s0["num_steps"].tolist()
    assert np.abs(d1[:2] - d0[:2]).max() < 1e-9 and np.abs(d1 - d0).max() < 1e-4
    # two chains side by side (one grid row each), still without adaptation: each is the chain it is alone
    res = hip_ctx.nuts_run_chains(cfg, [(0, 11), (0, 12)], z0=np.stack([z0, z0]))
    assert np.abs(res[0][0] - d1).max() < 1e-6 and res[0][1]["num_steps"].tolist() == s1["num_steps"].tolist()
    # adaptation (dual averaging, Welford mass matrix, window ends) through the grid-row team
    cfg.num_warmup, cfg.num_samples, cfg.step_size, cfg.max_tree_depth = 80, 10, 1.0, 6
    d, st = hip_ctx.nuts_run(cfg, (0, 31))
    assert np.isfinite(d).all() and st["total_divergences"] <= 2 and 0.5 < st["mean_accept_prob"] <= 1.0


def test_dynamic_chains_together_equal_chains_alone(hip_ctx):
    """Several dynamic-model chains share the wide leaf launches (grid.y = chain); each must be
    the chain it would be alone: same seed -> the same draws, also with thinning and a depth cap
    that trees actually hit."""
    from bpl._ffi import default_nuts_cfg

    T, G, n = 12, 6, 500
    rs = np.random.RandomState(8)
    h = rs.randint(0, T, n)
    a = (h + 1 + rs.randint(0, T - 1, n)) % T
    fx = DO.DynFixtures(h, a, rs.poisson(1.4, n), rs.poisson(1.1, n), np.sort(rs.randint(0, G, n)),
                        np.zeros(n, int), T, G)
    hip_ctx.set_fixtures_dynamic(fx.home_idx, fx.away_idx, fx.home_goals, fx.away_goals, fx.gameweek,
                                 fx.neutral, T, G)
    cfg = default_nuts_cfg()
    cfg.num_warmup, cfg.num_samples, cfg.thinning, cfg.max_tree_depth = 40, 21, 2, 4
    seeds = [(0, 21), (0, 22), (0, 23)]
    together = hip_ctx.nuts_run_chains(cfg, seeds)
    for (d, st), sd in zip(together, seeds):
        d1, s1 = hip_ctx.nuts_run(cfg, sd)
        assert d.shape == (10, hip_ctx.dim)
        # (not bit for bit: this model's evaluation adds with float64 atomics in arbitrary order)
        assert np.abs(d[:3] - d1[:3]).max() < 1e-8 and np.abs(d - d1).max() < 1e-3
        assert st["num_steps"][:3].tolist() == s1["num_steps"][:3].tolist()
        assert st["num_steps"].max() == 2 ** 4 - 1  # the cap is reached
