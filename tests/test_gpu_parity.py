"""GPU parity: the HIP path (through the C-ABI) vs the float64 oracle on the same seeded
inputs.  Floating point path -> tolerances, stated here:

  U     : |U_hip - U_oracle| <= cases.u_tolerance(N, U) = 2 (1e-6 sqrt(N) + 1e-9 |U|): twice SURVEY.md
          section 8c's tolerance, for EVERY point incl. the rate-clip ones; measured <= 0.61 of it
          (profiles/r03/parity_errors.txt).  Round 2's gate, 3e-7 (|U| + 4N) / sqrt(P) (x20 at the clip
          points), was 0.12 at N = 1e6 against 9e-3 now.
  gradU : max|dg| <= 5e-7 * max|g| + 1e-7   (float32 tables: measured <= 2.3e-7 relative; round 2: 3e-6)
Measured errors are printed with -s; profiles/r03/parity_errors.txt holds a full run.
"""
import numpy as np
import pytest

import cases
import dc_oracle as O

pytestmark = pytest.mark.gpu


def _run(ctx, model, fx, zs):
    import torch

    w32 = None if fx.weights is None else fx.weights.astype(np.float32)
    cov = None
    if model == O.MODEL_EXTENDED and fx.covariates is not None:
        cov = O.standardise_covariates(fx.covariates)
    ctx.set_fixtures(model, fx.home_idx.astype(np.uint16), fx.away_idx.astype(np.uint16),
                     fx.home_goals.astype(np.uint8), fx.away_goals.astype(np.uint8),
                     fx.n_teams, weights=w32, covariates_std=cov)
    z = torch.tensor(np.stack(zs), dtype=torch.float64, device=ctx.device)
    outs = []
    for i in range(z.shape[0]):
        U, g, aux = ctx.logp_grad(z[i].contiguous())
        outs.append((U.cpu().numpy()[0], g.cpu().numpy(), aux.cpu().numpy()[0]))
    ctx.set_option("vec_min_chains", 0)      # batched = grid.y copies of the single launch
    Ub, gb, auxb = ctx.logp_grad(z)
    ctx.set_option("vec_min_chains", 1)      # chain-vectorised kernel (dc_vec.hip.h)
    Uv, gv, auxv = ctx.logp_grad(z)
    ctx.set_option("vec_min_chains", 12)
    return outs, (Ub.cpu().numpy(), gb.cpu().numpy(), auxb.cpu().numpy()), \
        (Uv.cpu().numpy(), gv.cpu().numpy(), auxv.cpu().numpy())


def _tolU(model, fx, auxo, Uo):
    return cases.u_tolerance(fx.n, Uo)


def _check(model, fx, name, z, U, g, aux, cond=False):
    # (ties="first_pair": the product's choice where extremal rates tie exactly -- identical to the
    # reference's even split everywhere else; see dc_oracle.likelihood_and_adjoint)
    Uo, go, auxo = O.potential_and_grad(model, fx, z, ties="first_pair")
    tolU = _tolU(model, fx, auxo, Uo)
    gerr = np.abs(g - go).max()
    gtol = 5e-7 * np.abs(go).max() + 1e-7
    if cond:  # near a bound of rho: the gates follow the tau term's conditioning (cases.u_tolerance_cond)
        tolU, gtol = cases.u_tolerance_cond(fx.n, Uo, auxo), cases.g_tolerance_cond(go, auxo)
    print(f"{name:28s} N={fx.n:8d} U={Uo:.6f} dU={U - Uo:+.3e} (tol {tolU:.1e}, {abs(U - Uo) / tolU:.2f}) "
          f"dg={gerr:.3e} (tol {gtol:.1e}, {gerr / gtol:.2f}) rho={auxo['rho']:+.6f}")
    if not np.isfinite(Uo):
        assert not np.isfinite(U) or U > 1e300
        return
    assert abs(U - Uo) <= tolU
    assert gerr <= gtol
    assert abs(aux[0] - auxo["rho"]) <= 1e-6
    assert abs(aux[1] - auxo["LB"]) <= 1e-6 and abs(aux[2] - auxo["UB"]) <= 1e-6


CASES = [
    (O.MODEL_BASIC, "dummy"),
    (O.MODEL_BASIC, "timed"),
    (O.MODEL_BASIC, "ragged_1"),
    (O.MODEL_BASIC, "ragged_777"),
    (O.MODEL_BASIC, "ragged_5000"),
    (O.MODEL_BASIC, "league_1e5"),
    (O.MODEL_EXTENDED, "dummy"),
    (O.MODEL_EXTENDED, "dummy_cov"),
    (O.MODEL_EXTENDED, "dummy_w"),
    (O.MODEL_EXTENDED, "timed_w"),
    (O.MODEL_EXTENDED, "ragged_5000"),
    (O.MODEL_EXTENDED, "league_1e5"),
    (O.MODEL_EXTENDED, "leaguew_3e4"),
    # more than 64 teams: the general (LDS-resident) tail epilogue, host-side NUTS tree
    (O.MODEL_BASIC, "wide_4000_100"),
    (O.MODEL_EXTENDED, "wide_4000_100"),
    (O.MODEL_BASIC, "wide_30000_700"),
    # covariate counts around the epilogue's one-pass coefficient sums (K <= 8: lane = coefficient x team residue;
    # more: four wave sums per round), at 20 and at 64 teams
    (O.MODEL_EXTENDED, "dummy_covk_1"),
    (O.MODEL_EXTENDED, "dummy_covk_8"),
    (O.MODEL_EXTENDED, "dummy_covk_9"),
    (O.MODEL_EXTENDED, "ragged64cov_3"),
    (O.MODEL_EXTENDED, "ragged64cov_8"),
    # ... with time weights and covariates (the lanes' exact rate products past 64 teams, weighted and clipped forms)
    (O.MODEL_EXTENDED, "widewc_20000_80"),
    (O.MODEL_EXTENDED, "widewc_60000_150"),
]


@pytest.mark.parametrize("model,name", CASES)
def test_logp_grad_matches_oracle(hip_ctx, model, name):
    fx = cases.fixtures(name)
    pts = cases.z_points(model, fx)
    outs, (Ub, gb, auxb), (Uv, gv, auxv) = _run(hip_ctx, model, fx, [p[1] for p in pts])
    for i, ((pname, z), (U, g, aux)) in enumerate(zip(pts, outs)):
        _check(model, fx, f"{name}/{pname}", z, U, g, aux)
        # batched launch == single launches, bitwise
        assert U == Ub[i] and np.array_equal(g, gb[i]) and np.array_equal(aux, auxb[i])
        # chain-vectorised kernel: same tolerances against the oracle
        _check(model, fx, f"{name}/{pname} [vec]", z, Uv[i], gv[i], auxv[i])


@pytest.mark.parametrize("model,name", [(O.MODEL_BASIC, "dummy"), (O.MODEL_BASIC, "league_1e5"),
                                        (O.MODEL_EXTENDED, "dummy_cov"), (O.MODEL_EXTENDED, "league_1e5"),
                                        (O.MODEL_EXTENDED, "leaguew_3e4"), (O.MODEL_BASIC, "wide_4000_100")])
def test_near_bound_points(hip_ctx, model, name):
    """rho 1e-2 / 1e-4 / 1e-6 from its upper bound (both branches: UB = 1/M and UB = 1) and from its lower
    bound (cases.near_bound_points; SURVEY.md section 8c: "one with rho within 1e-6 of a bound"): there a tau
    argument of bpl/_util.py:58-85 is ~the distance, tol = 0.  The float32 stream leaves such classes out
    and the tail workgroup works them out in float64 from the exact rates (dc_kernels.hip.h: class_terms,
    ill_core); the gates are the ordinary ones plus the conditioning of what float32 still sees
    (cases.u_tolerance_cond / g_tolerance_cond).  Single launches, the grid.y batch and the chain-vectorised
    kernel (whose prior workgroups file the float64 part for a tail launch of its own)."""
    fx = cases.fixtures(name)
    pts = cases.near_bound_points(model, fx)
    outs, (Ub, gb, auxb), (Uv, gv, auxv) = _run(hip_ctx, model, fx, [p[1] for p in pts])
    for i, ((pname, z), (U, g, aux)) in enumerate(zip(pts, outs)):
        _check(model, fx, f"{name}/{pname}", z, U, g, aux, cond=True)
        assert U == Ub[i] and np.array_equal(g, gb[i]) and np.array_equal(aux, auxb[i])
        _check(model, fx, f"{name}/{pname} [vec]", z, Uv[i], gv[i], auxv[i], cond=True)


@pytest.mark.parametrize("model,name", [(O.MODEL_BASIC, "wide_4000_100"), (O.MODEL_EXTENDED, "widewc_20000_80"),
                                        (O.MODEL_BASIC, "league_1e5")])
def test_fixture_layouts_agree(hip_ctx, model, name):
    """Option pair_order: the fixtures in (home, away) order (0) and along the Z-order curve over (home, away)
    (1; the default past 64 teams) are the same evaluation -- same rho, bounds and arg-extremal pairs (the pair
    table keeps its order: the tie rule), U and the gradient equal up to the order of the float32 run sums --
    and both agree with the oracle.  (league_1e5: 20 teams, where the default is (home, away) order.)"""
    import torch

    fx = cases.fixtures(name)
    pts = cases.z_points(model, fx)
    cov = None
    if model == O.MODEL_EXTENDED and fx.covariates is not None:
        cov = (fx.covariates - fx.covariates.mean(axis=0)) / fx.covariates.std(axis=0)
    w32 = None if fx.weights is None else fx.weights.astype(np.float32)
    outs = {}
    try:
        for order in (0, 1):
            hip_ctx.set_option("pair_order", order)
            hip_ctx.set_fixtures(model, fx.home_idx.astype(np.uint16), fx.away_idx.astype(np.uint16),
                                 fx.home_goals.astype(np.uint8), fx.away_goals.astype(np.uint8), fx.n_teams,
                                 weights=w32, covariates_std=cov)
            outs[order] = [tuple(t.cpu().numpy().copy() for t in
                                 hip_ctx.logp_grad(torch.tensor(z, dtype=torch.float64, device=hip_ctx.device)))
                           for _, z in pts]
    finally:
        hip_ctx.set_option("pair_order", -1)
    for i, (pname, z) in enumerate(pts):
        (U0, g0, a0), (U1, g1, a1) = outs[0][i], outs[1][i]
        assert np.array_equal(a0, a1), "rho, LB, UB, q"
        if np.isfinite(U0[0]):
            tol = cases.u_tolerance(fx.n, float(U0[0]))
            assert abs(U1[0] - U0[0]) <= 0.2 * tol, (pname, U0[0], U1[0])
            assert np.abs(g1 - g0).max() <= 0.5 * (5e-7 * np.abs(g0).max() + 1e-7)
        _check(model, fx, f"{name}/{pname} [(home, away)]", z, float(U0[0]), g0.reshape(-1), a0[0])
        _check(model, fx, f"{name}/{pname} [Z-order]", z, float(U1[0]), g1.reshape(-1), a1[0])


def test_full_size_1e6(hip_ctx):
    """BASELINE size (N = 1e6): oracle comparison at two points for both models."""
    h, a, x, y = O.synthetic_league(1_000_000)
    fx = O.Fixtures(h, a, x, y, 20)
    for model in (O.MODEL_BASIC, O.MODEL_EXTENDED):
        pts = cases.z_points(model, fx, n_random=1)[:2]
        outs, _, (Uv, gv, auxv) = _run(hip_ctx, model, fx, [p[1] for p in pts])
        for i, ((pname, z), (U, g, aux)) in enumerate(zip(pts, outs)):
            _check(model, fx, f"league_1e6/{pname}", z, U, g, aux)
            _check(model, fx, f"league_1e6/{pname} [vec]", z, Uv[i], gv[i], auxv[i])


@pytest.mark.parametrize("model,name", [(O.MODEL_BASIC, "dummy"), (O.MODEL_BASIC, "league_1e5"),
                                        (O.MODEL_EXTENDED, "dummy_cov"), (O.MODEL_EXTENDED, "leaguew_3e4")])
def test_wild_region_matches_oracle(hip_ctx, model, name):
    """init_to_uniform(radius=2) starts a chain at z ~ U(-2, 2)^D: rates up to e^30 (basic model;
    the extended model clips them at 15), potentials of 1e15 and more.  The accumulator rows
    carry such sums in their hi word (dc_kernels.hip.h, GA_ROW); the float32 tables limit the
    agreement with the float64 oracle to ~1e-6 relative there."""
    import torch

    fx = cases.fixtures(name)
    D = O.latent_dim(model, fx.n_teams, fx.k if model == O.MODEL_EXTENDED else 0)
    zs = [np.random.RandomState(40 + i).uniform(-2.0, 2.0, D) for i in range(6)]
    outs, (Ub, gb, _), (Uv, gv, _) = _run(hip_ctx, model, fx, zs)
    big = 0.0
    for i, (z, (U, g, aux)) in enumerate(zip(zs, outs)):
        Uo, go, auxo = O.potential_and_grad(model, fx, z)
        big = max(big, abs(Uo))
        print(f"{name}/wild{i}: U={Uo:.6e} dU/U={(U - Uo) / abs(Uo):+.2e} "
              f"dg/|g|={np.abs(g - go).max() / np.abs(go).max():.2e} rho={auxo['rho']:+.4f}")
        assert np.isfinite(Uo) and np.isfinite(U)
        assert abs(U - Uo) <= 5e-6 * abs(Uo)
        assert np.abs(g - go).max() <= 2e-5 * np.abs(go).max()
        # (exact fixed-point sums up to 2^53 units of 2^-30 per row, i.e. 8e6; beyond, float64
        # rounding makes the grid.y batch's other partition differ in the last bits)
        assert abs(U - Ub[i]) <= 1e-14 * abs(U) and np.abs(g - gb[i]).max() <= 1e-13 * np.abs(g).max()
        assert abs(Uv[i] - Uo) <= 5e-6 * abs(Uo)
    if model == O.MODEL_BASIC:
        assert big > 1e6 * fx.n / 380.0  # beyond the accumulator rows' lo word (1.6e7 per row)


@pytest.mark.parametrize("weighted", [False, True])
def test_config3_extended_covariates_1e6(hip_ctx, weighted):
    """BASELINE.json configs[2] exactly as stated: extended model, 5 covariates
    (`RandomState(0).normal((20, 5))`, standardised as bpl/extended_dixon_coles.py:124-127),
    N = 1e6 fixtures; and the time-weighted line beside it (`time_diff = linspace(5, 0, N)`,
    epsilon = 1, bpl/extended_dixon_coles.py:202-215).  Every latent point of cases.z_points
    (three uniform, the UB branch, the rate clip at 15), single launch / grid.y batch /
    chain-vectorised kernel, against the float64 oracle."""
    h, a, x, y = O.synthetic_league(1_000_000)
    fx = O.Fixtures(h, a, x, y, 20)
    fx.covariates = np.random.RandomState(0).normal(size=(20, 5))
    if weighted:
        fx.weights = cases.float32_weights(np.linspace(5, 0, fx.n), 1.0)
    pts = cases.z_points(O.MODEL_EXTENDED, fx)
    assert pts[0][1].size == 77  # D = 3T + 2K + 7
    outs, (Ub, gb, auxb), (Uv, gv, auxv) = _run(hip_ctx, O.MODEL_EXTENDED, fx, [p[1] for p in pts])
    tag = "c3w_1e6" if weighted else "c3_1e6"
    for i, ((pname, z), (U, g, aux)) in enumerate(zip(pts, outs)):
        _check(O.MODEL_EXTENDED, fx, f"{tag}/{pname}", z, U, g, aux)
        assert U == Ub[i] and np.array_equal(g, gb[i]) and np.array_equal(aux, auxb[i])
        _check(O.MODEL_EXTENDED, fx, f"{tag}/{pname} [vec]", z, Uv[i], gv[i], auxv[i])


@pytest.mark.parametrize("name,model,chains", [("league_1e5", O.MODEL_BASIC, 19),
                                               ("leaguew_3e4", O.MODEL_EXTENDED, 9),
                                               ("ragged_777", O.MODEL_EXTENDED, 8)])
def test_chain_vectorised_matches_single(hip_ctx, name, model, chains):
    """dc_vec (8 chains per workgroup, partial last group) vs the single-chain launch on the
    same points: same arithmetic per run, different grouping of float32 partial sums."""
    import torch

    fx = cases.fixtures(name)
    D = O.latent_dim(model, fx.n_teams, 0 if fx.covariates is None else fx.covariates.shape[1])
    zs = [np.random.RandomState(100 + i).uniform(-0.5, 0.5, D) for i in range(chains)]
    outs, _, (Uv, gv, auxv) = _run(hip_ctx, model, fx, zs)
    for i, (U, g, aux) in enumerate(outs):
        print(f"chain {i}: U={U:.6f} dU={Uv[i] - U:+.3e} dg={np.abs(gv[i] - g).max():.3e} "
              f"|g|={np.abs(g).max():.3e}")
        Uo, _, auxo = O.potential_and_grad(model, fx, zs[i])
        assert abs(Uv[i] - U) <= _tolU(model, fx, auxo, Uo)
        assert np.abs(gv[i] - g).max() <= 1e-6 * np.abs(g).max()
        assert np.array_equal(auxv[i], aux)
    # deterministic
    z = torch.tensor(np.stack(zs), dtype=torch.float64, device=hip_ctx.device)
    hip_ctx.set_option("vec_min_chains", 1)
    U1, g1, _ = hip_ctx.logp_grad(z)
    U2, g2, _ = hip_ctx.logp_grad(z)
    hip_ctx.set_option("vec_min_chains", 12)
    assert torch.equal(U1, U2) and torch.equal(g1, g2)


def test_order_invariance_and_determinism(hip_ctx):
    """Size-independent properties at N = 1e6: (i) a launch repeated is bitwise equal;
    (ii) permuting the fixtures leaves U and gradU unchanged (the library re-sorts);
    (iii) additivity: U(A u B) - U(A) - U(B) + U(empty-likelihood part) is consistent,
    checked through duplicated data: the likelihood part doubles."""
    import torch

    h, a, x, y = O.synthetic_league(1_000_000)
    z = np.random.RandomState(3).uniform(-0.3, 0.3, 45)
    zt = torch.tensor(z, dtype=torch.float64, device=hip_ctx.device)

    def run(hh, aa, xx, yy):
        hip_ctx.set_fixtures(O.MODEL_BASIC, hh, aa, xx, yy, 20)
        U1, g1, _ = hip_ctx.logp_grad(zt)
        U2, g2, _ = hip_ctx.logp_grad(zt)
        assert torch.equal(U1, U2) and torch.equal(g1, g2)
        return U1.cpu().numpy()[0], g1.cpu().numpy()

    U0, g0 = run(h, a, x, y)
    perm = np.random.RandomState(1).permutation(h.size)
    Up, gp = run(h[perm], a[perm], x[perm], y[perm])
    assert abs(Up - U0) <= 1e-8 * abs(U0)
    assert np.abs(gp - g0).max() <= 1e-7 * np.abs(g0).max()

    # priors-only part: likelihood of zero fixtures is not expressible (n >= 1), so use
    # U(2x data) - U(data) = U(data) - U_prior  ->  U_prior = 2 U(data) - U(2x data)
    U2x, _ = run(np.tile(h, 2), np.tile(a, 2), np.tile(x, 2), np.tile(y, 2))
    fx_small = O.Fixtures(h[:1], a[:1], x[:1], y[:1], 20)
    # oracle prior part = U_small - (likelihood of the single fixture): compute directly
    sl = O.site_slices(O.MODEL_BASIC, 20)
    U_small, _, _ = O.potential_and_grad(O.MODEL_BASIC, fx_small, z)
    fx2 = O.Fixtures(np.tile(h[:1], 2), np.tile(a[:1], 2), np.tile(x[:1], 2), np.tile(y[:1], 2), 20)
    U_small2, _, _ = O.potential_and_grad(O.MODEL_BASIC, fx2, z)
    prior_oracle = 2 * U_small - U_small2
    prior_hip = 2 * U0 - U2x
    assert abs(prior_hip - prior_oracle) <= 2e-8 * abs(U0)


def test_graph_replay_matches_direct(hip_ctx):
    import torch

    fx = cases.fixtures("league_1e5")
    hip_ctx.set_fixtures(O.MODEL_BASIC, fx.home_idx.astype(np.uint16), fx.away_idx.astype(np.uint16),
                         fx.home_goals.astype(np.uint8), fx.away_goals.astype(np.uint8), 20)
    zs = np.random.RandomState(7).uniform(-0.5, 0.5, (8, 45))
    z = torch.tensor(zs, dtype=torch.float64, device=hip_ctx.device)
    hip_ctx.set_option("vec_min_chains", 0)  # direct = 8 single-chain launches (grid.y)
    Ud, gd, _ = hip_ctx.logp_grad(z)
    hip_ctx.set_option("vec_min_chains", 12)
    U = torch.zeros(8, dtype=torch.float64, device=hip_ctx.device)
    g = torch.zeros_like(z)
    hip_ctx.logp_grad_graph(16, z, U, g, replays=3)
    torch.cuda.synchronize()
    assert torch.equal(U, Ud) and torch.equal(g, gd)


def test_graph_survives_slab_growth(hip_ctx):
    """A cached hipGraph holds the hand-off slab / ticket pointers in its kernel arguments.  A
    batched call with more chains than the slabs were sized for reallocates them, so the cached
    graphs must be dropped and re-captured: graph, 8-chain batch, graph again == direct path."""
    import torch

    fx = cases.fixtures("league_1e5")
    hip_ctx.set_fixtures(O.MODEL_BASIC, fx.home_idx.astype(np.uint16), fx.away_idx.astype(np.uint16),
                         fx.home_goals.astype(np.uint8), fx.away_goals.astype(np.uint8), 20)
    z = torch.tensor(np.random.RandomState(11).uniform(-0.5, 0.5, (8, 45)), dtype=torch.float64,
                     device=hip_ctx.device)
    direct = [hip_ctx.logp_grad(z[i].contiguous()) for i in range(8)]
    Ud = torch.cat([d[0] for d in direct])
    gd = torch.stack([d[1] for d in direct])
    U = torch.zeros(8, dtype=torch.float64, device=hip_ctx.device)
    g = torch.zeros_like(z)
    hip_ctx.logp_grad_graph(8, z, U, g, replays=2)
    torch.cuda.synchronize()
    assert torch.equal(U, Ud) and torch.equal(g, gd)
    hip_ctx.set_option("vec_min_chains", 0)
    try:
        Ub, gb, _ = hip_ctx.logp_grad(z)  # 8 chains as grid.y copies: the slabs grow
    finally:
        hip_ctx.set_option("vec_min_chains", 12)
    assert torch.equal(Ub, Ud) and torch.equal(gb, gd)
    U.zero_()
    g.zero_()
    hip_ctx.logp_grad_graph(8, z, U, g, replays=2)  # same key as before the growth
    torch.cuda.synchronize()
    assert torch.equal(U, Ud) and torch.equal(g, gd)


def test_nonfinite_is_not_an_error(hip_ctx):
    """tol=0 in the reference (bpl/_util.py:42): rho on a bound gives log(0) = -inf ->
    U = +inf is returned, not raised (numpyro treats it as a divergence)."""
    import torch

    fx = cases.fixtures("dummy")
    hip_ctx.set_fixtures(O.MODEL_BASIC, fx.home_idx.astype(np.uint16), fx.away_idx.astype(np.uint16),
                         fx.home_goals.astype(np.uint8), fx.away_goals.astype(np.uint8), 20)
    z = np.random.RandomState(7).uniform(-0.5, 0.5, 45)
    z[20] = 40.0  # corr_coef_raw -> q clipped to 1-eps: rho ~ UB, 1 - rho*lh*la ~ 0 at argmax
    z[41] = 1.0   # make M > 1 so UB = 1/M
    Uo, go, auxo = O.potential_and_grad(O.MODEL_BASIC, fx, z)
    U, g, _ = hip_ctx.logp_grad(torch.tensor(z, dtype=torch.float64, device=hip_ctx.device))
    U, g = U.cpu().numpy()[0], g.cpu().numpy()
    print("U oracle", Uo, "U hip", U, "tau_min", auxo["tau_min"])
    # The oracle (float64) stays finite here: q is clipped at 1 - eps, so rho = UB (1 - 1.2e-7) + ...
    # and 1 - rho*lh*la ~ 3e-7 at the arg-max pair.  Round 3 evaluated that difference in float32 -- a
    # handful of ulps of 1: +inf or a value within 2e-2 |U| of the oracle, and the test said so.  Round 4:
    # tau arguments below 1/64 are worked out in float64 by the tail workgroup (dc_kernels.hip.h ill_core),
    # so the kernel is held to the same derived gate as everywhere (cases.u_tolerance_cond).
    assert np.isfinite(Uo) and np.isfinite(U)
    assert abs(U - Uo) <= cases.u_tolerance_cond(fx.n, Uo, auxo)
    assert np.abs(g - go).max() <= cases.g_tolerance_cond(go, auxo)
    # a wilder point of the same kind: mean_defence = -8 makes every rate ~e^8, M >> 1, UB = 1/M
    z2 = z.copy()
    z2[42] = -8.0
    Uo2, go2, auxo2 = O.potential_and_grad(O.MODEL_BASIC, fx, z2)
    U2, g2, _ = hip_ctx.logp_grad(torch.tensor(z2, dtype=torch.float64, device=hip_ctx.device))
    U2 = U2.cpu().numpy()[0]
    print("U oracle", Uo2, "U hip", U2, "tau_min", auxo2["tau_min"])
    assert np.isfinite(Uo2) and np.isfinite(U2)
    assert abs(U2 - Uo2) <= 5e-6 * abs(Uo2)     # (rates of e^8: the wild-region gate, test_wild_region_matches_oracle)
    assert np.abs(g2.cpu().numpy() - go2).max() <= 2e-5 * np.abs(go2).max()
    # beyond the bound (only reachable by a rho that the clip of q rules out -- but a NaN-free -inf is
    # what tol = 0 asks for): checked on the CPU oracle, tests/test_oracle.py
    zn = z.copy()
    zn[0] = np.nan
    U, g, _ = hip_ctx.logp_grad(torch.tensor(zn, dtype=torch.float64, device=hip_ctx.device))
    assert np.isnan(U.cpu().numpy()[0])


def test_bad_arguments_raise(hip_ctx):
    from bpl._ffi import BplHipError

    with pytest.raises(BplHipError):
        hip_ctx.set_fixtures(O.MODEL_BASIC, np.array([0, 5], np.uint16), np.array([1, 0], np.uint16),
                             np.array([1, 1], np.uint8), np.array([0, 0], np.uint8), 3)  # index 5 >= T
    with pytest.raises(BplHipError):
        hip_ctx.set_fixtures(O.MODEL_BASIC, np.array([0], np.uint16), np.array([1], np.uint16),
                             np.array([1], np.uint8), np.array([0], np.uint8), 2,
                             weights=np.ones(1, np.float32))  # basic model takes no weights


def test_reserved_scoreline_and_lane_padding(hip_ctx):
    """Pair runs are padded to the lane width with null fixtures (goals 255-255, weight 0).  A REAL
    255-255 fixture is legal all the same (the reference accepts any goals): run lengths 1..17
    (every padding amount), with extreme scorelines among them, agree with the oracle."""
    import torch

    rs = np.random.RandomState(5)
    hh, aa = [], []
    for k, (p, q) in enumerate([(i, j) for i in range(5) for j in range(5) if i != j][:17]):
        hh += [p] * (k + 1)
        aa += [q] * (k + 1)
    hh, aa = np.array(hh), np.array(aa)
    x, y = rs.poisson(1.5, hh.size), rs.poisson(1.2, hh.size)
    x[3], y[3] = 255, 0      # extremes, one of them the padding's own scoreline
    x[40], y[40] = 255, 255
    x[41], y[41] = 0, 255
    fx = O.Fixtures(hh, aa, x, y, 5)
    for model in (O.MODEL_BASIC, O.MODEL_EXTENDED):
        hip_ctx.set_fixtures(model, hh.astype(np.uint16), aa.astype(np.uint16), x.astype(np.uint8),
                             y.astype(np.uint8), 5)
        z = np.random.RandomState(9).uniform(-0.4, 0.4, hip_ctx.dim)
        U, g, aux = hip_ctx.logp_grad(torch.tensor(z, dtype=torch.float64, device=hip_ctx.device))
        _check(model, fx, "lane_padding/u", z, float(U.cpu()[0]), g.cpu().numpy(), aux.cpu().numpy()[0])


@pytest.mark.parametrize("model", [O.MODEL_BASIC, O.MODEL_EXTENDED])
@pytest.mark.parametrize("teams,n", [(20, 100_000), (70, 100_000), (91, 150_000), (100, 200_000), (200, 400_000),
                                     (300, 200_000)])   # (300 teams: more than 256 -- a second round in the top-two jobs)
def test_separable_bounds_equal_pair_walk(hip_ctx, model, teams, n):
    """A league in which every ordered pair has met takes the O(teams) bounds (top two table
    entries per role instead of a walk over all pairs, dc_kernels.hip.h dense_maxima_f32): same rho,
    same bounds, same arg-extremal pairs -- hence the same U and gradient -- as the pair walk
    (option dense_pairs = 0), at ordinary points and where the extended model's rate clip binds
    (there the separable path must hand over to the walk by itself); and both agree with the oracle.
    (Taken from 4096 pairs on, i.e. past 64 teams: 20 teams walk their 380 pairs either way.)"""
    import torch

    h, a, x, y = O.synthetic_league(n, teams)
    fx = O.Fixtures(h, a, x, y, teams)
    assert len(set(zip(h.tolist(), a.tolist()))) == teams * (teams - 1)
    D = O.latent_dim(model, teams, 0)
    zs = [np.random.RandomState(70 + i).uniform(-s, s, D) for i, s in enumerate((0.3, 0.8, 2.0, 2.0))]
    outs = {}
    for dense in (1, 0):
        hip_ctx.set_option("dense_pairs", dense)
        hip_ctx.set_fixtures(model, h.astype(np.uint16), a.astype(np.uint16), x.astype(np.uint8), y.astype(np.uint8), teams)
        outs[dense] = [tuple(t.cpu().numpy().copy() for t in hip_ctx.logp_grad(torch.tensor(z, dtype=torch.float64, device=hip_ctx.device)))
                       for z in zs]
    hip_ctx.set_option("dense_pairs", 1)
    for i, z in enumerate(zs):
        (U1, g1, a1), (U0, g0, a0) = outs[1][i], outs[0][i]
        print(f"T={teams} model={model} point {i}: U={U1[0]:.6e} dU={U1[0] - U0[0]:+.2e} dg={np.abs(g1 - g0).max():.2e} "
              f"rho {a1[0][0]:+.6f} / {a0[0][0]:+.6f}")
        # (rho, LB, UB, q: the maximum of the rate product may come out one ulp apart -- the walk takes
        # the largest ROUNDED product, the separable path the pair with the largest exact one)
        assert np.abs(a1 - a0).max() <= 4e-16 * np.abs(a0).max(), "rho, LB, UB, q"
        assert abs(U1[0] - U0[0]) <= 1e-9 * abs(U0[0]) and np.abs(g1 - g0).max() <= 1e-9 * np.abs(g0).max()
        if i < 2 and n <= 200_000:
            _check(model, fx, f"dense T={teams}/{i}", z, float(U1[0]), g1.reshape(-1), a1[0])


def test_nuts_with_many_teams(hip_ctx):
    """T > 64: the leaf does not fit dc_eval's tail, so the chain still lives on the device but
    books its leaves in separate launches -- one wave per chain up to 256 latent entries
    (T = 100: D = 205), GW workgroups per chain beyond (T = 700: D = 1405).  With a fixed step
    size both must build the same trees as the host tree engine (same threefry streams)."""
    from bpl._ffi import BPLHIP_EUNSUPPORTED, BplHipError, default_nuts_cfg

    for name in ("wide_4000_100", "wide_30000_700"):
        fx = cases.fixtures(name)
        hip_ctx.set_fixtures(O.MODEL_BASIC, fx.home_idx.astype(np.uint16), fx.away_idx.astype(np.uint16),
                             fx.home_goals.astype(np.uint8), fx.away_goals.astype(np.uint8), fx.n_teams)
        cfg = default_nuts_cfg()
        cfg.num_warmup, cfg.num_samples, cfg.step_size = 0, 6, 0.002
        z0 = np.random.RandomState(5).uniform(-0.1, 0.1, hip_ctx.dim)
        hip_ctx.set_option("device_nuts", 0)
        try:
            d0, s0 = hip_ctx.nuts_run(cfg, (0, 3), z0)
        finally:
            hip_ctx.set_option("device_nuts", 1)
        d1, s1 = hip_ctx.nuts_run(cfg, (0, 3), z0)
        assert s0["total_leapfrogs"] > 20
        assert s1["num_steps"].tolist() == s0["num_steps"].tolist()
        assert np.abs(d1[:3] - d0[:3]).max() < 1e-9 and np.abs(d1 - d0).max() < 1e-4
        assert np.abs(s1["accept_prob"][:3] - s0["accept_prob"][:3]).max() < 1e-8
        # with adaptation, and several chains on the device at once
        cfg.num_warmup, cfg.num_samples, cfg.step_size = 30, 20, 1.0
        d, st = hip_ctx.nuts_run(cfg, (0, 3))
        assert d.shape == (20, hip_ctx.dim) and np.isfinite(d).all() and st["total_leapfrogs"] > 50
        res = hip_ctx.nuts_run_chains(cfg, [(0, 3), (0, 4)])
        assert np.array_equal(res[0][0], d), "a chain does not depend on its neighbours"
        assert np.isfinite(res[1][0]).all()
    # persistent_nuts = 0: lock-step chains exist only for the models of dc_eval's tail
    hip_ctx.set_option("persistent_nuts", 0)
    try:
        with pytest.raises(BplHipError) as e:
            hip_ctx.nuts_run_chains(cfg, [(0, 1), (0, 2)])
        assert e.value.code == BPLHIP_EUNSUPPORTED
    finally:
        hip_ctx.set_option("persistent_nuts", 1)


def test_launch_partitions_agree(hip_ctx):
    """dc_eval has two launch partitions for short streams (4 or 8 waves of a workgroup own tiles;
    the launch picks by its workgroup count) and the "active_waves" option pins one: every choice
    must give the same potential and gradient up to summation order, for one chain and for several
    chains in one launch (which switch to the 8-wave partition on their own)."""
    import torch

    fx = cases.fixtures("league_200000")
    z = torch.tensor(np.random.RandomState(3).uniform(-0.4, 0.4, (6, 45)), dtype=torch.float64,
                     device=hip_ctx.device)
    ref = None
    try:
        for aw in (0, 8, 4, 2, 1):
            hip_ctx.set_option("active_waves", aw)
            hip_ctx.set_fixtures(O.MODEL_BASIC, fx.home_idx.astype(np.uint16), fx.away_idx.astype(np.uint16),
                                 fx.home_goals.astype(np.uint8), fx.away_goals.astype(np.uint8), fx.n_teams)
            U1, g1, _ = hip_ctx.logp_grad(z[0].contiguous())                      # one chain
            U = torch.zeros(6, dtype=torch.float64, device=hip_ctx.device)
            g = torch.zeros_like(z)
            aux = torch.zeros((6, 4), dtype=torch.float64, device=hip_ctx.device)
            hip_ctx.set_option("vec_min_chains", 0)                               # grid.y copies
            hip_ctx.logp_grad(z, U, g, aux)
            hip_ctx.set_option("vec_min_chains", 12)
            got = (U1.cpu().numpy()[0], g1.cpu().numpy(), U.cpu().numpy(), g.cpu().numpy())
            assert abs(got[2][0] - got[0]) <= 1e-12 * abs(got[0])
            if ref is None:
                ref = got
                continue
            assert abs(got[0] - ref[0]) <= 1e-12 * abs(ref[0])
            assert np.abs(got[1] - ref[1]).max() <= 1e-10 * np.abs(ref[1]).max()
            assert np.abs(got[2] - ref[2]).max() <= 1e-12 * np.abs(ref[2]).max()
            assert np.abs(got[3] - ref[3]).max() <= 1e-10 * np.abs(ref[3]).max()
    finally:
        hip_ctx.set_option("active_waves", 0)
        hip_ctx.set_option("vec_min_chains", 12)


def test_hand_off_timeout_is_reported_not_silent(hip_ctx):
    """A kernel whose bounded wait expires raises the context's fault word (host memory mapped into
    the device, dc::raise_fault); the next entry point turns it into BPLHIP_EHIP with a message and
    puts the hand-off state back, instead of NaN outputs that a sampler books as divergences.  The
    wait itself cannot be made to expire on a healthy GPU, so the word is raised through the test hook
    (option debug_raise_fault) exactly as the kernels raise it."""
    import torch
    from bpl._ffi import BPLHIP_EHIP, BplHipError, default_nuts_cfg

    fx = cases.fixtures("league_1e5")
    hip_ctx.set_fixtures(O.MODEL_BASIC, fx.home_idx.astype(np.uint16), fx.away_idx.astype(np.uint16),
                         fx.home_goals.astype(np.uint8), fx.away_goals.astype(np.uint8), 20)
    z = torch.tensor(np.random.RandomState(7).uniform(-0.5, 0.5, 45), dtype=torch.float64, device=hip_ctx.device)
    U0, g0, _ = hip_ctx.logp_grad(z)
    for code, word in ((1, "dc_eval arrivals"), (6, "dc_eval_loop")):
        hip_ctx.set_option("debug_raise_fault", code)
        with pytest.raises(BplHipError) as e:
            hip_ctx.logp_grad(z)
        assert e.value.code == BPLHIP_EHIP and "timed out" in str(e.value) and word in str(e.value)
        U1, g1, _ = hip_ctx.logp_grad(z)  # sticky only until reported; the hand-off state was reset
        assert torch.equal(U0, U1) and torch.equal(g0, g1)
    # raised while a sampler runs: the run stops with the error instead of returning a posterior
    cfg = default_nuts_cfg()
    cfg.num_warmup, cfg.num_samples = 20, 20
    hip_ctx.set_option("debug_raise_fault", 2)
    with pytest.raises(BplHipError) as e:
        hip_ctx.nuts_run(cfg, (0, 1))
    assert e.value.code == BPLHIP_EHIP
    d, st = hip_ctx.nuts_run(cfg, (0, 1))
    assert np.isfinite(d).all()
