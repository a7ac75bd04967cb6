"""GPU: soak of the in-launch hand-offs.  The evaluations have fixed summation orders (exact
fixed-point accumulator rows in dc_eval; host-built gather schedule in the neutral kernel), so a
repeat of the same z must reproduce U and the gradient BIT FOR BIT -- a rare race in the arrival
counters / write-through hand-off / polled granules would show up as a mismatch.  A short version of
tools/soak.py (8.5 M evaluations there) that runs with every `-m gpu`; the dynamic model's float64
atomics land in arbitrary order, so there the repeats must agree to rounding."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _league(n, teams=20):
    from bench import synthetic_league

    return synthetic_league(n, teams)


@pytest.mark.parametrize("name,n,extended,k,weighted,teams", [
    ("basic 1e6", 1_000_000, False, 0, False, 20),
    ("basic 3800", 3_800, False, 0, False, 20),
    ("extended 1e6, 5 covariates, weighted", 1_000_000, True, 5, True, 20),
    ("basic 2e5, 100 teams (separable bounds, general tail)", 200_000, False, 0, False, 100),
])
def test_repeats_are_bit_identical(hip_ctx, name, n, extended, k, weighted, teams):
    import torch
    from bpl._ffi import MODEL_BASIC, MODEL_EXTENDED

    h, a, x, y = _league(n, teams)
    cov = None
    if k:
        cov = np.random.RandomState(0).normal(size=(teams, k))
        cov = (cov - cov.mean(0)) / cov.std(0)
    w = np.exp(-np.linspace(5.0, 0.0, n)).astype(np.float32) if weighted else None
    hip_ctx.set_fixtures(MODEL_EXTENDED if extended else MODEL_BASIC, h, a, x, y, teams, weights=w, covariates_std=cov)
    z = torch.tensor(np.random.RandomState(7).uniform(-0.5, 0.5, (64, hip_ctx.dim)), dtype=torch.float64,
                     device=hip_ctx.device)
    U = torch.zeros(64, dtype=torch.float64, device=hip_ctx.device)
    g = torch.zeros_like(z)
    hip_ctx.logp_grad_graph(64, z, U, g, replays=1)
    torch.cuda.synchronize()
    U0, g0 = U.clone(), g.clone()
    assert torch.isfinite(U0).all()
    for _ in range(6):  # 6 x 25 replays x 64 = 9600 evaluations per round of checks
        U.zero_()
        g.zero_()
        hip_ctx.logp_grad_graph(64, z, U, g, replays=25)
        torch.cuda.synchronize()
        assert torch.equal(U, U0) and torch.equal(g, g0), name


def test_persistent_kernel_chain_repeats(hip_ctx):
    """The same chain twice inside the persistent evaluation kernel (granule hand-off, launch-specific
    tags): identical draws, and identical to the chain run with one launch per leapfrog."""
    from bpl._ffi import MODEL_BASIC, default_nuts_cfg

    h, a, x, y = _league(100_000)
    hip_ctx.set_fixtures(MODEL_BASIC, h, a, x, y, 20)
    cfg = default_nuts_cfg()
    cfg.num_warmup, cfg.num_samples = 60, 40
    d1, s1 = hip_ctx.nuts_run(cfg, (0, 11))
    d2, s2 = hip_ctx.nuts_run(cfg, (0, 11))
    assert np.array_equal(d1, d2) and s1["total_leapfrogs"] == s2["total_leapfrogs"] > 500
    hip_ctx.set_option("persistent_kernel", 0)
    try:
        d3, s3 = hip_ctx.nuts_run(cfg, (0, 11))
    finally:
        hip_ctx.set_option("persistent_kernel", 1)
    assert s3["total_leapfrogs"] == s1["total_leapfrogs"] and np.array_equal(d1, d3)


def test_single_launch_models_repeat(hip_ctx):
    """Neutral-venue kernel: bit-identical repeats (fixed gather order).  Dynamic kernel (data-flagged
    cell records, two grid barriers, self-clearing scratch): thousands of back-to-back launches agree
    to rounding and leave the scratch clean for the next one."""
    import torch

    rs = np.random.RandomState(11)
    N, T = 570, 20
    h = rs.randint(0, T, N)
    a = (h + 1 + rs.randint(0, T - 1, N)) % T
    hip_ctx.set_fixtures_neutral(h, a, rs.poisson(1.4, N), rs.poisson(1.1, N), rs.randint(0, 2, N), T,
                                 weights=rs.uniform(0.2, 3.0, N).astype(np.float32))
    z = torch.tensor(np.random.RandomState(7).uniform(-0.3, 0.3, (8, hip_ctx.dim)), dtype=torch.float64,
                     device=hip_ctx.device)
    U = torch.zeros(8, dtype=torch.float64, device=hip_ctx.device)
    g = torch.zeros_like(z)
    hip_ctx.logp_grad_graph(16, z, U, g, replays=1)
    torch.cuda.synchronize()
    U0, g0 = U.clone(), g.clone()
    for _ in range(4):
        hip_ctx.logp_grad_graph(16, z, U, g, replays=100)
        torch.cuda.synchronize()
        assert torch.equal(U, U0) and torch.equal(g, g0)

    Tn, G = 100, 50
    rs = np.random.RandomState(4)
    hh, aa, gw = [], [], []
    for gk in range(G):
        p = rs.permutation(Tn)
        hh += list(p[0::2]); aa += list(p[1::2]); gw += [gk] * (Tn // 2)
    n = len(hh)
    hip_ctx.set_fixtures_dynamic(np.array(hh), np.array(aa), rs.poisson(1.5, n), rs.poisson(1.2, n), np.array(gw),
                                 np.zeros(n, np.uint8), Tn, G)
    z = torch.tensor(np.random.RandomState(7).uniform(-0.3, 0.3, (8, hip_ctx.dim)), dtype=torch.float64,
                     device=hip_ctx.device)
    U = torch.zeros(8, dtype=torch.float64, device=hip_ctx.device)
    g = torch.zeros_like(z)
    hip_ctx.logp_grad_graph(16, z, U, g, replays=1)
    torch.cuda.synchronize()
    U0, g0 = U.clone(), g.clone()
    assert torch.isfinite(U0).all()
    for _ in range(4):
        hip_ctx.logp_grad_graph(16, z, U, g, replays=60)
        torch.cuda.synchronize()
        assert ((U - U0).abs() <= 1e-12 * U0.abs()).all()
        assert float((g - g0).abs().max()) <= 1e-11 * float(g0.abs().max())
