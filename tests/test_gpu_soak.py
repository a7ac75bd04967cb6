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
    # (round 4's tail past 64 teams: rows polled beside the record, two per thread at 200 teams; no barrier behind
    # separable bounds; the pair walk of an incomplete table; exact rate products with weights and covariates)
    ("basic 4e5, 200 teams", 400_000, False, 0, False, 200),
    ("basic 5e4, 300 teams (incomplete pair table)", 50_000, False, 0, False, 300),
    ("extended 3e5, 150 teams, 3 covariates, weighted", 300_000, True, 3, True, 150),
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


@pytest.mark.parametrize("extended", [False, True])
def test_leaf_one_step_behind_is_bitwise_neutral(hip_ctx, extended):
    """Inside the persistent kernel the next position goes out from the epilogue and the leaf is booked one
    step behind its evaluation, from forwarded registers (option persist_spec, default 1; dc_eval_loop).
    Same draws, bit for bit, as with the leaf booked first and its state re-read from memory (0) -- through
    warm-up (trees that end by U-turn: the dropped evaluation), adaptation and the launch boundaries of a
    chain of several thousand leapfrogs; 45 and 77 latent entries (one and two elements per leaf lane)."""
    from bpl._ffi import MODEL_BASIC, MODEL_EXTENDED, default_nuts_cfg

    h, a, x, y = _league(100_000)
    cov = None
    if extended:
        cov = np.random.RandomState(0).normal(size=(20, 5))
        cov = (cov - cov.mean(0)) / cov.std(0)
    hip_ctx.set_fixtures(MODEL_EXTENDED if extended else MODEL_BASIC, h, a, x, y, 20, covariates_std=cov)
    cfg = default_nuts_cfg()
    cfg.num_warmup, cfg.num_samples = 120, 60
    runs = {}
    try:
        for spec in (1, 0):
            hip_ctx.set_option("persist_spec", spec)
            runs[spec] = hip_ctx.nuts_run(cfg, (0, 23))
    finally:
        hip_ctx.set_option("persist_spec", 1)
    (d1, s1), (d0, s0) = runs[1], runs[0]
    assert s1["total_leapfrogs"] == s0["total_leapfrogs"] > 3000
    assert np.array_equal(d1, d0)
    assert np.array_equal(s1["potential_energy"], s0["potential_energy"])
    assert np.array_equal(s1["num_steps"], s0["num_steps"])


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


def test_sliced_single_launches_repeat(hip_ctx):
    """The single launches sliced over all CUs (dyn_fused<true>, dcn::neu_big: tree barriers, a counted cell
    hand-off, copies of the scratch words, scratch and counters cleared by the launch's last workgroup):
    thousands of back-to-back launches agree to rounding (float64 atomics land in arbitrary order), raise no
    fault and leave scratch and counters clean for the next one -- which the four-launch path, run right after
    on the same context, confirms from the other side."""
    import torch

    def soak(rounds, replays, tol_u, tol_g):
        z = torch.tensor(np.random.RandomState(7).uniform(-0.3, 0.3, (8, hip_ctx.dim)), dtype=torch.float64,
                         device=hip_ctx.device)
        U = torch.zeros(8, dtype=torch.float64, device=hip_ctx.device)
        g = torch.zeros_like(z)
        hip_ctx.logp_grad_graph(16, z, U, g, replays=1)
        torch.cuda.synchronize()
        U0, g0 = U.clone(), g.clone()
        assert torch.isfinite(U0).all() and torch.isfinite(g0).all()
        for _ in range(rounds):
            U.zero_()
            g.zero_()
            hip_ctx.logp_grad_graph(16, z, U, g, replays=replays)
            torch.cuda.synchronize()
            assert ((U - U0).abs() <= tol_u * U0.abs()).all()
            assert float((g - g0).abs().max()) <= tol_g * float(g0.abs().max())
        hip_ctx.set_option("fused_small", 0)   # the same points through the multi-launch path
        try:
            U4, g4, _ = hip_ctx.logp_grad(z)
        finally:
            hip_ctx.set_option("fused_small", 1)
        assert ((U4 - U0).abs() <= 1e-9 * U0.abs()).all()
        assert float((g4 - g0).abs().max()) <= 1e-9 * float(g0.abs().max())

    rs = np.random.RandomState(5)
    Tn, G, n = 100, 50, 300_000
    h = rs.randint(0, Tn, n)
    a = (h + 1 + rs.randint(0, Tn - 1, n)) % Tn
    hip_ctx.set_fixtures_dynamic(h, a, rs.poisson(1.5, n), rs.poisson(1.2, n), rs.randint(0, G, n),
                                 (rs.rand(n) < 0.1).astype(np.uint8), Tn, G)
    soak(4, 40, 1e-12, 1e-11)    # 4 x 40 x 16 = 2560 launches of dyn_fused<true>

    T, n = 20, 300_000
    h = rs.randint(0, T, n)
    a = (h + 1 + rs.randint(0, T - 1, n)) % T
    hip_ctx.set_fixtures_neutral(h, a, rs.poisson(1.4, n), rs.poisson(1.1, n), rs.randint(0, 2, n), T,
                                 weights=rs.uniform(0.2, 3.0, n).astype(np.float32))
    soak(4, 40, 1e-12, 1e-11)    # 2560 launches of neu_big
