# Soak test of the hand-off protocol: the evaluation is deterministic (fixed-order reductions), so
# every repeat of the same z must reproduce U and the gradient bit for bit.  A rare race in the
# tickets / sc1 hand-off would show up as a mismatch.  python tools/soak.py [replays]
import sys, os
ROOT = os.environ.get('GRAFT_REPO_ROOT', '/root/repo'); sys.path[:0] = [ROOT + '/bpl-next_amd', ROOT]
import numpy as np, torch
from bench import synthetic_league
from bpl._ffi import HipContext, MODEL_BASIC, MODEL_EXTENDED, default_nuts_cfg

REPLAYS = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
c = HipContext(0)
for name, n, model, k, weighted, T in (("basic 1e6", 1_000_000, MODEL_BASIC, 0, False, 20),
                                       ("basic 3800", 3_800, MODEL_BASIC, 0, False, 20),
                                       ("extended 1e6 K=5", 1_000_000, MODEL_EXTENDED, 5, False, 20),
                                       ("extended 1e6 K=5 weighted", 1_000_000, MODEL_EXTENDED, 5, True, 20),
                                       ("basic 1e7", 10_000_000, MODEL_BASIC, 0, False, 20),
                                       # past 64 teams (round 4: rows polled beside the record, no barrier behind
                                       # separable bounds, fused top-two jobs; an incomplete pair table: the walk)
                                       ("basic 2e5, 100 teams", 200_000, MODEL_BASIC, 0, False, 100),
                                       ("basic 4e5, 200 teams", 400_000, MODEL_BASIC, 0, False, 200),
                                       ("basic 5e4, 300 teams", 50_000, MODEL_BASIC, 0, False, 300),
                                       ("extended 3e5, 150 teams K=3 w", 300_000, MODEL_EXTENDED, 3, True, 150)):
    h, a, x, y = synthetic_league(n, T)
    cov = None
    if k:
        cov = np.random.RandomState(0).normal(size=(T, k)); cov = (cov - cov.mean(0)) / cov.std(0)
    w = np.exp(-np.linspace(5.0, 0.0, n)).astype(np.float32) if weighted else None
    c.set_fixtures(model, h, a, x, y, T, weights=w, covariates_std=cov)
    D = c.dim
    z = torch.tensor(np.random.RandomState(7).uniform(-.5, .5, (64, D)), dtype=torch.float64, device=c.device)
    U = torch.zeros(64, dtype=torch.float64, device=c.device); g = torch.zeros_like(z)
    c.logp_grad_graph(64, z, U, g, replays=1); torch.cuda.synchronize()
    U0, g0 = U.clone(), g.clone()
    bad = 0; done = 0
    reps = REPLAYS if n <= 1_000_000 else max(REPLAYS // 10, 10)
    for chunk in range(0, reps, 50):
        r = min(50, reps - chunk)
        U.zero_(); g.zero_()
        c.logp_grad_graph(64, z, U, g, replays=r); torch.cuda.synchronize()
        done += 64 * r
        if not (torch.equal(U, U0) and torch.equal(g, g0)): bad += 1
    print(f"{name:28s}: {done:8d} evaluations, mismatching batches {bad}", flush=True)
    assert bad == 0
# chain-vectorised kernel (64 chains per launch) and grid.y copies (8 chains)
h, a, x, y = synthetic_league(1_000_000, 20)
c.set_fixtures(MODEL_BASIC, h, a, x, y, 20)
for C in (64, 8):
    z = torch.tensor(np.random.RandomState(9).uniform(-.5, .5, (C, c.dim)), dtype=torch.float64, device=c.device)
    U = torch.zeros(C, dtype=torch.float64, device=c.device); g = torch.zeros_like(z)
    aux = torch.zeros((C, 4), dtype=torch.float64, device=c.device)
    c.logp_grad(z, U, g, aux); torch.cuda.synchronize(); U0, g0 = U.clone(), g.clone(); bad = 0
    n_rep = max(REPLAYS // 4, 10)
    for _ in range(n_rep):
        U.zero_(); g.zero_(); c.logp_grad(z, U, g, aux); torch.cuda.synchronize()
        bad += not (torch.equal(U, U0) and torch.equal(g, g0))
    print(f"batched launch, {C:2d} chains      : {n_rep * C:8d} evaluations, mismatching launches {bad}", flush=True)
    assert bad == 0
# NUTS: two runs of the same chains must give the same draws (persistent chains, 8 at once)
h, a, x, y = synthetic_league(100_000, 20)
c.set_fixtures(MODEL_BASIC, h, a, x, y, 20)
cfg = default_nuts_cfg(); cfg.num_warmup, cfg.num_samples = 150, 50
seeds = [(0, 100 + i) for i in range(8)]
r1 = c.nuts_run_chains(cfg, seeds); r2 = c.nuts_run_chains(cfg, seeds)
same = all(np.array_equal(p[0], q[0]) for p, q in zip(r1, r2))
print(f"8 persistent chains twice: identical draws = {same}; leapfrogs {sum(p[1]['total_leapfrogs'] for p in r1)}")
assert same
# ... and through the chain-vectorised kernel (more chains than gridy_max_chains: the chunk hipGraph of dc_vec launches)
seeds = [(0, 300 + i) for i in range(40)]
cfg.num_warmup, cfg.num_samples = 100, 30
r1 = c.nuts_run_chains(cfg, seeds); r2 = c.nuts_run_chains(cfg, seeds)
same = all(np.array_equal(p[0], q[0]) for p, q in zip(r1, r2))
print(f"40 chains through dc_vec twice: identical draws = {same}; leapfrogs {sum(p[1]['total_leapfrogs'] for p in r1)}")
assert same
print("soak ok")
