# the BASELINE.json configurations (1-GPU ones), one line each: evaluation latency and in-situ NUTS
import sys, os, time, itertools
ROOT = os.environ.get('GRAFT_REPO_ROOT', '/root/repo'); sys.path[:0] = [ROOT + '/bpl-next_amd', ROOT]
import numpy as np, torch
from bench import synthetic_league
from bpl._ffi import HipContext, MODEL_BASIC, MODEL_EXTENDED, default_nuts_cfg

def timed(c, D, k=64):
    z = torch.tensor(np.random.RandomState(7).uniform(-.5, .5, (64, D)), dtype=torch.float64, device=c.device)
    U = torch.zeros(64, dtype=torch.float64, device=c.device); g = torch.zeros_like(z)
    c.logp_grad_graph(k, z, U, g, replays=2); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); c.logp_grad_graph(k, z, U, g, replays=8); e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / (8 * k)

def insitu(c, warm=150, samp=100):
    cfg = default_nuts_cfg(); cfg.num_warmup, cfg.num_samples = warm, samp
    d, st = c.nuts_run(cfg, (0, 42))
    return st['total_leapfrogs'] / st['wall_seconds']

c = HipContext(0)
T = 20
def std(cov): return (cov - cov.mean(0)) / cov.std(0)
cov5 = std(np.random.RandomState(0).normal(size=(T, 5)))
rows = []
# C1: 380 matches (the reference's dummy_data size)
perms = list(itertools.permutations(range(T), 2)); rs = np.random.RandomState(42)
h = np.array([p[0] for p in perms]); a = np.array([p[1] for p in perms])
c.set_fixtures(MODEL_BASIC, h, a, rs.poisson(2.1, 380), rs.poisson(1.7, 380), T)
rows.append(("C1 basic, 380 matches", 380, 6, timed(c, c.dim), insitu(c, 500, 500)))
# C2: 1e5 basic
h, a, x, y = synthetic_league(100_000, T)
c.set_fixtures(MODEL_BASIC, h, a, x, y, T)
rows.append(("C2 basic, 1e5 fixtures", 100_000, 6, timed(c, c.dim), insitu(c)))
# C3: extended, 5 covariates, 1e6 (+ time-weighted line)
h, a, x, y = synthetic_league(1_000_000, T)
c.set_fixtures(MODEL_EXTENDED, h, a, x, y, T, covariates_std=cov5)
rows.append(("C3 extended, 5 covariates, 1e6", 1_000_000, 6, timed(c, c.dim), insitu(c)))
w = np.exp(-1.0 * np.linspace(5, 0, h.size)).astype(np.float32)
c.set_fixtures(MODEL_EXTENDED, h, a, x, y, T, weights=w, covariates_std=cov5)
rows.append(("C3w extended, covariates + time weights, 1e6", 1_000_000, 10, timed(c, c.dim), insitu(c)))
# C5 (per GPU): basic 1e6
c.set_fixtures(MODEL_BASIC, h, a, x, y, T)
rows.append(("C5 basic, 1e6 (per GPU)", 1_000_000, 6, timed(c, c.dim), insitu(c)))
for name, n, b, t, ls in rows:
    print(f"{name:46s} {t:7.2f} us/eval {1e6 / t:10.0f} evals/s  algorithmic {n * b / t / 1e3:8.1f} GB/s  in-situ {ls:8.0f} leapfrogs/s", flush=True)
