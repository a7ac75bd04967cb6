# N-sweep of the single-chain evaluation: where does the stream become HBM bound?
# usage: python tools/n_sweep.py [N ...]   (default 1e5 1e6 1e7 1e8)
import sys, os
ROOT = os.environ.get('GRAFT_REPO_ROOT', '/root/repo'); sys.path[:0] = [ROOT + '/bpl-next_amd', ROOT]
import numpy as np, torch
from bench import synthetic_league
from bpl._ffi import HipContext, MODEL_BASIC

Ns = [int(float(v)) for v in sys.argv[1:]] or [100_000, 1_000_000, 10_000_000, 100_000_000]
c = HipContext(0)
h0, a0, x0, y0 = synthetic_league(10_000_000, 20)
for N in Ns:
    reps = (N + h0.size - 1) // h0.size
    h, a, x, y = (np.tile(v, reps)[:N] for v in (h0, a0, x0, y0))
    c.set_fixtures(MODEL_BASIC, h, a, x, y, 20)
    z = torch.tensor(np.random.RandomState(7).uniform(-.5, .5, (64, 45)), dtype=torch.float64, device=c.device)
    U = torch.zeros(64, dtype=torch.float64, device=c.device); g = torch.zeros_like(z)
    k = 64 if N <= 10_000_000 else 16
    c.logp_grad_graph(k, z, U, g, replays=2)
    torch.cuda.synchronize(); ts = []
    for rnd in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); c.logp_grad_graph(k, z, U, g, replays=4); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3 / (4 * k))
    t = float(np.median(ts))
    print(f"N={N:>11d}  us/eval={t:10.2f}  evals/s={1e6 / t:10.1f}  algorithmic GB/s={N * 6 / t / 1e3:8.1f}"
          f"  frac of 8 TB/s={N * 6 / t / 1e3 / 8000:6.3f}", flush=True)
