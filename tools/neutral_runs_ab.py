import sys, os
ROOT = os.environ.get('GRAFT_REPO_ROOT', '/root/repo'); sys.path[:0] = [ROOT + '/bpl-next_amd', ROOT]
import numpy as np, torch
from bpl._ffi import HipContext
N, T = 1_000_000, 20
rs = np.random.RandomState(11)
h = rs.randint(0, T, N); a = (h + 1 + rs.randint(0, T - 1, N)) % T
x, y, nv, w = rs.poisson(1.4, N), rs.poisson(1.1, N), rs.randint(0, 2, N), rs.uniform(0.2, 3.0, N).astype(np.float32)
for runs in (1, 0, 1, 0):
    c = HipContext(0); c.set_option('neu_runs', runs)
    c.set_fixtures_neutral(h, a, x, y, nv, T, weights=w)
    D = c.dim
    z = torch.tensor(np.random.RandomState(7).uniform(-.3, .3, (8, D)), dtype=torch.float64, device=c.device)
    U = torch.zeros(8, dtype=torch.float64, device=c.device); g = torch.zeros_like(z)
    c.logp_grad_graph(16, z, U, g, replays=2); torch.cuda.synchronize(); ts = []
    for rnd in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); c.logp_grad_graph(16, z, U, g, replays=8); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3 / 128)
    print(f"neu_runs={runs}: {np.median(ts):.2f} us/eval  U[0]={float(U[0]):.9f}", flush=True)
    c.close()
