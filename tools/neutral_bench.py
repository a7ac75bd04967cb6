# row f-4: the neutral-venue model -- evals/s at the reference's recipe size and at N = 1e6, in-situ NUTS
import sys, os
ROOT = os.environ.get('GRAFT_REPO_ROOT', '/root/repo'); sys.path[:0] = [ROOT + '/bpl-next_amd', ROOT]
import numpy as np, torch
from bpl._ffi import HipContext, default_nuts_cfg
c = HipContext(0)
for N, T in ((570, 20), (1_000_000, 20)):
    rs = np.random.RandomState(11)
    h = rs.randint(0, T, N); a = (h + 1 + rs.randint(0, T - 1, N)) % T
    c.set_fixtures_neutral(h, a, rs.poisson(1.4, N), rs.poisson(1.1, N), rs.randint(0, 2, N), T,
                           weights=rs.uniform(0.2, 3.0, N).astype(np.float32))
    D = c.dim
    z = torch.tensor(np.random.RandomState(7).uniform(-.3, .3, (8, D)), dtype=torch.float64, device=c.device)
    U = torch.zeros(8, dtype=torch.float64, device=c.device); g = torch.zeros_like(z)
    c.logp_grad_graph(16, z, U, g, replays=2); torch.cuda.synchronize(); ts = []
    for rnd in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); c.logp_grad_graph(16, z, U, g, replays=8); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3 / 128)
    t = float(np.median(ts))
    cfg = default_nuts_cfg(); cfg.num_warmup, cfg.num_samples = 100, 50
    d, st = c.nuts_run(cfg, (0, 42))
    print(f"neutral N={N:8d} D={D}: {t:8.2f} us/eval {1e6 / t:10.1f} evals/s  algorithmic GB/s={N * 11 / t / 1e3:8.2f}"
          f"  in-situ (persistent chain) {st['total_leapfrogs'] / st['wall_seconds']:8.0f} leapfrogs/s", flush=True)
