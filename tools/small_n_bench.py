# evaluation latency and in-situ NUTS at the reference's own sizes (a season = 380 matches)
import sys, os, time
ROOT = os.environ.get('GRAFT_REPO_ROOT', '/root/repo'); sys.path[:0] = [ROOT + '/bpl-next_amd', ROOT]
import numpy as np, torch, itertools
from bpl._ffi import HipContext, MODEL_BASIC, MODEL_EXTENDED, default_nuts_cfg
c = HipContext(0)
T = 20
perms = list(itertools.permutations(range(T), 2))
for seasons in (1, 10, 100):
    rs = np.random.RandomState(42)
    h = np.array([p[0] for p in perms] * seasons); a = np.array([p[1] for p in perms] * seasons)
    x = rs.poisson(2.1, h.size); y = rs.poisson(1.7, h.size)
    for model, name in ((MODEL_BASIC, "basic"), (MODEL_EXTENDED, "extended")):
        c.set_fixtures(model, h, a, x, y, T)
        D = c.dim
        z = torch.tensor(np.random.RandomState(7).uniform(-.3, .3, (8, D)), dtype=torch.float64, device=c.device)
        U = torch.zeros(8, dtype=torch.float64, device=c.device); g = torch.zeros_like(z)
        c.logp_grad_graph(16, z, U, g, replays=2); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); c.logp_grad_graph(16, z, U, g, replays=16); e1.record(); torch.cuda.synchronize()
        t = e0.elapsed_time(e1) * 1e3 / 256
        cfg = default_nuts_cfg(); cfg.num_warmup, cfg.num_samples = 500, 1000
        t0 = time.perf_counter(); d, st = c.nuts_run(cfg, (0, 42)); wall = time.perf_counter() - t0
        print(f"{name:8s} N={h.size:6d}: {t:6.2f} us/eval; fit 500+1000: {wall:6.2f} s, {st['total_leapfrogs']} leapfrogs, "
              f"{st['total_leapfrogs'] / st['wall_seconds']:8.0f} leapfrogs/s", flush=True)
