"""Diagnostic: host overhead of the wall-clock bracket around K = 20 evaluations
(graph replay vs direct launches, synchronize vs event polling)."""
import os, sys, time
ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT + "/bpl-next_amd", ROOT]
import numpy as np, torch
from bench import synthetic_league
from bpl._ffi import HipContext, MODEL_BASIC

K = int(os.environ.get("K", "20"))
h, a, x, y = synthetic_league(1_000_000, 20)
c = HipContext(0); c.set_fixtures(MODEL_BASIC, h, a, x, y, 20)
D = c.dim
z = torch.tensor(np.random.RandomState(7).uniform(-.5, .5, (64, D)), dtype=torch.float64, device=c.device)
U = torch.zeros(64, dtype=torch.float64, device=c.device); g = torch.zeros_like(z)

def graph(k): c.logp_grad_graph(k, z, U, g, replays=1)
def direct(k):
    for i in range(k): c.logp_grad(z[i % 64], U[i % 64:i % 64 + 1], g[i % 64], None)

def measure(run, wait):
    res = []
    for rep in range(7):
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        e0.record(); run(K); e1.record()
        if wait == "poll":
            while not e1.query(): pass
        torch.cuda.synchronize()
        w = (time.perf_counter() - t0) * 1e6
        res.append((w, e0.elapsed_time(e1) * 1e3))
    res.sort()
    return res[len(res) // 2]

graph(K); graph(K); direct(K)
for name, run in (("graph", graph), ("direct", direct)):
    for wait in ("sync", "poll"):
        w, e = measure(run, wait)
        print(f"K={K} {name:6s} {wait:4s}: wall {w:7.1f} us ({w / K:5.2f}/eval)  events {e:7.1f} us ({e / K:5.2f}/eval)")
