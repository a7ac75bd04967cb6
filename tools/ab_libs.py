# A/B of two (or more) builds of the library ON ONE BOX: op-level graph replay of the basic model at N = 1e6
# (what bench.py times) and the in-situ single chain, each build in a process of its own, interleaved.
#   python tools/ab_libs.py libbplhip.so libbplhip_base.so     (files under bpl-next_amd/bpl/)
import os, subprocess, sys
ROOT = os.environ.get('GRAFT_REPO_ROOT', '/root/repo')
CHILD = r'''
import sys, os
ROOT = os.environ.get('GRAFT_REPO_ROOT', '/root/repo'); sys.path[:0] = [ROOT + '/bpl-next_amd', ROOT]
import numpy as np, torch
from bench import synthetic_league
from bpl._ffi import HipContext, MODEL_BASIC, MODEL_EXTENDED, default_nuts_cfg
N = int(float(os.environ.get('N', '1e6')))
h, a, x, y = synthetic_league(N, 20)
cov = np.random.RandomState(0).normal(size=(20, 5)); cov = (cov - cov.mean(0)) / cov.std(0)
out = []
for name, model, kw in (("basic", MODEL_BASIC, {}), ("ext5", MODEL_EXTENDED, {"covariates_std": cov})):
    c = HipContext(0); c.set_fixtures(model, h, a, x, y, 20, **kw)
    D = c.dim
    z = torch.tensor(np.random.RandomState(7).uniform(-.5, .5, (64, D)), dtype=torch.float64, device=c.device)
    U = torch.zeros(64, dtype=torch.float64, device=c.device); g = torch.zeros_like(z)
    c.logp_grad_graph(1024, z, U, g, replays=2); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for rep in range(5):
        e0.record(); c.logp_grad_graph(1024, z, U, g, replays=4); e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1e3 / 4096)
    cfg = default_nuts_cfg(); cfg.num_warmup, cfg.num_samples = 300, 100
    d, st = c.nuts_run(cfg, (0, 42))
    out.append(f"{name}: {best:.3f} us/eval, in situ {1e6 * st['wall_seconds'] / st['total_leapfrogs']:.3f} us/leapfrog")
    c.close()
print(" | ".join(out), flush=True)
'''
libs = sys.argv[1:] or ["libbplhip.so", "libbplhip_base.so"]
for rnd in range(int(os.environ.get("ROUNDS", "3"))):
    for lib in libs:
        env = dict(os.environ, BPLHIP_LIB=lib)
        r = subprocess.run([sys.executable, "-c", CHILD], env=env, capture_output=True, text=True)
        print(f"{lib:24s} {r.stdout.strip() or r.stderr.strip()[-300:]}", flush=True)
