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
if os.environ.get('AB_MODELS', '1') == '1':   # the float64 models' single launches (tree barriers, flagged records)
    def timed(c, D):
        z = torch.tensor(np.random.RandomState(7).uniform(-.3, .3, (8, D)), dtype=torch.float64, device=c.device)
        U = torch.zeros(8, dtype=torch.float64, device=c.device); g = torch.zeros_like(z)
        c.logp_grad_graph(16, z, U, g, replays=2); torch.cuda.synchronize(); ts = []
        for rnd in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); c.logp_grad_graph(16, z, U, g, replays=8); e1.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) * 1e3 / 128)
        return float(np.median(ts))
    c = HipContext(0)
    rs = np.random.RandomState(4); hh, aa, gg = [], [], []
    for w in range(50):
        p = rs.permutation(100); hh += list(p[0::2]); aa += list(p[1::2]); gg += [w] * 50
    c.set_fixtures_dynamic(np.array(hh), np.array(aa), rs.poisson(1.5, 2500), rs.poisson(1.2, 2500), np.array(gg), np.zeros(2500, int), 100, 50)
    out.append(f"dyn c4: {timed(c, c.dim):.2f} us")
    rs = np.random.RandomState(5); hb = rs.randint(0, 100, N); ab = (hb + 1 + rs.randint(0, 99, N)) % 100
    c.set_fixtures_dynamic(hb, ab, rs.poisson(1.5, N), rs.poisson(1.2, N), np.sort(rs.randint(0, 50, N)), np.zeros(N, int), 100, 50)
    out.append(f"dyn 1e6: {timed(c, c.dim):.2f} us")
    rs = np.random.RandomState(11); hn = rs.randint(0, 20, N); an = (hn + 1 + rs.randint(0, 19, N)) % 20
    c.set_fixtures_neutral(hn, an, rs.poisson(1.4, N), rs.poisson(1.1, N), rs.randint(0, 2, N), 20, weights=rs.uniform(0.2, 3.0, N).astype(np.float32))
    out.append(f"neutral 1e6: {timed(c, c.dim):.2f} us")
    c.close()
print(" | ".join(out), flush=True)
'''
libs = sys.argv[1:] or ["libbplhip.so", "libbplhip_base.so"]
for rnd in range(int(os.environ.get("ROUNDS", "3"))):
    for lib in libs:
        env = dict(os.environ, BPLHIP_LIB=lib)
        r = subprocess.run([sys.executable, "-c", CHILD], env=env, capture_output=True, text=True)
        print(f"{lib:24s} {r.stdout.strip() or r.stderr.strip()[-300:]}", flush=True)
