"""Timing only (the variants compute garbage): evaluation time of the basic and the extended model (5 covariates),
N = 1e6, per library built with -DDC_WHATIF=n -- which part of an evaluation is on its critical path?
python tools/whatif_ab.py libbplhip.so libbplhip_whatif7.so ...   (three interleaved rounds, a process per library)
Build a variant:  git apply tools/whatif.patch && hipcc <the Makefile's FLAGS> -DDC_WHATIF=7 -o bpl-next_amd/bpl/libbplhip_whatif7.so
bpl-next_amd/csrc/bplhip.hip && git checkout bpl-next_amd/csrc/dc_kernels.hip.h   (variants: the patch's comments and
profiles/r04/whatif.txt; 21-24 concern leagues of more than 64 teams: time them with BPLHIP_LIB=... tools/teams_sweep.py)"""
import os, sys, subprocess
CHILD = r'''
import sys, os
ROOT = os.environ.get('GRAFT_REPO_ROOT', '/root/repo'); sys.path[:0] = [ROOT + '/bpl-next_amd', ROOT]
import numpy as np, torch
from bench import synthetic_league
from bpl._ffi import HipContext, MODEL_BASIC, MODEL_EXTENDED
c = HipContext(0)
out = []
for model, k in ((MODEL_BASIC, 0), (MODEL_EXTENDED, 5)):
    h, a, x, y = synthetic_league(1_000_000, 20)
    cov = None
    if k:
        cov = np.random.RandomState(0).normal(size=(20, k)); cov = (cov - cov.mean(0)) / cov.std(0)
    c.set_fixtures(model, h, a, x, y, 20, covariates_std=cov)
    D = c.dim
    z = torch.tensor(np.random.RandomState(7).uniform(-.5, .5, (64, D)), dtype=torch.float64, device=c.device)
    U = torch.zeros(64, dtype=torch.float64, device=c.device); g = torch.zeros_like(z)
    c.logp_grad_graph(64, z, U, g, replays=4); torch.cuda.synchronize(); ts = []
    for r in range(7):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); c.logp_grad_graph(64, z, U, g, replays=8); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3 / 512)
    out.append("%.3f" % np.median(ts))
print(" | ".join(out))
'''
for rnd in range(3):
    for lib in sys.argv[1:]:
        env = dict(os.environ, BPLHIP_LIB=lib)
        r = subprocess.run([sys.executable, "-c", CHILD], env=env, capture_output=True, text=True)
        print(f"{lib:28s} basic | ext5 (us/eval): {r.stdout.strip().splitlines()[-1] if r.stdout.strip() else r.stderr[-300:]}", flush=True)
