import sys, os
ROOT=os.environ.get('GRAFT_REPO_ROOT','/root/repo'); sys.path[:0]=[ROOT+'/bpl-next_amd', ROOT]
import numpy as np, torch
from bench import synthetic_league
from bpl._ffi import HipContext, MODEL_BASIC, MODEL_EXTENDED
c=HipContext(0)
EXT = os.environ.get('EXT', '0') == '1'
WEIGHTED = os.environ.get('WEIGHTED', '0') == '1'
cov = None
if EXT:
    cov = np.random.RandomState(0).normal(size=(20, 5)); cov = (cov - cov.mean(0)) / cov.std(0)
for n in (1_000_000, 100_000, 2_000_000, 3_000_000):
    h,a,x,y = synthetic_league(n, 20)
    for aw in (8, 4, 2, 8, 4):
        c.set_option('active_waves', aw)
        w = np.exp(-np.linspace(5.0, 0.0, n)).astype(np.float32) if WEIGHTED else None
        c.set_fixtures(MODEL_EXTENDED if EXT else MODEL_BASIC,h,a,x,y,20, weights=w, covariates_std=cov)
        D=c.dim
        z=torch.tensor(np.random.RandomState(7).uniform(-.5,.5,(64,D)),dtype=torch.float64,device=c.device)
        U=torch.zeros(64,dtype=torch.float64,device=c.device); g=torch.zeros_like(z)
        c.logp_grad_graph(64,z,U,g,replays=8); torch.cuda.synchronize(); ts=[]
        for r in range(5):
            e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
            e0.record(); c.logp_grad_graph(64,z,U,g,replays=16); e1.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1)*1e3/1024)
        print(f"N={n:8d} active_waves={aw}: {np.median(ts):6.2f} us/eval", flush=True)
