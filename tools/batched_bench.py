# evals/s of the chain-batched launch (grid.y = chains) at N=1e6: one process, interleaved
import sys, os
ROOT=os.environ.get('GRAFT_REPO_ROOT','/root/repo'); sys.path[:0]=[ROOT+'/bpl-next_amd', ROOT]
import numpy as np, torch
from bench import synthetic_league
from bpl._ffi import HipContext, MODEL_BASIC
h,a,x,y = synthetic_league(1_000_000, 20)
c=HipContext(0)
VEC=int(os.environ.get('VEC','1')); TPW=int(os.environ.get('VEC_TPW','0'))
c.set_option('vec_tiles_per_wave',TPW); c.set_option('vec_min_chains', 1 if VEC else 0)
c.set_fixtures(MODEL_BASIC,h,a,x,y,20)
print(f"vec={VEC} vec_tiles_per_wave={TPW}")
res={}
for C in [int(v) for v in os.environ.get('CHAINS','1,2,4,8,16,32,64,128,256').split(',')]:
    z=torch.tensor(np.random.RandomState(7).uniform(-.5,.5,(C,45)),dtype=torch.float64,device=c.device)
    U=torch.zeros(C,dtype=torch.float64,device=c.device); g=torch.zeros_like(z); aux=torch.zeros((C,4),dtype=torch.float64,device=c.device)
    for _ in range(20): c.logp_grad(z,U,g,aux)
    torch.cuda.synchronize(); ts=[]
    for rnd in range(5):
        e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(200): c.logp_grad(z,U,g,aux)
        e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1)*1e3/200)
    t=np.median(ts); print(f"chains={C:3d}  us/launch={t:8.2f}  evals/s={C/t*1e6:10.0f}  alg GB/s={C*6e6/t/1e3:8.1f}")
