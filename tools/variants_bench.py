# one-process A/B of launch geometries at N=1e6 (interleaved rounds); BPLHIP_LIB selects the build
import sys, os
ROOT=os.environ.get('GRAFT_REPO_ROOT','/root/repo'); sys.path[:0]=[ROOT+'/bpl-next_amd', ROOT]
import numpy as np, torch
from bench import synthetic_league
from bpl._ffi import HipContext, MODEL_BASIC
N=int(float(os.environ.get('NFIX','1e6')))
h,a,x,y = synthetic_league(N, 20)
zs=np.random.RandomState(7).uniform(-.5,.5,(64,45))
res={}; ctxs={}
for name,maxwg in {'wg255':255,'wg191':191,'wg383':383,'wg510':510}.items():
    c=HipContext(0); c.set_option('max_wg',maxwg); c.set_fixtures(MODEL_BASIC,h,a,x,y,20)
    z=torch.tensor(zs,dtype=torch.float64,device=c.device); U=torch.zeros(64,dtype=torch.float64,device=c.device); g=torch.zeros_like(z)
    ctxs[name]=(c,z,U,g); res[name]=[]
for rnd in range(5):
    for name,(c,z,U,g) in ctxs.items():
        c.logp_grad_graph(64,z,U,g,replays=4); torch.cuda.synchronize()
        e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
        e0.record(); c.logp_grad_graph(64,z,U,g,replays=32); e1.record(); torch.cuda.synchronize()
        res[name].append(e0.elapsed_time(e1)*1e3/2048)
ref=None
for name,(c,z,U,g) in ctxs.items():
    if ref is None: ref=(U.clone(),g.clone())
    print(os.environ.get('BPLHIP_LIB','default'), name, 'us/eval median %.2f min %.2f'%(np.median(res[name]),min(res[name])), 'maxdU %.2e'%float((U-ref[0]).abs().max()), 'maxdg %.2e'%float((g-ref[1]).abs().max()))
