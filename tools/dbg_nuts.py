import sys, os
ROOT=os.environ.get('GRAFT_REPO_ROOT','/root/repo'); sys.path[:0]=[ROOT+'/bpl-next_amd', ROOT+'/oracle', ROOT+'/tests']
import numpy as np, torch
import cases, dc_oracle as O
from bpl._ffi import HipContext, MODEL_BASIC, default_nuts_cfg
fx=cases.fixtures("dummy")
c=HipContext(0); c.set_fixtures(MODEL_BASIC, fx.home_idx.astype(np.uint16), fx.away_idx.astype(np.uint16), fx.home_goals.astype(np.uint8), fx.away_goals.astype(np.uint8), 20)
z0=np.random.RandomState(2).uniform(-.2,.2,45)
cfg=default_nuts_cfg(); cfg.num_warmup=0; cfg.num_samples=6; cfg.step_size=0.02
out={}
for mode in (1,0):
    c.set_option("device_nuts",mode); out[mode]=c.nuts_run(cfg,(0,11),z0)
(d1,s1),(d0,s0)=out[1],out[0]
print('steps dev', s1['num_steps'], 'host', s0['num_steps'])
print('pe dev', s1['potential_energy'], '\npe host', s0['potential_energy'])
print('acc dev', s1['accept_prob'], '\nacc host', s0['accept_prob'])
print('max|dz| per draw', np.abs(d1-d0).max(axis=1))
print('leapfrogs', s1['total_leapfrogs'], s0['total_leapfrogs'])
for (w,sz,ns_) in ((0,1.0,6),(0,0.3,6),(6,0.02,6),(25,0.02,5),(40,1.0,5)):
    cfg=default_nuts_cfg(); cfg.num_warmup=w; cfg.num_samples=ns_; cfg.step_size=sz
    out={}
    for mode in (1,0):
        c.set_option("device_nuts",mode); out[mode]=c.nuts_run(cfg,(0,11),z0)
    (d1,s1),(d0,s0)=out[1],out[0]
    print(f'--- warm={w} step={sz}: steps dev {s1["num_steps"].tolist()} host {s0["num_steps"].tolist()} div {s1["diverging"].tolist()} {s0["diverging"].tolist()}')
    print('   max|dz|', np.abs(d1-d0).max(axis=1), 'final step', s1['final_step_size'], s0['final_step_size'])
