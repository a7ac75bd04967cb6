# evaluation latency against the number of teams at N = 1e6 (basic model): python tools/teams_sweep.py
import sys, os
ROOT = os.environ.get('GRAFT_REPO_ROOT', '/root/repo'); sys.path[:0] = [ROOT + '/bpl-next_amd', ROOT]
import numpy as np, torch
from bench import synthetic_league
from bpl._ffi import HipContext, MODEL_BASIC
c = HipContext(0)
if "ACTIVE_WAVES" in os.environ:   # experiment: a fixed partition (8 = all waves own tiles, half the workgroups)
    c.set_option("active_waves", int(os.environ["ACTIVE_WAVES"]))
if "MAX_WG" in os.environ:
    c.set_option("max_wg", int(os.environ["MAX_WG"]))
if "PAIR_ORDER" in os.environ:    # 0: fixtures in (home, away) order whatever the league's size (default: Z-order past 64 teams)
    c.set_option("pair_order", int(os.environ["PAIR_ORDER"]))
N = int(float(os.environ.get("N", "1e6")))
for T in [int(t) for t in os.environ.get("TEAMS_LIST", "20,32,48,64,100,200").split(",")]:
    h, a, x, y = synthetic_league(N, T)
    c.set_fixtures(MODEL_BASIC, h, a, x, y, T)
    D = c.dim
    z = torch.tensor(np.random.RandomState(7).uniform(-.5, .5, (64, D)), dtype=torch.float64, device=c.device)
    U = torch.zeros(64, dtype=torch.float64, device=c.device); g = torch.zeros_like(z)
    c.logp_grad_graph(64, z, U, g, replays=4); torch.cuda.synchronize(); ts = []
    for r in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); c.logp_grad_graph(64, z, U, g, replays=8); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3 / 512)
    print(f"N={N} T={T:4d} D={D:4d} pairs={T * (T - 1):6d}: {np.median(ts):7.2f} us/eval", flush=True)
