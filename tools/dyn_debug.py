import os, sys
ROOT = os.environ.get('GRAFT_REPO_ROOT', '/root/repo'); sys.path[:0] = [ROOT + '/bpl-next_amd', ROOT, ROOT + '/oracle']
import numpy as np, torch
import dc_dynamic_oracle as DO
from bpl._ffi import HipContext
c = HipContext(0)
fx = DO.small_recipe()
D = DO.latent_dim(fx.n_gameweeks, fx.n_teams, fx.k)
sl = DO.site_slices(fx.n_gameweeks, fx.n_teams, fx.k)
z = np.random.RandomState(int(os.environ.get("SEED","7"))).uniform(-0.3, 0.3, D)
if os.environ.get("SEED","7") == "2": z[sl["mean_home_attack"]] = 1.2
c.set_fixtures_dynamic(fx.home_idx, fx.away_idx, fx.home_goals, fx.away_goals, fx.gameweek, fx.neutral, fx.n_teams, fx.n_gameweeks, random_walk=bool(int(os.environ.get("RW", "1"))))
zt = torch.tensor(z, dtype=torch.float64, device=c.device)
out = {}
for fused in (0, 1):
    c.set_option("fused_small", fused)
    U, g, aux = c.logp_grad(zt)
    out[fused] = (float(U[0]), g.cpu().numpy().copy(), aux.cpu().numpy().copy())
print("G,T", fx.n_gameweeks, fx.n_teams, "U", out[0][0], out[1][0], "aux", out[0][2], out[1][2])
for name, s_ in sl.items():
    d = np.abs(out[0][1][s_] - out[1][1][s_])
    if d.max() > 1e-9:
        print(name, "max diff", d.max(), "at", np.argmax(d), "of", d.size)
        a = out[0][1][s_].reshape(-1); b = out[1][1][s_].reshape(-1)
        idx = np.argsort(-d.reshape(-1))[:6]
        print("   ", [(int(i), round(a[i], 4), round(b[i], 4)) for i in idx])
print("--- repeated single-launch evaluations")
c.set_option("fused_small", 1)
prev = None
for rep in range(4):
    U, g, aux = c.logp_grad(zt)
    cur = (float(U[0]), g.cpu().numpy().copy())
    if prev is not None:
        print("rep", rep, "dU", cur[0] - prev[0], "max dg", np.abs(cur[1] - prev[1]).max())
        for name, s_ in sl.items():
            d = np.abs(cur[1][s_] - prev[1][s_])
            if d.max() > 1e-9:
                print("   ", name, d.max())
    prev = cur
