import sys, os
ROOT = os.environ.get('GRAFT_REPO_ROOT', '/root/repo'); sys.path[:0] = [ROOT + '/bpl-next_amd', ROOT, ROOT + '/tests', ROOT + '/oracle']
import numpy as np, torch
import cases
from bpl._ffi import HipContext, MODEL_BASIC, default_nuts_cfg
fx = cases.fixtures("dummy")
c = HipContext(0)
c.set_fixtures(MODEL_BASIC, fx.home_idx.astype(np.uint16), fx.away_idx.astype(np.uint16),
               fx.home_goals.astype(np.uint8), fx.away_goals.astype(np.uint8), 20)
for (ass, amm) in ((1, 0), (0, 1), (1, 1)):
    cfg = default_nuts_cfg(); cfg.num_warmup, cfg.num_samples = 150, 10
    cfg.adapt_step_size = ass; cfg.adapt_mass_matrix = amm
    if not ass: cfg.step_size = 0.05
    z0 = np.random.RandomState(2).uniform(-0.2, 0.2, (3, 45))
    keys = [(0, 11), (0, 12), (0, 13)]
    print("adapt step", ass, "adapt mass", amm)
    for C in (1, 3):
        multi = c.nuts_run_chains(cfg, keys[:C], z0[:C])
        for ch in range(C):
            d1, s1 = c.nuts_run(cfg, keys[ch], z0[ch])
            dm, sm = multi[ch]
            print(f" C={C} chain {ch}: leap single {s1['total_leapfrogs']} lock {sm['total_leapfrogs']}  eps {s1['final_step_size']:.4f} {sm['final_step_size']:.4f}"
                  f" acc {s1['mean_accept_prob']:.3f} {sm['mean_accept_prob']:.3f}  div {s1['total_divergences']} {sm['total_divergences']}"
                  f"  imm[:4] {np.round(s1['inverse_mass_matrix'][:4],4).tolist()} {np.round(sm['inverse_mass_matrix'][:4],4).tolist()}")
