# in-situ NUTS throughput of lock-step chains on one GPU (bplhip_nuts_run_chains) at N = 1e6
import sys, os, time
ROOT = os.environ.get('GRAFT_REPO_ROOT', '/root/repo'); sys.path[:0] = [ROOT + '/bpl-next_amd', ROOT]
import numpy as np, torch
from bench import synthetic_league
from bpl._ffi import HipContext, MODEL_BASIC, default_nuts_cfg

N = int(float(os.environ.get('N', '1e6')))
h, a, x, y = synthetic_league(N, 20)
c = HipContext(0)
c.set_option('vec_tiles_per_wave', int(os.environ.get('VEC_TPW', '0')))
c.set_fixtures(MODEL_BASIC, h, a, x, y, 20)
cfg = default_nuts_cfg(); cfg.num_warmup, cfg.num_samples = 300, 100
for sd in (42, 43):
    d, st = c.nuts_run(cfg, (0, sd))
    print(f"single chain seed {sd} (device tree): {st['total_leapfrogs'] / st['wall_seconds']:10.0f} leapfrogs/s "
          f"({st['total_leapfrogs']} in {st['wall_seconds']:.3f} s) eps {st['final_step_size']:.2e}", flush=True)
c.set_option('persistent_nuts', int(os.environ.get('PERSIST', '1')))
if os.environ.get('GRIDY'): c.set_option('gridy_max_chains', int(os.environ['GRIDY']))   # (default: the library's, 8)
for C in [int(v) for v in os.environ.get('CHAINS', '2,4,8,16,32,64').split(',')]:
    res = c.nuts_run_chains(cfg, [(0, 42 + i) for i in range(C)])
    leap = sum(r[1]['total_leapfrogs'] for r in res); wall = res[0][1]['wall_seconds']
    acc = np.mean([r[1]['mean_accept_prob'] for r in res])
    print(f"chains on one GPU={C:3d}: {leap / wall:10.0f} leapfrogs/s aggregate ({leap} in {wall:.3f} s, "
          f"mean accept {acc:.3f}, steps/transition {leap / (C * 400):.1f})", flush=True)
    if C <= 4:
        print("    per chain:", [(r[1]['total_leapfrogs'], float('%.2e' % r[1]['final_step_size'])) for r in res], flush=True)
