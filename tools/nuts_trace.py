# one short single-chain NUTS run at N = 1e6 (for rocprofv3 --kernel-trace: kernel durations vs gaps)
import sys, os
ROOT = os.environ.get('GRAFT_REPO_ROOT', '/root/repo'); sys.path[:0] = [ROOT + '/bpl-next_amd', ROOT]
import numpy as np, torch
from bench import synthetic_league
from bpl._ffi import HipContext, MODEL_BASIC, default_nuts_cfg
h, a, x, y = synthetic_league(1_000_000, 20)
c = HipContext(0); c.set_fixtures(MODEL_BASIC, h, a, x, y, 20)
cfg = default_nuts_cfg(); cfg.num_warmup, cfg.num_samples = 60, 20
d, st = c.nuts_run(cfg, (0, 42))
print(st['total_leapfrogs'], st['wall_seconds'], st['total_leapfrogs'] / st['wall_seconds'])
