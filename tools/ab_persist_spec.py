# A/B on one box: the persistent chain with the next position published before the leaf is booked
# (persist_spec = 1, round 4) and after it (0): same chains bit for bit, leapfrogs/s in situ.
import sys, os
ROOT = os.environ.get('GRAFT_REPO_ROOT', '/root/repo'); sys.path[:0] = [ROOT + '/bpl-next_amd', ROOT, ROOT + '/tests', ROOT + '/oracle']
import numpy as np
from bench import synthetic_league
from bpl._ffi import HipContext, MODEL_BASIC, MODEL_EXTENDED, default_nuts_cfg

N = int(float(os.environ.get('N', '1e6')))
h, a, x, y = synthetic_league(N, 20)
cov = np.random.RandomState(0).normal(size=(20, 5)); cov = (cov - cov.mean(0)) / cov.std(0)
for name, model, kw in (("basic", MODEL_BASIC, {}), ("extended K=5", MODEL_EXTENDED, {"covariates_std": cov})):
    draws = {}
    for rep in range(2):
        for spec in (1, 0):
            c = HipContext(0)
            c.set_option('persist_spec', spec)
            c.set_fixtures(model, h, a, x, y, 20, **kw)
            cfg = default_nuts_cfg(); cfg.num_warmup, cfg.num_samples = 300, 100
            d, st = c.nuts_run(cfg, (0, 42))
            print(f"{name:14s} persist_spec={spec}: {st['total_leapfrogs'] / st['wall_seconds']:10.0f} leapfrogs/s "
                  f"({st['total_leapfrogs']} in {st['wall_seconds']:.3f} s, {1e6 * st['wall_seconds'] / st['total_leapfrogs']:.2f} us each) "
                  f"eps {st['final_step_size']:.3e} div {st['total_divergences']}", flush=True)
            draws.setdefault(spec, d)
            c.close()
    same = np.array_equal(draws[1], draws[0])
    print(f"{name:14s} draws identical with and without speculation: {same}   max|diff| {np.abs(draws[1] - draws[0]).max():.3e}", flush=True)
