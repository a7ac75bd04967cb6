# evals/s of the dynamic (time-varying) model: BASELINE config 4 (T=100, G=50, N=2500) and the N=1e6 variant
import sys, os
ROOT = os.environ.get('GRAFT_REPO_ROOT', '/root/repo'); sys.path[:0] = [ROOT + '/bpl-next_amd', ROOT]
import numpy as np, torch
from bpl._ffi import HipContext

def config4(T=100, G=50, seed=4):
    rs = np.random.RandomState(seed); h, a, g = [], [], []
    for w in range(G):
        p = rs.permutation(T); h += list(p[0::2]); a += list(p[1::2]); g += [w] * (T // 2)
    n = len(h)
    return np.array(h), np.array(a), rs.poisson(1.5, n), rs.poisson(1.2, n), np.array(g), np.zeros(n, int)

def big(N=1_000_000, T=100, G=50, seed=5):
    rs = np.random.RandomState(seed); h = rs.randint(0, T, N); a = (h + 1 + rs.randint(0, T - 1, N)) % T
    return h, a, rs.poisson(1.5, N), rs.poisson(1.2, N), np.sort(rs.randint(0, G, N)), np.zeros(N, int)

c = HipContext(0)
c.set_option('dyn_gather', int(os.environ.get('DYN_GATHER', '1')))   # 0: float64 atomics into the cells (round 3)
CASES = [("config4 N=2500", config4(), 0, 1), ("N=1e6", big(), 0, 1)]
if os.environ.get('SWEEP', '0') == '1':   # the sliced single launch at other grid sizes, and the four-launch path
    CASES += [(f"N=1e6 wgs={w}", big(), w, 1) for w in (128, 384, 512, 768)] + [("N=1e6 four launches", big(), 0, 0)]
    CASES += [("N=2e5", big(200_000), 0, 1), ("N=4e6", big(4_000_000), 0, 1)]
for name, (h, a, x, y, g, nv), wgs, fused in CASES:
    c.set_option('dyn_big_wgs', wgs); c.set_option('fused_small', fused)
    c.set_fixtures_dynamic(h, a, x, y, g, nv, 100, 50)
    D = c.dim
    z = torch.tensor(np.random.RandomState(7).uniform(-.3, .3, (8, D)), dtype=torch.float64, device=c.device)
    U = torch.zeros(8, dtype=torch.float64, device=c.device); gr = torch.zeros_like(z)
    c.logp_grad_graph(16, z, U, gr, replays=2); torch.cuda.synchronize(); ts = []
    for rnd in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); c.logp_grad_graph(16, z, U, gr, replays=8); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3 / 128)
    t = float(np.median(ts))
    print(f"dynamic {name:16s} D={D}: {t:9.2f} us/eval  {1e6 / t:10.1f} evals/s  algorithmic GB/s={h.size * 9 / t / 1e3:8.2f}", flush=True)
    if os.environ.get('INSITU', '1') == '1' and h.size <= 10000:
        from bpl._ffi import default_nuts_cfg
        cfg = default_nuts_cfg(); cfg.num_warmup, cfg.num_samples, cfg.max_tree_depth = 150, 50, 8
        for eng, dn in (("chain on the device (wide leaf launches)", 1), ("host tree engine", 0)):
            if dn == 0 and os.environ.get('HOST_ENGINE', '1') != '1': continue
            c.set_option('device_nuts', dn)
            _, st = c.nuts_run(cfg, (0, 42))
            c.set_option('device_nuts', 1)
            print(f"    in situ, {eng}: NUTS {cfg.num_warmup}+{cfg.num_samples} transitions, depth <= 8: "
                  f"{st['total_leapfrogs'] / st['wall_seconds']:9.0f} leapfrogs/s "
                  f"({st['total_leapfrogs']} in {st['wall_seconds']:.2f} s)", flush=True)
