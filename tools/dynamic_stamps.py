# diagnostic (needs `make stamps`): timeline of the single-launch dynamic kernel at config 4
import os, sys
ROOT = os.environ.get('GRAFT_REPO_ROOT', '/root/repo'); sys.path[:0] = [ROOT + '/bpl-next_amd', ROOT]
os.environ["BPLHIP_LIB"] = os.environ.get("STAMPS_LIB", "libbplhip_stamps.so")
import numpy as np, torch
from bpl._ffi import HipContext
c = HipContext(0)
if os.environ.get('WGS'): c.set_option('dyn_big_wgs', int(os.environ['WGS']))   # the sliced form at another grid size
Tn, G = 100, 50
rs = np.random.RandomState(4)
h, a, gw = [], [], []
for g in range(G):
    p = rs.permutation(Tn)
    h += list(p[0::2]); a += list(p[1::2]); gw += [g] * (Tn // 2)
n = len(h)
if os.environ.get('BIGN'):   # the sliced form (dyn_fused<true>) on N random fixtures: stamps of the 25 workgroups with teams
    n = int(float(os.environ['BIGN'])); rs = np.random.RandomState(5)
    h = rs.randint(0, Tn, n); a = (h + 1 + rs.randint(0, Tn - 1, n)) % Tn; gw = np.sort(rs.randint(0, G, n))
    print(f"N = {n}")
c.set_fixtures_dynamic(np.array(h), np.array(a), rs.poisson(1.5, n), rs.poisson(1.2, n), np.array(gw),
                       np.zeros(n, np.uint8), Tn, G)
D = c.dim
o_u = D - G * Tn
names = "entry cells-stored sigmoid-done (same) rates-done barrier-2 adjoint-done barrier-3 back-done final-done".split()
z = torch.tensor(np.random.RandomState(7).uniform(-.3, .3, (8, D)), dtype=torch.float64, device=c.device)
U = torch.zeros(8, dtype=torch.float64, device=c.device); gbuf = torch.zeros_like(z)
rows, ends = [], []
for mode in ("graph", "single"):
    rows, ends = [], []
    for _ in range(10):
        if mode == "graph":   # steady state: 16 evaluations per replay, the last 8 leave their stamps
            c.logp_grad_graph(16, z, U, gbuf, replays=4); torch.cuda.synchronize()
            gs = gbuf.cpu().numpy()
        else:
            _, g1, _ = c.logp_grad(z[0].contiguous()); gs = g1.cpu().numpy().reshape(1, -1)
        prev_end = None
        order = np.argsort([gs[r][o_u] for r in range(gs.shape[0])])
        for r in order:
            g = gs[r]
            st = np.array([[g[o_u + k * Tn + 4 * b] for k in range(10)] for b in range(Tn // 4)])
            t0 = st[:, 0].min()
            rel = (st - t0) * 0.01
            rel[st == 0] = np.nan
            rows.append(np.concatenate([np.nanmedian(rel, axis=0), np.nanmax(rel, axis=0)]))
            if prev_end is not None:
                ends.append((t0 - prev_end) * 0.01)
            prev_end = np.nanmax(st[:, 9])
    med = np.median(np.array(rows), axis=0)
    print(f"[{mode}] point      median-over-WGs   latest-WG   (us since the first workgroup's entry)")
    for k, nm in enumerate(names):
        print(f"  {nm:14s} {med[k]:10.2f} {med[10 + k]:12.2f}")
    if ends:
        print(f"  gap between one evaluation's final-done and the next one's first entry: median {np.median(ends):.2f} us")
