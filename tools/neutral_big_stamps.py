# diagnostic (needs `make stamps`): the last workgroup's way through dcn::neu_big at N = 1e6
import os, sys
ROOT = os.environ.get('GRAFT_REPO_ROOT', '/root/repo'); sys.path[:0] = [ROOT + '/bpl-next_amd', ROOT]
os.environ["BPLHIP_LIB"] = os.environ.get("STAMPS_LIB", "libbplhip_stamps.so")
import numpy as np, torch
from bpl._ffi import HipContext
c = HipContext(0)
N, T = int(float(os.environ.get('BIGN', '1e6'))), 20
rs = np.random.RandomState(11)
h = rs.randint(0, T, N); a = (h + 1 + rs.randint(0, T - 1, N)) % T
c.set_fixtures_neutral(h, a, rs.poisson(1.4, N), rs.poisson(1.1, N), rs.randint(0, 2, N), T,
                       weights=rs.uniform(0.2, 3.0, N).astype(np.float32))
D = c.dim
o_sat = 2 * T + 1 + 2 * T + 5       # NeuLayout (K = 0, C = 0): aat, adf, corr, hat, hdf, five means, then standardised_attack
names = ("entry fixtures-staged cells-done rates-done barrier-passed adjoint+flush-issued last-arrival epilogue-done "
         "epi:entry epi:scalar-sites epi:coupling-built epi:team-loop epi:sums-folded phase3:maxima-known phase3:loop-done").split()
z = torch.tensor(np.random.RandomState(7).uniform(-.3, .3, (8, D)), dtype=torch.float64, device=c.device)
U = torch.zeros(8, dtype=torch.float64, device=c.device); g = torch.zeros_like(z)
rows = []
for _ in range(10):
    c.logp_grad_graph(16, z, U, g, replays=4); torch.cuda.synchronize()
    for r in g.cpu().numpy():
        st = r[o_sat:o_sat + 15]
        rows.append((st - st[0]) * 0.01)
med = np.median(np.array(rows), axis=0)
print(f"N = {N}: the LAST workgroup of dcn::neu_big, us since its entry (median of {len(rows)} launches under graph replay)")
for k, nm in enumerate(names):
    print(f"  {nm:24s} {med[k]:8.2f}")
