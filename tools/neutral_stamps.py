# diagnostic: phase timeline of the single-launch neutral kernel (needs `make stamps`)
import os, sys
ROOT = os.environ.get('GRAFT_REPO_ROOT', '/root/repo'); sys.path[:0] = [ROOT + '/bpl-next_amd', ROOT]
os.environ["BPLHIP_LIB"] = os.environ.get("STAMPS_LIB", "libbplhip_stamps.so")
import numpy as np, torch
from bpl._ffi import HipContext
c = HipContext(0)
for N, T in ((570, 20), (4_000, 100)):
    rs = np.random.RandomState(11)
    h = rs.randint(0, T, N); a = (h + 1 + rs.randint(0, T - 1, N)) % T
    c.set_fixtures_neutral(h, a, rs.poisson(1.4, N), rs.poisson(1.1, N), rs.randint(0, 2, N), T,
                           weights=rs.uniform(0.2, 3.0, N).astype(np.float32))
    z = torch.tensor(np.random.RandomState(7).uniform(-.3, .3, c.dim), dtype=torch.float64, device=c.device)
    rows = []
    for _ in range(20):
        _, g, _ = c.logp_grad(z)
        rows.append(g.cpu().numpy()[:10] * 0.01)
    print(f"N={N} T={T}: phase starts (us since entry, median of 20): "
          + " ".join(f"{v:.2f}" for v in np.median(rows, axis=0)))
