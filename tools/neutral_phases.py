# diagnostic (needs `make stamps`): cost of each step of the single-launch neutral kernel, measured
# as the change in time per evaluation (graph replays) when the kernel leaves after that step
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
    D = c.dim
    z = torch.tensor(np.random.RandomState(7).uniform(-.3, .3, (8, D)), dtype=torch.float64, device=c.device)
    U = torch.zeros(8, dtype=torch.float64, device=c.device); g = torch.zeros_like(z)
    out = []
    for stop in (1, 2, 3, 4, 5, 6, 9):
        c.set_option("debug_stop", stop)
        c.logp_grad_graph(16, z, U, g, replays=2); torch.cuda.synchronize(); ts = []
        for rnd in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); c.logp_grad_graph(16, z, U, g, replays=8); e1.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) * 1e3 / 128)
        out.append(float(np.median(ts)))
    print(f"N={N} T={T}: us/eval when leaving after step A C D E1 E2 F | whole: " + " ".join(f"{v:.2f}" for v in out))
