# row f-2: the device predict path -- predict_score_proba over a season's score grids
import sys, os, time
ROOT = os.environ.get('GRAFT_REPO_ROOT', '/root/repo'); sys.path[:0] = [ROOT + '/bpl-next_amd', ROOT]
import numpy as np, torch
from bpl._ffi import HipContext
S, T, MG = 1000, 20, 16
rs = np.random.RandomState(0)
att, dfn = rs.normal(0, .3, (S, T)), rs.normal(0, .3, (S, T))
ha, cc = rs.normal(.25, .05, S), rs.uniform(-.1, .05, S)
pairs = [(h, a) for h in range(T) for a in range(T) if h != a]          # 380 fixtures
gx, gy = np.meshgrid(np.arange(MG), np.arange(MG), indexing="ij")
h = np.repeat([p[0] for p in pairs], MG * MG); a = np.repeat([p[1] for p in pairs], MG * MG)
x = np.tile(gx.ravel(), len(pairs)); y = np.tile(gy.ravel(), len(pairs))
c = HipContext(0); c.predict_set_posterior(att, dfn, ha, cc)
out = c.predict_score_proba(h, a, x, y)
ts = []
for _ in range(5):
    t0 = time.perf_counter(); out = c.predict_score_proba(h, a, x, y); ts.append(time.perf_counter() - t0)
t = float(np.median(ts))
# host numpy reference of the same quantity (float64), for the time comparison only
t0 = time.perf_counter()
lh = np.exp(att[:, h] - dfn[:, a] + ha[:, None]); la = np.exp(att[:, a] - dfn[:, h])
th = time.perf_counter() - t0
print(f"predict_score_proba: {h.size} entries x {S} draws: {t * 1e3:.2f} ms end to end (H2D + kernel + D2H), "
      f"{h.size * S / t / 1e9:.2f} G entry-draws/s; grids sum to {out.reshape(len(pairs), -1).sum(1).mean():.6f}; "
      f"(numpy: the two rate arrays alone take {th * 1e3:.0f} ms)")
# the grid kernel (one wave per fixture on the matrix cores): the same season of grids
g = c.predict_score_grid([p[0] for p in pairs], [p[1] for p in pairs], MG - 1)
ts = []
for _ in range(5):
    t0 = time.perf_counter(); g = c.predict_score_grid([p[0] for p in pairs], [p[1] for p in pairs], MG - 1); ts.append(time.perf_counter() - t0)
tg = float(np.median(ts))
print(f"predict_score_grid: {len(pairs)} fixtures x {MG}x{MG} x {S} draws: {tg * 1e3:.2f} ms end to end; "
      f"max |grid - pointwise| = {np.abs(g.reshape(-1) - out).max():.2e}")
big = np.tile(np.array(pairs), (256, 1))  # 97 280 fixtures
t0 = time.perf_counter(); gb = c.predict_score_grid(big[:, 0], big[:, 1], MG - 1); tb = time.perf_counter() - t0
print(f"predict_score_grid: {len(big)} fixtures ({gb.size * 8 / 1e6:.0f} MB of grids): {tb * 1e3:.1f} ms end to end, "
      f"{len(big) * MG * MG * S / tb / 1e12:.2f} T cell-draws/s")
t0 = time.perf_counter(); g32 = c.predict_score_grid(big[:, 0], big[:, 1], MG - 1, dtype=np.float32); t32 = time.perf_counter() - t0
print(f"predict_score_grid, float32 output: {len(big)} fixtures ({g32.size * 4 / 1e6:.0f} MB of grids): {t32 * 1e3:.1f} ms end to end; "
      f"equal to the rounded float64 grids: {bool(np.array_equal(g32, gb.astype(np.float32)))}")
