"""Diagnostic: per-phase timeline of dc_eval from the stamped build (make -C bpl-next_amd/csrc stamps).
Never used for reported numbers: the stamps perturb the kernel; read the SHARES."""
import ctypes as C, os, sys
ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT + "/bpl-next_amd", ROOT]
import numpy as np, torch
from bench import synthetic_league
from bpl import _ffi
_ffi._LIB_NAME = os.environ.get("STAMPS_LIB", "libbplhip_stamps.so")
from bpl._ffi import HipContext, MODEL_BASIC, MODEL_EXTENDED

TEAMS = int(os.environ.get('TEAMS', '20'))
def run(n, model=MODEL_BASIC, max_wg=255, k=0, weighted=False):
    h, a, x, y = synthetic_league(n, TEAMS)
    w = np.exp(-np.linspace(5.0, 0.0, n)).astype(np.float32) if weighted else None
    cov = None
    if k:
        cov = np.random.RandomState(0).normal(size=(TEAMS, k)); cov = (cov - cov.mean(0)) / cov.std(0)
    c = HipContext(0); c.set_option("max_wg", max_wg); c.set_fixtures(model, h, a, x, y, TEAMS, weights=w, covariates_std=cov)
    lib = c._lib
    lib.bplhip_debug_stamps.argtypes = [C.c_void_p, C.c_void_p, C.c_int]; lib.bplhip_debug_stamps.restype = C.c_int
    nwg = lib.bplhip_debug_stamps(c._h, None, 0)
    z = torch.tensor(np.random.RandomState(7).uniform(-.5, .5, c.dim), dtype=torch.float64, device=c.device)
    if os.environ.get("STAMPS_GRAPH"):   # the last launch of a replayed graph (dependent launches, as bench.py times them)
        zz = z.repeat(8, 1).contiguous(); UU = torch.zeros(8, dtype=torch.float64, device=c.device); gg = torch.zeros_like(zz)
        c.logp_grad_graph(8, zz, UU, gg, replays=3)
    else:
        for _ in range(5): c.logp_grad(z)
    torch.cuda.synchronize()
    buf = np.zeros((nwg + 1) * 16, dtype=np.uint64)
    lib.bplhip_debug_stamps(c._h, buf.ctypes.data_as(C.c_void_p), buf.size)
    st = buf[: nwg * 16].reshape(nwg, 16).astype(np.int64)
    t0 = st[:, 0].min()
    rel = (st - t0) * 0.01  # us (100 MHz)
    names = ["entry", "tables", "bounds", "stream", "slab", "drain", "ticket", "tail:start", "tail:loads", "tail:adj", "tail:end", "barrier", "loads-landed", "lane-math"]
    print(f"--- N={n} model={model} covariates={k} weighted={weighted} blocks={nwg} (block 0 = prior workgroup)")
    print("  prior WG: entry=%.2f scalars=%.2f cells=%.2f bounds=%.2f sums=%.2f amx=%.2f div=%.2f done=%.2f drain=%.2f ticket=%.2f" % (rel[0, 0], rel[0, 1], rel[0, 2], rel[0, 3], rel[0, 12], rel[0, 13], rel[0, 15], rel[0, 4], rel[0, 5], rel[0, 6]))
    names = names + ["epi", "sums-done"]
    for k in [0, 1, 2, 12, 13, 3, 15, 11, 4, 5, 6]:
        col = rel[1:, k][st[1:, k] > 0]
        if col.size: print(f"  {names[k]:10s} median {np.median(col):7.2f} us  min {col.min():7.2f}  max {col.max():7.2f}")
    life = (st[1:, 6] - st[1:, 0]) * 10.0  # ns, entry -> arrival
    print("  shader clock over the streaming workgroups' lives: median %.2f GHz" % np.median(st[1:, 9] / life))
    last = int(np.argmax(st[:, 10]))
    print("  tail WG", last, " ".join(f"{names[k]}={rel[last, k]:.2f}" for k in range(11)),
          f"epi:sums={rel[last, 11]:.2f} epi:puts={rel[last, 14]:.2f}",
          f"| past 64 teams: early poll (wave 1) {rel[0, 11]:.2f} -> {rel[0, 6]:.2f}")
    pw = (buf[nwg * 16: (nwg + 1) * 16].astype(np.int64) - t0) * 0.01
    print("  prior part, per wave, arrival at the barrier that ends the bounds: " + " ".join(f"{v:.2f}" for v in pw[8:]))
    print("  prior part, per wave, own top-two jobs done (complete pair tables): " + " ".join(f"{v:.2f}" for v in pw[:8]))
    print("  general epilogue (past 64 teams; overwrites waves 4..6 of the line before last): header read %.2f, teams done %.2f, sums done %.2f" % tuple(pw[12:15]))
    c.close()

def run_nuts(n=1_000_000):
    """timeline of an ordinary leapfrog of a persistent chain (NUTS-aware launch): the run is
    cut after BPLHIP_DEBUG_MAX_STEPS launches, so the record is the last launch's"""
    from bpl._ffi import default_nuts_cfg, BplHipError
    os.environ["BPLHIP_DEBUG_MAX_STEPS"] = "3000"
    h, a, x, y = synthetic_league(n, 20)
    k = int(os.environ.get("NUTS_K", "-1"))   # -1: basic model; >= 0: extended model with k covariates
    cov = None
    if k > 0:
        cov = np.random.RandomState(0).normal(size=(20, k)); cov = (cov - cov.mean(0)) / cov.std(0)
    w = np.exp(-np.linspace(5.0, 0.0, n)).astype(np.float32) if os.environ.get("NUTS_W") else None
    c = HipContext(0); c.set_option("persist_spec", int(os.environ.get("PERSIST_SPEC", "1")))
    c.set_fixtures(MODEL_BASIC if k < 0 else MODEL_EXTENDED, h, a, x, y, 20, weights=w, covariates_std=cov)
    lib = c._lib
    lib.bplhip_debug_stamps.argtypes = [C.c_void_p, C.c_void_p, C.c_int]; lib.bplhip_debug_stamps.restype = C.c_int
    nwg = lib.bplhip_debug_stamps(c._h, None, 0)
    cfg = default_nuts_cfg(); cfg.num_warmup, cfg.num_samples = 300, 100
    try:
        c.nuts_run(cfg, (0, 42))
    except BplHipError as e:
        print("   (run cut:", e, ")")
    buf = np.zeros((nwg + 1) * 16, dtype=np.uint64)
    lib.bplhip_debug_stamps(c._h, buf.ctypes.data_as(C.c_void_p), buf.size)
    st = buf[: nwg * 16].reshape(nwg, 16).astype(np.int64)
    rel = (st - st[:, 0].min()) * 0.01
    print(f"--- NUTS-aware launch, N={n}, blocks={nwg}, model={'basic' if k < 0 else f'extended K={k}'}{' weighted' if w is not None else ''}")
    print("  prior WG: entry=%.2f scalars=%.2f cells=%.2f bounds=%.2f done=%.2f ticket=%.2f" % (rel[0, 0], rel[0, 1], rel[0, 2], rel[0, 3], rel[0, 4], rel[0, 6]))
    for k, nm in [(0, "entry"), (1, "tables"), (2, "bounds"), (12, "loads-landed"), (3, "stream"), (4, "slab"), (6, "ticket")]:
        col = rel[1:, k]
        print(f"  {nm:12s} median {np.median(col):7.2f} us  min {col.min():7.2f}  max {col.max():7.2f}")
    last = int(np.argmax(st[:, 10]))
    print("  tail WG", last, " ".join(f"{nm}={rel[last, k]:.2f}" for k, nm in [(6, "ticket"), (7, "tail:start"), (8, "tail:loads"), (9, "tail:colsums"), (14, "outputs"), (13, "leaf:prepared"), (12, "leaf:start"), (11, "moves:end"), (15, "weights:end"), (5, "published (persistent kernel)")]))
    print("  (persistent kernel: a step's record; entry = the step's start in that workgroup; tail WG entry=%.2f)" % rel[0, 0])
    print("  tail WG raw slots (us since its step start): " + " ".join(f"{k}:{(st[0, k] - st[0, 0]) * 0.01:.2f}" for k in range(16)))
    pw = (buf[nwg * 16: (nwg + 1) * 16].astype(np.int64) - st[:, 0].min()) * 0.01
    print("  prior part, thread 0 after the bounds barrier (team sums read | records read | combined | arg-pairs | record written): " + " ".join(f"{v:.2f}" for v in pw[:5]))
    print("  prior part, per wave, arrival at the barrier that ends the bounds: " + " ".join(f"{v:.2f}" for v in pw[8:]))
    print("  seq (thread 0): " + " ".join(f"{v:.2f}" for v in pw[:4]))
    print("  deferred leaf (hook) since the tail WG's step start: begin %.2f prepared %.2f moves done %.2f" % tuple((buf[nwg * 16 + k].astype(np.int64) - st[0, 0]) * 0.01 for k in range(3)))
    c.close()

if len(sys.argv) > 1 and sys.argv[1] == "nuts":
    run_nuts()
elif len(sys.argv) > 1 and sys.argv[1] == "basic":
    run(1_000_000)
else:
    for n in (1_000_000,):
        run(n)
    run(1_000_000, MODEL_EXTENDED)
    run(1_000_000, MODEL_EXTENDED, k=5)
    run(1_000_000, MODEL_EXTENDED, k=5, weighted=True)
