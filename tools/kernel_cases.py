"""Workload driver for the per-kernel profiles (tools/profile_all.sh): `kernel_cases.py <case>`
runs a fixed number of launches of ONE kernel family so that rocprofv3's per-kernel averages
and PMC counters belong to it.  Prints one JSON line with the case's byte accounting."""
import json, os, sys
ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT + "/bpl-next_amd", ROOT]
import numpy as np, torch
from bench import synthetic_league
from bpl._ffi import HipContext, MODEL_BASIC, MODEL_EXTENDED, default_nuts_cfg

case = sys.argv[1]
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 512
T = 20
c = HipContext(0)
meta = {"case": case, "launches": steps}


def evals(D, chains=1):
    z = torch.tensor(np.random.RandomState(7).uniform(-.5, .5, (max(chains, 64), D)), dtype=torch.float64, device=c.device)
    if chains == 1:
        U = torch.zeros(64, dtype=torch.float64, device=c.device); g = torch.zeros_like(z)
        c.logp_grad_graph(64, z, U, g, replays=max(1, steps // 64))
    else:
        zc = z[:chains].contiguous()
        U = torch.zeros(chains, dtype=torch.float64, device=c.device); g = torch.zeros_like(zc)
        aux = torch.zeros((chains, 4), dtype=torch.float64, device=c.device)
        for _ in range(steps):
            c.logp_grad(zc, U, g, aux)
    torch.cuda.synchronize()


def league(model, n=1_000_000, weighted=False, cov=False, T=T):
    h, a, x, y = synthetic_league(n, T)
    w = np.exp(-1.0 * np.linspace(5, 0, n)).astype(np.float32) if weighted else None
    cv = None
    if cov:
        cv = np.random.RandomState(0).normal(size=(T, 5)); cv = (cv - cv.mean(0)) / cv.std(0)
    c.set_fixtures(model, h, a, x, y, T, weights=w, covariates_std=cv)
    meta["teams"] = T
    pad = 1.006  # pair runs padded to 32 fixtures, the whole to 2048
    meta.update(n=n, algorithmic_bytes_per_launch=n * (10 if weighted else 6),
                library_copy_bytes_per_launch=int(n * pad * (6.25 if weighted else 2.25)))


if case == "basic":
    league(MODEL_BASIC); evals(c.dim)
elif case in ("t100", "t200"):
    league(MODEL_BASIC, T=int(case[1:])); evals(c.dim)
elif case == "c3":
    league(MODEL_EXTENDED, cov=True); evals(c.dim)
elif case == "c3w":
    league(MODEL_EXTENDED, weighted=True, cov=True); evals(c.dim)
elif case == "nuts":
    league(MODEL_BASIC)
    cfg = default_nuts_cfg(); cfg.num_warmup, cfg.num_samples = 60, 40
    _, st = c.nuts_run(cfg, (0, 42))
    meta["launches"] = int(st["total_leapfrogs"]); meta["leapfrogs_per_s"] = st["total_leapfrogs"] / st["wall_seconds"]
elif case == "vec64":
    league(MODEL_BASIC); evals(c.dim, chains=64)
    meta["chains_per_launch"] = 64
    meta["algorithmic_bytes_per_launch"] *= 64          # 64 evaluations per launch
    meta["library_copy_bytes_per_launch"] *= 8          # the fixtures are read once per 8 chains
elif case == "dyn_c4":
    Tn, G = 100, 50
    rs = np.random.RandomState(4)
    h, a, gw = [], [], []
    for g in range(G):
        p = rs.permutation(Tn)
        h += list(p[0::2]); a += list(p[1::2]); gw += [g] * (Tn // 2)
    n = len(h)
    c.set_fixtures_dynamic(np.array(h), np.array(a), rs.poisson(1.5, n), rs.poisson(1.2, n), np.array(gw),
                           np.zeros(n, np.uint8), Tn, G)
    meta.update(n=n, algorithmic_bytes_per_launch=n * 8, latent_dim=c.dim,
                z_side_bytes_per_evaluation=c.dim * 8 * 2 + Tn * G * 12 * 8)
    z = torch.tensor(np.random.RandomState(7).uniform(-.3, .3, c.dim), dtype=torch.float64, device=c.device)
    for _ in range(steps):
        c.logp_grad(z)
    torch.cuda.synchronize()
elif case == "dyn_nuts":   # config 4 in situ: evaluation + the wide leaf's launches per leapfrog
    Tn, G = 100, 50
    rs = np.random.RandomState(4)
    h, a, gw = [], [], []
    for g in range(G):
        p = rs.permutation(Tn)
        h += list(p[0::2]); a += list(p[1::2]); gw += [g] * (Tn // 2)
    n = len(h)
    c.set_fixtures_dynamic(np.array(h), np.array(a), rs.poisson(1.5, n), rs.poisson(1.2, n), np.array(gw),
                           np.zeros(n, np.uint8), Tn, G)
    cfg = default_nuts_cfg(); cfg.num_warmup, cfg.num_samples, cfg.max_tree_depth = 60, 20, 8
    _, st = c.nuts_run(cfg, (0, 42))
    meta.update(n=n, latent_dim=c.dim, launches=int(st["total_leapfrogs"]),
                leapfrogs_per_s=st["total_leapfrogs"] / st["wall_seconds"])
elif case == "neutral":
    N = 570
    rs = np.random.RandomState(11)
    h = rs.randint(0, T, N); a = (h + 1 + rs.randint(0, T - 1, N)) % T
    c.set_fixtures_neutral(h, a, rs.poisson(1.4, N), rs.poisson(1.1, N), rs.randint(0, 2, N), T,
                           weights=rs.uniform(0.2, 3.0, N).astype(np.float32))
    meta.update(n=N, algorithmic_bytes_per_launch=N * 11, latent_dim=c.dim)
    z = torch.tensor(np.random.RandomState(7).uniform(-.3, .3, c.dim), dtype=torch.float64, device=c.device)
    for _ in range(steps):
        c.logp_grad(z)
    torch.cuda.synchronize()
elif case == "dyn_1e6":   # BASELINE config 4's throughput variant: N = 1e6 over the same 100 teams x 50 gameweeks
    Tn, G, N = 100, 50, 1_000_000
    rs = np.random.RandomState(5)
    h = rs.randint(0, Tn, N); a = (h + 1 + rs.randint(0, Tn - 1, N)) % Tn
    x, y, gw = rs.poisson(1.5, N), rs.poisson(1.2, N), np.sort(rs.randint(0, G, N))
    c.set_fixtures_dynamic(h, a, x, y, gw, np.zeros(N, np.uint8), Tn, G)
    meta.update(n=N, algorithmic_bytes_per_launch=N * 8, latent_dim=c.dim)
    z = torch.tensor(np.random.RandomState(7).uniform(-.3, .3, c.dim), dtype=torch.float64, device=c.device)
    for _ in range(steps):
        c.logp_grad(z)
    torch.cuda.synchronize()
elif case == "neutral_1e6":
    N = 1_000_000
    rs = np.random.RandomState(11)
    h = rs.randint(0, T, N); a = (h + 1 + rs.randint(0, T - 1, N)) % T
    c.set_fixtures_neutral(h, a, rs.poisson(1.4, N), rs.poisson(1.1, N), rs.randint(0, 2, N), T,
                           weights=rs.uniform(0.2, 3.0, N).astype(np.float32))
    meta.update(n=N, algorithmic_bytes_per_launch=N * 11, latent_dim=c.dim)
    z = torch.tensor(np.random.RandomState(7).uniform(-.3, .3, c.dim), dtype=torch.float64, device=c.device)
    for _ in range(steps):
        c.logp_grad(z)
    torch.cuda.synchronize()
elif case == "predict":
    S = 1000
    rs = np.random.RandomState(0)
    c.predict_set_posterior(rs.normal(0, .3, (S, T)), rs.normal(0, .3, (S, T)), rs.normal(.25, .05, S), rs.uniform(-.1, .05, S))
    pairs = np.array([(h, a) for h in range(T) for a in range(T) if h != a])
    big = np.tile(pairs, (64, 1))
    for _ in range(max(1, steps // 64)):
        c.predict_score_grid(big[:, 0], big[:, 1], 15)
    meta.update(fixtures_per_launch=len(big), draws=S, posterior_bytes=S * (2 * T + 2) * 4,
                output_bytes_per_launch=len(big) * 256 * 8)
elif case == "predict_venue":   # the neutral-venue family's rate form, with confederations
    S, C = 1000, 5
    rs = np.random.RandomState(0)
    tabs = [rs.normal(0, .2, (S, T)) for _ in range(6)]
    c.predict_set_posterior_venue(*tabs, rs.uniform(-.1, .05, S), confederation_strength=rs.normal(0, .2, (S, C)))
    pairs = np.array([(h, a) for h in range(T) for a in range(T) if h != a])
    big = np.tile(pairs, (64, 1))
    nv = (np.arange(len(big)) % 3 == 0).astype(np.uint8)
    for _ in range(max(1, steps // 64)):
        c.predict_score_grid(big[:, 0], big[:, 1], 15, neutral=nv, conf=(big[:, 0] % C, big[:, 1] % C))
    meta.update(fixtures_per_launch=len(big), draws=S, posterior_bytes=S * (6 * T + C + 1) * 4,
                output_bytes_per_launch=len(big) * 256 * 8)
else:
    raise SystemExit(f"unknown case {case}")
print(json.dumps(meta))
