#!/usr/bin/env python3
"""bench.py -- log-density+gradient evaluations per second of the Dixon-Coles hot path.

Contract: `python bench.py --gpus N --steps K --warmup W` (for N > 1 launched by
torch.distributed.run, one rank per GPU).  A *step* is ONE evaluation of (U, grad U) of
the basic Dixon-Coles model over the whole synthetic fixture set (N_fix = 1e6 fixtures,
T = 20 teams -- the configuration BASELINE.json's metric is quoted on), inputs resident
in HBM.  Every rank is an independent chain (weak scaling, no data-path collective):
rank 0 generates the fixtures, broadcasts them over RCCL, each rank binds them to its own
libbplhip context and evaluates a fixed cycle of 64 latent points back to back through
the hipGraph path the NUTS driver uses for tree doublings.  Prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "bpl-next_amd"))

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md)
BYTES_PER_FIXTURE = 6  # u16 + u16 + u8 + u8 (10 with f32 weights)


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=4096)
    ap.add_argument("--warmup", type=int, default=512)
    ap.add_argument("--fixtures", type=int, default=1_000_000)
    ap.add_argument("--teams", type=int, default=20)
    ap.add_argument("--graph-len", type=int, default=64,
                    help="evaluations captured per hipGraph (0 = direct launches)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-insitu", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=15.0)
    ap.add_argument("--no-cpu-torch", action="store_true",
                    help="skip the torch-CPU float64 autograd comparator of the cpu_baseline leg")
    return ap.parse_args()


def synthetic_league(n, n_teams, seed=2024):
    """the T(T-1) ordered pairs tiled cyclically; truth strengths
    N(0, 0.3^2), home advantage 0.25, Poisson goals (RandomState(seed))."""
    import itertools

    import numpy as np

    teams = sorted(str(i) for i in range(n_teams))  # string-sorted, bpl/_util.py:131
    idx = {t: i for i, t in enumerate(teams)}
    perms = list(itertools.permutations([str(i) for i in range(n_teams)], 2))
    ph = np.array([idx[p[0]] for p in perms], dtype=np.uint16)
    pa = np.array([idx[p[1]] for p in perms], dtype=np.uint16)
    k = np.arange(n) % len(perms)
    h, a = ph[k], pa[k]
    rs = np.random.RandomState(seed)
    att = rs.normal(0.0, 0.3, n_teams)
    dfn = rs.normal(0.0, 0.3, n_teams)
    x = np.minimum(rs.poisson(np.exp(att[h] - dfn[a] + 0.25)), 255).astype(np.uint8)
    y = np.minimum(rs.poisson(np.exp(att[a] - dfn[h])), 255).astype(np.uint8)
    return h, a, x, y


def cpu_baseline(h, a, x, y, n_teams, zs, budget_s, with_torch=True):
    """The CPU path timed beside the GPU, on this box's host cores, same fixtures, same z cycle,
    bounded samples.  Three comparators:
      value / threads_sweep  oracle/dc_cpu_port.c -- the HIP kernel's algorithm written for host
                             cores (pair-sorted goal bytes, float32 per-fixture arithmetic, float64
                             accumulation, OpenMP with per-thread accumulators, no allocation per
                             evaluation): kind "port".  `value` = the best thread count of the sweep,
                             `cores` = that count; two timed repeats each (spread reported).
      float64_checker        oracle/dc_oracle.c, the float64 restatement the parity tests use
                             (not written for speed; round 1's baseline).
      torch_autograd_f64     "framework AD on CPU": torch float64 autograd of a literal
                             transcription of the reference model (the JAX-on-CPU analogue)."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import dc_oracle as O
    import dc_oracle_c as OC

    fx = O.Fixtures(h, a, x, y, n_teams)
    cf = OC.CFixtures(O.MODEL_BASIC, fx)
    ncpu = OC.max_threads()
    sweep = {}
    # (round 4: 1/16 and 3/16 of the threads join the sweep -- on the 128-thread box the best count was 16 with
    # 32 already half of it, so the optimum could sit at 8 or 24 -- and the best count is timed once more
    # with OMP_WAIT_POLICY=passive, in a process of its own: the policy is read when the runtime starts)
    counts = sorted({1, max(1, ncpu // 16), max(1, ncpu // 8), max(1, 3 * ncpu // 16), max(1, ncpu // 4),
                     max(1, ncpu // 2), ncpu})
    share = budget_s * 0.6 / (2 * len(counts))
    for nt in counts:
        port = OC.CpuPort(cf, nt)
        port.eval_many(zs, 8)  # warm (thread team, caches)
        rates = []
        for _ in range(2):
            k = 16
            while True:  # grow the sample until it fills this repeat's share of the budget
                t0 = time.perf_counter()
                port.eval_many(zs, k)
                el = time.perf_counter() - t0
                if el >= share or k >= 1 << 16:
                    break
                k = min(1 << 16, max(k * 2, int(k * share / max(el, 1e-6))))
            rates.append((k / el, k, el))
        port.close()
        sweep[nt] = rates
    best = max(sweep, key=lambda nt: min(r[0] for r in sweep[nt]))
    v = min(r[0] for r in sweep[best])
    out = {
        "value": v,
        "unit": "evals/s",
        "cores": best,
        "kind": "port",
        "sample": f"{sweep[best][0][1]} evals of the same {len(h)}-fixture workload in {sweep[best][0][2]:.2f}s "
                  f"(x2 repeats), oracle/dc_cpu_port.c, OpenMP {best} of {ncpu} threads "
                  f"(OMP_PROC_BIND={os.environ.get('OMP_PROC_BIND')}, OMP_PLACES={os.environ.get('OMP_PLACES')})",
        "single_thread_value": min(r[0] for r in sweep[1]),
        "threads_sweep": {str(nt): [round(r[0], 1) for r in sweep[nt]] for nt in sweep},
        "repeat_spread": max(r[0] for r in sweep[best]) / v - 1.0,
    }
    try:   # passive wait policy at the best thread count (a fresh OpenMP runtime: a child process)
        import subprocess

        child = (
            "import sys, time, numpy as np; sys.path[:0] = [%r, %r]\n"
            "import bench, dc_oracle as O, dc_oracle_c as OC\n"
            "h, a, x, y = bench.synthetic_league(%d, %d)\n"
            "cf = OC.CFixtures(O.MODEL_BASIC, O.Fixtures(h, a, x, y, %d))\n"
            "zs = np.random.RandomState(7).uniform(-0.5, 0.5, (64, cf.dim if hasattr(cf, 'dim') else 2 * %d + 5))\n"
            "port = OC.CpuPort(cf, %d); port.eval_many(zs, 8)\n"
            "k = %d; t0 = time.perf_counter(); port.eval_many(zs, k); print(k / (time.perf_counter() - t0))\n"
        ) % (ROOT, os.path.join(ROOT, "oracle"), len(h), n_teams, n_teams, n_teams, best,
             max(64, int(v * budget_s * 0.005)))   # (passive waits are ~20x slower: a small sample)
        env = dict(os.environ, OMP_WAIT_POLICY="passive")
        r = subprocess.run([sys.executable, "-c", child], env=env, capture_output=True, text=True, timeout=300)
        out["passive_wait"] = {"threads": best, "value": float(r.stdout.strip().splitlines()[-1]), "unit": "evals/s"}
    except Exception as e:  # pylint: disable=broad-except
        out["passive_wait"] = {"error": f"{type(e).__name__}: {e}"[:200]}
    # the float64 checker (round 1's baseline), all threads and one
    chk = {}
    for label, nt in (("all", ncpu), ("one", 1)):
        OC.potential_and_grad(cf, zs[0], nt)
        t0 = time.perf_counter()
        k = 0
        while time.perf_counter() - t0 < budget_s * 0.1:
            OC.potential_and_grad(cf, zs[k % len(zs)], nt)
            k += 1
        chk[label] = k / (time.perf_counter() - t0)
    out["float64_checker"] = {"value": chk["all"], "threads": ncpu, "single_thread_value": chk["one"],
                              "unit": "evals/s", "source": "oracle/dc_oracle.c"}
    if with_torch:
        import torch
        import dc_torch_ref as TR

        TR.potential_and_grad(O.MODEL_BASIC, fx, zs[0])
        t0 = time.perf_counter()
        kt = 0
        while time.perf_counter() - t0 < budget_s * 0.2 and kt < 64:
            TR.potential_and_grad(O.MODEL_BASIC, fx, zs[kt % len(zs)])
            kt += 1
        out["torch_autograd_f64"] = {"value": kt / (time.perf_counter() - t0), "unit": "evals/s",
                                     "threads": torch.get_num_threads(), "evals": kt,
                                     "source": "oracle/dc_torch_ref.py (torch CPU float64 autograd)"}
    return out


def init_collectives(world, rank, local, args):
    """RCCL over xGMI for N > 1: the fixture broadcast + the timing barrier / max (no data-path
    collective).  A rank whose RCCL init or probe fails exits non-zero: an N > 1 line is an RCCL
    measurement or it is not printed (no silent fall-back to another backend).
    Returns (label for config.collectives or None, device the control tensors live on)."""
    import torch
    import torch.distributed as dist

    dev = torch.device("cuda", local)
    if world == 1:
        return None, dev
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    torch.cuda.set_device(local)
    try:
        dist.init_process_group("nccl", device_id=dev)
        probe = torch.ones(1, device=dev)
        dist.all_reduce(probe)
        torch.cuda.synchronize()
        if int(probe.item()) != world:
            raise RuntimeError(f"all_reduce over RCCL saw {int(probe.item())} of {world} ranks")
    except Exception as e:  # pylint: disable=broad-except
        print(f"# rank {rank}: RCCL unavailable ({type(e).__name__}: {e}); --gpus {args.gpus} "
              "needs RCCL, exiting non-zero", file=sys.stderr)
        sys.stderr.flush()
        os._exit(3)  # (the other ranks fail their collective and torchrun tears the job down)
    return "nccl", dev


def main():
    args = parse_args()
    # OpenMP placement of the CPU comparator (read when libgomp loads, so set before anything else)
    os.environ.setdefault("OMP_PROC_BIND", "close")
    os.environ.setdefault("OMP_PLACES", "cores")
    os.environ.setdefault("OMP_WAIT_POLICY", "active")
    import numpy as np
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    backend, ctl_dev = init_collectives(world, rank, local, args)
    if world != args.gpus and rank == 0:
        print(f"# note: --gpus {args.gpus} but WORLD_SIZE={world}; using {world}", file=sys.stderr)

    from bpl import _dist
    from bpl._ffi import MODEL_BASIC, HipContext, default_nuts_cfg, prng_key, threefry_split

    ctx = HipContext(local)
    dev = ctx.device
    T = args.teams
    if rank == 0:
        h, a, x, y = synthetic_league(args.fixtures, T)
    else:
        h = a = np.zeros(0, np.uint16)
        x = y = np.zeros(0, np.uint8)
    # RCCL broadcast of the fixture SoA from rank 0 (the only data collective)
    bc = _dist.broadcast_fixtures({"home_idx": h, "away_idx": a, "home_goals": x,
                                   "away_goals": y}, device=ctl_dev)
    bc = {k: (v.to(dev) if v is not None else None) for k, v in bc.items()}
    ctx.set_fixtures(MODEL_BASIC, bc["home_idx"], bc["away_idx"], bc["home_goals"],
                     bc["away_goals"], T)
    D = ctx.dim
    n_fix = int(bc["home_idx"].numel())

    # fixed z cycle: RandomState(7).uniform(-0.5, 0.5, (64, D))  
    zs = np.random.RandomState(7).uniform(-0.5, 0.5, (64, D))
    z = torch.tensor(zs, dtype=torch.float64, device=dev)
    U = torch.zeros(64, dtype=torch.float64, device=dev)
    g = torch.zeros_like(z)

    # K evaluations = whole replays of a hipGraph of `graph_len` evaluations + one shorter graph
    # for the remainder, so any --steps / --warmup is timed at the kernel's rate (a graph of
    # gcd(K, W, graph_len) evaluations would be launch bound for odd counts)
    glen = min(args.graph_len, 64)

    replayers = {}

    def run(k):
        if glen > 0:
            q, r = divmod(k, glen)
            if q:
                replayers[glen](q)
            if r:
                replayers[r](1)
        else:
            for i in range(k):
                j = i % 64
                ctx.logp_grad(z[j], U[j:j + 1], g[j], None)

    if glen > 0:  # capture, instantiate AND launch once every graph the run uses, untimed: the
        # first launch of an instantiated graph uploads it (a cold graph inside the timed region
        # cost ~1.3 us per evaluation at --steps 20)
        for k in (args.warmup, args.steps):
            for n in ((glen,) if k >= glen else ()) + ((k % glen,) if k % glen else ()):
                ctx.logp_grad_graph(n, z, U, g, replays=1)
                replayers[n] = ctx.graph_replayer(n, z, U, g)   # (arguments marshalled once)
        torch.cuda.synchronize()

    def bracket(k):
        """k evaluations between barrier + synchronize on both sides: (wall seconds, HIP-event ms)."""
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record()   # (torch creates the HIP event at its first record: not inside the region)
        ev1.record()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()
        t0 = time.perf_counter()
        ev0.record()
        run(k)
        ev1.record()
        # (the host spins on the end event instead of sleeping in the synchronize: a blocked thread's
        # wake-up is 10-20 us, a tenth of the timed region at the driver's --steps 20; the synchronize
        # that brackets the region then returns at once)
        while not ev1.query():
            pass
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()
        return time.perf_counter() - t0, ev0.elapsed_time(ev1)

    # the warm-up goes through the SAME bracket as the timed steps (the first query, the first
    # synchronize after a record: one-off host costs that are not the path's).  At --steps 20 the
    # region was 178 us for 134 us of kernels; rehearsed 168, with the events created and the
    # arguments marshalled outside it 162 (what is left -- ~25 us -- is hipGraphLaunch reaching an
    # idle GPU and the host seeing the end event)
    if args.warmup:
        bracket(args.warmup)
    wall, ev_ms = bracket(args.steps)
    tmax = torch.tensor([wall], dtype=torch.float64, device=ctl_dev)
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    wall_max = float(tmax.item())

    # sanity: the timed outputs are the real thing (compare one point with a direct call)
    # (a point the timed run did evaluate: the graph covers z[0 .. glen-1])
    chk = min(args.steps, glen if glen > 0 else 64) - 1
    Uc, gc, _ = ctx.logp_grad(z[chk].contiguous())
    assert torch.equal(Uc[0], U[chk]) and torch.equal(gc, g[chk]), "graph path != direct path"

    # the evaluation is ONE kernel (dc_eval: streaming + prior + tail workgroups), so the HIP-event
    # timed period of back-to-back evaluations on this stream, ev_ms / steps, is that kernel's
    # average launch duration plus the dependent-launch gap (conservative for the roofline)
    per_eval_us = ev_ms * 1e3 / args.steps
    achieved = n_fix * BYTES_PER_FIXTURE / (per_eval_us * 1e-6) / 1e9

    extra = {}
    if rank == 0 and not args.no_insitu and world == 1:
        # in situ: leapfrogs/s of a real NUTS chain on the same data -- SURVEY.md section 8d's 500 + 500
        # (about 2 s at N = 1e6: ~290 leapfrogs per transition)
        cfg = default_nuts_cfg()
        cfg.num_warmup, cfg.num_samples = 500, 500
        _, st = ctx.nuts_run(cfg, prng_key(42))
        extra["insitu_leapfrogs_per_s"] = st["total_leapfrogs"] / st["wall_seconds"]
        extra["insitu_leapfrogs"] = st["total_leapfrogs"]
        # beside the headline (1 chain per GPU): 64 chains through the chain-vectorised
        # kernel (bplhip_logp_grad_batched, numpyro chain_method="vectorized")
        zc = torch.tensor(np.random.RandomState(7).uniform(-0.5, 0.5, (64, D)),
                          dtype=torch.float64, device=dev)
        Uc2 = torch.zeros(64, dtype=torch.float64, device=dev)
        gc2 = torch.zeros_like(zc)
        ac2 = torch.zeros((64, 4), dtype=torch.float64, device=dev)
        for _ in range(20):
            ctx.logp_grad(zc, Uc2, gc2, ac2)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(200):
            ctx.logp_grad(zc, Uc2, gc2, ac2)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / 200
        extra["vectorised_64_chains"] = {
            "evals_per_s": 64 / us * 1e6,
            "algorithmic_GBps": 64 * n_fix * BYTES_PER_FIXTURE / us / 1e3,
            "us_per_launch": us,
        }

    traffic = None
    tpath = os.path.join(ROOT, "profiles", "traffic.json")
    if os.path.exists(tpath):
        with open(tpath) as f:
            traffic = json.load(f).get("hbm_bytes_per_eval")

    if rank == 0:
        out = {
            "metric": "log-density+grad evals/sec, 1e6 fixtures, 1/2/4/8 chains x MI355X",
            "value": world * args.steps / wall_max,
            "unit": "evals/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": wall_max * 1e3 / args.steps,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32 per-fixture arithmetic, f64 accumulation/epilogue",
            "data": "synthetic",
            "config": {
                "workload": f"dixon_coles_basic logp+grad, {n_fix} fixtures, {T} teams, "
                            f"D={D}, 1 chain per GPU, hipGraph of {min(glen, args.steps)} evals",
                "fixtures": n_fix,
                "teams": T,
                "parallelism": f"{world} independent chain(s), one per GPU",
                "collectives": backend or "none (one process)",
            },
            "roofline": {
                "bound": "hbm",
                "achieved": achieved,
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS,
                "traffic": traffic,
                "kernel": "dc_eval (one launch per evaluation: streaming + prior + tail)",
                "us_per_eval_events": per_eval_us,
                "algorithmic_bytes_per_eval": n_fix * BYTES_PER_FIXTURE,
                # the same fraction on the bytes the kernel actually moves (PMC, profiles/traffic.json)
                "frac_traffic": (traffic / per_eval_us / 1e3 / HBM_PEAK_GBS) if traffic else None,
                "duration_note": "HIP events around the replayed graph divided by its evaluations: kernel "
                                 "duration plus the dependent-launch gap; rocprofv3's per-kernel average "
                                 "(profiles/r04/kernels.md) is the duration alone",
                "regime": "latency bound at this size (an empty kernel in the same graph is 2.06 us of the "
                          "~5.9 us launch); the same kernel reaches ~83% of peak at N=1e7 and ~205% "
                          "(algorithmic) at N=1e8: profiles/r04/n_sweep.txt",
            },
        }
        out.update(extra)
        if not args.no_cpu_baseline and world == 1:  # rank 0 at N = 1 only
            hh = bc["home_idx"].cpu().numpy().view(np.uint16)
            aa = bc["away_idx"].cpu().numpy().view(np.uint16)
            out["cpu_baseline"] = cpu_baseline(hh, aa, bc["home_goals"].cpu().numpy(),
                                               bc["away_goals"].cpu().numpy(), T, zs,
                                               args.cpu_seconds, with_torch=not args.no_cpu_torch)
        print(json.dumps(out))
    ctx.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
