"""The N > 1 path on CPU: world_size-2 gloo jobs exercising bpl/_dist.py (broadcast of the
fixture arrays from rank 0, round-robin chain ownership, all-gather of the draws in chain
order) and bpl/_mcmc.py's multi-chain orchestration with a TEST stand-in context."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _setup(rank, world, port):
    for p in (os.path.join(os.path.dirname(HERE), "bpl-next_amd"),
              os.path.join(os.path.dirname(HERE), "oracle"), HERE):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank),
                      WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    dist.init_process_group("gloo", rank=rank, world_size=world)


def _worker_collectives(rank, world, port, q):
    _setup(rank, world, port)
    from bpl import _dist

    assert _dist.world() == (rank, world)
    # rank 0 owns the data; rank 1 passes dtype-only placeholders
    if rank == 0:
        arrays = {"home_idx": np.arange(10, dtype=np.uint16) + 60000, "away_idx": np.arange(10, dtype=np.uint16),
                  "home_goals": np.arange(10, dtype=np.uint8), "away_goals": np.ones(10, np.uint8),
                  "weights": np.linspace(0, 1, 10).astype(np.float32), "covariates": None}
    else:
        arrays = {"home_idx": np.zeros(0, np.uint16), "away_idx": np.zeros(0, np.uint16),
                  "home_goals": np.zeros(0, np.uint8), "away_goals": np.zeros(0, np.uint8),
                  "weights": np.zeros(0, np.float32), "covariates": None}
    bc = _dist.broadcast_fixtures(arrays)
    h = bc["home_idx"].numpy().view(np.uint16)
    assert h.tolist() == (np.arange(10) + 60000).tolist()  # uint16 survives the int16 view
    assert bc["covariates"] is None and bc["weights"].dtype == torch.float32
    assert np.allclose(bc["weights"].numpy(), np.linspace(0, 1, 10))
    # 5 chains over 2 ranks: rank0 -> 0,2,4  rank1 -> 1,3
    mine = _dist.chains_of_rank(5, rank, world)
    local = np.stack([np.full((3, 2), float(c)) for c in mine])
    full = _dist.gather_chains(local, 5)
    assert full.shape == (5, 3, 2)
    assert [full[c, 0, 0] for c in range(5)] == [0.0, 1.0, 2.0, 3.0, 4.0]
    q.put((rank, "ok"))
    dist.destroy_process_group()


def _worker_mcmc(rank, world, port, q):
    _setup(rank, world, port)
    import dc_oracle as O
    from bpl._ffi import MODEL_BASIC
    from bpl._mcmc import run_mcmc
    from fake_ctx import FakeCtx

    td = O.dummy_data_recipe()
    fx, _ = O.fixtures_from_training_data(td)
    if rank != 0:  # only rank 0's data counts: scramble the others
        fx.home_goals = fx.home_goals[::-1].copy()
    samples, info = run_mcmc(MODEL_BASIC, fx.home_idx, fx.away_idx, fx.home_goals, fx.away_goals,
                             20, random_state=42, num_warmup=40, num_samples=30,
                             mcmc_kwargs={"num_chains": 2}, context_factory=FakeCtx)
    q.put((rank, samples["attack"].shape, float(samples["attack"].sum()),
           float(samples["home_advantage"].mean()), info["total_leapfrogs"]))
    dist.destroy_process_group()


def _run(fn, world=2):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=fn, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=240)
        assert p.exitcode == 0, f"worker exit code {p.exitcode}"
    return sorted(q.get(timeout=5) for _ in range(world))


def test_broadcast_and_gather_world2():
    assert _run(_worker_collectives) == [(0, "ok"), (1, "ok")]


def test_two_chains_on_two_ranks_match_single_process():
    out = _run(_worker_mcmc)
    (r0, shape0, sum0, ha0, lf0), (r1, shape1, sum1, ha1, lf1) = out
    assert shape0 == shape1 == (60, 20)           # 2 chains x 30 draws, on every rank
    assert sum0 == sum1 and lf0 == lf1            # identical gathered posterior on both ranks
    # the same job in ONE process (both chains on rank 0) gives the same draws
    import dc_oracle as O
    from bpl._ffi import MODEL_BASIC
    from bpl._mcmc import run_mcmc
    from fake_ctx import FakeCtx

    fx, _ = O.fixtures_from_training_data(O.dummy_data_recipe())
    samples, info = run_mcmc(MODEL_BASIC, fx.home_idx, fx.away_idx, fx.home_goals, fx.away_goals,
                             20, random_state=42, num_warmup=40, num_samples=30,
                             mcmc_kwargs={"num_chains": 2}, context_factory=FakeCtx)
    assert float(samples["attack"].sum()) == pytest.approx(sum0, rel=1e-12)
    assert info["total_leapfrogs"] == lf0
    assert abs(ha0 - np.log(2.1 / 1.7)) < 0.2
