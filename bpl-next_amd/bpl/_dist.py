"""Multi-GPU plumbing: one process per GPU, independent chains, no data-path collective.

The reference's only parallel axis is independent chains (numpyro `num_chains`,
reachable through `mcmc_kwargs` at bpl/dixon_coles.py:105; chain_method="parallel" is a
jax.pmap over chains with replicated model args and stacked samples).  Here rank r of a
`torch.distributed` job (backend "nccl" = RCCL over xGMI on ROCm; "gloo" in the CPU
tests) owns GPU r and the chains c with c % world == r.  Collectives are used for exactly
two things: broadcasting the fixture arrays from rank 0 and gathering the draws.
"""

from __future__ import annotations

import os
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np


def world() -> Tuple[int, int]:
    """(rank, world_size) of the current torch.distributed job, (0, 1) if none."""
    try:
        import torch.distributed as dist
    except ImportError:  # pragma: no cover
        return 0, 1
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def local_device_index() -> int:
    return int(os.environ.get("LOCAL_RANK", "0"))


def chains_of_rank(num_chains: int, rank: int, world_size: int) -> List[int]:
    """Round-robin chain ownership: chain c runs on rank c % world."""
    return [c for c in range(num_chains) if c % world_size == rank]


def _comm_device(device=None):
    import torch
    import torch.distributed as dist

    if device is not None:
        return device
    return torch.device("cuda", local_device_index()) if dist.get_backend() == "nccl" else torch.device("cpu")


def broadcast_fixtures(arrays: Dict[str, Optional[np.ndarray]], device=None, src: int = 0):
    """Broadcast the fixture arrays (and their shapes) from rank `src` to every rank.

    `arrays` maps name -> numpy array or None; dtypes must agree across ranks (uint16
    indices, uint8 goals, float32 weights, float64 covariates).  Returns the same dict
    holding rank src's data as torch tensors on `device` (uint16 as int16 bit patterns).
    With world_size == 1 this is just the upload.
    """
    import torch
    import torch.distributed as dist

    rank, ws = world()
    dev = _comm_device(device) if ws > 1 else (device or torch.device("cpu"))
    names = sorted(arrays)
    out = {}
    # 1. shapes (-1 = None), so receivers can allocate
    meta = torch.full((len(names), 3), -1, dtype=torch.int64)
    if rank == src:
        for i, k in enumerate(names):
            a = arrays[k]
            if a is not None:
                a = np.asarray(a)
                meta[i, 0] = a.ndim
                for d in range(a.ndim):
                    meta[i, 1 + d] = a.shape[d]
    if ws > 1:
        meta = meta.to(dev)
        dist.broadcast(meta, src=src)
        meta = meta.cpu()
    # 2. payloads
    for i, k in enumerate(names):
        nd = int(meta[i, 0])
        if nd < 0:
            out[k] = None
            continue
        shape = tuple(int(meta[i, 1 + d]) for d in range(nd))
        ref = arrays[k]
        if rank == src:
            a = np.ascontiguousarray(ref)
            if a.dtype == np.uint16:
                a = a.view(np.int16)
            t = torch.from_numpy(a).to(dev)
        else:
            dt = np.asarray(ref).dtype if ref is not None else None
            if dt is None:
                raise ValueError(f"rank {rank}: array '{k}' is None here but not on rank {src}")
            tdt = {
                np.dtype(np.uint16): torch.int16,
                np.dtype(np.uint8): torch.uint8,
                np.dtype(np.float32): torch.float32,
                np.dtype(np.float64): torch.float64,
            }[np.dtype(dt)]
            t = torch.empty(shape, dtype=tdt, device=dev)
        if ws > 1:
            # as raw bytes: neither RCCL nor gloo broadcasts 16-bit integer tensors
            dist.broadcast(t.view(torch.uint8), src=src)
        out[k] = t
    return out


def gather_chains(local: np.ndarray, num_chains: int, device=None) -> np.ndarray:
    """All-gather per-chain arrays.  `local` is [n_local_chains, ...] holding this rank's
    chains (ascending chain id); returns [num_chains, ...] in chain order on every rank."""
    import torch
    import torch.distributed as dist

    rank, ws = world()
    if ws == 1:
        return local
    dev = _comm_device(device)
    per = (num_chains + ws - 1) // ws  # max chains on any rank
    pad_shape = (per,) + tuple(local.shape[1:])
    buf = np.full(pad_shape, np.nan, dtype=np.float64)
    buf[: local.shape[0]] = local
    t = torch.from_numpy(buf).to(dev)
    outs = [torch.empty_like(t) for _ in range(ws)]
    dist.all_gather(outs, t)
    full = np.empty((num_chains,) + tuple(local.shape[1:]), dtype=np.float64)
    for r in range(ws):
        o = outs[r].cpu().numpy()
        for j, c in enumerate(chains_of_rank(num_chains, r, ws)):
            full[c] = o[j]
    return full
