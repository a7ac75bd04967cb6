"""The `NUTS(model)` + `MCMC(...).run(...)` + `get_samples()` sequence of the reference
(bpl/dixon_coles.py:100-122, bpl/extended_dixon_coles.py:293-331), driven through
libbplhip.so.  Host Python only orchestrates: fixture upload / broadcast, one
`bplhip_nuts_run` per chain, gather, and the constrained/deterministic site map.
"""

from __future__ import annotations

from typing import Any, Callable, Dict, Optional

import numpy as np

from bpl import _dist
from bpl._ffi import MODEL_BASIC, MODEL_EXTENDED, default_nuts_cfg, prng_key, threefry_split

_MCMC_KEYS = {"num_chains", "thinning", "progress_bar", "chain_method", "jit_model_args",
              "postprocess_fn"}
_RUN_KEYS = {"init_params", "extra_fields"}


def latent_sites(model: int, T: int, K: int):
    """(name, size) of every latent site in flat (sorted-name) order."""
    if model == MODEL_BASIC:
        return [("attack_decentered", T), ("corr_coef_raw", 1), ("defence_decentered", T),
                ("home_advantage", 1), ("mean_defence", 1), ("std_attack", 1),
                ("std_defence", 1)]
    s = []
    if K:
        s.append(("attack_coefficients", K))
    s.append(("corr_coef_raw", 1))
    if K:
        s.append(("defence_coefficients", K))
    s += [("home_advantage_decentered", T), ("mean_defence", 1), ("mean_home_advantage", 1),
          ("standardised_attack", T), ("standardised_defence", T), ("std_attack", 1),
          ("std_defence", 1), ("std_home_advantage", 1), ("u", 1)]
    return s


def _flatten_init(init_params, model, T, K):
    if init_params is None:
        return None
    if isinstance(init_params, dict):
        parts = []
        for name, size in latent_sites(model, T, K):
            if name not in init_params:
                raise KeyError(f"init_params is missing site '{name}'")
            v = np.asarray(init_params[name], dtype=np.float64).reshape(-1)
            if v.size != size:
                raise ValueError(f"init_params['{name}'] has size {v.size}, expected {size}")
            parts.append(v)
        return np.concatenate(parts)
    return np.asarray(init_params, dtype=np.float64).reshape(-1)


def _sigmoid_clipped(x):
    s = 1.0 / (1.0 + np.exp(-x))
    return np.clip(s, np.finfo(np.float32).tiny, 1.0 - np.finfo(np.float32).eps)


def constrained_samples(model, T, K, z, site) -> Dict[str, np.ndarray]:
    """numpyro `get_samples()`: latent sites in constrained space + deterministic sites.
    `site` holds attack/defence/home_advantage/corr_coef from bplhip_constrain."""
    out = dict(site)
    o = 0
    for name, size in latent_sites(model, T, K):
        v = z[:, o:o + size]
        o += size
        if name.startswith("std_"):
            v = np.exp(v)  # HalfNormal sites: ExpTransform
        elif name in ("corr_coef_raw", "u"):
            v = _sigmoid_clipped(v)  # Beta sites: SigmoidTransform
        if size == 1 and not name.endswith("_coefficients"):
            v = v[:, 0]
        out.setdefault(name, v)
    if model == MODEL_EXTENDED:
        out["rho"] = 2.0 * out["u"] - 1.0
    return out


def run_mcmc(
    model: int,
    home_ind: np.ndarray,
    away_ind: np.ndarray,
    home_goals,
    away_goals,
    n_teams: int,
    *,
    weights: Optional[np.ndarray] = None,
    covariates_std: Optional[np.ndarray] = None,
    random_state: int = 42,
    num_warmup: int = 500,
    num_samples: int = 1000,
    mcmc_kwargs: Optional[Dict[str, Any]] = None,
    run_kwargs: Optional[Dict[str, Any]] = None,
    context_factory: Optional[Callable[[int], Any]] = None,
):
    """Returns (samples: dict of [chains*S, ...] arrays, info: dict)."""
    mcmc_kwargs = dict(mcmc_kwargs or {})
    run_kwargs = dict(run_kwargs or {})
    bad = set(mcmc_kwargs) - _MCMC_KEYS
    if bad:
        raise TypeError(f"MCMC got unexpected keyword argument(s) {sorted(bad)}")
    bad = set(run_kwargs) - _RUN_KEYS
    if bad:
        raise TypeError(f"MCMC.run got unexpected keyword argument(s) {sorted(bad)}")
    num_chains = int(mcmc_kwargs.get("num_chains", 1))
    thinning = int(mcmc_kwargs.get("thinning", 1))
    if num_chains < 1 or thinning < 1:
        raise ValueError("num_chains and thinning must be >= 1")
    # numpyro's MCMC takes these too; none is silently ignored here (round 3 accepted and dropped them):
    #   postprocess_fn  the map from unconstrained draws to the sites of `get_samples()`: the library's own
    #                   (bplhip_constrain + the closed-form transforms below) is the only one there is
    #   jit_model_args  a compilation knob of the JAX path; nothing is traced here, either value is a no-op
    #   extra_fields    (MCMC.run) per-draw sampler statistics: every one this sampler keeps is returned in
    #                   `info` anyway; the names asked for are checked against them and echoed back
    if mcmc_kwargs.get("postprocess_fn") is not None:
        raise NotImplementedError("postprocess_fn: the device sampler constrains its draws itself "
                                  "(bplhip_constrain); transform the returned samples instead")
    extra = tuple(run_kwargs.get("extra_fields") or ())
    known = {"potential_energy", "accept_prob", "mean_accept_prob", "adapt_state.step_size", "step_size",
             "num_steps", "diverging", "energy"}
    bad = [f for f in extra if f not in known]
    if bad:
        raise ValueError(f"extra_fields {bad} are not collected by this sampler; available: {sorted(known - {'energy'})}")
    if "energy" in extra:
        raise ValueError("extra_fields 'energy' (the Hamiltonian of the proposal) is not kept per draw; "
                         "'potential_energy' is")

    hg = np.asarray(home_goals)
    ag = np.asarray(away_goals)
    if hg.size and (hg.min() < 0 or ag.min() < 0 or hg.max() > 255 or ag.max() > 255):
        raise ValueError("goals must be integers in [0, 255]")
    arrays = {
        "home_idx": np.asarray(home_ind, dtype=np.uint16),
        "away_idx": np.asarray(away_ind, dtype=np.uint16),
        "home_goals": hg.astype(np.uint8),
        "away_goals": ag.astype(np.uint8),
        "weights": None if weights is None else np.asarray(weights, dtype=np.float32),
        "covariates": None if covariates_std is None else np.asarray(covariates_std, np.float64),
    }
    rank, ws = _dist.world()
    dev_index = _dist.local_device_index() if ws > 1 else 0
    if context_factory is None:
        from bpl._ffi import HipContext

        context_factory = HipContext
    ctx = context_factory(dev_index)
    try:
        bc = _dist.broadcast_fixtures(arrays, device=ctx.device)
        cov = None if bc["covariates"] is None else bc["covariates"].cpu().numpy()
        ctx.set_fixtures(model, bc["home_idx"], bc["away_idx"], bc["home_goals"],
                         bc["away_goals"], n_teams, weights=bc["weights"], covariates_std=cov)
        K = 0 if cov is None else cov.shape[1]
        D = ctx.dim

        cfg = default_nuts_cfg()
        cfg.num_warmup = int(num_warmup)
        cfg.num_samples = int(num_samples)
        cfg.thinning = thinning
        key = prng_key(random_state)
        keys = [key] if num_chains == 1 else threefry_split(key, num_chains)
        z0 = _flatten_init(run_kwargs.get("init_params"), model, n_teams, K)
        if z0 is not None and z0.size == num_chains * D and num_chains > 1:
            z0 = z0.reshape(num_chains, D)

        mine = _dist.chains_of_rank(num_chains, rank, ws)
        kept = cfg.num_samples // thinning
        draws = np.empty((len(mine), kept, D))
        stat_names = ("potential_energy", "accept_prob", "step_size", "num_steps",
                      "diverging", "corr_coef")
        stats = np.empty((len(mine), kept, len(stat_names)))
        scal = np.zeros((len(mine), 4))
        # chains that share this GPU run in lock step (numpyro chain_method="vectorized":
        # one chain-vectorised evaluation per leapfrog of all of them) unless
        # chain_method="sequential" or the bound model does not support it
        results = None
        method = mcmc_kwargs.get("chain_method", "parallel")
        if method not in ("parallel", "sequential", "vectorized"):
            raise ValueError("Only supporting the following methods to draw chains: "
                             '"sequential", "parallel", or "vectorized"')
        if len(mine) > 1 and method != "sequential" and hasattr(ctx, "nuts_run_chains"):
            from bpl._ffi import BPLHIP_EUNSUPPORTED, BplHipError

            zm = None if z0 is None else (z0[list(mine)] if z0.ndim == 2 else z0)
            try:
                results = ctx.nuts_run_chains(cfg, [keys[c] for c in mine], zm)
            except BplHipError as e:
                if e.code != BPLHIP_EUNSUPPORTED:
                    raise
        for j, c in enumerate(mine):
            if results is not None:
                d, st = results[j]
            else:
                zc = None if z0 is None else (z0[c] if z0.ndim == 2 else z0)
                d, st = ctx.nuts_run(cfg, keys[c], zc)
            draws[j] = d
            for i, nm in enumerate(stat_names):
                stats[j, :, i] = st[nm]
            scal[j] = (st["total_leapfrogs"], st["wall_seconds"], st["final_step_size"],
                       st["total_divergences"])
        draws = _dist.gather_chains(draws, num_chains, device=ctx.device)
        stats = _dist.gather_chains(stats, num_chains, device=ctx.device)
        scal = _dist.gather_chains(scal, num_chains, device=ctx.device)

        z = draws.reshape(num_chains * kept, D)  # chain-major, numpyro get_samples order
        samples = constrained_samples(model, n_teams, K, z, ctx.constrain(z))
        info = {
            "num_chains": num_chains,
            "unconstrained": z,
            "total_leapfrogs": int(scal[:, 0].sum()),
            "wall_seconds": float(scal[:, 1].max()),
            "step_size": scal[:, 2].copy(),
            "divergences": int(scal[:, 3].sum()),
        }
        for i, nm in enumerate(stat_names):
            info[nm] = stats[:, :, i].reshape(-1)
        alias = {"adapt_state.step_size": "step_size", "mean_accept_prob": "accept_prob"}
        info["extra_fields"] = {f: info[alias.get(f, f)] for f in extra}
        return samples, info
    finally:
        close = getattr(ctx, "close", None)
        if close:
            close()
