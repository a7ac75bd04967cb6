"""The neutral-venue model with time-varying (per gameweek) parameters -- host-side
mirror of the reference's bpl/dynamic_dixon_coles.py:23-584
(`DynamicNeutralDixonColesMatchPredictor`), which is unfinished there: not exported from
bpl/__init__.py, untested, and its random walk is a no-op because the results of
`attack.at[j].set(...)` are discarded (:192-218).  `fit(random_walk=True)` (default) runs the
INTENDED model, attack[g] = attack[g-1] + standardised_attack[g]*std_attack[g];
`random_walk=False` runs the model as the reference code computes it.  Other defects of the
reference are not reproduced: num_gameweeks = max(gameweek)+1 (reference: max, :287), the
debug prints (:303-307) are dropped, `mean_away_attack` reads its own site (:324 reads
mean_home_attack), the predict side uses the signs of `_model` (:220-231) and takes the
gameweek to predict for (default: the last one).  `_model` + NUTS run in libbplhip.so.
Like the reference, the class is importable from this module, not from `bpl`."""

from __future__ import annotations

import warnings

from typing import Any, Dict, Iterable, Optional, Tuple, Union

import numpy as np

from bpl import _dist
from bpl._ffi import default_nuts_cfg, prng_key, threefry_split
from bpl.base import MAX_GOALS, PosteriorOnDevice

__all__ = ["DynamicNeutralDixonColesMatchPredictor"]

_G = "g"
_GT = "gt"


def latent_sites(G: int, T: int, K: int):
    """(name, shape) in flat (sorted-name) order; D = 7GT + 10G + 2 + 2K."""
    s = []
    if K:
        s.append(("attack_coefficients", (K,)))
    s += [("away_attack_decentered", (G, T)), ("away_defence_decentered", (G, T)),
          ("corr_coef_raw", ())]
    if K:
        s.append(("defence_coefficients", (K,)))
    s += [("home_attack_decentered", (G, T)), ("home_defence_decentered", (G, T)),
          ("mean_away_attack", (G,)), ("mean_away_defence", (G,)), ("mean_defence", ()),
          ("mean_home_attack", (G,)), ("mean_home_defence", (G,)),
          ("standardised_attack", (G, T)), ("standardised_defence", (G, T)),
          ("std_attack", (G,)), ("std_away_attack", (G,)), ("std_away_defence", (G,)),
          ("std_defence", (G,)), ("std_home_attack", (G,)), ("std_home_defence", (G,)),
          ("u", (G, T))]
    return s


# pylint: disable=too-many-instance-attributes
class DynamicNeutralDixonColesMatchPredictor(PosteriorOnDevice):
    """Dixon-Coles with neutral venues, separate home/away attack/defence offsets and a
    random walk of the team strengths over gameweeks."""

    def __init__(self):
        self.teams = None
        self.attack = None        # [S, G, T]
        self.defence = None       # [S, G, T]
        self.home_attack = None
        self.away_attack = None
        self.home_defence = None
        self.away_defence = None
        self.corr_coef = None
        self.u = None
        self.rho = None
        self.attack_coefficients = None
        self.defence_coefficients = None
        self.mean_defence = None
        self.std_defence = None
        self.std_attack = None
        self.mean_home_attack = None
        self.mean_away_attack = None
        self.mean_home_defence = None
        self.mean_away_defence = None
        self.std_home_attack = None
        self.std_away_attack = None
        self.std_home_defence = None
        self.std_away_defence = None
        self.standardised_attack = None
        self.standardised_defence = None
        self._team_covariates_mean = None
        self._team_covariates_std = None
        self.num_gameweeks = None
        self.mcmc_info_ = None

    # pylint: disable=too-many-arguments,too-many-locals
    def fit(
        self,
        training_data: Dict[str, Union[Iterable[str], Iterable[float]]],
        random_state: int = 42,
        num_warmup: int = 500,
        num_samples: int = 1000,
        mcmc_kwargs: Optional[Dict[str, Any]] = None,
        run_kwargs: Optional[Dict[str, Any]] = None,
        random_walk: bool = True,
    ) -> "DynamicNeutralDixonColesMatchPredictor":
        """Fit the model.  training_data keys: home_team, away_team, home_goals,
        away_goals, gameweek (0-based ints), neutral_venue (0/1), optional team_covariates.

        random_walk=True (default) fits the INTENDED model, attack[g] = attack[g-1] + increment;
        the reference's code discards those updates (bpl/dynamic_dixon_coles.py:192-218 assign
        `attack.at[j].set(...)` to nothing), so its posterior is that of random_walk=False.  A
        warning says so once per fit; pass random_walk=False for the reference's behaviour."""
        from bpl._ffi import HipContext

        if random_walk:
            warnings.warn(
                "DynamicNeutralDixonColesMatchPredictor.fit(random_walk=True) fits the intended "
                "random-walk model; the upstream implementation discards the walk (its attack and "
                "defence stay at zero). Results differ from upstream: pass random_walk=False to "
                "reproduce it.", stacklevel=2)

        home_team = list(training_data["home_team"])
        away_team = list(training_data["away_team"])
        team_covariates = training_data.get("team_covariates")
        self.teams = sorted(list(set(home_team) | set(away_team)))
        tidx = {t: i for i, t in enumerate(self.teams)}
        home_ind = np.array([tidx[t] for t in home_team], dtype=np.uint16)
        away_ind = np.array([tidx[t] for t in away_team], dtype=np.uint16)
        T = len(self.teams)

        cov_std = None
        if team_covariates:
            if set(team_covariates.keys()) != set(self.teams):
                raise ValueError("team_covariates must contain all the teams in the data.")
            cov = np.array([team_covariates[t] for t in self.teams], dtype=np.float64)
            self._team_covariates_mean = cov.mean(axis=0)
            self._team_covariates_std = cov.std(axis=0)
            cov_std = (cov - self._team_covariates_mean) / self._team_covariates_std
        K = 0 if cov_std is None else cov_std.shape[1]

        gameweek = np.array(training_data["gameweek"], dtype=int)
        if gameweek.min() < 0:
            raise ValueError("gameweek must be >= 0")
        G = int(gameweek.max()) + 1
        self.num_gameweeks = G
        hg, ag = np.asarray(training_data["home_goals"]), np.asarray(training_data["away_goals"])
        if hg.min() < 0 or ag.min() < 0 or hg.max() > 255 or ag.max() > 255:
            raise ValueError("goals must be integers in [0, 255]")
        nv = np.asarray(training_data["neutral_venue"]).astype(np.uint8)

        mcmc_kwargs = dict(mcmc_kwargs or {})
        run_kwargs = dict(run_kwargs or {})
        num_chains = int(mcmc_kwargs.get("num_chains", 1))
        thinning = int(mcmc_kwargs.get("thinning", 1))
        rank, ws = _dist.world()
        ctx = HipContext(_dist.local_device_index() if ws > 1 else 0)
        try:
            ctx.set_fixtures_dynamic(home_ind, away_ind, hg, ag, gameweek, nv, T, G,
                                     covariates_std=cov_std, random_walk=random_walk)
            D = ctx.dim
            cfg = default_nuts_cfg()
            cfg.num_warmup, cfg.num_samples, cfg.thinning = int(num_warmup), int(num_samples), thinning
            key = prng_key(random_state)
            keys = [key] if num_chains == 1 else threefry_split(key, num_chains)
            z0 = run_kwargs.get("init_params")
            if isinstance(z0, dict):
                z0 = np.concatenate([np.asarray(z0[n], dtype=np.float64).reshape(-1)
                                     for n, _ in latent_sites(G, T, K)])
            mine = _dist.chains_of_rank(num_chains, rank, ws)
            kept = cfg.num_samples // thinning
            draws = np.empty((len(mine), kept, D))
            corr = np.empty((len(mine), kept))
            leap = np.zeros((len(mine), 2))
            for j, c in enumerate(mine):
                d, st = ctx.nuts_run(cfg, keys[c], None if z0 is None else np.asarray(z0, np.float64))
                draws[j], corr[j] = d, st["corr_coef"]
                leap[j] = (st["total_leapfrogs"], st["wall_seconds"])
            draws = _dist.gather_chains(draws, num_chains, device=ctx.device)
            corr = _dist.gather_chains(corr, num_chains, device=ctx.device)
            leap = _dist.gather_chains(leap, num_chains, device=ctx.device)
            z = draws.reshape(num_chains * kept, D)
            sites = ctx.constrain_dynamic(z)
        finally:
            ctx.close()

        # constrained latent sites (numpyro get_samples): exp for HalfNormal, sigmoid for Beta/Uniform
        o = 0
        lat = {}
        for name, shape in latent_sites(G, T, K):
            n = int(np.prod(shape)) if shape else 1
            v = z[:, o:o + n].reshape((z.shape[0],) + tuple(shape))
            o += n
            if name.startswith("std_"):
                v = np.exp(v)
            elif name in ("u", "corr_coef_raw"):
                v = np.clip(1.0 / (1.0 + np.exp(-v)), np.finfo(np.float32).tiny,
                            1.0 - np.finfo(np.float32).eps)
            lat[name] = v
        self.attack, self.defence = sites["attack"], sites["defence"]
        self.home_attack, self.away_attack = sites["home_attack"], sites["away_attack"]
        self.home_defence, self.away_defence = sites["home_defence"], sites["away_defence"]
        self.corr_coef = corr.reshape(-1)
        self.u = lat["u"]
        self.rho = 2.0 * lat["u"] - 1.0
        self.attack_coefficients = lat.get("attack_coefficients")
        self.defence_coefficients = lat.get("defence_coefficients")
        for nm in ("mean_defence", "std_attack", "std_defence", "mean_home_attack",
                   "mean_away_attack", "mean_home_defence", "mean_away_defence",
                   "std_home_attack", "std_away_attack", "std_home_defence", "std_away_defence",
                   "standardised_attack", "standardised_defence"):
            setattr(self, nm, lat[nm])
        self.mcmc_info_ = {"unconstrained": z, "total_leapfrogs": int(leap[:, 0].sum()),
                           "wall_seconds": float(leap[:, 1].max())}
        return self

    # ---- predict side: the tables of ONE gameweek through the venue-aware device kernels
    # (csrc/dc_predict.hip.h, the same entry points the neutral-venue classes use)
    _VENUE_TABLES = ("attack", "defence", "home_attack", "away_attack", "home_defence", "away_defence")
    _predict_gameweek = None   # the gameweek whose tables the device holds (part of the upload stamp, a plain int)

    def _posterior_arrays(self):
        g = self._predict_gameweek
        return tuple(getattr(self, nm)[:, g, :] for nm in self._VENUE_TABLES) + (self.corr_coef, int(g))

    def _upload_posterior(self, ctx):
        g = self._predict_gameweek
        ctx.predict_set_posterior_venue(*(getattr(self, nm)[:, g, :] for nm in self._VENUE_TABLES), self.corr_coef)

    def __getstate__(self):
        state = super().__getstate__()
        state.pop("_predict_gameweek", None)   # (a cache key of the device side, not model state)
        return state

    def _week(self, gameweek: Optional[int]) -> int:
        g = self.num_gameweeks - 1 if gameweek is None else int(gameweek)
        if not 0 <= g < self.num_gameweeks:
            raise IndexError(f"gameweek {g} outside 0..{self.num_gameweeks - 1}")
        return g

    def _fixture_indices(self, home_team, away_team):
        home_team = [home_team] if isinstance(home_team, str) else list(home_team)
        away_team = [away_team] if isinstance(away_team, str) else list(away_team)
        return (np.array([self.teams.index(t) for t in home_team], dtype=np.uint16),
                np.array([self.teams.index(t) for t in away_team], dtype=np.uint16))

    def _calculate_expected_goals(self, home_team, away_team, neutral_venue,
                                  gameweek: Optional[int] = None) -> Tuple[np.ndarray, np.ndarray]:
        g = self._week(gameweek)
        h, a = self._fixture_indices(home_team, away_team)
        at_home = 1.0 - np.asarray(neutral_venue, dtype=np.float64)
        # signs as in `_model` (bpl/dynamic_dixon_coles.py:220-231)
        log_home = (self.attack[:, g, h] - self.defence[:, g, a]
                    + at_home * self.home_attack[:, g, h] - at_home * self.away_defence[:, g, a])
        log_away = (self.attack[:, g, a] - self.defence[:, g, h]
                    + at_home * self.away_attack[:, g, a] - at_home * self.home_defence[:, g, h])
        return np.exp(log_home), np.exp(log_away)

    def predict_score_proba(self, home_team, away_team, home_goals, away_goals, neutral_venue,
                            gameweek: Optional[int] = None) -> np.ndarray:
        """Probabilities of the given scorelines (mean over posterior draws)."""
        self._predict_gameweek = self._week(gameweek)
        h, a = self._fixture_indices(home_team, away_team)
        m = max(len(h), np.size(home_goals), np.size(away_goals))
        spread = lambda v: np.broadcast_to(np.asarray(v), (m,))
        return self._device().predict_score_proba(spread(h), spread(a), spread(home_goals), spread(away_goals),
                                                  neutral=spread(neutral_venue))

    def predict_outcome_proba(self, home_team, away_team, neutral_venue,
                              gameweek: Optional[int] = None) -> Dict[str, np.ndarray]:
        """Home win, draw and away win probabilities: the triangles of each fixture's scoreline grid."""
        self._predict_gameweek = self._week(gameweek)
        h, a = self._fixture_indices(home_team, away_team)
        grid = self._device().predict_score_grid(h, a, MAX_GOALS, neutral=np.broadcast_to(np.asarray(neutral_venue), (len(h),)))
        return {"home_win": np.tril(grid, -1).sum(axis=(1, 2)), "draw": np.trace(grid, axis1=1, axis2=2),
                "away_win": np.triu(grid, 1).sum(axis=(1, 2))}
