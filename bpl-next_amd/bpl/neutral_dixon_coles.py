"""Neutral-venue Dixon-Coles model (host side), SURVEY.md §8 row f-4.

Mirrors the reference's bpl/neutral_dixon_coles.py:30-902 (`NeutralDixonColesMatchPredictor`)
method for method: same names, arguments, return shapes and error behaviour; arrays are
numpy instead of jax.  `fit` drives libbplhip (bplhip_set_fixtures_neutral + bplhip_nuts_run);
the predict methods run on the device like the league models' (bpl/base.py here): ONE primitive,
the per-fixture scoreline grid of `bplhip_predict_score_grid_venue` (csrc/dc_predict.hip.h, the
venue-aware rate form), of which outcomes, n-goal marginals and the sampling methods are
reductions; arbitrary scorelines go through the pointwise kernel.  No host fallback.
"""

from __future__ import annotations

import warnings
from typing import Any, Dict, Iterable, Optional, Tuple, Union

import numpy as np

from bpl import _dist
from bpl._util import map_choice, parse_teams, str_to_list
from bpl.base import DTYPES, GRID_MAX_GOALS, MAX_GOALS, PosteriorOnDevice, _prng_key, _wall_clock_seed, grid_from_pointwise

__all__ = ["NeutralDixonColesMatchPredictor"]

_MCMC_KEYS = {"num_chains", "thinning", "progress_bar", "chain_method", "jit_model_args",
              "postprocess_fn"}
_RUN_KEYS = {"init_params", "extra_fields"}


def latent_sites(T: int, K: int, C: int = 0):
    """(name, size) of every latent site in flat (sorted-name) order; D = 6T + 2K + C + 13
    (C confederations: World-Cup variant)."""
    s = []
    if K:
        s.append(("attack_coefficients", K))
    s += [("away_attack_decentered", T), ("away_defence_decentered", T)]
    if C:
        s.append(("confederation_strength_decentered", C))
    s.append(("corr_coef_raw", 1))
    if K:
        s.append(("defence_coefficients", K))
    s += [("home_attack_decentered", T), ("home_defence_decentered", T),
          ("mean_away_attack", 1), ("mean_away_defence", 1), ("mean_defence", 1),
          ("mean_home_attack", 1), ("mean_home_defence", 1),
          ("standardised_attack", T), ("standardised_defence", T),
          ("std_attack", 1), ("std_away_attack", 1), ("std_away_defence", 1),
          ("std_defence", 1), ("std_home_attack", 1), ("std_home_defence", 1), ("u", 1)]
    return s


def make_weights(n, time_diff, epsilon, game_weights, rescale_weights):
    """bpl/neutral_dixon_coles.py:251-257 (parameter independent, so computed once)."""
    w = np.ones(n)
    if epsilon is not None:
        w = w * np.exp(-epsilon * np.asarray(time_diff, dtype=np.float64))
        if rescale_weights:
            w = n * w / w.sum()
    if game_weights is None:
        # the reference multiplies by `game_weights` unconditionally (:256-257)
        raise TypeError("unsupported operand type(s) for *: 'Array' and 'NoneType' "
                        "(training_data['game_weights'] is required)")
    return w * np.asarray(game_weights, dtype=np.float64)


# pylint: disable=too-many-instance-attributes
class NeutralDixonColesMatchPredictor(PosteriorOnDevice):
    """Dixon-Coles with rho-correlated attack/defence, optional covariates, separate home and
    away attack/defence offsets per team that vanish at neutral venues, time decay and
    per-game weights (see bpl/neutral_dixon_coles.py:30-52)."""

    def __init__(self):
        self.teams = None
        self._teams_dict = None
        for nm in ("attack", "defence", "home_attack", "away_attack", "home_defence", "away_defence",
                   "time_diff", "epsilon", "rescale_weights", "game_weights", "corr_coef", "u", "rho",
                   "attack_coefficients", "defence_coefficients", "mean_attack", "mean_defence",
                   "std_attack", "std_defence", "mean_home_attack", "mean_away_attack",
                   "mean_home_defence", "mean_away_defence", "std_home_attack", "std_away_attack",
                   "std_home_defence", "std_away_defence", "standardised_attack",
                   "standardised_defence", "_team_covariates_mean", "_team_covariates_std",
                   "confederation_strength"):
            setattr(self, nm, None)
        self.mcmc_info_ = None

    # pylint: disable=arguments-differ,too-many-arguments,too-many-statements,too-many-locals
    def fit(
        self,
        training_data: Dict[str, Union[Iterable[str], Iterable[float]]],
        epsilon: Optional[float] = None,
        rescale_weights: Optional[bool] = False,
        random_state: int = 42,
        num_warmup: int = 500,
        num_samples: int = 1000,
        mcmc_kwargs: Optional[Dict[str, Any]] = None,
        run_kwargs: Optional[Dict[str, Any]] = None,
    ) -> "NeutralDixonColesMatchPredictor":
        """Fit model to data (bpl/neutral_dixon_coles.py:286-384)."""
        self.epsilon = epsilon
        self.rescale_weights = rescale_weights
        self.time_diff = training_data.get("time_diff", None)
        if epsilon is not None and self.time_diff is None:
            raise ValueError(
                """
                    time_diff must be provided in training_data
                    to include exponential time decay in model.
                    """
            )
        self.game_weights = training_data.get("game_weights", None)
        n = len(list(training_data["home_goals"]))
        weights = make_weights(n, self.time_diff, epsilon, self.game_weights, rescale_weights)
        return self._fit(training_data, weights, None, random_state, num_warmup, num_samples,
                         mcmc_kwargs, run_kwargs)

    def _fit(self, training_data, weights, conf, random_state, num_warmup, num_samples,
             mcmc_kwargs, run_kwargs):
        """Shared by the neutral and the World-Cup model.  `conf`: None or
        (home_conf_idx, away_conf_idx, n_conf)."""
        from bpl._ffi import HipContext, default_nuts_cfg, prng_key, threefry_split

        self.teams, self._teams_dict, home_ind, away_ind = parse_teams(
            training_data["home_team"], training_data["away_team"], DTYPES["teams"]
        )
        team_covariates = training_data.get("team_covariates")
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
        C = 0 if conf is None else int(conf[2])

        hg = np.asarray(training_data["home_goals"])
        ag = np.asarray(training_data["away_goals"])
        if hg.size and (hg.min() < 0 or ag.min() < 0 or hg.max() > 255 or ag.max() > 255):
            raise ValueError("goals must be integers in [0, 255]")
        nv = np.asarray(training_data["neutral_venue"]).astype(np.uint8)

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
        chain_method = mcmc_kwargs.get("chain_method", "parallel")
        if chain_method not in ("parallel", "sequential", "vectorized"):
            raise ValueError("Only supporting the following methods to draw chains:"
                             ' "sequential", "parallel", or "vectorized"')
        if num_chains < 1 or thinning < 1:
            raise ValueError("num_chains and thinning must be positive")
        rank, ws = _dist.world()
        ctx = HipContext(_dist.local_device_index() if ws > 1 else 0)
        try:
            ctx.set_fixtures_neutral(home_ind, away_ind, hg, ag, nv, T, weights=weights,
                                     covariates_std=cov_std,
                                     home_conf=None if conf is None else conf[0],
                                     away_conf=None if conf is None else conf[1], n_conf=C)
            D = ctx.dim
            cfg = default_nuts_cfg()
            cfg.num_warmup, cfg.num_samples, cfg.thinning = int(num_warmup), int(num_samples), thinning
            key = prng_key(random_state)
            keys = [key] if num_chains == 1 else threefry_split(key, num_chains)
            z0 = run_kwargs.get("init_params")
            if isinstance(z0, dict):
                z0 = np.concatenate([np.asarray(z0[n], dtype=np.float64).reshape(-1)
                                     for n, _ in latent_sites(T, K, C)])
            mine = _dist.chains_of_rank(num_chains, rank, ws)
            kept = cfg.num_samples // thinning
            draws = np.empty((len(mine), kept, D))
            corr = np.empty((len(mine), kept))
            leap = np.zeros((len(mine), 3))
            # init_params: one point for every chain ([D]) or one per chain ([num_chains, D]),
            # sliced per chain like bpl/_mcmc.py:run_mcmc does
            z0a = None
            if z0 is not None:
                z0a = np.asarray(z0, np.float64)
                if z0a.size == D:
                    z0a = np.tile(z0a.reshape(1, D), (num_chains, 1))
                elif z0a.size == num_chains * D:
                    z0a = z0a.reshape(num_chains, D)
                else:
                    raise ValueError(f"init_params must have {D} or {num_chains}x{D} entries, got {z0a.size}")
            results = None
            if len(mine) > 1 and chain_method != "sequential":
                from bpl._ffi import BPLHIP_EUNSUPPORTED, BplHipError

                try:  # chains of this rank run concurrently on the device
                    results = ctx.nuts_run_chains(cfg, [keys[c] for c in mine],
                                                  None if z0a is None else z0a[list(mine)])
                except BplHipError as e:
                    if e.code != BPLHIP_EUNSUPPORTED:
                        raise
            for j, c in enumerate(mine):
                d, st = results[j] if results is not None else ctx.nuts_run(
                    cfg, keys[c], None if z0a is None else z0a[c])
                draws[j], corr[j] = d, st["corr_coef"]
                leap[j] = (st["total_leapfrogs"], st["wall_seconds"], st["total_divergences"])
            draws = _dist.gather_chains(draws, num_chains, device=ctx.device)
            corr = _dist.gather_chains(corr, num_chains, device=ctx.device)
            leap = _dist.gather_chains(leap, num_chains, device=ctx.device)
        finally:
            ctx.close()

        # numpyro get_samples(): constrained latent sites + deterministic sites
        z = draws.reshape(num_chains * kept, D)
        o = 0
        lat = {}
        for name, size in latent_sites(T, K, C):
            v = z[:, o:o + size]
            o += size
            if name.startswith("std_"):
                v = np.exp(v)  # HalfNormal sites: ExpTransform
            elif name in ("u", "corr_coef_raw"):
                v = np.clip(1.0 / (1.0 + np.exp(-v)), np.finfo(np.float32).tiny,
                            1.0 - np.finfo(np.float32).eps)  # Beta sites: SigmoidTransform
            if size == 1 and name.startswith(("mean_", "std_", "u", "corr_coef_raw")):
                v = v[:, 0]
            lat[name] = v
        att_mean, def_mean = 0.0, lat["mean_defence"][:, None]
        if K:
            att_mean = lat["attack_coefficients"] @ cov_std.T
            def_mean = def_mean + lat["defence_coefficients"] @ cov_std.T
        self.attack = att_mean + lat["standardised_attack"] * lat["std_attack"][:, None]
        self.defence = def_mean + lat["standardised_defence"] * lat["std_defence"][:, None]
        for nm in ("home_attack", "away_attack", "home_defence", "away_defence"):
            setattr(self, nm, lat["mean_" + nm][:, None]
                    + lat["std_" + nm][:, None] * lat[nm + "_decentered"])
        if C:  # LocScaleReparam(centered=0) of Normal(0, 1): the value is the decentered site
            self.confederation_strength = lat["confederation_strength_decentered"]
        self.corr_coef = corr.reshape(-1)
        self.u = lat["u"]
        self.rho = 2.0 * lat["u"] - 1.0
        self.attack_coefficients = lat.get("attack_coefficients", None)
        self.defence_coefficients = lat.get("defence_coefficients", None)
        for nm in ("mean_defence", "std_attack", "std_defence", "mean_home_attack",
                   "mean_away_attack", "mean_home_defence", "mean_away_defence", "std_home_attack",
                   "std_home_defence", "std_away_attack", "std_away_defence",
                   "standardised_attack", "standardised_defence"):
            setattr(self, nm, lat[nm])
        self.mcmc_info_ = {"unconstrained": z, "num_chains": num_chains,
                           "total_leapfrogs": int(leap[:, 0].sum()),
                           "wall_seconds": float(leap[:, 1].max()),
                           "divergences": int(leap[:, 2].sum())}
        return self

    def _parse_fixture_args(self, home_team, away_team, neutral_venue):
        home_team, away_team = str_to_list(home_team, away_team)
        neutral_venue = np.array(neutral_venue, DTYPES["venue"])
        if isinstance(home_team[0], str):
            home_team = np.array([self._teams_dict[t] for t in home_team], DTYPES["teams"])
        if isinstance(away_team[0], str):
            away_team = np.array([self._teams_dict[t] for t in away_team], DTYPES["teams"])
        return np.asarray(home_team), np.asarray(away_team), neutral_venue

    # ---- internals on index arrays; `conf` = None or (home_conf_idx, away_conf_idx)
    _VENUE_TABLES = ("attack", "defence", "home_attack", "away_attack", "home_defence", "away_defence")

    def _posterior_arrays(self):
        return tuple(getattr(self, nm) for nm in self._VENUE_TABLES) + (self.corr_coef, self.confederation_strength)

    def _upload_posterior(self, ctx):
        ctx.predict_set_posterior_venue(*(getattr(self, nm) for nm in self._VENUE_TABLES), self.corr_coef,
                                        confederation_strength=self.confederation_strength)

    def _rates(self, home_team, away_team, neutral_venue, conf=None):
        """Scoring rates [draws, fixtures]: the venue offsets count only away from neutral ground."""
        at_home = 1.0 - np.asarray(neutral_venue, dtype=np.float64)
        log_home = (self.attack[:, home_team] - self.defence[:, away_team]
                    + at_home * self.home_attack[:, home_team] - at_home * self.away_defence[:, away_team])
        log_away = (self.attack[:, away_team] - self.defence[:, home_team]
                    + at_home * self.away_attack[:, away_team] - at_home * self.home_defence[:, home_team])
        if conf is not None:
            edge = self.confederation_strength[:, conf[0]] - self.confederation_strength[:, conf[1]]
            log_home, log_away = log_home + edge, log_away - edge
        return np.exp(log_home), np.exp(log_away)

    def _score_proba(self, home_team, away_team, home_goals, away_goals, neutral_venue, conf=None):
        """Posterior-mean probability of each (fixture, scoreline) query: the pointwise kernel."""
        home_team, away_team = np.atleast_1d(home_team), np.atleast_1d(away_team)
        m = max(len(home_team), np.size(home_goals), np.size(away_goals))
        spread = lambda v: np.broadcast_to(np.asarray(v), (m,))
        return self._device().predict_score_proba(
            spread(home_team), spread(away_team), spread(home_goals), spread(away_goals),
            neutral=spread(neutral_venue), conf=None if conf is None else (spread(conf[0]), spread(conf[1])))

    def _grid_probs(self, home_team, away_team, neutral_venue, conf, max_goals) -> np.ndarray:
        """[fixtures, max_goals+1, max_goals+1]: P(home scores x, away scores y)."""
        max_goals = int(max_goals)
        if max_goals < 0:
            raise ValueError("max_goals must be >= 0")
        m = len(home_team)
        nv = np.broadcast_to(np.asarray(neutral_venue), (m,))
        dev = self._device()
        if max_goals <= GRID_MAX_GOALS:
            return dev.predict_score_grid(home_team, away_team, max_goals, neutral=nv, conf=conf)
        pick = (lambda f: None) if conf is None else (lambda f: (np.asarray(conf[0])[f], np.asarray(conf[1])[f]))
        return grid_from_pointwise(
            lambda f, x, y: dev.predict_score_proba(home_team[f], away_team[f], x, y, neutral=nv[f], conf=pick(f)),
            m, max_goals)

    def _grid(self, home_team, away_team, neutral_venue, conf, max_goals):
        counts = np.arange(max_goals + 1)
        return (self._grid_probs(home_team, away_team, neutral_venue, conf, max_goals),
                *np.meshgrid(counts, counts, indexing="ij"))

    def _outcome(self, home_team, away_team, neutral_venue, conf, knockout, max_goals):
        grid = self._grid_probs(home_team, away_team, neutral_venue, conf, max_goals)
        home_win = np.tril(grid, -1).sum(axis=(1, 2))   # home goals (axis 1) > away goals (axis 2)
        away_win = np.triu(grid, 1).sum(axis=(1, 2))
        if knockout:  # no draws: the two wins renormalised
            decided = home_win + away_win
            return {"home_win": home_win / decided, "away_win": away_win / decided}
        return {"home_win": home_win, "draw": np.trace(grid, axis1=1, axis2=2), "away_win": away_win}

    def _sample_score(self, home_team, away_team, neutral_venue, conf, num_samples, random_state,
                      max_goals):
        seed = _wall_clock_seed() if random_state is None else random_state
        width = max_goals + 1
        flat = self._grid_probs(home_team, away_team, neutral_venue, conf, max_goals).reshape(len(home_team), width * width)
        cell = map_choice(_prng_key(seed), np.arange(width * width, dtype="uint32"), num_samples, flat)
        rows, cols = np.divmod(cell, width)
        return {"home_score": rows.astype(DTYPES["goals"]), "away_score": cols.astype(DTYPES["goals"])}

    def _sample_outcome(self, home_team, away_team, neutral_venue, conf, knockout, num_samples,
                        random_state, max_goals):
        seed = _wall_clock_seed() if random_state is None else random_state
        p = self._outcome(home_team, away_team, neutral_venue, conf, knockout, max_goals)
        order = ("home_win", "away_win") if knockout else ("home_win", "draw", "away_win")
        table = np.column_stack([p[k] for k in order])
        pick = map_choice(_prng_key(seed), np.arange(len(order), dtype="uint32"), num_samples, table)
        labels = np.append(self.teams, "Draw")
        home_col, away_col = np.asarray(home_team)[:, None], np.asarray(away_team)[:, None]
        who = np.where(pick == 0, home_col, np.where(pick == len(order) - 1, away_col, len(self.teams)))
        return labels[who]

    def _n_proba(self, n, team, opponent, conf, home, neutral_venue, max_goals, scored: bool):
        """P(`team` scores [concedes] n), the other side's goals summed over 0..max_goals: a row or
        column sum of ONE fixture's grid (the reference sums predict_score_proba over the same cells,
        bpl/neutral_dixon_coles.py:782-902)."""
        wanted = np.atleast_1d(np.asarray(n, dtype=np.int64))
        if wanted.size and wanted.min() < 0:
            raise ValueError("n must be >= 0")
        depth = max(int(max_goals), int(wanted.max()))
        nv = np.atleast_1d(np.asarray(neutral_venue))[:1]
        if home:
            grid = self._grid_probs(team[:1], opponent[:1], nv, conf, depth)[0]
        else:
            grid = self._grid_probs(opponent[:1], team[:1], nv, None if conf is None else (conf[1], conf[0]), depth)[0]
        # axis 0 counts the home side's goals: the team's when it is at home and we count its own
        own_axis = 0 if bool(home) == scored else 1
        other = np.take(grid, np.arange(max_goals + 1), axis=1 - own_axis)
        return other.sum(axis=1 - own_axis)[wanted]

    def _new_team_draws(self, team_name: str, team_covariates):
        """Parameters of a new team drawn from the fitted priors
        (bpl/neutral_dixon_coles.py:490-560)."""
        if team_name in self.teams:
            raise ValueError(f"Team {team_name} already known to model.")
        if self.attack_coefficients is not None:
            if team_covariates is None:
                warnings.warn(
                    f"You haven't provided features for {team_name}."
                    " Assuming team_covariates are the average of known teams."
                    " For better forecasts, provide team_covariates."
                )
                team_covariates = np.zeros(self.attack_coefficients.shape[1])
            else:
                team_covariates = (0.5 * (np.asarray(team_covariates) - self._team_covariates_mean)
                                   / self._team_covariates_std)
            mean_attack = np.dot(self.attack_coefficients, team_covariates.ravel())
            mean_defence = self.mean_defence + np.dot(self.defence_coefficients, team_covariates.ravel())
        else:
            mean_attack = 0.0
            mean_defence = self.mean_defence
        log_a_tilde = np.random.normal(loc=0.0, scale=1.0, size=len(self.std_attack))
        log_b_tilde = np.random.normal(loc=self.rho * log_a_tilde, scale=np.sqrt(1 - self.rho ** 2.0))
        home_attack = np.random.normal(loc=self.mean_home_attack, scale=self.std_home_attack)
        away_attack = np.random.normal(loc=self.mean_away_attack, scale=self.std_away_attack)
        home_defence = np.random.normal(loc=self.mean_home_defence, scale=self.std_home_defence)
        away_defence = np.random.normal(loc=self.mean_away_defence, scale=self.std_away_defence)
        attack = mean_attack + log_a_tilde * self.std_attack
        defence = mean_defence + log_b_tilde * self.std_defence
        self.teams = np.append(self.teams, team_name)
        self._teams_dict[team_name] = len(self._teams_dict)
        self.attack = np.concatenate((self.attack, attack[:, None]), axis=1)
        self.defence = np.concatenate((self.defence, defence[:, None]), axis=1)
        self.home_attack = np.concatenate((self.home_attack, home_attack[:, None]), axis=1)
        self.away_attack = np.concatenate((self.away_attack, away_attack[:, None]), axis=1)
        self.home_defence = np.concatenate((self.home_defence, home_defence[:, None]), axis=1)
        self.away_defence = np.concatenate((self.away_defence, away_defence[:, None]), axis=1)

    # ---- public API (bpl/neutral_dixon_coles.py:399-902)
    def _calculate_expected_goals(self, home_team, away_team, neutral_venue) -> Tuple[np.ndarray, np.ndarray]:
        """Poisson rates of the home and away goals."""
        return self._rates(*self._parse_fixture_args(home_team, away_team, neutral_venue))

    def predict_score_proba(self, home_team, away_team, home_goals, away_goals, neutral_venue) -> np.ndarray:
        """Probability of a particular scoreline between two teams (mean over draws)."""
        h, a, nv = self._parse_fixture_args(home_team, away_team, neutral_venue)
        return self._score_proba(h, a, home_goals, away_goals, nv)

    def add_new_team(self, team_name: str, team_covariates: Optional[np.ndarray] = None):
        """Add another team with parameters drawn from the fitted priors."""
        self._new_team_draws(team_name, team_covariates)

    def predict_score_grid_proba(self, home_team, away_team, neutral_venue,
                                 max_goals: Optional[int] = MAX_GOALS) -> Tuple[np.ndarray, np.ndarray, np.ndarray]:
        """Scoreline probabilities on the (max_goals+1)^2 grid for every fixture."""
        h, a, nv = self._parse_fixture_args(home_team, away_team, neutral_venue)
        return self._grid(h, a, nv, None, max_goals)

    def predict_outcome_proba(self, home_team, away_team, neutral_venue, knockout: bool = False,
                              max_goals: Optional[int] = MAX_GOALS) -> Dict[str, np.ndarray]:
        """Home win, away win and draw probabilities; `knockout` renormalises over the wins."""
        h, a, nv = self._parse_fixture_args(home_team, away_team, neutral_venue)
        return self._outcome(h, a, nv, None, knockout, max_goals)

    def sample_score(self, home_team, away_team, neutral_venue, num_samples: int = 1,
                     random_state: int = None, max_goals: Optional[int] = MAX_GOALS) -> Dict[str, np.ndarray]:
        """Sample scorelines between two teams."""
        h, a, nv = self._parse_fixture_args(home_team, away_team, neutral_venue)
        return self._sample_score(h, a, nv, None, num_samples, random_state, max_goals)

    def sample_outcome(self, home_team, away_team, neutral_venue, knockout: bool = False,
                       num_samples: int = 1, random_state: int = None,
                       max_goals: Optional[int] = MAX_GOALS) -> np.ndarray:
        """Sample the winner ('Draw' unless `knockout`) of matches between two teams."""
        h, a, nv = self._parse_fixture_args(home_team, away_team, neutral_venue)
        return self._sample_outcome(h, a, nv, None, knockout, num_samples, random_state, max_goals)

    def predict_score_n_proba(self, n, team, opponent, home: Optional[bool] = True,
                              neutral_venue: Optional[int] = 0,
                              max_goals: Optional[int] = MAX_GOALS) -> np.ndarray:
        """Probability that `team` scores n goals against `opponent`."""
        t, o, _ = self._parse_fixture_args(team, opponent, neutral_venue)
        return self._n_proba(n, t, o, None, home, neutral_venue, max_goals, scored=True)

    def predict_concede_n_proba(self, n, team, opponent, home: Optional[bool] = True,
                                neutral_venue: Optional[int] = 0,
                                max_goals: Optional[int] = MAX_GOALS) -> np.ndarray:
        """Probability that `team` concedes n goals against `opponent`."""
        t, o, _ = self._parse_fixture_args(team, opponent, neutral_venue)
        return self._n_proba(n, t, o, None, home, neutral_venue, max_goals, scored=False)
