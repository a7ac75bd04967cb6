"""Predict side of the match predictors: everything the reference's BaseMatchPredictor offers
(bpl/base.py:25-348 -- same method names, arguments, return shapes, error behaviour), built on
ONE device primitive instead of re-tiling scoreline queries: the per-fixture scoreline grid
`bplhip_predict_score_grid` (csrc/dc_predict.hip.h, one wave per fixture on the matrix cores).
Outcome probabilities, n-goal marginals and the sampling methods are reductions of that grid;
`predict_score_proba` for arbitrary scorelines uses the pointwise kernel.  Arrays are numpy
(the reference returns jax arrays).  There is no host fallback: the predict path needs the HIP
library and a GPU, like `fit`.
"""

from __future__ import annotations

from datetime import datetime
from typing import Dict, Iterable, Optional, Tuple, Union

import numpy as np

from bpl._util import map_choice

MAX_GOALS = 15
GRID_MAX_GOALS = 63  # depth of the device grid kernel (csrc/dc_predict.hip.h); deeper grids go pointwise
DTYPES = {
    "goals": "uint8",
    "teams": "uint16",
    "conferences": "uint8",
    "venue": "uint8",
    "outcome": "uint8",
}

TeamArg = Union[str, int, Iterable[str], Iterable[int]]


def _prng_key(seed: int):
    """jax.random.PRNGKey(seed) as the (hi, lo) pair the library's threefry takes."""
    seed = int(seed) & 0xFFFFFFFFFFFFFFFF
    return (seed >> 32) & 0xFFFFFFFF, seed & 0xFFFFFFFF


def _wall_clock_seed() -> int:
    # the reference seeds from the clock when random_state is None (bpl/base.py:173-174)
    return int(datetime.now().timestamp() * 100)


def _fingerprint(arrays) -> tuple:
    """Identity of a set of posterior arrays BY CONTENT: shape, dtype and a 128-bit hash of the whole
    buffer -- any in-place edit (`model.attack[:, j] = ...`, a swap of two columns) changes it, and
    neither a recycled `id` nor a temporary made by `np.asarray` enters it.  blake2b runs at ~1 GB/s: 0.6 ms
    for the four [1000, 20] tables of a league fit (the dynamic class hashes one gameweek's slices).
    Plain Python values (the dynamic class's gameweek) are part of the stamp as they are."""
    import hashlib

    out = []
    for a in arrays:
        if a is None or isinstance(a, (int, float, str, bool)):
            out.append(a)
            continue
        a = np.ascontiguousarray(a)
        out.append((a.shape, a.dtype.str, hashlib.blake2b(memoryview(a).cast("B"), digest_size=16).digest()))
    return tuple(out)


class PosteriorOnDevice:
    """The posterior draws of a fitted model, resident on one GPU for the predict kernels.  A mixin:
    the class says which arrays make up its posterior (`_posterior_arrays`) and how a context takes
    them (`_upload_posterior`).  The device context is created on first use, re-fed whenever the
    arrays change (assignment, `add_new_team`, in-place edits: `_fingerprint`), and is NOT part of the
    model's state: a fitted model pickles and deep-copies like the reference's (plain arrays), the
    copy re-creating its context on its first predict call."""

    #: GPU index for the predict path; None = this rank's GPU (bpl._dist) / GPU 0 on one process
    predict_device: Optional[int] = None
    _predict_ctx = None   # bpl._ffi.HipContext holding the uploaded posterior
    _uploaded = None      # fingerprint of the arrays last uploaded

    def _posterior_arrays(self) -> tuple:
        raise NotImplementedError

    def _upload_posterior(self, ctx) -> None:
        raise NotImplementedError

    def invalidate_predict_cache(self) -> None:
        """Force the next predict call to upload the posterior again."""
        self._uploaded = None

    def _device(self):
        """The context with this model's current posterior draws on the GPU."""
        if self._predict_ctx is None:
            from bpl import _dist
            from bpl._ffi import HipContext

            index = self.predict_device
            if index is None:
                index = _dist.local_device_index() if _dist.world()[1] > 1 else 0
            self._predict_ctx = HipContext(int(index))
            self._uploaded = None
        stamp = _fingerprint(self._posterior_arrays())
        if stamp != self._uploaded:
            self._upload_posterior(self._predict_ctx)
            self._uploaded = stamp
        return self._predict_ctx

    def __getstate__(self):
        state = dict(self.__dict__)
        state.pop("_predict_ctx", None)
        state.pop("_uploaded", None)
        return state

    def __deepcopy__(self, memo):
        import copy

        new = self.__class__.__new__(self.__class__)
        memo[id(self)] = new
        for key, value in self.__getstate__().items():
            setattr(new, key, copy.deepcopy(value, memo))
        return new


def grid_from_pointwise(score_proba, n_fixtures: int, max_goals: int) -> np.ndarray:
    """A scoreline grid deeper than the grid kernel goes (max_goals > GRID_MAX_GOALS) through the
    pointwise kernel: `score_proba(fixture_index, x, y)` over every cell."""
    width = max_goals + 1
    x, y = np.divmod(np.arange(width * width), width)
    which = np.repeat(np.arange(n_fixtures), width * width)
    return score_proba(which, np.tile(x, n_fixtures), np.tile(y, n_fixtures)).reshape(n_fixtures, width, width)


class BaseMatchPredictor(PosteriorOnDevice):
    """Common predict API of the team-level models.  A subclass provides `fit` and the four
    posterior arrays (`attack`, `defence` [draws, teams]; `home_advantage` [draws] or
    [draws, teams]; `corr_coef` [draws])."""

    def __init__(self):
        self.teams = None          # sorted unique team names
        self._teams_dict = None    # name -> index

    # ------------------------------------------------------------------ plumbing
    def fit(self, training_data, **kwargs) -> "BaseMatchPredictor":
        raise NotImplementedError("subclasses implement fit()")

    def _team_indices(self, *team_args: TeamArg):
        """Names (or ready indices), scalar or iterable -> uint16 index arrays.  Unknown names
        raise KeyError, as the reference's dictionary lookup does."""
        out = []
        for arg in team_args:
            items = [arg] if isinstance(arg, (str, int, np.integer)) else list(arg)
            out.append(np.fromiter((self._teams_dict[t] if isinstance(t, str) else int(t) for t in items),
                                   dtype=DTYPES["teams"], count=len(items)))
        return out if len(out) > 1 else out[0]

    # (name kept: the reference's helper of the same role, bpl/base.py:62-72)
    def _parse_fixture_args(self, home_team: TeamArg, away_team: TeamArg):
        return tuple(self._team_indices(home_team, away_team))

    def _posterior_arrays(self):
        return (self.attack, self.defence, self.home_advantage, self.corr_coef)

    def _upload_posterior(self, ctx):
        ctx.predict_set_posterior(self.attack, self.defence, self.home_advantage, self.corr_coef)

    def _grid(self, home_idx: np.ndarray, away_idx: np.ndarray, max_goals: int) -> np.ndarray:
        """[fixtures, max_goals+1, max_goals+1]: P(home scores x, away scores y)."""
        max_goals = int(max_goals)
        if max_goals < 0:
            raise ValueError("max_goals must be >= 0")
        dev = self._device()
        if max_goals <= GRID_MAX_GOALS:
            return dev.predict_score_grid(home_idx, away_idx, max_goals)
        return grid_from_pointwise(lambda f, x, y: dev.predict_score_proba(home_idx[f], away_idx[f], x, y),
                                   len(home_idx), max_goals)

    def _calculate_expected_goals(self, home_team: TeamArg, away_team: TeamArg) -> Tuple[np.ndarray, np.ndarray]:
        """Home and away scoring rates, [draws, fixtures] (bpl/dixon_coles.py:126-137,
        bpl/extended_dixon_coles.py:335-358)."""
        h, a = self._team_indices(home_team, away_team)
        edge = self.home_advantage[:, None] if np.ndim(self.home_advantage) == 1 else self.home_advantage[:, h]
        log_home = self.attack[:, h] - self.defence[:, a] + edge
        log_away = self.attack[:, a] - self.defence[:, h]
        return np.exp(log_home), np.exp(log_away)

    # ------------------------------------------------------------------ probabilities
    def predict_score_proba(self, home_team: TeamArg, away_team: TeamArg,
                            home_goals: Union[int, Iterable[int]],
                            away_goals: Union[int, Iterable[int]]) -> np.ndarray:
        """Probability of each requested scoreline: posterior mean of tau * Poisson * Poisson
        (bpl/dixon_coles.py:139-163)."""
        h, a = self._team_indices(home_team, away_team)
        x = np.broadcast_to(np.asarray(home_goals), h.shape)
        y = np.broadcast_to(np.asarray(away_goals), h.shape)
        return self._device().predict_score_proba(h, a, x, y)

    def predict_score_grid_proba(self, home_team: TeamArg, away_team: TeamArg,
                                 max_goals: Optional[int] = MAX_GOALS,
                                 ) -> Tuple[np.ndarray, np.ndarray, np.ndarray]:
        """(probabilities [fixtures, G+1, G+1], home-goal grid, away-goal grid)
        (bpl/base.py:74-111)."""
        h, a = self._team_indices(home_team, away_team)
        counts = np.arange(max_goals + 1)
        return (self._grid(h, a, max_goals),) + tuple(np.meshgrid(counts, counts, indexing="ij"))

    def predict_outcome_proba(self, home_team: TeamArg, away_team: TeamArg,
                              max_goals: Optional[int] = MAX_GOALS) -> Dict[str, np.ndarray]:
        """Home win / draw / away win (bpl/base.py:113-148): the strictly lower triangle, the
        diagonal and the strictly upper triangle of the scoreline grid."""
        h, a = self._team_indices(home_team, away_team)
        grid = self._grid(h, a, max_goals)
        return {
            "home_win": np.tril(grid, -1).sum(axis=(1, 2)),
            "draw": np.trace(grid, axis1=1, axis2=2),
            "away_win": np.triu(grid, 1).sum(axis=(1, 2)),
        }

    def _goal_marginal(self, n, team: TeamArg, opponent: TeamArg, team_is_home: bool,
                       count_team_goals: bool, max_goals: int) -> np.ndarray:
        """P(`team` scores [concedes] n) with the other side's goals summed over 0..max_goals:
        a row or column sum of ONE fixture's grid."""
        wanted = np.atleast_1d(np.asarray(n, dtype=np.int64))
        if wanted.size and wanted.min() < 0:
            raise ValueError("n must be >= 0")
        t, o = self._team_indices(team, opponent)
        depth = max(int(max_goals), int(wanted.max()))
        grid = self._grid(*((t, o) if team_is_home else (o, t)), depth)[0]
        # axis 0 counts the home side's goals: the team's when it is at home and we count its own
        own_axis = 0 if team_is_home == count_team_goals else 1
        other = np.take(grid, np.arange(max_goals + 1), axis=1 - own_axis)
        return other.sum(axis=1 - own_axis)[wanted]

    def predict_score_n_proba(self, n: Union[int, Iterable[int]], team: TeamArg, opponent: TeamArg,
                              home: Optional[bool] = True,
                              max_goals: Optional[int] = MAX_GOALS) -> np.ndarray:
        """Probability that `team` scores n goals against `opponent` (bpl/base.py:248-297)."""
        return self._goal_marginal(n, team, opponent, bool(home), True, max_goals)

    def predict_concede_n_proba(self, n: Union[int, Iterable[int]], team: TeamArg, opponent: TeamArg,
                                home: Optional[bool] = True,
                                max_goals: Optional[int] = MAX_GOALS) -> np.ndarray:
        """Probability that `team` concedes n goals against `opponent` (bpl/base.py:299-348)."""
        return self._goal_marginal(n, team, opponent, bool(home), False, max_goals)

    # ------------------------------------------------------------------ sampling
    def sample_score(self, home_team: TeamArg, away_team: TeamArg, num_samples: int = 1,
                     random_state: int = None, max_goals: Optional[int] = MAX_GOALS,
                     ) -> Dict[str, np.ndarray]:
        """Scorelines drawn from each fixture's grid, [fixtures, num_samples] per side
        (bpl/base.py:150-195): one categorical draw over the flattened grid, then
        cell -> (row, column)."""
        h, a = self._team_indices(home_team, away_team)
        seed = _wall_clock_seed() if random_state is None else random_state
        width = max_goals + 1
        grid = self._grid(h, a, max_goals).reshape(len(h), width * width)
        cell = map_choice(_prng_key(seed), np.arange(width * width, dtype="uint32"), num_samples, grid)
        rows, cols = np.divmod(cell, width)
        return {"home_score": rows.astype(DTYPES["goals"]), "away_score": cols.astype(DTYPES["goals"])}

    def sample_outcome(self, home_team: TeamArg, away_team: TeamArg, num_samples: int = 1,
                       random_state: int = None, max_goals: Optional[int] = MAX_GOALS) -> np.ndarray:
        """Winner's name, or 'Draw', [fixtures, num_samples] (bpl/base.py:197-246)."""
        h, a = self._team_indices(home_team, away_team)
        seed = _wall_clock_seed() if random_state is None else random_state
        p = self.predict_outcome_proba(h, a, max_goals=max_goals)
        table = np.column_stack([p["home_win"], p["draw"], p["away_win"]])
        pick = map_choice(_prng_key(seed), np.arange(3, dtype="uint32"), num_samples, table)
        labels = np.append(self.teams, "Draw")
        draw_slot = len(self.teams)
        # 0 -> the home side's name, 1 -> 'Draw', 2 -> the away side's
        who = np.where(pick == 0, h[:, None], np.where(pick == 2, a[:, None], draw_slot))
        return labels[who]
