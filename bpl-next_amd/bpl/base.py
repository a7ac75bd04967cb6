"""Base class of the match predictors (host side).

Mirrors the reference's bpl/base.py:25-348 method for method (same names, arguments,
return shapes and error behaviour); arrays are numpy instead of jax.  Everything here
is post-processing of posterior draws -- it runs once per predict_* call, not per
leapfrog -- and stays on the host in this round.
"""

from __future__ import annotations

from abc import abstractmethod
from datetime import datetime
from typing import Dict, Iterable, Optional, Tuple, Union

import numpy as np

from bpl._util import map_choice, str_to_list

MAX_GOALS = 15
DTYPES = {
    "goals": "uint8",
    "teams": "uint16",
    "conferences": "uint8",
    "venue": "uint8",
    "outcome": "uint8",
}


def _prng_key(seed: int):
    seed = int(seed) & 0xFFFFFFFFFFFFFFFF
    return (seed >> 32) & 0xFFFFFFFF, seed & 0xFFFFFFFF


class BaseMatchPredictor:
    """Abstract class for models of football matches."""

    def __init__(self):
        # unique team names (sorted) and the name -> integer index map
        self.teams = None
        self._teams_dict = None
        # True: predict_score_proba (and everything built on it) runs on the GPU through
        # libbplhip's predict kernel instead of host numpy (same results to ~1e-13)
        self.predict_on_device = False
        self._predict_ctx = None
        self._predict_key = None

    def _device_score_proba(self, home_ind, away_ind, home_goals, away_goals) -> np.ndarray:
        """predict_score_proba on the device for index arrays; scalars are broadcast."""
        from bpl._ffi import HipContext

        m = len(home_ind)
        hg = np.broadcast_to(np.asarray(home_goals), (m,)) if np.ndim(home_goals) == 0 else np.asarray(home_goals)
        ag = np.broadcast_to(np.asarray(away_goals), (m,)) if np.ndim(away_goals) == 0 else np.asarray(away_goals)
        if self._predict_ctx is None:
            self._predict_ctx = HipContext(0)
        key = (id(self.attack), id(self.defence), id(self.home_advantage), id(self.corr_coef),
               np.shape(self.attack))
        if key != self._predict_key:
            self._predict_ctx.predict_set_posterior(self.attack, self.defence,
                                                    self.home_advantage, self.corr_coef)
            self._predict_key = key
        return self._predict_ctx.predict_score_proba(home_ind, away_ind, hg, ag)

    @abstractmethod
    def fit(
        self, training_data: Dict[str, Union[Iterable[str], Iterable[float]]], **kwargs
    ) -> "BaseMatchPredictor":
        """Fit the model to data and return self."""

    @abstractmethod
    def predict_score_proba(
        self,
        home_team: Union[str, Iterable[str]],
        away_team: Union[str, Iterable[str]],
        home_goals: Union[int, Iterable[int]],
        away_goals: Union[int, Iterable[int]],
    ) -> np.ndarray:
        """Return the probability of a particular scoreline."""

    def _parse_fixture_args(self, home_team, away_team):
        home_team, away_team = str_to_list(home_team, away_team)
        if isinstance(home_team[0], str):
            home_team = np.array([self._teams_dict[t] for t in home_team], DTYPES["teams"])
        if isinstance(away_team[0], str):
            away_team = np.array([self._teams_dict[t] for t in away_team], DTYPES["teams"])
        return home_team, away_team

    def predict_score_grid_proba(
        self,
        home_team: Union[str, Iterable[str]],
        away_team: Union[str, Iterable[str]],
        max_goals: Optional[int] = MAX_GOALS,
    ) -> Tuple[np.ndarray, np.ndarray, np.ndarray]:
        """Scoreline probabilities on the (max_goals+1)^2 grid for each fixture."""
        home_team, away_team = self._parse_fixture_args(home_team, away_team)

        n_goals = np.arange(0, max_goals + 1)
        home_goals, away_goals = np.meshgrid(n_goals, n_goals, indexing="ij")
        home_goals_flat = np.tile(home_goals.reshape((max_goals + 1) ** 2), len(home_team))
        away_goals_flat = np.tile(away_goals.reshape((max_goals + 1) ** 2), len(home_team))
        home_team_rep = np.repeat(home_team, (max_goals + 1) ** 2)
        away_team_rep = np.repeat(away_team, (max_goals + 1) ** 2)

        probs = self.predict_score_proba(
            home_team_rep, away_team_rep, home_goals_flat, away_goals_flat
        ).reshape(len(home_team), max_goals + 1, max_goals + 1)
        return probs, home_goals, away_goals

    def predict_outcome_proba(
        self,
        home_team: Union[str, Iterable[str]],
        away_team: Union[str, Iterable[str]],
        max_goals: Optional[int] = MAX_GOALS,
    ) -> Dict[str, np.ndarray]:
        """Home win, draw and away win probabilities."""
        home_team, away_team = self._parse_fixture_args(home_team, away_team)
        probs, home_goals, away_goals = self.predict_score_grid_proba(
            home_team, away_team, max_goals=max_goals
        )
        home_win = probs[:, home_goals > away_goals].sum(axis=-1)
        draw = probs[:, home_goals == away_goals].sum(axis=-1)
        away_win = probs[:, home_goals < away_goals].sum(axis=-1)
        return {"home_win": home_win, "draw": draw, "away_win": away_win}

    def sample_score(
        self,
        home_team: Union[str, Iterable[str]],
        away_team: Union[str, Iterable[str]],
        num_samples: int = 1,
        random_state: int = None,
        max_goals: Optional[int] = MAX_GOALS,
    ) -> Dict[str, np.ndarray]:
        """Sample scorelines between two teams."""
        home_team, away_team = self._parse_fixture_args(home_team, away_team)
        if random_state is None:
            random_state = int(datetime.now().timestamp() * 100)

        probs, home_goals, away_goals = self.predict_score_grid_proba(
            home_team, away_team, max_goals=max_goals
        )
        home_goals = np.array(home_goals.flatten(), DTYPES["goals"])
        away_goals = np.array(away_goals.flatten(), DTYPES["goals"])

        sample_idx = map_choice(
            _prng_key(random_state),
            np.arange(len(home_goals), dtype="uint32"),
            num_samples,
            probs.reshape((len(home_team), -1)),
        )
        return {"home_score": home_goals[sample_idx], "away_score": away_goals[sample_idx]}

    def sample_outcome(
        self,
        home_team: Union[str, Iterable[str]],
        away_team: Union[str, Iterable[str]],
        num_samples: int = 1,
        random_state: int = None,
        max_goals: Optional[int] = MAX_GOALS,
    ) -> np.ndarray:
        """Sample the winner ('Draw' for a draw) of matches between two teams."""
        home_team, away_team = self._parse_fixture_args(home_team, away_team)
        if random_state is None:
            random_state = int(datetime.now().timestamp() * 100)

        probs = self.predict_outcome_proba(home_team, away_team, max_goals=max_goals)
        probs = np.array([probs["home_win"], probs["draw"], probs["away_win"]]).T

        sample_idx = map_choice(
            _prng_key(random_state), np.arange(probs.shape[1], dtype="uint32"), num_samples, probs
        )

        home_team = np.asarray(home_team)
        away_team = np.asarray(away_team)
        winner = np.empty((len(home_team), num_samples), dtype=DTYPES["teams"])
        home_team_rep = home_team.repeat(num_samples).reshape((len(home_team), num_samples))
        away_team_rep = away_team.repeat(num_samples).reshape((len(home_team), num_samples))
        winner[sample_idx == 0] = home_team_rep[sample_idx == 0]
        winner[sample_idx == 2] = away_team_rep[sample_idx == 2]
        winner[sample_idx == 1] = len(self.teams)  # temporary index for 'Draw'

        _teams_with_draw = np.append(self.teams, "Draw")
        return _teams_with_draw[winner]

    def predict_score_n_proba(
        self,
        n: Union[int, Iterable[int]],
        team: Union[str, Iterable[str]],
        opponent: Union[str, Iterable[str]],
        home: Optional[bool] = True,
        max_goals: Optional[int] = MAX_GOALS,
    ) -> np.ndarray:
        """Probability that `team` scores n goals against `opponent`."""
        n = [n] if isinstance(n, (int, np.integer)) else n
        team, opponent = self._parse_fixture_args(team, opponent)
        team_rep = np.repeat(team, (max_goals + 1) * len(n))
        opponent_rep = np.repeat(opponent, (max_goals + 1) * len(n))
        n_rep = np.resize(n, (max_goals + 1) * len(n))
        x_rep = np.repeat(np.arange(max_goals + 1), len(n))

        probs = (
            self.predict_score_proba(team_rep, opponent_rep, n_rep, x_rep)
            if home
            else self.predict_score_proba(opponent_rep, team_rep, x_rep, n_rep)
        ).reshape(max_goals + 1, len(n))
        return probs.sum(axis=0)

    def predict_concede_n_proba(
        self,
        n: Union[int, Iterable[int]],
        team: Union[str, Iterable[str]],
        opponent: Union[str, Iterable[str]],
        home: Optional[bool] = True,
        max_goals: Optional[int] = MAX_GOALS,
    ) -> np.ndarray:
        """Probability that `team` concedes n goals against `opponent`."""
        n = [n] if isinstance(n, (int, np.integer)) else n
        team, opponent = self._parse_fixture_args(team, opponent)
        team_rep = np.repeat(team, (max_goals + 1) * len(n))
        opponent_rep = np.repeat(opponent, (max_goals + 1) * len(n))
        n_rep = np.resize(n, (max_goals + 1) * len(n))
        x_rep = np.repeat(np.arange(max_goals + 1), len(n))

        probs = (
            self.predict_score_proba(team_rep, opponent_rep, x_rep, n_rep)
            if home
            else self.predict_score_proba(opponent_rep, team_rep, n_rep, x_rep)
        ).reshape(max_goals + 1, len(n))
        return probs.sum(axis=0)
