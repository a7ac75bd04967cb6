"""World-Cup variant of the neutral-venue model (host side), SURVEY.md §8 row f-4.

Mirrors the reference's bpl/neutral_dixon_coles_WC.py:21-968
(`NeutralDixonColesMatchPredictorWC`): the neutral model plus a per-confederation strength
~ N(0,1) entering both rates (:188-203), always time-weighted (:206-208).  Same method
names, argument order and return shapes; arrays are numpy instead of jax.
"""

from __future__ import annotations

from typing import Any, Dict, Iterable, Optional, Tuple, Union

import numpy as np

from bpl._util import str_to_list
from bpl.base import DTYPES, MAX_GOALS
from bpl.neutral_dixon_coles import NeutralDixonColesMatchPredictor

__all__ = ["NeutralDixonColesMatchPredictorWC"]


class NeutralDixonColesMatchPredictorWC(NeutralDixonColesMatchPredictor):
    """Neutral-venue Dixon-Coles with confederation (league) strengths."""

    def __init__(self):
        super().__init__()
        self.conferences = None
        self._conferences_dict = None
        self.conferences_ref = None
        self.confederation_strength = None

    # pylint: disable=arguments-differ,too-many-arguments
    def fit(
        self,
        training_data: Dict[str, Union[Iterable[str], Iterable[float]]],
        epsilon: float = 0.0,
        rescale_weights: Optional[bool] = False,
        random_state: int = 42,
        num_warmup: int = 500,
        num_samples: int = 1000,
        mcmc_kwargs: Optional[Dict[str, Any]] = None,
        run_kwargs: Optional[Dict[str, Any]] = None,
    ) -> "NeutralDixonColesMatchPredictorWC":
        """Fit the model (bpl/neutral_dixon_coles_WC.py:235-336)."""
        home_team_conf = training_data["home_conf"]
        away_team_conf = training_data["away_conf"]
        self.conferences = np.array(sorted(set(home_team_conf) | set(away_team_conf)))
        self._conferences_dict = {c: i for i, c in enumerate(self.conferences)}
        # lookup for what each number represents
        self.conferences_ref = dict(zip(range(len(self.conferences)), self.conferences))
        home_conf_ind = np.array([self._conferences_dict[hc] for hc in home_team_conf],
                                 DTYPES["conferences"])
        away_conf_ind = np.array([self._conferences_dict[ac] for ac in away_team_conf],
                                 DTYPES["conferences"])
        self.epsilon = epsilon
        self.rescale_weights = rescale_weights
        self.time_diff = training_data["time_diff"]
        self.game_weights = training_data["game_weights"]
        # weights (:206-208): always time-weighted, rescaled after the game weights
        n = len(home_conf_ind)
        weights = (np.exp(-epsilon * np.asarray(self.time_diff, dtype=np.float64))
                   * np.asarray(self.game_weights, dtype=np.float64))
        if rescale_weights:
            weights = n * weights / weights.sum()
        return self._fit(training_data, weights, (home_conf_ind, away_conf_ind, len(self.conferences)),
                         random_state, num_warmup, num_samples, mcmc_kwargs, run_kwargs)

    def _parse_fixture_args(self, home_team, away_team, home_conf, away_conf, neutral_venue):
        home_team, away_team, home_conf, away_conf = str_to_list(home_team, away_team, home_conf, away_conf)
        neutral_venue = np.array(neutral_venue, DTYPES["venue"])
        if isinstance(home_team[0], str):
            home_team = np.array([self._teams_dict[t] for t in home_team], DTYPES["teams"])
        if isinstance(away_team[0], str):
            away_team = np.array([self._teams_dict[t] for t in away_team], DTYPES["teams"])
        if isinstance(home_conf[0], str):
            home_conf = np.array([self._conferences_dict[hc] for hc in home_conf], DTYPES["conferences"])
        if isinstance(away_conf[0], str):
            away_conf = np.array([self._conferences_dict[ac] for ac in away_conf], DTYPES["conferences"])
        return (np.asarray(home_team), np.asarray(away_team), np.asarray(home_conf),
                np.asarray(away_conf), neutral_venue)

    def _calculate_expected_goals(self, home_team, away_team, home_conf, away_conf,
                                  neutral_venue) -> Tuple[np.ndarray, np.ndarray]:
        """Poisson rates of the home and away goals (:363-422)."""
        h, a, hc, ac, nv = self._parse_fixture_args(home_team, away_team, home_conf, away_conf, neutral_venue)
        return self._rates(h, a, nv, (hc, ac))

    def predict_score_proba(self, home_team, away_team, home_conf, away_conf, home_goals, away_goals,
                            neutral_venue) -> np.ndarray:
        """Probability of a particular scoreline between two teams."""
        h, a, hc, ac, nv = self._parse_fixture_args(home_team, away_team, home_conf, away_conf, neutral_venue)
        return self._score_proba(h, a, home_goals, away_goals, nv, (hc, ac))

    def add_new_team(self, team_name: str, team_covariates: Optional[np.ndarray] = None):
        """Add another team with parameters drawn from the fitted priors (:476-546)."""
        self._new_team_draws(team_name, team_covariates)

    def predict_score_grid_proba(self, home_team, away_team, home_conf, away_conf, neutral_venue,
                                 max_goals: Optional[int] = MAX_GOALS):
        """Scoreline probabilities on the (max_goals+1)^2 grid for every fixture."""
        h, a, hc, ac, nv = self._parse_fixture_args(home_team, away_team, home_conf, away_conf, neutral_venue)
        return self._grid(h, a, nv, (hc, ac), max_goals)

    def predict_outcome_proba(self, home_team, away_team, home_conf, away_conf, neutral_venue,
                              knockout: bool = False, max_goals: Optional[int] = MAX_GOALS):
        """Home win, away win and draw probabilities; `knockout` renormalises over the wins."""
        h, a, hc, ac, nv = self._parse_fixture_args(home_team, away_team, home_conf, away_conf, neutral_venue)
        return self._outcome(h, a, nv, (hc, ac), knockout, max_goals)

    def sample_score(self, home_team, away_team, home_conf, away_conf, neutral_venue,
                     num_samples: int = 1, random_state: int = None,
                     max_goals: Optional[int] = MAX_GOALS):
        """Sample scorelines between two teams."""
        h, a, hc, ac, nv = self._parse_fixture_args(home_team, away_team, home_conf, away_conf, neutral_venue)
        return self._sample_score(h, a, nv, (hc, ac), num_samples, random_state, max_goals)

    def sample_outcome(self, home_team, away_team, home_conf, away_conf, neutral_venue,
                       knockout: bool = False, num_samples: int = 1, random_state: int = None,
                       max_goals: Optional[int] = MAX_GOALS):
        """Sample the winner ('Draw' unless `knockout`) of matches between two teams."""
        h, a, hc, ac, nv = self._parse_fixture_args(home_team, away_team, home_conf, away_conf, neutral_venue)
        return self._sample_outcome(h, a, nv, (hc, ac), knockout, num_samples, random_state, max_goals)

    def predict_score_n_proba(self, n, team, opponent, team_conf, opponent_conf,
                              home: Optional[bool] = True, neutral_venue: Optional[int] = 0,
                              max_goals: Optional[int] = MAX_GOALS) -> np.ndarray:
        """Probability that `team` scores n goals against `opponent`."""
        t, o, tc, oc, _ = self._parse_fixture_args(team, opponent, team_conf, opponent_conf, neutral_venue)
        return self._n_proba(n, t, o, (tc, oc), home, neutral_venue, max_goals, scored=True)

    def predict_concede_n_proba(self, n, team, opponent, team_conf, opponent_conf,
                                home: Optional[bool] = True, neutral_venue: Optional[int] = 0,
                                max_goals: Optional[int] = MAX_GOALS) -> np.ndarray:
        """Probability that `team` concedes n goals against `opponent`."""
        t, o, tc, oc, _ = self._parse_fixture_args(team, opponent, team_conf, opponent_conf, neutral_venue)
        return self._n_proba(n, t, o, (tc, oc), home, neutral_venue, max_goals, scored=False)
