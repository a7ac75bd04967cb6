"""A simple team-level Dixon-Coles model -- drop-in for the reference's
bpl/dixon_coles.py:26-163.  `fit()` keeps the reference signature; the model function
(`_model`, bpl/dixon_coles.py:39-84) and numpyro's NUTS are replaced by the HIP path in
libbplhip.so (dc_stream / dc_epilogue kernels + the C++ NUTS driver)."""

from __future__ import annotations

from typing import Any, Dict, Iterable, Optional, Tuple, Union

import numpy as np

from bpl._ffi import MODEL_BASIC
from bpl._mcmc import run_mcmc
from bpl._util import dixon_coles_correlation_term, parse_teams, poisson_log_prob
from bpl.base import DTYPES, BaseMatchPredictor

__all__ = ["DixonColesMatchPredictor"]


class DixonColesMatchPredictor(BaseMatchPredictor):
    """A Dixon-Coles like model for predicting match outcomes."""

    # pylint: disable=duplicate-code
    def __init__(self):
        super().__init__()
        self.attack = None
        self.defence = None
        self.home_advantage = None
        self.corr_coef = None
        self.mcmc_info_ = None

    # pylint: disable=arguments-differ,too-many-arguments,duplicate-code
    def fit(
        self,
        training_data: Dict[str, Union[Iterable[str], Iterable[float]]],
        random_state: int = 42,
        num_warmup: int = 500,
        num_samples: int = 1000,
        mcmc_kwargs: Optional[Dict[str, Any]] = None,
        run_kwargs: Optional[Dict[str, Any]] = None,
    ) -> "DixonColesMatchPredictor":
        self.teams, self._teams_dict, home_ind, away_ind = parse_teams(
            training_data["home_team"], training_data["away_team"], DTYPES["teams"]
        )
        samples, info = run_mcmc(
            MODEL_BASIC,
            home_ind,
            away_ind,
            np.array(training_data["home_goals"]),
            np.array(training_data["away_goals"]),
            len(self.teams),
            random_state=random_state,
            num_warmup=num_warmup,
            num_samples=num_samples,
            mcmc_kwargs=mcmc_kwargs,
            run_kwargs=run_kwargs,
        )
        self.attack = samples["attack"]
        self.defence = samples["defence"]
        self.home_advantage = samples["home_advantage"]
        self.corr_coef = samples["corr_coef"]
        self.mcmc_info_ = info
        return self

    def _calculate_expected_goals(
        self, home_team: Union[str, Iterable[str]], away_team: Union[str, Iterable[str]]
    ) -> Tuple[np.ndarray, np.ndarray]:
        home_ind, away_ind = self._parse_fixture_args(home_team, away_team)

        attack_home, defence_home = self.attack[:, home_ind], self.defence[:, home_ind]
        attack_away, defence_away = self.attack[:, away_ind], self.defence[:, away_ind]

        home_rate = np.exp(attack_home - defence_away + self.home_advantage[:, None])
        away_rate = np.exp(attack_away - defence_home)
        return home_rate, away_rate

    def predict_score_proba(
        self,
        home_team: Union[str, Iterable[str]],
        away_team: Union[str, Iterable[str]],
        home_goals: Union[int, Iterable[int]],
        away_goals: Union[int, Iterable[int]],
    ) -> np.ndarray:
        home_team, away_team = self._parse_fixture_args(home_team, away_team)
        if self.predict_on_device:
            return self._device_score_proba(home_team, away_team, home_goals, away_goals)

        expected_home_goals, expected_away_goals = self._calculate_expected_goals(
            home_team, away_team
        )
        corr_term = dixon_coles_correlation_term(
            home_goals, away_goals, expected_home_goals, expected_away_goals, self.corr_coef
        )
        home_probs = np.exp(poisson_log_prob(expected_home_goals, home_goals))
        away_probs = np.exp(poisson_log_prob(expected_away_goals, away_goals))

        sampled_probs = np.exp(corr_term) * home_probs * away_probs
        return sampled_probs.mean(axis=0)
