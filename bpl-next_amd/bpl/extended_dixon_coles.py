"""The extended Dixon-Coles model -- drop-in for the reference's
bpl/extended_dixon_coles.py:27-457: per-team home advantage, rho-correlated
attack/defence prior, optional team covariates, optional exponential time weights,
rates clipped at 15.  `_model` (bpl/extended_dixon_coles.py:78-248) + NUTS run in
libbplhip.so; this class is the host-side mirror of the reference interface."""

from __future__ import annotations

import warnings
from typing import Any, Dict, Iterable, Optional, Tuple, Union

import numpy as np

from bpl._ffi import MODEL_EXTENDED
from bpl._mcmc import run_mcmc
from bpl._util import dixon_coles_correlation_term, parse_teams, poisson_log_prob
from bpl.base import DTYPES, BaseMatchPredictor

__all__ = ["ExtendedDixonColesMatchPredictor"]


# pylint: disable=too-many-instance-attributes
class ExtendedDixonColesMatchPredictor(BaseMatchPredictor):
    """A Dixon-Coles like model with correlated attack/defence abilities, per-team home
    advantage, optional team covariates and optional exponential time weighting."""

    # pylint: disable=duplicate-code
    def __init__(self):
        super().__init__()
        self.attack = None
        self.defence = None
        self.home_advantage = None
        self.corr_coef = None
        self.rho = None
        self.attack_coefficients = None
        self.defence_coefficients = None
        self.mean_defence = None
        self.std_defence = None
        self.std_attack = None
        self.mean_home_advantage = None
        self.std_home_advantage = None

        self._team_covariates_mean = None
        self._team_covariates_std = None

        self.epsilon = None
        self.time_diff = None
        self.rescale_weights = None
        self.mcmc_info_ = None

    # pylint: disable=arguments-differ,too-many-arguments,duplicate-code
    def fit(
        self,
        training_data: Dict[str, Union[Iterable[str], Iterable[float]]],
        random_state: int = 42,
        num_warmup: int = 500,
        num_samples: int = 1000,
        epsilon: Optional[float] = None,
        rescale_weights: Optional[bool] = False,
        mcmc_kwargs: Optional[Dict[str, Any]] = None,
        run_kwargs: Optional[Dict[str, Any]] = None,
    ) -> "ExtendedDixonColesMatchPredictor":
        """Fit model to data."""
        self.teams, self._teams_dict, home_ind, away_ind = parse_teams(
            training_data["home_team"], training_data["away_team"], DTYPES["teams"]
        )
        team_covariates = training_data.get("team_covariates", None)

        self.epsilon = epsilon
        self.time_diff = training_data.get("time_diff", None)
        self.rescale_weights = rescale_weights
        if epsilon is not None:
            if self.time_diff is None:
                raise ValueError(
                    "time_diff must be provided in training_data to include "
                    "exponential time decay in model."
                )

        covariates_std = None
        if team_covariates:
            if set(team_covariates.keys()) == set(self.teams):
                team_covariates = np.array(
                    [team_covariates[t] for t in self.teams], dtype=np.float64
                )
                self._team_covariates_mean = team_covariates.mean(axis=0)
                self._team_covariates_std = team_covariates.std(axis=0)
                # standardisation of bpl/extended_dixon_coles.py:124-127 is data-only
                covariates_std = (
                    team_covariates - self._team_covariates_mean
                ) / self._team_covariates_std
            else:
                raise ValueError("team_covariates must contain all the teams in the data.")

        weights = None
        if epsilon is not None:
            # bpl/extended_dixon_coles.py:202-205: parameter independent, so computed once
            time_diff = np.asarray(self.time_diff, dtype=np.float64)
            weights = np.exp(-epsilon * time_diff)
            if rescale_weights:
                weights = time_diff.shape[0] * weights / weights.sum()

        samples, info = run_mcmc(
            MODEL_EXTENDED,
            home_ind,
            away_ind,
            np.array(training_data["home_goals"]),
            np.array(training_data["away_goals"]),
            len(self.teams),
            weights=weights,
            covariates_std=covariates_std,
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
        self.rho = samples["rho"]
        self.attack_coefficients = samples.get("attack_coefficients", None)
        self.defence_coefficients = samples.get("defence_coefficients", None)
        self.mean_defence = samples["mean_defence"]
        self.std_defence = samples["std_defence"]
        self.std_attack = samples["std_attack"]
        self.mean_home_advantage = samples["mean_home_advantage"]
        self.std_home_advantage = samples["std_home_advantage"]
        self.mcmc_info_ = info
        return self

    def _calculate_expected_goals(
        self, home_team: Union[str, Iterable[str]], away_team: Union[str, Iterable[str]]
    ) -> Tuple[np.ndarray, np.ndarray]:
        """Expected goals [draws, fixtures] for home and away teams."""
        home_ind, away_ind = self._parse_fixture_args(home_team, away_team)

        attack_home, defence_home = self.attack[:, home_ind], self.defence[:, home_ind]
        attack_away, defence_away = self.attack[:, away_ind], self.defence[:, away_ind]

        home_rate = np.exp(attack_home - defence_away + self.home_advantage[:, home_ind])
        away_rate = np.exp(attack_away - defence_home)
        return home_rate, away_rate

    def predict_score_proba(
        self,
        home_team: Union[str, Iterable[str]],
        away_team: Union[str, Iterable[str]],
        home_goals: Union[int, Iterable[int]],
        away_goals: Union[int, Iterable[int]],
    ) -> np.ndarray:
        """Return the probability of a particular scoreline."""
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

    def add_new_team(self, team_name: str, team_covariates: Optional[np.ndarray] = None) -> None:
        """Build attack/defence/home_advantage draws for a team not seen in training, from
        the priors (informed by team covariates when coefficients were estimated)."""
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
                team_covariates = (
                    0.5
                    * (np.asarray(team_covariates) - self._team_covariates_mean)
                    / self._team_covariates_std
                )
            mean_attack = np.dot(self.attack_coefficients, team_covariates.ravel())
            mean_defence = self.mean_defence + np.dot(
                self.defence_coefficients, team_covariates.ravel()
            )
        else:
            mean_attack = 0.0
            mean_defence = self.mean_defence

        log_a_tilde = np.random.normal(loc=0.0, scale=1.0, size=len(self.std_attack))
        log_b_tilde = np.random.normal(
            loc=self.rho * log_a_tilde, scale=np.sqrt(1 - self.rho**2.0)
        )
        home_advantage = np.random.normal(
            loc=self.mean_home_advantage, scale=self.std_home_advantage
        )

        attack = mean_attack + log_a_tilde * self.std_attack
        defence = mean_defence + log_b_tilde * self.std_defence

        self.teams = np.append(self.teams, team_name)
        self._teams_dict[team_name] = len(self._teams_dict)
        self.attack = np.concatenate((self.attack, attack[:, None]), axis=1)
        self.defence = np.concatenate((self.defence, defence[:, None]), axis=1)
        self.home_advantage = np.concatenate(
            (self.home_advantage, home_advantage[:, None]), axis=1
        )
