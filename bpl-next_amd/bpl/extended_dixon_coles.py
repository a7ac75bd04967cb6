"""The extended Dixon-Coles model behind the reference's class name, `fit` signature and
attributes (bpl/extended_dixon_coles.py:27-457): per-team home advantage, rho-correlated
attack / defence prior, optional team covariates, optional exponential time weights, rates
clipped at 15.  `_model` (:78-248) and NUTS run in libbplhip.so; everything that depends on
the data only -- covariate standardisation (:124-127), the time weights (:202-205) -- is
computed here once and handed to the library."""

from __future__ import annotations

import warnings
from typing import Any, Dict, Iterable, Optional, Union

import numpy as np

from bpl._ffi import MODEL_EXTENDED
from bpl._mcmc import run_mcmc
from bpl._util import parse_teams
from bpl.base import BaseMatchPredictor, DTYPES

__all__ = ["ExtendedDixonColesMatchPredictor"]
TrainingData = Dict[str, Union[Iterable[str], Iterable[float]]]

# posterior sites kept as attributes (bpl/extended_dixon_coles.py:319-331); the two coefficient
# blocks exist only when team covariates were given
_KEPT_SITES = ("attack", "defence", "home_advantage", "corr_coef", "rho", "mean_defence", "std_defence",
               "std_attack", "mean_home_advantage", "std_home_advantage")
_OPTIONAL_SITES = ("attack_coefficients", "defence_coefficients")


# pylint: disable=too-many-instance-attributes
class ExtendedDixonColesMatchPredictor(BaseMatchPredictor):
    """Dixon-Coles with correlated abilities, per-team home advantage, team covariates and
    time-decayed likelihood weights."""

    def __init__(self):
        super().__init__()
        for site in _KEPT_SITES + _OPTIONAL_SITES:
            setattr(self, site, None)
        self._team_covariates_mean = None
        self._team_covariates_std = None
        self.epsilon = None
        self.time_diff = None
        self.rescale_weights = None
        self.mcmc_info_ = None

    def _standardised_covariates(self, by_team: Optional[dict]) -> Optional[np.ndarray]:
        """[teams, K] in `self.teams` order, centred and scaled (population std) -- or None."""
        if not by_team:
            return None
        if set(by_team) != set(self.teams):
            raise ValueError("team_covariates must contain all the teams in the data.")
        table = np.array([by_team[name] for name in self.teams], dtype=np.float64)
        self._team_covariates_mean, self._team_covariates_std = table.mean(axis=0), table.std(axis=0)
        return (table - self._team_covariates_mean) / self._team_covariates_std

    def _time_weights(self) -> Optional[np.ndarray]:
        """exp(-epsilon * time_diff), optionally rescaled to sum to the number of fixtures."""
        if self.epsilon is None:
            return None
        if self.time_diff is None:
            raise ValueError(
                "time_diff must be provided in training_data to include exponential time decay in model."
            )
        age = np.asarray(self.time_diff, dtype=np.float64)
        w = np.exp(-self.epsilon * age)
        return w * (age.shape[0] / w.sum()) if self.rescale_weights else w

    # pylint: disable=arguments-differ,too-many-arguments
    def fit(self, training_data: TrainingData, random_state: int = 42, num_warmup: int = 500,
            num_samples: int = 1000, epsilon: Optional[float] = None,
            rescale_weights: Optional[bool] = False, mcmc_kwargs: Optional[Dict[str, Any]] = None,
            run_kwargs: Optional[Dict[str, Any]] = None) -> "ExtendedDixonColesMatchPredictor":
        """Same arguments and defaults as the reference's fit (bpl/extended_dixon_coles.py:251-261)."""
        names = parse_teams(training_data["home_team"], training_data["away_team"], DTYPES["teams"])
        self.teams, self._teams_dict, home_idx, away_idx = names
        self.epsilon, self.rescale_weights = epsilon, rescale_weights
        self.time_diff = training_data.get("time_diff")
        weights = self._time_weights()  # (raises before any device work, like the reference)
        covariates = self._standardised_covariates(training_data.get("team_covariates"))

        draws, self.mcmc_info_ = run_mcmc(
            MODEL_EXTENDED, home_idx, away_idx, np.array(training_data["home_goals"]),
            np.array(training_data["away_goals"]), len(self.teams), weights=weights,
            covariates_std=covariates, random_state=random_state, num_warmup=num_warmup,
            num_samples=num_samples, mcmc_kwargs=mcmc_kwargs, run_kwargs=run_kwargs,
        )
        for site in _KEPT_SITES:
            setattr(self, site, draws[site])
        for site in _OPTIONAL_SITES:
            setattr(self, site, draws.get(site))
        return self

    def add_new_team(self, team_name: str, team_covariates: Optional[np.ndarray] = None) -> None:
        """Append a team the model has not seen, with abilities drawn from the hierarchical prior
        of every posterior draw (bpl/extended_dixon_coles.py:401-457).  Consumes numpy's global
        random stream exactly as the reference does: three blocks of `draws` standard normals
        (attack, defence given attack, home advantage)."""
        if team_name in self.teams:
            raise ValueError(f"Team {team_name} already known to model.")

        prior_attack, prior_defence = 0.0, self.mean_defence
        if self.attack_coefficients is not None:
            if team_covariates is None:
                warnings.warn(
                    f"You haven't provided features for {team_name}."
                    " Assuming team_covariates are the average of known teams."
                    " For better forecasts, provide team_covariates."
                )
                z = np.zeros(self.attack_coefficients.shape[1])
            else:
                # (the reference halves the standardised covariates of a new team, :427-431)
                z = 0.5 * (np.asarray(team_covariates).ravel() - self._team_covariates_mean) / self._team_covariates_std
            prior_attack = self.attack_coefficients @ z
            prior_defence = self.mean_defence + self.defence_coefficients @ z

        n_draws = len(self.std_attack)
        e_att, e_def, e_home = (np.random.standard_normal(n_draws) for _ in range(3))
        std_attack = e_att
        std_defence = self.rho * e_att + np.sqrt(1.0 - self.rho**2.0) * e_def
        columns = {
            "attack": prior_attack + std_attack * self.std_attack,
            "defence": prior_defence + std_defence * self.std_defence,
            "home_advantage": self.mean_home_advantage + self.std_home_advantage * e_home,
        }
        self._teams_dict[team_name] = len(self._teams_dict)
        self.teams = np.append(self.teams, team_name)
        for site, column in columns.items():
            setattr(self, site, np.column_stack([getattr(self, site), column]))
