"""The team-level Dixon-Coles model behind the reference's class name and `fit` signature
(bpl/dixon_coles.py:26-163).  The model function (`_model`, :39-84) and numpyro's NUTS are
replaced by libbplhip.so -- the `dc_eval` kernel evaluates the log-density and its gradient,
the chain lives on the device -- and the predict methods come from `BaseMatchPredictor`
(one device grid per fixture)."""

from __future__ import annotations

from typing import Any, Dict, Iterable, Optional, Union

import numpy as np

from bpl._ffi import MODEL_BASIC
from bpl._mcmc import run_mcmc
from bpl._util import parse_teams
from bpl.base import BaseMatchPredictor, DTYPES

__all__ = ["DixonColesMatchPredictor"]
TrainingData = Dict[str, Union[Iterable[str], Iterable[float]]]

# posterior sites the reference keeps as attributes after a fit (bpl/dixon_coles.py:118-122)
_KEPT_SITES = ("attack", "defence", "home_advantage", "corr_coef")


class DixonColesMatchPredictor(BaseMatchPredictor):
    """Attack and defence ability per team, one home advantage, Dixon-Coles low-score
    correlation."""

    def __init__(self):
        super().__init__()
        for site in _KEPT_SITES:
            setattr(self, site, None)
        self.mcmc_info_ = None  # sampler statistics (no reference counterpart)

    # pylint: disable=arguments-differ,too-many-arguments
    def fit(self, training_data: TrainingData, random_state: int = 42, num_warmup: int = 500,
            num_samples: int = 1000, mcmc_kwargs: Optional[Dict[str, Any]] = None,
            run_kwargs: Optional[Dict[str, Any]] = None) -> "DixonColesMatchPredictor":
        """Same arguments and defaults as the reference's fit (bpl/dixon_coles.py:87-95)."""
        names = parse_teams(training_data["home_team"], training_data["away_team"], DTYPES["teams"])
        self.teams, self._teams_dict, home_idx, away_idx = names
        goals = [np.array(training_data[side]) for side in ("home_goals", "away_goals")]
        draws, self.mcmc_info_ = run_mcmc(
            MODEL_BASIC, home_idx, away_idx, goals[0], goals[1], len(self.teams),
            random_state=random_state, num_warmup=num_warmup, num_samples=num_samples,
            mcmc_kwargs=mcmc_kwargs, run_kwargs=run_kwargs,
        )
        for site in _KEPT_SITES:
            setattr(self, site, draws[site])
        return self
