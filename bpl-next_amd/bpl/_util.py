"""Private utility functions (host side, numpy).

Mirrors the reference's bpl/_util.py: same names, argument meaning and broadcasting.
The training-time use of compute_corr_coef_bounds / dixon_coles_correlation_term
(bpl/dixon_coles.py:79-84) runs inside the HIP kernel; the functions here serve the
predict path (posterior post-processing), which is host numpy in this round.
"""

from typing import Iterable, Optional, Tuple, Union

import numpy as np


def str_to_list(*args):
    """convert all elements of a list into strings.  (bpl/_util.py:10-14)"""
    return ([x] if isinstance(x, str) else x for x in args)


def compute_corr_coef_bounds(
    expected_home_goals: np.ndarray, expected_away_goals: np.ndarray
) -> Tuple[float, float]:
    """Bounds of the correlation coefficient from the Dixon & Coles paper
    (bpl/_util.py:17-31)."""
    UB = np.min(np.array([np.min(1.0 / (expected_home_goals * expected_away_goals)), 1]))
    LB = np.max(
        np.array([np.max(-1.0 / expected_home_goals), np.max(-1.0 / expected_away_goals)])
    )
    return LB, UB


# pylint: disable=too-many-arguments
def dixon_coles_correlation_term(
    home_goals: Union[int, Iterable[int]],
    away_goals: Union[int, Iterable[int]],
    home_rate: np.ndarray,
    away_rate: np.ndarray,
    corr_coef: np.ndarray,
    weights: Optional[np.ndarray] = None,
    tol: Optional[float] = 0,
) -> np.ndarray:
    """Correlation (tau) term of the Dixon & Coles paper (bpl/_util.py:35-93)."""
    if isinstance(home_goals, (int, np.integer)):
        home_goals = np.array(home_goals).reshape((1,))
    if isinstance(away_goals, (int, np.integer)):
        away_goals = np.array(away_goals).reshape((1,))
    home_goals = np.asarray(home_goals)
    away_goals = np.asarray(away_goals)
    home_rate = np.asarray(home_rate, dtype=np.float64)
    away_rate = np.asarray(away_rate, dtype=np.float64)
    corr_coef = np.asarray(corr_coef, dtype=np.float64)
    if weights is None:
        weights = np.ones(len(home_goals))
    weights = np.asarray(weights, dtype=np.float64)

    corr_term = np.zeros_like(home_rate)
    cc = corr_coef[..., None]
    with np.errstate(divide="ignore", invalid="ignore"):
        nil_nil = (home_goals == 0) & (away_goals == 0)
        corr_term[..., nil_nil] = weights[..., nil_nil] * np.log(
            np.clip(1.0 - cc * home_rate[..., nil_nil] * away_rate[..., nil_nil], tol, None)
        )
        one_nil = (home_goals == 1) & (away_goals == 0)
        corr_term[..., one_nil] = weights[..., one_nil] * np.log(
            np.clip(1.0 + cc * away_rate[..., one_nil], tol, None)
        )
        nil_one = (home_goals == 0) & (away_goals == 1)
        corr_term[..., nil_one] = weights[..., nil_one] * np.log(
            np.clip(1.0 + cc * home_rate[..., nil_one], tol, None)
        )
        one_one = (home_goals == 1) & (away_goals == 1)
        corr_term[..., one_one] = weights[..., one_one] * np.log(
            np.clip(1.0 - cc + 0.0 * home_rate[..., one_one], tol, None)
        )
    return corr_term


def map_choice(key, a, num_samples, p):
    """One categorical draw set per row of p (bpl/_util.py:96-112).

    `key` is a threefry key (hi, lo).  Follows jax.random.split + jax.random.choice
    (replace=True, p given): r = cumsum(p)[-1] * (1 - uniform(key)); searchsorted.
    """
    from bpl._ffi import threefry_bits, threefry_split  # host-side threefry in the lib

    a = np.asarray(a)
    p = np.asarray(p)
    keys = threefry_split(key, p.shape[0])
    out = np.empty((p.shape[0], num_samples), dtype=a.dtype)
    one = np.float32(1.0)
    for i, k in enumerate(keys):
        bits = threefry_bits(k, num_samples)
        u = ((bits >> np.uint32(9)) | np.uint32(0x3F800000)).view(np.float32) - one
        p_cuml = np.cumsum(p[i].astype(np.float32), dtype=np.float32)
        r = p_cuml[-1] * (one - u)
        ind = np.searchsorted(p_cuml, r, side="left")
        out[i] = a[np.minimum(ind, len(a) - 1)]
    return out


def poisson_log_prob(rate, k):
    """numpyro Poisson.log_prob: log(rate)*k - gammaln(k+1) - rate."""
    from scipy.special import gammaln

    k = np.asarray(k, dtype=np.float64)
    with np.errstate(divide="ignore", invalid="ignore"):
        return np.log(rate) * k - gammaln(k + 1.0) - rate


def parse_teams(
    home_team: Iterable[str], away_team: Iterable[str], dtype: str
) -> Tuple[np.ndarray, dict, np.ndarray, np.ndarray]:
    """Unique team names (string-sorted), name->index map and the per-fixture indices
    (bpl/_util.py:115-135)."""
    home_team = list(home_team)
    away_team = list(away_team)
    teams = np.array(sorted(set(home_team) | set(away_team)))
    teams_dict = {t: i for i, t in enumerate(teams)}
    home_ind = np.array([teams_dict[t] for t in home_team], dtype)
    away_ind = np.array([teams_dict[t] for t in away_team], dtype)
    return teams, teams_dict, home_ind, away_ind
