"""Host-side helpers of the predictors (numpy).

What the reference's bpl/_util.py holds for the MODEL -- `compute_corr_coef_bounds` (:17-31) and
`dixon_coles_correlation_term` (:35-93) -- lives in the HIP kernels here (csrc/dc_kernels.hip.h for
`fit`, csrc/dc_predict.hip.h for the predict methods); this module keeps only the argument
plumbing: team-name parsing and the categorical sampling behind `sample_score` / `sample_outcome`.
"""

from typing import Iterable, Tuple

import numpy as np


def str_to_list(*args):
    """A bare string stands for a one-element list; anything else passes through
    (role of bpl/_util.py:10-14)."""
    return tuple([arg] if isinstance(arg, str) else arg for arg in args)


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


def parse_teams(
    home_team: Iterable[str], away_team: Iterable[str], dtype: str
) -> Tuple[np.ndarray, dict, np.ndarray, np.ndarray]:
    """(team names in string-sorted order, name -> index, home indices, away indices), the
    contract of bpl/_util.py:115-135: with teams "0".."19" team "2" has index 12."""
    home, away = np.asarray(list(home_team)), np.asarray(list(away_team))
    names = np.unique(np.concatenate([home, away]))  # sorted by code point, like sorted(set(...))
    lookup = {str(name): index for index, name in enumerate(names)}
    return names, lookup, np.searchsorted(names, home).astype(dtype), np.searchsorted(names, away).astype(dtype)
