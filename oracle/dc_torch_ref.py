"""ORACLE CROSS-CHECK (test infrastructure, NOT product code).

A literal torch-float64 transcription of the reference's model functions, written
op-for-op in the order the reference declares them, so that torch autograd plays
the role jax.grad plays in the reference.  It exists only to validate the
hand-derived adjoint in oracle/dc_oracle.py (PARITY UNPINNED -- see that header).

Follows:
  basic    bpl/dixon_coles.py:39-84
  extended bpl/extended_dixon_coles.py:78-248
  bounds   bpl/_util.py:17-31
  tau      bpl/_util.py:35-93
and numpyro 0.13.2 potential_energy semantics.
"""

from __future__ import annotations

import math

import numpy as np
import torch

from dc_oracle import (
    MODEL_BASIC,
    SIG_HI,
    SIG_LO,
    Fixtures,
    site_slices,
)

DT = torch.float64
HALF_LOG_2PI = 0.5 * math.log(2.0 * math.pi)


def _normal_lp(v, mu, sd):
    return -0.5 * ((v - mu) / sd) ** 2 - torch.log(sd) - HALF_LOG_2PI


def _t(x):
    return torch.as_tensor(x, dtype=DT)


def _poisson_lp(rate, k):
    # numpyro Poisson.log_prob: log(rate)*k - gammaln(k+1) - rate
    return torch.log(rate) * k - torch.lgamma(k + 1.0) - rate


def _halfnormal1_lp(v):
    return _normal_lp(v, _t(0.0), _t(1.0)) + math.log(2.0)


def _beta_lp(v, a, b):
    norm = math.lgamma(a + b) - math.lgamma(a) - math.lgamma(b)
    return (a - 1.0) * torch.log(v) + (b - 1.0) * torch.log1p(-v) + norm


def _sigmoid_site(zc):
    """SigmoidTransform: clipped expit + log|J| = -softplus(z) - softplus(-z)."""
    v = torch.clamp(torch.sigmoid(zc), SIG_LO, SIG_HI)
    ladj = -torch.nn.functional.softplus(zc) - torch.nn.functional.softplus(-zc)
    return v, ladj


def compute_corr_coef_bounds(lh, la):
    # bpl/_util.py:23-30
    UB = torch.minimum(torch.amin(1.0 / (lh * la)), _t(1.0))
    LB = torch.maximum(torch.amax(-1.0 / lh), torch.amax(-1.0 / la))
    return LB, UB


def dixon_coles_correlation_term(x, y, lh, la, corr, w):
    # bpl/_util.py:54-93 with tol = 0.  The reference gathers with static boolean
    # masks (`.at[..., mask].set(...)`), so unselected fixtures never enter log().
    term = torch.zeros_like(lh)
    nil_nil = torch.as_tensor(np.nonzero((x == 0) & (y == 0))[0])
    one_nil = torch.as_tensor(np.nonzero((x == 1) & (y == 0))[0])
    nil_one = torch.as_tensor(np.nonzero((x == 0) & (y == 1))[0])
    one_one = torch.as_tensor(np.nonzero((x == 1) & (y == 1))[0])
    term = term.index_put(
        (nil_nil,),
        w[nil_nil]
        * torch.log(torch.clamp(1.0 - corr * lh[nil_nil] * la[nil_nil], min=0.0)),
    )
    term = term.index_put(
        (one_nil,),
        w[one_nil] * torch.log(torch.clamp(1.0 + corr * la[one_nil], min=0.0)),
    )
    term = term.index_put(
        (nil_one,),
        w[nil_one] * torch.log(torch.clamp(1.0 + corr * lh[nil_one], min=0.0)),
    )
    term = term.index_put(
        (one_one,),
        w[one_one]
        * torch.log(torch.clamp(1.0 - corr + 0.0 * lh[one_one], min=0.0)),
    )
    return term


def log_density(model: int, fx: Fixtures, z: torch.Tensor):
    """Joint log density in unconstrained space (= -potential_energy)."""
    T = fx.n_teams
    K = fx.k if model != MODEL_BASIC else 0
    sl = site_slices(model, T, K)
    h = torch.as_tensor(fx.home_idx)
    a = torch.as_tensor(fx.away_idx)
    x_np, y_np = fx.home_goals, fx.away_goals
    x = _t(x_np)
    y = _t(y_np)
    lp = _t(0.0)

    if model == MODEL_BASIC:
        home_advantage = z[sl["home_advantage"]][0]
        mean_defence = z[sl["mean_defence"]][0]
        std_attack = torch.exp(z[sl["std_attack"]][0])
        std_defence = torch.exp(z[sl["std_defence"]][0])
        lp = lp + _normal_lp(home_advantage, _t(0.1), _t(0.2))
        lp = lp + _normal_lp(mean_defence, _t(0.0), _t(1.0))
        lp = lp + _halfnormal1_lp(std_attack) + z[sl["std_attack"]][0]
        lp = lp + _halfnormal1_lp(std_defence) + z[sl["std_defence"]][0]
        a_dec = z[sl["attack_decentered"]]
        d_dec = z[sl["defence_decentered"]]
        lp = lp + _normal_lp(a_dec, _t(0.0), _t(1.0)).sum()
        lp = lp + _normal_lp(d_dec, _t(0.0), _t(1.0)).sum()
        attack = 0.0 + std_attack * a_dec
        defence = mean_defence + std_defence * d_dec
        ehg = torch.exp(attack[h] - defence[a] + home_advantage)
        eag = torch.exp(attack[a] - defence[h])
        w = torch.ones_like(ehg)
        lp = lp + _poisson_lp(ehg, x).sum() + _poisson_lp(eag, y).sum()
    else:
        mean_home_advantage = z[sl["mean_home_advantage"]][0]
        std_home_advantage = torch.exp(z[sl["std_home_advantage"]][0])
        mean_defence = z[sl["mean_defence"]][0]
        std_attack = torch.exp(z[sl["std_attack"]][0])
        std_defence = torch.exp(z[sl["std_defence"]][0])
        lp = lp + _normal_lp(mean_home_advantage, _t(0.1), _t(0.2))
        lp = lp + _halfnormal1_lp(std_home_advantage) + z[sl["std_home_advantage"]][0]
        lp = lp + _normal_lp(mean_defence, _t(0.0), _t(1.0))
        lp = lp + _halfnormal1_lp(std_attack) + z[sl["std_attack"]][0]
        lp = lp + _halfnormal1_lp(std_defence) + z[sl["std_defence"]][0]
        if K:
            tc = _t(fx.covariates)
            sc = (tc - tc.mean(dim=0)) / tc.std(dim=0, unbiased=False)
            b_a = z[sl["attack_coefficients"]]
            b_d = z[sl["defence_coefficients"]]
            lp = lp + _normal_lp(b_a, _t(0.0), _t(1.0)).sum()
            lp = lp + _normal_lp(b_d, _t(0.0), _t(1.0)).sum()
            attack_prior_mean = torch.matmul(sc, b_a[:, None]).squeeze(-1)
            defence_prior_mean = mean_defence + torch.matmul(sc, b_d[:, None]).squeeze(-1)
        else:
            attack_prior_mean = _t(0.0)
            defence_prior_mean = mean_defence
        u, ladj_u = _sigmoid_site(z[sl["u"]][0])
        lp = lp + _beta_lp(u, 2.0, 4.0) + ladj_u
        rho = 2.0 * u - 1.0
        sa = z[sl["standardised_attack"]]
        sd = z[sl["standardised_defence"]]
        lp = lp + _normal_lp(sa, _t(0.0), _t(1.0)).sum()
        lp = lp + _normal_lp(sd, rho * sa, torch.sqrt(1.0 - rho**2.0)).sum()
        ha_dec = z[sl["home_advantage_decentered"]]
        lp = lp + _normal_lp(ha_dec, _t(0.0), _t(1.0)).sum()
        home_advantage = mean_home_advantage + std_home_advantage * ha_dec
        attack = attack_prior_mean + sa * std_attack
        defence = defence_prior_mean + sd * std_defence
        ehg = torch.exp(attack[h] - defence[a] + home_advantage[h])
        eag = torch.exp(attack[a] - defence[h])
        ehg = torch.clamp(ehg, max=15.0)
        eag = torch.clamp(eag, max=15.0)
        if fx.weights is not None:
            w = _t(fx.weights)
        else:
            w = torch.ones_like(ehg)
        lp = lp + (w * _poisson_lp(ehg, x)).sum() + (w * _poisson_lp(eag, y)).sum()

    raw, ladj = _sigmoid_site(z[sl["corr_coef_raw"]][0])
    lp = lp + _beta_lp(raw, 2.0, 2.0) + ladj
    LB, UB = compute_corr_coef_bounds(ehg, eag)
    corr_coef = LB + raw * (UB - LB)
    corr_term = dixon_coles_correlation_term(x_np, y_np, ehg, eag, corr_coef, w)
    lp = lp + corr_term.sum()
    return lp, corr_coef


def potential_and_grad(model: int, fx: Fixtures, z: np.ndarray):
    zt = torch.tensor(np.asarray(z, dtype=np.float64), dtype=DT, requires_grad=True)
    lp, corr = log_density(model, fx, zt)
    (g,) = torch.autograd.grad(-lp, zt)
    return float(-lp.detach()), g.numpy().copy(), float(corr.detach())
