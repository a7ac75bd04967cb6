"""ORACLE (test infrastructure, NOT product code) -- float64 numpy restatement of
the Dixon-Coles joint log-density and its reverse-mode gradient.

PARITY UNPINNED: the reference (anguswilliams91/bpl-next) holds no golden numbers for
this path (tests/ only assert properties after a real MCMC fit) and its arithmetic
lives in numpyro==0.13.2 / jax==0.4.24 (poetry.lock), which are not installed here
(`import bpl` -> ModuleNotFoundError: No module named 'jax', an ordinary error).
This file restates the published algorithm and is pinned only by (i) a literal
torch-float64 transcription of the reference model differentiated by autograd
(oracle/dc_torch_ref.py), (ii) central finite differences, (iii) the restatement
known-answers of SURVEY.md Appendix C.  Only tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg may import this module.

What it follows (reference file:line):
  * basic model      bpl/dixon_coles.py:39-84
  * extended model   bpl/extended_dixon_coles.py:78-248
  * rho bounds       bpl/_util.py:17-31   (compute_corr_coef_bounds)
  * tau term         bpl/_util.py:35-93   (dixon_coles_correlation_term, tol=0)
  * team indexing    bpl/_util.py:115-135 (parse_teams: sorted unique names)
  * numpyro 0.13.2 semantics (potential_energy = -(sum log_prob + sum log|J|),
    ExpTransform for HalfNormal sites, SigmoidTransform (clipped expit) for Beta
    sites, LocScaleReparam(centered=0), handlers.scale, factor; flat latent order =
    sorted site names).

Latent layout (flat, sorted site names):
  basic    : attack_decentered[T], corr_coef_raw, defence_decentered[T],
             home_advantage, mean_defence, std_attack, std_defence        D = 2T+5
  extended : attack_coefficients[K], corr_coef_raw, defence_coefficients[K],
             home_advantage_decentered[T], mean_defence, mean_home_advantage,
             standardised_attack[T], standardised_defence[T], std_attack,
             std_defence, std_home_advantage, u                           D = 3T+2K+7
"""

from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Dict, Iterable, Optional, Tuple

import numpy as np
from scipy.special import gammaln

MODEL_BASIC = 0
MODEL_EXTENDED = 1

HALF_LOG_2PI = 0.5 * math.log(2.0 * math.pi)
LOG2 = math.log(2.0)
RATE_CLIP = 15.0  # bpl/extended_dixon_coles.py:197-198
# numpyro SigmoidTransform clips expit() to [finfo.tiny, 1 - finfo.eps]; the
# reference runs in float32 (JAX default), so these are the float32 constants.
SIG_LO = float(np.finfo(np.float32).tiny)
SIG_HI = 1.0 - float(np.finfo(np.float32).eps)


# --------------------------------------------------------------------------- data


def parse_teams(home_team: Iterable[str], away_team: Iterable[str]):
    """bpl/_util.py:115-135 -- string-sorted unique team names -> indices."""
    teams = np.array(sorted(set(home_team) | set(away_team)))
    teams_dict = {t: i for i, t in enumerate(teams)}
    home_ind = np.array([teams_dict[t] for t in home_team], dtype=np.uint16)
    away_ind = np.array([teams_dict[t] for t in away_team], dtype=np.uint16)
    return teams, teams_dict, home_ind, away_ind


@dataclass
class Fixtures:
    """Concrete (non-traced) model arguments -- SURVEY.md fact 0.6."""

    home_idx: np.ndarray  # [N] int
    away_idx: np.ndarray  # [N] int
    home_goals: np.ndarray  # [N] int
    away_goals: np.ndarray  # [N] int
    n_teams: int
    weights: Optional[np.ndarray] = None  # [N] float64 or None
    covariates: Optional[np.ndarray] = None  # [T,K] RAW covariates or None

    def __post_init__(self):
        self.home_idx = np.asarray(self.home_idx).astype(np.int64)
        self.away_idx = np.asarray(self.away_idx).astype(np.int64)
        self.home_goals = np.asarray(self.home_goals).astype(np.int64)
        self.away_goals = np.asarray(self.away_goals).astype(np.int64)
        if self.weights is not None:
            self.weights = np.asarray(self.weights, dtype=np.float64)
        if self.covariates is not None:
            self.covariates = np.asarray(self.covariates, dtype=np.float64)

    @property
    def n(self) -> int:
        return int(self.home_idx.shape[0])

    @property
    def k(self) -> int:
        return 0 if self.covariates is None else int(self.covariates.shape[1])


def time_weights(time_diff, epsilon: float, rescale: bool = False) -> np.ndarray:
    """bpl/extended_dixon_coles.py:202-205."""
    td = np.asarray(time_diff, dtype=np.float64)
    w = np.exp(-float(epsilon) * td)
    if rescale:
        w = td.shape[0] * w / w.sum()
    return w


def standardise_covariates(x: np.ndarray) -> np.ndarray:
    """bpl/extended_dixon_coles.py:124-127 (population std, ddof=0)."""
    x = np.asarray(x, dtype=np.float64)
    return (x - x.mean(axis=0)) / x.std(axis=0)


def latent_dim(model: int, n_teams: int, k: int = 0) -> int:
    if model == MODEL_BASIC:
        return 2 * n_teams + 5
    return 3 * n_teams + 2 * k + 7


def site_slices(model: int, T: int, K: int = 0) -> Dict[str, slice]:
    """Flat offsets of every latent site (sorted-site-name order)."""
    out: Dict[str, slice] = {}
    o = 0

    def put(name, n):
        nonlocal o
        out[name] = slice(o, o + n)
        o += n

    if model == MODEL_BASIC:
        put("attack_decentered", T)
        put("corr_coef_raw", 1)
        put("defence_decentered", T)
        put("home_advantage", 1)
        put("mean_defence", 1)
        put("std_attack", 1)
        put("std_defence", 1)
    else:
        if K:
            put("attack_coefficients", K)
        put("corr_coef_raw", 1)
        if K:
            put("defence_coefficients", K)
        put("home_advantage_decentered", T)
        put("mean_defence", 1)
        put("mean_home_advantage", 1)
        put("standardised_attack", T)
        put("standardised_defence", T)
        put("std_attack", 1)
        put("std_defence", 1)
        put("std_home_advantage", 1)
        put("u", 1)
    return out


# ------------------------------------------------------------------ small helpers


def _sigmoid(x: float) -> float:
    if x >= 0:
        return 1.0 / (1.0 + math.exp(-x))
    e = math.exp(x)
    return e / (1.0 + e)


def _softplus(x: float) -> float:
    return max(x, 0.0) + math.log1p(math.exp(-abs(x)))


def _clipped_sigmoid(x: float) -> Tuple[float, float]:
    """numpyro SigmoidTransform: value and d(value)/dx (0 where the clip binds)."""
    s = _sigmoid(x)
    if s < SIG_LO:
        return SIG_LO, 0.0
    if s > SIG_HI:
        return SIG_HI, 0.0
    return s, s * (1.0 - s)


def _normal_lp(v, mu, sd):
    return -0.5 * ((v - mu) / sd) ** 2 - np.log(sd) - HALF_LOG_2PI


# ------------------------------------------------------- likelihood + tau (shared)


def likelihood_and_adjoint(
    fx: Fixtures,
    attack: np.ndarray,
    defence: np.ndarray,
    home_adv: np.ndarray,  # [T] (basic: the scalar broadcast to T)
    q: float,  # constrained corr_coef_raw in (0,1)
    clip_rates: bool,
    ties: str = "split",
):
    """Poisson + tau part of L, and dL/d(attack, defence, home_adv[T], q).

    Follows bpl/dixon_coles.py:63-84 / bpl/extended_dixon_coles.py:191-248 and
    SURVEY.md Appendix A.2-A.4.  Returns a dict with L_lik, the adjoints and
    diagnostics (rho, LB, UB).
    """
    h, a = fx.home_idx, fx.away_idx
    x, y = fx.home_goals, fx.away_goals
    w = np.ones(fx.n) if fx.weights is None else fx.weights

    eta_h = attack[h] - defence[a] + home_adv[h]
    eta_a = attack[a] - defence[h]
    lam_h_raw = np.exp(eta_h)
    lam_a_raw = np.exp(eta_a)
    if clip_rates:
        clip_h = lam_h_raw > RATE_CLIP
        clip_a = lam_a_raw > RATE_CLIP
        lam_h = np.where(clip_h, RATE_CLIP, lam_h_raw)
        lam_a = np.where(clip_a, RATE_CLIP, lam_a_raw)
    else:
        clip_h = np.zeros(fx.n, dtype=bool)
        clip_a = np.zeros(fx.n, dtype=bool)
        lam_h, lam_a = lam_h_raw, lam_a_raw
    # d lam / d eta  (0 where the clip binds)
    dlh = np.where(clip_h, 0.0, lam_h)
    dla = np.where(clip_a, 0.0, lam_a)

    # Poisson(lam).log_prob(k) = k log lam - lgamma(k+1) - lam   (numpyro)
    pois = w * (x * np.log(lam_h) - lam_h - gammaln(x + 1.0)) + w * (
        y * np.log(lam_a) - lam_a - gammaln(y + 1.0)
    )
    # adjoint of the Poisson part wrt lam
    with np.errstate(divide="ignore", invalid="ignore"):
        dP_dlh = w * (x / lam_h - 1.0)
        dP_dla = w * (y / lam_a - 1.0)

    # bounds (bpl/_util.py:23-30)
    prod = lam_h * lam_a
    iP = int(np.argmax(prod))
    M = prod[iP]
    UB = 1.0 / M if M > 1.0 else 1.0
    iQ = int(np.argmax(lam_h))
    iR = int(np.argmax(lam_a))
    Lh, La = lam_h[iQ], lam_a[iR]
    LB = -1.0 / max(Lh, La)
    rho = LB + q * (UB - LB)
    # Where the extrema are attained.  bpl/_util.py:23-30 takes them with jnp.min / jnp.max, whose
    # derivative at a tie is split EVENLY over the tied entries (jax's reduce-min/-max JVP divides the
    # indicator of the attained positions by their count) -- over tied fixtures inside min_i / max_i, and
    # half / half between the two entries of jnp.array([min_i 1/(lh la), 1]) and of
    # jnp.array([max_i -1/lh, max_i -1/la]).  ties="split" (default) restates that.  Ties are exact
    # floating-point equalities: z = 0 (every rate 1), or teams with identical parameters.  The product
    # path keeps ONE arg-extremal pair, the smallest (home, away) key -- another element of the same
    # subdifferential; ties="first_pair" restates THAT (tests compare gradients at tied points with it;
    # U, rho and the bounds do not depend on the rule).
    wP = np.zeros(fx.n)   # d UB / d (1 / (lh la))_i ... as weights over fixtures
    wQ = np.zeros(fx.n)   # d LB / d (-1 / lh)_i
    wR = np.zeros(fx.n)   # d LB / d (-1 / la)_i
    tP, tQ, tR = prod == M, lam_h == Lh, lam_a == La
    tied = False
    if ties == "split":
        shareP = 1.0 if M > 1.0 else (0.5 if M == 1.0 else 0.0)
        wP[tP] = shareP / tP.sum()
        shareQ = 1.0 if Lh > La else (0.5 if Lh == La else 0.0)
        wQ[tQ] = shareQ / tQ.sum()
        wR[tR] = (1.0 - shareQ) / tR.sum()
    elif ties == "first_pair":
        key = (h.astype(np.int64) << 16) + a.astype(np.int64)   # the product's pair table is sorted by (home, away)

        def first(mask):
            idx = np.flatnonzero(mask)
            return idx[np.argmin(key[idx])]

        if M > 1.0:
            wP[first(tP)] = 1.0
        if Lh >= La:
            wQ[first(tQ)] = 1.0
        else:
            wR[first(tR)] = 1.0
    else:
        raise ValueError(ties)
    # (more than one PAIR attains an extremum that matters, or two branches tie)
    npairs = lambda mask: len(set(zip(h[mask].tolist(), a[mask].tolist())))
    tied = bool((M >= 1.0 and npairs(tP) > 1) or M == 1.0 or Lh == La or
                (Lh >= La and npairs(tQ) > 1) or (La >= Lh and npairs(tR) > 1))

    # tau (bpl/_util.py:58-91), tol = 0
    c00 = (x == 0) & (y == 0)
    c10 = (x == 1) & (y == 0)
    c01 = (x == 0) & (y == 1)
    c11 = (x == 1) & (y == 1)
    arg = np.ones(fx.n)
    arg[c00] = 1.0 - rho * prod[c00]
    arg[c10] = 1.0 + rho * lam_a[c10]
    arg[c01] = 1.0 + rho * lam_h[c01]
    arg[c11] = 1.0 - rho
    low = c00 | c10 | c01 | c11
    clipped_tau = low & (arg <= 0.0)
    with np.errstate(divide="ignore"):
        logtau = np.where(low, np.log(np.maximum(arg, 0.0)), 0.0)
    tau_sum = float(np.sum(w * logtau))

    # d logtau / d(lam_h, lam_a, rho); zero where max(.,0) is active
    inv = np.where(low & ~clipped_tau, 1.0 / np.where(arg > 0, arg, 1.0), 0.0)
    dT_dlh = np.zeros(fx.n)
    dT_dla = np.zeros(fx.n)
    dT_drho = np.zeros(fx.n)
    dT_dlh[c00] = -rho * lam_a[c00] * inv[c00]
    dT_dla[c00] = -rho * lam_h[c00] * inv[c00]
    dT_drho[c00] = -prod[c00] * inv[c00]
    dT_dla[c10] = rho * inv[c10]
    dT_drho[c10] = lam_a[c10] * inv[c10]
    dT_dlh[c01] = rho * inv[c01]
    dT_drho[c01] = lam_h[c01] * inv[c01]
    dT_drho[c11] = -inv[c11]
    G_rho = float(np.sum(w * dT_drho))
    # Conditioning of the tau part (test infrastructure: the gates of tests/test_gpu_parity.py follow it).
    # A relative perturbation d of the product rho*c moves log(1 + rho c) by |rho c| / t * d and its
    # derivative rho c / t by |rho c| / t^2 * d (t = 1 + rho c): near a bound of rho (t -> 0,
    # bpl/_util.py:58-70 with tol = 0) float32 inputs cannot hold more than that.  The product path
    # therefore works classes with t < 1/64 out in float64 (dc_kernels.hip.h, TAU_ILL): the amplification
    # of its float32 part is capped there, and cond_val / cond_grad are the capped sums.
    rc = np.where(low & ~clipped_tau, np.abs(arg - 1.0), 0.0)
    cond_val_raw = float(np.sum(w * rc * inv))          # uncapped: what float64 arithmetic itself feels
    cond_grad_raw = float(max(np.bincount(h, w * rc * inv * inv, fx.n_teams).max(),
                              np.bincount(a, w * rc * inv * inv, fx.n_teams).max()))
    amp = np.minimum(inv, 64.0)
    cond_val = float(np.sum(w * rc * amp))
    per_fix = w * rc * amp * amp
    cond_grad = float(max(np.bincount(h, per_fix, fx.n_teams).max(), np.bincount(a, per_fix, fx.n_teams).max()))

    # total adjoint wrt lam (before rho coupling)
    bar_lh = dP_dlh + w * dT_dlh
    bar_la = dP_dla + w * dT_dla
    # rho = LB + q (UB - LB): d rho/d UB = q, d rho/d LB = 1-q
    #   UB = 1/M (M>1): dUB/d lam_h[P] = -lam_a[P]/M^2, dUB/d lam_a[P] = -lam_h[P]/M^2
    #   LB = -1/max(Lh,La): dLB/d lam[arg] = 1/lam^2
    bar_lh += G_rho * q * wP * (-lam_a / (prod * prod))
    bar_la += G_rho * q * wP * (-lam_h / (prod * prod))
    bar_lh += G_rho * (1.0 - q) * wQ / (lam_h * lam_h)
    bar_la += G_rho * (1.0 - q) * wR / (lam_a * lam_a)

    g_h = bar_lh * dlh  # dL/d eta_h
    g_a = bar_la * dla
    # x/lam*lam is exact only when not clipped; when clipped dlh==0 -> 0 already
    T = fx.n_teams
    g_att = np.bincount(h, g_h, T) + np.bincount(a, g_a, T)
    g_def = -np.bincount(a, g_h, T) - np.bincount(h, g_a, T)
    g_ha = np.bincount(h, g_h, T)
    return {
        "L": float(np.sum(pois)) + tau_sum,
        "g_attack": g_att,
        "g_defence": g_def,
        "g_home_adv": g_ha,
        "g_q": G_rho * (UB - LB),
        "rho": rho,
        "LB": LB,
        "UB": UB,
        "cond_val": cond_val,
        "cond_grad": cond_grad,
        "cond_val_raw": cond_val_raw,
        "cond_grad_raw": cond_grad_raw,
        "tau_min": float(np.min(np.where(low, arg, 1.0))),
        "tied": tied,
    }


# ------------------------------------------------------------------------ models


def _beta_site(zc: float, a: float, b: float):
    """Beta(a,b) prior on sigmoid(zc): returns (q, dq/dz, logp+log|J|, d/dz)."""
    q, dq = _clipped_sigmoid(zc)
    lognorm = math.lgamma(a + b) - math.lgamma(a) - math.lgamma(b)
    lp = (a - 1.0) * math.log(q) + (b - 1.0) * math.log1p(-q) + lognorm
    jac = -_softplus(zc) - _softplus(-zc)
    s = _sigmoid(zc)
    dlp = ((a - 1.0) / q - (b - 1.0) / (1.0 - q)) * dq + (1.0 - 2.0 * s)
    return q, dq, lp + jac, dlp


def _halfnormal_exp_site(zs: float):
    """HalfNormal(1) on exp(zs): (s, logp + log|J|, d/dz)."""
    s = math.exp(zs)
    lp = -0.5 * s * s - HALF_LOG_2PI + LOG2 + zs
    return s, lp, -s * s + 1.0


def potential_and_grad(model: int, fx: Fixtures, z: np.ndarray, ties: str = "split"):
    """U(z) = -log p(z, data) in unconstrained space, and dU/dz (float64).

    Returns (U, grad[D], aux) with aux = {rho, LB, UB, attack, defence,
    home_advantage, corr_coef} (constrained / deterministic sites).
    """
    z = np.asarray(z, dtype=np.float64)
    T = fx.n_teams
    K = fx.k if model == MODEL_EXTENDED else 0
    sl = site_slices(model, T, K)
    assert z.shape == (latent_dim(model, T, K),), (z.shape, latent_dim(model, T, K))
    g = np.zeros_like(z)
    L = 0.0

    if model == MODEL_BASIC:
        a_dec = z[sl["attack_decentered"]]
        d_dec = z[sl["defence_decentered"]]
        gamma = float(z[sl["home_advantage"]][0])
        m = float(z[sl["mean_defence"]][0])
        s_a, lp_sa, dlp_sa = _halfnormal_exp_site(float(z[sl["std_attack"]][0]))
        s_d, lp_sd, dlp_sd = _halfnormal_exp_site(float(z[sl["std_defence"]][0]))
        q, dq, lp_q, dlp_q = _beta_site(float(z[sl["corr_coef_raw"]][0]), 2.0, 2.0)
        attack = s_a * a_dec
        defence = m + s_d * d_dec
        ha = np.full(T, gamma)

        L += float(_normal_lp(gamma, 0.1, 0.2)) + float(_normal_lp(m, 0.0, 1.0))
        L += lp_sa + lp_sd + lp_q
        L += float(np.sum(_normal_lp(a_dec, 0.0, 1.0)))
        L += float(np.sum(_normal_lp(d_dec, 0.0, 1.0)))

        lik = likelihood_and_adjoint(fx, attack, defence, ha, q, clip_rates=False, ties=ties)
        L += lik["L"]
        ga, gd, gh = lik["g_attack"], lik["g_defence"], lik["g_home_adv"]

        g[sl["attack_decentered"]] = s_a * ga - a_dec
        g[sl["defence_decentered"]] = s_d * gd - d_dec
        g[sl["home_advantage"]] = gh.sum() - (gamma - 0.1) / 0.04
        g[sl["mean_defence"]] = gd.sum() - m
        g[sl["std_attack"]] = s_a * float(a_dec @ ga) + dlp_sa
        g[sl["std_defence"]] = s_d * float(d_dec @ gd) + dlp_sd
        g[sl["corr_coef_raw"]] = lik["g_q"] * dq + dlp_q
        aux_ha = gamma
    else:
        ha_dec = z[sl["home_advantage_decentered"]]
        sa = z[sl["standardised_attack"]]
        sd = z[sl["standardised_defence"]]
        m = float(z[sl["mean_defence"]][0])
        mha = float(z[sl["mean_home_advantage"]][0])
        s_a, lp_sa, dlp_sa = _halfnormal_exp_site(float(z[sl["std_attack"]][0]))
        s_d, lp_sd, dlp_sd = _halfnormal_exp_site(float(z[sl["std_defence"]][0]))
        s_h, lp_sh, dlp_sh = _halfnormal_exp_site(
            float(z[sl["std_home_advantage"]][0])
        )
        q, dq, lp_q, dlp_q = _beta_site(float(z[sl["corr_coef_raw"]][0]), 2.0, 2.0)
        u, du, lp_u, dlp_u = _beta_site(float(z[sl["u"]][0]), 2.0, 4.0)
        rho_p = 2.0 * u - 1.0
        v = 1.0 - rho_p * rho_p
        if K:
            Xs = standardise_covariates(fx.covariates)
            b_a = z[sl["attack_coefficients"]]
            b_d = z[sl["defence_coefficients"]]
            apm = Xs @ b_a
            dpm = m + Xs @ b_d
            L += float(np.sum(_normal_lp(b_a, 0.0, 1.0)))
            L += float(np.sum(_normal_lp(b_d, 0.0, 1.0)))
        else:
            apm = np.zeros(T)
            dpm = np.full(T, m)
        attack = apm + sa * s_a
        defence = dpm + sd * s_d
        ha = mha + s_h * ha_dec

        L += float(_normal_lp(mha, 0.1, 0.2)) + float(_normal_lp(m, 0.0, 1.0))
        L += lp_sa + lp_sd + lp_sh + lp_q + lp_u
        L += float(np.sum(_normal_lp(sa, 0.0, 1.0)))
        e = sd - rho_p * sa
        L += float(np.sum(-0.5 * e * e / v - 0.5 * math.log(v) - HALF_LOG_2PI))
        L += float(np.sum(_normal_lp(ha_dec, 0.0, 1.0)))

        lik = likelihood_and_adjoint(fx, attack, defence, ha, q, clip_rates=True, ties=ties)
        L += lik["L"]
        ga, gd, gh = lik["g_attack"], lik["g_defence"], lik["g_home_adv"]

        g[sl["standardised_attack"]] = s_a * ga - sa + rho_p * e / v
        g[sl["standardised_defence"]] = s_d * gd - e / v
        g[sl["home_advantage_decentered"]] = s_h * gh - ha_dec
        g[sl["mean_home_advantage"]] = gh.sum() - (mha - 0.1) / 0.04
        g[sl["std_home_advantage"]] = s_h * float(ha_dec @ gh) + dlp_sh
        g[sl["mean_defence"]] = gd.sum() - m
        g[sl["std_attack"]] = s_a * float(sa @ ga) + dlp_sa
        g[sl["std_defence"]] = s_d * float(sd @ gd) + dlp_sd
        g[sl["corr_coef_raw"]] = lik["g_q"] * dq + dlp_q
        dL_drho_p = float(np.sum(e * sa / v - rho_p * e * e / (v * v) + rho_p / v))
        g[sl["u"]] = 2.0 * dL_drho_p * du + dlp_u
        if K:
            g[sl["attack_coefficients"]] = Xs.T @ ga - b_a
            g[sl["defence_coefficients"]] = Xs.T @ gd - b_d
        aux_ha = ha

    aux = {
        "rho": lik["rho"],
        "LB": lik["LB"],
        "UB": lik["UB"],
        "attack": attack,
        "defence": defence,
        "home_advantage": aux_ha,
        "corr_coef": lik["rho"],
        "cond_val": lik["cond_val"],
        "cond_grad": lik["cond_grad"],
        "cond_val_raw": lik["cond_val_raw"],
        "cond_grad_raw": lik["cond_grad_raw"],
        "tau_min": lik["tau_min"],
        "tied": lik["tied"],
    }
    return -L, -g, aux


def potential(model: int, fx: Fixtures, z: np.ndarray) -> float:
    return potential_and_grad(model, fx, z)[0]


def finite_difference_grad(model, fx, z, h=1e-6):
    z = np.asarray(z, dtype=np.float64)
    out = np.zeros_like(z)
    for i in range(z.size):
        zp = z.copy()
        zm = z.copy()
        zp[i] += h
        zm[i] -= h
        out[i] = (potential(model, fx, zp) - potential(model, fx, zm)) / (2 * h)
    return out


# ------------------------------------------------- synthetic fixture recipes (§8d)


def dummy_data_recipe():
    """Exactly tests/conftest.py:7-29 of the reference (BASELINE config 1)."""
    import itertools

    rs = np.random.RandomState(42)  # == legacy np.random.seed(42) stream
    home_goals = rs.poisson(2.1, size=380)
    away_goals = rs.poisson(1.7, size=380)
    teams = [str(i) for i in range(20)]
    home_team, away_team = [], []
    for a, b in itertools.permutations(teams, 2):
        home_team.append(a)
        away_team.append(b)
    return {
        "home_team": home_team,
        "away_team": away_team,
        "home_goals": home_goals,
        "away_goals": away_goals,
    }


def timed_dummy_data_recipe():
    """tests/conftest.py:32-62 of the reference."""
    mpp = 20
    home_team = ["A", "B"] * (mpp // 2) * 3
    away_team = ["B", "A"] * (mpp // 2) * 3
    home_goals = [2, 0] * (mpp // 2) + [1] * mpp + [0, 2] * (mpp // 2)
    away_goals = [0, 2] * (mpp // 2) + [1] * mpp + [2, 0] * (mpp // 2)
    return {
        "home_team": home_team,
        "away_team": away_team,
        "home_goals": home_goals,
        "away_goals": away_goals,
        "time_diff": np.linspace(5, 0, num=mpp * 3),
    }


def synthetic_league(n: int, n_teams: int = 20, seed: int = 2024):
    """SURVEY.md §8(d) C2/C3/C5 recipe: the T(T-1) ordered pairs tiled cyclically,
    truth attack/defence ~ N(0, 0.3^2), home_adv 0.25, goals Poisson."""
    import itertools

    teams = [str(i) for i in range(n_teams)]
    _, tdict, _, _ = parse_teams(teams, teams)
    perms = list(itertools.permutations(teams, 2))
    ph = np.array([tdict[p[0]] for p in perms], dtype=np.uint16)
    pa = np.array([tdict[p[1]] for p in perms], dtype=np.uint16)
    idx = np.arange(n) % len(perms)
    h, a = ph[idx], pa[idx]
    rs = np.random.RandomState(seed)
    att = rs.normal(0.0, 0.3, n_teams)
    dfn = rs.normal(0.0, 0.3, n_teams)
    lam_h = np.exp(att[h] - dfn[a] + 0.25)
    lam_a = np.exp(att[a] - dfn[h])
    x = np.minimum(rs.poisson(lam_h), 255).astype(np.uint8)
    y = np.minimum(rs.poisson(lam_a), 255).astype(np.uint8)
    return h, a, x, y


def fixtures_from_training_data(td: dict, **kw) -> Tuple[Fixtures, np.ndarray]:
    teams, _, h, a = parse_teams(td["home_team"], td["away_team"])
    fx = Fixtures(h, a, td["home_goals"], td["away_goals"], len(teams), **kw)
    return fx, teams
