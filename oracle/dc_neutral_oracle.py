"""ORACLE (test infrastructure, NOT product code) -- float64 restatement of the
neutral-venue Dixon-Coles model (bpl/neutral_dixon_coles.py:102-283,
`NeutralDixonColesMatchPredictor._model`), SURVEY.md §8 row f-4.

PARITY UNPINNED: numpyro/jax are not installed and the reference's tests hold property
asserts only (tests/test_neutral_dixon_coles.py).  Two independent restatements live here
and are checked against each other and against finite differences
(tests/test_oracle_neutral.py):
  * potential_and_grad        numpy, hand-derived adjoint
  * torch_potential_and_grad  literal op-for-op transcription of the model function under
                              numpyro 0.13.2 semantics, differentiated by torch autograd

Model (bpl/neutral_dixon_coles.py): as the extended model (rho-correlated standardised
attack/defence, optional covariates) with
  * four per-team non-centred offsets home_attack, away_attack, home_defence, away_defence
    (:204-225), switched off at neutral venues (:236-247):
      eta_h = attack[h] - defence[a] + (1-nv)(home_attack[h] - away_defence[a])
      eta_a = attack[a] - defence[h] + (1-nv)(away_attack[a] - home_defence[h])
  * HalfNormal(0.5) on std_attack / std_defence (:140-141), no rate clip,
  * an always-weighted likelihood: w = [exp(-eps*dt) (rescaled)] * game_weights (:251-257).

The World-Cup variant (bpl/neutral_dixon_coles_WC.py:83-232) adds a per-confederation
strength ~ N(0,1): eta_h += cs[home_conf] - cs[away_conf], eta_a += cs[away_conf] - cs[home_conf]
(:188-203), and always time-weights (:206-208).

Latent layout (flat, sorted site names), T teams, K covariates, C confederations (0 for the
plain neutral model); D = 6T + 2K + C + 13:
  attack_coefficients[K], away_attack_decentered[T], away_defence_decentered[T],
  confederation_strength_decentered[C], corr_coef_raw, defence_coefficients[K], home_attack_decentered[T],
  home_defence_decentered[T], mean_away_attack, mean_away_defence, mean_defence,
  mean_home_attack, mean_home_defence, standardised_attack[T], standardised_defence[T],
  std_attack, std_away_attack, std_away_defence, std_defence, std_home_attack,
  std_home_defence, u
"""

from __future__ import annotations

import itertools
import math
from dataclasses import dataclass
from typing import Dict, Optional

import numpy as np
from scipy.special import gammaln

from dc_oracle import HALF_LOG_2PI, LOG2, SIG_HI, SIG_LO, standardise_covariates  # noqa: F401

MODEL_NEUTRAL = 3


@dataclass
class NeutralFixtures:
    home_idx: np.ndarray
    away_idx: np.ndarray
    home_goals: np.ndarray
    away_goals: np.ndarray
    neutral: np.ndarray  # 0/1
    weights: np.ndarray  # final per-fixture weights (time decay x game weights)
    n_teams: int
    covariates: Optional[np.ndarray] = None  # RAW [T,K]
    # World-Cup variant (bpl/neutral_dixon_coles_WC.py): confederation of each side
    home_conf: Optional[np.ndarray] = None
    away_conf: Optional[np.ndarray] = None
    n_conf: int = 0

    def __post_init__(self):
        for f in ("home_idx", "away_idx", "home_goals", "away_goals", "neutral"):
            setattr(self, f, np.asarray(getattr(self, f)).astype(np.int64))
        self.weights = np.asarray(self.weights, dtype=np.float64)
        if self.covariates is not None:
            self.covariates = np.asarray(self.covariates, dtype=np.float64)
        if self.n_conf:
            self.home_conf = np.asarray(self.home_conf).astype(np.int64)
            self.away_conf = np.asarray(self.away_conf).astype(np.int64)

    @property
    def n(self):
        return self.home_idx.size

    @property
    def k(self):
        return 0 if self.covariates is None else self.covariates.shape[1]


def make_weights(n, time_diff=None, epsilon=None, game_weights=None, rescale_weights=False):
    """bpl/neutral_dixon_coles.py:251-257."""
    w = np.ones(n)
    if epsilon is not None:
        w = w * np.exp(-epsilon * np.asarray(time_diff, dtype=np.float64))
        if rescale_weights:
            w = n * w / w.sum()
    return w * np.asarray(game_weights, dtype=np.float64)


def site_list(T: int, K: int = 0, C: int = 0):
    s = []
    if K:
        s.append(("attack_coefficients", K))
    s += [("away_attack_decentered", T), ("away_defence_decentered", T)]
    if C:
        s.append(("confederation_strength_decentered", C))
    s.append(("corr_coef_raw", 1))
    if K:
        s.append(("defence_coefficients", K))
    s += [("home_attack_decentered", T), ("home_defence_decentered", T),
          ("mean_away_attack", 1), ("mean_away_defence", 1), ("mean_defence", 1),
          ("mean_home_attack", 1), ("mean_home_defence", 1),
          ("standardised_attack", T), ("standardised_defence", T),
          ("std_attack", 1), ("std_away_attack", 1), ("std_away_defence", 1),
          ("std_defence", 1), ("std_home_attack", 1), ("std_home_defence", 1), ("u", 1)]
    return s


def latent_dim(T, K=0, C=0):
    return sum(n for _, n in site_list(T, K, C))


def site_slices(T, K=0, C=0) -> Dict[str, slice]:
    out, o = {}, 0
    for name, n in site_list(T, K, C):
        out[name] = slice(o, o + n)
        o += n
    return out


def _sig(x):
    return np.where(x >= 0, 1.0 / (1.0 + np.exp(-np.abs(x))), 1.0 - 1.0 / (1.0 + np.exp(-np.abs(x))))


def _softplus(x):
    return np.maximum(x, 0.0) + np.log1p(np.exp(-np.abs(x)))


def _clip_sig(zr):
    s = float(_sig(np.float64(zr)))
    if s < SIG_LO:
        return SIG_LO, 0.0, s
    if s > SIG_HI:
        return SIG_HI, 0.0, s
    return s, s * (1.0 - s), s


def potential_and_grad(fx: NeutralFixtures, z: np.ndarray):
    """U(z) = -log p(z, data) and its gradient; aux: rho (corr_coef), LB, UB and sites."""
    T, K, N, C = fx.n_teams, fx.k, fx.n, fx.n_conf
    sl = site_slices(T, K, C)
    z = np.asarray(z, dtype=np.float64)
    g = np.zeros_like(z)
    L = 0.0
    h, a, x, y, nv, w = fx.home_idx, fx.away_idx, fx.home_goals, fx.away_goals, fx.neutral, fx.weights
    on = 1.0 - nv

    # ---- scalar hypers
    m_def = z[sl["mean_defence"]][0]
    L += -0.5 * m_def ** 2 - HALF_LOG_2PI
    g[sl["mean_defence"]] += -m_def  # dL/d (accumulate dL, negate at the end)
    stds = {}
    for name, scale in (("std_attack", 0.5), ("std_defence", 0.5), ("std_home_attack", 1.0),
                        ("std_away_attack", 1.0), ("std_home_defence", 1.0), ("std_away_defence", 1.0)):
        zs = z[sl[name]][0]
        s = math.exp(zs)
        stds[name] = s
        # HalfNormal(scale): log 2 - log(scale) - 0.5 log 2pi - s^2/(2 scale^2); + Jacobian zs
        L += LOG2 - math.log(scale) - HALF_LOG_2PI - 0.5 * (s / scale) ** 2 + zs
        g[sl[name]] += -(s / scale) ** 2 + 1.0
    means = {}
    for name, mu in (("mean_home_attack", 0.1), ("mean_away_attack", -0.1),
                     ("mean_home_defence", 0.1), ("mean_away_defence", -0.1)):
        m = z[sl[name]][0]
        means[name] = m
        r = (m - mu) / 0.2
        L += -0.5 * r * r - math.log(0.2) - HALF_LOG_2PI
        g[sl[name]] += -(m - mu) / 0.04

    # ---- covariate prior means
    att_mean = np.zeros(T)
    def_mean = np.full(T, m_def)
    xs = None
    if K:
        xs = standardise_covariates(fx.covariates)
        bA, bD = z[sl["attack_coefficients"]], z[sl["defence_coefficients"]]
        L += (-0.5 * bA ** 2 - HALF_LOG_2PI).sum() + (-0.5 * bD ** 2 - HALF_LOG_2PI).sum()
        g[sl["attack_coefficients"]] += -bA
        g[sl["defence_coefficients"]] += -bD
        att_mean = xs @ bA
        def_mean = m_def + xs @ bD

    # ---- u ~ Beta(2,4), rho_p = 2u - 1
    zu = z[sl["u"]][0]
    u, du, su = _clip_sig(zu)
    L += math.log(u) + 3.0 * math.log1p(-u) + math.log(20.0) - _softplus(zu) - _softplus(-zu)
    g_u = (1.0 / u - 3.0 / (1.0 - u)) * du + (1.0 - 2.0 * su)
    rp = 2.0 * u - 1.0
    vv = 1.0 - rp * rp

    # ---- per-team sites
    sa, sd = z[sl["standardised_attack"]], z[sl["standardised_defence"]]
    e = sd - rp * sa
    L += (-0.5 * sa ** 2 - HALF_LOG_2PI).sum()
    L += (-0.5 * e ** 2 / vv - 0.5 * math.log(vv) - HALF_LOG_2PI).sum()
    g[sl["standardised_attack"]] += -sa + rp * e / vv
    g[sl["standardised_defence"]] += -e / vv
    dL_drp = (e * sa / vv - rp * e ** 2 / vv ** 2 + rp / vv).sum()
    g_u += 2.0 * dL_drp * du
    dec = {}
    for nm in ("home_attack", "away_attack", "home_defence", "away_defence"):
        d_ = z[sl[nm + "_decentered"]]
        dec[nm] = d_
        L += (-0.5 * d_ ** 2 - HALF_LOG_2PI).sum()
        g[sl[nm + "_decentered"]] += -d_
    s_att, s_def = stds["std_attack"], stds["std_defence"]
    attack = att_mean + sa * s_att
    defence = def_mean + sd * s_def
    hat = means["mean_home_attack"] + stds["std_home_attack"] * dec["home_attack"]
    aat = means["mean_away_attack"] + stds["std_away_attack"] * dec["away_attack"]
    hdf = means["mean_home_defence"] + stds["std_home_defence"] * dec["home_defence"]
    adf = means["mean_away_defence"] + stds["std_away_defence"] * dec["away_defence"]

    # ---- likelihood
    eh = attack[h] - defence[a] + on * (hat[h] - adf[a])
    ea = attack[a] - defence[h] + on * (aat[a] - hdf[h])
    if C:  # confederation_strength ~ N(0,1), non-centred with loc 0 / scale 1 (WC :178-203)
        cs = z[sl["confederation_strength_decentered"]]
        L += (-0.5 * cs ** 2 - HALF_LOG_2PI).sum()
        g[sl["confederation_strength_decentered"]] += -cs
        eh = eh + cs[fx.home_conf] - cs[fx.away_conf]
        ea = ea + cs[fx.away_conf] - cs[fx.home_conf]
    lh, la = np.exp(eh), np.exp(ea)
    L += (w * (x * eh - lh - gammaln(x + 1.0))).sum() + (w * (y * ea - la - gammaln(y + 1.0))).sum()
    gh = w * (x - lh)  # dL/d eta_h
    ga = w * (y - la)

    # ---- corr_coef_raw ~ Beta(2,2); bounds; tau
    zc = z[sl["corr_coef_raw"]][0]
    q, dq, sq = _clip_sig(zc)
    L += math.log(q) + math.log1p(-q) + math.log(6.0) - _softplus(zc) - _softplus(-zc)
    g_c = (1.0 / q - 1.0 / (1.0 - q)) * dq + (1.0 - 2.0 * sq)
    prod = lh * la
    iP, iQ, iR = int(np.argmax(prod)), int(np.argmax(lh)), int(np.argmax(la))
    M, Lh, La = prod[iP], lh[iQ], la[iR]
    UB = 1.0 / M if M > 1.0 else 1.0
    LB = -1.0 / max(Lh, La)
    rho = LB + q * (UB - LB)
    low = (x <= 1) & (y <= 1)
    c = np.where(x == 0, np.where(y == 0, -lh * la, lh), np.where(y == 0, la, -1.0))
    arg = 1.0 + rho * c
    with np.errstate(divide="ignore", invalid="ignore"):
        lt = np.where(low, np.log(np.clip(arg, 0.0, None)), 0.0)
        uu = np.where(low & (arg > 0), c / arg, 0.0)
    L += (w * lt).sum()
    G_rho = (w * uu).sum()
    gh += np.where(low & (x == 0), w * rho * uu, 0.0)
    ga += np.where(low & (y == 0), w * rho * uu, 0.0)
    # adjoint of the bounds
    g_c += G_rho * (UB - LB) * dq
    if M > 1.0:
        v = G_rho * q * (-UB)
        gh[iP] += v
        ga[iP] += v
    v = G_rho * (1.0 - q) * (-LB)
    if Lh >= La:
        gh[iQ] += v
    else:
        ga[iR] += v

    # ---- scatter to team tables
    G_att = np.bincount(h, gh, T) + np.bincount(a, ga, T)
    G_def = -(np.bincount(a, gh, T) + np.bincount(h, ga, T))
    G_hat = np.bincount(h, gh * on, T)
    G_adf = -np.bincount(a, gh * on, T)
    G_aat = np.bincount(a, ga * on, T)
    G_hdf = -np.bincount(h, ga * on, T)
    g[sl["standardised_attack"]] += s_att * G_att
    g[sl["standardised_defence"]] += s_def * G_def
    g[sl["std_attack"]] += s_att * (sa * G_att).sum()
    g[sl["std_defence"]] += s_def * (sd * G_def).sum()
    g[sl["mean_defence"]] += G_def.sum()
    if K:
        g[sl["attack_coefficients"]] += xs.T @ G_att
        g[sl["defence_coefficients"]] += xs.T @ G_def
    for nm, Gt in (("home_attack", G_hat), ("away_attack", G_aat), ("home_defence", G_hdf),
                   ("away_defence", G_adf)):
        s = stds["std_" + nm]
        g[sl[nm + "_decentered"]] += s * Gt
        g[sl["mean_" + nm]] += Gt.sum()
        g[sl["std_" + nm]] += s * (dec[nm] * Gt).sum()
    if C:
        g[sl["confederation_strength_decentered"]] += (
            np.bincount(fx.home_conf, gh - ga, C) + np.bincount(fx.away_conf, ga - gh, C))
    g[sl["u"]] += g_u
    g[sl["corr_coef_raw"]] += g_c
    aux = {"rho": rho, "LB": LB, "UB": UB, "q": q, "attack": attack, "defence": defence,
           "home_attack": hat, "away_attack": aat, "home_defence": hdf, "away_defence": adf,
           "corr_coef": rho}
    return -L, -g, aux


def torch_potential_and_grad(fx: NeutralFixtures, z: np.ndarray):
    """Literal transcription of bpl/neutral_dixon_coles.py:136-283 + numpyro's transforms,
    differentiated by autograd (float64)."""
    import torch

    T, K, C = fx.n_teams, fx.k, fx.n_conf
    sl = site_slices(T, K, C)
    zt = torch.tensor(np.asarray(z, dtype=np.float64), requires_grad=True)
    N01 = lambda v, loc=0.0, sc=1.0: -0.5 * ((v - loc) / sc) ** 2 - math.log(sc) - HALF_LOG_2PI
    logp = torch.zeros((), dtype=torch.float64)

    def halfnormal(name, scale):
        nonlocal logp
        zs = zt[sl[name]][0]
        s = torch.exp(zs)
        logp = logp + (LOG2 + N01(s, 0.0, scale)) + zs
        return s

    def normal(name, loc, sc):
        nonlocal logp
        v = zt[sl[name]][0]
        logp = logp + N01(v, loc, sc)
        return v

    def beta(name, c1, c0):
        nonlocal logp
        zr = zt[sl[name]][0]
        s = torch.sigmoid(zr)
        v = torch.clamp(s, SIG_LO, SIG_HI)
        lbeta = math.lgamma(c1) + math.lgamma(c0) - math.lgamma(c1 + c0)
        logp = logp + (c1 - 1) * torch.log(v) + (c0 - 1) * torch.log1p(-v) - lbeta
        logp = logp - torch.nn.functional.softplus(zr) - torch.nn.functional.softplus(-zr)
        return v

    mean_defence = normal("mean_defence", 0.0, 1.0)
    std_attack = halfnormal("std_attack", 0.5)
    std_defence = halfnormal("std_defence", 0.5)
    mean_home_attack = normal("mean_home_attack", 0.1, 0.2)
    mean_away_attack = normal("mean_away_attack", -0.1, 0.2)
    mean_home_defence = normal("mean_home_defence", 0.1, 0.2)
    mean_away_defence = normal("mean_away_defence", -0.1, 0.2)
    std_home_attack = halfnormal("std_home_attack", 1.0)
    std_away_attack = halfnormal("std_away_attack", 1.0)
    std_home_defence = halfnormal("std_home_defence", 1.0)
    std_away_defence = halfnormal("std_away_defence", 1.0)
    if K:
        cov = torch.tensor(fx.covariates)
        xs = (cov - cov.mean(0)) / cov.std(0, unbiased=False)
        bA, bD = zt[sl["attack_coefficients"]], zt[sl["defence_coefficients"]]
        logp = logp + N01(bA).sum() + N01(bD).sum()
        attack_prior_mean = xs @ bA
        defence_prior_mean = mean_defence + xs @ bD
    else:
        attack_prior_mean = 0.0
        defence_prior_mean = mean_defence
    u = beta("u", 2.0, 4.0)
    rho_p = 2.0 * u - 1.0
    sa = zt[sl["standardised_attack"]]
    sd = zt[sl["standardised_defence"]]
    logp = logp + N01(sa).sum()
    sc = torch.sqrt(1.0 - rho_p ** 2)
    logp = logp + (-0.5 * ((sd - rho_p * sa) / sc) ** 2 - torch.log(sc) - HALF_LOG_2PI).sum()

    def noncentred(name, loc, scale):
        nonlocal logp
        d_ = zt[sl[name + "_decentered"]]
        logp = logp + N01(d_).sum()
        return loc + scale * d_

    home_attack = noncentred("home_attack", mean_home_attack, std_home_attack)
    away_attack = noncentred("away_attack", mean_away_attack, std_away_attack)
    home_defence = noncentred("home_defence", mean_home_defence, std_home_defence)
    away_defence = noncentred("away_defence", mean_away_defence, std_away_defence)
    attack = attack_prior_mean + sa * std_attack
    defence = defence_prior_mean + sd * std_defence

    h = torch.tensor(fx.home_idx)
    a = torch.tensor(fx.away_idx)
    nv = torch.tensor(fx.neutral, dtype=torch.float64)
    x = torch.tensor(fx.home_goals, dtype=torch.float64)
    y = torch.tensor(fx.away_goals, dtype=torch.float64)
    w = torch.tensor(fx.weights)
    conf_h = conf_a = 0.0
    if C:
        confederation_strength = noncentred("confederation_strength", 0.0, 1.0)
        conf_h = confederation_strength[torch.tensor(fx.home_conf)]
        conf_a = confederation_strength[torch.tensor(fx.away_conf)]
    lam_h = torch.exp(attack[h] - defence[a] + conf_h - conf_a
                      + (1 - nv) * home_attack[h] - (1 - nv) * away_defence[a])
    lam_a = torch.exp(attack[a] - defence[h] + conf_a - conf_h
                      + (1 - nv) * away_attack[a] - (1 - nv) * home_defence[h])
    pois = lambda k, lam: k * torch.log(lam) - torch.lgamma(k + 1.0) - lam
    logp = logp + (w * pois(x, lam_h)).sum() + (w * pois(y, lam_a)).sum()

    q = beta("corr_coef_raw", 2.0, 2.0)
    UB = torch.clamp(torch.min(1.0 / (lam_h * lam_a)), max=1.0)
    LB = torch.maximum(torch.max(-1.0 / lam_h), torch.max(-1.0 / lam_a))
    corr = LB + q * (UB - LB)
    # dixon_coles_correlation_term (bpl/_util.py:35-93), static index sets
    xi, yi = fx.home_goals, fx.away_goals
    term = torch.zeros(fx.n, dtype=torch.float64)
    for mask, fn in (((xi == 0) & (yi == 0), lambda i: 1.0 - corr * lam_h[i] * lam_a[i]),
                     ((xi == 1) & (yi == 0), lambda i: 1.0 + corr * lam_a[i]),
                     ((xi == 0) & (yi == 1), lambda i: 1.0 + corr * lam_h[i]),
                     ((xi == 1) & (yi == 1), lambda i: 1.0 - corr + 0.0 * lam_h[i])):
        idx = torch.tensor(np.nonzero(mask)[0])
        if idx.numel():
            term = term.index_put((idx,), torch.log(torch.clamp(fn(idx), min=0.0)))
    logp = logp + (w * term).sum()
    U = -logp
    (gr,) = torch.autograd.grad(U, zt)
    return float(U.detach()), gr.numpy(), {"rho": float(corr.detach()), "LB": float(LB.detach()),
                                           "UB": float(UB.detach())}


def neutral_dummy_recipe(epsilon=None, rescale_weights=False):
    """tests/conftest.py:66-117 (`neutral_dummy_data`): 380 league + 190 neutral cup matches."""
    np.random.seed(42)
    neutral_venue = np.array([0] * 380 + [1] * 190)
    home_means = [2.1 if v == 0 else 1.9 for v in neutral_venue]
    away_means = [1.7 if v == 0 else 1.9 for v in neutral_venue]
    home_goals = np.random.poisson(home_means)
    away_goals = np.random.poisson(away_means)
    time_diff = np.concatenate([np.array([1.0] * 380), np.linspace(0, 10, num=190)])
    game_weights = np.concatenate([np.array([1.0] * 380), np.random.uniform(0, 10, size=190)])
    teams = [str(i) for i in range(20)]
    home_team, away_team = [], []
    for p, q in itertools.permutations(teams, 2):
        home_team.append(p)
        away_team.append(q)
    for p, q in itertools.combinations(teams, 2):
        home_team.append(p)
        away_team.append(q)
    # deterministic assignment of teams to conferences (tests/conftest.py:103-105)
    return {"home_team": home_team, "away_team": away_team, "home_goals": home_goals,
            "away_goals": away_goals, "neutral_venue": neutral_venue, "time_diff": time_diff,
            "game_weights": game_weights,
            "home_conf": [str(int(t) // 4) for t in home_team],
            "away_conf": [str(int(t) // 4) for t in away_team]}


def fixtures_from_data(dd, epsilon=None, rescale_weights=False, covariates=None):
    teams = sorted(set(dd["home_team"]) | set(dd["away_team"]))
    idx = {t: i for i, t in enumerate(teams)}
    h = np.array([idx[t] for t in dd["home_team"]])
    a = np.array([idx[t] for t in dd["away_team"]])
    w = make_weights(len(h), dd.get("time_diff"), epsilon, dd["game_weights"], rescale_weights)
    return NeutralFixtures(h, a, dd["home_goals"], dd["away_goals"], dd["neutral_venue"], w,
                           len(teams), covariates=covariates)


def make_weights_wc(n, time_diff, epsilon, game_weights, rescale_weights=False):
    """bpl/neutral_dixon_coles_WC.py:206-208 (always time-weighted; rescaled AFTER the game weights)."""
    w = np.exp(-epsilon * np.asarray(time_diff, dtype=np.float64)) * np.asarray(game_weights, np.float64)
    if rescale_weights:
        w = n * w / w.sum()
    return w


def fixtures_from_data_wc(dd, epsilon=0.0, rescale_weights=False, covariates=None):
    """World-Cup variant: confederations from dd['home_conf'] / dd['away_conf'] (string sorted)."""
    fx = fixtures_from_data(dd, covariates=covariates)
    confs = sorted(set(dd["home_conf"]) | set(dd["away_conf"]))
    cidx = {c: i for i, c in enumerate(confs)}
    fx.home_conf = np.array([cidx[c] for c in dd["home_conf"]])
    fx.away_conf = np.array([cidx[c] for c in dd["away_conf"]])
    fx.n_conf = len(confs)
    fx.weights = make_weights_wc(fx.n, dd["time_diff"], epsilon, dd["game_weights"], rescale_weights)
    return fx


def synthetic_neutral(n, n_teams=20, seed=11, k=0, n_conf=0):
    rs = np.random.RandomState(seed)
    h = rs.randint(0, n_teams, n)
    a = (h + 1 + rs.randint(0, n_teams - 1, n)) % n_teams
    nv = rs.randint(0, 2, n)
    w = rs.uniform(0.2, 3.0, n)
    cov = rs.normal(size=(n_teams, k)) if k else None
    x, y = rs.poisson(1.4, n), rs.poisson(1.1, n)
    conf_of = rs.randint(0, max(n_conf, 1), n_teams)
    return NeutralFixtures(h, a, x, y, nv, w, n_teams, covariates=cov,
                           home_conf=conf_of[h] if n_conf else None,
                           away_conf=conf_of[a] if n_conf else None, n_conf=n_conf)
