"""ORACLE (test infrastructure, NOT product code) -- float64 restatement of the dynamic
(time-varying) neutral Dixon-Coles model, value and hand-derived gradient.

PARITY UNPINNED, and more: the reference class
(bpl/dynamic_dixon_coles.py:23-584, `DynamicNeutralDixonColesMatchPredictor`) is
unfinished -- not exported (bpl/__init__.py:4-7), untested, and as written its random
walk is a no-op: `attack.at[j].set(...)` results are discarded (:192-218), so `attack`
and `defence` stay `jnp.empty` and the rates depend on the four offset tables only
(SURVEY.md Appendix D1).  `random_walk=True` (default) restates the INTENDED model
(attack[g] = attack[g-1] + standardised_attack[g] * std_attack[g], same for defence);
`random_walk=False` restates the behaviour of the code as written (attack = defence = 0).
Other as-written defects are not reproduced: num_gameweeks = max(gameweek)+1 (D2).

Follows bpl/dynamic_dixon_coles.py:63-247 under numpyro 0.13.2 semantics.  Checked
against torch autograd of a literal transcription (tests/test_oracle_dynamic.py).

Latent layout (flat, sorted site names), G gameweeks, T teams, K covariates:
  attack_coefficients[K], away_attack_decentered[G,T], away_defence_decentered[G,T],
  corr_coef_raw, defence_coefficients[K], home_attack_decentered[G,T],
  home_defence_decentered[G,T], mean_away_attack[G], mean_away_defence[G], mean_defence,
  mean_home_attack[G], mean_home_defence[G], standardised_attack[G,T],
  standardised_defence[G,T], std_attack[G], std_away_attack[G], std_away_defence[G],
  std_defence[G], std_home_attack[G], std_home_defence[G], u[G,T]
  D = 7GT + 10G + 2 + 2K
"""

from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Dict, Optional

import numpy as np
from scipy.special import gammaln

from dc_oracle import HALF_LOG_2PI, LOG2, SIG_HI, SIG_LO, standardise_covariates

MODEL_DYNAMIC = 2


@dataclass
class DynFixtures:
    home_idx: np.ndarray
    away_idx: np.ndarray
    home_goals: np.ndarray
    away_goals: np.ndarray
    gameweek: np.ndarray  # 0-based
    neutral: np.ndarray  # 0/1
    n_teams: int
    n_gameweeks: int
    covariates: Optional[np.ndarray] = None  # RAW [T,K]

    def __post_init__(self):
        for f in ("home_idx", "away_idx", "home_goals", "away_goals", "gameweek", "neutral"):
            setattr(self, f, np.asarray(getattr(self, f)).astype(np.int64))
        if self.covariates is not None:
            self.covariates = np.asarray(self.covariates, dtype=np.float64)

    @property
    def n(self):
        return int(self.home_idx.shape[0])

    @property
    def k(self):
        return 0 if self.covariates is None else int(self.covariates.shape[1])


def site_list(G: int, T: int, K: int = 0):
    s = []
    if K:
        s.append(("attack_coefficients", (K,)))
    s += [("away_attack_decentered", (G, T)), ("away_defence_decentered", (G, T)),
          ("corr_coef_raw", ())]
    if K:
        s.append(("defence_coefficients", (K,)))
    s += [("home_attack_decentered", (G, T)), ("home_defence_decentered", (G, T)),
          ("mean_away_attack", (G,)), ("mean_away_defence", (G,)), ("mean_defence", ()),
          ("mean_home_attack", (G,)), ("mean_home_defence", (G,)),
          ("standardised_attack", (G, T)), ("standardised_defence", (G, T)),
          ("std_attack", (G,)), ("std_away_attack", (G,)), ("std_away_defence", (G,)),
          ("std_defence", (G,)), ("std_home_attack", (G,)), ("std_home_defence", (G,)),
          ("u", (G, T))]
    return s


def latent_dim(G, T, K=0):
    return 7 * G * T + 10 * G + 2 + 2 * K


def site_slices(G, T, K=0) -> Dict[str, slice]:
    out, o = {}, 0
    for name, shape in site_list(G, T, K):
        n = int(np.prod(shape)) if shape else 1
        out[name] = slice(o, o + n)
        o += n
    return out


def unpack(z, G, T, K=0):
    sl = site_slices(G, T, K)
    d = {}
    for name, shape in site_list(G, T, K):
        v = z[sl[name]]
        d[name] = v.reshape(shape) if shape else float(v[0])
    return d


def _sig(x):
    s = 1.0 / (1.0 + np.exp(-x))
    v = np.clip(s, SIG_LO, SIG_HI)
    dv = np.where((s < SIG_LO) | (s > SIG_HI), 0.0, s * (1.0 - s))
    return v, dv, s


def _softplus(x):
    return np.maximum(x, 0.0) + np.log1p(np.exp(-np.abs(x)))


def potential_and_grad(fx: DynFixtures, z: np.ndarray, random_walk: bool = True):
    G, T, K = fx.n_gameweeks, fx.n_teams, fx.k
    z = np.asarray(z, dtype=np.float64)
    assert z.shape == (latent_dim(G, T, K),)
    sl = site_slices(G, T, K)
    p = unpack(z, G, T, K)
    g = np.zeros_like(z)
    L = 0.0

    def gset(name, val):
        g[sl[name]] = np.asarray(val, dtype=np.float64).reshape(-1)

    # ---- per-gameweek hypers
    hyp = {}
    for nm, mu in (("home_attack", 0.1), ("away_attack", -0.1), ("home_defence", 0.1),
                   ("away_defence", -0.1)):
        mean = p["mean_" + nm]
        zs = p["std_" + nm]
        s = np.exp(zs)
        L += np.sum(-0.5 * ((mean - mu) / 0.2) ** 2 - math.log(0.2) - HALF_LOG_2PI)
        L += np.sum(-0.5 * s * s - HALF_LOG_2PI + LOG2 + zs)
        hyp[nm] = (mean, s, mu)
    s_att, s_def = np.exp(p["std_attack"]), np.exp(p["std_defence"])
    L += np.sum(-0.5 * s_att**2 - HALF_LOG_2PI + LOG2 + p["std_attack"])
    L += np.sum(-0.5 * s_def**2 - HALF_LOG_2PI + LOG2 + p["std_defence"])
    m = p["mean_defence"]
    L += -0.5 * m * m - HALF_LOG_2PI
    if K:
        Xs = standardise_covariates(fx.covariates)
        b_a, b_d = p["attack_coefficients"], p["defence_coefficients"]
        L += np.sum(-0.5 * b_a**2 - HALF_LOG_2PI) + np.sum(-0.5 * b_d**2 - HALF_LOG_2PI)
        apm, dpm = Xs @ b_a, m + Xs @ b_d
    else:
        apm, dpm = np.zeros(T), np.full(T, m)

    # ---- per-cell sites
    zu = p["u"]
    u, du, su = _sig(zu)
    L += np.sum(np.log(u) + 3.0 * np.log1p(-u) + math.log(20.0) - _softplus(zu) - _softplus(-zu))
    rp = 2.0 * u - 1.0
    v = 1.0 - rp * rp
    sa, sd = p["standardised_attack"], p["standardised_defence"]
    e = sd - rp * sa
    L += np.sum(-0.5 * sa * sa - HALF_LOG_2PI)
    L += np.sum(-0.5 * e * e / v - 0.5 * np.log(v) - HALF_LOG_2PI)
    off = {}
    for nm in ("home_attack", "away_attack", "home_defence", "away_defence"):
        dec = p[nm + "_decentered"]
        L += np.sum(-0.5 * dec * dec - HALF_LOG_2PI)
        mean, s, _ = hyp[nm]
        off[nm] = mean[:, None] + s[:, None] * dec
    if random_walk:
        attack = apm[None, :] + np.cumsum(sa * s_att[:, None], axis=0)
        defence = dpm[None, :] + np.cumsum(sd * s_def[:, None], axis=0)
    else:  # the reference as written: jnp.empty -> zeros, never updated
        attack = np.zeros((G, T))
        defence = np.zeros((G, T))

    # ---- likelihood + tau
    gw, h, a = fx.gameweek, fx.home_idx, fx.away_idx
    x, y = fx.home_goals, fx.away_goals
    nn = 1.0 - fx.neutral
    eta_h = attack[gw, h] - defence[gw, a] + nn * (off["home_attack"][gw, h] - off["away_defence"][gw, a])
    eta_a = attack[gw, a] - defence[gw, h] + nn * (off["away_attack"][gw, a] - off["home_defence"][gw, h])
    lh, la = np.exp(eta_h), np.exp(eta_a)
    L += np.sum(x * eta_h - lh - gammaln(x + 1.0) + y * eta_a - la - gammaln(y + 1.0))
    zc = p["corr_coef_raw"]
    q, dq, sq = _sig(np.array(zc))
    q, dq, sq = float(q), float(dq), float(sq)
    L += -float(_softplus(np.array(zc))) - float(_softplus(np.array(-zc)))  # Uniform(0,1): log_prob 0
    prod = lh * la
    iP, iQ, iR = int(np.argmax(prod)), int(np.argmax(lh)), int(np.argmax(la))
    M, Lh, La = prod[iP], lh[iQ], la[iR]
    UB = 1.0 / M if M > 1.0 else 1.0
    LB = -1.0 / max(Lh, La)
    rho = LB + q * (UB - LB)
    c = np.zeros(fx.n)
    c00, c10, c01, c11 = (x == 0) & (y == 0), (x == 1) & (y == 0), (x == 0) & (y == 1), (x == 1) & (y == 1)
    c[c00], c[c10], c[c01], c[c11] = -prod[c00], la[c10], lh[c01], -1.0
    arg = 1.0 + rho * c
    pos = arg > 0
    with np.errstate(divide="ignore"):
        L += np.sum(np.where(c != 0, np.log(np.maximum(arg, 0.0)), 0.0))
    uu = np.where(pos, c / np.where(pos, arg, 1.0), 0.0)
    G_rho = float(np.sum(uu))
    gh = x - lh + np.where(x == 0, rho * uu, 0.0) * ((x <= 1) & (y <= 1))
    ga = y - la + np.where(y == 0, rho * uu, 0.0) * ((x <= 1) & (y <= 1))
    if M > 1.0:
        gh[iP] += G_rho * q * (-UB)
        ga[iP] += G_rho * q * (-UB)
    if Lh >= La:
        gh[iQ] += G_rho * (1.0 - q) * (-LB)
    else:
        ga[iR] += G_rho * (1.0 - q) * (-LB)

    def scat(idx_t, w):
        out = np.zeros((G, T))
        np.add.at(out, (gw, idx_t), w)
        return out

    G_att = scat(h, gh) + scat(a, ga)
    G_def = -scat(a, gh) - scat(h, ga)
    G_hatt, G_adef = scat(h, nn * gh), -scat(a, nn * gh)
    G_aatt, G_hdef = scat(a, nn * ga), -scat(h, nn * ga)
    if not random_walk:
        G_att[:] = 0.0
        G_def[:] = 0.0

    # ---- chain rule
    for nm, Gc in (("home_attack", G_hatt), ("away_attack", G_aatt), ("home_defence", G_hdef),
                   ("away_defence", G_adef)):
        mean, s, mu = hyp[nm]
        dec = p[nm + "_decentered"]
        gset(nm + "_decentered", s[:, None] * Gc - dec)
        gset("mean_" + nm, Gc.sum(axis=1) - (mean - mu) / 0.04)
        gset("std_" + nm, s * np.sum(dec * Gc, axis=1) + 1.0 - s * s)
    RA = np.cumsum(G_att[::-1], axis=0)[::-1]  # RA[j] = sum_{g>=j} G_att[g]
    RD = np.cumsum(G_def[::-1], axis=0)[::-1]
    gset("standardised_attack", s_att[:, None] * RA - sa + rp * e / v)
    gset("standardised_defence", s_def[:, None] * RD - e / v)
    gset("std_attack", s_att * np.sum(sa * RA, axis=1) + 1.0 - s_att**2)
    gset("std_defence", s_def * np.sum(sd * RD, axis=1) + 1.0 - s_def**2)
    gset("mean_defence", RD[0].sum() - m)
    if K:
        gset("attack_coefficients", Xs.T @ RA[0] - b_a)
        gset("defence_coefficients", Xs.T @ RD[0] - b_d)
    dL_drp = e * sa / v - rp * e * e / (v * v) + rp / v
    gset("u", (1.0 / u - 3.0 / (1.0 - u)) * du + 2.0 * dL_drp * du + (1.0 - 2.0 * su))
    gset("corr_coef_raw", G_rho * (UB - LB) * dq + (1.0 - 2.0 * sq))
    aux = {"rho": rho, "LB": LB, "UB": UB, "attack": attack, "defence": defence, **off}
    return -float(L), -g, aux


# ----------------------------------------------------------- torch literal transcription


def torch_potential_and_grad(fx: DynFixtures, z: np.ndarray, random_walk: bool = True):
    """bpl/dynamic_dixon_coles.py:63-247 op for op in torch float64 (+ the intended walk),
    differentiated by autograd."""
    import torch

    G, T, K = fx.n_gameweeks, fx.n_teams, fx.k
    sl = site_slices(G, T, K)
    zt = torch.tensor(np.asarray(z, np.float64), dtype=torch.float64, requires_grad=True)

    def site(name, shape=None):
        v = zt[sl[name]]
        return v.reshape(shape) if shape else v[0]

    def nlp(v, mu, sd):
        return -0.5 * ((v - mu) / sd) ** 2 - torch.log(torch.as_tensor(sd, dtype=torch.float64)) - HALF_LOG_2PI

    def hn(zs):
        s = torch.exp(zs)
        return s, nlp(s, 0.0, 1.0) + LOG2 + zs

    lp = torch.zeros((), dtype=torch.float64)
    means, stds = {}, {}
    for nm, mu in (("home_attack", 0.1), ("away_attack", -0.1), ("home_defence", 0.1), ("away_defence", -0.1)):
        means[nm] = site("mean_" + nm, (G,))
        lp = lp + nlp(means[nm], mu, 0.2).sum()
    for nm in ("home_attack", "away_attack", "home_defence", "away_defence"):
        stds[nm], l = hn(site("std_" + nm, (G,)))
        lp = lp + l.sum()
    std_attack, l = hn(site("std_attack", (G,)))
    lp = lp + l.sum()
    std_defence, l = hn(site("std_defence", (G,)))
    lp = lp + l.sum()
    mean_defence = site("mean_defence")
    lp = lp + nlp(mean_defence, 0.0, 1.0)
    if K:
        tc = torch.as_tensor(fx.covariates)
        sc = (tc - tc.mean(dim=0)) / tc.std(dim=0, unbiased=False)
        b_a, b_d = site("attack_coefficients", (K,)), site("defence_coefficients", (K,))
        lp = lp + nlp(b_a, 0.0, 1.0).sum() + nlp(b_d, 0.0, 1.0).sum()
        apm = torch.matmul(sc, b_a[:, None]).squeeze(-1)
        dpm = mean_defence + torch.matmul(sc, b_d[:, None]).squeeze(-1)
    else:
        apm, dpm = torch.zeros(T, dtype=torch.float64), mean_defence + torch.zeros(T, dtype=torch.float64)
    zu = site("u", (G, T))
    u = torch.clamp(torch.sigmoid(zu), SIG_LO, SIG_HI)
    sp = torch.nn.functional.softplus
    lp = lp + (torch.log(u) + 3 * torch.log1p(-u) + math.log(20.0) - sp(zu) - sp(-zu)).sum()
    rho_p = 2.0 * u - 1.0
    sa, sd = site("standardised_attack", (G, T)), site("standardised_defence", (G, T))
    lp = lp + nlp(sa, 0.0, 1.0).sum()
    scale = torch.sqrt(1.0 - rho_p**2.0)
    lp = lp + (-0.5 * ((sd - rho_p * sa) / scale) ** 2 - torch.log(scale) - HALF_LOG_2PI).sum()
    off = {}
    for nm in ("home_attack", "away_attack", "home_defence", "away_defence"):
        dec = site(nm + "_decentered", (G, T))
        lp = lp + nlp(dec, 0.0, 1.0).sum()
        off[nm] = means[nm][:, None] + stds[nm][:, None] * dec
    if random_walk:
        rows_a = [apm + sa[0] * std_attack[0]]
        rows_d = [dpm + sd[0] * std_defence[0]]
        for j in range(1, G):
            rows_a.append(rows_a[-1] + sa[j] * std_attack[j])
            rows_d.append(rows_d[-1] + sd[j] * std_defence[j])
        attack, defence = torch.stack(rows_a), torch.stack(rows_d)
    else:
        attack = torch.zeros((G, T), dtype=torch.float64)
        defence = torch.zeros((G, T), dtype=torch.float64)
    gw, h, a = (torch.as_tensor(v) for v in (fx.gameweek, fx.home_idx, fx.away_idx))
    nv = torch.as_tensor(fx.neutral, dtype=torch.float64)
    x, y = torch.as_tensor(fx.home_goals, dtype=torch.float64), torch.as_tensor(fx.away_goals, dtype=torch.float64)
    ehg = torch.exp(attack[gw, h] - defence[gw, a] + (1 - nv) * off["home_attack"][gw, h]
                    - (1 - nv) * off["away_defence"][gw, a])
    eag = torch.exp(attack[gw, a] - defence[gw, h] + (1 - nv) * off["away_attack"][gw, a]
                    - (1 - nv) * off["home_defence"][gw, h])
    lp = lp + (torch.log(ehg) * x - torch.lgamma(x + 1) - ehg).sum()
    lp = lp + (torch.log(eag) * y - torch.lgamma(y + 1) - eag).sum()
    zc = site("corr_coef_raw")
    raw = torch.clamp(torch.sigmoid(zc), SIG_LO, SIG_HI)
    lp = lp - sp(zc) - sp(-zc)
    UB = torch.minimum(torch.amin(1.0 / (ehg * eag)), torch.ones((), dtype=torch.float64))
    LB = torch.maximum(torch.amax(-1.0 / ehg), torch.amax(-1.0 / eag))
    corr = LB + raw * (UB - LB)
    xi, yi = fx.home_goals, fx.away_goals
    for mask, fn in (((xi == 0) & (yi == 0), lambda i: 1.0 - corr * ehg[i] * eag[i]),
                     ((xi == 1) & (yi == 0), lambda i: 1.0 + corr * eag[i]),
                     ((xi == 0) & (yi == 1), lambda i: 1.0 + corr * ehg[i]),
                     ((xi == 1) & (yi == 1), lambda i: 1.0 - corr + 0.0 * ehg[i])):
        idx = torch.as_tensor(np.nonzero(mask)[0])
        if idx.numel():
            lp = lp + torch.log(torch.clamp(fn(idx), min=0.0)).sum()
    (gr,) = torch.autograd.grad(-lp, zt)
    return float(-lp.detach()), gr.numpy().copy(), float(corr.detach())


# ------------------------------------------------------------------ fixture recipes


def config4_recipe(n_teams=100, n_gameweeks=50, seed=4):
    """one round per gameweek of T/2 disjoint random pairings."""
    rs = np.random.RandomState(seed)
    h, a, g = [], [], []
    for w in range(n_gameweeks):
        p = rs.permutation(n_teams)
        h += list(p[0::2])
        a += list(p[1::2])
        g += [w] * (n_teams // 2)
    n = len(h)
    x = rs.poisson(1.5, n)
    y = rs.poisson(1.2, n)
    return DynFixtures(h, a, x, y, g, np.zeros(n, int), n_teams, n_gameweeks)


def small_recipe(n=300, n_teams=7, n_gameweeks=5, seed=1, k=0):
    rs = np.random.RandomState(seed)
    h = rs.randint(0, n_teams, n)
    a = (h + 1 + rs.randint(0, n_teams - 1, n)) % n_teams
    g = np.sort(rs.randint(0, n_gameweeks, n))
    cov = rs.normal(size=(n_teams, k)) if k else None
    return DynFixtures(h, a, rs.poisson(1.3, n), rs.poisson(1.1, n), g, rs.randint(0, 2, n),
                       n_teams, n_gameweeks, covariates=cov)
