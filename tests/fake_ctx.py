"""TEST-ONLY stand-in for bpl._ffi.HipContext: same surface, but the potential is the
CPU oracle (through the NUTS harness).  Lets the host orchestration (bpl/_mcmc.py,
bpl/_dist.py: chain ownership, key splitting, broadcast, gather, constrain) run under
gloo on a box without a GPU.  Never imported by product code."""
import numpy as np
import torch

import dc_oracle as O
import dc_oracle_c as OC


class FakeCtx:
    def __init__(self, device_index=0):
        self.device = torch.device("cpu")
        self.dim = 0

    def set_fixtures(self, model, h, a, x, y, n_teams, weights=None, covariates_std=None):
        def np_(t, dt):
            t = t.cpu().numpy() if isinstance(t, torch.Tensor) else np.asarray(t)
            return t.view(np.uint16) if (dt == np.uint16 and t.dtype == np.int16) else t.astype(dt)

        fx = O.Fixtures(np_(h, np.uint16), np_(a, np.uint16), np_(x, np.uint8), np_(y, np.uint8),
                        n_teams, weights=None if weights is None else np_(weights, np.float64))
        self.fx, self.model, self.n_teams = fx, model, n_teams
        self.cf = OC.CFixtures(model, fx)
        if covariates_std is not None:  # already standardised by the caller
            self.cf.K = covariates_std.shape[1]
            self.cf.xs = np.ascontiguousarray(covariates_std, dtype=np.float64)
            self.cf.D = O.latent_dim(model, n_teams, self.cf.K)
        self.dim = self.cf.D
        return self

    def nuts_run(self, cfg, key, z0=None):
        rc, draws, stats, summ = OC.nuts_dc(self.cf, cfg.num_warmup, cfg.num_samples, key,
                                            thin=cfg.thinning, z0=z0)
        assert rc == 0
        kept = draws.shape[0]
        out = {"potential_energy": stats[:, 0], "accept_prob": stats[:, 1],
               "step_size": np.full(kept, summ[0]), "num_steps": stats[:, 2].astype(np.int32),
               "diverging": stats[:, 3].astype(np.int32), "corr_coef": np.zeros(kept),
               "inverse_mass_matrix": summ[4:], "final_step_size": summ[0],
               "mean_accept_prob": summ[1], "total_leapfrogs": int(summ[2]),
               "total_divergences": int(summ[3]), "wall_seconds": 1.0}
        return draws, out

    def constrain(self, z):
        T, sl = self.n_teams, O.site_slices(self.model, self.n_teams, self.cf.K)
        s = z.shape[0]
        att, dfn, corr = np.empty((s, T)), np.empty((s, T)), np.empty(s)
        ha = np.empty(s) if self.model == O.MODEL_BASIC else np.empty((s, T))
        fx = self.fx
        if self.cf.K:  # un-standardised copy is not kept: rebuild a Fixtures with xs as raw
            fx = O.Fixtures(fx.home_idx, fx.away_idx, fx.home_goals, fx.away_goals, T,
                            weights=fx.weights, covariates=self.cf.xs)
        for i in range(s):
            _, _, aux = O.potential_and_grad(self.model, fx, z[i])
            att[i], dfn[i], ha[i], corr[i] = aux["attack"], aux["defence"], aux["home_advantage"], aux["corr_coef"]
        return {"attack": att, "defence": dfn, "home_advantage": ha, "corr_coef": corr}

    def close(self):
        pass


class FakePredictCtx:
    """TEST-ONLY numpy (float64) predict backend with bpl._ffi.HipContext's predict surface:
    a restatement of the reference's predict_score_proba -- bpl/dixon_coles.py:126-163 for the
    league models, bpl/neutral_dixon_coles.py:399-488 / bpl/neutral_dixon_coles_WC.py:363-470 for the
    venue-aware family -- with the tau term of bpl/_util.py:35-93.  The CPU API tests inject it as
    `model._predict_ctx`; the GPU tests use it as the float64 comparator of the device kernels."""

    def predict_set_posterior(self, attack, defence, home_advantage, corr_coef):
        self.att, self.dfn = np.asarray(attack, float), np.asarray(defence, float)
        self.ha, self.cc = np.asarray(home_advantage, float), np.asarray(corr_coef, float)
        self.venue = None

    def predict_set_posterior_venue(self, attack, defence, home_attack, away_attack, home_defence,
                                    away_defence, corr_coef, confederation_strength=None):
        self.att, self.dfn = np.asarray(attack, float), np.asarray(defence, float)
        self.venue = [np.asarray(t, float) for t in (home_attack, away_attack, home_defence, away_defence)]
        self.conf = None if confederation_strength is None else np.asarray(confederation_strength, float)
        self.cc = np.asarray(corr_coef, float)

    def _log_rates(self, h, a, neutral, conf):
        eh = self.att[:, h] - self.dfn[:, a]
        ea = self.att[:, a] - self.dfn[:, h]
        if self.venue is None:
            assert neutral is None and conf is None
            return eh + (self.ha[:, None] if self.ha.ndim == 1 else self.ha[:, h]), ea
        assert neutral is not None and (conf is None) == (self.conf is None)
        hat, aat, hdf, adf = self.venue
        on = 1.0 - np.broadcast_to(np.asarray(neutral, float), h.shape)
        eh = eh + on * hat[:, h] - on * adf[:, a]
        ea = ea + on * aat[:, a] - on * hdf[:, h]
        if conf is not None:
            hc = np.broadcast_to(np.asarray(conf[0], int), h.shape)
            ac = np.broadcast_to(np.asarray(conf[1], int), h.shape)
            d = self.conf[:, hc] - self.conf[:, ac]
            eh, ea = eh + d, ea - d
        return eh, ea

    def predict_score_proba(self, h, a, x, y, neutral=None, conf=None):
        from scipy.special import gammaln

        h, a = np.asarray(h, int), np.asarray(a, int)
        x, y = np.asarray(x, int), np.asarray(y, int)
        eh, ea = self._log_rates(h, a, neutral, conf)
        lh, la = np.exp(eh), np.exp(ea)
        rho = self.cc[:, None]
        tau = np.ones_like(lh)
        tau = np.where((x == 0) & (y == 0), np.clip(1 - rho * lh * la, 0, None), tau)
        tau = np.where((x == 1) & (y == 0), np.clip(1 + rho * la, 0, None), tau)
        tau = np.where((x == 0) & (y == 1), np.clip(1 + rho * lh, 0, None), tau)
        tau = np.where((x == 1) & (y == 1), np.clip(1 - rho + 0 * lh, 0, None), tau)
        ph = np.exp(eh * x - gammaln(x + 1.0) - lh)
        pa = np.exp(ea * y - gammaln(y + 1.0) - la)
        return (tau * ph * pa).mean(axis=0)

    def predict_score_grid(self, h, a, max_goals, neutral=None, conf=None):
        h, a = np.asarray(h, int), np.asarray(a, int)
        g1 = max_goals + 1
        xs, ys = np.meshgrid(np.arange(g1), np.arange(g1), indexing="ij")
        out = np.empty((len(h), g1, g1))
        nv = None if neutral is None else np.broadcast_to(np.asarray(neutral), h.shape)
        for i in range(len(h)):
            ci = None if conf is None else (np.broadcast_to(np.asarray(conf[0]), h.shape)[i],
                                            np.broadcast_to(np.asarray(conf[1]), h.shape)[i])
            out[i] = self.predict_score_proba(np.full(g1 * g1, h[i]), np.full(g1 * g1, a[i]),
                                              xs.ravel(), ys.ravel(),
                                              None if nv is None else nv[i], ci).reshape(g1, g1)
        return out
