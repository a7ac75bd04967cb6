"""ctypes binding of libbplhip.so (include/bplhip.h) and a thin context wrapper.

Python holds the device buffers (torch-ROCm tensors) and passes raw device pointers
and the current HIP stream across the C-ABI.  There is NO CPU fallback: if the shared
library is missing, or there is no GPU, constructing a `HipContext` raises.
"""

from __future__ import annotations

import ctypes as C
import os
from typing import Optional, Sequence, Tuple

import numpy as np

MODEL_BASIC = 0
MODEL_EXTENDED = 1
MODEL_DYNAMIC = 2
MODEL_NEUTRAL = 3
# error codes of include/bplhip.h
BPLHIP_EINVAL, BPLHIP_ESTATE, BPLHIP_EHIP, BPLHIP_ENOMEM, BPLHIP_EUNSUPPORTED, BPLHIP_ENUMERIC = -1, -2, -3, -4, -5, -6

_LIB_NAME = os.environ.get("BPLHIP_LIB", "libbplhip.so")  # override: diagnostic builds only
_lib = None

# every symbol include/bplhip.h declares (checked by tests/test_abi.py without a GPU)
ABI_SYMBOLS = (
    "bplhip_abi_version",
    "bplhip_create",
    "bplhip_destroy",
    "bplhip_last_error",
    "bplhip_set_fixtures",
    "bplhip_set_fixtures_dynamic",
    "bplhip_set_fixtures_neutral",
    "bplhip_constrain_dynamic",
    "bplhip_set_option",
    "bplhip_latent_dim",
    "bplhip_logp_grad",
    "bplhip_logp_grad_batched",
    "bplhip_logp_grad_graph",
    "bplhip_nuts_default_cfg",
    "bplhip_nuts_run",
    "bplhip_nuts_run_chains",
    "bplhip_constrain",
    "bplhip_predict_set_posterior",
    "bplhip_predict_score_proba",
    "bplhip_predict_score_grid",
    "bplhip_predict_score_grid_f32",
    "bplhip_predict_set_posterior_venue",
    "bplhip_predict_score_proba_venue",
    "bplhip_predict_score_grid_venue",
    "bplhip_predict_score_grid_venue_f32",
    "bplhip_selftest_math",
    "bplhip_threefry_split",
    "bplhip_threefry_bits",
)


class BplHipError(RuntimeError):
    """A libbplhip call failed (code < 0); the message is bplhip_last_error()."""

    def __init__(self, code: int, msg: str):
        super().__init__(f"libbplhip error {code}: {msg}")
        self.code = code


class NutsCfg(C.Structure):
    _fields_ = [
        ("num_warmup", C.c_int32),
        ("num_samples", C.c_int32),
        ("max_tree_depth", C.c_int32),
        ("adapt_step_size", C.c_int32),
        ("adapt_mass_matrix", C.c_int32),
        ("thinning", C.c_int32),
        ("step_size", C.c_double),
        ("target_accept_prob", C.c_double),
        ("init_radius", C.c_double),
        ("max_delta_energy", C.c_double),
    ]


class NutsStats(C.Structure):
    _fields_ = [
        ("potential_energy", C.POINTER(C.c_double)),
        ("accept_prob", C.POINTER(C.c_double)),
        ("step_size", C.POINTER(C.c_double)),
        ("num_steps", C.POINTER(C.c_int32)),
        ("diverging", C.POINTER(C.c_int32)),
        ("corr_coef", C.POINTER(C.c_double)),
        ("final_step_size", C.c_double),
        ("mean_accept_prob", C.c_double),
        ("total_leapfrogs", C.c_int64),
        ("total_divergences", C.c_int64),
        ("wall_seconds", C.c_double),
        ("inverse_mass_matrix", C.POINTER(C.c_double)),
    ]


def lib_path() -> str:
    return os.path.join(os.path.dirname(os.path.abspath(__file__)), _LIB_NAME)


def load_library():
    """dlopen libbplhip.so (built in-tree by `make -C bpl-next_amd/csrc`)."""
    global _lib
    if _lib is not None:
        return _lib
    path = lib_path()
    if not os.path.exists(path):
        raise ImportError(
            f"{path} not found: build the HIP extension first "
            "(python -c 'import __graft_entry__ as g; g.build()' or make -C bpl-next_amd/csrc)"
        )
    lib = C.CDLL(path)
    vp, i32, i64, u32 = C.c_void_p, C.c_int32, C.c_int64, C.c_uint32
    lib.bplhip_abi_version.restype = C.c_int
    lib.bplhip_create.argtypes = [C.POINTER(vp), C.c_int]
    lib.bplhip_create.restype = C.c_int
    lib.bplhip_destroy.argtypes = [vp]
    lib.bplhip_destroy.restype = None
    lib.bplhip_last_error.argtypes = [vp]
    lib.bplhip_last_error.restype = C.c_char_p
    lib.bplhip_set_fixtures.argtypes = [vp, C.c_int, i64, i32, vp, vp, vp, vp, vp, vp, i32, vp]
    lib.bplhip_set_fixtures.restype = C.c_int
    lib.bplhip_set_fixtures_dynamic.argtypes = [vp, i64, i32, i32, vp, vp, vp, vp, vp, vp, vp, i32, i32, vp]
    lib.bplhip_set_fixtures_dynamic.restype = C.c_int
    lib.bplhip_set_fixtures_neutral.argtypes = [vp, i64, i32, vp, vp, vp, vp, vp, vp, vp, i32, vp, vp, i32, vp]
    lib.bplhip_set_fixtures_neutral.restype = C.c_int
    lib.bplhip_constrain_dynamic.argtypes = [vp, vp, i64, vp, vp, vp, vp, vp, vp]
    lib.bplhip_constrain_dynamic.restype = C.c_int
    lib.bplhip_set_option.argtypes = [vp, C.c_char_p, C.c_int]
    lib.bplhip_set_option.restype = C.c_int
    lib.bplhip_latent_dim.argtypes = [vp]
    lib.bplhip_latent_dim.restype = C.c_int
    lib.bplhip_logp_grad.argtypes = [vp, vp, vp, vp, vp, vp]
    lib.bplhip_logp_grad.restype = C.c_int
    lib.bplhip_logp_grad_batched.argtypes = [vp, i32, vp, vp, vp, vp, vp]
    lib.bplhip_logp_grad_batched.restype = C.c_int
    lib.bplhip_logp_grad_graph.argtypes = [vp, i32, i32, vp, vp, vp, i32, vp]
    lib.bplhip_logp_grad_graph.restype = C.c_int
    lib.bplhip_nuts_default_cfg.argtypes = [C.POINTER(NutsCfg)]
    lib.bplhip_nuts_default_cfg.restype = None
    lib.bplhip_nuts_run.argtypes = [
        vp, C.POINTER(NutsCfg), vp, u32, u32, vp, C.POINTER(NutsStats), vp,
    ]
    lib.bplhip_nuts_run.restype = C.c_int
    lib.bplhip_nuts_run_chains.argtypes = [
        vp, C.POINTER(NutsCfg), i32, vp, vp, vp, C.POINTER(NutsStats), vp,
    ]
    lib.bplhip_nuts_run_chains.restype = C.c_int
    lib.bplhip_constrain.argtypes = [vp, vp, i64, vp, vp, vp, vp]
    lib.bplhip_constrain.restype = C.c_int
    lib.bplhip_predict_set_posterior.argtypes = [vp, i32, i32, vp, vp, vp, i32, vp]
    lib.bplhip_predict_set_posterior.restype = C.c_int
    lib.bplhip_predict_score_proba.argtypes = [vp, i64, vp, vp, vp, vp, vp, vp]
    lib.bplhip_predict_score_proba.restype = C.c_int
    lib.bplhip_predict_score_grid.argtypes = [vp, i64, vp, vp, i32, vp, vp]
    lib.bplhip_predict_score_grid.restype = C.c_int
    lib.bplhip_predict_score_grid_f32.argtypes = [vp, i64, vp, vp, i32, vp, vp]
    lib.bplhip_predict_score_grid_f32.restype = C.c_int
    lib.bplhip_predict_set_posterior_venue.argtypes = [vp, i32, i32, vp, vp, vp, vp, vp, vp, i32, vp, vp]
    lib.bplhip_predict_set_posterior_venue.restype = C.c_int
    lib.bplhip_predict_score_proba_venue.argtypes = [vp, i64, vp, vp, vp, vp, vp, vp, vp, vp, vp]
    lib.bplhip_predict_score_proba_venue.restype = C.c_int
    lib.bplhip_predict_score_grid_venue.argtypes = [vp, i64, vp, vp, vp, vp, vp, i32, vp, vp]
    lib.bplhip_predict_score_grid_venue.restype = C.c_int
    lib.bplhip_predict_score_grid_venue_f32.argtypes = [vp, i64, vp, vp, vp, vp, vp, i32, vp, vp]
    lib.bplhip_predict_score_grid_venue_f32.restype = C.c_int
    lib.bplhip_selftest_math.argtypes = [vp, i32, i64, vp, vp]
    lib.bplhip_selftest_math.restype = C.c_int
    lib.bplhip_threefry_split.argtypes = [u32, u32, i32, C.POINTER(u32)]
    lib.bplhip_threefry_split.restype = None
    lib.bplhip_threefry_bits.argtypes = [u32, u32, i32, C.POINTER(u32)]
    lib.bplhip_threefry_bits.restype = None
    if lib.bplhip_abi_version() != 1:
        raise ImportError(f"{path}: ABI version {lib.bplhip_abi_version()} != 1")
    _lib = lib
    return lib


def default_nuts_cfg() -> NutsCfg:
    cfg = NutsCfg()
    load_library().bplhip_nuts_default_cfg(C.byref(cfg))
    return cfg


def prng_key(seed: int) -> Tuple[int, int]:
    """jax.random.PRNGKey(seed) -> (hi, lo)."""
    seed = int(seed) & 0xFFFFFFFFFFFFFFFF
    return (seed >> 32) & 0xFFFFFFFF, seed & 0xFFFFFFFF


def threefry_split(key: Tuple[int, int], n: int):
    """jax.random.split(key, n) -> list of (hi, lo)."""
    out = (C.c_uint32 * (2 * n))()
    load_library().bplhip_threefry_split(key[0], key[1], n, out)
    return [(int(out[2 * i]), int(out[2 * i + 1])) for i in range(n)]


def threefry_bits(key: Tuple[int, int], n: int) -> np.ndarray:
    out = (C.c_uint32 * n)()
    load_library().bplhip_threefry_bits(key[0], key[1], n, out)
    return np.frombuffer(out, dtype=np.uint32).copy()


def _np_ptr(a: Optional[np.ndarray]):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


class HipContext:
    """One libbplhip context on one GPU.  All tensors are torch CUDA(=HIP) tensors."""

    def __init__(self, device_index: int = 0):
        import torch

        self._torch = torch
        self._lib = load_library()
        if not torch.cuda.is_available():
            raise RuntimeError(
                "bpl (MI355X build) needs a HIP GPU: torch.cuda.is_available() is False "
                "and there is no CPU fallback"
            )
        self.device = torch.device("cuda", device_index)
        h = C.c_void_p()
        rc = self._lib.bplhip_create(C.byref(h), device_index)
        if rc != 0:
            raise BplHipError(rc, self._lib.bplhip_last_error(None).decode())
        self._h = h
        self.dim = 0
        self.n_teams = 0
        self.model = None

    # -- plumbing
    def close(self):
        if getattr(self, "_h", None):
            self._lib.bplhip_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:  # pylint: disable=broad-except
            pass

    def _check(self, rc: int):
        if rc != 0:
            raise BplHipError(rc, self._lib.bplhip_last_error(self._h).decode())

    def _stream(self):
        return C.c_void_p(self._torch.cuda.current_stream(self.device).cuda_stream)

    def set_option(self, name: str, value: int):
        self._check(self._lib.bplhip_set_option(self._h, name.encode(), int(value)))

    # -- model arguments
    def set_fixtures(
        self,
        model: int,
        home_idx,
        away_idx,
        home_goals,
        away_goals,
        n_teams: int,
        weights=None,
        covariates_std: Optional[np.ndarray] = None,
    ):
        """Bind fixtures.  Index/goal arrays: torch tensors on this device (uint16 is
        stored as int16 bit patterns / uint8) or numpy arrays (uploaded here)."""
        torch = self._torch

        def dev(a, np_dtype, t_dtype):
            if isinstance(a, torch.Tensor):
                if a.dtype != t_dtype or a.device != self.device or not a.is_contiguous():
                    raise ValueError(f"tensor must be contiguous {t_dtype} on {self.device}")
                return a
            arr = np.ascontiguousarray(np.asarray(a).astype(np_dtype))
            return torch.from_numpy(arr.view(np.int16) if np_dtype == np.uint16 else arr).to(
                self.device
            )

        h = dev(home_idx, np.uint16, torch.int16)
        a = dev(away_idx, np.uint16, torch.int16)
        x = dev(home_goals, np.uint8, torch.uint8)
        y = dev(away_goals, np.uint8, torch.uint8)
        n = h.numel()
        if not (a.numel() == x.numel() == y.numel() == n):
            raise ValueError("fixture arrays must have equal length")
        w = None
        if weights is not None:
            w = dev(weights, np.float32, torch.float32)
            if w.numel() != n:
                raise ValueError("weights must have one entry per fixture")
        cov = None
        k = 0
        if covariates_std is not None:
            cov = np.ascontiguousarray(covariates_std, dtype=np.float64)
            if cov.ndim != 2 or cov.shape[0] != n_teams:
                raise ValueError("covariates must be [n_teams, k]")
            k = cov.shape[1]
        with torch.cuda.device(self.device):
            self._check(
                self._lib.bplhip_set_fixtures(
                    self._h, model, n, n_teams,
                    h.data_ptr(), a.data_ptr(), x.data_ptr(), y.data_ptr(),
                    None if w is None else w.data_ptr(), _np_ptr(cov), k, self._stream(),
                )
            )
        self.dim = self._lib.bplhip_latent_dim(self._h)
        self.n_teams = n_teams
        self.model = model
        self.n = n
        return self

    def set_fixtures_dynamic(self, home_idx, away_idx, home_goals, away_goals, gameweek,
                             neutral_venue, n_teams: int, n_gameweeks: int,
                             covariates_std: Optional[np.ndarray] = None, random_walk: bool = True):
        """Bind the dynamic (time-varying) model (bpl/dynamic_dixon_coles.py)."""
        torch = self._torch

        def dev(a, np_dtype):
            arr = np.ascontiguousarray(np.asarray(a).astype(np_dtype))
            return torch.from_numpy(arr.view(np.int16) if np_dtype == np.uint16 else arr).to(self.device)

        h, a, g = dev(home_idx, np.uint16), dev(away_idx, np.uint16), dev(gameweek, np.uint16)
        x, y, nv = dev(home_goals, np.uint8), dev(away_goals, np.uint8), dev(neutral_venue, np.uint8)
        n = h.numel()
        if not (a.numel() == g.numel() == x.numel() == y.numel() == nv.numel() == n):
            raise ValueError("fixture arrays must have equal length")
        cov, k = None, 0
        if covariates_std is not None:
            cov = np.ascontiguousarray(covariates_std, dtype=np.float64)
            k = cov.shape[1]
        with torch.cuda.device(self.device):
            self._check(self._lib.bplhip_set_fixtures_dynamic(
                self._h, n, n_teams, n_gameweeks, h.data_ptr(), a.data_ptr(), x.data_ptr(),
                y.data_ptr(), g.data_ptr(), nv.data_ptr(), _np_ptr(cov), k, int(random_walk),
                self._stream()))
        self.dim = self._lib.bplhip_latent_dim(self._h)
        self.n_teams, self.n_gameweeks, self.model, self.n = n_teams, n_gameweeks, MODEL_DYNAMIC, n
        return self

    def set_fixtures_neutral(self, home_idx, away_idx, home_goals, away_goals, neutral_venue,
                             n_teams: int, weights=None, covariates_std: Optional[np.ndarray] = None,
                             home_conf=None, away_conf=None, n_conf: int = 0):
        """Bind the neutral-venue model (bpl/neutral_dixon_coles.py).  `weights`: the final
        per-fixture weights (time decay x game weights) or None.  `home_conf`, `away_conf`,
        `n_conf`: confederation indices of the World-Cup variant."""
        torch = self._torch

        def dev(a, np_dtype):
            arr = np.ascontiguousarray(np.asarray(a).astype(np_dtype))
            return torch.from_numpy(arr.view(np.int16) if np_dtype == np.uint16 else arr).to(self.device)

        h, a = dev(home_idx, np.uint16), dev(away_idx, np.uint16)
        x, y, nv = dev(home_goals, np.uint8), dev(away_goals, np.uint8), dev(neutral_venue, np.uint8)
        n = h.numel()
        if not (a.numel() == x.numel() == y.numel() == nv.numel() == n):
            raise ValueError("fixture arrays must have equal length")
        w = None
        if weights is not None:
            w = dev(weights, np.float32)
            if w.numel() != n:
                raise ValueError("weights must have one entry per fixture")
        cov, k = None, 0
        if covariates_std is not None:
            cov = np.ascontiguousarray(covariates_std, dtype=np.float64)
            k = cov.shape[1]
        hc = ac = None
        if n_conf:
            hc, ac = dev(home_conf, np.uint8), dev(away_conf, np.uint8)
            if hc.numel() != n or ac.numel() != n:
                raise ValueError("confederation arrays must have one entry per fixture")
        with torch.cuda.device(self.device):
            self._check(self._lib.bplhip_set_fixtures_neutral(
                self._h, n, n_teams, h.data_ptr(), a.data_ptr(), x.data_ptr(), y.data_ptr(),
                nv.data_ptr(), None if hc is None else hc.data_ptr(),
                None if ac is None else ac.data_ptr(), int(n_conf),
                None if w is None else w.data_ptr(), _np_ptr(cov), k, self._stream()))
        self.dim = self._lib.bplhip_latent_dim(self._h)
        self.n_teams, self.model, self.n = n_teams, MODEL_NEUTRAL, n
        return self

    def constrain_dynamic(self, z_draws: np.ndarray):
        z = np.ascontiguousarray(z_draws, dtype=np.float64)
        s, g, t = z.shape[0], self.n_gameweeks, self.n_teams
        names = ("attack", "defence", "home_attack", "away_attack", "home_defence", "away_defence")
        out = {nm: np.empty((s, g, t)) for nm in names}
        self._check(self._lib.bplhip_constrain_dynamic(self._h, _np_ptr(z), s,
                                                       *[_np_ptr(out[nm]) for nm in names]))
        return out

    # -- the hot path
    def logp_grad(self, z, potential=None, grad=None, aux=None):
        """U(z), dU/dz for z [D] or [C, D] (float64 tensors on this device)."""
        torch = self._torch
        if z.dtype != torch.float64 or not z.is_contiguous() or z.device != self.device:
            raise ValueError("z must be a contiguous float64 tensor on the context device")
        batched = z.dim() == 2
        c = z.shape[0] if batched else 1
        if z.shape[-1] != self.dim:
            raise ValueError(f"z has {z.shape[-1]} columns, model has D={self.dim}")
        if potential is None:
            potential = torch.empty(c, dtype=torch.float64, device=self.device)
        if grad is None:
            grad = torch.empty_like(z)
        if aux is None:
            aux = torch.empty((c, 4), dtype=torch.float64, device=self.device)
        self._check(
            self._lib.bplhip_logp_grad_batched(
                self._h, c, z.data_ptr(), potential.data_ptr(), grad.data_ptr(),
                aux.data_ptr(), self._stream(),
            )
        )
        return potential, grad, aux

    def logp_grad_graph(self, count: int, z, potential, grad, replays: int = 1):
        """Replay a captured chain of `count` evaluations over the rows of z [n_z, D]."""
        self._check(
            self._lib.bplhip_logp_grad_graph(
                self._h, count, z.shape[0], z.data_ptr(), potential.data_ptr(),
                grad.data_ptr(), replays, self._stream(),
            )
        )

    def graph_replayer(self, count: int, z, potential, grad):
        """`logp_grad_graph` with its arguments marshalled once: returns `replay(replays=1)` for the
        current stream (the per-call Python cost -- stream lookup, three data_ptr(), argument
        conversion -- is ~5 us, which a caller replaying short graphs back to back can skip)."""
        fn, check = self._lib.bplhip_logp_grad_graph, self._check
        args = (self._h, C.c_int32(count), C.c_int32(z.shape[0]), C.c_void_p(z.data_ptr()),
                C.c_void_p(potential.data_ptr()), C.c_void_p(grad.data_ptr()))
        stream = self._stream()
        keep = (z, potential, grad)

        def replay(replays: int = 1, _keep=keep):
            check(fn(*args, replays, stream))

        return replay

    # -- predict path on the device
    def predict_set_posterior(self, attack, defence, home_advantage, corr_coef):
        att = np.ascontiguousarray(attack, dtype=np.float64)
        dfn = np.ascontiguousarray(defence, dtype=np.float64)
        ha = np.ascontiguousarray(home_advantage, dtype=np.float64)
        cc = np.ascontiguousarray(corr_coef, dtype=np.float64)
        s, t = att.shape
        if dfn.shape != (s, t) or cc.shape != (s,) or ha.shape not in ((s,), (s, t)):
            raise ValueError("posterior arrays have inconsistent shapes")
        with self._torch.cuda.device(self.device):
            self._check(self._lib.bplhip_predict_set_posterior(
                self._h, s, t, _np_ptr(att), _np_ptr(dfn), _np_ptr(ha), int(ha.ndim == 2), _np_ptr(cc)))

    def predict_set_posterior_venue(self, attack, defence, home_attack, away_attack, home_defence,
                                    away_defence, corr_coef, confederation_strength=None):
        """Posterior of the neutral-venue family (six [draws, teams] tables, optional
        [draws, confederations] strengths); queries then take `neutral` (and `conf`)."""
        tabs = [np.ascontiguousarray(t, dtype=np.float64)
                for t in (attack, defence, home_attack, away_attack, home_defence, away_defence)]
        cc = np.ascontiguousarray(corr_coef, dtype=np.float64)
        s, t = tabs[0].shape
        if any(x.shape != (s, t) for x in tabs) or cc.shape != (s,):
            raise ValueError("posterior arrays have inconsistent shapes")
        conf = None
        if confederation_strength is not None:
            conf = np.ascontiguousarray(confederation_strength, dtype=np.float64)
            if conf.ndim != 2 or conf.shape[0] != s:
                raise ValueError("confederation_strength must be [draws, confederations]")
        with self._torch.cuda.device(self.device):
            self._check(self._lib.bplhip_predict_set_posterior_venue(
                self._h, s, t, *(_np_ptr(x) for x in tabs), 0 if conf is None else conf.shape[1],
                None if conf is None else _np_ptr(conf), _np_ptr(cc)))

    @staticmethod
    def _venue_args(m, neutral, conf):
        nv = np.ascontiguousarray(np.broadcast_to(np.asarray(neutral), (m,)), dtype=np.uint8)
        if conf is None:
            return nv, None, None
        hc = np.ascontiguousarray(np.broadcast_to(np.asarray(conf[0]), (m,)), dtype=np.uint16)
        ac = np.ascontiguousarray(np.broadcast_to(np.asarray(conf[1]), (m,)), dtype=np.uint16)
        return nv, hc, ac

    def predict_score_proba(self, home_idx, away_idx, home_goals, away_goals, neutral=None,
                            conf=None) -> np.ndarray:
        """Mean over the draws of tau * Poisson * Poisson per query.  `neutral` (0/1 per query) and
        `conf` = (home, away confederation indices) select the venue-aware rates and must be given
        exactly when the posterior was set with predict_set_posterior_venue."""
        h = np.ascontiguousarray(home_idx, dtype=np.uint16)
        a = np.ascontiguousarray(away_idx, dtype=np.uint16)
        x = np.ascontiguousarray(home_goals, dtype=np.uint16)
        y = np.ascontiguousarray(away_goals, dtype=np.uint16)
        m = h.size
        if not (a.size == x.size == y.size == m):
            raise ValueError("query arrays must have equal length")
        out = np.empty(m, dtype=np.float64)
        with self._torch.cuda.device(self.device):
            if neutral is None:
                self._check(self._lib.bplhip_predict_score_proba(
                    self._h, m, _np_ptr(h), _np_ptr(a), _np_ptr(x), _np_ptr(y), _np_ptr(out), self._stream()))
            else:
                nv, hc, ac = self._venue_args(m, neutral, conf)
                self._check(self._lib.bplhip_predict_score_proba_venue(
                    self._h, m, _np_ptr(h), _np_ptr(a), _np_ptr(x), _np_ptr(y), _np_ptr(nv),
                    None if hc is None else _np_ptr(hc), None if ac is None else _np_ptr(ac),
                    _np_ptr(out), self._stream()))
        return out

    def predict_score_grid(self, home_idx, away_idx, max_goals: int, neutral=None, conf=None,
                           dtype=np.float64) -> np.ndarray:
        """[m, max_goals+1, max_goals+1] scoreline probabilities of the m fixtures; dtype float64 (default) or
        float32 (the reference's own: half the bytes over PCIe)."""
        h = np.ascontiguousarray(home_idx, dtype=np.uint16)
        a = np.ascontiguousarray(away_idx, dtype=np.uint16)
        if h.size != a.size:
            raise ValueError("home and away index arrays must have equal length")
        dtype = np.dtype(dtype)
        if dtype not in (np.dtype(np.float64), np.dtype(np.float32)):
            raise ValueError("predict_score_grid: dtype is float64 or float32")
        f32 = dtype == np.dtype(np.float32)
        g1 = int(max_goals) + 1
        out = np.empty((h.size, g1, g1), dtype=dtype)
        with self._torch.cuda.device(self.device):
            if neutral is None:
                fn = self._lib.bplhip_predict_score_grid_f32 if f32 else self._lib.bplhip_predict_score_grid
                self._check(fn(
                    self._h, h.size, _np_ptr(h), _np_ptr(a), int(max_goals), _np_ptr(out), self._stream()))
            else:
                nv, hc, ac = self._venue_args(h.size, neutral, conf)
                fn = self._lib.bplhip_predict_score_grid_venue_f32 if f32 else self._lib.bplhip_predict_score_grid_venue
                self._check(fn(
                    self._h, h.size, _np_ptr(h), _np_ptr(a), _np_ptr(nv),
                    None if hc is None else _np_ptr(hc), None if ac is None else _np_ptr(ac),
                    int(max_goals), _np_ptr(out), self._stream()))
        return out

    def selftest_math(self, which: int, x) -> np.ndarray:
        """The library's short float64 device math on x: 0 exp, 1 log, 2 log1p (x >= 0), 3 1/x."""
        x = np.ascontiguousarray(x, dtype=np.float64)
        out = np.empty_like(x)
        with self._torch.cuda.device(self.device):
            self._check(self._lib.bplhip_selftest_math(self._h, int(which), x.size, _np_ptr(x), _np_ptr(out)))
        return out

    # -- sampler
    @staticmethod
    def _stats_buffers(kept: int, d: int):
        out = {
            "potential_energy": np.empty(kept),
            "accept_prob": np.empty(kept),
            "step_size": np.empty(kept),
            "num_steps": np.empty(kept, dtype=np.int32),
            "diverging": np.empty(kept, dtype=np.int32),
            "corr_coef": np.empty(kept),
            "inverse_mass_matrix": np.empty(d),
        }
        st = NutsStats()
        dp = C.POINTER(C.c_double)
        ip = C.POINTER(C.c_int32)
        st.potential_energy = out["potential_energy"].ctypes.data_as(dp)
        st.accept_prob = out["accept_prob"].ctypes.data_as(dp)
        st.step_size = out["step_size"].ctypes.data_as(dp)
        st.num_steps = out["num_steps"].ctypes.data_as(ip)
        st.diverging = out["diverging"].ctypes.data_as(ip)
        st.corr_coef = out["corr_coef"].ctypes.data_as(dp)
        st.inverse_mass_matrix = out["inverse_mass_matrix"].ctypes.data_as(dp)
        return out, st

    @staticmethod
    def _stats_scalars(out, st):
        out.update(
            final_step_size=st.final_step_size,
            mean_accept_prob=st.mean_accept_prob,
            total_leapfrogs=int(st.total_leapfrogs),
            total_divergences=int(st.total_divergences),
            wall_seconds=st.wall_seconds,
        )

    def nuts_run(self, cfg: NutsCfg, key: Tuple[int, int], z0: Optional[np.ndarray] = None):
        kept = cfg.num_samples // cfg.thinning
        d = self.dim
        draws = np.empty((kept, d), dtype=np.float64)
        out, st = self._stats_buffers(kept, d)
        z0c = None if z0 is None else np.ascontiguousarray(z0, dtype=np.float64)
        if z0c is not None and z0c.shape != (d,):
            raise ValueError(f"init_params must have shape ({d},)")
        with self._torch.cuda.device(self.device):
            self._check(
                self._lib.bplhip_nuts_run(
                    self._h, C.byref(cfg), _np_ptr(z0c), key[0], key[1],
                    draws.ctypes.data_as(C.c_void_p), C.byref(st), self._stream(),
                )
            )
        self._stats_scalars(out, st)
        return draws, out

    def nuts_run_chains(self, cfg: NutsCfg, keys, z0: Optional[np.ndarray] = None):
        """Lock-step chains on this GPU (numpyro chain_method="vectorized").  Returns a list
        of (draws, stats) per chain, same content as nuts_run.  Raises BplHipError with
        code EUNSUPPORTED when the bound model cannot run in lock step."""
        n = len(keys)
        kept = cfg.num_samples // cfg.thinning
        d = self.dim
        draws = np.empty((n, kept, d), dtype=np.float64)
        bufs = [self._stats_buffers(kept, d) for _ in range(n)]
        st_arr = (NutsStats * n)()
        for i, (_, st) in enumerate(bufs):
            st_arr[i] = st
        seeds = np.ascontiguousarray(np.asarray(keys, dtype=np.uint32).reshape(n, 2))
        z0c = None if z0 is None else np.ascontiguousarray(z0, dtype=np.float64)
        if z0c is not None:
            if z0c.shape == (d,):
                z0c = np.ascontiguousarray(np.tile(z0c, (n, 1)))
            if z0c.shape != (n, d):
                raise ValueError(f"init_params must have shape ({n}, {d})")
        with self._torch.cuda.device(self.device):
            self._check(
                self._lib.bplhip_nuts_run_chains(
                    self._h, C.byref(cfg), n, _np_ptr(z0c), _np_ptr(seeds),
                    draws.ctypes.data_as(C.c_void_p), st_arr, self._stream(),
                )
            )
        res = []
        for i, (out, _) in enumerate(bufs):
            self._stats_scalars(out, st_arr[i])
            res.append((draws[i], out))
        return res

    def constrain(self, z_draws: np.ndarray):
        z = np.ascontiguousarray(z_draws, dtype=np.float64)
        s, t = z.shape[0], self.n_teams
        attack = np.empty((s, t))
        defence = np.empty((s, t))
        ha = np.empty(s) if self.model == MODEL_BASIC else np.empty((s, t))
        corr = np.empty(s)
        self._check(
            self._lib.bplhip_constrain(
                self._h, _np_ptr(z), s, _np_ptr(attack), _np_ptr(defence), _np_ptr(ha),
                _np_ptr(corr),
            )
        )
        return {"attack": attack, "defence": defence, "home_advantage": ha, "corr_coef": corr}
