"""TEST INFRASTRUCTURE -- an independent float64 numpy restatement of numpyro 0.13.2's NUTS.

The reference builds its sampler with `NUTS(self._model)` / `MCMC(...).run(PRNGKey(s), ...)`
(bpl/dixon_coles.py:100-116); numpyro and jax are pip dependencies that are not in the
reference tree (poetry.lock: numpyro 0.13.2, jax 0.4.24) and not importable here, so this file
restates their *published* algorithm: numpyro/infer/hmc.py (init_kernel, sample_kernel,
_nuts_next), numpyro/infer/hmc_util.py (velocity_verlet, build_tree, _double_tree,
_iterative_build_subtree, _build_basetree, _combine_tree, _is_turning,
_leaf_idx_to_ckpt_idxs, dual_averaging, welford_covariance, build_adaptation_schedule,
warmup_adapter), numpyro/infer/util.py (find_valid_initial_params, the init_to_uniform
fast branch), jax/_src/random.py + prng.py (threefry2x32, split, bits, uniform, normal,
bernoulli).  It was written from SURVEY.md Appendix B and the published sources' structure
WITHOUT reading the product's driver (bpl-next_amd/csrc/nuts.hpp, nuts_dev.hip.h): it exists
so that the product's tree builder is checked against something that is not itself.

PARITY UNPINNED against numpyro itself (nothing reference-held pins a sampler trajectory:
the reference's tests are property tests after a fit).  What IS pinned
from outside: the Threefry block (Random123 known answers) and jax's published values for
split / bits / normal (tests/test_nuts_oracle.py).

Only tests/ may import this module.

Differences from numpyro that are deliberate and shared with the product: the chain state is
float64 (numpyro: float32); uniform draws are jax's float32 values exactly (mantissa trick);
a normal draw is sqrt(2) * erfinv(u) of jax's float32 u evaluated in float64 -- jax evaluates a
float32 polynomial for erfinv, so its draw equals this one up to the last float32 bit.
"""
import math

import numpy as np
from scipy.special import erfinv, expit

U32 = np.uint32
MASK = 0xFFFFFFFF

# ------------------------------------------------------------------ threefry2x32 / jax.random


def _rotl(x, r):
    return ((x << r) | (x >> (32 - r))) & MASK


def threefry_block(k0, k1, c0, c1):
    """Threefry-2x32, 20 rounds (Salmon et al. 2011; jax/_src/prng.py threefry2x32)."""
    ks = (k0, k1, k0 ^ k1 ^ 0x1BD11BDA)
    rot = ((13, 15, 26, 6), (17, 29, 16, 24))
    x0 = (c0 + ks[0]) & MASK
    x1 = (c1 + ks[1]) & MASK
    for g in range(5):
        for r in rot[g % 2]:
            x0 = (x0 + x1) & MASK
            x1 = _rotl(x1, r)
            x1 ^= x0
        x0 = (x0 + ks[(g + 1) % 3]) & MASK
        x1 = (x1 + ks[(g + 2) % 3] + g + 1) & MASK
    return x0, x1


def _hash_counts(key, counts):
    """threefry_2x32(key, counts): odd lengths are zero padded, the two halves of the padded
    vector are the block inputs, the two output halves are concatenated and cut."""
    n = len(counts)
    c = list(counts) + [0] * (n % 2)
    m = len(c) // 2
    o0, o1 = [], []
    for i in range(m):
        a, b = threefry_block(key[0], key[1], c[i], c[m + i])
        o0.append(a)
        o1.append(b)
    return (o0 + o1)[:n]


def prng_key(seed):
    seed = int(seed) & 0xFFFFFFFFFFFFFFFF
    return ((seed >> 32) & MASK, seed & MASK)


def random_bits(key, n):
    return _hash_counts(key, range(n))


def split(key, num=2):
    flat = _hash_counts(key, range(2 * num))
    return [(flat[2 * i], flat[2 * i + 1]) for i in range(num)]


def _unit_floats(bits):
    """float32 in [0, 1): (bits >> 9) | bits(1.0f), minus one."""
    fb = (np.asarray(bits, dtype=np.uint64).astype(np.uint32) >> U32(9)) | U32(0x3F800000)
    return fb.view(np.float32) - np.float32(1.0)


def uniform(key, n, minval=0.0, maxval=1.0):
    """jax.random.uniform(key, (n,), float32, minval, maxval), float32 arithmetic."""
    lo, hi = np.float32(minval), np.float32(maxval)
    f = _unit_floats(random_bits(key, n))
    return np.maximum(lo, f * np.float32(hi - lo) + lo).astype(np.float32)


def normal(key, n, float32_result=False):
    """jax.random.normal(key, (n,), float32) = sqrt(2) * erfinv(u), u = uniform(key, (n,),
    nextafter(-1, 0), 1) in float32.  Default: the float64 inverse of the float32 u (the chain
    state is float64; jax itself evaluates a float32 polynomial, so its value is this one up to
    the last float32 bit).  float32_result=True rounds like jax's output dtype."""
    lo = np.nextafter(np.float32(-1.0), np.float32(0.0))
    u = uniform(key, n, lo, 1.0)
    v = math.sqrt(2.0) * erfinv(u.astype(np.float64))
    if float32_result:
        return (np.float32(math.sqrt(2.0)) * erfinv(u.astype(np.float64)).astype(np.float32)).astype(np.float32)
    return v


def bernoulli(key, p):
    """jax.random.bernoulli(key, p) for a scalar p: uniform(key, ()) < p."""
    return bool(np.float64(uniform(key, 1)[0]) < p)


# ------------------------------------------------------------------ adaptation


def build_adaptation_schedule(num_steps):
    """Stan's windowed schedule as numpyro builds it: list of (start, end) inclusive."""
    if num_steps < 20:
        return [(0, num_steps - 1)]
    start_buffer, end_buffer, init_window = 75, 50, 25
    if start_buffer + end_buffer + init_window > num_steps:
        start_buffer = int(0.15 * num_steps)
        end_buffer = int(0.1 * num_steps)
        init_window = num_steps - start_buffer - end_buffer
    out = [(0, start_buffer - 1)]
    end_window_start = num_steps - end_buffer
    next_size, next_start = init_window, start_buffer
    while next_start < end_window_start:
        cur_start, cur_size = next_start, next_size
        if 3 * cur_size <= end_window_start - cur_start:
            next_size = 2 * cur_size
        else:
            cur_size = end_window_start - cur_start
        next_start = cur_start + cur_size
        out.append((cur_start, next_start - 1))
    out.append((end_window_start, num_steps - 1))
    return out


class DualAveraging:
    """Nesterov dual averaging of log(step size) (Hoffman & Gelman 2014, section 3.2.1) with
    numpyro's constants t0 = 10, kappa = 0.75, gamma = 0.05."""

    T0, KAPPA, GAMMA = 10.0, 0.75, 0.05

    def __init__(self, prox_center):
        self.x = 0.0
        self.x_avg = 0.0
        self.g_avg = 0.0
        self.t = 0
        self.prox_center = prox_center

    def update(self, g):
        self.t += 1
        t = self.t
        self.g_avg = (1.0 - 1.0 / (t + self.T0)) * self.g_avg + g / (t + self.T0)
        self.x = self.prox_center - math.sqrt(t) / self.GAMMA * self.g_avg
        w = t ** (-self.KAPPA)
        self.x_avg = (1.0 - w) * self.x_avg + w * self.x


class Welford:
    """Running diagonal variance; `final` applies Stan's shrinkage."""

    def __init__(self, d):
        self.mean = np.zeros(d)
        self.m2 = np.zeros(d)
        self.n = 0

    def update(self, x):
        self.n += 1
        d0 = x - self.mean
        self.mean = self.mean + d0 / self.n
        self.m2 = self.m2 + d0 * (x - self.mean)

    def final(self, regularize=True):
        var = self.m2 / (self.n - 1)
        if regularize:
            var = (self.n / (self.n + 5.0)) * var + 1e-3 * (5.0 / (self.n + 5.0))
        return var


TINY32 = float(np.finfo(np.float32).tiny)
MAX32 = float(np.finfo(np.float32).max)


class WarmupAdapter:
    """numpyro.infer.hmc_util.warmup_adapter with the defaults bpl reaches (adapt both,
    diagonal mass, regularised, no heuristic step size search)."""

    def __init__(self, key, num_warmup, step_size, d, target=0.8, adapt_step_size=True,
                 adapt_mass_matrix=True):
        self.num_warmup = num_warmup
        self.schedule = build_adaptation_schedule(num_warmup)
        self.target = target
        self.adapt_step_size = adapt_step_size
        self.adapt_mass_matrix = adapt_mass_matrix
        self.key, _unused = split(key)  # rng_key, rng_key_ss = split(rng_key)
        self.step_size = step_size
        self.inv_mass = np.ones(d)
        self.ss = DualAveraging(math.log(10.0 * step_size))
        self.mm = Welford(d)
        self.window = 0

    def update(self, t, accept_prob, z):
        self.key, _key_ss = split(self.key)
        if self.adapt_step_size:
            self.ss.update(self.target - accept_prob)
            log_ss = self.ss.x_avg if t == self.num_warmup - 1 else self.ss.x
            self.step_size = min(max(math.exp(log_ss) if log_ss < 700 else math.inf, TINY32), MAX32)
        middle = 0 < self.window < len(self.schedule) - 1
        if self.adapt_mass_matrix and middle:
            self.mm.update(z)
        at_end = t == self.schedule[self.window][1]
        if at_end:
            self.window += 1
        if at_end and middle:
            if self.adapt_mass_matrix:
                self.inv_mass = self.mm.final(True)
                self.mm = Welford(len(z))
            if self.adapt_step_size:
                self.ss = DualAveraging(math.log(10.0 * self.step_size))


# ------------------------------------------------------------------ the tree


def leaf_idx_to_ckpt_idxs(n):
    idx_max = bin(n >> 1).count("1")
    num_subtrees = 0
    m = n
    while m & 1:
        m >>= 1
        num_subtrees += 1
    return idx_max - num_subtrees + 1, idx_max


def is_turning(inv_mass, r_left, r_right, r_sum):
    rs = r_sum - (r_left + r_right) / 2.0
    return bool(np.dot(inv_mass * r_left, rs) <= 0.0) or bool(np.dot(inv_mass * r_right, rs) <= 0.0)


class Tree:
    __slots__ = ("z_left", "r_left", "g_left", "z_right", "r_right", "g_right", "z_prop", "pe_prop",
                 "g_prop", "e_prop", "depth", "weight", "r_sum", "turning", "diverging", "sum_accept",
                 "num_proposals")

    def copy(self):
        t = Tree()
        for k in self.__slots__:
            setattr(t, k, getattr(self, k))
        return t


def kinetic(inv_mass, r):
    return 0.5 * float(np.dot(inv_mass * r, r))


def leapfrog(pot, eps, inv_mass, z, r, g):
    r = r - 0.5 * eps * g
    z = z + eps * (inv_mass * r)
    pe, g = pot(z)
    r = r - 0.5 * eps * g
    return z, r, pe, g


def build_basetree(pot, z, r, g, inv_mass, step_size, going_right, energy_current, max_delta_energy):
    eps = step_size if going_right else -step_size
    z, r, pe, g = leapfrog(pot, eps, inv_mass, z, r, g)
    energy = pe + kinetic(inv_mass, r)
    delta = energy - energy_current
    if math.isnan(delta):
        delta = math.inf
    t = Tree()
    t.z_left = t.z_right = t.z_prop = z
    t.r_left = t.r_right = r
    t.g_left = t.g_right = t.g_prop = g
    t.pe_prop, t.e_prop = pe, energy
    t.depth, t.weight, t.r_sum = 0, -delta, r
    t.turning, t.diverging = False, delta > max_delta_energy
    t.sum_accept = min(1.0, math.exp(-delta)) if delta > -700 else 1.0
    t.num_proposals = 1
    return t


def _logaddexp(a, b):
    m = max(a, b)
    if m == -math.inf:
        return -math.inf
    return m + math.log(math.exp(a - m) + math.exp(b - m))


def combine_tree(cur, new, inv_mass, going_right, key, biased):
    out = Tree()
    if going_right:
        out.z_left, out.r_left, out.g_left = cur.z_left, cur.r_left, cur.g_left
        out.z_right, out.r_right, out.g_right = new.z_right, new.r_right, new.g_right
    else:
        out.z_left, out.r_left, out.g_left = new.z_left, new.r_left, new.g_left
        out.z_right, out.r_right, out.g_right = cur.z_right, cur.r_right, cur.g_right
    out.r_sum = cur.r_sum + new.r_sum
    if biased:  # main tree: min(1, w_new / w_cur), never into a turning / diverging subtree
        d = new.weight - cur.weight
        p = 0.0 if (new.turning or new.diverging) else min(1.0, math.exp(d) if d < 700 else math.inf)
        out.turning = new.turning or is_turning(inv_mass, out.r_left, out.r_right, out.r_sum)
    else:  # inside a subtree: w_new / (w_new + w_cur)
        p = float(expit(new.weight - cur.weight))
        out.turning = cur.turning
    take_new = bernoulli(key, p)
    src = new if take_new else cur
    out.z_prop, out.pe_prop, out.g_prop, out.e_prop = src.z_prop, src.pe_prop, src.g_prop, src.e_prop
    out.depth = cur.depth + 1
    out.weight = _logaddexp(cur.weight, new.weight)
    out.diverging = new.diverging
    out.sum_accept = cur.sum_accept + new.sum_accept
    out.num_proposals = cur.num_proposals + new.num_proposals
    return out


def iterative_build_subtree(pot, proto, inv_mass, step_size, going_right, key, energy_current,
                            max_delta_energy, max_depth):
    d = len(proto.z_left)
    r_ckpts = np.zeros((max_depth, d))
    r_sum_ckpts = np.zeros((max_depth, d))
    max_num = 2 ** proto.depth
    tree = proto.copy()
    tree.num_proposals = 0
    turning = False
    while tree.num_proposals < max_num and not turning and not tree.diverging:
        key, tkey = split(key)
        if going_right:
            z, r, g = tree.z_right, tree.r_right, tree.g_right
        else:
            z, r, g = tree.z_left, tree.r_left, tree.g_left
        leaf = build_basetree(pot, z, r, g, inv_mass, step_size, going_right, energy_current,
                              max_delta_energy)
        leaf_idx = tree.num_proposals
        new_tree = leaf if leaf_idx == 0 else combine_tree(tree, leaf, inv_mass, going_right, tkey, False)
        idx_min, idx_max = leaf_idx_to_ckpt_idxs(leaf_idx)
        if leaf_idx % 2 == 0:
            r_ckpts[idx_max] = leaf.r_right
            r_sum_ckpts[idx_max] = new_tree.r_sum
        turning = False
        i = idx_max
        while i >= idx_min and not turning:
            sub_sum = new_tree.r_sum - r_sum_ckpts[i] + r_ckpts[i]
            turning = is_turning(inv_mass, r_ckpts[i], leaf.r_right, sub_sum)
            i -= 1
        tree = new_tree
    tree = tree.copy()
    tree.depth = proto.depth
    tree.turning = turning
    return tree


def build_tree(pot, z, r, pe, g, inv_mass, step_size, key, max_delta_energy=1000.0, max_depth=10):
    energy_current = pe + kinetic(inv_mass, r)
    tree = Tree()
    tree.z_left = tree.z_right = tree.z_prop = z
    tree.r_left = tree.r_right = tree.r_sum = r
    tree.g_left = tree.g_right = tree.g_prop = g
    tree.pe_prop, tree.e_prop = pe, energy_current
    tree.depth, tree.weight = 0, 0.0
    tree.turning = tree.diverging = False
    tree.sum_accept, tree.num_proposals = 0.0, 0
    while tree.depth < max_depth and not tree.turning and not tree.diverging:
        key, direction_key, doubling_key = split(key, 3)
        going_right = bernoulli(direction_key, 0.5)
        sub_key, transition_key = split(doubling_key)
        new_tree = iterative_build_subtree(pot, tree, inv_mass, step_size, going_right, sub_key,
                                           energy_current, max_delta_energy, max_depth)
        tree = combine_tree(tree, new_tree, inv_mass, going_right, transition_key, True)
    return tree


# ------------------------------------------------------------------ initial point


def init_to_uniform(key, site_shapes, radius=2.0):
    """One attempt of find_valid_initial_params' init_to_uniform branch (numpyro/infer/util.py:
    "this branch doesn't require tracing the model").  `site_shapes`: [(name, size)] in MODEL
    TRACE order.  Returns (carried key, {name: float32-valued array})."""
    key, sub = split(key)
    out = {}
    for name, size in site_shapes:
        out[name] = uniform(sub, size, -radius, radius).astype(np.float64)
        key, sub = split(key)
    return key, out


def find_valid_initial_params(pot, key, site_shapes, radius=2.0):
    """Retry (at most 100 times) until the potential and its gradient are finite.  The flat
    vector is in sorted-site-name order (ravel_pytree of a dict)."""
    z = None
    for _ in range(100):
        key, vals = init_to_uniform(key, site_shapes, radius)
        z = np.concatenate([vals[k] for k in sorted(vals)])
        pe, g = pot(z)
        if np.isfinite(pe) and np.isfinite(g).all():
            return z, True
    return z, False


# ------------------------------------------------------------------ the chain


def run_chain(pot, key, num_warmup, num_samples, z0=None, site_shapes=None, step_size=1.0,
              max_depth=10, max_delta_energy=1000.0, target_accept=0.8, adapt_step_size=True,
              adapt_mass_matrix=True, thinning=1):
    """MCMC(NUTS(...), num_warmup, num_samples).run(key) for one chain.  Returns a dict with
    the post-warm-up draws and per-iteration statistics of EVERY iteration (warm-up
    included): num_steps, accept_prob, diverging, step_size (used for the iteration),
    potential_energy, depth."""
    key, key_init_model = split(key)                         # NUTS.init
    if z0 is None:
        z0, ok = find_valid_initial_params(pot, key_init_model, site_shapes)
        if not ok:
            raise RuntimeError("no finite initial point after 100 tries")
    z = np.asarray(z0, dtype=np.float64).copy()
    d = z.size
    key_hmc, key_wa, _key_momentum = split(key, 3)           # hmc init_kernel
    wa = WarmupAdapter(key_wa, num_warmup, step_size, d, target_accept, adapt_step_size,
                       adapt_mass_matrix)
    pe, g = pot(z)
    key = key_hmc
    n_iter = num_warmup + num_samples
    stats = {k: [] for k in ("num_steps", "accept_prob", "diverging", "step_size", "potential_energy",
                             "depth")}
    draws = []
    mean_accept = 0.0
    for i in range(n_iter):
        key, key_momentum, key_transition = split(key, 3)    # sample_kernel
        mass_sqrt = 1.0 / np.sqrt(wa.inv_mass)
        r = mass_sqrt * normal(key_momentum, d).astype(np.float64)
        eps = wa.step_size
        tree = build_tree(pot, z, r, pe, g, wa.inv_mass, eps, key_transition, max_delta_energy, max_depth)
        accept_prob = tree.sum_accept / tree.num_proposals
        z, pe, g = tree.z_prop, tree.pe_prop, tree.g_prop
        stats["num_steps"].append(tree.num_proposals)
        stats["accept_prob"].append(accept_prob)
        stats["diverging"].append(bool(tree.diverging))
        stats["step_size"].append(eps)
        stats["potential_energy"].append(pe)
        stats["depth"].append(tree.depth)
        if i < num_warmup:
            wa.update(i, accept_prob, z)
        n = i + 1 if i < num_warmup else i + 1 - num_warmup
        mean_accept = mean_accept + (accept_prob - mean_accept) / n if n == 1 or True else mean_accept
        if i >= num_warmup and (i - num_warmup + 1) % thinning == 0:
            draws.append(z.copy())
    out = {k: np.array(v) for k, v in stats.items()}
    out["draws"] = np.array(draws).reshape(len(draws), d)
    out["final_step_size"] = wa.step_size
    out["inverse_mass_matrix"] = wa.inv_mass.copy()
    out["z0"] = np.asarray(z0, dtype=np.float64)
    return out


# ------------------------------------------------------------------ the reference's models

def site_shapes(model, n_teams, k=0):
    """Latent sample sites in MODEL TRACE order (the order find_valid_initial_params walks them)
    with their sizes.  basic: bpl/dixon_coles.py:46-78 (`attack` / `defence` are reparametrised:
    the traced latent sites are `*_decentered`); extended: bpl/extended_dixon_coles.py:112-235."""
    T = n_teams
    if model == 0:
        return [("home_advantage", 1), ("mean_defence", 1), ("std_attack", 1), ("std_defence", 1),
                ("attack_decentered", T), ("defence_decentered", T), ("corr_coef_raw", 1)]
    out = [("mean_home_advantage", 1), ("std_home_advantage", 1), ("mean_defence", 1),
           ("std_attack", 1), ("std_defence", 1)]
    if k:
        out += [("attack_coefficients", k), ("defence_coefficients", k)]
    out += [("u", 1), ("standardised_attack", T), ("standardised_defence", T),
            ("home_advantage_decentered", T), ("corr_coef_raw", 1)]
    return out
