# Evidence behind tests/test_gpu_fit.py::test_epsilon.  The reference asserts, on ONE numpyro
# chain (seed 42, 500 + 1000), delta_attack(epsilon=2) > 1.5 * delta_attack(epsilon=1)
# (/root/reference/tests/test_extended_dixon_coles.py:28-47).  This script fits the same model on
# the same recipe (tests/conftest.py:32-62, restated in oracle/dc_oracle.py) at the reference's
# 500 + 1000 for many seeds and reports the distribution of that ratio, plus one long run
# (8 chains x 5000 draws) for the exact posterior gaps.
#   python tools/epsilon_ratio.py [n_seeds] > profiles/r03/epsilon_ratio.txt
import os
import sys

ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path[:0] = [ROOT + "/bpl-next_amd", ROOT + "/oracle", ROOT]
import numpy as np

import dc_oracle as O
from bpl import ExtendedDixonColesMatchPredictor

n_seeds = int(sys.argv[1]) if len(sys.argv) > 1 else 48
td = O.timed_dummy_data_recipe()


def gaps(seed, eps, warm=500, samp=1000, chains=1):
    m = ExtendedDixonColesMatchPredictor().fit(td, random_state=seed, num_warmup=warm, num_samples=samp,
                                               epsilon=eps, mcmc_kwargs={"num_chains": chains} if chains > 1 else None)
    a, d = m.attack.mean(axis=0), m.defence.mean(axis=0)
    return abs(a[1] - a[0]), abs(d[1] - d[0])


print("# seed  gap_attack(eps=1)  gap_attack(eps=2)  ratio   gap_defence(eps=1)  gap_defence(eps=2)")
rows = []
for seed in [42] + [s for s in range(n_seeds) if s != 42][: n_seeds - 1]:
    a1, d1 = gaps(seed, 1.0)
    a2, d2 = gaps(seed, 2.0)
    rows.append((seed, a1, a2, a2 / a1, d1, d2))
    print(f"{seed:5d}  {a1:.4f}  {a2:.4f}  {a2 / a1:.4f}  {d1:.4f}  {d2:.4f}", flush=True)
r = np.array(rows)
ratio = r[:, 3]
print(f"# {len(rows)} seeds at 500+1000: ratio mean {ratio.mean():.4f} sd {ratio.std(ddof=1):.4f} "
      f"min {ratio.min():.4f} max {ratio.max():.4f}; > 1.5 for {int((ratio > 1.5).sum())} of {len(rows)} seeds "
      f"({[int(s) for s in r[ratio > 1.5, 0]]})")
print(f"# gap_attack eps=1: mean {r[:, 1].mean():.4f} sd {r[:, 1].std(ddof=1):.4f};  "
      f"eps=2: mean {r[:, 2].mean():.4f} sd {r[:, 2].std(ddof=1):.4f}")
A1, D1 = gaps(1000, 1.0, 1000, 5000, 8)
A2, D2 = gaps(1000, 2.0, 1000, 5000, 8)
print(f"# long run, 8 chains x 5000 draws: gap_attack {A1:.4f} (eps=1) {A2:.4f} (eps=2) ratio {A2 / A1:.4f}; "
      f"gap_defence {D1:.4f} {D2:.4f}")
