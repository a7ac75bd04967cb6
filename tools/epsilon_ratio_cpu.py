# Independent check of the posterior gaps behind test_epsilon (tools/epsilon_ratio.py is the GPU
# side): the INDEPENDENT numpy NUTS of oracle/nuts_oracle.py driving torch AUTOGRAD of the literal
# op-for-op transcription of the reference's model function (oracle/dc_torch_ref.py) -- no line of
# the product, no hand-derived adjoint.  CPU only, a few minutes.
#   python tools/epsilon_ratio_cpu.py [chains] [draws] > profiles/r03/epsilon_ratio_cpu.txt
import os
import sys

ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path[:0] = [ROOT + "/oracle", ROOT]
import numpy as np
import torch

import dc_oracle as O
import dc_torch_ref as TR
import nuts_oracle as NO

torch.set_num_threads(1)
chains = int(sys.argv[1]) if len(sys.argv) > 1 else 4
draws = int(sys.argv[2]) if len(sys.argv) > 2 else 4000
td = O.timed_dummy_data_recipe()
base = {k: td[k] for k in ("home_team", "away_team", "home_goals", "away_goals")}
sl = O.site_slices(O.MODEL_EXTENDED, 2)
shapes = NO.site_shapes(O.MODEL_EXTENDED, 2)


def gaps(eps):
    fx, _ = O.fixtures_from_training_data(base)
    fx.weights = O.time_weights(td["time_diff"], eps)
    pot = lambda z: TR.potential_and_grad(O.MODEL_EXTENDED, fx, z)[:2]
    att, dfn = [], []
    for c in range(chains):
        out = NO.run_chain(pot, NO.prng_key(100 + c), 500, draws, site_shapes=shapes)
        for z in out["draws"]:
            aux = O.potential_and_grad(O.MODEL_EXTENDED, fx, z)[2]
            att.append(aux["attack"])
            dfn.append(aux["defence"])
        a, d = np.array(att), np.array(dfn)
        print(f"# eps={eps} after chain {c}: gap_attack {abs(a[:, 1].mean() - a[:, 0].mean()):.4f} "
              f"gap_defence {abs(d[:, 1].mean() - d[:, 0].mean()):.4f} "
              f"(mean accept {out['accept_prob'][500:].mean():.3f}, divergences {int(out['diverging'].sum())})", flush=True)
    a, d = np.array(att), np.array(dfn)
    return abs(a[:, 1].mean() - a[:, 0].mean()), abs(d[:, 1].mean() - d[:, 0].mean())


a1, d1 = gaps(1.0)
a2, d2 = gaps(2.0)
print(f"# independent NUTS x autograd of the literal model, {chains} chains x {draws} draws: "
      f"gap_attack {a1:.4f} (eps=1) {a2:.4f} (eps=2) ratio {a2 / a1:.4f}; gap_defence {d1:.4f} {d2:.4f}")
