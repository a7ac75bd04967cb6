"""TEST INFRASTRUCTURE: writes tests/golden/nuts_*.npz from the independent NUTS restatement
(oracle/nuts_oracle.py) on (i) a diagonal Gaussian and (ii) the reference's dummy_data recipe
(tests/conftest.py:7-29) with the float64 C restatement of the Dixon-Coles potential.  Each file
holds the inputs (key, z0, step size, counts) and the oracle's trajectory: draws, tree sizes,
acceptance statistics, energies, step sizes.  Run: python oracle/make_nuts_golden.py"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path[:0] = [HERE, os.path.join(os.path.dirname(HERE), "tests")]
import cases  # noqa: E402
import dc_oracle as O  # noqa: E402
import dc_oracle_c as OC  # noqa: E402
import nuts_oracle as NO  # noqa: E402

OUT = os.path.join(os.path.dirname(HERE), "tests", "golden")
GAUSS_SD = np.array([1.0, 2.0, 0.5, 3.0, 1.5, 0.25, 4.0])


def gauss_pot(z):
    return 0.5 * float(np.sum(z * z / GAUSS_SD ** 2)), z / GAUSS_SD ** 2


def dc_pot(model, name):
    fx = cases.fixtures(name)
    cf = OC.CFixtures(model, fx)

    def pot(z):
        U, g, _ = OC.potential_and_grad(cf, z, 1)
        return U, g

    return pot, cf


def save(name, pot, key, warm, samp, z0=None, step=1.0, sites=None, max_depth=10, **extra):
    o = NO.run_chain(pot, key, warm, samp, z0=z0, site_shapes=sites, step_size=step, max_depth=max_depth)
    np.savez_compressed(
        os.path.join(OUT, name + ".npz"), key=np.array(key, dtype=np.uint32), num_warmup=warm,
        num_samples=samp, step_size0=step, max_tree_depth=max_depth, z0=o["z0"], z0_given=z0 is not None, draws=o["draws"],
        num_steps=o["num_steps"], accept_prob=o["accept_prob"], diverging=o["diverging"],
        step_size=o["step_size"], potential_energy=o["potential_energy"], depth=o["depth"],
        final_step_size=o["final_step_size"], inverse_mass_matrix=o["inverse_mass_matrix"], **extra)
    print(f"{name}: {o['num_steps'].sum()} leapfrogs, tree sizes {o['num_steps'][:12].tolist()} ...")


def main():
    os.makedirs(OUT, exist_ok=True)
    zg = np.array([0.3, -0.2, 0.1, 0.5, -0.4, 0.05, 1.0])
    # Gaussian: fixed step (pure tree builder), then the windowed adaptation
    save("nuts_gauss_fixed", gauss_pot, (0, 11), 0, 12, z0=zg, step=0.4, sd=GAUSS_SD)
    save("nuts_gauss_adapt", gauss_pot, (0, 7), 40, 10, z0=zg, step=1.0, sd=GAUSS_SD)
    # the reference's dummy_data, both in-scope models, fixed small step (deep trees)
    zb = np.random.RandomState(2).uniform(-0.2, 0.2, 45)
    pot, _ = dc_pot(O.MODEL_BASIC, "dummy")
    save("nuts_dummy_basic_fixed", pot, (0, 11), 0, 10, z0=zb, step=0.02)
    save("nuts_dummy_basic_adapt", pot, (0, 5), 30, 10, z0=zb, step=1.0)
    # the same with trees of at most 3 leapfrogs: rounding differences are amplified far less per
    # transition, so a float32-table potential (the GPU's) still tracks the whole warm-up --
    # one slow window (mass matrix update + dual-averaging restart at t = 19) and the final
    # averaging of the step size
    save("nuts_dummy_basic_adapt_shallow", pot, (0, 5), 22, 6, z0=zb, step=1.0, max_depth=2)
    # initial point: init_to_uniform(radius=2) + retry, then a short adapted run
    save("nuts_dummy_basic_init", pot, NO.prng_key(42), 20, 5, z0=None,
         sites=NO.site_shapes(0, 20))
    pot, cf = dc_pot(O.MODEL_EXTENDED, "dummy_cov")
    ze = np.random.RandomState(2).uniform(-0.2, 0.2, cf.D)
    save("nuts_dummy_ext_fixed", pot, (0, 11), 0, 8, z0=ze, step=0.02)
    save("nuts_dummy_ext_init", pot, NO.prng_key(7), 20, 5, z0=None,
         sites=NO.site_shapes(1, 20, 5))


if __name__ == "__main__":
    main()
