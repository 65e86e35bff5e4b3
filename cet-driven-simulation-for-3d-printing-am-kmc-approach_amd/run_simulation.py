"""Entry script with the reference's name (run_simulation.py:1-30).

Default behaviour is the reference's: build a 10^3 toy lattice and print the metrics dict
(no KMC).  ``python run_simulation.py --kmc [L]`` additionally runs the KMC plumbing case of
BASELINE config 1 -- one reference-initialised lattice (default 32^3), one thermal update,
one full rate sweep + selection + update on the GPU -- and prints the sweep summary.
"""
import sys

import numpy as np

from metrics import compute_metrics

state = np.zeros((10, 10, 10), dtype=int)
state[2:6, 2:6, 2:6] = 1
state[6:9, 6:9, 6:9] = 2
theta = np.zeros_like(state, dtype=float)
phi = np.zeros_like(state, dtype=float)

W_mask = np.zeros_like(state, dtype=bool)
Re_mask = np.zeros_like(state, dtype=bool)
C_mask = np.zeros_like(state, dtype=bool)
W_mask[3, 3, 3] = True
Re_mask[7, 7, 7] = True
C_mask[5, 5, 5] = True
grain_ids = state.copy()

m = compute_metrics(state, theta, phi, defects=None, W_mask=W_mask, Re_mask=Re_mask, C_mask=C_mask,
                    grain_ids=grain_ids, rng_seed=42)
print("Keys:", m.keys())
print("Values:", m)


def kmc_plumbing(L=32):
    from kmc_simulation import run_kmc
    out = run_kmc(L=L, n_steps=1, temp=2800, defect_fraction=0.0, n_seeds=5, impurity_c=0.0,
                  output_prefix=f"plumbing_L{L}")
    print("occupied voxels after 1 step:", int((out[0] != 0).sum()))
    return out


if __name__ == "__main__" and "--kmc" in sys.argv:
    args = [a for a in sys.argv[1:] if a.isdigit()]
    kmc_plumbing(int(args[0]) if args else 32)
