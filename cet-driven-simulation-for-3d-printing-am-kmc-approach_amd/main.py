"""Carbon-level sweep driver (equivalent of the reference ``main.py:17-108``).

For each carbon level: save the initial lattice, run the GPU KMC (kmc_simulation.run_kmc), read
back ``metrics.csv`` and classify the final microstructure.  Plotting is optional (matplotlib is
only imported when ``--plots`` is given); the reference's visualization / graphs modules are
outside the accelerated path and are not part of this package.

    python main.py [--L 30] [--steps 20000] [--levels 0.0 0.1 0.2] [--plots] [--mode B --box 8]
"""
import argparse
import os
import time

import numpy as np
import pandas as pd

from constants import DEFECT_PROB, LATTICE_SIZE, N_SEEDS, N_STEPS, T_SUB
from kmc_simulation import run_kmc
from lattice_init import initialize_lattice, save_lattice
from metrics import detect_CET_transition


def main(L=LATTICE_SIZE, n_steps=N_STEPS, carbon_levels=(0.0, 0.1, 0.2), plots=False, **run_kw):
    print("Starting KMC simulation for microstructure control...")
    t_start = time.time()
    summary = {"carbon_levels": [], "grain_sizes": [], "defect_densities": [], "aspect_ratios": []}
    for c in carbon_levels:
        prefix = f"impurity_c_{int(c * 100)}"
        print(f"\nRunning simulation with {c * 100:.1f}% carbon")
        out_dir = f"outputs/{prefix}"
        os.makedirs(f"{out_dir}/microstructures", exist_ok=True)
        os.makedirs("output_images", exist_ok=True)
        init = initialize_lattice(lattice_size=L, n_seeds=N_SEEDS, T_sub=T_SUB, random_seed=42, impurity_c=c)
        save_lattice(*init[:4], init[4], prefix=f"{out_dir}/init")
        t0 = time.time()
        state, atom_type, total_time, theta, phi = run_kmc(L=L, n_steps=n_steps, temp=T_SUB, defect_fraction=DEFECT_PROB,
                                                           n_seeds=N_SEEDS, impurity_c=c, output_prefix=prefix, **run_kw)
        t1 = time.time()
        csv_path = f"outputs/{prefix}/metrics.csv"
        status = "Undetected"
        if os.path.exists(csv_path):
            df = pd.read_csv(csv_path)
            if not df.empty:
                last = df.iloc[-1].to_dict()
                status = "Equiaxed" if detect_CET_transition(last) else "Columnar"
                summary["carbon_levels"].append(c)
                summary["grain_sizes"].append(last.get("AvgGrainSize", np.nan))
                summary["defect_densities"].append(last.get("DefectDensity", np.nan))
                summary["aspect_ratios"].append(last.get("AspectRatio", np.nan))
        else:
            print(f"Metrics file not found at {csv_path}")
        if plots:
            from lattice_init import visualize_initial_seeds
            visualize_initial_seeds(state, atom_type, title=f"Final State: {status} (C={c * 100:.0f}%)",
                                    filename=f"output_images/final_state_{prefix}.png")
        print(f"Completed {prefix} in {t1 - t0:.2f}s ({status})")
    print(f"\nTotal runtime: {time.time() - t_start:.2f} seconds")
    return summary


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--L", type=int, default=LATTICE_SIZE)
    ap.add_argument("--steps", type=int, default=N_STEPS)
    ap.add_argument("--levels", type=float, nargs="*", default=[0.0, 0.1, 0.2])
    ap.add_argument("--plots", action="store_true")
    ap.add_argument("--mode", choices=("A", "B"), default="A", help="A: exact loop (one event per sweep); B: super-steps")
    ap.add_argument("--box", type=int, default=8)
    a = ap.parse_args()
    main(a.L, a.steps, tuple(a.levels), a.plots, **(dict(mode="B", box=a.box) if a.mode == "B" else {}))
