"""G-V sweep driver (BASELINE config 5: "full CET G-V sweep").

The reference only sweeps the carbon level (main.py:23); its thermal gradient G and growth velocity R are
constants derived in kmc_simulation.py:236-239 (G from T_SUB and the lattice height, R from NU_DEP).  This
driver varies both through the arguments the engine already takes: the substrate temperature ``temp`` of
run_kmc (initial T ramp, lattice_init.py:31) for G, and the deposition attempt frequency ``nu_dep`` for V, and
collects the CET classification of the last metrics row of every run into ``outputs/gv_sweep/gv_map.csv``.

    python gv_sweep.py [--L 30] [--steps 2000] [--temps 2800 3100 3400] [--nu-dep 2e12 2e13 2e14] [--carbon 0.2]
                       [--mode B --box 8]
"""
import argparse
import os

import pandas as pd

from constants import ATOMIC_SPACING_W, DEFECT_PROB, N_SEEDS, T_MELT, VOXEL_SIZE
from kmc_simulation import run_kmc


def gv_sweep(L=30, n_steps=2000, temps=(2800.0, 3100.0, 3400.0), nu_deps=(2e12, 2e13, 2e14), carbon=0.2,
             defect_fraction=DEFECT_PROB, n_seeds=N_SEEDS, out_dir="outputs/gv_sweep", **run_kw):
    """``run_kw`` goes to run_kmc unchanged -- e.g. ``mode="B", box=8`` runs every point of the map through the super-step
    engine (same metrics.csv columns; n_steps stays the number of executed events)."""
    rows = []
    for T_sub in temps:
        for nu_dep in nu_deps:
            prefix = f"gv_sweep/T{int(T_sub)}_V{nu_dep:.0e}_c_{int(carbon * 100)}"
            run_kmc(L=L, n_steps=n_steps, temp=T_sub, defect_fraction=defect_fraction, n_seeds=n_seeds,
                    impurity_c=carbon, output_prefix=prefix, nu_dep=nu_dep, **run_kw)
            last = pd.read_csv(f"outputs/{prefix}/metrics.csv").iloc[-1]
            G = (T_MELT - T_sub) / (L * VOXEL_SIZE)                 # gradient of the initial ramp of this run
            V = nu_dep * ATOMIC_SPACING_W
            rows.append({"T_sub": T_sub, "nu_dep": nu_dep, "G_K_per_m": G, "V_m_per_s": V, "G_over_V": G / V,
                         "AspectRatio": last["AspectRatio"], "EquiaxedFraction": last["EquiaxedFraction"],
                         "GrainCount": last["GrainCount"], "NucleationCount": last["NucleationCount"],
                         "CET_Class": last["CET_Class"], "CET_Detected": last["CET_Detected"]})
    os.makedirs(out_dir, exist_ok=True)
    df = pd.DataFrame(rows)
    df.to_csv(os.path.join(out_dir, "gv_map.csv"), index=False)
    return df


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--L", type=int, default=30)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--temps", type=float, nargs="*", default=[2800.0, 3100.0, 3400.0])
    ap.add_argument("--nu-dep", type=float, nargs="*", default=[2e12, 2e13, 2e14])
    ap.add_argument("--carbon", type=float, default=0.2)
    ap.add_argument("--mode", choices=("A", "B"), default="A", help="A: exact loop (one event per sweep); B: super-steps")
    ap.add_argument("--box", type=int, default=8)
    a = ap.parse_args()
    kw = dict(mode="B", box=a.box) if a.mode == "B" else {}
    print(gv_sweep(a.L, a.steps, tuple(a.temps), tuple(a.nu_dep), a.carbon, **kw).to_string(index=False))
