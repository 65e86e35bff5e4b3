#!/usr/bin/env python3
"""Wall time of the drop-in run_kmc at the reference's own default problem (constants.py: L=30, 20000 steps;
main.py sweeps three carbon levels).  GPU box only.  Usage: python tools/time_run_kmc.py [L] [steps]"""
import contextlib
import io
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "cet-driven-simulation-for-3d-printing-am-kmc-approach_amd"))
import constants  # noqa: E402
import kmc_simulation  # noqa: E402

L = int(sys.argv[1]) if len(sys.argv) > 1 else constants.LATTICE_SIZE
steps = int(sys.argv[2]) if len(sys.argv) > 2 else constants.N_STEPS
os.chdir(tempfile.mkdtemp())
for c in (0.0, 0.1, 0.2):
    buf = io.StringIO()
    t0 = time.perf_counter()
    with contextlib.redirect_stdout(buf):
        out = kmc_simulation.run_kmc(L=L, n_steps=steps, temp=constants.T_SUB, defect_fraction=constants.DEFECT_PROB,
                                     n_seeds=constants.N_SEEDS, impurity_c=c, output_prefix=f"impurity_c_{int(c * 100)}")
    dt = time.perf_counter() - t0
    last = [ln for ln in buf.getvalue().splitlines() if ln.startswith("Step ")][-1]
    print(f"L={L} steps={steps} carbon={c}: {dt:.2f} s wall ({steps / dt:.0f} steps/s)  | {last}", flush=True)
