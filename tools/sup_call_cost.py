"""per-call cost of cetkmc_run_supersteps: one super-step per call vs batches (GPU box)"""
import sys, time
import numpy as np
sys.path.insert(0, "cet-driven-simulation-for-3d-printing-am-kmc-approach_amd")
import cetkmc
from cetkmc import synthetic
for L in (64, 256):
    st, th, ph, T, df = synthetic.planes(L, 0, L, seed=5)
    e = cetkmc.Engine(L, impurity_c=0.2)
    e.upload_planes(0, L, st, th, ph, T, df)
    e.set_prev_state(None)
    e.run_supersteps(0, 8, 8, 0.0, seed=1, thermal_mode=0, null_events=True)
    g = 8
    for nb in (1, 20):
        e.sync(); t0 = time.perf_counter(); calls = 40 if nb == 1 else 4
        for _ in range(calls):
            r = e.run_supersteps(g, nb, 8, 0.0, seed=1, thermal_mode=0, null_events=True); g += nb
        e.sync(); dt = time.perf_counter() - t0
        print(f"L={L} {nb:2d} super-step(s) per call: {dt / (calls * nb) * 1e6:8.1f} us per super-step (host wall)", flush=True)
    e.close()
