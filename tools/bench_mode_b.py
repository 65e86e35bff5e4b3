#!/usr/bin/env python3
"""Mode B (super-steps over boxes) on the bench workload: executed events/s.  GPU box only."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "cet-driven-simulation-for-3d-printing-am-kmc-approach_amd"))
import cetkmc  # noqa: E402
from cetkmc import synthetic  # noqa: E402

L = int(sys.argv[1]) if len(sys.argv) > 1 else 256
box = int(sys.argv[2]) if len(sys.argv) > 2 else 8
n = int(sys.argv[3]) if len(sys.argv) > 3 else 40
st, th, ph, T, df = synthetic.planes(L, 0, L, seed=42)
e = cetkmc.Engine(L, impurity_c=0.2)
e.upload_planes(0, L, st, th, ph, T, df)
e.set_prev_state(None)
step = 0
for rep in range(int(os.environ.get("REPS", "4"))):
    q = synthetic.laser_planes(L, step, n)
    r = e.run_supersteps(step, n, box, 3e-3, seed=42, thermal_mode=2, q_planes=q)
    ex = int(r["n_exec"].sum())
    print(f"steps {step}..{step + r['done']}: {r['wall_ms'] / max(r['done'], 1) * 1e3:.1f} us/super-step, "
          f"{ex / max(r['done'], 1):.0f} events/super-step, {ex / (r['wall_ms'] * 1e-3):.3e} executed events/s, "
          f"empty voxels left {e.species_counts()[0]}", flush=True)
    step += r["done"]
    if r["status"]:
        break
