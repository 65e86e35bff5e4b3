#!/usr/bin/env python3
"""In-sweep block reduction (sc1 hand-off between workgroups of one launch) against the separate k_plane_reduce launch:
two engines on the same lattice and streams, every step's total / event / count compared bit for bit.  GPU box only.
Usage: stress_reduce.py [L] [steps]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "cet-driven-simulation-for-3d-printing-am-kmc-approach_amd"))
import cetkmc  # noqa: E402
from cetkmc import synthetic  # noqa: E402

L = int(sys.argv[1]) if len(sys.argv) > 1 else 256
total_steps = int(sys.argv[2]) if len(sys.argv) > 2 else 20000
st, th, ph, T, df = synthetic.planes(L, 0, L, seed=42)
engs = []
for fused in (1, 0):
    e = cetkmc.Engine(L, impurity_c=0.2)
    e.set_option("reduce_in_sweep", fused)
    e.upload_planes(0, L, st, th, ph, T, df)
    e.set_prev_state(None)
    engs.append(e)
rs = np.random.RandomState(7)
step, pos, bad = 0, 0, 0
n = 500
u_np = rs.random_sample(2 * total_steps + 4 * n)
while step < total_steps:
    u_pick, u_def = rs.random_sample(n), rs.random_sample(n)
    q = synthetic.laser_planes(L, step, n)
    out = []
    for e in engs:
        r = e.run_steps(step, n, 3e-3, u_pick, u_def, u_np[pos:pos + 2 * n + 2], rng_mode=1, seed=42, thermal_mode=2, q_planes=q)
        assert r["done"] == n, r
        out.append(r)
    same = (out[0]["totals"].tobytes() == out[1]["totals"].tobytes() and out[0]["events"].tobytes() == out[1]["events"].tobytes()
            and out[0]["n_events"].tobytes() == out[1]["n_events"].tobytes())
    if not same:
        bad += 1
        d = np.nonzero(out[0]["totals"] != out[1]["totals"])[0]
        print(f"MISMATCH in steps {step}..{step + n}: first differing step {step + (d[0] if len(d) else -1)}", flush=True)
    pos += out[0]["np_used"]
    step += n
    if step % 5000 == 0:
        print(f"{step} steps, {bad} mismatching batches, {out[0]['wall_ms'] / n * 1e3:.1f} vs {out[1]['wall_ms'] / n * 1e3:.1f} us/step", flush=True)
print("RESULT", "ok" if bad == 0 else f"{bad} bad batches")
sys.exit(1 if bad else 0)
