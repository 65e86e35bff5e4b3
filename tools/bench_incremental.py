#!/usr/bin/env python3
"""Full-sweep stepping vs exact incremental stepping (run_steps incremental=1) on the bench workload:
same inputs, per-step totals / chosen events compared bit for bit, device time per step of both.  GPU box only."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "cet-driven-simulation-for-3d-printing-am-kmc-approach_amd"))
import cetkmc  # noqa: E402
from cetkmc import synthetic  # noqa: E402

L = int(sys.argv[1]) if len(sys.argv) > 1 else 256
n = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
thermal = int(os.environ.get("THERMAL", "2"))
st, th, ph, T, df = synthetic.planes(L, 0, L, seed=42)
rs = np.random.RandomState(1)
u_pick, u_def, u_np = rs.random_sample(n), rs.random_sample(n), rs.random_sample(2 * n + 2)
q = synthetic.laser_planes(L, 0, n)
out = {}
for inc in (0, 1, 0, 1):
    e = cetkmc.Engine(L, impurity_c=0.2)
    e.upload_planes(0, L, st, th, ph, T, df)
    e.set_prev_state(None)
    r = e.run_steps(0, n, 3e-3, u_pick, u_def, u_np, rng_mode=1, seed=42, thermal_mode=thermal, q_planes=q, incremental=inc)
    sig = (r["done"], r["totals"].tobytes(), r["events"].tobytes(), e.download_planes(0, L)["state"].tobytes())
    out.setdefault(inc, []).append((r["wall_ms"] / max(r["done"], 1), sig, r["full_sweeps"]))
    print(f"incremental={inc}: {r['done']} steps, {r['wall_ms'] / max(r['done'], 1) * 1e3:.1f} us/step, "
          f"full sweeps {r['full_sweeps']}", flush=True)
    e.close()
print("bit-identical:", all(a[1] == out[0][0][1] for v in out.values() for a in v), flush=True)
print(f"speed-up {min(a[0] for a in out[0]) / min(a[0] for a in out[1]):.2f}x", flush=True)
