#!/usr/bin/env python3
"""A/B of the rate-sweep kernel variants in ONE process (interleaved rounds), plus a
bit-exactness cross-check of their row sums.  GPU box only."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "cet-driven-simulation-for-3d-printing-am-kmc-approach_amd"))
import cetkmc  # noqa: E402
from cetkmc import synthetic  # noqa: E402

L = int(sys.argv[1]) if len(sys.argv) > 1 else 256
variants = [int(v) for v in (sys.argv[2].split(",") if len(sys.argv) > 2 else "1,3,4".split(","))]
rounds = int(sys.argv[3]) if len(sys.argv) > 3 else 3
e = cetkmc.Engine(L, impurity_c=0.2)
fill = float(os.environ.get("FILL", "0.25"))
st, th, ph, T, df = synthetic.planes(L, 0, L, seed=42, fill_frac=fill)
print("fill_frac", fill, flush=True)
e.upload_planes(0, L, st, th, ph, T, df)
evolve = int(os.environ.get("EVOLVE", "0"))
if evolve:
    rs = np.random.RandomState(1)
    q = synthetic.laser_planes(L, 0, evolve)
    r = e.run_steps(0, evolve, 3e-3, rs.random_sample(evolve), rs.random_sample(evolve), rs.random_sample(2 * evolve + 2),
                    rng_mode=1, seed=42, thermal_mode=int(os.environ.get("THERMAL", "2")), q_planes=q)
    print("evolved", r["done"], "steps, status", r["status"], flush=True)
ref = None
for v in variants:
    e.set_option("sweep_variant", v)
    info = e.rate_sweep()
    rs, rc = e.row_sums()
    if ref is None:
        ref = (info, rs, rc)
    else:
        ok = info == ref[0] and np.array_equal(rs, ref[1]) and np.array_equal(rc, ref[2])
        print(f"variant {v} vs {variants[0]}: bit-identical={ok}  info={info}", flush=True)
print("info", ref[0], flush=True)
res = {v: [] for v in variants}
for r in range(rounds):
    for v in variants:
        e.set_option("sweep_variant", v)
        n = 5 if v == 0 else 30
        e.time_sweeps(2)
        res[v].append(e.time_sweeps(n) / n)
for v in variants:
    ms = np.array(res[v])
    print(f"variant {v}: sweep+reduce median {np.median(ms)*1e3:.1f} us  min {ms.min()*1e3:.1f} us  ({L}^3)", flush=True)
