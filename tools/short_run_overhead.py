#!/usr/bin/env python3
"""Host-side cost of ONE short batch (the driver's `bench.py --steps 20`): wall time around Engine.run_steps against the
device time between the batch's first and last hipEvent, with the inputs handed over in the call and staged ahead
(cetkmc_stage_inputs).  GPU box only."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "cet-driven-simulation-for-3d-printing-am-kmc-approach_amd"))
import cetkmc  # noqa: E402
from cetkmc import synthetic  # noqa: E402

L = int(sys.argv[1]) if len(sys.argv) > 1 else 256
n = int(sys.argv[2]) if len(sys.argv) > 2 else 20
st, th, ph, T, df = synthetic.planes(L, 0, L, seed=42)
e = cetkmc.Engine(L, impurity_c=0.2)
e.upload_planes(0, L, st, th, ph, T, df)
e.set_prev_state(None)
rs = np.random.RandomState(1)
step, pos = 0, 0
u_np = rs.random_sample(200000)
e.set_option("reserve_batch", n)
for staged, prof in ((False, 1), (True, 1), (True, 0), (True, 3), (True, 1), (True, 0)):
    walls, devs = [], []
    for rep in range(12):
        u_pick, u_def = rs.random_sample(n), rs.random_sample(n)
        q = synthetic.laser_planes(L, step, n)
        args = (step, n, 3e-3, u_pick, u_def, u_np[pos:pos + 2 * n + 2])
        kw = dict(rng_mode=1, seed=42, thermal_mode=2, q_planes=q, profile=prof)
        if staged:
            e.stage_inputs(*args, **kw)
        e.sync()
        t0 = time.perf_counter()
        r = e.run_steps(*args, staged=staged, **kw)
        e.sync()
        walls.append(1e3 * (time.perf_counter() - t0))
        devs.append(r["wall_ms"])
        assert r["done"] == n
        step += n
        pos += r["np_used"]
    w, d = np.median(walls[2:]), np.median(devs[2:])
    print(f"staged={staged} profile={prof}: wall {w:.3f} ms, device {d:.3f} ms, host-side {w - d:.3f} ms per {n}-step batch -> {n / w * 1e3:.0f} steps/s", flush=True)
