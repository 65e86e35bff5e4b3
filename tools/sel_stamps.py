#!/usr/bin/env python3
"""Where the fused selection + application kernel spends its time: thread 0's 100 MHz wall-clock stamps at the phase
boundaries (alternative build: bash tools/ab_build.sh -DCETKMC_SEL_STAMPS; run with CETKMC_LIB=.../libcetkmc_hip_alt.so).
GPU box only."""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "cet-driven-simulation-for-3d-printing-am-kmc-approach_amd"))
import cetkmc  # noqa: E402
from cetkmc import _lib, synthetic  # noqa: E402

L = int(sys.argv[1]) if len(sys.argv) > 1 else 256
inc = int(sys.argv[2]) if len(sys.argv) > 2 else 0
e = cetkmc.Engine(L, impurity_c=0.2)
st, th, ph, T, df = synthetic.planes(L, 0, L, seed=42)
e.upload_planes(0, L, st, th, ph, T, df)
e.set_prev_state(None)
lib = _lib.load()
names = ["start", "blocks loaded", "block heap", "block descent", "rows loaded", "row heap+descent", "voxels loaded",
         "voxel heap", "descent+slot+record", "apply start", "writes done", "touch done",
         "  touch: neighbourhood words", "  touch: flags / append issued", "  touch: evaluated + stored"]
acc = np.zeros(15)
rs = np.random.RandomState(1)
n_samples = 0
step = 0
for rep in range(60):
    n = 7                       # the stamps of the LAST step of each batch are read back
    q = synthetic.laser_planes(L, step, n)
    r = e.run_steps(step, n, 3e-3, rs.random_sample(n), rs.random_sample(n), rs.random_sample(2 * n + 2), rng_mode=1, seed=42,
                    thermal_mode=2, q_planes=q, incremental=bool(inc))
    step += r["done"]
    out = (C.c_longlong * 16)()
    assert lib.cetkmc_debug_sel_stamps(out) == 0
    v = np.array(out[:15], dtype=np.float64)
    if rep >= 5:
        acc += (v - v[0]) / 100.0      # us
        n_samples += 1
acc /= n_samples
for a, b, nm in zip(acc[:12], np.diff(np.concatenate([[0.0], acc[:12]])), names[:12]):
    print(f"{nm:24s} t = {a:7.2f} us   (+{b:5.2f})")
for a, nm in zip(acc[12:], names[12:]):
    print(f"{nm:34s} t = {a:7.2f} us")
