#!/usr/bin/env python3
"""End-to-end wall time of the drop-in entry point at the reference's own default call (kmc_simulation.run_kmc: L=30,
n_steps=20000 -- /root/reference/kmc_simulation.py:203) and at a larger lattice: what a user who switches the import sees,
metrics rows, CSV and the reference's random streams included.  GPU box only; writes under the current directory's outputs/."""
import contextlib
import io
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "cet-driven-simulation-for-3d-printing-am-kmc-approach_amd"))
import kmc_simulation  # noqa: E402

out = []
for kw in (dict(L=30, n_steps=20000), dict(L=30, n_steps=20000, defect_fraction=0.003, impurity_c=0.1),
           dict(L=64, n_steps=20000, impurity_c=0.1), dict(L=128, n_steps=20000, impurity_c=0.1)):
    buf = io.StringIO()
    t0 = time.perf_counter()
    with contextlib.redirect_stdout(buf):
        state, atom_type, total_time, theta, phi = kmc_simulation.run_kmc(output_prefix="timing", **kw)
    dt = time.perf_counter() - t0
    info = dict(kmc_simulation.last_run_info)
    r = dict(kw, wall_s=round(dt, 2), executed=info.get("executed_events"), us_per_step=round(dt / max(info.get("executed_events") or 1, 1) * 1e6, 1),
             filled=int((state != 0).sum()), simulated_time=total_time, metric_rows=buf.getvalue().count("\nStep ") + buf.getvalue().startswith("Step "))
    print(json.dumps(r), flush=True)
    out.append(r)
print(json.dumps({"runs": out}))
