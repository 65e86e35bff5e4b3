#!/usr/bin/env python3
"""What ONE rank of an N-rank run spends in its rate sweep: the engine of rank r alone on the GPU (collective callbacks that
do nothing, so only its own slab's kernels run), e.g. L=512, 8 ranks -> 64 planes of 512^2 per rank (config 5's per-GPU
share).  The all-gathered block sums are incomplete, nothing is selected: timing of launch_sweep only.  GPU box only.
Usage: python tools/slab_sweep_time.py [L] [nranks] [rank]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "cet-driven-simulation-for-3d-printing-am-kmc-approach_amd"))
import cetkmc  # noqa: E402
from cetkmc import synthetic  # noqa: E402

L = int(sys.argv[1]) if len(sys.argv) > 1 else 512
nranks = int(sys.argv[2]) if len(sys.argv) > 2 else 8
rank = int(sys.argv[3]) if len(sys.argv) > 3 else 3
e = cetkmc.Engine(L, impurity_c=0.2, rank=rank, nranks=nranks, host_comm=(lambda *a: 0, lambda *a: 0))
a0, a1 = max(0, e.i0 - 2), min(L, e.i1 + 2)
st, th, ph, T, df = synthetic.planes(L, a0, a1, seed=42)
e.upload_planes(a0, a1, st, th, ph, T, df)
e.set_prev_state(None)
e.time_sweeps(3)
ts = [e.time_sweeps(20) / 20 for _ in range(5)]
vox = (e.i1 - e.i0) * L * L
print(f"L={L} rank {rank}/{nranks}: planes {e.i0}..{e.i1} ({vox/1e6:.1f} M voxels): sweep + reduce + host-relay no-op "
      f"{np.median(ts)*1e3:.1f} us -> {9.0 * vox / (np.median(ts) * 1e-3) / 1e12:.2f} TB/s algorithmic (incl. reduce and relay overhead)")
