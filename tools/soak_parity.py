#!/usr/bin/env python3
"""Long GPU <-> oracle parity soak (one-off evidence, not part of the suite; GPU box only):
Mode A (exact loop, incremental on and off) and Mode B (null events) on a reference-initialised lattice until more than half
of the voxels are filled; every batch compares chosen events, event counts, totals (<= 1e-11) and at the end all fields.

    python tools/soak_parity.py [--L 48] [--steps 60000] [--supersteps 400] [--threads 16]
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "cet-driven-simulation-for-3d-printing-am-kmc-approach_amd"))
sys.path.insert(0, ROOT)
import cetkmc  # noqa: E402
import defects as host_defects  # noqa: E402
import lattice_init  # noqa: E402
from oracle import oracle  # noqa: E402  (checker only)


def relerr(a, b):
    a, b = np.asarray(a, float), np.asarray(b, float)
    out = np.abs(a - b) / np.maximum(np.abs(b), 1e-300)
    out[a == b] = 0.0
    return out


def lattice(L, seed, c=0.15):
    np.random.seed(seed)
    st, th, ph, T, at = lattice_init.initialize_lattice(lattice_size=L, n_seeds=20, T_sub=2800, impurity_c=c)
    mask, _ = host_defects.introduce_defects(st, at, T, apply_to_state=False)
    return st, th, ph, T, mask, c


def soak_mode_a(L, n_steps, incremental, batch=2000):
    st, th, ph, T, mask, c = lattice(L, 1)
    e = cetkmc.Engine(L, impurity_c=c)
    e.upload(st, th, ph, T, mask)
    lat = oracle.Lattice(st, th, ph, T, mask, impurity_c=c)
    rs = np.random.RandomState(7)
    step, worst, margin = 0, 0.0, 1.0
    while step < n_steps:
        n = min(batch, n_steps - step)
        u_pick, u_def, u_np = rs.random_sample(n), rs.random_sample(n), rs.random_sample(2 * n + 2)
        rg = e.run_steps(step, n, 3e-3, u_pick, u_def, u_np, rng_mode=1, seed=5, thermal_mode=1, incremental=incremental)
        ro = lat.run_steps(step, n, 3e-3, u_pick, u_def, u_np, rng_mode=1, seed=5, thermal_mode=1)
        assert rg["done"] == ro["done"] and rg["status"] == ro["status"], (step, rg["done"], ro["done"])
        for f in ("type", "pos", "target", "atom"):
            assert np.array_equal(rg["events"][f], ro["events"][f]), (step, f)
        assert np.array_equal(rg["n_events"], ro["n_events"]) and rg["np_used"] == ro["np_used"]
        worst = max(worst, float(relerr(rg["totals"], ro["totals"]).max()))
        margin = min(margin, rg["min_margin"])
        step += rg["done"]
        if rg["status"]:
            break
    d = e.download()
    assert np.array_equal(d["state"], lat.state) and np.array_equal(d["theta"], lat.theta) and np.array_equal(d["T"], lat.T)
    fill = float((lat.state != 0).mean())
    e.close()
    return dict(mode="A", incremental=incremental, L=L, steps=step, fill=fill, worst_total_relerr=worst, min_margin=margin)


def soak_mode_b(L, n_super, box=8, batch=50):
    st, th, ph, T, mask, c = lattice(L, 2)
    e = cetkmc.Engine(L, impurity_c=c)
    e.upload(st, th, ph, T, mask)
    lat = oracle.Lattice(st, th, ph, T, mask, impurity_c=c)
    g, worst, executed = 0, 0.0, 0
    while g < n_super:
        n = min(batch, n_super - g)
        rg = e.run_supersteps(g, n, box, 3e-3, seed=9, thermal_mode=1, want_events=True, null_events=True)
        ro = lat.run_supersteps(g, n, box, 3e-3, 9, thermal_mode=1, null_events=True)
        assert rg["done"] == ro["done"] and rg["status"] == ro["status"]
        for f in ("type", "pos", "target", "atom"):
            assert np.array_equal(rg["events"][f], ro["events"][f]), (g, f)
        assert np.array_equal(rg["n_exec"], ro["n_exec"])
        worst = max(worst, float(relerr(rg["totals"], ro["totals"]).max()), float(relerr(rg["dt_event"], ro["dt_event"]).max()))
        executed += int(rg["n_exec"].sum())
        g += rg["done"]
        if rg["status"]:
            break
    d = e.download()
    assert np.array_equal(d["state"], lat.state) and np.array_equal(d["theta"], lat.theta) and np.array_equal(d["T"], lat.T)
    fill = float((lat.state != 0).mean())
    e.close()
    return dict(mode="B", null_events=True, L=L, box=box, supersteps=g, executed=executed, fill=fill, worst_relerr=worst)


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--L", type=int, default=48)
    ap.add_argument("--steps", type=int, default=60000)
    ap.add_argument("--supersteps", type=int, default=400)
    ap.add_argument("--threads", type=int, default=16)
    a = ap.parse_args()
    oracle.set_threads(a.threads)
    out = []
    for fn, args in ((soak_mode_a, (a.L, a.steps, True)), (soak_mode_a, (a.L, a.steps // 4, False)), (soak_mode_b, (a.L, a.supersteps))):
        t0 = time.time()
        r = fn(*args)
        r["seconds"] = round(time.time() - t0, 1)
        out.append(r)
        print(json.dumps(r), flush=True)
    print(json.dumps(dict(ok=True, runs=out)))
