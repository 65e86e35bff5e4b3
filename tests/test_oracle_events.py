"""Oracle (oracle/cet_oracle.c) vs the reference's get_event_rates output (fixture F1).

Bar: event SET and ORDER exact (type, position, target, species), rates <= 1e-12
relative (the spec tolerance is 1e-6; +,-,*,/ are bit-identical and libm's exp/sin/cos/
acos differ from NumPy's by <= a few ulp), sequential total likewise.
"""
import glob
import os

import numpy as np
import pytest

from helpers import GOLDEN, load, relerr

EVENT_CASES = sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN, "events_*.npz")))


def _u_dep(z, n_dep):
    if "u_dep" in z.files:
        return z["u_dep"]
    rs = np.random.RandomState(int(z["np_seed"]))
    return rs.random_sample(n_dep)


@pytest.mark.parametrize("name", EVENT_CASES)
def test_enumerate_matches_reference(oracle_mod, name):
    z = load(name)
    lat = oracle_mod.Lattice(z["state"], z["theta"], z["phi"], z["T"], z["defects"], impurity_c=float(z["impurity_c"]))
    n_dep_ref = int(np.sum(z["etype"] == 0))
    ev, n_dep = lat.enumerate(_u_dep(z, n_dep_ref))
    assert n_dep == n_dep_ref
    assert len(ev) == len(z["rate"])
    assert np.array_equal(ev["type"], z["etype"])
    assert np.array_equal(ev["pos"], z["pos"])
    assert np.array_equal(ev["target"], z["target"])
    assert np.array_equal(ev["atom"], z["atom"])
    if len(ev):
        assert relerr(ev["rate"], z["rate"]).max() <= 1e-12
    # sequential total (kmc_simulation.py:259)
    _, tot = lat.select_sequential(ev, 0.5)
    assert abs(tot - float(z["seq_total"])) <= 1e-12 * max(abs(float(z["seq_total"])), 1e-300)


@pytest.mark.parametrize("name", EVENT_CASES)
def test_canonical_tree_consistent_with_list(oracle_mod, name):
    """The canonical tree (what the GPU implements) must describe the same event list:
    counts equal, tree total ~ sequential total, and tree selection == linear-scan
    selection for a spread of uniforms."""
    z = load(name)
    lat = oracle_mod.Lattice(z["state"], z["theta"], z["phi"], z["T"], z["defects"], impurity_c=float(z["impurity_c"]))
    ev, n_dep = lat.enumerate(None)
    sw = lat.sweep()
    assert sw["n_events"] == len(ev) and sw["n_dep"] == n_dep
    assert int(sw["rowcnt"].sum()) == len(ev)
    if len(ev) == 0:
        assert lat.select_tree(sw["blocksum"], sw["blockcnt"], sw["rowsum"], sw["rowcnt"], 0.0) is None
        return
    _, seq_total = lat.select_sequential(ev, 0.5)
    if np.isfinite(seq_total) and seq_total > 0:
        assert abs(sw["total"] - seq_total) <= 1e-12 * seq_total
    rs = np.random.RandomState(123)
    us = np.concatenate([rs.random_sample(200), [0.0, 1e-300, 0.999999999999, 1.0 - 2.0 ** -53]])
    cum = np.cumsum(ev["rate"])
    for u in us:
        idx, tot = lat.select_sequential(ev, u)
        e = lat.select_tree(sw["blocksum"], sw["blockcnt"], sw["rowsum"], sw["rowcnt"], u * sw["total"])
        r = u * tot
        # skip draws that land within rounding distance of an event boundary
        margin = np.min(np.abs(cum - r)) / tot if tot > 0 else 1.0
        if margin < 1e-9 and u not in (0.0,):
            continue
        ref = ev[idx]
        assert (e.type, tuple(e.pos), tuple(e.target)) == (int(ref["type"]), tuple(ref["pos"]), tuple(ref["target"])), (u, margin)
        assert e.rate == ref["rate"]
        if e.type == 0:
            assert e.dep_rank == ref["dep_rank"]


def test_neighbors_and_misorientation(oracle_mod):
    import json
    meta = json.load(open(os.path.join(GOLDEN, "metrics_meta.json")))
    for key, want in meta["neighbors"].items():
        i, j, k, L = map(int, key.split(","))
        got = oracle_mod.neighbors(i, j, k, L)
        assert got.tolist() == want
    z = load("metrics")
    got = np.array([oracle_mod.misorientation(*row) for row in z["misor_in"]])
    assert np.allclose(got, z["misor_out"], rtol=0, atol=5e-16 * np.pi) or np.max(np.abs(got - z["misor_out"])) < 1e-7
    assert np.max(np.abs(got - z["misor_out"])) < 2e-8   # acos is ill-conditioned at |dot|~1


def test_oracle_is_clean_under_address_and_ub_sanitizers():
    """`make -C oracle asan`: cet_oracle.c + oracle/selftest_main.c under -fsanitize=address,undefined on the CPU (every
    exported entry point on a 16^3 lattice; SURVEY section 5 "race detection / sanitizers")."""
    import os
    import shutil
    import subprocess
    if not shutil.which("gcc"):
        pytest.skip("no gcc")
    here = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle")
    r = subprocess.run(["make", "-s", "-C", here, "asan"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "oracle sanitizer self-test ok" in r.stdout
