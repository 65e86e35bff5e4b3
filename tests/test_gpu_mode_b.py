"""Mode B on the device vs its CPU comparator (oracle orc_run_supersteps): per-box events, totals and all
lattice fields after every batch.  Mode B is not in the reference; the comparator's single-box case is pinned
to Mode A in tests/test_oracle_mode_b.py."""
import numpy as np
import pytest

from helpers import random_lattice, relerr

pytestmark = pytest.mark.gpu
RATE_RTOL = 1e-11


def _pair(oracle_mod, L, seed, fill, c, n_slabs=1):
    import cetkmc
    state, theta, phi, T, defects = random_lattice(L, seed, fill=fill)
    e = cetkmc.Engine(L, impurity_c=c, n_slabs=n_slabs)
    e.upload(state, theta, phi, T, defects)
    lat = oracle_mod.Lattice(state, theta, phi, T, defects, impurity_c=c)
    return e, lat


@pytest.mark.parametrize("L,box,fill,n_slabs,df", [(16, 8, 0.2, 1, 0.05), (20, 10, 0.1, 2, 0.0), (32, 16, 0.3, 1, 0.1),
                                                  (24, 8, 0.02, 3, 0.02), (24, 12, 0.5, 1, 0.05)])
def test_supersteps_vs_oracle(oracle_mod, L, box, fill, n_slabs, df):
    e, lat = _pair(oracle_mod, L, 100 + L, fill, 0.2, n_slabs)
    step = 5
    for n in (19, 26):           # two batches: the second starts mid-cadence / mid-octant
        rg = e.run_supersteps(step, n, box, df, seed=4242, thermal_mode=1, want_events=True)
        ro = lat.run_supersteps(step, n, box, df, 4242, thermal_mode=1)
        assert rg["done"] == ro["done"] == n and rg["status"] == ro["status"] == 0
        for f in ("type", "pos", "target", "atom"):
            assert np.array_equal(rg["events"][f], ro["events"][f]), f
        live = ro["events"]["type"] >= 0
        assert relerr(rg["events"]["rate"][live], ro["events"]["rate"][live]).max() <= RATE_RTOL
        assert np.array_equal(rg["n_exec"], ro["n_exec"])
        assert relerr(rg["totals"], ro["totals"]).max() <= RATE_RTOL
        d = e.download()
        assert np.array_equal(d["state"], lat.state)
        assert np.array_equal(d["theta"], lat.theta) and np.array_equal(d["phi"], lat.phi)
        assert np.array_equal(d["T"], lat.T)
        assert rg["nucleation_count"] == lat.nuc_count
        step += n
    assert rg["n_exec"].max() > 1
    # the engine is in a consistent state for Mode A afterwards (interface list, class array)
    e.set_option("sweep_variant", 0)
    a = (e.rate_sweep(),) + tuple(x.tobytes() for x in e.row_sums())
    e.set_option("sweep_variant", 1)
    b = (e.rate_sweep(),) + tuple(x.tobytes() for x in e.row_sums())
    assert a == b
    e.close()


def test_supersteps_laser_mode_and_slab_invariance(oracle_mod):
    import cetkmc
    from cetkmc import synthetic
    L, box, n = 32, 8, 24
    st, th, ph, T, df = synthetic.planes(L, 0, L, seed=3)
    q = synthetic.laser_planes(L, 0, n)
    outs, evs = [], []
    for ns in (1, 4):
        e = cetkmc.Engine(L, impurity_c=0.2, n_slabs=ns)
        e.upload_planes(0, L, st, th, ph, T, df)
        e.set_prev_state(None)
        r = e.run_supersteps(0, n, box, 3e-3, seed=7, thermal_mode=2, q_planes=q, want_events=True)
        assert r["done"] == n
        d = e.download()
        evs.append(r["events"])
        outs.append((r["events"].tobytes(), r["totals"].tobytes(), r["n_exec"].tobytes()) + tuple(d[k].tobytes() for k in sorted(d)))
        e.close()
    assert outs[0] == outs[1]
    lat = oracle_mod.Lattice(st.astype(np.int64), th, ph, T, df.astype(np.int64), impurity_c=0.2)
    ro = lat.run_supersteps(0, n, box, 3e-3, 7, thermal_mode=2, q_planes=q)
    ev = evs[0]
    for f in ("type", "pos", "target", "atom"):
        assert np.array_equal(ev[f], ro["events"][f]), f
    assert np.array_equal(np.frombuffer(outs[0][2], np.int64), ro["n_exec"])


def test_supersteps_argument_errors():
    import cetkmc
    e = cetkmc.Engine(24, impurity_c=0.1)
    st, th, ph, T, df = random_lattice(24, 1)
    e.upload(st, th, ph, T, df)
    for box in (6, 9, 16, 24):
        with pytest.raises(RuntimeError):
            e.run_supersteps(0, 1, box, 0.0, seed=1)
    e.close()


def test_supersteps_full_size_properties(oracle_mod):
    """256^3, box 8 (32768 boxes): executed events = non-idle boxes, every event inside its window, written
    voxels pairwise distinct, species histogram moves by exactly the executed events; the first two super-steps'
    per-box events equal the CPU comparator's."""
    import os

    import cetkmc
    from cetkmc import synthetic
    L, box, n = 256, 8, 6
    st, th, ph, T, df = synthetic.planes(L, 0, L, seed=42)
    oracle_mod.set_threads(min(16, os.cpu_count() or 1))
    try:
        lat = oracle_mod.Lattice(st.astype(np.int64), th, ph, T, df.astype(np.int64), impurity_c=0.2)
        ro = lat.run_supersteps(0, 2, box, 3e-3, 42, thermal_mode=2, q_planes=synthetic.laser_planes(L, 0, 2))
    finally:
        oracle_mod.set_threads(1)
    del lat
    e = cetkmc.Engine(L, impurity_c=0.2)
    e.upload_planes(0, L, st, th, ph, T, df)
    e.set_prev_state(None)
    c0 = e.species_counts()
    r = e.run_supersteps(0, n, box, 3e-3, seed=42, thermal_mode=2, q_planes=synthetic.laser_planes(L, 0, n), want_events=True)
    assert r["done"] == n and r["domains"] == 32 ** 3
    for f in ("type", "pos", "target", "atom"):
        assert np.array_equal(r["events"][f][:2], ro["events"][f]), f
    nb, H = L // box, box // 2
    for s in range(n):
        ev = r["events"][s]
        d = np.nonzero(ev["type"] >= 0)[0]
        live = ev[d]
        assert len(live) == r["n_exec"][s] > 10000
        sh = np.array([(s >> 2) & 1, (s >> 1) & 1, s & 1]) * H
        org = np.stack([d // (nb * nb), (d // nb) % nb, d % nb], 1) * box + sh
        assert np.all(live["pos"] >= org) and np.all(live["pos"] < org + H)
        w = np.concatenate([live["pos"], live["target"][live["type"] == 1]]).astype(np.int64)
        key = (w[:, 0] * L + w[:, 1]) * L + w[:, 2]
        assert len(np.unique(key)) == len(key)
    c1 = e.species_counts()
    assert c1[:5].sum() == c0[:5].sum() == L ** 3
    filled = sum(int((r["events"][s]["type"][r["events"][s]["type"] >= 0] != 1).sum()) for s in range(n))
    assert c0[0] - c1[0] == filled          # every dep/nuc/att event fills one empty voxel; diffusion moves one
    e.close()


def test_supersteps_vs_oracle_128(oracle_mod):
    """128^3, box 8 (4096 boxes), 10 super-steps = all eight octants + a temperature update: per-box events,
    executed counts and every field equal the CPU comparator's."""
    import os

    import cetkmc
    from cetkmc import synthetic
    L, box, n = 128, 8, 10
    st, th, ph, T, df = synthetic.planes(L, 0, L, seed=8)
    q = synthetic.laser_planes(L, 15, n)
    e = cetkmc.Engine(L, impurity_c=0.2, n_slabs=2)
    e.upload_planes(0, L, st, th, ph, T, df)
    e.set_prev_state(None)
    rg = e.run_supersteps(15, n, box, 3e-3, seed=77, thermal_mode=2, q_planes=q, want_events=True)
    d = e.download()
    e.close()
    oracle_mod.set_threads(min(16, os.cpu_count() or 1))
    try:
        lat = oracle_mod.Lattice(st.astype(np.int64), th, ph, T, df.astype(np.int64), impurity_c=0.2)
        ro = lat.run_supersteps(15, n, box, 3e-3, 77, thermal_mode=2, q_planes=q)
    finally:
        oracle_mod.set_threads(1)
    assert rg["done"] == ro["done"] == n
    for f in ("type", "pos", "target", "atom"):
        assert np.array_equal(rg["events"][f], ro["events"][f]), f
    assert np.array_equal(rg["n_exec"], ro["n_exec"]) and rg["n_exec"].min() > 2000
    assert relerr(rg["totals"], ro["totals"]).max() <= RATE_RTOL
    assert np.array_equal(d["state"], lat.state) and np.array_equal(d["T"], lat.T)
    assert np.array_equal(d["theta"], lat.theta) and np.array_equal(d["phi"], lat.phi)
