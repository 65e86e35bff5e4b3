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


@pytest.mark.parametrize("null_events", [False, True])
@pytest.mark.parametrize("L,box,fill,n_slabs,df", [(16, 8, 0.2, 1, 0.05), (20, 10, 0.1, 2, 0.0), (32, 16, 0.3, 1, 0.1),
                                                  (24, 8, 0.02, 3, 0.02), (24, 12, 0.5, 1, 0.05)])
def test_supersteps_vs_oracle(oracle_mod, L, box, fill, n_slabs, df, null_events):
    e, lat = _pair(oracle_mod, L, 100 + L, fill, 0.2, n_slabs)
    step = 5
    for n in (19, 26):           # two batches: the second starts mid-cadence / mid-octant
        rg = e.run_supersteps(step, n, box, df, seed=4242, thermal_mode=1, want_events=True, null_events=null_events)
        ro = lat.run_supersteps(step, n, box, df, 4242, thermal_mode=1, null_events=null_events)
        assert rg["done"] == ro["done"] == n and rg["status"] == ro["status"] == 0
        for f in ("type", "pos", "target", "atom"):
            assert np.array_equal(rg["events"][f], ro["events"][f]), f
        live = ro["events"]["type"] != -1           # executed picks and null events (type -2) both carry their rate
        assert relerr(rg["events"]["rate"][live], ro["events"]["rate"][live]).max() <= RATE_RTOL
        assert np.array_equal(rg["n_exec"], ro["n_exec"])
        assert relerr(rg["totals"], ro["totals"]).max() <= RATE_RTOL
        # time advance (kmc_simulation.py:331-332 per executed event): host libm on both sides, totals within RATE_RTOL
        assert relerr(rg["dt_event"], ro["dt_event"]).max() <= RATE_RTOL
        if null_events:
            assert (ro["events"]["type"] == -2).sum() > 0
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
    for box in (6, 9, 16, 48):
        with pytest.raises(RuntimeError):
            e.run_supersteps(0, 1, box, 0.0, seed=1)
    e.close()


@pytest.mark.parametrize("L,fill,df,n_slabs,thermal_mode", [(8, 0.2, 0.1, 1, 1), (20, 0.05, 0.0, 2, 1), (33, 0.4, 0.05, 1, 2),
                                                            (64, 0.1, 0.02, 1, 1)])
def test_single_box_on_the_device_is_mode_a(oracle_mod, L, fill, df, n_slabs, thermal_mode):
    """Device twin of tests/test_oracle_mode_b.py::test_single_box_is_mode_a: cetkmc_run_supersteps(box == L) -- one domain,
    no octants -- equals cetkmc_run_steps (Mode A, kmc_simulation.py:246-332) fed the same counter uniforms, step by step:
    events, totals, every field, nucleation count.  Both also equal the CPU comparator's single-box run (up to L = 33)."""
    import cetkmc
    from cetkmc import synthetic
    n, seed, step0 = 45, 77, 3
    state, theta, phi, T, defects = random_lattice(L, 5, fill=fill)
    q = synthetic.laser_planes(L, step0, n) if thermal_mode == 2 else None

    def engine():
        e = cetkmc.Engine(L, impurity_c=0.2, n_slabs=n_slabs)
        e.upload(state, theta, phi, T, defects)
        e.set_prev_state(None)
        return e
    b = engine()
    rb = b.run_supersteps(step0, n, L, df, seed, thermal_mode=thermal_mode, q_planes=q, want_events=True)
    assert rb["done"] == n and np.all(rb["n_exec"] == 1) and rb["domains"] == 1
    ev = rb["events"][:, 0]
    u_pick = np.array([oracle_mod.counter_uniform(seed, step0 + s, oracle_mod.KEY_PICK) for s in range(n)])
    u_def = np.array([oracle_mod.counter_uniform(seed, step0 + s, oracle_mod.KEY_DEFECT) for s in range(n)])
    u_np = []
    for s in range(n):
        if ev["type"][s] in (0, 2):
            u_np += [oracle_mod.counter_uniform(seed, step0 + s, oracle_mod.KEY_THETA),
                     oracle_mod.counter_uniform(seed, step0 + s, oracle_mod.KEY_PHI)]
    a = engine()
    ra = a.run_steps(step0, n, df, u_pick, u_def, np.array(u_np + [0.0, 0.0]), rng_mode=1, seed=seed, thermal_mode=thermal_mode,
                     q_planes=q)
    assert ra["done"] == n and ra["np_used"] == len(u_np)
    assert ra["events"].tobytes() == np.ascontiguousarray(ev).tobytes()          # the whole 64-byte records
    assert np.array_equal(ra["totals"], rb["totals"])
    da, db = a.download(), b.download()
    for f in ("state", "theta", "phi", "T"):
        assert np.array_equal(da[f], db[f]), f
    assert a.nucleation_count() == b.nucleation_count()
    # the all-counter stream through the Mode A entry point itself (rng_mode 2, no host streams)
    c = engine()
    rc = c.run_steps(step0, n, df, None, None, None, rng_mode=2, seed=seed, thermal_mode=thermal_mode, q_planes=q)
    assert rc["events"].tobytes() == ra["events"].tobytes() and np.array_equal(rc["totals"], ra["totals"])
    if L <= 33:
        lat = oracle_mod.Lattice(state, theta, phi, T, defects, impurity_c=0.2)
        ro = lat.run_supersteps(step0, n, L, df, seed, thermal_mode=thermal_mode, q_planes=q)
        for f in ("type", "pos", "target", "atom"):
            assert np.array_equal(ev[f], ro["events"][:, 0][f]), f
        assert relerr(rb["totals"], ro["totals"]).max() <= RATE_RTOL and np.array_equal(db["state"], lat.state)
        assert relerr(rb["dt_event"], ro["dt_event"]).max() <= RATE_RTOL
    for x in (a, b, c):
        x.close()


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


@pytest.mark.parametrize("null_events", [False, True])
def test_supersteps_vs_oracle_128(oracle_mod, null_events):
    """128^3, box 8 (4096 boxes), 10 super-steps = all eight octants + a temperature update: per-box events (incl. the null
    events of the acceptance test against R_max over 4096 window totals), executed counts and every field equal the CPU
    comparator's."""
    import os

    import cetkmc
    from cetkmc import synthetic
    L, box, n = 128, 8, 10
    st, th, ph, T, df = synthetic.planes(L, 0, L, seed=8)
    q = synthetic.laser_planes(L, 15, n)
    e = cetkmc.Engine(L, impurity_c=0.2, n_slabs=2)
    e.upload_planes(0, L, st, th, ph, T, df)
    e.set_prev_state(None)
    rg = e.run_supersteps(15, n, box, 3e-3, seed=77, thermal_mode=2, q_planes=q, want_events=True, null_events=null_events)
    d = e.download()
    e.close()
    oracle_mod.set_threads(min(16, os.cpu_count() or 1))
    try:
        lat = oracle_mod.Lattice(st.astype(np.int64), th, ph, T, df.astype(np.int64), impurity_c=0.2)
        ro = lat.run_supersteps(15, n, box, 3e-3, 77, thermal_mode=2, q_planes=q, null_events=null_events)
    finally:
        oracle_mod.set_threads(1)
    assert rg["done"] == ro["done"] == n
    for f in ("type", "pos", "target", "atom"):
        assert np.array_equal(rg["events"][f], ro["events"][f]), f
    assert np.array_equal(rg["n_exec"], ro["n_exec"]) and (null_events or rg["n_exec"].min() > 2000)
    assert (ro["events"]["type"] == -2).any() == null_events
    assert relerr(rg["totals"], ro["totals"]).max() <= RATE_RTOL
    assert np.array_equal(d["state"], lat.state) and np.array_equal(d["T"], lat.T)
    assert np.array_equal(d["theta"], lat.theta) and np.array_equal(d["phi"], lat.phi)
