"""HIP engine (through the C ABI) vs the oracle and the reference fixtures.  All tests here
need a real MI355X: run with `pytest -m gpu` on the GPU box.

Bars: integer/index results (event set, order, targets, species, counts, lattice state)
bit-exact; rates within 1e-6 relative (the tests assert a much tighter 1e-11; device libm
(ocml) and NumPy/glibc differ by a few ulp); temperature fields bit-exact.
"""
import glob
import os

import numpy as np
import pytest

from helpers import GOLDEN, GpuBackend, OracleBackend, load, random_lattice, relerr, replay_run_kmc

pytestmark = pytest.mark.gpu

RATE_RTOL = 1e-11
EVENT_CASES = sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN, "events_*.npz")))
TRAJ = sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN, "traj_*.npz")))


def _engine(z_or_L, impurity_c=0.0, n_slabs=1):
    import cetkmc
    return cetkmc.Engine(int(z_or_L), impurity_c=impurity_c, n_slabs=n_slabs)


@pytest.mark.parametrize("name", EVENT_CASES)
def test_enumerate_vs_reference_fixture(name):
    z = load(name)
    L = int(z["L"])
    e = _engine(L, float(z["impurity_c"]))
    e.upload(z["state"], z["theta"], z["phi"], z["T"], z["defects"])
    ev, n = e.enumerate_events()
    assert n == len(z["rate"]) == len(ev)
    assert np.array_equal(ev["type"], z["etype"])
    assert np.array_equal(ev["pos"], z["pos"])
    assert np.array_equal(ev["target"], z["target"])
    nd = z["etype"] != 0
    assert np.array_equal(ev["atom"][nd], z["atom"][nd])
    dep = ~nd
    assert np.array_equal(ev["dep_rank"][dep], np.arange(int(dep.sum())))
    if n:
        assert relerr(ev["rate"], z["rate"]).max() <= RATE_RTOL


@pytest.mark.parametrize("name", EVENT_CASES)
def test_sweep_and_select_vs_oracle(oracle_mod, name):
    z = load(name)
    L = int(z["L"])
    c = float(z["impurity_c"])
    e = _engine(L, c)
    e.upload(z["state"], z["theta"], z["phi"], z["T"], z["defects"])
    lat = oracle_mod.Lattice(z["state"], z["theta"], z["phi"], z["T"], z["defects"], impurity_c=c)
    sw = lat.sweep()
    total, n_events, n_dep = e.rate_sweep()
    assert (n_events, n_dep) == (sw["n_events"], sw["n_dep"])
    rs, rc = e.row_sums()
    assert np.array_equal(rc, sw["rowcnt"])
    assert relerr(rs, sw["rowsum"]).max() <= RATE_RTOL
    if n_events == 0:
        return
    assert abs(total - sw["total"]) <= RATE_RTOL * abs(sw["total"])
    rng = np.random.RandomState(7)
    for u in list(rng.random_sample(60)) + [0.0, 1.0 - 2.0 ** -53]:
        want = lat.select_tree(sw["blocksum"], sw["blockcnt"], sw["rowsum"], sw["rowcnt"], u * sw["total"])
        got = e.select(u * total)
        assert (got.type, tuple(got.pos), tuple(got.target), got.dep_rank) == \
               (want.type, tuple(want.pos), tuple(want.target), want.dep_rank), u
        assert abs(got.rate - want.rate) <= RATE_RTOL * abs(want.rate)


@pytest.mark.parametrize("L,seed", [(20, 1), (33, 2), (64, 3), (100, 4), (130, 5)])
def test_sweep_vs_oracle_larger(oracle_mod, L, seed):
    state, theta, phi, T, defects = random_lattice(L, seed, fill=0.2 if L < 100 else 0.05)
    e = _engine(L, 0.15)
    e.upload(state, theta, phi, T, defects)
    lat = oracle_mod.Lattice(state, theta, phi, T, defects, impurity_c=0.15)
    sw = lat.sweep()
    total, n_events, n_dep = e.rate_sweep()
    rs, rc = e.row_sums()
    assert np.array_equal(rc, sw["rowcnt"])
    assert (n_events, n_dep) == (sw["n_events"], sw["n_dep"])
    assert relerr(rs, sw["rowsum"]).max() <= RATE_RTOL
    assert abs(total - sw["total"]) <= RATE_RTOL * abs(sw["total"])


@pytest.mark.parametrize("L,seed,fill", [(24, 1, 0.3), (40, 2, 0.05), (70, 3, 0.5), (8, 4, 0.9)])
def test_sweep_variants_bit_identical(L, seed, fill):
    """The LDS-census streaming kernel with the rate table (1, default), the census-free table sweep (3: class bytes + rate
    table only; 4: the same with one block per plane and the block sums folded in the sweep launch), the one that recomputes
    the nucleation rates per sweep (2) and the simple kernel (0) produce identical row
    sums, also after events were applied (interface-list / rate-table / count-byte maintenance by the apply kernel: the
    table sweep trusts the event counts in the class bytes)."""
    state, theta, phi, T, defects = random_lattice(L, seed, fill=fill)
    e = _engine(L, 0.2)
    e.upload(state, theta, phi, T, defects)
    rs = np.random.RandomState(seed)
    for rnd in range(3):
        out = []
        for v in (1, 2, 0, 3, 4):
            e.set_option("sweep_variant", v)
            info = e.rate_sweep()
            out.append((info,) + e.row_sums())
        for o in out[1:]:
            assert out[0][0] == o[0]
            assert np.array_equal(out[0][1], o[1]) and np.array_equal(out[0][2], o[2])
        e.set_option("sweep_variant", 1)
        n = 25
        res = e.run_steps(rnd * n, n, 0.1, rs.random_sample(n), rs.random_sample(n), rs.random_sample(n * (L * L + 2)),
                          rng_mode=0, thermal_mode=1)
        assert res["status"] in (0, 1) and (res["done"] == n or res["status"] == 1)   # dense lattices run out of events


def test_thermal_cet_fixtures():
    z = load("thermal")
    for L in (1, 2, 3, 7, 16):
        Tin = z[f"cet_rand_L{L}_in"]
        zero = np.zeros((L, L, L), np.int64)
        for dt, key in ((1e-6, "out"), (3e-7, "out_dt3e-7")):
            e = _engine(L)
            e.upload(zero, Tin * 0, Tin * 0, Tin, zero)
            e.thermal_cet(dt, scrub_nan=False)
            assert np.array_equal(e.download()["T"], z[f"cet_rand_L{L}_{key}"]), (L, dt)
    seq = z["cet_ramp_L12_seq"]
    e = _engine(12)
    zero = np.zeros((12, 12, 12), np.int64)
    e.upload(zero, seq[0] * 0, seq[0] * 0, seq[0], zero)
    for n in range(1, len(seq)):
        e.thermal_cet(1e-6, scrub_nan=True)
        assert np.array_equal(e.download()["T"], seq[n]), n
    Tn = z["cet_nan_L6_in"]
    e = _engine(6)
    zero = np.zeros((6, 6, 6), np.int64)
    e.upload(zero, zero.astype(float), zero.astype(float), Tn, zero)
    e.thermal_cet(1e-6, scrub_nan=True)
    assert np.array_equal(e.download()["T"], z["cet_nan_L6_out"])


@pytest.mark.parametrize("L", [8, 13, 16])
def test_thermal_laser(oracle_mod, L):
    z = load("thermal")
    for tag in ("dt1e-06", "dt1e-09"):
        key = f"laser_L{L}_{tag}"
        dt, i0, j0, P, rb, ab = z[key + "_par"]
        T, prev, cur = z[key + "_T"], z[key + "_prev"], z[key + "_cur"]
        q = oracle_mod.laser_source_plane(L, (i0, j0), P, rb, ab)
        lat = oracle_mod.Lattice(cur, T * 0, T * 0, T)
        want = lat.thermal_laser(dt, q, prev_state=prev)
        e = _engine(L)
        e.upload(cur.astype(np.int64), T * 0, T * 0, T, np.zeros_like(cur, dtype=np.int64))
        e.set_prev_state(prev.astype(np.int64))
        e.thermal_laser(dt, q, use_latent=True, scrub_nan=False)
        got = e.download()["T"]
        assert np.array_equal(got, want)                     # same source plane -> bit-exact
        assert np.allclose(got, z[key + "_out"], rtol=1e-12, atol=0)


@pytest.mark.parametrize("name", TRAJ)
def test_replay_reference_trajectory_stepwise(name):
    """run_kmc re-enacted with rate_sweep/select/apply through the C ABI, checked per step
    against the reference-generated fixture (state diffs, T snapshots, counts, totals)."""
    z = load(name)
    every = int(z["L"]) <= 16
    replay_run_kmc(z, lambda *a: GpuBackend(*a), check_every_step=every, rate_rtol=RATE_RTOL)


def _drive_batched(engine_like, z, rng_mode=0, seed=0, batch=64):
    """Batched protocol driver shared by the oracle and the device (same host RNG handling)."""
    import random

    import defects as host_defects
    import lattice_init as host_init

    L, n_steps = int(z["L"]), int(z["n_steps"])
    temp, df = float(z["temp"]), float(z["defect_fraction"])
    n_seeds, c = int(z["n_seeds"]), float(z["impurity_c"])
    np.random.seed(42)
    random.seed(42)
    state, theta, phi, T, atom = host_init.initialize_lattice(lattice_size=L, n_seeds=n_seeds, T_sub=temp, impurity_c=c)
    mask, _ = host_defects.introduce_defects(state, atom, T, apply_to_state=False)
    be = engine_like(state, theta, phi, T, mask, c)
    step, total_time = 0, 0.0
    log = []
    while step < n_steps:
        nxt_refresh = step + 1 if step % 200 == 0 else (step // 200 + 1) * 200 + 1
        n = min(n_steps, nxt_refresh, step + batch) - step
        per = 3 if df > 0 else 2
        draws = np.array([random.random() for _ in range(per * n)]).reshape(n, per)
        np_state = np.random.get_state()
        u_np = np.random.random(n * (L * L + 2))
        res = be.run_steps(step, n, df, draws[:, 0], draws[:, 1] if df > 0 else None, u_np, rng_mode=rng_mode,
                           seed=seed, thermal_mode=1)
        assert res["status"] == 0 and res["done"] == n, res
        np.random.set_state(np_state)
        np.random.random(res["np_used"])
        for s in range(n):
            total_time += max(-np.log(max(1e-12, draws[s, -1])) / res["totals"][s], 1e-12)
        log.append(res)
        step += n
        if (step - 1) % 200 == 0:
            st, Tnow = be.state_and_T()
            m, _ = host_defects.introduce_defects(st, st, Tnow, apply_to_state=False)
            be.set_defects(m)
    return be, total_time, log


class _GpuBatched:
    def __init__(self, state, theta, phi, T, defects, c, n_slabs=1):
        import cetkmc
        self.e = cetkmc.Engine(int(state.shape[0]), impurity_c=c, n_slabs=n_slabs)
        self.e.upload(state, theta, phi, T, defects)

    def run_steps(self, *a, **k):
        return self.e.run_steps(*a, **k)

    def state_and_T(self):
        d = self.e.download()
        return d["state"], d["T"]

    def set_defects(self, m):
        self.e.set_defects(m)

    def final(self):
        d = self.e.download()
        return d["state"], d["theta"], d["phi"], d["T"]


class _OracleBatched:
    def __init__(self, oracle_mod, state, theta, phi, T, defects, c):
        self.lat = oracle_mod.Lattice(state, theta, phi, T, defects, impurity_c=c)

    def run_steps(self, *a, **k):
        return self.lat.run_steps(*a, **k)

    def state_and_T(self):
        return self.lat.state.astype(np.int64), self.lat.T

    def set_defects(self, m):
        self.lat.defects = np.ascontiguousarray(m, dtype=np.int8)

    def final(self):
        return self.lat.state, self.lat.theta, self.lat.phi, self.lat.T


@pytest.mark.parametrize("name", ["traj_L8_n60", "traj_L12_n40", "traj_L16_n100", "traj_L10_n450", "traj_L7_n230_T3400"])
@pytest.mark.parametrize("n_slabs", [1, 2])
def test_batched_loop_reference_trajectory(name, n_slabs):
    """cetkmc_run_steps (no host round trips) reproduces the reference run, incl. RNG positions."""
    import random
    z = load(name)
    be, total_time, _ = _drive_batched(lambda *a: _GpuBatched(*a, n_slabs=n_slabs), z, rng_mode=0)
    st, th, ph, _ = be.final()
    assert np.array_equal(st, z["final_state"])
    assert np.array_equal(th, z["final_theta"]) and np.array_equal(ph, z["final_phi"])
    assert total_time == float(z["total_time"])
    assert np.array_equal(np.random.random(4), z["np_next"])
    assert np.array_equal(np.array([random.random() for _ in range(4)]), z["py_next"])


@pytest.mark.parametrize("rng_mode", [0, 1])
def test_batched_loop_vs_oracle_medium(oracle_mod, rng_mode):
    """Larger lattice, 120 steps, both RNG modes: per-step chosen events, totals, counts and the
    final lattice equal the oracle's batched loop."""
    z = dict(L=24, n_steps=120, temp=2800.0, defect_fraction=0.02, n_seeds=30, impurity_c=0.25)
    g, tg, lg = _drive_batched(lambda *a: _GpuBatched(*a), z, rng_mode=rng_mode, seed=99, batch=50)
    o, to, lo = _drive_batched(lambda *a: _OracleBatched(oracle_mod, *a), z, rng_mode=rng_mode, seed=99, batch=50)
    for rg, ro in zip(lg, lo):
        for f in ("type", "pos", "target", "atom", "dep_rank"):
            assert np.array_equal(rg["events"][f], ro["events"][f]), f
        assert np.array_equal(rg["events"]["theta"][np.isin(rg["events"]["type"], (0, 2))] >= 0, np.ones(int(np.isin(rg["events"]["type"], (0, 2)).sum()), bool))
        assert np.array_equal(rg["n_events"], ro["n_events"])
        assert relerr(rg["totals"], ro["totals"]).max() <= RATE_RTOL
        assert rg["np_used"] == ro["np_used"]
    for a, b in zip(g.final(), o.final()):
        assert np.array_equal(a, b)
    assert tg == to


@pytest.mark.parametrize("n_slabs", [2, 3, 5])
def test_slab_decomposition_bit_identical(n_slabs):
    """Axis-0 slabs (halo 2) give bit-identical totals, events and fields to the undivided lattice."""
    L = 20
    state, theta, phi, T, defects = random_lattice(L, 11, fill=0.25)
    rs = np.random.RandomState(5)
    n = 90
    u_pick, u_def, u_np = rs.random_sample(n), rs.random_sample(n), rs.random_sample(n * (L * L + 2))
    outs = []
    for ns in (1, n_slabs):
        e = _engine(L, 0.2, n_slabs=ns)
        e.upload(state, theta, phi, T, defects)
        res = e.run_steps(0, n, 0.05, u_pick, u_def, u_np, rng_mode=0, thermal_mode=1)
        outs.append((res, e.download(defects=True)))
    (r1, d1), (r2, d2) = outs
    assert r1["done"] == r2["done"] == n
    assert np.array_equal(r1["totals"], r2["totals"])            # bit-identical sums
    assert r1["events"].tobytes() == r2["events"].tobytes()
    for k in d1:
        assert np.array_equal(d1[k], d2[k]), k


def test_transfers_and_errors():
    import cetkmc
    L = 9
    state, theta, phi, T, defects = random_lattice(L, 3)
    e = _engine(L, 0.1, n_slabs=2)
    e.upload(state, theta, phi, T, defects)
    d = e.download(defects=True)
    for k, want in (("state", state), ("theta", theta), ("phi", phi), ("T", T), ("defects", defects)):
        assert np.array_equal(d[k], want), k
    p = e.download_planes(2, 7, state=True, theta=True, T=True, defects=True)
    assert np.array_equal(p["state"], state[2:7]) and np.array_equal(p["T"], T[2:7])
    e2 = _engine(L, 0.1)
    e2.upload_planes(0, L, state.astype(np.uint8), theta, phi, T, defects.astype(np.uint8))
    assert e2.rate_sweep() == e.rate_sweep()
    bad = state.copy()
    bad[0, 0, 0] = 300
    with pytest.raises(RuntimeError):
        e.upload(bad, theta, phi, T, defects)
    with pytest.raises(RuntimeError):
        cetkmc.Engine(2000)
    e3 = _engine(4)
    with pytest.raises(RuntimeError):
        e3.select(0.5)                     # no sweep yet
    z = np.zeros((4, 4, 4))
    occ = np.full((4, 4, 4), 4, dtype=np.int64)
    e3.upload(occ, z, z, z + 3000.0, occ * 0)
    assert e3.rate_sweep()[1] == 0         # defect-only lattice: no events
    with pytest.raises(RuntimeError):
        e3.select(0.0)
    res = e3.run_steps(0, 5, 0.0, np.full(5, 0.5), None, np.zeros(64), thermal_mode=0)
    assert res["status"] == 1 and res["done"] == 0     # kmc_simulation.py:260-262


@pytest.mark.parametrize("L,n_slabs,n", [(128, 4, 40), (256, 2, 12), (256, 4, 24)])
def test_full_size_determinism_and_slab_invariance(L, n_slabs, n):
    """BASELINE-size lattices (config-3 workload: pre-filled k<L/4, moving melt pool, latent heat,
    counter-mode species): run-to-run determinism and slab-count invariance of every per-step
    total, chosen event and of the final fields -- size-independent properties that need no oracle."""
    import cetkmc
    from cetkmc import synthetic
    st, th, ph, T, df = synthetic.planes(L, 0, L, seed=7)
    q = synthetic.laser_planes(L, 0, n)
    rs = np.random.RandomState(2)
    u_pick, u_def, u_np = rs.random_sample(n), rs.random_sample(n), rs.random_sample(2 * n + 2)
    outs = []
    for ns in (1, 1, n_slabs):
        e = cetkmc.Engine(L, impurity_c=0.2, n_slabs=ns)
        e.upload_planes(0, L, st, th, ph, T, df)
        e.set_prev_state(None)
        r = e.run_steps(0, n, 3e-3, u_pick, u_def, u_np, rng_mode=1, seed=11, thermal_mode=2, q_planes=q)
        assert r["done"] == n and r["status"] == 0
        d = e.download_planes(0, L, state=True, theta=True, T=True)
        outs.append((r["totals"].tobytes(), r["events"].tobytes(), r["n_events"].tobytes(),
                     d["state"].tobytes(), d["theta"].tobytes(), d["T"].tobytes()))
        e.close()
        # conservation: one executed event per step, every event changes at most 2 voxels
        changed = int(np.count_nonzero(d["state"] != st))
        assert 0 < changed <= 2 * n
    assert outs[0] == outs[1]          # determinism
    assert outs[0] == outs[2]          # slab-count invariance (bit-identical sums and picks)


def test_config2_constant_T_128(oracle_mod):
    """BASELINE config 2: 128^3, constant T=3000 K, pre-filled k<32: rate sweep + selection only,
    against the oracle on the full lattice (counts exact, row sums/total <= 1e-11, same picks)."""
    import cetkmc
    from cetkmc import synthetic
    L = 128
    st, th, ph, T, df = synthetic.planes(L, 0, L, seed=1, constant_T=3000.0)
    e = cetkmc.Engine(L, impurity_c=0.2)
    e.upload_planes(0, L, st, th, ph, T, df)
    lat = oracle_mod.Lattice(st, th, ph, T, df, impurity_c=0.2)
    sw = lat.sweep()
    total, n_events, n_dep = e.rate_sweep()
    rs, rc = e.row_sums()
    assert (n_events, n_dep) == (sw["n_events"], sw["n_dep"]) and np.array_equal(rc, sw["rowcnt"])
    assert relerr(rs, sw["rowsum"]).max() <= RATE_RTOL
    assert abs(total - sw["total"]) <= RATE_RTOL * sw["total"]
    for u in np.random.RandomState(4).random_sample(100):
        want = lat.select_tree(sw["blocksum"], sw["blockcnt"], sw["rowsum"], sw["rowcnt"], u * sw["total"])
        got = e.select(u * total)
        assert (got.type, tuple(got.pos), tuple(got.target)) == (want.type, tuple(want.pos), tuple(want.target)), u


@pytest.mark.parametrize("L,n_slabs", [(5, 1), (19, 2), (64, 1), (130, 3), (300, 1), (256, 1), (256, 4)])
def test_thermal_kernel_variants_identical(L, n_slabs):
    """Plane-marching LDS thermal kernels (1: k_thermal_tiles16 with 2 rows per thread where the tiles cover the lattice
    exactly -- L = 256 here -- else k_thermal_march; 2: k_thermal_march everywhere; 3: 16-row tiles with 4 rows per thread;
    4: the 8-row k_thermal_tiles; 5: 16 x 128 tiles; 1x: the same with 5 planes per block, a march that does not divide the
    slab) == one-thread-per-voxel kernel, bit for bit (cet, laser with and without the latent-heat term, NaN scrubbing)."""
    rs = np.random.RandomState(L)
    T = rs.uniform(2700.0, 4100.0, (L, L, L))
    T[rs.random_sample((L, L, L)) < 0.01] = np.nan
    state = (rs.random_sample((L, L, L)) < 0.3).astype(np.int64) * rs.randint(1, 5, (L, L, L))
    prev = state * (rs.random_sample((L, L, L)) < 0.7)
    q = rs.uniform(0, 1e15, (L, L))
    z = np.zeros((L, L, L))
    outs = []
    for v in (0, 1, 2, 3, 4, 5, 11, 13, 15):
        e = _engine(L, n_slabs=n_slabs)
        e.set_option("thermal_variant", v % 10)
        if v >= 11:
            e.set_option("thermal_planes_per_block16", 5)
        e.upload(state, z, z, T, state * 0)
        e.set_prev_state(prev)
        e.thermal_cet(1e-6, scrub_nan=True)
        a = e.download()["T"]
        e.thermal_laser(1e-6, q, use_latent=True, scrub_nan=False)
        b = e.download()["T"]
        e.thermal_cet(3e-7, scrub_nan=False)
        c = e.download()["T"]
        e.thermal_laser(2e-6, q, use_latent=False, scrub_nan=True)
        outs.append((a, b, c, e.download()["T"]))
        e.close()
    for o in outs[1:]:
        for x, y in zip(outs[0], o):
            assert np.array_equal(x, y)


@pytest.mark.parametrize("L", [264, 344])
def test_large_L_two_chunks_per_row(L):
    """L > 256: rows span two 256-voxel chunks in the streaming kernel (the N=2/4/8 bench sizes 320/408/512
    take this path); L = 344 additionally needs the 2048-leaf block heap of the selection (3L > 1024, as at
    408 and 512).  Streaming == simple kernel bit for bit, slab-count invariant, incl. after stepping."""
    import cetkmc
    from cetkmc import synthetic
    st, th, ph, T, df = synthetic.planes(L, 0, L, seed=3)
    rs = np.random.RandomState(9)
    # sprinkle isolated atoms / defects into the empty region so every event family occurs there too
    idx = rs.randint(0, L, (4000, 3))
    st[idx[:, 0], idx[:, 1], idx[:, 2]] = rs.randint(1, 5, 4000)
    n = 10
    q = synthetic.laser_planes(L, 0, n)
    u_pick, u_def, u_np = rs.random_sample(n), rs.random_sample(n), rs.random_sample(2 * n + 2)
    outs = []
    for ns, variant in ((1, 1), (1, 0), (3, 1), (2, 2), (2, 3), (1, 4), (3, 4)):
        e = cetkmc.Engine(L, impurity_c=0.2, n_slabs=ns)
        e.set_option("sweep_variant", variant)
        e.upload_planes(0, L, st, th, ph, T, df)
        e.set_prev_state(None)
        info = e.rate_sweep()
        rsum, rcnt = e.row_sums()
        r = e.run_steps(0, n, 3e-3, u_pick, u_def, u_np, rng_mode=1, seed=5, thermal_mode=2, q_planes=q)
        assert r["done"] == n
        outs.append((info, rsum.tobytes(), rcnt.tobytes(), r["totals"].tobytes(), r["events"].tobytes()))
        e.close()
    assert outs[0] == outs[1] and outs[0] == outs[2] and outs[0] == outs[3]


def test_interface_every_step_option_identical():
    """Default: the interface list is evaluated in full only after a temperature update, in between the apply
    kernel re-evaluates the <= 30 listed voxels an event touches.  interface_every_step=1 (the whole list before
    every full sweep) must not change a single bit."""
    L = 40
    state, theta, phi, T, defects = random_lattice(L, 31, fill=0.2)
    rs = np.random.RandomState(8)
    n = 70
    u_pick, u_def, u_np = rs.random_sample(n), rs.random_sample(n), rs.random_sample(n * (L * L + 2))
    outs = []
    for ov in (0, 1):
        e = _engine(L, 0.2)
        e.set_option("interface_every_step", ov)
        e.upload(state, theta, phi, T, defects)
        res = e.run_steps(0, n, 0.05, u_pick, u_def, u_np, rng_mode=0, thermal_mode=1)
        assert res["done"] == n
        info = e.rate_sweep()
        outs.append((res["totals"].tobytes(), res["events"].tobytes(), info, e.row_sums()[0].tobytes(), e.download()["theta"].tobytes()))
    assert outs[0] == outs[1]


@pytest.mark.parametrize("L", [36, 255, 264, 520])
def test_stream_kernel_row_shapes(L):
    """Rows of <= 256 voxels occupy half a wave (two rows per wave), longer rows one or two 512-voxel chunks of a
    full wave; ragged L leaves the last lane partly outside the lattice.  Both streaming variants equal the simple
    kernel bit for bit."""
    import cetkmc
    e = cetkmc.Engine(L, impurity_c=0.2)
    if L < 400:
        state, theta, phi, T, defects = random_lattice(L, 77, fill=0.15 if L < 100 else 0.02)
        e.upload(state, theta, phi, T, defects)
    else:               # narrow host arrays (u8 state): 520^3 in the reference's int64 would be 1.1 GB per field
        from cetkmc import synthetic
        st, th, ph, T, df = synthetic.planes(L, 0, L, seed=3)
        idx = np.random.RandomState(9).randint(0, L, (20000, 3))
        st[idx[:, 0], idx[:, 1], idx[:, 2]] = np.random.RandomState(10).randint(1, 5, 20000)
        e.upload_planes(0, L, st, th, ph, T, df)
        e.set_prev_state(None)
        del st, th, ph, T, df
    out = []
    for v in (1, 2, 0, 3, 4):
        e.set_option("sweep_variant", v)
        out.append((e.rate_sweep(),) + e.row_sums())
    for o in out[1:]:
        assert out[0][0] == o[0]
        assert np.array_equal(out[0][1], o[1]) and np.array_equal(out[0][2], o[2])
    e.close()


@pytest.mark.parametrize("L,fill,n_slabs,thermal", [(20, 0.25, 1, 1), (40, 0.2, 3, 1), (33, 0.6, 1, 0), (70, 0.1, 2, 1)])
def test_incremental_mode_bit_identical(L, fill, n_slabs, thermal):
    """incremental=1 (only the rows an event made stale are re-evaluated between temperature updates)
    is exact: totals, chosen events, counts, row sums and all fields equal the full-sweep run bit for bit."""
    state, theta, phi, T, defects = random_lattice(L, 41 + L, fill=fill)
    rs = np.random.RandomState(8)
    n = 130
    u_pick, u_def, u_np = rs.random_sample(n), rs.random_sample(n), rs.random_sample(n * (L * L + 2))
    outs = []
    for inc, ns in ((False, 1), (True, n_slabs)):
        e = _engine(L, 0.2, n_slabs=ns)
        e.upload(state, theta, phi, T, defects)
        res = e.run_steps(3, n, 0.05, u_pick, u_def, u_np, rng_mode=0, thermal_mode=thermal, incremental=inc)
        assert res["done"] == n
        n_full = n if not inc else 1 + (sum(1 for s in range(1, n) if (3 + s) % 20 == 0) if thermal else 0)
        assert res["full_sweeps"] == n_full
        d = e.download(defects=True)
        info = e.rate_sweep()
        outs.append((res["totals"].tobytes(), res["events"].tobytes(), res["n_events"].tobytes(), res["np_used"], info,
                     e.row_sums()[0].tobytes(), e.row_sums()[1].tobytes()) + tuple(d[k].tobytes() for k in sorted(d)))
        e.close()
    assert outs[0] == outs[1]


def test_incremental_mode_large_rows_and_laser():
    """Incremental stepping with two 256-voxel chunks per row (L > 256), the laser thermal mode and the
    counter-hash species draw."""
    import cetkmc
    from cetkmc import synthetic
    L = 264
    st, th, ph, T, df = synthetic.planes(L, 0, L, seed=3)
    rs = np.random.RandomState(9)
    idx = rs.randint(0, L, (4000, 3))
    st[idx[:, 0], idx[:, 1], idx[:, 2]] = rs.randint(1, 5, 4000)
    n = 45
    q = synthetic.laser_planes(L, 0, n)
    u_pick, u_def, u_np = rs.random_sample(n), rs.random_sample(n), rs.random_sample(2 * n + 2)
    outs = []
    for inc in (False, True):
        e = cetkmc.Engine(L, impurity_c=0.2)
        e.upload_planes(0, L, st, th, ph, T, df)
        e.set_prev_state(None)
        r = e.run_steps(0, n, 3e-3, u_pick, u_def, u_np, rng_mode=1, seed=5, thermal_mode=2, q_planes=q, incremental=inc)
        assert r["done"] == n
        outs.append((r["totals"].tobytes(), r["events"].tobytes(), e.rate_sweep(), e.row_sums()[0].tobytes()))
        e.close()
    assert outs[0] == outs[1]


def test_counters_and_phase_profile():
    """cetkmc_get_counters: work issued, bytes moved, and the per-phase device times of a profile=2 batch;
    profiling must not change results."""
    L = 48
    state, theta, phi, T, defects = random_lattice(L, 5, fill=0.2)
    rs = np.random.RandomState(2)
    n = 60
    u_pick, u_def, u_np = rs.random_sample(n), rs.random_sample(n), rs.random_sample(n * (L * L + 2))
    outs = []
    for prof, inc in ((False, False), (2, False), (2, True)):
        e = _engine(L, 0.2)
        e.upload(state, theta, phi, T, defects)
        c0 = e.counters(reset=True)
        assert c0["bytes_h2d"] == L ** 3 * (8 + 8 + 8 + 8 + 8)          # int64 state/defects + three f64 fields
        r = e.run_steps(0, n, 0.05, u_pick, u_def, u_np, rng_mode=0, thermal_mode=1, profile=prof, incremental=inc)
        c = e.counters()
        assert r["done"] == n and c["steps"] == n and c["thermal_updates"] == 3
        assert c["sweeps"] == (n if not inc else 3) and c["incremental_steps"] == (0 if not inc else n - 3)
        assert c["alg_bytes_sweep"] == c["sweeps"] * 9 * L ** 3 and c["alg_bytes_thermal"] == 3 * 16 * L ** 3
        # rate table + full interface-list evaluation only after the three temperature updates
        assert c["table_updates"] == 3 and c["alg_bytes_table"] == 3 * 16 * L ** 3 and c["interface_launches"] == 3
        if prof == 2:
            assert c["profiled_steps"] == n
            assert c["ms_sweep"] > 0 and c["ms_interface"] > 0 and c["ms_select_apply"] > 0 and c["ms_thermal"] > 0
            assert (c["ms_dirty_rows"] > 0) == inc
            parts = sum(c[k] for k in ("ms_thermal", "ms_interface", "ms_sweep", "ms_dirty_rows", "ms_reduce", "ms_select_apply"))
            assert 0.5 * r["wall_ms"] < parts <= 1.05 * r["wall_ms"]
        outs.append((r["totals"].tobytes(), r["events"].tobytes(), e.download()["theta"].tobytes()))
        e.close()
    assert outs[0] == outs[1] == outs[2]


def test_selection_with_2048_leaf_block_heap_vs_oracle(oracle_mod):
    """3L > 1024 (here L = 344; the 4- and 8-GPU bench sizes 408 and 512 as well): the selection kernel's block
    heap has 2048 leaves, 8 per thread.  Chosen events, totals and counts of a few steps equal the oracle's."""
    import cetkmc
    from cetkmc import synthetic
    L, n = 344, 3
    st, th, ph, T, df = synthetic.planes(L, 0, L, seed=11)
    rs = np.random.RandomState(4)
    u_pick, u_def, u_np = rs.random_sample(n), rs.random_sample(n), rs.random_sample(2 * n + 2)
    q = synthetic.laser_planes(L, 0, n)
    e = cetkmc.Engine(L, impurity_c=0.2)
    e.upload_planes(0, L, st, th, ph, T, df)
    e.set_prev_state(None)
    rg = e.run_steps(0, n, 3e-3, u_pick, u_def, u_np, rng_mode=1, seed=5, thermal_mode=2, q_planes=q)
    e.close()
    oracle_mod.set_threads(min(16, os.cpu_count() or 1))
    try:
        lat = oracle_mod.Lattice(st.astype(np.int64), th, ph, T, df.astype(np.int64), impurity_c=0.2)
        ro = lat.run_steps(0, n, 3e-3, u_pick, u_def, u_np, rng_mode=1, seed=5, thermal_mode=2, q_planes=q)
    finally:
        oracle_mod.set_threads(1)
    assert rg["done"] == ro["done"] == n
    for f in ("type", "pos", "target", "atom"):
        assert np.array_equal(rg["events"][f], ro["events"][f]), f
    assert np.array_equal(rg["n_events"], ro["n_events"])
    assert relerr(rg["totals"], ro["totals"]).max() <= RATE_RTOL


@pytest.mark.parametrize("incremental", [False, True])
def test_long_run_vs_oracle(oracle_mod, incremental):
    """3000 steps (15 defect-mask refreshes, 150 temperature updates, the interface list growing all the way) through
    the batched protocol: every chosen event, count and total, the stream positions and the final fields equal the
    oracle's."""
    z = dict(L=18, n_steps=3000, temp=2800.0, defect_fraction=0.01, n_seeds=12, impurity_c=0.15)

    class _Gpu(_GpuBatched):
        def run_steps(self, *a, **k):
            return self.e.run_steps(*a, incremental=incremental, **k)

    oracle_mod.set_threads(min(8, os.cpu_count() or 1))
    try:
        g, tg, lg = _drive_batched(lambda *a: _Gpu(*a), z, rng_mode=0, batch=190)
        o, to, lo = _drive_batched(lambda *a: _OracleBatched(oracle_mod, *a), z, rng_mode=0, batch=190)
    finally:
        oracle_mod.set_threads(1)
    assert len(lg) == len(lo)
    for rg, ro in zip(lg, lo):
        for f in ("type", "pos", "target", "atom", "dep_rank"):
            assert np.array_equal(rg["events"][f], ro["events"][f]), f
        assert np.array_equal(rg["n_events"], ro["n_events"]) and rg["np_used"] == ro["np_used"]
        assert relerr(rg["totals"], ro["totals"]).max() <= RATE_RTOL
    for a, b in zip(g.final(), o.final()):
        assert np.array_equal(a, b)
    assert tg == to


def test_incremental_256_vs_oracle(oracle_mod):
    """The bench workload itself (256^3, laser mode, counter species draw), 45 steps of exact incremental stepping:
    chosen events, counts and totals equal the oracle's full evaluation of every step."""
    import cetkmc
    from cetkmc import synthetic
    L, n = 256, 45
    st, th, ph, T, df = synthetic.planes(L, 0, L, seed=42)
    rs = np.random.RandomState(6)
    u_pick, u_def, u_np = rs.random_sample(n), rs.random_sample(n), rs.random_sample(2 * n + 2)
    q = synthetic.laser_planes(L, 0, n)
    e = cetkmc.Engine(L, impurity_c=0.2)
    e.upload_planes(0, L, st, th, ph, T, df)
    e.set_prev_state(None)
    rg = e.run_steps(0, n, 3e-3, u_pick, u_def, u_np, rng_mode=1, seed=42, thermal_mode=2, q_planes=q, incremental=True)
    e.close()
    assert rg["done"] == n and rg["full_sweeps"] == 3
    oracle_mod.set_threads(min(16, os.cpu_count() or 1))
    try:
        lat = oracle_mod.Lattice(st.astype(np.int64), th, ph, T, df.astype(np.int64), impurity_c=0.2)
        ro = lat.run_steps(0, n, 3e-3, u_pick, u_def, u_np, rng_mode=1, seed=42, thermal_mode=2, q_planes=q)
    finally:
        oracle_mod.set_threads(1)
    for f in ("type", "pos", "target", "atom"):
        assert np.array_equal(rg["events"][f], ro["events"][f]), f
    assert np.array_equal(rg["n_events"], ro["n_events"])
    assert relerr(rg["totals"], ro["totals"]).max() <= RATE_RTOL


def test_terminating_step_consumes_its_species_draws(oracle_mod):
    """kmc_event_rates.py:65 draws one NumPy uniform per finite-rate deposition candidate BEFORE run_kmc tests the total
    (kmc_simulation.py:259-262): a run that terminates with candidates left has consumed them.  run_kmc itself cannot
    get there (an empty site at T >= T_SUB always owns a large nucleation rate), so no reference fixture covers this
    step -- parity unpinned; device and oracle are compared with each other.  Here: I0 = 0 (no nucleation), cold
    lattice (every deposition rate underflows to 0.0, which is finite and therefore a candidate), total = 0."""
    import cetkmc
    L = 8
    state = np.zeros((L, L, L), np.int64)
    state[:, :, 0] = 4                                  # a defect layer: no events of its own
    zeros = np.zeros((L, L, L))
    T = np.full((L, L, L), 1000.0)
    params = cetkmc.default_params(0.2)
    params.I0 = 0.0
    e = cetkmc.Engine(L, impurity_c=0.2, params=params)
    e.upload(state, zeros, zeros, T, np.zeros((L, L, L), np.int64))
    lat = oracle_mod.Lattice(state, zeros, zeros, T, None, impurity_c=0.2)
    lat.params.I0 = 0.0
    n_top = L * (L - 1)                                 # empty sites of plane L-1
    assert e.rate_sweep() == (0.0, n_top, n_top)
    rs = np.random.RandomState(0)
    n = 3
    u_pick, u_np = rs.random_sample(n), rs.random_sample(n * (L * L + 2))
    rg = e.run_steps(0, n, 0.0, u_pick, None, u_np, rng_mode=0, thermal_mode=0)
    ro = lat.run_steps(0, n, 0.0, u_pick, None, u_np, rng_mode=0, thermal_mode=0)
    assert rg["done"] == ro["done"] == 0 and rg["status"] == ro["status"] == 1
    assert rg["np_used"] == ro["np_used"] == n_top
    # the stream is too short for the terminating step's draws: status 2 (refill), nothing consumed
    rg = e.run_steps(0, n, 0.0, u_pick, None, u_np[:n_top - 1], rng_mode=0, thermal_mode=0)
    ro = lat.run_steps(0, n, 0.0, u_pick, None, u_np[:n_top - 1], rng_mode=0, thermal_mode=0)
    assert rg["status"] == ro["status"] == 2 and rg["np_used"] == ro["np_used"] == 0
    # counter mode draws nothing from the stream
    rg = e.run_steps(0, n, 0.0, u_pick, None, u_np, rng_mode=1, seed=3, thermal_mode=0)
    assert rg["status"] == 1 and rg["np_used"] == 0
    e.close()


def test_config5_workload_at_size(oracle_mod):
    """BASELINE config 5's workload at its full size on ONE GPU: 512^3, eight in-process axis-0 slabs (the 8-GPU
    decomposition), impurity_c = 0.2, laser thermal mode, counter RNG.  (a) the first steps equal the oracle's (16
    threads): chosen events, counts, totals <= 1e-11; (b) the defect refresh without moving the lattice
    (defects.py:4-19 on the carbon sites) gives the mask the host path gives; (c) the same steps with ONE slab are
    bit-identical, and that engine clusters the lattice (utils.py:28-84) with every occupied voxel labelled."""
    import cetkmc
    import defects as defects_mod
    from cetkmc import synthetic
    from oracle import oracle
    L, n = 512, 3
    st, th, ph, T, df = synthetic.planes(L, 0, L, seed=42)
    rs = np.random.RandomState(4)
    u_pick, u_def, u_np = rs.random_sample(n), rs.random_sample(n), rs.random_sample(2 * n + 2)
    q = synthetic.laser_planes(L, 0, n)
    outs = []
    for ns in (8, 1):
        e = cetkmc.Engine(L, impurity_c=0.2, n_slabs=ns)
        e.upload_planes(0, L, st, th, ph, T, df)
        e.set_prev_state(None)
        info = e.rate_sweep()
        r = e.run_steps(0, n, 3e-3, u_pick, u_def, u_np, rng_mode=1, seed=42, thermal_mode=2, q_planes=q)
        assert r["done"] == n and r["status"] == 0
        np.random.seed(7)
        n_flag, _ = defects_mod.refresh_defects_device(e)
        outs.append((info, r["totals"].tobytes(), r["events"].tobytes(), r["n_events"].tobytes(), n_flag, e.rate_sweep()))
        if ns == 1:
            cl = e.clusters(labels=True)
            occ = e.download_planes(0, L, state=True)["state"] != 0
            assert np.array_equal(cl["labels"] > 0, occ) and cl["size"].sum() == occ.sum()
            del cl, occ
        if ns == 8:
            keep = (r, e.download_planes(0, L, defects=True)["defects"].copy())
        e.close()
    assert outs[0] == outs[1]
    rg, mask_dev = keep
    # oracle on the same steps
    oracle.set_threads(16)
    try:
        lat = oracle.Lattice(st, th, ph, T, df, impurity_c=0.2)
        ro = lat.run_steps(0, n, 3e-3, u_pick, u_def, u_np, rng_mode=1, seed=42, thermal_mode=2, q_planes=q)
    finally:
        oracle.set_threads(1)
    assert ro["done"] == n
    for f in ("type", "pos", "target", "atom"):
        assert np.array_equal(rg["events"][f], ro["events"][f]), f
    assert np.array_equal(rg["n_events"], ro["n_events"])
    assert relerr(rg["totals"], ro["totals"]).max() <= RATE_RTOL
    # host path of the defect refresh on the oracle's lattice (same stream position)
    np.random.seed(7)
    mask_host = defects_mod.track_defects(lat.state.astype(np.int64), lat.state.astype(np.int64), L, lat.T)
    assert np.array_equal(mask_dev != 0, mask_host != 0)


@pytest.mark.parametrize("n_slabs,thermal_mode", [(1, 2), (3, 2), (1, 1)])
def test_thermal_lookahead_option_identical(n_slabs, thermal_mode):
    """thermal_lookahead=1: the next temperature update of a batch and its rate table are computed ahead on a second
    stream without the latent-heat term; at the update k_thermal_fix recomputes the voxels the term concerns.  Must not
    change a bit (events, totals, T, every field), across batches that start off the 20-step cadence and through a Mode B
    batch.  (Batches that stop early -- status 1 / 2 -- and their continuation are covered by
    test_batch_that_runs_out_of_stream_continues_bit_identically and test_terminated_batch_leaves_a_consistent_engine.)"""
    import cetkmc
    from cetkmc import synthetic
    L, n = 48, 130
    st, th, ph, T, df = synthetic.planes(L, 0, L, seed=5)
    rs = np.random.RandomState(6)
    idx = rs.randint(0, L, (300, 3))
    st[idx[:, 0], idx[:, 1], idx[:, 2]] = rs.randint(1, 5, 300)
    u_pick, u_def, u_np = rs.random_sample(n), rs.random_sample(n), rs.random_sample(2 * n + 2)
    outs = []
    for la in (0, 1):
        e = cetkmc.Engine(L, impurity_c=0.2, n_slabs=n_slabs)
        e.set_option("thermal_lookahead", la)
        e.upload_planes(0, L, st, th, ph, T, df)
        e.set_prev_state(None)
        res, pos, step = [], 0, 3
        for nb in (47, 83):                        # two batches, starting off the 20-step cadence
            q = synthetic.laser_planes(L, step, nb) if thermal_mode == 2 else None
            r = e.run_steps(step, nb, 0.05, u_pick[step - 3:], u_def[step - 3:], u_np[pos:], rng_mode=1, seed=11,
                            thermal_mode=thermal_mode, q_planes=q, incremental=bool(la))
            assert r["done"] == nb
            pos += r["np_used"]
            step += nb
            res.append((r["totals"].tobytes(), r["events"].tobytes()))
        rb = e.run_supersteps(step, 45, 8, 0.02, seed=3, thermal_mode=thermal_mode,
                              q_planes=synthetic.laser_planes(L, step, 45) if thermal_mode == 2 else None)
        d = e.download_planes(0, L, state=True, theta=True, phi=True, T=True, defects=True)
        outs.append((res, rb["totals"].tobytes(), rb["n_exec"].tobytes(), {k: v.tobytes() for k, v in d.items()}, e.rate_sweep()))
        e.close()
    assert outs[0] == outs[1]


@pytest.mark.parametrize("L,n_slabs,thermal_mode", [(256, 1, 2), (256, 2, 1), (512, 1, 2)])
def test_thermal_table_fusion_identical(L, n_slabs, thermal_mode):
    """thermal_table=1 (default): the 16 x 256 temperature tiles also write the new field's rate table and plane L-1's
    deposition rates -- k_rate_table's work without a second pass over T.  Same functions on the same values: every total,
    event, field and the stand-alone sweep are bit-identical to the separate launch (thermal_table=0), through temperature
    updates with the laser source and latent heat, the cet update, several slabs, and the direct thermal calls."""
    import cetkmc
    from cetkmc import synthetic
    n = 45 if L == 256 else 24
    st, th, ph, T, df = synthetic.planes(L, 0, L, seed=5)
    rs = np.random.RandomState(6)
    u_pick, u_def, u_np = rs.random_sample(n), rs.random_sample(n), rs.random_sample(2 * n + 2)
    q = synthetic.laser_planes(L, 0, n) if thermal_mode == 2 else None
    outs = []
    for fuse in (0, 1):
        e = cetkmc.Engine(L, impurity_c=0.2, n_slabs=n_slabs)
        e.set_option("thermal_table", fuse)
        e.upload_planes(0, L, st, th, ph, T, df)
        e.set_prev_state(None)
        r = e.run_steps(0, n, 0.05, u_pick, u_def, u_np, rng_mode=1, seed=11, thermal_mode=thermal_mode, q_planes=q)
        assert r["done"] == n and r["status"] == 0
        e.thermal_cet(1e-6, True)                       # the direct calls of the drop-in thermal_solver
        sw1 = e.rate_sweep()
        e.thermal_laser(1e-6, synthetic.laser_planes(L, 0, 1)[0], use_latent=True)
        sw2 = e.rate_sweep()
        c = e.counters()
        assert c["table_updates"] > 0
        d = e.download_planes(0, L, state=True, theta=True, T=True)
        outs.append((r["totals"].tobytes(), r["events"].tobytes(), r["n_events"].tobytes(), sw1, sw2,
                     d["state"].tobytes(), d["theta"].tobytes(), d["T"].tobytes()))
        e.close()
    assert outs[0] == outs[1]


def _fuzz_case(seed):
    """A random small problem: size, fill, species mix, model parameters and a temperature field with cold / hot / non-finite
    voxels -- what no fixture of the reference holds in one place."""
    rs = np.random.RandomState(1000 + seed)
    L = int(rs.choice([5, 6, 7, 9, 11, 13, 16, 20]))
    fill = float(rs.choice([0.0, 0.02, 0.2, 0.5, 0.9, 1.0]))
    state = np.zeros((L, L, L), np.int64)
    occ = rs.random_sample((L, L, L)) < fill
    species = rs.choice([1, 2, 3, 4], size=(L, L, L), p=[0.5, 0.2, 0.2, 0.1])
    state[occ] = species[occ]
    every = bool(rs.randint(2))                           # orientations on empty sites too (the reference reads them)
    theta = np.where(every | ((state != 0) & (state != 4)), rs.uniform(0, np.pi, (L, L, L)), 0.0)
    phi = np.where(every | ((state != 0) & (state != 4)), rs.uniform(0, 2 * np.pi, (L, L, L)), 0.0)
    kind = rs.randint(4)
    if kind == 0:
        T = rs.uniform(2600.0, 4064.5, (L, L, L))
    elif kind == 1:
        T = rs.uniform(0.2, 6000.0, (L, L, L))            # below 1 K (clamped by max(T, 1)) up to far above the melting point
    elif kind == 2:
        T = np.full((L, L, L), float(rs.choice([2800.0, 3000.0, 3684.9, 3695.0, 3700.0])))
    else:
        T = 2800.0 + (3695.0 - 2800.0) / L * np.arange(L)[None, None, :] + rs.uniform(-5, 5, (L, L, L))
    weird = rs.random_sample((L, L, L)) < float(rs.choice([0.0, 0.0, 0.01]))
    T = np.where(weird, rs.choice([np.nan, np.inf, -np.inf, 0.0, -40.0, 1e308], size=(L, L, L)), T)
    defects = ((state != 0) & (rs.random_sample((L, L, L)) < 0.2)).astype(np.int64)
    tweak = dict(nu_dep=float(rs.choice([1e6, 1e10, 1e13])), I0=float(rs.choice([0.0, 5e13, 1e20])),
                 delta_T_c=float(rs.choice([0.0, 10.0, 400.0])), anisotropy=float(rs.choice([0.0, 0.25, 3.0])),
                 rate_threshold=float(rs.choice([1e-30, 1e-3, 1e9])), K_nuc=float(rs.choice([50.0, 500.0, 5000.0])))
    c = float(rs.choice([0.0, 0.1, 0.45]))
    return L, (state, theta, phi, T, defects), tweak, c, rs


@pytest.mark.parametrize("seed", range(36))
def test_fuzz_random_problems_vs_oracle(oracle_mod, seed):
    """Seeded random problems (see _fuzz_case): event list, row sums / counts / total, tree picks and 30 steps of the exact
    loop with the reference's stream bookkeeping and temperature updates -- device against oracle, same bar as the fixtures
    (types, positions, targets, counts and chosen events equal; rates / sums within 1e-11)."""
    import cetkmc
    L, fields, tweak, c, rs = _fuzz_case(seed)
    params = cetkmc.default_params(c)
    for k, v in tweak.items():
        setattr(params, k, v)
    e = cetkmc.Engine(L, impurity_c=c, params=params)
    e.upload(*fields)
    lat = oracle_mod.Lattice(*fields, impurity_c=c)
    for k, v in tweak.items():
        setattr(lat.params, k, v)
    ev_o, nd_o = lat.enumerate()
    ev_g, n_g = e.enumerate_events()
    assert n_g == len(ev_o)
    for f in ("type", "pos", "target"):
        assert np.array_equal(ev_g[f], ev_o[f]), f
    if n_g:
        fin = np.isfinite(ev_o["rate"])
        assert np.array_equal(np.isfinite(ev_g["rate"]), fin)
        assert relerr(ev_g["rate"][fin], ev_o["rate"][fin]).max() <= RATE_RTOL
    sw = lat.sweep()
    total, n_events, n_dep = e.rate_sweep()
    rsum, rcnt = e.row_sums()
    assert (n_events, n_dep) == (sw["n_events"], sw["n_dep"]) and np.array_equal(rcnt, sw["rowcnt"])
    fin = np.isfinite(sw["rowsum"])
    assert np.array_equal(np.isfinite(rsum), fin)
    assert relerr(rsum[fin], sw["rowsum"][fin]).max() <= RATE_RTOL if fin.any() else True
    if n_events and np.isfinite(sw["total"]) and sw["total"] >= 1e-25:
        for u in rs.random_sample(12):
            want = lat.select_tree(sw["blocksum"], sw["blockcnt"], sw["rowsum"], sw["rowcnt"], u * sw["total"])
            got = e.select(u * total)
            assert (got.type, tuple(got.pos), tuple(got.target), got.dep_rank) == \
                   (want.type, tuple(want.pos), tuple(want.target), want.dep_rank), u
    n = 30
    u_pick, u_def, u_np = rs.random_sample(n), rs.random_sample(n), rs.random_sample(n * (L * L + 2))
    rg = e.run_steps(0, n, 0.1, u_pick, u_def, u_np, rng_mode=0, thermal_mode=1)
    ro = lat.run_steps(0, n, 0.1, u_pick, u_def, u_np, rng_mode=0, thermal_mode=1)
    assert (rg["done"], rg["status"], rg["np_used"]) == (ro["done"], ro["status"], ro["np_used"])
    for f in ("type", "pos", "target", "atom"):
        assert np.array_equal(rg["events"][f], ro["events"][f]), f
    assert np.array_equal(rg["n_events"], ro["n_events"])
    if rg["done"]:
        assert relerr(rg["totals"][:rg["done"]], ro["totals"][:rg["done"]]).max() <= RATE_RTOL
    d = e.download()
    assert np.array_equal(d["state"], lat.state) and np.array_equal(d["theta"], lat.theta) and np.array_equal(d["phi"], lat.phi)
    assert np.array_equal(d["T"], lat.T, equal_nan=True)
    e.close()


@pytest.mark.parametrize("seed", range(16))
def test_fuzz_modes_vs_oracle(oracle_mod, seed):
    """The same random problems through the other ways of stepping: several slabs, the exact incremental loop with the counter
    species draw, and Mode B super-steps (box 8, with and without null events) -- all against the oracle's full evaluation."""
    import cetkmc
    L0, fields, tweak, c, rs = _fuzz_case(100 + seed)
    L = 16 if L0 <= 11 else 24                           # a multiple of the box
    rs2 = np.random.RandomState(77 + seed)
    reps = -(-L // L0)
    fields = tuple(np.tile(f, (reps, reps, reps))[:L, :L, :L].copy() for f in fields)
    params = cetkmc.default_params(c)
    for k, v in tweak.items():
        setattr(params, k, v)
    n_slabs = int(rs2.choice([1, 2, 3]))
    e = cetkmc.Engine(L, impurity_c=c, params=params, n_slabs=n_slabs)
    e.upload(*fields)
    lat = oracle_mod.Lattice(*fields, impurity_c=c)
    for k, v in tweak.items():
        setattr(lat.params, k, v)
    n = 45
    u_pick, u_def, u_np = rs2.random_sample(n), rs2.random_sample(n), rs2.random_sample(2 * n + 2)
    rg = e.run_steps(3, n, 0.05, u_pick, u_def, u_np, rng_mode=1, seed=seed, thermal_mode=1, incremental=True)
    ro = lat.run_steps(3, n, 0.05, u_pick, u_def, u_np, rng_mode=1, seed=seed, thermal_mode=1)
    assert (rg["done"], rg["status"], rg["np_used"]) == (ro["done"], ro["status"], ro["np_used"])
    for f in ("type", "pos", "target", "atom"):
        assert np.array_equal(rg["events"][f], ro["events"][f]), f
    assert np.array_equal(rg["n_events"], ro["n_events"])
    if rg["status"] == 0:
        for null in (False, True):
            g0 = 3 + n + (20 if null else 0)
            bg = e.run_supersteps(g0, 12, 8, 0.05, seed=seed, thermal_mode=1, want_events=True, null_events=null)
            bo = lat.run_supersteps(g0, 12, 8, 0.05, seed, thermal_mode=1, null_events=null)
            assert (bg["done"], bg["status"]) == (bo["done"], bo["status"])
            for f in ("type", "pos", "target", "atom"):
                assert np.array_equal(bg["events"][f], bo["events"][f]), (null, f)
            assert np.array_equal(bg["n_exec"], bo["n_exec"])
            if bg["done"]:
                assert relerr(bg["totals"][:bg["done"]], bo["totals"][:bg["done"]]).max() <= RATE_RTOL
            if bg["status"]:
                break
    d = e.download()
    assert np.array_equal(d["state"], lat.state) and np.array_equal(d["theta"], lat.theta) and np.array_equal(d["phi"], lat.phi)
    assert np.array_equal(d["T"], lat.T, equal_nan=True)
    e.close()


def test_staged_inputs_identical_and_checked():
    """cetkmc_stage_inputs: a batch whose streams / source planes were copied ahead gives the same bits as the plain call;
    the library refuses a staged call whose shape differs from the staged batch, and a staged batch is used once."""
    import cetkmc
    from cetkmc import synthetic
    L, n = 32, 45
    st, th, ph, T, df = synthetic.planes(L, 0, L, seed=8)
    rs = np.random.RandomState(2)
    u_pick, u_def, u_np = rs.random_sample(n), rs.random_sample(n), rs.random_sample(2 * n + 2)
    q = synthetic.laser_planes(L, 10, n)
    args = (10, n, 0.05, u_pick, u_def, u_np)
    kw = dict(rng_mode=1, seed=4, thermal_mode=2, q_planes=q)
    outs = []
    for staged in (False, True):
        e = cetkmc.Engine(L, impurity_c=0.2)
        e.upload_planes(0, L, st, th, ph, T, df)
        e.set_prev_state(None)
        if staged:
            with pytest.raises(RuntimeError, match="no staged batch"):
                e.run_steps(*args, staged=True, **kw)
            e.stage_inputs(*args, **kw)
            with pytest.raises(RuntimeError, match="does not match"):
                e.run_steps(11, n, 0.05, u_pick, u_def, u_np, staged=True, **kw)
            e.stage_inputs(*args, **kw)             # (the refused call dropped the staged batch)
        r = e.run_steps(*args, staged=staged, **kw)
        assert r["done"] == n and r["status"] == 0
        if staged:
            with pytest.raises(RuntimeError, match="no staged batch"):
                e.run_steps(*args, staged=True, **kw)
        d = e.download_planes(0, L, state=True, theta=True, phi=True, T=True, defects=True)
        outs.append((r["totals"].tobytes(), r["events"].tobytes(), r["np_used"], {k: v.tobytes() for k, v in d.items()}))
        e.close()
    assert outs[0] == outs[1]


def _fresh_engine(L, st, th, ph, T, df, n_slabs=1):
    import cetkmc
    e = cetkmc.Engine(L, impurity_c=0.2, n_slabs=n_slabs)
    e.upload_planes(0, L, st, th, ph, T, df)
    e.set_prev_state(None)
    return e


def _consumed_before(events, upto):
    """rng_mode 1: doubles of u_np consumed by the first `upto` steps (2 per deposition / nucleation)"""
    return int(2 * np.isin(events["type"][:upto], (0, 2)).sum())


@pytest.mark.parametrize("thermal_mode,incremental,n_slabs,lookahead", [(1, False, 1, 0), (2, False, 2, 0), (1, True, 1, 0), (2, True, 1, 0),
                                                                        (2, False, 1, 1), (2, True, 1, 1)])
@pytest.mark.parametrize("stop_at", [25, 40, 60])
def test_batch_that_runs_out_of_stream_continues_bit_identically(oracle_mod, thermal_mode, incremental, n_slabs, lookahead, stop_at):
    """A batch whose NumPy stream runs short stops with status 2; the steps still queued behind the stop run as
    pass-throughs on the device while the host keeps flipping the temperature buffers and freshness flags.  The
    continuation must (a) sweep from a rate table of the CURRENT field (stop_at = 25: an odd number -- one, step 40's --
    of skipped temperature updates follow the stop; the table / deposition buffers are paired with the buffer parity) and
    (b) not apply a temperature update twice when the stream ran out ON an update step (stop_at = 40, 60: the update of
    that step ran before the selection noticed the shortage; kmc_simulation.py:248-250 updates T once per 20 steps).
    lookahead = 1: the same with the look-ahead temperature update (second stream, k_thermal_fix and its pass-through
    when the batch has stopped).
    Compared with one unbroken batch (bitwise: events, totals, every field) and with the oracle."""
    from cetkmc import synthetic
    L, n = 32, 50 if stop_at < 50 else 70
    st, th, ph, T, df = synthetic.planes(L, 0, L, seed=13)
    rs = np.random.RandomState(9)
    u_pick, u_def, u_np = rs.random_sample(n), rs.random_sample(n), rs.random_sample(2 * n + 2)
    q = synthetic.laser_planes(L, 0, n) if thermal_mode == 2 else None
    kw = dict(rng_mode=1, seed=21, thermal_mode=thermal_mode, incremental=incremental)
    ref = _fresh_engine(L, st, th, ph, T, df, n_slabs)
    ref.set_option("thermal_lookahead", lookahead)
    r0 = ref.run_steps(0, n, 0.05, u_pick, u_def, u_np, q_planes=q, **kw)
    assert r0["done"] == n and r0["status"] == 0
    d0 = ref.download_planes(0, L, state=True, theta=True, phi=True, T=True)
    sweep0 = ref.rate_sweep()
    ref.close()
    # the same run, the first call given a stream that ends right before step `stop_at` needs its two orientation slots
    cap = _consumed_before(r0["events"], stop_at) + 1
    e = _fresh_engine(L, st, th, ph, T, df, n_slabs)
    e.set_option("thermal_lookahead", lookahead)
    r1 = e.run_steps(0, n, 0.05, u_pick, u_def, u_np[:cap], q_planes=q, **kw)
    assert r1["status"] == 2 and r1["done"] == stop_at and r1["np_used"] == cap - 1
    q2 = synthetic.laser_planes(L, stop_at, n - stop_at) if thermal_mode == 2 else None
    r2 = e.run_steps(stop_at, n - stop_at, 0.05, u_pick[stop_at:], u_def[stop_at:], u_np[r1["np_used"]:], q_planes=q2, **kw)
    assert r2["status"] == 0 and r2["done"] == n - stop_at
    assert np.concatenate([r1["events"], r2["events"]]).tobytes() == r0["events"].tobytes()
    assert np.array_equal(np.concatenate([r1["totals"], r2["totals"]]), r0["totals"])
    d1 = e.download_planes(0, L, state=True, theta=True, phi=True, T=True)
    for f in d0:
        assert np.array_equal(d0[f], d1[f]), f
    assert e.rate_sweep() == sweep0
    e.close()
    lat = oracle_mod.Lattice(st.astype(np.int64), th, ph, T, df.astype(np.int64), impurity_c=0.2)
    ro = lat.run_steps(0, n, 0.05, u_pick, u_def, u_np, rng_mode=1, seed=21, thermal_mode=thermal_mode, q_planes=q)
    for f in ("type", "pos", "target", "atom"):
        assert np.array_equal(r0["events"][f], ro["events"][f]), f
    assert np.array_equal(d0["T"], lat.T) and np.array_equal(d0["state"], lat.state)


def test_terminated_batch_leaves_a_consistent_engine(oracle_mod):
    """status 1 (no valid events, kmc_simulation.py:260-262) in the middle of a batch: the queued rest passes through;
    afterwards the rate sweep of the engine equals the oracle's on the same lattice, and stepping on after the caller has
    made events possible again (a new temperature field) matches the oracle step by step."""
    import cetkmc
    L, n = 16, 70
    # solid lattice with defects only (state 4: no events) except two vacancies whose neighbours are defects too:
    # the only events are the vacancies' nucleations; once both are filled the run terminates
    state = np.full((L, L, L), 4, np.int64)
    state[3, 4, 5] = state[9, 9, 9] = 0
    zeros = np.zeros((L, L, L))
    T = np.full((L, L, L), 3000.0)
    rs = np.random.RandomState(2)
    u_pick, u_np = rs.random_sample(n), rs.random_sample(2 * n + 2)
    e = cetkmc.Engine(L, impurity_c=0.2)
    e.upload(state, zeros, zeros, T, None)
    lat = oracle_mod.Lattice(state, zeros, zeros, T, None, impurity_c=0.2)
    rg = e.run_steps(0, n, 0.0, u_pick, None, u_np, rng_mode=1, seed=1, thermal_mode=1)
    ro = lat.run_steps(0, n, 0.0, u_pick, None, u_np, rng_mode=1, seed=1, thermal_mode=1)
    assert rg["status"] == ro["status"] == 1 and rg["done"] == ro["done"] == 2
    sw = lat.sweep()
    assert e.rate_sweep() == (sw["total"], sw["n_events"], sw["n_dep"]) == (0.0, 0, 0)
    d = e.download()
    assert np.array_equal(d["state"], lat.state) and np.array_equal(d["T"], lat.T)
    # open a few voxels again and continue: tables and interface sums must be those of the new state
    s2 = lat.state.astype(np.int64).copy()
    s2[5:8, 5:8, 5:8] = 0
    s2[6, 6, 6] = 1
    e.upload(s2, d["theta"], d["phi"], d["T"], None)
    lat2 = oracle_mod.Lattice(s2, d["theta"], d["phi"], d["T"], None, impurity_c=0.2)
    rg = e.run_steps(2, 20, 0.0, u_pick[2:], None, u_np, rng_mode=1, seed=1, thermal_mode=1)
    ro = lat2.run_steps(2, 20, 0.0, u_pick[2:], None, u_np, rng_mode=1, seed=1, thermal_mode=1)
    assert rg["done"] == ro["done"] == 20
    for f in ("type", "pos", "target", "atom"):
        assert np.array_equal(rg["events"][f], ro["events"][f]), f
    e.close()


def test_selection_margin_is_reported(oracle_mod):
    """cetkmc_run_result.min_margin (hard part 3): the smallest distance of a pick r = u * total from the
    nearer end of the chosen event's interval of the cumulative rate sum, relative to the total.  A u placed 1e-13 (relative)
    inside an event boundary of the canonical sum is flagged; ordinary uniforms are not.  (The reference scans a sequentially
    rounded sum, kmc_simulation.py:259,265-274, which differs from the tree sum by ~1e-13 relative: a smaller margin means
    its pick could be the neighbouring event.)"""
    import cetkmc
    L = 16
    state, theta, phi, T, defects = random_lattice(L, 3, fill=0.2)
    lat = oracle_mod.Lattice(state, theta, phi, T, defects, impurity_c=0.2)
    sw = lat.sweep()

    def one_step(u):
        e = cetkmc.Engine(L, impurity_c=0.2)
        e.upload(state, theta, phi, T, defects)
        r = e.run_steps(0, 1, 0.0, np.array([u]), None, np.zeros(4), rng_mode=1, seed=1, thermal_mode=0)
        e.close()
        assert r["done"] == 1
        return r
    r0 = one_step(0.4321)
    assert 1e-9 < r0["min_margin"] <= 1.0
    ev = r0["events"][0]
    # the chosen event's interval ends `margin_hi` above r: move u to just below that end
    total = r0["totals"][0]
    want = lat.select_tree(sw["blocksum"], sw["blockcnt"], sw["rowsum"], sw["rowcnt"], 0.4321 * sw["total"])
    assert tuple(ev["pos"]) == tuple(want.pos) and ev["type"] == want.type
    # bisect u upwards until the pick changes: the boundary of the chosen event's interval in the canonical sum
    lo, hi = 0.4321, 0.4321 + 2.0 * ev["rate"] / total
    assert one_step(hi)["events"][0].tobytes() != ev.tobytes()
    for _ in range(60):
        mid = 0.5 * (lo + hi)
        if one_step(mid)["events"][0].tobytes() == ev.tobytes():
            lo = mid
        else:
            hi = mid
    near = one_step(lo)
    assert near["events"][0].tobytes() == ev.tobytes() and near["min_margin"] < 1e-12
    # a batch reports the minimum over its steps
    e = cetkmc.Engine(L, impurity_c=0.2)
    e.upload(state, theta, phi, T, defects)
    rs = np.random.RandomState(1)
    u = rs.random_sample(30)
    u[0] = lo
    r = e.run_steps(0, 30, 0.0, u, None, rs.random_sample(70), rng_mode=1, seed=1, thermal_mode=0)
    assert r["done"] == 30 and r["min_margin"] == near["min_margin"]
    r = e.run_steps(30, 30, 0.0, rs.random_sample(30), None, rs.random_sample(70), rng_mode=1, seed=1, thermal_mode=0)
    assert r["min_margin"] > 1e-9                                   # reset per batch
    e.close()
