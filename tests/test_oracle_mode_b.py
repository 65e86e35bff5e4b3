"""Mode B (synchronous super-steps over spatial boxes; NOT in the reference) -- checks of its CPU comparator:
the single-box case is Mode A, boxes stay inside their windows, events of a super-step are disjoint."""
import numpy as np
import pytest

from helpers import random_lattice


def _lat(oracle_mod, L, seed, fill, c=0.2):
    state, theta, phi, T, defects = random_lattice(L, seed, fill=fill)
    return oracle_mod.Lattice(state, theta, phi, T, defects, impurity_c=c)


@pytest.mark.parametrize("L,fill,df", [(8, 0.2, 0.1), (10, 0.05, 0.0), (12, 0.4, 0.05)])
def test_single_box_is_mode_a(oracle_mod, L, fill, df):
    """box == L: one domain, no sectors.  Feeding Mode A (run_steps, counter species draw) the same
    uniforms gives the same events, totals and lattice, step by step."""
    n, seed = 45, 77
    b = _lat(oracle_mod, L, 5, fill)
    rb = b.run_supersteps(3, n, L, df, seed, thermal_mode=1)
    assert rb["done"] == n and np.all(rb["n_exec"] == 1)
    ev = rb["events"][:, 0]
    u_pick = np.array([oracle_mod.counter_uniform(seed, 3 + s, oracle_mod.KEY_PICK) for s in range(n)])
    u_def = np.array([oracle_mod.counter_uniform(seed, 3 + s, oracle_mod.KEY_DEFECT) for s in range(n)])
    u_np = []
    for s in range(n):
        if ev["type"][s] in (0, 2):
            u_np += [oracle_mod.counter_uniform(seed, 3 + s, oracle_mod.KEY_THETA),
                     oracle_mod.counter_uniform(seed, 3 + s, oracle_mod.KEY_PHI)]
    a = _lat(oracle_mod, L, 5, fill)
    ra = a.run_steps(3, n, df, u_pick, u_def, np.array(u_np + [0.0, 0.0]), rng_mode=1, seed=seed, thermal_mode=1)
    assert ra["done"] == n and ra["np_used"] == len(u_np)
    for f in ("type", "pos", "target", "atom", "rate"):
        assert np.array_equal(ra["events"][f], ev[f]), f
    assert np.array_equal(ra["totals"], rb["totals"])
    for x, y in ((a.state, b.state), (a.theta, b.theta), (a.phi, b.phi), (a.T, b.T)):
        assert np.array_equal(x, y)
    assert a.nuc_count == b.nuc_count


@pytest.mark.parametrize("L,box", [(16, 8), (20, 10), (16, 16)])
def test_windows_and_disjointness(oracle_mod, L, box):
    n, seed = 17, 3
    lat = _lat(oracle_mod, L, 9, 0.15)
    before = lat.state.copy()
    r = lat.run_supersteps(0, n, box, 0.05, seed, thermal_mode=1)
    assert r["done"] == n
    nb, H = L // box, (L if box == L else box // 2)
    ev = r["events"]
    assert ev.shape == (n, nb ** 3)
    for s in range(n):
        sec = s % 8
        sh = np.array([(sec >> 2) & 1, (sec >> 1) & 1, sec & 1]) * (0 if box == L else H)
        live = ev[s][ev[s]["type"] >= 0]
        assert len(live) == r["n_exec"][s]
        d = np.nonzero(ev[s]["type"] >= 0)[0]
        org = np.stack([d // (nb * nb), (d // nb) % nb, d % nb], 1) * box + sh
        assert np.all(live["pos"] >= org) and np.all(live["pos"] < org + H)
        # written voxels (pos, and target of a diffusion) are pairwise distinct
        w = [tuple(p) for p in live["pos"]] + [tuple(t) for t in live["target"][live["type"] == 1]]
        assert len(w) == len(set(w))
    assert r["n_exec"].sum() > n or box == L          # several events per sweep
    assert (lat.state != before).sum() > 0


def test_oracle_threads_do_not_change_results(oracle_mod):
    """The oracle's row-sum and thermal loops may run on several host threads (bench.py's all-core CPU
    baseline); rows are independent, so every output is bit-identical to the scalar run."""
    outs = []
    for nt in (1, 4):
        assert oracle_mod.set_threads(nt) == nt
        lat = _lat(oracle_mod, 14, 21, 0.3)
        rs = np.random.RandomState(3)
        n = 25
        r = lat.run_steps(0, n, 0.05, rs.random_sample(n), rs.random_sample(n), rs.random_sample(n * 200), rng_mode=0,
                          thermal_mode=1)
        outs.append((r["totals"].tobytes(), r["events"].tobytes(), lat.state.tobytes(), lat.T.tobytes()))
    oracle_mod.set_threads(1)
    assert outs[0] == outs[1]


def test_time_advance_is_the_reference_formula_per_executed_event(oracle_mod):
    """dt_event[g] = max(-ln(max(1e-12, u_g)) / totals[g], 1e-12) with u_g = u(seed, g, KEY_DT): kmc_simulation.py:331-332
    restated per executed event (one draw per super-step).  At this model's rates the 1e-12 floor binds, as it does in every
    reference trajectory; a cold lattice with tiny rates exercises the other branch."""
    import math
    lat = _lat(oracle_mod, 16, 4, 0.1)
    r = lat.run_supersteps(7, 12, 8, 0.0, 99, thermal_mode=1)
    for s in range(12):
        u = oracle_mod.counter_uniform(99, 7 + s, oracle_mod.KEY_DT)
        want = max(-math.log(max(1e-12, u)) / r["totals"][s], 1e-12)
        assert r["dt_event"][s] == want
    assert np.all(r["dt_event"] == 1e-12)
    # a lattice whose total stays below 1e12 / s, where -ln(u) / total exceeds the floor: solid tungsten pointing "down"
    # (theta = pi: attachment barrier E_b, kmc_event_rates.py:154), three isolated vacancies, 5 K below the melting point
    # (no nucleation: dT <= DELTA_T_C, :120)
    L = 8
    state = np.ones((L, L, L), np.int64)
    state[2, 3, 4] = state[5, 5, 1] = state[6, 1, 6] = 0
    theta = np.where(state != 0, np.pi, 0.0)
    slow = oracle_mod.Lattice(state, theta, np.zeros((L, L, L)), np.full((L, L, L), 3690.0), None, impurity_c=0.0)
    rc = slow.run_supersteps(0, 3, 8, 0.0, 5, thermal_mode=0)
    assert rc["done"] == 3 and rc["totals"].max() < 1e12
    for s in range(3):
        u = oracle_mod.counter_uniform(5, s, oracle_mod.KEY_DT)
        assert rc["dt_event"][s] == -math.log(u) / rc["totals"][s] > 1e-12


def test_null_events_accept_a_subset_of_the_picks(oracle_mod):
    """null_events: the picks of the first super-step are those of the plain mode (same frozen lattice, same uniforms);
    the accepted ones are a subset, the others are logged as type -2 with the same voxel and rate; the single-box case
    (R_d == R_max) accepts everything, so it stays Mode A."""
    a, b = _lat(oracle_mod, 16, 9, 0.3), _lat(oracle_mod, 16, 9, 0.3)
    ra = a.run_supersteps(0, 1, 8, 0.0, 11, thermal_mode=1)
    rb = b.run_supersteps(0, 1, 8, 0.0, 11, thermal_mode=1, null_events=True)
    ea, eb = ra["events"][0], rb["events"][0]
    assert np.array_equal(ea["pos"], eb["pos"]) and np.array_equal(ea["rate"], eb["rate"])
    assert set(np.unique(eb["type"])) <= {-2, -1, 0, 1, 2, 3}
    live = eb["type"] >= 0
    assert np.array_equal(ea["type"][live], eb["type"][live]) and np.all(ea["type"][eb["type"] == -2] >= 0)
    assert np.array_equal(ea["type"] == -1, eb["type"] == -1)
    assert rb["n_exec"][0] == live.sum() >= 1                      # the box holding R_max always executes
    assert np.array_equal(ra["dt_event"], rb["dt_event"])
    one_a, one_b = _lat(oracle_mod, 10, 2, 0.2), _lat(oracle_mod, 10, 2, 0.2)
    r1 = one_a.run_supersteps(0, 30, 10, 0.05, 3, thermal_mode=1)
    r2 = one_b.run_supersteps(0, 30, 10, 0.05, 3, thermal_mode=1, null_events=True)
    assert r1["events"].tobytes() == r2["events"].tobytes() and np.array_equal(one_a.state, one_b.state)


def test_null_events_acceptance_follows_the_window_rate(oracle_mod):
    """Boxes 0..3 (planes 0..7) hold ONE empty voxel per active window among defects (state 4: no events), boxes 4..7 are
    empty: at constant temperature every empty voxel owns the same nucleation rate r, so R_d = r in the sparse boxes and
    64 r in the empty ones (= R_max, always accepted).  The sparse boxes must execute with probability 1/64."""
    L, box = 16, 8
    state = np.zeros((L, L, L), np.int64)
    state[:8] = 4
    for dj in range(2):
        for dk in range(2):
            state[1, dj * 8 + 2, dk * 8 + 1] = 0            # inside the octant-0 window of boxes (0, dj, dk)
    zeros = np.zeros((L, L, L))
    T = np.full((L, L, L), 3000.0)
    hits, trials = 0, 0
    for seed in range(160):
        lat = oracle_mod.Lattice(state, zeros, zeros, T, None, impurity_c=0.0)
        r = lat.run_supersteps(0, 1, box, 0.0, seed, thermal_mode=0, null_events=True)
        t = r["events"][0]["type"]
        assert np.all(t[4:] >= 0) and np.all(t[:4] != -1)    # empty boxes always execute; sparse boxes always pick
        hits += int((t[:4] >= 0).sum())
        trials += 4
    assert trials == 640 and 2 <= hits <= 25                 # binomial(640, 1/64): mean 10, sigma 3.1
