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
