"""Shared test helpers: fixture loading and a run_kmc replay driver.

``replay_run_kmc`` re-enacts kmc_simulation.py:203-398 of the reference with the host
RNG streams (CPython ``random`` + NumPy legacy global) handled exactly as the reference
does, on top of an abstract *backend* offering sweep / select / apply / thermal.  The
same driver is used with the CPU oracle (not-gpu tests) and with the HIP engine (gpu
tests), so both are checked against the same reference-generated trajectories.
"""
import os
import random

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
TYPE_NAMES = ("dep", "diff", "nuc", "att")


def load(name):
    return np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)


def relerr(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    den = np.maximum(np.abs(b), 1e-300)
    out = np.abs(a - b) / den
    out[(a == b)] = 0.0
    return out


def fixture_step_diffs(z):
    """step -> {flat_idx: (state, theta, phi)} from a traj fixture."""
    d = {}
    for s, fi, st, th, ph in zip(z["diff_step"], z["diff_idx"], z["diff_state"], z["diff_theta"], z["diff_phi"]):
        d.setdefault(int(s), {})[int(fi)] = (int(st), float(th), float(ph))
    return d


class OracleBackend:
    """Backend protocol over oracle.Lattice (canonical-tree selection, like the GPU)."""

    def __init__(self, oracle_mod, state, theta, phi, T, defects, impurity_c):
        self.o = oracle_mod
        self.lat = oracle_mod.Lattice(state, theta, phi, T, defects, impurity_c=impurity_c)
        self._sw = None

    def thermal_cet(self, dt):
        self.lat.thermal_cet(dt=dt, scrub_nan=True)

    def sweep(self):
        self._sw = self.lat.sweep()
        return self._sw["total"], self._sw["n_events"], self._sw["n_dep"]

    def select(self, r):
        sw = self._sw
        return self.lat.select_tree(sw["blocksum"], sw["blockcnt"], sw["rowsum"], sw["rowcnt"], r)

    def apply(self, ev, theta_new, phi_new, make_defect):
        self.lat.apply(ev, theta_new, phi_new, make_defect)

    def set_defects(self, mask):
        self.lat.defects = np.ascontiguousarray(mask, dtype=np.int8)

    def fields(self):
        return self.lat.state, self.lat.theta, self.lat.phi, self.lat.T


def dep_species(u, impurity_c, impurity_re=0.10):
    """kmc_event_rates.py:66-71"""
    if u < impurity_c:
        return 3
    if u < impurity_c + impurity_re:
        return 2
    return 1


def replay_run_kmc(z, make_backend, check_every_step=True, rate_rtol=1e-12):
    """Re-enact run_kmc (kmc_simulation.py:203-398) on a backend and compare with the
    trajectory fixture ``z`` step by step.  Returns the backend."""
    import defects as host_defects
    import lattice_init as host_init

    L, n_steps = int(z["L"]), int(z["n_steps"])
    temp, df = float(z["temp"]), float(z["defect_fraction"])
    n_seeds, c = int(z["n_seeds"]), float(z["impurity_c"])
    diffs = fixture_step_diffs(z)
    Tsnap = {int(s): t for s, t in zip(z["T_steps"], z["T_snaps"])}
    Dsnap = {int(s): d for s, d in zip(z["D_steps"], z["D_snaps"])}

    np.random.seed(42)
    random.seed(42)
    state, theta, phi, T, atom = host_init.initialize_lattice(lattice_size=L, n_seeds=n_seeds, T_sub=temp, impurity_c=c)
    mask, _ = host_defects.introduce_defects(state, atom, T, apply_to_state=False)
    be = make_backend(state, theta, phi, T, mask, c)
    total_time = 0.0
    prev = None
    for step in range(n_steps):
        if step % 20 == 0:
            be.thermal_cet(1e-6)
        if check_every_step:
            if step in Tsnap:
                assert np.array_equal(be.fields()[3], Tsnap[step]), f"T mismatch at step {step}"
            if step in Dsnap:
                assert np.array_equal(mask, Dsnap[step]), f"defect mask mismatch at step {step}"
        total, n_events, n_dep = be.sweep()
        assert n_events == int(z["n_events"][step]), (step, n_events, int(z["n_events"][step]))
        assert n_dep == int(z["n_dep"][step]), step
        ref_tot = float(z["seq_total"][step])
        assert abs(total - ref_tot) <= rate_rtol * abs(ref_tot), (step, total, ref_tot)
        if n_events == 0 or total < 1e-25 or not np.isfinite(total):
            break
        u_dep = np.random.random(n_dep)                    # kmc_event_rates.py:65, one per candidate
        ev = be.select(random.random() * total)            # kmc_simulation.py:265
        if ev.type == 0:
            ev.atom = dep_species(u_dep[ev.dep_rank], c)
        th = ph = 0.0
        if ev.type in (0, 2):
            th = np.random.uniform(0, np.pi)               # :283-284 / :308-309
            ph = np.random.uniform(0, 2 * np.pi)
        mk = bool(df > 0.0 and random.random() < df)       # :323
        if check_every_step:
            prev = tuple(a.copy() for a in be.fields()[:3])
        be.apply(ev, th, ph, mk)
        dt = max(-np.log(max(1e-12, random.random())) / total, 1e-12)   # :331
        total_time += dt
        if check_every_step:
            cur = be.fields()[:3]
            ch = np.flatnonzero((cur[0].ravel() != prev[0].ravel()) | (cur[1].ravel() != prev[1].ravel())
                                | (cur[2].ravel() != prev[2].ravel()))
            got = {int(fi): (int(cur[0].ravel()[fi]), float(cur[1].ravel()[fi]), float(cur[2].ravel()[fi])) for fi in ch}
            assert got == diffs.get(step, {}), (step, TYPE_NAMES[ev.type], got, diffs.get(step, {}))
        if step % 200 == 0:                                 # :335-338
            s_now = be.fields()[0].astype(np.int64)
            mask, _ = host_defects.introduce_defects(s_now, s_now, be.fields()[3], apply_to_state=False)
            be.set_defects(mask)
    st, th_, ph_, _ = be.fields()
    assert np.array_equal(st, z["final_state"])
    assert np.array_equal(th_, z["final_theta"]) and np.array_equal(ph_, z["final_phi"])
    assert total_time == float(z["total_time"])
    assert np.array_equal(np.array([random.random() for _ in range(4)]), z["py_next"])
    assert np.array_equal(np.random.random(4), z["np_next"])
    return be


class GpuBackend:
    """Backend protocol over cetkmc.Engine (single-step C-ABI entry points)."""

    def __init__(self, state, theta, phi, T, defects, impurity_c, n_slabs=1):
        import cetkmc
        self.e = cetkmc.Engine(int(state.shape[0]), impurity_c=impurity_c, n_slabs=n_slabs)
        self.e.upload(state, theta, phi, T, defects)

    def thermal_cet(self, dt):
        self.e.thermal_cet(dt, scrub_nan=True)

    def sweep(self):
        return self.e.rate_sweep()

    def select(self, r):
        return self.e.select(r)

    def apply(self, ev, theta_new, phi_new, make_defect):
        self.e.apply(ev, theta_new, phi_new, make_defect)

    def set_defects(self, mask):
        self.e.set_defects(mask)

    def fields(self):
        d = self.e.download()
        return d["state"], d["theta"], d["phi"], d["T"]


def random_lattice(L, seed, fill=0.3, t_lo=2600.0, t_hi=3690.0, hot_frac=0.05):
    """Synthetic lattice exercising all four event families (test-only generator)."""
    rs = np.random.RandomState(seed)
    state = np.zeros((L, L, L), dtype=np.int64)
    occ = rs.random_sample((L, L, L)) < fill
    species = rs.choice([1, 2, 3, 4], size=(L, L, L), p=[0.6, 0.15, 0.2, 0.05])
    state[occ] = species[occ]
    theta = np.where((state != 0) & (state != 4), rs.uniform(0, np.pi, (L, L, L)), 0.0)
    phi = np.where((state != 0) & (state != 4), rs.uniform(0, 2 * np.pi, (L, L, L)), 0.0)
    T = rs.uniform(t_lo, t_hi, (L, L, L))
    hot = rs.random_sample((L, L, L)) < hot_frac
    T[hot] = rs.uniform(3690.0, 4064.5, int(hot.sum()))
    defects = ((state == 3) & (rs.random_sample((L, L, L)) < 0.3)).astype(np.int64)
    return state, theta, phi, T, defects
