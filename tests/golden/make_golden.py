#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ from the *reference itself*.

Runs ONLY in the build container (needs /root/reference); the GPU box and the
test-suite only ever read the committed ``*.npz`` / ``*.json`` files.

The reference imports ``numba`` (kmc_event_rates.py:2), which is not installed
in this image.  ``numba.jit`` is semantically an identity decorator, so this
script puts a 6-line module named ``numba`` (identity ``jit``) on ``sys.path``
in a temporary directory and then imports the reference *unmodified* from
/root/reference.  Nothing of the reference is copied: the fixtures are pure
data (inputs + the outputs the reference produced for them).

Semantic gap (documented in DESIGN.md): under this "pure-Python mode" the
deposition-species draw (kmc_event_rates.py:65) comes from NumPy's seeded global
MT19937 stream; under real numba it would come from numba's own generator.

Usage:  python tests/golden/make_golden.py [--only NAME]
"""
import argparse
import contextlib
import hashlib
import io
import json
import os
import random
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"

sys.dont_write_bytecode = True
_stub_dir = tempfile.mkdtemp(prefix="numba_identity_")
os.makedirs(os.path.join(_stub_dir, "numba"))
with open(os.path.join(_stub_dir, "numba", "__init__.py"), "w") as f:
    f.write(
        "import numpy as _np\n"
        "def jit(*a, **k):\n"
        "    if len(a) == 1 and callable(a[0]) and not k:\n"
        "        return a[0]\n"
        "    return lambda fn: fn\n"
        "int64 = _np.int64\nfloat64 = _np.float64\n"
    )
sys.path.insert(0, _stub_dir)
sys.path.insert(0, REF)

import matplotlib  # noqa: E402

matplotlib.use("Agg")
import numpy as np  # noqa: E402

_work = tempfile.mkdtemp(prefix="golden_work_")
os.chdir(_work)  # the reference writes outputs/<prefix>/metrics.csv relative to cwd

import constants as C  # noqa: E402
import defects as ref_defects  # noqa: E402
import kmc_event_rates as ref_rates  # noqa: E402
import kmc_simulation as ref_sim  # noqa: E402
import lattice_init as ref_init  # noqa: E402
import metrics as ref_metrics  # noqa: E402
import thermal_solver as ref_thermal  # noqa: E402
import utils as ref_utils  # noqa: E402

TYPE_CODE = {b"dep": 0, b"diff": 1, b"nuc": 2, b"att": 3}


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def pack_events(events):
    n = len(events)
    out = dict(
        etype=np.zeros(n, np.uint8),
        pos=np.zeros((n, 3), np.int16),
        rate=np.zeros(n, np.float64),
        target=np.zeros((n, 3), np.int16),
        atom=np.zeros(n, np.uint8),
    )
    for m, (t, p, r, tg, a) in enumerate(events):
        out["etype"][m] = TYPE_CODE[t]
        out["pos"][m] = p
        out["rate"][m] = r
        out["target"][m] = tg
        out["atom"][m] = a
    return out


def save(name, **arrays):
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **arrays)
    print(f"  wrote {name}.npz  ({os.path.getsize(path)/1024:.1f} KiB)")


# --------------------------------------------------------------------------
# F1: ordered event lists from get_event_rates on crafted lattices
# --------------------------------------------------------------------------
def crafted_lattice(L, seed, fill, nonzero_empty_orient=False, t_mode="mixed"):
    rs = np.random.RandomState(seed)
    state = np.zeros((L, L, L), dtype=np.int64)
    occ = rs.random_sample((L, L, L)) < fill
    species = rs.choice([1, 2, 3, 4], size=(L, L, L), p=[0.55, 0.15, 0.2, 0.1])
    state[occ] = species[occ]
    theta = np.where(state != 0, rs.uniform(0, np.pi, (L, L, L)), 0.0)
    phi = np.where(state != 0, rs.uniform(0, 2 * np.pi, (L, L, L)), 0.0)
    theta[state == 4] = 0.0
    phi[state == 4] = 0.0
    if nonzero_empty_orient:
        theta = rs.uniform(0, np.pi, (L, L, L))
        phi = rs.uniform(0, 2 * np.pi, (L, L, L))
    if t_mode == "const":
        T = np.full((L, L, L), 3000.0)
    elif t_mode == "ramp":
        T = np.repeat(np.repeat((2800.0 + (895.0 / L) * np.arange(L))[None, None, :], L, 0), L, 1).copy()
    else:
        T = rs.uniform(2500.0, 3690.0, (L, L, L))
        flat = T.reshape(-1)
        n = flat.size
        pick = rs.choice(n, size=max(8, n // 6), replace=False)
        specials = [0.5, -5.0, 0.0, 1.0, 3685.0, 3685.0000001, 3684.9999999, 3694.9, 3695.0,
                    3696.0, 3700.0, 3800.0, 4064.5, 2800.0, 3694.0, 3690.0, 5000.0, 1e-3, 300.0, 30.0, 12.0]
        for q, idx in enumerate(pick):
            flat[idx] = specials[q % len(specials)]
    defects = np.zeros((L, L, L), dtype=np.int64)
    dm = (state == 3) & (rs.random_sample((L, L, L)) < 0.5)
    defects[dm] = 1
    if nonzero_empty_orient:  # also exercise defect flags on non-C sites (API allows any mask)
        defects[(state == 1) & (rs.random_sample((L, L, L)) < 0.1)] = 1
    return state, theta, phi, T, defects


def gen_events():
    cases = [
        # name, L, seed, fill, nonzero-empty-orientation, T mode, impurity_c
        ("events_L6_mixed", 6, 11, 0.35, False, "mixed", 0.1),
        ("events_L10_mixed", 10, 12, 0.30, False, "mixed", 0.2),
        ("events_L13_general", 13, 13, 0.40, True, "mixed", 0.0),
        ("events_L9_const", 9, 14, 0.25, False, "const", 0.3),
        ("events_L12_ramp_sparse", 12, 15, 0.03, False, "ramp", 0.1),
        ("events_L5_dense", 5, 16, 0.85, False, "mixed", 0.2),
        ("events_L3_tiny", 3, 17, 0.4, False, "mixed", 0.1),
        ("events_L1", 1, 18, 0.0, False, "ramp", 0.1),
        ("events_L2", 2, 19, 0.5, False, "mixed", 0.1),
        ("events_L16_mixed", 16, 20, 0.2, False, "mixed", 0.15),
    ]
    for name, L, seed, fill, nze, tmode, c in cases:
        state, theta, phi, T, defects = crafted_lattice(L, seed, fill, nze, tmode)
        np.random.seed(1000 + seed)
        with np.errstate(all="ignore"):
            ev = ref_rates.get_event_rates(state, theta, phi, T, state.copy(), defects, L, 1, 2, 3,
                                           step=0, debug_step=1000, impurity_c=c)
        seq_total = 0.0
        for e in ev:
            seq_total += e[2]
        save(name, L=L, state=state.astype(np.int8), theta=theta, phi=phi, T=T,
             defects=defects.astype(np.int8), impurity_c=c, np_seed=1000 + seed,
             seq_total=seq_total, **pack_events(ev))


def gen_events_odd():
    """Event lists on inputs the crafted cases above do not hold: non-finite temperatures (NaN, +-inf, 1e308) scattered over
    the lattice and on plane L-1, a completely filled and a completely empty lattice, orientations on empty sites."""
    cases = [
        # name, L, seed, fill, nonzero-empty-orientation, impurity_c, fraction of non-finite temperatures
        ("events_L7_nonfinite", 7, 31, 0.30, False, 0.1, 0.08),
        ("events_L8_nonfinite_orient", 8, 32, 0.45, True, 0.2, 0.05),
        ("events_L5_full", 5, 33, 1.0, False, 0.1, 0.0),
        ("events_L6_empty_nonfinite", 6, 34, 0.0, False, 0.3, 0.10),
        ("events_L4_dense_nonfinite", 4, 35, 0.8, True, 0.0, 0.15),
    ]
    for name, L, seed, fill, nze, c, frac in cases:
        state, theta, phi, T, defects = crafted_lattice(L, seed, fill, nze, "mixed")
        rs = np.random.RandomState(500 + seed)
        odd = rs.random_sample((L, L, L)) < frac
        T = np.where(odd, rs.choice([np.nan, np.inf, -np.inf, 1e308, -1e308], size=(L, L, L)), T)
        np.random.seed(1000 + seed)
        with np.errstate(all="ignore"):
            ev = ref_rates.get_event_rates(state, theta, phi, T, state.copy(), defects, L, 1, 2, 3,
                                           step=0, debug_step=1000, impurity_c=c)
        seq_total = 0.0
        for e in ev:
            seq_total += e[2]
        save(name, L=L, state=state.astype(np.int8), theta=theta, phi=phi, T=T,
             defects=defects.astype(np.int8), impurity_c=c, np_seed=1000 + seed,
             seq_total=seq_total, **pack_events(ev))


# --------------------------------------------------------------------------
# helper lattice used by several generators: the bench "config 2/3" fill rule
# --------------------------------------------------------------------------
def gen_events_initlattice():
    """Single sweep on the reference's own initial lattice (SURVEY 8c known answer:
    L=16, seeds=5, c=0.1, after one _cet update -> 4439 events)."""
    L = 16
    np.random.seed(42)
    state, theta, phi, T, atom = ref_init.initialize_lattice(lattice_size=L, n_seeds=5, T_sub=2800, impurity_c=0.1)
    mask, _ = ref_defects.introduce_defects(state, atom, T, apply_to_state=False)
    T1 = ref_thermal.update_temperature_cet(np.nan_to_num(T, nan=C.T_SUB), state, dt=1e-6)
    st = np.random.get_state()
    with np.errstate(all="ignore"):
        ev = ref_rates.get_event_rates(state, theta, phi, T1, atom, mask, L, 1, 2, 3, impurity_c=0.1)
    seq_total = 0.0
    for e in ev:
        seq_total += e[2]
    np.random.set_state(st)
    n_dep = sum(1 for e in ev if e[0] == b"dep")
    u_dep = np.random.random(n_dep)
    save("events_L16_init", L=L, state=state.astype(np.int8), theta=theta, phi=phi, T=T1,
         defects=mask.astype(np.int8), impurity_c=0.1, u_dep=u_dep, seq_total=seq_total, **pack_events(ev))


# --------------------------------------------------------------------------
# F2: run_kmc trajectories (per-step records via a recording wrapper around
# the module attribute kmc_simulation.get_event_rates; the reference source is
# not modified and its RNG consumption is unchanged)
# --------------------------------------------------------------------------
class Recorder:
    def __init__(self, orig):
        self.orig = orig
        self.prev = None
        self.steps = []          # per call: n_events, n_dep, seq_total
        self.diffs = []          # (step_applied, flat_idx, state, theta, phi)
        self.T_snaps = {}        # step -> T copy (only when T changed)
        self.defect_snaps = {}   # step -> defects mask copy (when it changed)
        self._lastT = None
        self._lastD = None

    def __call__(self, state, theta, phi, T, atom_type, defects_mask, L, w, re, c, step=0, debug_step=1000, impurity_c=0.0):
        self.observe(state, theta, phi, step)
        if self._lastT is None or not np.array_equal(self._lastT, T):
            self.T_snaps[step] = T.copy()
            self._lastT = T.copy()
        if self._lastD is None or not np.array_equal(self._lastD, defects_mask):
            self.defect_snaps[step] = defects_mask.astype(np.int8)
            self._lastD = defects_mask.copy()
        assert np.array_equal(state, atom_type)
        ev = self.orig(state, theta, phi, T, atom_type, defects_mask, L, w, re, c,
                       step=step, debug_step=debug_step, impurity_c=impurity_c)
        tot = 0.0
        for e in ev:
            tot += e[2]
        self.steps.append((len(ev), sum(1 for e in ev if e[0] == b"dep"), tot))
        return ev

    def observe(self, state, theta, phi, step):
        cur = (state.copy(), theta.copy(), phi.copy())
        if self.prev is not None:
            ch = np.flatnonzero((cur[0] != self.prev[0]).ravel() | (cur[1] != self.prev[1]).ravel()
                                | (cur[2] != self.prev[2]).ravel())
            for fi in ch:
                self.diffs.append((step - 1, int(fi), int(cur[0].ravel()[fi]), float(cur[1].ravel()[fi]),
                                   float(cur[2].ravel()[fi])))
        self.prev = cur


def gen_traj():
    cases = [
        ("traj_L8_n60", dict(L=8, n_steps=60, temp=2800, defect_fraction=0.05, n_seeds=3, impurity_c=0.1)),
        ("traj_L12_n40", dict(L=12, n_steps=40, temp=2800, defect_fraction=3e-3, n_seeds=5, impurity_c=0.1)),
        ("traj_L16_n100", dict(L=16, n_steps=100, temp=2800, defect_fraction=0.0, n_seeds=5, impurity_c=0.2)),
        ("traj_L10_n450", dict(L=10, n_steps=450, temp=2800, defect_fraction=0.01, n_seeds=6, impurity_c=0.2)),
        ("traj_L7_n230_T3400", dict(L=7, n_steps=230, temp=3400, defect_fraction=0.02, n_seeds=4, impurity_c=0.3)),
        ("traj_L30_n3", dict(L=30, n_steps=3, temp=2800, defect_fraction=3e-3, n_seeds=20, impurity_c=0.2)),
        ("traj_L32_n1", dict(L=32, n_steps=1, temp=2800, defect_fraction=0.0, n_seeds=5, impurity_c=0.0)),
        # the 9^3 lattice is full after 760 events: the run ends through the termination branch at the default temperature
        ("traj_L9_n2500", dict(L=9, n_steps=2500, temp=2800, defect_fraction=0.004, n_seeds=4, impurity_c=0.15)),
        # a long one: 2500 steps -- twelve defect refreshes, 14 metrics rows, 82 % of the lattice filled at the end
        ("traj_L14_n2500", dict(L=14, n_steps=2500, temp=2800, defect_fraction=0.004, n_seeds=5, impurity_c=0.15)),
        # substrate temperature within delta_T_c of the melting point: no nucleation anywhere; the small lattice fills up and
        # the run ends through the reference's "no valid events" branch (kmc_simulation.py:259-262) before n_steps
        ("traj_L3_n80_T3690_terminates", dict(L=3, n_steps=80, temp=3690, defect_fraction=0.0, n_seeds=2, impurity_c=0.1)),
        ("traj_L4_n200_T3688_terminates", dict(L=4, n_steps=200, temp=3688, defect_fraction=0.05, n_seeds=3, impurity_c=0.3)),
    ]
    only = os.environ.get("TRAJ_ONLY")            # regenerate one case: TRAJ_ONLY=traj_L3 python make_golden.py --only traj
    orig = ref_sim.get_event_rates
    for name, kw in cases:
        if only and not name.startswith(only):
            continue
        rec = Recorder(orig)
        ref_sim.get_event_rates = rec
        buf = io.StringIO()
        try:
            with contextlib.redirect_stdout(buf), np.errstate(all="ignore"):
                state, atom, total_time, theta, phi = ref_sim.run_kmc(output_prefix=name, **kw)
        finally:
            ref_sim.get_event_rates = orig
        rec.observe(state, theta, phi, len(rec.steps))
        py_state_after = random.getstate()
        np_after = np.random.get_state()
        csv = open(os.path.join("outputs", name, "metrics.csv")).read()
        steps = np.array(rec.steps, dtype=np.float64).reshape(-1, 3)
        d = rec.diffs
        tkeys = sorted(rec.T_snaps)
        dkeys = sorted(rec.defect_snaps)
        save(name,
             L=kw["L"], n_steps=kw["n_steps"], temp=kw["temp"], defect_fraction=kw["defect_fraction"],
             n_seeds=kw["n_seeds"], impurity_c=kw["impurity_c"],
             final_state=state.astype(np.int8), final_theta=theta, final_phi=phi,
             total_time=total_time,
             n_events=steps[:, 0].astype(np.int64), n_dep=steps[:, 1].astype(np.int64), seq_total=steps[:, 2],
             diff_step=np.array([x[0] for x in d], np.int32), diff_idx=np.array([x[1] for x in d], np.int32),
             diff_state=np.array([x[2] for x in d], np.int8), diff_theta=np.array([x[3] for x in d], np.float64),
             diff_phi=np.array([x[4] for x in d], np.float64),
             T_steps=np.array(tkeys, np.int32), T_snaps=np.stack([rec.T_snaps[k] for k in tkeys]),
             D_steps=np.array(dkeys, np.int32), D_snaps=np.stack([rec.defect_snaps[k] for k in dkeys]),
             py_next=np.array([random.random() for _ in range(4)]),
             np_next=np.random.random(4),
             metrics_csv=np.array(csv), stdout=np.array(buf.getvalue()),
             sha_state=np.array(sha(state)), sha_theta=np.array(sha(theta)), sha_phi=np.array(sha(phi)))
        random.setstate(py_state_after)
        np.random.set_state(np_after)


# --------------------------------------------------------------------------
# F3: thermal updates
# --------------------------------------------------------------------------
def gen_thermal():
    out = {}
    rs = np.random.RandomState(5)
    for L in (1, 2, 3, 7, 16):
        T = rs.uniform(2700.0, 4100.0, (L, L, L))
        st = np.zeros((L, L, L), dtype=np.int64)
        out[f"cet_rand_L{L}_in"] = T
        out[f"cet_rand_L{L}_out"] = ref_thermal.update_temperature_cet(T, st, dt=1e-6)
        out[f"cet_rand_L{L}_out_dt3e-7"] = ref_thermal.update_temperature_cet(T, st, dt=3e-7)
    # ramp, iterated (the reference's own unstable regime: dt*alpha/dx^2 = 2.7)
    L = 12
    _, _, _, T, _ = ref_init.initialize_lattice(lattice_size=L, n_seeds=3, T_sub=2800)
    seq = [T]
    for _ in range(6):
        seq.append(ref_thermal.update_temperature_cet(np.nan_to_num(seq[-1], nan=C.T_SUB), None, dt=1e-6))
    out["cet_ramp_L12_seq"] = np.stack(seq)
    # nan / inf scrub as done by run_kmc before each update (kmc_simulation.py:249)
    Tn = rs.uniform(2800.0, 3600.0, (6, 6, 6))
    Tn[1, 2, 3] = np.nan
    Tn[0, 0, 0] = np.inf
    Tn[5, 5, 5] = -np.inf
    out["cet_nan_L6_in"] = Tn
    with np.errstate(all="ignore"):
        out["cet_nan_L6_out"] = ref_thermal.update_temperature_cet(np.nan_to_num(Tn, nan=C.T_SUB), None, dt=1e-6)
    # laser variant
    for L, j0, i0 in ((8, 3, 1), (13, 6.5, 0), (16, 20, 4)):
        T = rs.uniform(2800.0, 3690.0, (L, L, L))
        prev = (rs.random_sample((L, L, L)) < 0.3).astype(np.int64) * rs.randint(1, 5, (L, L, L))
        cur = prev.copy()
        newm = (prev == 0) & (rs.random_sample((L, L, L)) < 0.1)
        cur[newm] = 1
        for dt in (1e-6, 1e-9):
            key = f"laser_L{L}_dt{dt:g}"
            out[key + "_T"] = T
            out[key + "_prev"] = prev.astype(np.int8)
            out[key + "_cur"] = cur.astype(np.int8)
            out[key + "_par"] = np.array([dt, i0, j0, 200.0, 50e-6, 0.35])
            out[key + "_out"] = ref_thermal.update_temperature(T, cur, prev, dt, (i0, j0), 200.0, 50e-6, 0.35)
        key = f"laser_L{L}_nolatent"
        out[key + "_out"] = ref_thermal.update_temperature(T, cur, cur, 1e-6, (i0, j0), 120.0, 30e-6, 0.5)
        out[key + "_par"] = np.array([1e-6, i0, j0, 120.0, 30e-6, 0.5])
    save("thermal", **out)


# --------------------------------------------------------------------------
# F4/F5: initialize_lattice and track_defects
# --------------------------------------------------------------------------
def gen_init_defects():
    out = {}
    for L, ns, c, tsub in ((8, 3, 0.1, 2800), (12, 5, 0.1, 2800), (16, 5, 0.2, 2800), (30, 20, 0.2, 2800),
                           (10, 6, 0.2, 2800), (7, 4, 0.3, 3400), (32, 5, 0.0, 2800), (5, 25, 0.5, 3000)):
        key = f"init_L{L}_s{ns}_c{c}_t{tsub}"
        state, theta, phi, T, atom = ref_init.initialize_lattice(lattice_size=L, n_seeds=ns, T_sub=tsub, impurity_c=c)
        out[key + "_state"] = state.astype(np.int8)
        out[key + "_theta"] = theta
        out[key + "_phi"] = phi
        out[key + "_T"] = T
        assert np.array_equal(state, atom)
        # defects right after init, same global stream (as run_kmc does, kmc_simulation.py:231)
        mask, dens = ref_defects.introduce_defects(state, atom, T, apply_to_state=False)
        out[key + "_defects"] = mask.astype(np.int8)
        out[key + "_density"] = np.float64(dens)
    # standalone defect masks on a lattice with many C sites, with and without T
    rs = np.random.RandomState(3)
    L = 9
    atom = rs.choice([0, 1, 2, 3, 4], size=(L, L, L), p=[0.3, 0.3, 0.1, 0.25, 0.05]).astype(np.int64)
    T = rs.uniform(-100.0, 4000.0, (L, L, L))
    out["defects_L9_atom"] = atom.astype(np.int8)
    out["defects_L9_T"] = T
    np.random.seed(77)
    out["defects_L9_mask_T"] = ref_defects.track_defects(atom, atom, L, T).astype(np.int8)
    np.random.seed(78)
    out["defects_L9_mask_noT"] = ref_defects.track_defects(atom, atom, L, None).astype(np.int8)
    np.random.seed(79)
    st2 = atom.copy()
    m, dens = ref_defects.introduce_defects(st2, atom, T, apply_to_state=True)
    out["defects_L9_applied_state"] = st2.astype(np.int8)
    out["defects_L9_applied_density"] = np.float64(dens)
    save("init_defects", **out)


# --------------------------------------------------------------------------
# F6: metrics dicts / clustering / helpers
# --------------------------------------------------------------------------
def _jsonable(d):
    o = {}
    for k, v in d.items():
        if isinstance(v, (np.floating, float)):
            o[k] = float(v)
        elif isinstance(v, (np.integer, int)):
            o[k] = int(v)
        elif v is None:
            o[k] = None
        else:
            o[k] = v
    return o


def gen_metrics():
    out = {}
    meta = {}
    rs = np.random.RandomState(9)
    for name, L, fill in (("m_L6", 6, 0.5), ("m_L9", 9, 0.25), ("m_L12", 12, 0.1), ("m_empty", 5, 0.0)):
        state = np.zeros((L, L, L), dtype=np.int64)
        occ = rs.random_sample((L, L, L)) < fill
        state[occ] = rs.choice([1, 2, 3, 4], size=int(occ.sum()), p=[0.6, 0.15, 0.2, 0.05])
        # few distinct orientations so that grains of >1 voxel exist
        palette_t = rs.uniform(0, np.pi, 4)
        palette_p = rs.uniform(0, 2 * np.pi, 4)
        pick = rs.randint(0, 4, (L, L, L))
        theta = np.where(state != 0, palette_t[pick], 0.0)
        phi = np.where(state != 0, palette_p[pick], 0.0)
        defects = ((state == 3) & (rs.random_sample((L, L, L)) < 0.3)).astype(np.int64)
        clusters, visited = ref_utils.get_clusters(state, theta, phi, theta_threshold=0.5)
        m = ref_metrics.compute_metrics(state, theta, phi, defects=defects, W_mask=(state == 1), Re_mask=(state == 2),
                                        C_mask=(state == 3), grain_ids=visited, rng_seed=7)
        m2 = ref_metrics.compute_metrics(state, theta, phi)
        out[name + "_state"] = state.astype(np.int8)
        out[name + "_theta"] = theta
        out[name + "_phi"] = phi
        out[name + "_defects"] = defects.astype(np.int8)
        out[name + "_visited"] = np.asarray(visited, dtype=np.int32)
        out[name + "_cluster_sizes"] = np.array([len(c) for c in clusters], np.int32)
        out[name + "_cluster_first"] = np.array([c[0] for c in clusters], np.int32).reshape(-1, 3)
        out[name + "_cluster_ar"] = np.array([ref_utils.calculate_aspect_ratio(c) for c in clusters], np.float64)
        meta[name] = dict(full=_jsonable(m), default=_jsonable(m2),
                          cet=ref_metrics.compute_CET(state, theta, phi),
                          detect=bool(ref_metrics.detect_CET_transition(m)) )
    # run_simulation.py behaviour (reference smoke script): 17 keys
    state = np.zeros((10, 10, 10), dtype=int)
    state[2:6, 2:6, 2:6] = 1
    state[6:9, 6:9, 6:9] = 2
    theta = np.zeros_like(state, dtype=float)
    phi = np.zeros_like(state, dtype=float)
    W = np.zeros_like(state, dtype=bool); Re = W.copy(); Cm = W.copy()
    W[3, 3, 3] = True; Re[7, 7, 7] = True; Cm[5, 5, 5] = True
    m = ref_metrics.compute_metrics(state, theta, phi, defects=None, W_mask=W, Re_mask=Re, C_mask=Cm,
                                    grain_ids=state.copy(), rng_seed=42)
    meta["run_simulation"] = dict(keys=list(m.keys()), values=_jsonable(m))
    # misorientation / neighbour known answers
    ang = rs.uniform(0, 2 * np.pi, (64, 4))
    ang[:4] = [[0, 0, 0, 0], [0, 0, np.pi, 0], [1.0, 2.0, 1.0, 2.0], [0.3, 0.1, 0.3000001, 0.1]]
    out["misor_in"] = ang
    out["misor_out"] = np.array([ref_rates.compute_misorientation(*row) for row in ang])
    nb = {}
    for (i, j, k, L) in ((0, 0, 0, 5), (2, 2, 2, 5), (4, 0, 3, 5), (1, 1, 1, 3), (0, 0, 0, 1), (7, 3, 0, 8)):
        nb[f"{i},{j},{k},{L}"] = ref_rates.get_bcc_neighbors(i, j, k, L).tolist()
    meta["neighbors"] = nb
    save("metrics", **out)
    with open(os.path.join(HERE, "metrics_meta.json"), "w") as f:
        json.dump(meta, f, indent=1, sort_keys=True)
    print("  wrote metrics_meta.json")


GENS = dict(events=gen_events, events_odd=gen_events_odd, events_init=gen_events_initlattice, traj=gen_traj, thermal=gen_thermal,
            init_defects=gen_init_defects, metrics=gen_metrics)

if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", default=None)
    a = ap.parse_args()
    for k, g in GENS.items():
        if a.only and a.only != k:
            continue
        print(f"[{k}]")
        g()
    json.dump(dict(python=sys.version.split()[0], numpy=np.__version__,
                   scipy=__import__("scipy").__version__, pandas=__import__("pandas").__version__,
                   reference_snapshot="2025-12-05", mode="identity-jit (numba absent)"),
              open(os.path.join(HERE, "PROVENANCE.json"), "w"), indent=1)
