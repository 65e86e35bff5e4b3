"""ctypes front-end of oracle/cet_oracle.c -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this
module; the product package never does.  Parity status: pinned by tests/golden/*.npz
(generated from the reference by tests/golden/make_golden.py).

The constants below restate the reference's constants.py (values are part of the parity
contract); the oracle deliberately does not import the product's constants module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libcet_oracle.so")

# ---- constants.py (reference) ---------------------------------------------------
K_T = 8.617333262e-5       # constants.py:63
T_MELT = 3695              # :64
T_SUB = 2800               # :65
NU = 1e13                  # :70
NU_DEP = 2e13              # :71
E_B = (3.8, 4.2, 3.2)      # :74,80,84
E_DIFF = (0.35, 0.50, 0.30)  # :75,81,85
IMPURITY_RE = 0.10         # :82
MAX_IMP_FRACTION = 1.0     # :90
ANISOTROPY_FACTOR = 0.25   # :102
DELTA_T_C = 10             # :123
I0 = 5e13                  # :129
K_NUC = 500                # :130
BETA_IMP_NUC = 0.4         # :131
RATE_THRESHOLD = 1e-30     # :145
VOXEL_SIZE = 5e-6          # :54
# thermal_solver.py:6-9
K_COND, RHO, CP = 173.0, 19300.0, 132.0
ALPHA = K_COND / (RHO * CP)


class Params(C.Structure):
    _fields_ = [
        ("nu", C.c_double), ("nu_dep", C.c_double), ("E_b", C.c_double * 3), ("E_diff", C.c_double * 3),
        ("kT", C.c_double), ("T_melt", C.c_double), ("I0", C.c_double), ("delta_T_c", C.c_double),
        ("K_nuc", C.c_double), ("beta_imp_nuc", C.c_double), ("max_imp_frac", C.c_double),
        ("rate_threshold", C.c_double), ("anisotropy", C.c_double), ("impurity_re", C.c_double),
        ("impurity_c", C.c_double),
        ("alpha", C.c_double), ("inv_dx2", C.c_double), ("T_clip_lo", C.c_double), ("T_clip_hi", C.c_double),
        ("T_nan", C.c_double), ("rho_cp", C.c_double), ("latent_coef", C.c_double),
    ]


class Event(C.Structure):
    _fields_ = [("type", C.c_int32), ("pos", C.c_int32 * 3), ("target", C.c_int32 * 3), ("atom", C.c_int32),
                ("rate", C.c_double), ("dep_rank", C.c_int64)]

    def astuple(self):
        return (self.type, tuple(self.pos), self.rate, tuple(self.target), self.atom, self.dep_rank)


EVENT_DTYPE = np.dtype([("type", "<i4"), ("pos", "<i4", 3), ("target", "<i4", 3), ("atom", "<i4"),
                        ("rate", "<f8"), ("dep_rank", "<i8")], align=True)


def make_params(impurity_c=0.0):
    p = Params()
    p.nu, p.nu_dep = NU, NU_DEP
    p.E_b[:] = E_B
    p.E_diff[:] = E_DIFF
    p.kT, p.T_melt, p.I0, p.delta_T_c = K_T, float(T_MELT), I0, float(DELTA_T_C)
    p.K_nuc, p.beta_imp_nuc, p.max_imp_frac = float(K_NUC), BETA_IMP_NUC, MAX_IMP_FRACTION
    p.rate_threshold, p.anisotropy, p.impurity_re = RATE_THRESHOLD, ANISOTROPY_FACTOR, IMPURITY_RE
    p.impurity_c = float(impurity_c)
    p.alpha = ALPHA
    p.inv_dx2 = 1.0 / (VOXEL_SIZE * VOXEL_SIZE)
    p.T_clip_lo, p.T_clip_hi, p.T_nan = float(T_SUB), T_MELT * 1.1, float(T_SUB)
    p.rho_cp = RHO * CP
    p.latent_coef = 200e3 / CP
    return p


def build(force=False):
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(os.path.join(_HERE, "cet_oracle.c")):
        subprocess.check_call(["make", "-s", "-C", _HERE])
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        L = C.CDLL(_SO)
        assert L.orc_sizeof_params() == C.sizeof(Params)
        assert L.orc_sizeof_event() == C.sizeof(Event) == EVENT_DTYPE.itemsize
        L.orc_misorientation.restype = C.c_double
        L.orc_misorientation.argtypes = [C.c_double] * 4
        L.orc_enumerate.restype = C.c_int64
        L.orc_select_sequential.restype = C.c_int64
        L.orc_total.restype = C.c_double
        L.orc_run_steps.restype = C.c_int64
        L.orc_run_supersteps.restype = C.c_int64
        L.orc_set_threads.restype = C.c_int
        L.orc_set_threads(int(os.environ.get("CET_ORACLE_THREADS", "1")))   # scalar port unless asked otherwise
        L.orc_counter_uniform.restype = C.c_double
        L.orc_counter_uniform.argtypes = [C.c_uint64] * 3
        _lib = L
    return _lib


def _p(a, t):
    return a.ctypes.data_as(C.POINTER(t)) if a is not None else None


def _i8(a):
    return np.ascontiguousarray(a, dtype=np.int8)


def _f8(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def neighbors(i, j, k, L):
    out = (C.c_int32 * 3 * 14)()
    n = lib().orc_neighbors(int(i), int(j), int(k), int(L), out)
    return np.array([[out[m][0], out[m][1], out[m][2]] for m in range(n)], dtype=np.int64).reshape(n, 3)


def misorientation(t1, p1, t2, p2):
    return lib().orc_misorientation(t1, p1, t2, p2)


class Lattice:
    """Holds int8/f64 copies of the six lattice fields in reference layout."""

    def __init__(self, state, theta, phi, T, defects=None, impurity_c=0.0):
        self.L = int(state.shape[0])
        self.state = _i8(state).copy()
        self.theta = _f8(theta).copy()
        self.phi = _f8(phi).copy()
        self.T = _f8(T).copy()
        self.defects = _i8(defects).copy() if defects is not None else np.zeros_like(self.state)
        self.prev_state = self.state.copy()
        self.params = make_params(impurity_c)
        self.nuc_count = 0

    def _fields(self):
        return (_p(self.state, C.c_int8), _p(self.theta, C.c_double), _p(self.phi, C.c_double),
                _p(self.T, C.c_double), _p(self.defects, C.c_int8))

    # kmc_event_rates.py:162 get_event_rates
    def enumerate(self, u_dep=None):
        L = self.L
        cap = 16 * L ** 3 + 16
        ev = np.zeros(cap, dtype=EVENT_DTYPE)
        nd = C.c_int64(0)
        u = _f8(u_dep) if u_dep is not None else None
        n = lib().orc_enumerate(C.byref(self.params), L, *self._fields(), _p(u, C.c_double),
                                C.c_int64(len(u) if u is not None else 0), ev.ctypes.data_as(C.c_void_p),
                                C.c_int64(cap), C.byref(nd))
        return ev[:n].copy(), nd.value

    @staticmethod
    def select_sequential(ev, u):
        tot = C.c_double(0.0)
        evc = np.ascontiguousarray(ev)
        idx = lib().orc_select_sequential(evc.ctypes.data_as(C.c_void_p), C.c_int64(len(evc)), C.c_double(u),
                                          C.byref(tot))
        return idx, tot.value

    def row_sums(self, i0=0, i1=None, rowsum=None, rowcnt=None):
        L = self.L
        i1 = L if i1 is None else i1
        if rowsum is None:
            rowsum = np.zeros((L, 3, L), np.float64)
            rowcnt = np.zeros((L, 3, L), np.int32)
        lib().orc_row_sums(C.byref(self.params), L, *self._fields(), int(i0), int(i1), _p(rowsum, C.c_double),
                           _p(rowcnt, C.c_int32))
        return rowsum, rowcnt

    def block_sums(self, rowsum, rowcnt, i0=0, i1=None, blocksum=None, blockcnt=None):
        L = self.L
        i1 = L if i1 is None else i1
        if blocksum is None:
            blocksum = np.zeros(3 * L, np.float64)
            blockcnt = np.zeros(3 * L, np.int64)
        lib().orc_block_sums(L, _p(rowsum, C.c_double), _p(rowcnt, C.c_int32), int(i0), int(i1),
                             _p(blocksum, C.c_double), _p(blockcnt, C.c_int64))
        return blocksum, blockcnt

    def total(self, blocksum, blockcnt):
        ne, nd = C.c_int64(0), C.c_int64(0)
        t = lib().orc_total(self.L, _p(blocksum, C.c_double), _p(blockcnt, C.c_int64), C.byref(ne), C.byref(nd))
        return t, ne.value, nd.value

    def select_tree(self, blocksum, blockcnt, rowsum, rowcnt, r):
        ev = Event()
        rc = lib().orc_select_tree(C.byref(self.params), self.L, *self._fields(), _p(blocksum, C.c_double),
                                   _p(blockcnt, C.c_int64), _p(rowsum, C.c_double), _p(rowcnt, C.c_int32),
                                   C.c_double(r), C.byref(ev))
        return None if rc else ev

    def window_pick(self, i0, j0, k0, H, u, with_total=False):
        """Mode B: the pick of ONE box (window origin, edge H, uniform u) on this lattice copy; None for an idle box.
        with_total: (event or None, window total R_d) -- what the null-event acceptance compares with R_max."""
        ev = Event()
        R = C.c_double(0.0)
        idle = lib().orc_window_pick_r(C.byref(self.params), self.L, *self._fields(), int(i0), int(j0), int(k0), int(H),
                                       C.c_double(u), C.byref(ev), C.byref(R))
        if with_total:
            return (None if idle else ev), R.value
        return None if idle else ev

    def sweep(self):
        rs, rc = self.row_sums()
        bs, bc = self.block_sums(rs, rc)
        t, ne, nd = self.total(bs, bc)
        return dict(rowsum=rs, rowcnt=rc, blocksum=bs, blockcnt=bc, total=t, n_events=ne, n_dep=nd)

    def apply(self, ev, theta_new=0.0, phi_new=0.0, make_defect=False):
        nuc = lib().orc_apply(self.L, _p(self.state, C.c_int8), _p(self.theta, C.c_double), _p(self.phi, C.c_double),
                              C.byref(ev), C.c_double(theta_new), C.c_double(phi_new), int(bool(make_defect)))
        self.nuc_count += nuc
        return nuc

    # thermal_solver.py:107
    def thermal_cet(self, dt=1e-6, scrub_nan=True):
        out = np.empty_like(self.T)
        lib().orc_thermal_cet(C.byref(self.params), self.L, _p(self.T, C.c_double), C.c_double(dt), int(scrub_nan),
                              _p(out, C.c_double))
        self.T = out
        return out

    # thermal_solver.py:36
    def thermal_laser(self, dt, q_top, prev_state=None, scrub_nan=False, update_prev=True):
        out = np.empty_like(self.T)
        prev = _i8(prev_state) if prev_state is not None else self.prev_state
        q = _f8(q_top)
        lib().orc_thermal_laser(C.byref(self.params), self.L, _p(self.T, C.c_double), _p(self.state, C.c_int8),
                                _p(prev, C.c_int8), C.c_double(dt), _p(q, C.c_double), int(scrub_nan),
                                _p(out, C.c_double))
        self.T = out
        if update_prev:
            self.prev_state = self.state.copy()
        return out

    def run_steps(self, step0, n, defect_fraction, u_pick, u_defect, u_np, rng_mode=0, seed=0,
                  thermal_mode=1, thermal_dt=1e-6, q_planes=None):
        L = self.L
        u_pick = _f8(u_pick)
        u_defect = _f8(u_defect) if u_defect is not None else np.zeros(n)
        u_np = _f8(u_np)
        totals = np.zeros(n, np.float64)
        events = np.zeros(n, dtype=EVENT_DTYPE)
        nev = np.zeros(n, np.int64)
        np_used, q_used, nuc, status = C.c_int64(0), C.c_int64(0), C.c_int64(self.nuc_count), C.c_int(0)
        q = _f8(q_planes) if q_planes is not None else None
        done = lib().orc_run_steps(
            C.byref(self.params), L, _p(self.state, C.c_int8), _p(self.theta, C.c_double), _p(self.phi, C.c_double),
            _p(self.T, C.c_double), _p(self.defects, C.c_int8), _p(self.prev_state, C.c_int8),
            C.c_int64(step0), C.c_int64(n), C.c_double(defect_fraction), _p(u_pick, C.c_double),
            _p(u_defect, C.c_double), _p(u_np, C.c_double), C.c_int64(len(u_np)), C.byref(np_used),
            int(rng_mode), C.c_uint64(seed), int(thermal_mode), C.c_double(thermal_dt), _p(q, C.c_double),
            C.byref(q_used), _p(totals, C.c_double), events.ctypes.data_as(C.c_void_p), _p(nev, C.c_int64),
            C.byref(nuc), C.byref(status))
        self.nuc_count = nuc.value
        return dict(done=int(done), status=status.value, np_used=np_used.value, q_used=q_used.value,
                    totals=totals[:max(done, 0) + (1 if status.value == 1 else 0)], events=events[:done],
                    n_events=nev[:done])


    # Mode B (not in the reference): synchronous super-steps over nb^3 boxes, see cet_oracle.c
    def run_supersteps(self, step0, n, box, defect_fraction, seed, thermal_mode=1, thermal_dt=1e-6, q_planes=None,
                       want_events=True, null_events=False):
        L = self.L
        D = (L // box) ** 3
        totals = np.zeros(n, np.float64)
        dt_event = np.zeros(n, np.float64)
        events = np.zeros((n, D), dtype=EVENT_DTYPE) if want_events else None
        n_exec = np.zeros(n, np.int64)
        q_used, nuc, status = C.c_int64(0), C.c_int64(self.nuc_count), C.c_int(0)
        q = _f8(q_planes) if q_planes is not None else None
        done = lib().orc_run_supersteps(
            C.byref(self.params), L, _p(self.state, C.c_int8), _p(self.theta, C.c_double), _p(self.phi, C.c_double),
            _p(self.T, C.c_double), _p(self.defects, C.c_int8), _p(self.prev_state, C.c_int8),
            C.c_int64(step0), C.c_int64(n), int(box), C.c_double(defect_fraction), C.c_uint64(seed),
            int(thermal_mode), C.c_double(thermal_dt), _p(q, C.c_double), C.byref(q_used), _p(totals, C.c_double),
            events.ctypes.data_as(C.c_void_p) if want_events else None, _p(n_exec, C.c_int64), C.byref(nuc),
            C.byref(status), int(bool(null_events)), _p(dt_event, C.c_double))
        self.nuc_count = nuc.value
        return dict(done=int(done), status=status.value, q_used=q_used.value,
                    totals=totals[:max(done, 0) + (1 if status.value == 1 else 0)],
                    events=None if events is None else events[:done], n_exec=n_exec[:done], dt_event=dt_event[:done])


def laser_source_plane(L, laser_pos, laser_power, beam_radius=50e-6, absorptivity=0.35):
    """thermal_solver.py:82-95: volumetric source of plane i=L-1 (I_surface / VOXEL_SIZE).
    Note the reference uses j0 for BOTH in-plane axes and ignores i0 (:86)."""
    i0, j0 = laser_pos
    jj = np.arange(L, dtype=np.float64)
    kk = np.arange(L, dtype=np.float64)
    JJ, KK = np.meshgrid(jj, kk, indexing="ij")
    r_m = np.sqrt((JJ - j0) ** 2 + (KK - j0) ** 2) * VOXEL_SIZE
    area_norm = np.pi * beam_radius * beam_radius
    I_surface = (laser_power * absorptivity / area_norm) * np.exp(-(r_m ** 2) / (beam_radius ** 2))
    return I_surface / VOXEL_SIZE


KEY_PICK, KEY_THETA, KEY_PHI, KEY_DEFECT, KEY_ACCEPT, KEY_DT = (q << 40 for q in range(1, 7))    # Mode B uniform keys


def set_threads(n):
    """Worker threads of the row-sum / thermal loops (results do not depend on it); returns the count in effect."""
    return lib().orc_set_threads(int(n))


def counter_uniform(seed, step, site):
    return lib().orc_counter_uniform(int(seed), int(step), int(site))
