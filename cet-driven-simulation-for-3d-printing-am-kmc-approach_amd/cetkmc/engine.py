"""Engine: one lattice resident on the GPU, driven through the C ABI (include/cetkmc.h)."""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import Counters, Event, Params, RunArgs, RunResult, SuperArgs, SweepInfo, build_library  # noqa: F401

EVENT_DTYPE = np.dtype([("type", "<i4"), ("pos", "<i4", 3), ("target", "<i4", 3), ("atom", "<i4"),
                        ("rate", "<f8"), ("dep_rank", "<i8"), ("theta", "<f8"), ("phi", "<f8")], align=True)
assert EVENT_DTYPE.itemsize == C.sizeof(Event) == 64

TYPE_BYTES = (b"dep", b"diff", b"nuc", b"att")   # kmc_event_rates.py:72,109,132,158


def library_path():
    return _lib.SO_PATH


def default_params(impurity_c=0.0):
    """cetkmc_params filled from constants.py / thermal_solver.py of this package (same
    expressions as the reference so derived values round identically)."""
    import constants as K
    p = Params()
    p.nu, p.nu_dep = K.NU, K.NU_DEP
    p.E_b[:] = (K.E_B_W, K.E_B_RE, K.E_B_C)
    p.E_diff[:] = (K.E_DIFF_W, K.E_DIFF_RE, K.E_DIFF_C)
    p.kT, p.T_melt, p.I0, p.delta_T_c = K.K_T, float(K.T_MELT), K.I0, float(K.DELTA_T_C)
    p.K_nuc, p.beta_imp_nuc, p.max_imp_frac = float(K.K_NUC), K.BETA_IMP_NUC, K.MAX_IMP_FRACTION
    p.rate_threshold, p.anisotropy, p.impurity_re = K.RATE_THRESHOLD, K.ANISOTROPY_FACTOR, K.IMPURITY_RE
    p.impurity_c = float(impurity_c)
    k_cond, rho, cp = 173.0, 19300.0, 132.0            # thermal_solver.py:6-8
    p.alpha = k_cond / (rho * cp)
    p.inv_dx2 = 1.0 / (K.VOXEL_SIZE * K.VOXEL_SIZE)
    p.T_clip_lo, p.T_clip_hi, p.T_nan = float(K.T_SUB), K.T_MELT * 1.1, float(K.T_SUB)
    p.rho_cp = rho * cp
    p.latent_coef = 200e3 / cp
    return p


def _ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _dptr(a):
    return None if a is None else a.ctypes.data_as(C.POINTER(C.c_double))


def device_count():
    """Number of HIP devices this process can see (0 when there is none)."""
    n = C.c_int(0)
    if _lib.load().cetkmc_device_count(C.byref(n)):
        return 0
    return n.value


class Engine:
    """Owns a cetkmc handle.  ``n_slabs>1`` splits the lattice into axis-0 slabs on the same
    GPU (decomposition check); ``rank/nranks/unique_id`` selects the one-process-per-GPU
    RCCL mode."""

    def __init__(self, L, impurity_c=0.0, params=None, n_slabs=1, device=0, rank=None, nranks=None, unique_id=None,
                 host_comm=None):
        self.lib = _lib.load()
        self.L = int(L)
        self.params = params if params is not None else default_params(impurity_c)
        self.h = C.c_void_p()
        n = C.c_int(0)
        if self.lib.cetkmc_device_count(C.byref(n)) or n.value <= 0:
            raise RuntimeError("cetkmc: no usable HIP device (there is no CPU fallback): " + self.error())
        if rank is None:
            devs = (C.c_int * n_slabs)(*([device] * n_slabs))
            rc = self.lib.cetkmc_create(C.byref(self.params), self.L, n_slabs, devs, C.byref(self.h))
        elif host_comm is not None:
            # bring-up / test transport: collectives relayed through Python callables (see host_transport.py)
            allgather, exchange = host_comm
            self._hc = _lib.HostComm(_lib.ALLGATHER_FN(allgather), _lib.EXCHANGE_FN(exchange), None)     # keep alive
            rc = self.lib.cetkmc_create_rank_host(C.byref(self.params), self.L, int(rank), int(nranks), int(device),
                                                  C.byref(self._hc), C.byref(self.h))
        else:
            rc = self.lib.cetkmc_create_rank(C.byref(self.params), self.L, int(rank), int(nranks), int(device),
                                             unique_id, C.byref(self.h))
        if rc:
            raise RuntimeError("cetkmc_create: " + self.error())
        i0, i1 = C.c_int(0), C.c_int(0)
        self._ck(self.lib.cetkmc_owned_planes(self.h, C.byref(i0), C.byref(i1)))
        self.i0, self.i1 = i0.value, i1.value

    # -- plumbing ---------------------------------------------------------------------
    def error(self):
        return self.lib.cetkmc_last_error().decode(errors="replace")

    def _ck(self, rc):
        if rc:
            raise RuntimeError("cetkmc: " + self.error())

    def close(self):
        if getattr(self, "h", None) is not None and self.h:
            self.lib.cetkmc_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @staticmethod
    def unique_id():
        buf = C.create_string_buffer(128)
        lib = _lib.load()
        if lib.cetkmc_get_unique_id(buf):
            raise RuntimeError("cetkmc_get_unique_id: " + lib.cetkmc_last_error().decode())
        return buf.raw

    def set_impurity_c(self, c):
        self.params.impurity_c = float(c)
        self._ck(self.lib.cetkmc_set_params(self.h, C.byref(self.params)))

    # -- transfers --------------------------------------------------------------------
    def upload(self, state=None, theta=None, phi=None, T=None, defects=None):
        L = self.L

        def i64(a):
            if a is None:
                return None
            a = np.ascontiguousarray(a, dtype=np.int64)
            assert a.shape == (L, L, L)
            return a

        def f64(a):
            if a is None:
                return None
            a = np.ascontiguousarray(a, dtype=np.float64)
            assert a.shape == (L, L, L)
            return a

        s, th, ph, t, d = i64(state), f64(theta), f64(phi), f64(T), i64(defects)
        self._ck(self.lib.cetkmc_upload(self.h, _ptr(s), _ptr(th), _ptr(ph), _ptr(t), _ptr(d)))

    def download(self, state=True, theta=True, phi=True, T=True, defects=False):
        L = self.L
        out = {}
        if state:
            out["state"] = np.zeros((L, L, L), np.int64)
        if theta:
            out["theta"] = np.zeros((L, L, L), np.float64)
        if phi:
            out["phi"] = np.zeros((L, L, L), np.float64)
        if T:
            out["T"] = np.zeros((L, L, L), np.float64)
        if defects:
            out["defects"] = np.zeros((L, L, L), np.int64)
        self._ck(self.lib.cetkmc_download(self.h, _ptr(out.get("state")), _ptr(out.get("theta")), _ptr(out.get("phi")),
                                          _ptr(out.get("T")), _ptr(out.get("defects"))))
        return out

    def upload_planes(self, i_begin, i_end, state=None, theta=None, phi=None, T=None, defects=None):
        n, L = i_end - i_begin, self.L

        def u8(a):
            if a is None:
                return None
            a = np.ascontiguousarray(a, dtype=np.uint8)
            assert a.shape == (n, L, L)
            return a

        def f64(a):
            if a is None:
                return None
            a = np.ascontiguousarray(a, dtype=np.float64)
            assert a.shape == (n, L, L)
            return a

        s, th, ph, t, d = u8(state), f64(theta), f64(phi), f64(T), u8(defects)
        self._ck(self.lib.cetkmc_upload_planes(self.h, i_begin, i_end, _ptr(s), _ptr(th), _ptr(ph), _ptr(t), _ptr(d)))

    def download_planes(self, i_begin, i_end, state=True, theta=False, phi=False, T=False, defects=False):
        n, L = i_end - i_begin, self.L
        out = {}
        if state:
            out["state"] = np.zeros((n, L, L), np.uint8)
        if theta:
            out["theta"] = np.zeros((n, L, L), np.float64)
        if phi:
            out["phi"] = np.zeros((n, L, L), np.float64)
        if T:
            out["T"] = np.zeros((n, L, L), np.float64)
        if defects:
            out["defects"] = np.zeros((n, L, L), np.uint8)
        self._ck(self.lib.cetkmc_download_planes(self.h, i_begin, i_end, _ptr(out.get("state")), _ptr(out.get("theta")),
                                                 _ptr(out.get("phi")), _ptr(out.get("T")), _ptr(out.get("defects"))))
        return out

    def set_defects(self, mask):
        m = np.ascontiguousarray(mask, dtype=np.uint8)
        assert m.shape == (self.L,) * 3
        self._ck(self.lib.cetkmc_set_defects(self.h, _ptr(m)))

    def set_prev_state(self, prev=None):
        p = None if prev is None else np.ascontiguousarray(prev, dtype=np.int64)
        self._ck(self.lib.cetkmc_set_prev_state(self.h, _ptr(p)))

    # -- thermal ------------------------------------------------------------------------
    def thermal_cet(self, dt=1e-6, scrub_nan=True):
        self._ck(self.lib.cetkmc_thermal_cet(self.h, float(dt), int(bool(scrub_nan))))

    def thermal_laser(self, dt, q_top, use_latent=True, scrub_nan=False):
        q = np.ascontiguousarray(q_top, dtype=np.float64)
        assert q.shape == (self.L, self.L)
        self._ck(self.lib.cetkmc_thermal_laser(self.h, float(dt), _ptr(q), int(bool(use_latent)), int(bool(scrub_nan))))

    # -- single-step primitives -----------------------------------------------------------
    def rate_sweep(self):
        info = SweepInfo()
        self._ck(self.lib.cetkmc_rate_sweep(self.h, C.byref(info)))
        return info.total, info.n_events, info.n_dep

    def select(self, r):
        ev = Event()
        self._ck(self.lib.cetkmc_select(self.h, float(r), C.byref(ev)))
        return ev

    def apply(self, ev, theta_new=0.0, phi_new=0.0, make_defect=False):
        self._ck(self.lib.cetkmc_apply(self.h, C.byref(ev), float(theta_new), float(phi_new), int(bool(make_defect))))

    def enumerate_events(self, cap=None):
        n = C.c_int64(0)
        self._ck(self.lib.cetkmc_enumerate_events(self.h, None, 0, C.byref(n)))
        cap = n.value if cap is None else min(cap, n.value)
        buf = np.zeros(max(cap, 1), dtype=EVENT_DTYPE)
        if cap:
            self._ck(self.lib.cetkmc_enumerate_events(self.h, _ptr(buf), cap, C.byref(n)))
        return buf[:cap], n.value

    def row_sums(self):
        L = self.L
        rs = np.zeros((L, 3, L), np.float64)
        rc = np.zeros((L, 3, L), np.int32)
        self._ck(self.lib.cetkmc_row_sums(self.h, _ptr(rs), _ptr(rc)))
        return rs, rc

    # -- batched loop ------------------------------------------------------------------------
    def _run_args(self, step0, n, defect_fraction, u_pick, u_defect, u_np, rng_mode, seed, thermal_mode, thermal_dt, q_planes,
                  use_latent, profile, incremental):
        a = RunArgs()
        u_pick = None if u_pick is None else np.ascontiguousarray(u_pick, dtype=np.float64)     # rng_mode 2: no host streams
        u_defect = None if u_defect is None else np.ascontiguousarray(u_defect, dtype=np.float64)
        u_np = np.zeros(0) if u_np is None else np.ascontiguousarray(u_np, dtype=np.float64)
        q = None if q_planes is None else np.ascontiguousarray(q_planes, dtype=np.float64)
        a.step0, a.n_steps, a.defect_fraction = int(step0), int(n), float(defect_fraction)
        a.u_pick, a.u_defect, a.u_np = _dptr(u_pick), _dptr(u_defect), _dptr(u_np)
        a.np_cap, a.rng_mode, a.seed = len(u_np), int(rng_mode), int(seed)
        a.thermal_mode, a.thermal_dt = int(thermal_mode), float(thermal_dt)
        a.q_planes, a.n_q = _dptr(q), (0 if q is None else q.shape[0])
        a.use_latent, a.profile = int(bool(use_latent)), int(profile)       # profile: False/True/2 (per-phase, see counters())
        a.incremental = int(bool(incremental))
        return a, (u_pick, u_defect, u_np, q)       # (the arrays stay referenced while the pointers are in use)

    def stage_inputs(self, step0, n, defect_fraction, u_pick, u_defect, u_np, rng_mode=0, seed=0, thermal_mode=1,
                     thermal_dt=1e-6, q_planes=None, use_latent=True, profile=False, incremental=False):
        """cetkmc_stage_inputs: the batch's random streams and laser source planes go to the device NOW; the matching
        run_steps(..., staged=True) call copies nothing."""
        a, keep = self._run_args(step0, n, defect_fraction, u_pick, u_defect, u_np, rng_mode, seed, thermal_mode, thermal_dt,
                                 q_planes, use_latent, profile, incremental)
        self._ck(self.lib.cetkmc_stage_inputs(self.h, C.byref(a)))

    def run_steps(self, step0, n, defect_fraction, u_pick, u_defect, u_np, rng_mode=0, seed=0, thermal_mode=1,
                  thermal_dt=1e-6, q_planes=None, use_latent=True, profile=False, want_logs=True, incremental=False, staged=False):
        a, keep = self._run_args(step0, n, defect_fraction, u_pick, u_defect, u_np, rng_mode, seed, thermal_mode, thermal_dt,
                                 q_planes, use_latent, profile, incremental)
        if staged:       # same arguments as the stage_inputs call; the library checks the batch shape and copies nothing
            a.u_pick = a.u_defect = a.u_np = a.q_planes = None
        res = RunResult()
        totals = np.zeros(n + 1, np.float64) if want_logs else None
        events = np.zeros(max(n, 1), dtype=EVENT_DTYPE) if want_logs else None
        nev = np.zeros(max(n, 1), np.int64) if want_logs else None
        self._ck(self.lib.cetkmc_run_steps(self.h, C.byref(a), C.byref(res), _ptr(totals), _ptr(events), _ptr(nev)))
        done = res.steps_done
        out = dict(done=int(done), status=int(res.status), np_used=int(res.np_used), q_used=int(res.q_used),
                   nucleation_count=int(res.nucleation_count), sweep_ms_total=res.sweep_ms_total,
                   sweep_launches=int(res.sweep_launches), wall_ms=res.wall_ms, full_sweeps=int(res.full_sweeps),
                   min_margin=float(res.min_margin))
        if want_logs:
            out.update(totals=totals[:done + (1 if res.status == 1 else 0)], events=events[:done], n_events=nev[:done])
        return out

    def counters(self, reset=False):
        """cetkmc_get_counters as a dict: work issued, bytes moved, per-phase device ms of the profile=2 runs."""
        c = Counters()
        self._ck(self.lib.cetkmc_get_counters(self.h, C.byref(c), int(bool(reset))))
        return {n: getattr(c, n) for n, _ in Counters._fields_}

    # -- Mode B: synchronous super-steps over (L/box)^3 boxes (not in the reference; include/cetkmc.h) --------
    def run_supersteps(self, step0, n, box, defect_fraction, seed, thermal_mode=1, thermal_dt=1e-6, q_planes=None,
                       use_latent=True, want_events=False, null_events=False):
        """box == L: the single-domain case (one Mode A step per super-step, counter uniforms of box 0).
        Returns also ``dt_event`` (time increment per executed event of every super-step, kmc_simulation.py:331-332
        restated; identical on every rank) -- simulated time advances by ``n_exec[g] * dt_event[g]`` with n_exec summed
        over ranks."""
        a = SuperArgs()
        q = None if q_planes is None else np.ascontiguousarray(q_planes, dtype=np.float64)
        a.step0, a.n_steps, a.box, a.defect_fraction, a.seed = int(step0), int(n), int(box), float(defect_fraction), int(seed)
        a.thermal_mode, a.thermal_dt = int(thermal_mode), float(thermal_dt)
        a.q_planes, a.n_q, a.use_latent = _dptr(q), (0 if q is None else q.shape[0]), int(bool(use_latent))
        a.null_events = int(bool(null_events))
        # boxes of this handle: the box layers of its owned planes (across ranks every rank runs its own boxes and logs
        # their events / executed counts; global box order = rank order)
        nbx = self.L // int(box) if box and self.L % int(box) == 0 else 1
        D = max(1, (self.i1 - self.i0) // int(box)) * nbx * nbx if box and self.L % int(box) == 0 else 1
        res = RunResult()
        totals = np.zeros(n + 1, np.float64)
        n_exec = np.zeros(max(n, 1), np.int64)
        dt_event = np.zeros(max(n, 1), np.float64)
        events = np.zeros((max(n, 1), D), dtype=EVENT_DTYPE) if want_events else None
        self._ck(self.lib.cetkmc_run_supersteps(self.h, C.byref(a), C.byref(res), _ptr(totals), _ptr(events), _ptr(n_exec),
                                                _ptr(dt_event)))
        done = int(res.steps_done)
        return dict(done=done, status=int(res.status), q_used=int(res.q_used), nucleation_count=int(res.nucleation_count),
                    wall_ms=res.wall_ms, totals=totals[:done + (1 if res.status == 1 else 0)], n_exec=n_exec[:done],
                    events=None if events is None else events[:done], domains=D, dt_event=dt_event[:done])

    # -- grain clustering (utils.get_clusters on the device) --------------------------------------
    def clusters(self, threshold=0.5, labels=False):
        """dict(first (n,3) int32, size (n,) int64, bbox (n,6) int32[, labels (L,L,L) int32]); clusters are
        numbered in the reference's order (first voxel, row-major)."""
        n = C.c_int64(0)
        self._ck(self.lib.cetkmc_cluster(self.h, float(threshold), C.byref(n)))
        k = n.value
        first = np.zeros((max(k, 1), 3), np.int32)
        size = np.zeros(max(k, 1), np.int64)
        bbox = np.zeros((max(k, 1), 6), np.int32)
        if k:
            self._ck(self.lib.cetkmc_cluster_stats(self.h, k, _ptr(first), _ptr(size), _ptr(bbox)))
        out = dict(first=first[:k], size=size[:k], bbox=bbox[:k])
        if labels:
            lab = np.zeros((self.L,) * 3, np.int32)
            self._ck(self.lib.cetkmc_cluster_labels(self.h, _ptr(lab)))
            out["labels"] = lab
        return out

    # -- sparse site queries (defect model / species counts without full-lattice transfers) -------
    def species_counts(self):
        c = np.zeros(6, np.int64)
        self._ck(self.lib.cetkmc_species_counts(self.h, _ptr(c)))
        return c

    def gather_species(self, species):
        """(linear indices ascending, T at those voxels) of the owned voxels in state ``species``."""
        n = C.c_int64(0)
        self._ck(self.lib.cetkmc_gather_species(self.h, int(species), None, None, 0, C.byref(n)))
        k = n.value
        idx = np.zeros(max(k, 1), np.int64)
        Tv = np.zeros(max(k, 1), np.float64)
        if k:
            self._ck(self.lib.cetkmc_gather_species(self.h, int(species), _ptr(idx), _ptr(Tv), k, C.byref(n)))
        order = np.argsort(idx[:k], kind="stable")
        return idx[:k][order], Tv[:k][order]

    def set_defects_sparse(self, lin_idx):
        a = np.ascontiguousarray(lin_idx, dtype=np.int64)
        self._ck(self.lib.cetkmc_set_defects_sparse(self.h, _ptr(a) if len(a) else None, len(a)))

    def nucleation_count(self):
        return int(self.lib.cetkmc_nucleation_count(self.h))

    def reset_counters(self):
        self._ck(self.lib.cetkmc_reset_counters(self.h))

    def time_sweeps(self, n):
        ms = C.c_double(0.0)
        self._ck(self.lib.cetkmc_time_sweeps(self.h, int(n), C.byref(ms)))
        return ms.value

    def event_overhead(self, n=50):
        """ms between two back-to-back hipEvents around an EMPTY kernel: the part of a hipEvent-bracketed kernel time that is
        not the kernel's."""
        ms = C.c_double(0.0)
        self._ck(self.lib.cetkmc_event_overhead(self.h, int(n), C.byref(ms)))
        return ms.value

    def comm_selftest(self, nbytes=4096, timed=True):
        """Collective transport check of a multi-rank engine (every rank calls it): patterned all-gather and neighbour
        exchange verified on the host; raises with the library's message when the data is wrong.  Returns
        {"allgather_us", "exchange_us"} (averages of 20 further calls, stream synchronisation included) or None."""
        t = (C.c_double * 2)(0.0, 0.0)
        self._ck(self.lib.cetkmc_comm_selftest(self.h, int(nbytes), t if timed else None))
        return {"bytes": int(nbytes), "allgather_us": t[0], "exchange_us": t[1]} if timed else None

    def set_option(self, key, value):
        self._ck(self.lib.cetkmc_set_option(self.h, key.encode(), int(value)))

    def sync(self):
        self._ck(self.lib.cetkmc_sync(self.h))
