"""Loader + ctypes prototypes for include/cetkmc.h."""
import ctypes as C
import glob
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(os.path.dirname(_HERE), "csrc")
INCLUDE = os.path.join(os.path.dirname(os.path.dirname(_HERE)), "include")
SO_PATH = os.path.join(CSRC, "libcetkmc_hip.so")


def sources():
    """Every file the library is compiled from: csrc/*.hip, csrc/*.hpp, include/*.h (globbed, so that a new header
    cannot be forgotten by the staleness check)."""
    return sorted(glob.glob(os.path.join(CSRC, "*.hip")) + glob.glob(os.path.join(CSRC, "*.hpp")) +
                  glob.glob(os.path.join(INCLUDE, "*.h")))


HIPCC_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off",
               "-fno-fast-math", "-I/opt/rocm/include"]


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
                ("rate", C.c_double), ("dep_rank", C.c_int64), ("theta", C.c_double), ("phi", C.c_double)]


class SweepInfo(C.Structure):
    _fields_ = [("total", C.c_double), ("n_events", C.c_int64), ("n_dep", C.c_int64)]


class RunArgs(C.Structure):
    _fields_ = [
        ("step0", C.c_int64), ("n_steps", C.c_int64), ("defect_fraction", C.c_double),
        ("u_pick", C.POINTER(C.c_double)), ("u_defect", C.POINTER(C.c_double)), ("u_np", C.POINTER(C.c_double)),
        ("np_cap", C.c_int64), ("rng_mode", C.c_int32), ("seed", C.c_uint64), ("thermal_mode", C.c_int32),
        ("thermal_dt", C.c_double), ("q_planes", C.POINTER(C.c_double)), ("n_q", C.c_int64),
        ("use_latent", C.c_int32), ("profile", C.c_int32), ("incremental", C.c_int32),
    ]


class SuperArgs(C.Structure):
    _fields_ = [
        ("step0", C.c_int64), ("n_steps", C.c_int64), ("box", C.c_int32), ("defect_fraction", C.c_double),
        ("seed", C.c_uint64), ("thermal_mode", C.c_int32), ("thermal_dt", C.c_double),
        ("q_planes", C.POINTER(C.c_double)), ("n_q", C.c_int64), ("use_latent", C.c_int32), ("null_events", C.c_int32),
    ]


class Counters(C.Structure):
    _fields_ = [(n, C.c_int64) for n in ("steps", "sweeps", "incremental_steps", "thermal_updates", "supersteps", "bytes_h2d",
                                         "bytes_d2h", "alg_bytes_sweep", "alg_bytes_thermal", "profiled_steps")] + \
               [(n, C.c_double) for n in ("ms_thermal", "ms_interface", "ms_sweep", "ms_dirty_rows", "ms_reduce", "ms_select_apply")] + \
               [(n, C.c_int64) for n in ("alg_bytes_table", "table_updates", "interface_launches")] + \
               [("ms_comm", C.c_double), ("comm_calls", C.c_int64)]


ALLGATHER_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64)
EXCHANGE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64)


class HostComm(C.Structure):
    _fields_ = [("allgather", ALLGATHER_FN), ("exchange", EXCHANGE_FN), ("user", C.c_void_p)]


class RunResult(C.Structure):
    _fields_ = [
        ("steps_done", C.c_int64), ("status", C.c_int32), ("np_used", C.c_int64), ("q_used", C.c_int64),
        ("nucleation_count", C.c_int64), ("sweep_ms_total", C.c_double), ("sweep_launches", C.c_int64),
        ("wall_ms", C.c_double), ("full_sweeps", C.c_int64), ("min_margin", C.c_double),
    ]


STRUCT_MIRRORS = {"params": Params, "event": Event, "sweep_info": SweepInfo, "run_args": RunArgs, "run_result": RunResult,
                  "super_args": SuperArgs, "counters": Counters, "host_comm": HostComm}

# name -> (restype, argtypes); every symbol include/cetkmc.h declares
_P = C.POINTER
PROTOTYPES = {
    "cetkmc_last_error": (C.c_char_p, []),
    "cetkmc_abi_version": (C.c_int, []),
    "cetkmc_source_hash": (C.c_char_p, []),
    "cetkmc_struct_size": (C.c_int, [C.c_char_p]),
    "cetkmc_device_count": (C.c_int, [_P(C.c_int)]),
    "cetkmc_create": (C.c_int, [_P(Params), C.c_int, C.c_int, _P(C.c_int), _P(C.c_void_p)]),
    "cetkmc_get_unique_id": (C.c_int, [C.c_char_p]),
    "cetkmc_create_rank": (C.c_int, [_P(Params), C.c_int, C.c_int, C.c_int, C.c_int, C.c_char_p, _P(C.c_void_p)]),
    "cetkmc_create_rank_host": (C.c_int, [_P(Params), C.c_int, C.c_int, C.c_int, C.c_int, _P(HostComm), _P(C.c_void_p)]),
    "cetkmc_destroy": (C.c_int, [C.c_void_p]),
    "cetkmc_set_params": (C.c_int, [C.c_void_p, _P(Params)]),
    "cetkmc_sync": (C.c_int, [C.c_void_p]),
    "cetkmc_set_option": (C.c_int, [C.c_void_p, C.c_char_p, C.c_int64]),
    "cetkmc_owned_planes": (C.c_int, [C.c_void_p, _P(C.c_int), _P(C.c_int)]),
    "cetkmc_upload": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "cetkmc_download": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "cetkmc_upload_planes": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "cetkmc_download_planes": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "cetkmc_set_defects": (C.c_int, [C.c_void_p, C.c_void_p]),
    "cetkmc_set_prev_state": (C.c_int, [C.c_void_p, C.c_void_p]),
    "cetkmc_thermal_cet": (C.c_int, [C.c_void_p, C.c_double, C.c_int]),
    "cetkmc_thermal_laser": (C.c_int, [C.c_void_p, C.c_double, C.c_void_p, C.c_int, C.c_int]),
    "cetkmc_rate_sweep": (C.c_int, [C.c_void_p, _P(SweepInfo)]),
    "cetkmc_select": (C.c_int, [C.c_void_p, C.c_double, _P(Event)]),
    "cetkmc_apply": (C.c_int, [C.c_void_p, _P(Event), C.c_double, C.c_double, C.c_int]),
    "cetkmc_enumerate_events": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, _P(C.c_int64)]),
    "cetkmc_row_sums": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "cetkmc_run_steps": (C.c_int, [C.c_void_p, _P(RunArgs), _P(RunResult), C.c_void_p, C.c_void_p, C.c_void_p]),
    "cetkmc_stage_inputs": (C.c_int, [C.c_void_p, _P(RunArgs)]),
    "cetkmc_run_supersteps": (C.c_int, [C.c_void_p, _P(SuperArgs), _P(RunResult), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "cetkmc_get_counters": (C.c_int, [C.c_void_p, _P(Counters), C.c_int]),
    "cetkmc_cluster": (C.c_int, [C.c_void_p, C.c_double, _P(C.c_int64)]),
    "cetkmc_cluster_stats": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p]),
    "cetkmc_cluster_labels": (C.c_int, [C.c_void_p, C.c_void_p]),
    "cetkmc_species_counts": (C.c_int, [C.c_void_p, C.c_void_p]),
    "cetkmc_gather_species": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int64, _P(C.c_int64)]),
    "cetkmc_set_defects_sparse": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64]),
    "cetkmc_nucleation_count": (C.c_int64, [C.c_void_p]),
    "cetkmc_reset_counters": (C.c_int, [C.c_void_p]),
    "cetkmc_time_sweeps": (C.c_int, [C.c_void_p, C.c_int, _P(C.c_double)]),
    "cetkmc_event_overhead": (C.c_int, [C.c_void_p, C.c_int, _P(C.c_double)]),
    "cetkmc_comm_selftest": (C.c_int, [C.c_void_p, C.c_int64, _P(C.c_double)]),
}


LAST_BUILD = None      # "compiled" | "reused" after build_library()


def source_hash():
    """sha256 (16 hex digits) over every source of the library, in name order: what the library reports through
    cetkmc_source_hash() when it was compiled from exactly these files."""
    import hashlib
    h = hashlib.sha256()
    for f in sources():
        h.update(os.path.basename(f).encode() + b"\0")
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


def library_hash(so_path=None):
    """the source hash compiled into an existing library, or None (missing file / library from before the hash).  Read from
    the file's bytes (marker "cetkmc-source-hash="), NOT by loading it: a library loaded once stays the one the process
    sees under that path, also after build_library() has replaced the file."""
    so_path = so_path or SO_PATH
    if not os.path.exists(so_path):
        return None
    marker = b"cetkmc-source-hash="
    with open(so_path, "rb") as f:
        data = f.read()
    at = data.find(marker)
    if at < 0:
        return None
    end = data.find(b"\0", at)
    return data[at + len(marker):end].decode(errors="replace")


def build_library(force=False, verbose=False):
    """hipcc cross-compiles for gfx950 without a GPU; the .so stays in-tree (csrc/).  Recompiles when the library is missing
    or was compiled from other sources than the ones beside it (source hash compiled in; file times are not trusted: a
    checkout or a copy changes them)."""
    global LAST_BUILD
    want = source_hash()
    if not force and library_hash() == want:
        LAST_BUILD = "reused"
        return SO_PATH
    import fcntl
    # several processes may import at once (torchrun ranks, the multi-process tests): one compiles, the others wait for the
    # lock and then find the finished library; hipcc writes a temporary file that is renamed into place, so a concurrent
    # dlopen never sees a half-written library
    with open(SO_PATH + ".lock", "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            if not force and library_hash() == want:
                LAST_BUILD = "reused"
                return SO_PATH
            LAST_BUILD = "compiled"
            hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
            tmp = f"{SO_PATH}.tmp.{os.getpid()}"
            cmd = [hipcc] + HIPCC_FLAGS + [f'-DCETKMC_SRC_HASH="{want}"', os.path.join(CSRC, "cetkmc_hip.hip"), "-o", tmp, "-ldl"]
            if verbose:
                print(" ".join(cmd))
            try:
                subprocess.check_call(cmd, cwd=CSRC)
                os.replace(tmp, SO_PATH)
            finally:
                if os.path.exists(tmp):
                    os.remove(tmp)
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)
    return SO_PATH


_lib = None


def load():
    """Load libcetkmc_hip.so; raises (never falls back) when it is missing."""
    global _lib
    if _lib is not None:
        return _lib
    so_path = os.environ.get("CETKMC_LIB", SO_PATH)       # alternative build of the same library (A/B timing only)
    if "CETKMC_LIB" not in os.environ and os.path.exists(so_path) and library_hash(so_path) != source_hash() \
            and os.path.exists(os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")):
        build_library()        # a library compiled from other sources than the ones beside it: recompile (hipcc is here)
    if not os.path.exists(so_path):
        raise RuntimeError(
            f"{so_path} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950).  cetkmc has no CPU fallback.")
    lib = C.CDLL(so_path)
    for name, (res, args) in PROTOTYPES.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    if lib.cetkmc_abi_version() != 1:
        raise RuntimeError("libcetkmc_hip.so ABI version mismatch")
    if "CETKMC_LIB" not in os.environ:        # (an explicitly chosen alternative build is the caller's business)
        have, want = lib.cetkmc_source_hash().decode(), source_hash()
        if have != want:
            raise RuntimeError(f"{so_path} was compiled from other sources (hash {have}, sources {want}): rebuild it with "
                               "`python -c 'import __graft_entry__ as g; g.build()'`")
    for name, mirror in STRUCT_MIRRORS.items():
        if lib.cetkmc_struct_size(name.encode()) != C.sizeof(mirror):
            raise RuntimeError(f"libcetkmc_hip.so: struct {name} is {lib.cetkmc_struct_size(name.encode())} bytes, "
                               f"the Python mirror {C.sizeof(mirror)} (stale build?)")
    _lib = lib
    return lib
