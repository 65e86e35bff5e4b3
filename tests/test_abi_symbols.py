"""The C-ABI library loads without a GPU and exports every symbol include/cetkmc.h declares;
compute calls fail loudly (no CPU fallback)."""
import ctypes
import os
import re

import pytest

from conftest import PKG, ROOT


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "cetkmc.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(cetkmc_[a-z_0-9]+)\s*\(", text)))


def test_library_exports_header_symbols():
    from cetkmc import _lib
    _lib.build_library()
    lib = ctypes.CDLL(_lib.SO_PATH)
    names = _declared_symbols()
    assert len(names) >= 25
    for n in names:
        assert hasattr(lib, n), n
    assert set(names) == set(_lib.PROTOTYPES), set(names) ^ set(_lib.PROTOTYPES)
    lib.cetkmc_abi_version.restype = ctypes.c_int
    assert lib.cetkmc_abi_version() == 1


def test_library_hash_names_its_sources():
    """The hash compiled into the library is the hash of csrc/*.hip, csrc/*.hpp, include/*.h beside it; the binding reads
    it from the file's bytes (a dlopen of a stale file would pin that file for the rest of the process) and the loaded
    library reports the same one."""
    from cetkmc import _lib
    _lib.build_library()
    assert _lib.library_hash() == _lib.source_hash()
    lib = _lib.load()
    assert lib.cetkmc_source_hash().decode() == _lib.source_hash()
    assert _lib.library_hash(os.path.join(ROOT, "include", "cetkmc.h")) is None      # a file without the marker
    assert _lib.library_hash(os.path.join(ROOT, "no_such_file.so")) is None


def test_struct_layouts_match_header():
    from cetkmc import _lib
    assert ctypes.sizeof(_lib.Event) == 64
    assert ctypes.sizeof(_lib.Params) == 26 * 8
    assert ctypes.sizeof(_lib.SweepInfo) == 24


def test_no_cpu_fallback():
    """Without a GPU the product refuses to compute (skipped on the GPU box)."""
    import cetkmc
    from cetkmc import _lib
    lib = _lib.load()
    n = ctypes.c_int(0)
    rc = lib.cetkmc_device_count(ctypes.byref(n))
    if rc == 0 and n.value > 0:
        pytest.skip("GPU present")
    with pytest.raises(RuntimeError, match="no CPU fallback|no usable HIP device"):
        cetkmc.Engine(8)
    import numpy as np
    import thermal_solver
    with pytest.raises(RuntimeError):
        thermal_solver.update_temperature_cet(np.full((4, 4, 4), 3000.0), None)


def test_product_never_imports_oracle():
    """The product may MENTION the oracle in comments, but never imports, links or calls it."""
    pat = re.compile(r"import\s+oracle|from\s+oracle|oracle\.|oracle/|libcet_oracle|\borc_[a-z_]+\s*\(")
    for dirpath, _, files in os.walk(PKG):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".h")):
                src = open(os.path.join(dirpath, f)).read()
                assert not pat.search(src), os.path.join(dirpath, f)


def test_struct_sizes_match_python_mirrors():
    """cetkmc_struct_size: every ABI struct has the size of its ctypes mirror (the loader enforces it too)."""
    import ctypes as C

    from cetkmc import _lib
    lib = _lib.load()
    for name, mirror in _lib.STRUCT_MIRRORS.items():
        assert lib.cetkmc_struct_size(name.encode()) == C.sizeof(mirror), name
    assert lib.cetkmc_struct_size(b"no_such_struct") == -1
