"""cetkmc -- ctypes binding of libcetkmc_hip.so (hand-written HIP kernels for MI355X / gfx950).

The flat modules next to this package (kmc_simulation.py, kmc_event_rates.py,
thermal_solver.py, ...) keep the reference's names and call signatures and are built on
:class:`cetkmc.engine.Engine`.  There is no CPU fallback: loading fails loudly when the
shared library is missing, and every compute call fails when no GPU is present.
"""
from .engine import Engine, Event, default_params, device_count, build_library, library_path  # noqa: F401
