"""Temperature-field updates on the GPU (drop-in for the reference ``thermal_solver.py``).

Same pure-function interface (arrays in, new array out); the 7-point explicit-Euler stencil
runs in the ``k_thermal`` HIP kernel, bit-identical to ``scipy.ndimage.laplace`` + NumPy
(thermal_solver.py:36-117).  The Gaussian source plane is built on the host with NumPy, like
the reference, so its ``exp`` rounds as the reference's does.
"""
import numpy as np

from constants import LATTICE_SIZE, T_MELT, T_SUB, VOXEL_SIZE

K = 173.0        # [W/m-K]      thermal_solver.py:6
RHO = 19300.0    # [kg/m^3]     :7
CP = 132.0       # [J/kg-K]     :8
ALPHA = K / (RHO * CP)
DEFAULT_BEAM_RADIUS = 50e-6
DEFAULT_ABSORPTIVITY = 0.35

class _EngineCache:
    """Device lattices kept between calls of the pure-function drop-ins, least recently used first out: at most
    ``max_engines`` handles (a 256^3 handle owns ~1.3 GB of HBM); ``close_engines()`` releases them all."""

    def __init__(self, max_engines=2):
        from collections import OrderedDict
        self.max_engines = max_engines
        self._d = OrderedDict()

    def get(self, L, make):
        eng = self._d.pop(L, None)
        if eng is None:
            while len(self._d) >= self.max_engines:
                self._d.popitem(last=False)[1].close()
            eng = make()
        self._d[L] = eng
        return eng

    def close(self):
        while self._d:
            self._d.popitem()[1].close()

    def __len__(self):
        return len(self._d)


_engines = _EngineCache()


def close_engines():
    """Release the cached device lattices of this module."""
    _engines.close()


def _engine(L):
    import cetkmc
    return _engines.get(L, lambda: cetkmc.Engine(L))


def build_temperature_field(L=None):
    """Steady ramp along axis 0: T(i) = T_SUB + (T_MELT-T_SUB)/(L-1) * i (thermal_solver.py:15-34)."""
    if L is None:
        L = LATTICE_SIZE
    slope = (T_MELT - T_SUB) / (L - 1) if L > 1 else 0.0
    column = T_SUB + slope * np.arange(L, dtype=np.float64)
    return np.repeat(np.repeat(column[:, None, None], L, axis=1), L, axis=2)


def laser_source_plane(L, laser_pos, laser_power, beam_radius=DEFAULT_BEAM_RADIUS, absorptivity=DEFAULT_ABSORPTIVITY):
    """Volumetric source [W/m^3] of plane i=L-1 (thermal_solver.py:82-95).  As in the reference,
    ``j0`` centres the beam on BOTH in-plane axes and ``i0`` is ignored (:86)."""
    _i0, j0 = laser_pos
    axis = np.arange(L, dtype=np.float64)
    JJ, KK = np.meshgrid(axis, axis, indexing="ij")
    r_m = np.sqrt((JJ - j0) ** 2 + (KK - j0) ** 2) * VOXEL_SIZE
    peak = laser_power * absorptivity / (np.pi * beam_radius * beam_radius)
    surface = peak * np.exp(-(r_m ** 2) / (beam_radius ** 2))
    return surface / VOXEL_SIZE


def update_temperature(T, state, prev_state, dt, laser_pos, laser_power,
                       beam_radius=DEFAULT_BEAM_RADIUS, absorptivity=DEFAULT_ABSORPTIVITY):
    """Explicit Euler step with Laplacian + Gaussian surface source on i=L-1 + latent heat where
    ``prev_state==0 & state!=0``; clipped to [T_SUB, 1.1*T_MELT] (thermal_solver.py:36-105)."""
    L = T.shape[0]
    assert T.shape == (L, L, L)
    assert state.shape == T.shape and prev_state.shape == T.shape
    eng = _engine(L)
    eng.upload(state=state, T=T)
    eng.set_prev_state(prev_state)
    eng.thermal_laser(dt, laser_source_plane(L, laser_pos, laser_power, beam_radius, absorptivity),
                      use_latent=True, scrub_nan=False)
    return eng.download(state=False, theta=False, phi=False, T=True)["T"]


def update_temperature_cet(T, state, dt=1e-6):
    """Diffusion-only step used by run_kmc; ``state`` is unused, as in the reference
    (thermal_solver.py:107-117)."""
    L = T.shape[0]
    eng = _engine(L)
    eng.upload(T=T)
    eng.thermal_cet(dt, scrub_nan=False)
    return eng.download(state=False, theta=False, phi=False, T=True)["T"]
