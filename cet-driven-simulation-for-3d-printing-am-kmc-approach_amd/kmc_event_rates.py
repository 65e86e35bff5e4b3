"""Per-voxel Arrhenius event rates on the GPU (drop-in for the reference ``kmc_event_rates.py``).

``get_event_rates`` keeps the reference signature and return type (a Python list of
``(type_bytes, (i,j,k), rate, (ti,tj,tk), atom)`` tuples in the reference's order,
kmc_event_rates.py:162-176) but the rates are evaluated by the HIP kernels behind
``libcetkmc_hip.so`` (csrc/voxel.hpp restates kmc_event_rates.py:42-160).  The two small
geometry helpers that ``utils`` imports stay host functions.

There is no CPU fallback: without the library or a GPU the call raises.
"""
import numpy as np

from constants import IMPURITY_RE

_OFFSETS = np.array(
    [(1, 1, 0), (1, -1, 0), (-1, 1, 0), (-1, -1, 0), (0, 1, 1), (0, 1, -1), (0, -1, 1), (0, -1, -1),
     (2, 0, 0), (-2, 0, 0), (0, 2, 0), (0, -2, 0), (0, 0, 2), (0, 0, -2)], dtype=np.int64)

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


def _engine(L, impurity_c):
    """One cached device lattice per edge length (handles are reused across calls; bounded, see _EngineCache)."""
    import cetkmc
    eng = _engines.get(L, lambda: cetkmc.Engine(L, impurity_c=impurity_c))
    if eng.params.impurity_c != float(impurity_c):
        eng.set_impurity_c(impurity_c)
    return eng


def get_bcc_neighbors(i, j, k, L):
    """In-bounds members of the 14-offset stencil, reference order (kmc_event_rates.py:25-40)."""
    cand = _OFFSETS + np.array((i, j, k), dtype=np.int64)
    keep = np.all((cand >= 0) & (cand < L), axis=1)
    return cand[keep]


def compute_misorientation(theta1, phi1, theta2, phi2):
    """Angle between the two orientation unit vectors (kmc_event_rates.py:9-23)."""
    ax, ay, az = np.sin(theta1) * np.cos(phi1), np.sin(theta1) * np.sin(phi1), np.cos(theta1)
    bx, by, bz = np.sin(theta2) * np.cos(phi2), np.sin(theta2) * np.sin(phi2), np.cos(theta2)
    dot = ax * bx + ay * by + az * bz
    return np.arccos(max(min(dot, 1.0), -1.0))


def get_event_rates(state, orientation_theta, orientation_phi, T, atom_type, defects_mask, L,
                    states_w, states_re, states_c, step=0, debug_step=1000, impurity_c=0.0):
    """All candidate events of the lattice, in the reference's order.

    ``atom_type``, ``step`` and ``debug_step`` are accepted and unused, as in the reference
    (kmc_event_rates.py:43-46,162-163).  The deposition species is drawn on the host with one
    ``np.random.random()`` per candidate (kmc_event_rates.py:65) so the global NumPy stream
    advances exactly as in the reference's interpreter mode.
    """
    if (states_w, states_re, states_c) != (1, 2, 3):
        raise ValueError("species ids other than W=1, Re=2, C=3 are not supported by the device kernels")
    from cetkmc.engine import TYPE_BYTES
    eng = _engine(int(L), impurity_c)
    eng.upload(state, orientation_theta, orientation_phi, T, defects_mask)
    ev, n = eng.enumerate_events()
    n_dep = int(np.count_nonzero(ev["type"] == 0))
    u = np.random.random(n_dep) if n_dep else np.zeros(0)
    species = np.where(u < impurity_c, states_c, np.where(u < impurity_c + IMPURITY_RE, states_re, states_w))
    atoms = ev["atom"].copy()
    atoms[ev["type"] == 0] = species
    out = []
    for t, p, r, g, a in zip(ev["type"].tolist(), ev["pos"].tolist(), ev["rate"].tolist(), ev["target"].tolist(),
                             atoms.tolist()):
        out.append((TYPE_BYTES[t], tuple(p), r, tuple(g), a))
    return out
