"""Synthetic lattices for the benchmark configurations of BASELINE.json / SURVEY.md 8(d).

Every plane i is generated from its own ``RandomState(seed*1000003 + i)`` so a rank can build
exactly the planes of its slab (+halo) without materialising the whole lattice.

Fill rule (configs 2/3): voxels with k < L/4 are occupied, species drawn {W:0.7, Re:0.1,
C:0.2}, 0.5 % of the occupied voxels are defects (state 4, orientation 0), orientation
theta~U(0,pi), phi~U(0,2pi); defect mask Bernoulli(0.01) on C sites; T is either constant
or the lattice_init ramp T_sub + (T_melt-T_sub)/L * k (lattice_init.py:29-32).
"""
import numpy as np

from constants import T_MELT, T_SUB


def planes(L, i_begin, i_end, seed=42, constant_T=None, fill_frac=0.25):
    n = i_end - i_begin
    kfill = int(L * fill_frac)
    state = np.zeros((n, L, L), np.uint8)
    theta = np.zeros((n, L, L), np.float64)
    phi = np.zeros((n, L, L), np.float64)
    defects = np.zeros((n, L, L), np.uint8)
    if constant_T is None:
        ramp = T_SUB + ((T_MELT - T_SUB) / L) * np.arange(L)
        T = np.broadcast_to(ramp[None, None, :], (n, L, L)).copy()
    else:
        T = np.full((n, L, L), float(constant_T))
    for p in range(n if kfill > 0 else 0):
        rs = np.random.RandomState((seed * 1000003 + (i_begin + p)) % (2 ** 32))
        sp = rs.choice(np.array([1, 2, 3], np.uint8), size=(L, kfill), p=[0.7, 0.1, 0.2])
        dmask = rs.random_sample((L, kfill)) < 0.005
        th = rs.uniform(0, np.pi, (L, kfill))
        ph = rs.uniform(0, 2 * np.pi, (L, kfill))
        cdef = rs.random_sample((L, kfill)) < 0.01
        sp[dmask] = 4
        th[dmask] = 0.0
        ph[dmask] = 0.0
        state[p, :, :kfill] = sp
        theta[p, :, :kfill] = th
        phi[p, :, :kfill] = ph
        defects[p, :, :kfill] = (cdef & (sp == 3)).astype(np.uint8)
    return state, theta, phi, T, defects


def laser_planes(L, step0, n_steps, power=200.0, beam_radius=50e-6, absorptivity=0.35, j_start=32.0):
    """Source planes for every thermal update in [step0, step0+n): Gaussian spot moving along j,
    j0 = j_start*(L/256) + step/20."""
    from thermal_solver import laser_source_plane
    out = []
    for g in range(step0, step0 + n_steps):
        if g % 20 == 0:
            j0 = j_start * (L / 256.0) + g / 20
            out.append(laser_source_plane(L, (L - 1, j0), power, beam_radius, absorptivity))
    if not out:
        return np.zeros((0, L, L))
    return np.stack(out)
