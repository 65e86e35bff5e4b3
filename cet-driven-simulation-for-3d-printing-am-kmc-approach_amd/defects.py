"""Diagnostic point-defect mask on carbon sites (drop-in for the reference ``defects.py``).

Host-side NumPy: the mask is drawn from NumPy's *global* legacy stream in row-major order
of the C sites, exactly like the reference (defects.py:4-19), because ``run_kmc`` shares
that stream with the orientation draws.
"""
import numpy as np

from constants import DEFECT_PROB_BASE, K_T, T_SUB

_C_SITE = 3        # STATES['C'], hard-coded in the reference (defects.py:8)
_DEFECT_STATE = 4  # STATES['Defect'] (defects.py:29)


def track_defects(state, atom_type, L, T=None):
    """0/1 mask: each C site is flagged with p = 0.12*exp(-0.3/(kT*T)) (defects.py:4-19)."""
    if L == 0:
        return np.zeros((0, 0, 0), dtype=int)
    mask = np.zeros((L, L, L), dtype=int)
    sites = atom_type == _C_SITE
    n_sites = int(np.count_nonzero(sites))
    if n_sites:
        if T is None:
            p = np.full(n_sites, DEFECT_PROB_BASE)
        else:
            t_here = T[sites]
            with np.errstate(divide="ignore", invalid="ignore"):
                t_here = np.where(t_here > 0, t_here, T_SUB)
                p = DEFECT_PROB_BASE * np.exp(-0.3 / (K_T * t_here))
        p = np.clip(p, 0.0, 1.0)
        mask[sites] = (np.random.random(n_sites) < p).astype(int)
    return mask


def get_defect_density(defects, voxel_size=5e-6):
    """Flagged sites per cubic metre (defects.py:21-23)."""
    volume = defects.size * (voxel_size ** 3)
    return np.sum(defects) / volume if volume > 0 else 0.0


def introduce_defects(state, atom_type, T=None, apply_to_state=False, voxel_size=5e-6):
    """(mask, density); optionally writes state 4 into flagged sites (defects.py:25-31)."""
    mask = track_defects(state, atom_type, state.shape[0], T)
    if apply_to_state:
        state[mask == 1] = _DEFECT_STATE
    return mask, get_defect_density(mask, voxel_size)


def refresh_defects_device(engine, rank_counts=None, rank=0):
    """introduce_defects(state, atom_type, T) for the lattice resident on the GPU without moving the
    lattice: only the carbon sites (index + T) come to the host, the Bernoulli draws are taken from
    NumPy's global stream in row-major site order exactly as track_defects does (defects.py:8-18), and
    only the flagged indices go back.  Returns (number of flagged sites, density).

    Across ranks (axis-0 slabs, one engine per rank): ``rank_counts(n_mine)`` returns every rank's site count
    (an all-gather); the global row-major site order is the concatenation of the ranks' orders, so each rank
    draws the whole stream -- every rank holds the same generator state -- and keeps its own slice.  The
    returned count / density are then this rank's share."""
    idx, t_here = engine.gather_species(_C_SITE)
    L = engine.L
    n_before, n_total = 0, len(idx)
    if rank_counts is not None:
        counts = list(rank_counts(len(idx)))
        n_before, n_total = int(sum(counts[:rank])), int(sum(counts))
    flagged = np.zeros(0, dtype=np.int64)
    if n_total:
        u = np.random.random(n_total)[n_before:n_before + len(idx)]
        if len(idx):
            with np.errstate(divide="ignore", invalid="ignore"):
                t_here = np.where(t_here > 0, t_here, T_SUB)
                p = DEFECT_PROB_BASE * np.exp(-0.3 / (K_T * t_here))
            p = np.clip(p, 0.0, 1.0)
            flagged = idx[u < p]
    engine.set_defects_sparse(flagged)
    volume = (L ** 3) * (5e-6 ** 3)
    return int(len(flagged)), (len(flagged) / volume if volume > 0 else 0.0)
