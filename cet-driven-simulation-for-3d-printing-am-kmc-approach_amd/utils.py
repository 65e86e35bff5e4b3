"""Grain clustering and CET helper functions (drop-in for the parts of the reference ``utils.py``
that the hot path's callers use: utils.py:13-111, plus the small CET helpers :117-229).

Host-side analysis code (NumPy); written fresh with the reference's traversal semantics so that
cluster membership, discovery order and labels are identical.
"""
import numpy as np

from constants import CET_AR_THRESHOLD, CET_GR_THRESHOLD, NU_DEP, STATES
from kmc_event_rates import _OFFSETS, compute_misorientation, get_bcc_neighbors  # noqa: F401


def get_cluster_edges(cluster):
    """6-neighbour adjacency pairs inside one cluster (utils.py:13-26)."""
    members = set(tuple(int(x) for x in v) for v in cluster)
    edges = set()
    for (i, j, k) in members:
        for n in ((i + 1, j, k), (i - 1, j, k), (i, j + 1, k), (i, j - 1, k), (i, j, k + 1), (i, j, k - 1)):
            if n in members:
                edges.add(tuple(sorted([(i, j, k), n])))
    return list(edges)


def dfs_cluster(state, visited, cluster_label, orientation_theta, orientation_phi=None, theta_threshold=0.5):
    """Generator over grains: depth-first flood over the 14-offset stencil, joining a neighbour
    when its misorientation to the CURRENT voxel is below the threshold (utils.py:28-67).
    Seeds are visited in row-major order; ``visited`` receives labels cluster_label, +1, ..."""
    Lx, Ly, Lz = state.shape
    occupied = state != 0
    if orientation_phi is not None:
        st, ct = np.sin(orientation_theta), np.cos(orientation_theta)
        vx, vy, vz = st * np.cos(orientation_phi), st * np.sin(orientation_phi), ct
    offsets = [tuple(int(x) for x in o) for o in _OFFSETS]
    for (i, j, k) in np.argwhere(occupied):
        i, j, k = int(i), int(j), int(k)
        if visited[i, j, k] != 0:
            continue
        visited[i, j, k] = cluster_label
        cluster = [(i, j, k)]
        stack = [(i, j, k)]
        while stack:
            ci, cj, ck = stack.pop()
            for di, dj, dk in offsets:
                ni, nj, nk = ci + di, cj + dj, ck + dk
                # the reference enumerates neighbours with L = Lx on every axis (utils.py:45)
                if not (0 <= ni < Lx and 0 <= nj < Lx and 0 <= nk < Lx):
                    continue
                if not (nj < Ly and nk < Lz) or not occupied[ni, nj, nk] or visited[ni, nj, nk] != 0:
                    continue
                if orientation_phi is None:
                    mis = abs(orientation_theta[ci, cj, ck] - orientation_theta[ni, nj, nk])
                else:
                    dot = vx[ci, cj, ck] * vx[ni, nj, nk] + vy[ci, cj, ck] * vy[ni, nj, nk] + vz[ci, cj, ck] * vz[ni, nj, nk]
                    mis = np.arccos(max(min(dot, 1.0), -1.0))
                if mis < theta_threshold:
                    visited[ni, nj, nk] = cluster_label
                    cluster.append((ni, nj, nk))
                    stack.append((ni, nj, nk))
        yield cluster
        cluster_label += 1
    return cluster_label


def get_clusters(state, orientation_theta, orientation_phi=None, theta_threshold=0.5):
    """(list of clusters, label volume).  The second value is the int32 ``visited`` volume --
    the reference returns it under the name cluster_sizes (utils.py:69-84)."""
    if state.size == 0:
        return [], np.array([])
    visited = np.zeros(state.shape, dtype=np.int32)
    clusters = [list(c) for c in dfs_cluster(state, visited, 1, orientation_theta, orientation_phi, theta_threshold)]
    return clusters, visited


def compute_metrics(clusters, state, atom_type):
    """(size ratio, coverage, sizes, species counts) -- the older helper of utils.py:88-102."""
    L = state.shape[0]
    sizes = [len(c) for c in clusters if c]
    if not sizes:
        return 0.0, 0.0, [], {STATES["W"]: 0, STATES["Re"]: 0, STATES["C"]: 0}
    ratio = max(sizes) / max(min(sizes), 1)
    coverage = np.sum(state != STATES["Empty"]) / float(L * L * L)
    counts = {s: int(np.sum(atom_type == s)) for s in (STATES["W"], STATES["Re"], STATES["C"])}
    return ratio, coverage, sizes, counts


def calculate_aspect_ratio(cluster):
    """Longest / shortest bounding-box edge of a cluster (utils.py:104-111)."""
    coords = np.array(cluster)
    dims = coords.max(axis=0) - coords.min(axis=0) + 1
    return float(np.max(dims)) / float(max(np.min(dims), 1))


# --- CET helpers (utils.py:117-229) ------------------------------------------------------
def detect_CET_transition(state, orientation_theta, threshold=0.3):
    """True when >50 % of the grains in the top ``threshold`` fraction of axis 2 are wider than
    1.5x their height (utils.py:117-145)."""
    L = state.shape[2]
    z0 = max(0, L - int(L * threshold))
    clusters, _ = get_clusters(state[:, :, z0:], orientation_theta[:, :, z0:])
    if not clusters:
        return False
    wide = 0
    for c in clusters:
        xyz = np.array(c)
        if len(xyz) == 0:
            continue
        z = xyz[:, 2] + z0
        if max(np.ptp(xyz[:, 0]), np.ptp(xyz[:, 1])) > (np.max(z) - np.min(z)) * 1.5:
            wide += 1
    return wide / len(clusters) > 0.5


def calculate_G_over_R(G, R):
    return np.inf if R == 0 else float(G) / float(R)


def determine_CET(aspect_ratio, G_over_R, gr_threshold=CET_GR_THRESHOLD, ar_threshold=CET_AR_THRESHOLD):
    """Columnar iff the aspect ratio (and, when a finite G/R is known, G/R as well) reaches its threshold; prints the
    reference's debug line (utils.py:168-176)."""
    elongated = aspect_ratio >= ar_threshold
    have_gr = G_over_R is not None and not np.isinf(G_over_R)
    if have_gr:
        print(f"Debug: G_over_R={G_over_R:.2e}, AspectRatio={aspect_ratio:.2f}, thresholds={gr_threshold}/{ar_threshold}")
        elongated = elongated and G_over_R >= gr_threshold
    else:
        print(f"Debug: G_over_R invalid, using AspectRatio={aspect_ratio:.2f} vs threshold={ar_threshold}")
    return "Columnar" if elongated else "Equiaxed"


def validate_aspect_ratio(aspect_ratio, ar_threshold=CET_AR_THRESHOLD):
    return aspect_ratio >= ar_threshold


def validate_CET_with_GR(G, R, aspect_ratio, gr_threshold=CET_GR_THRESHOLD, ar_threshold=CET_AR_THRESHOLD):
    return determine_CET(aspect_ratio, calculate_G_over_R(G, R), gr_threshold, ar_threshold)


def overall_microstructure_classification(clusters, G=None, R=None, gr_threshold=CET_GR_THRESHOLD,
                                          ar_threshold=CET_AR_THRESHOLD):
    if not clusters:
        return "No grains detected"
    avg_ar = float(np.mean([calculate_aspect_ratio(c) for c in clusters]))
    if G is None or R is None:
        return determine_CET(avg_ar, None, gr_threshold, ar_threshold)
    return validate_CET_with_GR(G, R, avg_ar, gr_threshold, ar_threshold)


def compute_CET_metrics(state, orientation_theta, orientation_phi=None, G=None, R=None):
    if orientation_phi is None:
        orientation_phi = np.zeros_like(orientation_theta)
    clusters, _ = get_clusters(state, orientation_theta, orientation_phi)
    if not clusters:
        return {"avg_ar": 0.0, "f_eq": 0.0, "n_density": 0.0, "classification": "No grains"}
    ars = [calculate_aspect_ratio(c) for c in clusters]
    avg_ar = float(np.mean(ars))
    cls = validate_CET_with_GR(G, R, avg_ar) if (G is not None and R is not None) else determine_CET(avg_ar, None)
    return {"avg_ar": avg_ar, "f_eq": len([a for a in ars if a < CET_AR_THRESHOLD]) / len(ars),
            "n_density": len(clusters) / float(state.size), "classification": cls}


def estimate_temperature_gradient(T_field):
    grad_z = np.abs(np.gradient(T_field, 1.0, axis=2))
    nz = grad_z > 0
    return float(np.mean(grad_z[nz])) if np.any(nz) else 0.0


def estimate_growth_rate(previous_state, current_state, timestep):
    prev_idx = np.where(previous_state != STATES["Empty"])
    curr_idx = np.where(current_state != STATES["Empty"])
    prev_top = int(prev_idx[2].max()) if prev_idx[0].size > 0 else 0
    curr_top = int(curr_idx[2].max()) if curr_idx[0].size > 0 else 0
    if timestep <= 0 or curr_top == prev_top:
        return NU_DEP * 1e-16
    return float((curr_top - prev_top) * 1.0 / timestep)
