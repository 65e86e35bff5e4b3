"""Grain clustering and CET helper functions (drop-in for the parts of the reference ``utils.py``
that the hot path's callers use: utils.py:13-111; the reference's unused CET helpers :117-258 are not rebuilt).

Host-side analysis code (NumPy); written fresh with the reference's traversal semantics so that
cluster membership, discovery order and labels are identical.
"""
import numpy as np

from constants import STATES
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
