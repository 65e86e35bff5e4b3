"""Microstructure metrics (drop-in for the reference ``metrics.py``; host-side NumPy).

``compute_metrics`` returns the same 17-key dict (metrics.py:41-96) including the reference's
quirk that the grain-diameter percentiles are taken over the LABEL volume returned by
``get_clusters`` (metrics.py:43,76 <-> utils.py:84).
"""
import numpy as np

from constants import CET_AR_THRESHOLD, CET_EQ_THRESHOLD, VOXEL_SIZE
from utils import calculate_aspect_ratio, get_clusters


def grain_aspect_ratio(state, voxel_size=VOXEL_SIZE):
    """Height (axis 2) over the larger lateral extent of the occupied region (metrics.py:6-15)."""
    occ = np.argwhere(state > 0)
    if occ.size == 0:
        return 0.0
    ext = occ.max(axis=0) - occ.min(axis=0) + 1
    return (ext[2] * voxel_size) / (max(ext[0], ext[1]) * voxel_size + 1e-12)


def equiaxed_fraction(state, threshold=CET_AR_THRESHOLD, voxel_size=VOXEL_SIZE):
    if np.argwhere(state > 0).size == 0:
        return 0.0
    return 1.0 if grain_aspect_ratio(state, voxel_size) < threshold else 0.0


def nucleation_density(state, voxel_size=VOXEL_SIZE):
    volume = np.prod(state.shape) * (voxel_size ** 3)
    return np.count_nonzero(state > 0) / volume if volume > 0 else 0.0


def grain_sizes(state):
    occ = np.argwhere(state > 0)
    return [] if occ.size == 0 else [len(occ)]


def compute_voxel_fraction(count, total_voxels):
    return count / total_voxels if total_voxels > 0 else 0.0


def compute_boundary_fraction(impurity_mask, grain_ids):
    """Fraction of flagged voxels with a 6-neighbour of a different grain id, edges replicated
    (metrics.py:113-138)."""
    n_imp = np.count_nonzero(impurity_mask)
    if n_imp == 0:
        return 0.0
    g = np.pad(grain_ids, 1, mode="edge")
    core = g[1:-1, 1:-1, 1:-1]
    differs = np.zeros(grain_ids.shape, dtype=bool)
    for sl in ((slice(2, None), slice(1, -1), slice(1, -1)), (slice(None, -2), slice(1, -1), slice(1, -1)),
               (slice(1, -1), slice(2, None), slice(1, -1)), (slice(1, -1), slice(None, -2), slice(1, -1)),
               (slice(1, -1), slice(1, -1), slice(2, None)), (slice(1, -1), slice(1, -1), slice(None, -2))):
        differs |= g[sl] != core
    return np.count_nonzero(differs & np.asarray(impurity_mask, dtype=bool)) / n_imp


def equivalent_diameter_um(cluster_sizes, voxel_size):
    """(d50, d90) in micrometres of sphere-equivalent diameters (metrics.py:141-150)."""
    if len(cluster_sizes) == 0:
        return 0.0, 0.0
    volumes = np.array(cluster_sizes) * (voxel_size ** 3)
    d = ((6.0 * volumes / np.pi) ** (1.0 / 3.0)) * 1e6
    return np.median(d), np.percentile(d, 90)


def compute_metrics(state, theta, phi, defects=None, voxel_size=VOXEL_SIZE,
                    W_mask=None, Re_mask=None, C_mask=None, grain_ids=None, rng_seed=None):
    clusters, label_volume = get_clusters(state, theta, phi, theta_threshold=0.5)
    if not clusters:
        return {
            "AspectRatio": 0.0, "EquiaxedFraction": 0.0, "NucleationDensity": 0.0,
            "AvgGrainSize": 0.0, "GrainCount": 0, "DefectDensity": 0.0,
            "Frac_W": 0.0, "Frac_Re": 0.0, "Frac_C": 0.0,
            "C_boundary_frac": 0.0, "Re_boundary_frac": 0.0,
            "Defect_voxel_count": 0, "Defect_voxel_frac": 0.0,
            "Grain_d50_um": 0.0, "Grain_d90_um": 0.0,
            "VOXEL_SIZE_m": voxel_size, "RANDOM_SEED": rng_seed,
        }
    ars = [calculate_aspect_ratio(c) for c in clusters]
    volume = state.size * (voxel_size ** 3)
    n_vox = state.size
    n_def = np.sum(defects) if defects is not None else 0
    d50, d90 = equivalent_diameter_um(label_volume, voxel_size)

    def frac(mask):
        return compute_voxel_fraction(np.count_nonzero(mask), n_vox) if mask is not None else 0.0

    def bfrac(mask):
        return compute_boundary_fraction(mask, grain_ids) if (mask is not None and grain_ids is not None) else 0.0

    return {
        "AspectRatio": np.mean(ars),
        "EquiaxedFraction": np.mean(np.array(ars) < CET_AR_THRESHOLD),
        "NucleationDensity": len(clusters) / volume if volume > 0 else 0.0,
        "AvgGrainSize": np.mean([len(c) for c in clusters]) * voxel_size * 1e6,
        "GrainCount": len(clusters),
        "DefectDensity": n_def / volume if volume > 0 else 0.0,
        "Frac_W": frac(W_mask), "Frac_Re": frac(Re_mask), "Frac_C": frac(C_mask),
        "C_boundary_frac": bfrac(C_mask), "Re_boundary_frac": bfrac(Re_mask),
        "Defect_voxel_count": n_def,
        "Defect_voxel_frac": compute_voxel_fraction(n_def, n_vox),
        "Grain_d50_um": d50, "Grain_d90_um": d90,
        "VOXEL_SIZE_m": voxel_size, "RANDOM_SEED": rng_seed,
    }


def compute_metrics_device(engine, n_voxels, defects_count=0, voxel_size=VOXEL_SIZE, rng_seed=None):
    """Same dict as :func:`compute_metrics` (without the optional mask fractions), but the grain
    clustering runs on the GPU on the lattice resident in ``engine`` (cetkmc_cluster: connected
    components, numbered like the reference's DFS, utils.py:28-84).  Only the per-cluster sizes /
    bounding boxes and the int32 label volume (for the d50/d90 quirk, metrics.py:76) cross PCIe."""
    cl = engine.clusters(0.5, labels=True)
    n = len(cl["size"])
    if n == 0:
        return {
            "AspectRatio": 0.0, "EquiaxedFraction": 0.0, "NucleationDensity": 0.0,
            "AvgGrainSize": 0.0, "GrainCount": 0, "DefectDensity": 0.0,
            "Frac_W": 0.0, "Frac_Re": 0.0, "Frac_C": 0.0,
            "C_boundary_frac": 0.0, "Re_boundary_frac": 0.0,
            "Defect_voxel_count": 0, "Defect_voxel_frac": 0.0,
            "Grain_d50_um": 0.0, "Grain_d90_um": 0.0,
            "VOXEL_SIZE_m": voxel_size, "RANDOM_SEED": rng_seed,
        }
    dims = (cl["bbox"][:, 3:6] - cl["bbox"][:, 0:3] + 1).astype(np.int64)
    ars = [float(lo_hi[1]) / float(max(lo_hi[0], 1)) for lo_hi in zip(dims.min(axis=1).tolist(), dims.max(axis=1).tolist())]
    sizes = cl["size"].tolist()
    volume = n_voxels * (voxel_size ** 3)
    d50, d90 = equivalent_diameter_um(cl["labels"], voxel_size)
    return {
        "AspectRatio": np.mean(ars),
        "EquiaxedFraction": np.mean(np.array(ars) < CET_AR_THRESHOLD),
        "NucleationDensity": n / volume if volume > 0 else 0.0,
        "AvgGrainSize": np.mean(sizes) * voxel_size * 1e6,
        "GrainCount": n,
        "DefectDensity": defects_count / volume if volume > 0 else 0.0,
        "Frac_W": 0.0, "Frac_Re": 0.0, "Frac_C": 0.0,
        "C_boundary_frac": 0.0, "Re_boundary_frac": 0.0,
        "Defect_voxel_count": defects_count,
        "Defect_voxel_frac": compute_voxel_fraction(defects_count, n_voxels),
        "Grain_d50_um": d50, "Grain_d90_um": d90,
        "VOXEL_SIZE_m": voxel_size, "RANDOM_SEED": rng_seed,
    }


def compute_CET(state, theta, phi, voxel_size=VOXEL_SIZE):
    m = compute_metrics(state, theta, phi, voxel_size=voxel_size)
    ok = m["AspectRatio"] < CET_AR_THRESHOLD and m["EquiaxedFraction"] > CET_EQ_THRESHOLD
    return "Equiaxed" if ok else "Columnar"


def detect_CET_transition(metrics_dict):
    return (metrics_dict["AspectRatio"] < CET_AR_THRESHOLD and
            metrics_dict["EquiaxedFraction"] > CET_EQ_THRESHOLD)
