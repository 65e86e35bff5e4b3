"""Initial lattice for the CET runs (drop-in for the reference ``lattice_init.py``).

Host-side, O(L^3) once per run; the interface and the RNG call sequence are kept
verbatim because everything downstream (seed positions, species, orientations and the
position of NumPy's global stream) is part of the trajectory (lattice_init.py:10-59).
"""
import numpy as np

from constants import IMPURITY_RE, LATTICE_SIZE, N_SEEDS, STATES, T_MELT, T_SUB
from defects import introduce_defects  # noqa: F401  (re-exported like the reference, lattice_init.py:8)


def initialize_lattice(lattice_size=LATTICE_SIZE, n_seeds=N_SEEDS, T_sub=T_SUB, T_melt=T_MELT,
                       random_seed=42, impurity_c=0.0, verbose=False):
    """Returns (state, orientation_theta, orientation_phi, T, atom_type).

    Empty int lattice; T ramps linearly along AXIS 2 with slope (T_melt-T_sub)/L;
    ``n_seeds`` nuclei on the plane k=0 at (idx//L, idx%L) with species
    Re if u<0.10, C if u<0.10+impurity_c, else W, and random orientation.
    """
    np.random.seed(random_seed)
    L = lattice_size
    shape = (L, L, L)
    state = np.full(shape, STATES["Empty"], dtype=int)
    atom_type = np.full(shape, STATES["Empty"], dtype=int)
    theta = np.zeros(shape)
    phi = np.zeros(shape)

    slope = (T_melt - T_sub) / L
    column = T_sub + slope * np.arange(L)[np.newaxis, np.newaxis, :]
    T = np.repeat(column, L, axis=0).repeat(L, axis=1)

    picks = np.random.choice(L * L, n_seeds, replace=False)
    if verbose:
        print(f"Initializing {n_seeds} seeds with C={impurity_c}, Re={IMPURITY_RE}")
    for n, (x, y) in enumerate(zip(picks // L, picks % L)):
        u = np.random.random()
        if u < IMPURITY_RE:
            species = STATES["Re"]
        elif u < (IMPURITY_RE + impurity_c):
            species = STATES["C"]
        else:
            species = STATES["W"]
        state[x, y, 0] = species
        atom_type[x, y, 0] = species
        theta[x, y, 0] = np.random.uniform(0, np.pi)
        phi[x, y, 0] = np.random.uniform(0, 2 * np.pi)
        if verbose:
            print(f"Seed {n} at ({x}, {y}, 0): rand={u:.3f}, atom={species}")
    return state, theta, phi, T, atom_type


def visualize_initial_seeds(state, atom_type, title="Initial Nucleation Sites", filename="initial_seeds.png"):
    """3-D scatter of the occupied sites, one colour per species (lattice_init.py:61-96)."""
    import matplotlib
    matplotlib.use("Agg", force=False)
    import matplotlib.pyplot as plt

    style = {STATES["W"]: ("grey", "Tungsten"), STATES["Re"]: ("blue", "Rhenium"),
             STATES["C"]: ("red", "Carbon"), STATES["Defect"]: ("black", "Defect")}
    fig = plt.figure()
    ax = fig.add_subplot(111, projection="3d")
    for species, (colour, label) in style.items():
        ii, jj, kk = np.where(state == species)
        if ii.size:
            ax.scatter(kk, jj, ii, c=colour, label=label, alpha=0.6, s=10)
    ax.set(xlabel="X", ylabel="Y", zlabel="Z (Build Direction)", title=title)
    ax.legend()
    fig.tight_layout()
    fig.savefig(filename, dpi=150)
    plt.close(fig)


def save_lattice(state, orientation_theta, orientation_phi, T, atom_type, prefix="init"):
    """Five ``.npy`` files ``{prefix}_{state,orientation_theta,orientation_phi,temperature,atom_type}``
    (lattice_init.py:98-105)."""
    for suffix, arr in (("state", state), ("orientation_theta", orientation_theta),
                        ("orientation_phi", orientation_phi), ("temperature", T), ("atom_type", atom_type)):
        np.save(f"{prefix}_{suffix}.npy", arr)
