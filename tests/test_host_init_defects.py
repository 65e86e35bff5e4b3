"""Host-side drop-ins lattice_init / defects vs reference outputs (fixture F4/F5)."""
import numpy as np
import pytest

from helpers import load

import defects
import lattice_init

CASES = [(8, 3, 0.1, 2800), (12, 5, 0.1, 2800), (16, 5, 0.2, 2800), (30, 20, 0.2, 2800),
         (10, 6, 0.2, 2800), (7, 4, 0.3, 3400), (32, 5, 0.0, 2800), (5, 25, 0.5, 3000)]


@pytest.mark.parametrize("L,ns,c,tsub", CASES)
def test_initialize_lattice_and_first_defect_mask(L, ns, c, tsub):
    z = load("init_defects")
    key = f"init_L{L}_s{ns}_c{c}_t{tsub}"
    state, theta, phi, T, atom = lattice_init.initialize_lattice(lattice_size=L, n_seeds=ns, T_sub=tsub, impurity_c=c)
    assert state.dtype == np.int64 and T.dtype == np.float64 and state.shape == (L, L, L)
    assert np.array_equal(state, z[key + "_state"]) and np.array_equal(atom, state)
    assert np.array_equal(theta, z[key + "_theta"]) and np.array_equal(phi, z[key + "_phi"])
    assert np.array_equal(T, z[key + "_T"])
    mask, dens = defects.introduce_defects(state, atom, T, apply_to_state=False)
    assert np.array_equal(mask, z[key + "_defects"])
    assert dens == float(z[key + "_density"])


def test_track_defects_variants():
    z = load("init_defects")
    atom = z["defects_L9_atom"].astype(np.int64)
    T = z["defects_L9_T"]
    np.random.seed(77)
    assert np.array_equal(defects.track_defects(atom, atom, 9, T), z["defects_L9_mask_T"])
    np.random.seed(78)
    assert np.array_equal(defects.track_defects(atom, atom, 9, None), z["defects_L9_mask_noT"])
    np.random.seed(79)
    st = atom.copy()
    m, dens = defects.introduce_defects(st, atom, T, apply_to_state=True)
    assert np.array_equal(st, z["defects_L9_applied_state"])
    assert dens == float(z["defects_L9_applied_density"])
    assert defects.track_defects(atom, atom, 0).shape == (0, 0, 0)


def test_save_lattice_roundtrip(tmp_path):
    state, theta, phi, T, atom = lattice_init.initialize_lattice(lattice_size=5, n_seeds=3, impurity_c=0.2)
    prefix = str(tmp_path / "init")
    lattice_init.save_lattice(state, theta, phi, T, atom, prefix=prefix)
    for suffix, arr in (("state", state), ("orientation_theta", theta), ("orientation_phi", phi),
                        ("temperature", T), ("atom_type", atom)):
        assert np.array_equal(np.load(f"{prefix}_{suffix}.npy"), arr)
