"""Oracle thermal kernels vs the reference's thermal_solver outputs (fixture F3)."""
import numpy as np
import pytest

from helpers import load


@pytest.mark.parametrize("L", [1, 2, 3, 7, 16])
def test_thermal_cet_bit_exact(oracle_mod, L):
    z = load("thermal")
    Tin = z[f"cet_rand_L{L}_in"]
    zero = np.zeros((L, L, L), np.int8)
    for dt, key in ((1e-6, "out"), (3e-7, "out_dt3e-7")):
        lat = oracle_mod.Lattice(zero, Tin * 0, Tin * 0, Tin)
        got = lat.thermal_cet(dt=dt, scrub_nan=False)
        assert np.array_equal(got, z[f"cet_rand_L{L}_{key}"])


def test_thermal_cet_iterated_ramp_and_nan_scrub(oracle_mod):
    z = load("thermal")
    seq = z["cet_ramp_L12_seq"]
    zero = np.zeros(seq[0].shape, np.int8)
    lat = oracle_mod.Lattice(zero, seq[0] * 0, seq[0] * 0, seq[0])
    for n in range(1, len(seq)):
        got = lat.thermal_cet(dt=1e-6, scrub_nan=True)
        assert np.array_equal(got, seq[n]), n
    Tn = z["cet_nan_L6_in"]
    lat = oracle_mod.Lattice(np.zeros(Tn.shape, np.int8), Tn * 0, Tn * 0, Tn)
    assert np.array_equal(lat.thermal_cet(dt=1e-6, scrub_nan=True), z["cet_nan_L6_out"])


@pytest.mark.parametrize("L", [8, 13, 16])
def test_thermal_laser(oracle_mod, L):
    z = load("thermal")
    for tag in ("dt1e-06", "dt1e-09"):
        key = f"laser_L{L}_{tag}"
        dt, i0, j0, P, rb, ab = z[key + "_par"]
        T, prev, cur = z[key + "_T"], z[key + "_prev"], z[key + "_cur"]
        lat = oracle_mod.Lattice(cur, T * 0, T * 0, T)
        q = oracle_mod.laser_source_plane(L, (i0, j0), P, rb, ab)
        got = lat.thermal_laser(dt, q, prev_state=prev)
        want = z[key + "_out"]
        # the source plane carries NumPy's exp: allow 1e-12 rel in case the host CPU's SIMD
        # dispatch differs from the fixture machine; everything else is bit-exact
        assert np.allclose(got, want, rtol=1e-12, atol=0)
    key = f"laser_L{L}_nolatent"
    dt, i0, j0, P, rb, ab = z[key + "_par"]
    T, cur = z[f"laser_L{L}_dt1e-06_T"], z[f"laser_L{L}_dt1e-06_cur"]
    lat = oracle_mod.Lattice(cur, T * 0, T * 0, T)
    got = lat.thermal_laser(dt, oracle_mod.laser_source_plane(L, (i0, j0), P, rb, ab), prev_state=cur)
    assert np.allclose(got, z[key + "_out"], rtol=1e-12, atol=0)
