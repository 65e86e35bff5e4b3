"""The reference-named drop-in modules (kmc_simulation / kmc_event_rates / thermal_solver /
run_simulation), which call the HIP engine through the C ABI, against reference outputs."""
import glob
import io
import os
import random

import numpy as np
import pandas as pd
import pytest

from helpers import GOLDEN, load, relerr

pytestmark = pytest.mark.gpu

TRAJ = sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN, "traj_*.npz")))
EVENTS = sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN, "events_*.npz")))


@pytest.mark.parametrize("incremental", [True, False])
@pytest.mark.parametrize("name", TRAJ)
def test_run_kmc_matches_reference(name, incremental, tmp_path, monkeypatch, capsys):
    """kmc_simulation.run_kmc: same return tuple, metrics.csv, prints and RNG end state as the
    reference run the fixture was recorded from (kmc_simulation.py:203-398)."""
    import kmc_simulation
    z = load(name)
    monkeypatch.chdir(tmp_path)
    kw = dict(L=int(z["L"]), n_steps=int(z["n_steps"]), temp=float(z["temp"]), defect_fraction=float(z["defect_fraction"]),
              n_seeds=int(z["n_seeds"]), impurity_c=float(z["impurity_c"]), output_prefix=name)
    if float(z["temp"]) == int(z["temp"]):
        kw["temp"] = int(z["temp"])
    if int(z["L"]) >= 30 and not incremental:
        pytest.skip("large cases run once (default mode)")
    state, atom_type, total_time, theta, phi = kmc_simulation.run_kmc(**kw, incremental=incremental)
    assert state.dtype == np.int64 and theta.dtype == np.float64
    assert np.array_equal(state, z["final_state"]) and np.array_equal(atom_type, state)
    assert np.array_equal(theta, z["final_theta"]) and np.array_equal(phi, z["final_phi"])
    assert total_time == float(z["total_time"])
    assert np.array_equal(np.array([random.random() for _ in range(4)]), z["py_next"])
    assert np.array_equal(np.random.random(4), z["np_next"])
    got = pd.read_csv(os.path.join("outputs", name, "metrics.csv"))
    want = pd.read_csv(io.StringIO(str(z["metrics_csv"])))
    assert list(got.columns) == list(want.columns)
    assert len(got) == len(want)
    for col in want.columns:
        if want[col].dtype.kind == "f":
            assert np.allclose(got[col].values, want[col].values, rtol=1e-12, atol=0), col
        else:
            assert got[col].tolist() == want[col].tolist(), col
    out = capsys.readouterr().out.strip().splitlines()
    ref = str(z["stdout"]).strip().splitlines()
    assert out == ref
    tag = name.split("_")[-1]
    assert os.path.exists(os.path.join("outputs", name, f"metrics_{tag}.csv"))   # plot_cet.py glob contract


@pytest.mark.parametrize("name", EVENTS)
def test_get_event_rates_list_form(name):
    import kmc_event_rates
    z = load(name)
    L = int(z["L"])
    np.random.seed(int(z["np_seed"]) if "np_seed" in z.files else 0)
    if "u_dep" in z.files:
        pytest.skip("fixture drew its species from a mid-run stream")
    st = z["state"].astype(np.int64)
    ev = kmc_event_rates.get_event_rates(st, z["theta"], z["phi"], z["T"], st.copy(), z["defects"].astype(np.int64), L,
                                         1, 2, 3, step=0, debug_step=1000, impurity_c=float(z["impurity_c"]))
    names = (b"dep", b"diff", b"nuc", b"att")
    assert len(ev) == len(z["rate"])
    for m, e in enumerate(ev):
        assert isinstance(e, tuple) and len(e) == 5
        assert e[0] == names[int(z["etype"][m])]
        assert tuple(e[1]) == tuple(int(x) for x in z["pos"][m])
        assert tuple(e[3]) == tuple(int(x) for x in z["target"][m])
        assert e[4] == int(z["atom"][m])
    if ev:
        assert relerr(np.array([e[2] for e in ev]), z["rate"]).max() <= 1e-11
    with pytest.raises(ValueError):
        kmc_event_rates.get_event_rates(st, z["theta"], z["phi"], z["T"], st, st * 0, L, 3, 2, 1)


def test_thermal_solver_functions():
    import thermal_solver
    z = load("thermal")
    for L in (2, 7, 16):
        Tin = z[f"cet_rand_L{L}_in"]
        out = thermal_solver.update_temperature_cet(Tin, np.zeros((L, L, L), int), dt=1e-6)
        assert np.array_equal(out, z[f"cet_rand_L{L}_out"]) and out is not Tin
    for L in (8, 13, 16):
        key = f"laser_L{L}_dt1e-06"
        dt, i0, j0, P, rb, ab = z[key + "_par"]
        out = thermal_solver.update_temperature(z[key + "_T"], z[key + "_cur"].astype(np.int64), z[key + "_prev"].astype(np.int64),
                                                dt, (i0, j0), P, rb, ab)
        assert np.allclose(out, z[key + "_out"], rtol=1e-12, atol=0)
    T = thermal_solver.build_temperature_field(5)
    assert T.shape == (5, 5, 5) and T[0, 0, 0] == 2800 and T[4, 2, 1] == 3695


def test_run_simulation_kmc_plumbing(tmp_path, monkeypatch, capsys):
    """BASELINE config 1: 32^3 lattice, one KMC step through the run_simulation.py entry point."""
    import importlib
    monkeypatch.chdir(tmp_path)
    mod = importlib.import_module("run_simulation")
    z = load("traj_L32_n1")
    state, atom_type, total_time, theta, phi = mod.kmc_plumbing(32)
    assert np.array_equal(state, z["final_state"]) and total_time == float(z["total_time"])
    assert "Completed 1 steps" in capsys.readouterr().out


def test_rccl_single_rank_mode():
    """cetkmc_create_rank with a real RCCL communicator of size 1: the collective code path
    (all-gathers of block sums / event records) must reproduce the plain engine bit for bit."""
    import cetkmc
    from helpers import random_lattice
    L = 16
    state, theta, phi, T, defects = random_lattice(L, 21, fill=0.3)
    rs = np.random.RandomState(3)
    n = 40
    u_pick, u_def, u_np = rs.random_sample(n), rs.random_sample(n), rs.random_sample(n * (L * L + 2))
    outs = []
    for mode in ("plain", "rank", "rank+overlap", "rank+incremental"):
        if mode == "plain":
            e = cetkmc.Engine(L, impurity_c=0.2)
        else:
            e = cetkmc.Engine(L, impurity_c=0.2, rank=0, nranks=1, unique_id=cetkmc.Engine.unique_id())
            # patterned all-gather + a send/recv to itself through RCCL (the neighbour-exchange calls of N > 1), data checked
            t = e.comm_selftest(4096)
            assert t["allgather_us"] > 0
        if mode == "rank+overlap":
            e.set_option("interface_every_step", 1)
        e.upload(state, theta, phi, T, defects)
        res = e.run_steps(0, n, 0.05, u_pick, u_def, u_np, rng_mode=0, thermal_mode=1, incremental=(mode == "rank+incremental"))
        outs.append((res["totals"].copy(), res["events"].tobytes(), e.download()))
        e.close()
    for o in outs[1:]:
        assert np.array_equal(outs[0][0], o[0]) and outs[0][1] == o[1]
        for k in outs[0][2]:
            assert np.array_equal(outs[0][2][k], o[2][k])


def test_checkpoint_resume_is_bit_identical(tmp_path, monkeypatch):
    """a run interrupted at a checkpoint and resumed equals the uninterrupted run
    (lattice, orientations, time, metrics rows, both RNG streams)."""
    import kmc_simulation
    monkeypatch.chdir(tmp_path)
    kw = dict(L=10, temp=2800, defect_fraction=0.01, n_seeds=6, impurity_c=0.2)
    full = kmc_simulation.run_kmc(n_steps=450, output_prefix="full", **kw)
    py_end, np_end = random.random(), np.random.random()
    kmc_simulation.run_kmc(n_steps=131, output_prefix="part", checkpoint_every=130, **kw)   # leaves the checkpoint of step 130
    part = kmc_simulation.run_kmc(n_steps=450, output_prefix="part", resume_from="outputs/part/checkpoint.npz", **kw)
    assert (random.random(), np.random.random()) == (py_end, np_end)
    for a, b in zip(full, part):
        assert np.array_equal(a, b)
    z = load("traj_L10_n450")
    assert np.array_equal(part[0], z["final_state"]) and part[2] == float(z["total_time"])
    a = pd.read_csv("outputs/full/metrics.csv")
    b = pd.read_csv("outputs/part/metrics.csv")
    assert a.equals(b)


def test_gv_sweep_driver(tmp_path, monkeypatch, capsys):
    """gv_sweep.py (BASELINE config 5 driver): substrate temperature and deposition frequency reach the engine
    (G and R columns follow them, deposition events follow nu_dep), one map row per (T_sub, nu_dep)."""
    import constants
    import gv_sweep
    monkeypatch.chdir(tmp_path)
    df = gv_sweep.gv_sweep(L=10, n_steps=45, temps=(2800.0, 3300.0), nu_deps=(2e13, 2e15), carbon=0.2)
    capsys.readouterr()
    assert len(df) == 4 and os.path.exists("outputs/gv_sweep/gv_map.csv")
    assert sorted(set(df["V_m_per_s"])) == [2e13 * constants.ATOMIC_SPACING_W, 2e15 * constants.ATOMIC_SPACING_W]
    m = pd.read_csv("outputs/gv_sweep/T3300_V2e+15_c_20/metrics.csv")
    assert np.allclose(m["R_phys"], 2e15 * constants.ATOMIC_SPACING_W)
    # default nu_dep reproduces the plain run_kmc of the same arguments
    import kmc_simulation
    a = kmc_simulation.run_kmc(L=10, n_steps=45, temp=2800.0, defect_fraction=constants.DEFECT_PROB, n_seeds=constants.N_SEEDS,
                               impurity_c=0.2, output_prefix="plain")
    b = pd.read_csv("outputs/gv_sweep/T2800_V2e+13_c_20/metrics.csv")
    assert b.equals(pd.read_csv("outputs/plain/metrics.csv"))
    assert a[0].shape == (10, 10, 10)
