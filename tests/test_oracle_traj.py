"""Oracle vs reference run_kmc trajectories (fixture F2), step by step: event counts,
totals, the voxels each step touches, thermal snapshots, defect masks, final arrays,
total_time and the final position of both RNG streams."""
import glob
import os

import numpy as np
import pytest

from helpers import GOLDEN, OracleBackend, load, replay_run_kmc

TRAJ = sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN, "traj_*.npz")))


@pytest.mark.parametrize("name", TRAJ)
def test_replay_matches_reference(oracle_mod, name):
    z = load(name)
    replay_run_kmc(z, lambda *a: OracleBackend(oracle_mod, *a))


@pytest.mark.parametrize("name", ["traj_L8_n60", "traj_L10_n450", "traj_L7_n230_T3400", "traj_L16_n100"])
def test_batched_loop_equals_stepwise(oracle_mod, name):
    """orc_run_steps (the batched protocol mirrored by the device) reproduces the same
    trajectory when fed the pre-drawn streams."""
    import random

    import defects as host_defects
    import lattice_init as host_init

    z = load(name)
    L, n_steps = int(z["L"]), int(z["n_steps"])
    temp, df = float(z["temp"]), float(z["defect_fraction"])
    n_seeds, c = int(z["n_seeds"]), float(z["impurity_c"])
    np.random.seed(42)
    random.seed(42)
    state, theta, phi, T, atom = host_init.initialize_lattice(lattice_size=L, n_seeds=n_seeds, T_sub=temp, impurity_c=c)
    mask, _ = host_defects.introduce_defects(state, atom, T, apply_to_state=False)
    lat = oracle_mod.Lattice(state, theta, phi, T, mask, impurity_c=c)
    step = 0
    total_time = 0.0
    while step < n_steps:
        # batches end right after a step with step % 200 == 0 (defect-mask refresh point)
        end = min(n_steps, (step // 200) * 200 + 1 if step % 200 == 0 else ((step // 200) + 1) * 200 + 1)
        n = end - step
        py_state = random.getstate()
        per = 3 if df > 0 else 2
        draws = np.array([random.random() for _ in range(per * n)]).reshape(n, per)
        u_pick = draws[:, 0]
        u_def = draws[:, 1] if df > 0 else None
        u_dt = draws[:, -1]
        np_state = np.random.get_state()
        cap = n * (L * L + 2)
        u_np = np.random.random(cap)
        res = lat.run_steps(step, n, df, u_pick, u_def, u_np, rng_mode=0, thermal_mode=1)
        assert res["status"] == 0 and res["done"] == n
        np.random.set_state(np_state)
        np.random.random(res["np_used"])
        for s in range(n):
            total_time += max(-np.log(max(1e-12, u_dt[s])) / res["totals"][s], 1e-12)
        step = end
        if (step - 1) % 200 == 0:
            s_now = lat.state.astype(np.int64)
            m, _ = host_defects.introduce_defects(s_now, s_now, lat.T, apply_to_state=False)
            lat.defects = m.astype(np.int8)
    assert np.array_equal(lat.state, z["final_state"])
    assert np.array_equal(lat.theta, z["final_theta"]) and np.array_equal(lat.phi, z["final_phi"])
    assert total_time == float(z["total_time"])
    assert np.array_equal(np.random.random(4), z["np_next"])
    assert np.array_equal(np.array([random.random() for _ in range(4)]), z["py_next"])
