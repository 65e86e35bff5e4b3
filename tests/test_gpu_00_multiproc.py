"""Real multi-process runs of the per-rank device code path on ONE GPU: 2 and 3 ranks share device 0, each owns an
axis-0 slab (+2 halo planes), and the engine's three collectives (all-gather of block sums, all-gather of event
records, temperature-halo exchange) travel through the host-relay transport over gloo instead of RCCL
(cetkmc_create_rank_host).  Everything else -- kernels, owner selection, event application from the gathered
record, RNG cursors -- is exactly what runs under the RCCL communicator.  Result: bit-identical to the single-process
engine.  (First in collection order on purpose: the parent spawns its workers before anything initialises the GPU
in this process.)"""
import os
import socket
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _inputs(L, n, seed):
    sys.path.insert(0, HERE)
    from helpers import random_lattice
    # (at 128^2 top-plane candidates the 5 % melt-hot voxels of the generator make the deposition rates sum to inf, which
    # ends a run at its first step -- kmc_simulation.py:260-262; the large case keeps every temperature below the melting point)
    fields = random_lattice(L, seed, fill=0.25, hot_frac=0.0 if L >= 128 else 0.05)
    rs = np.random.RandomState(seed + 1)
    return fields, (rs.random_sample(n), rs.random_sample(n), rs.random_sample(n * (L * L + 2)))


def _run(eng, a0, a1, fields, streams, n, mode):
    state, theta, phi, T, defects = fields
    u_pick, u_def, u_np = streams
    eng.upload_planes(a0, a1, state[a0:a1], theta[a0:a1], phi[a0:a1], T[a0:a1], defects[a0:a1])
    kw = dict(rng_mode=0, thermal_mode=1, incremental=(mode == "incremental"))
    if mode == "laser":
        from cetkmc import synthetic
        eng.set_prev_state(None)
        kw = dict(rng_mode=1, seed=9, thermal_mode=2, q_planes=synthetic.laser_planes(eng.L, 0, n))
    r = eng.run_steps(0, n, 0.05, u_pick, u_def, u_np, **kw)
    assert r["done"] == n and r["status"] == 0, r
    d = eng.download_planes(eng.i0, eng.i1, state=True, theta=True, phi=True, T=True, defects=True)
    info = eng.rate_sweep()
    return r, d, info


def _worker(rank, world, port, L, n, seed, mode, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    root = os.path.dirname(HERE)
    sys.path.insert(0, os.path.join(root, "cet-driven-simulation-for-3d-printing-am-kmc-approach_amd"))
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import cetkmc
    from cetkmc import host_transport
    fields, streams = _inputs(L, n, seed)
    eng = cetkmc.Engine(L, impurity_c=0.2, device=0, rank=rank, nranks=world, host_comm=host_transport.torch_callbacks())
    a0, a1 = max(0, eng.i0 - 2), min(L, eng.i1 + 2)
    for nbytes in (64, 5000):       # transport self-test: patterned all-gather + neighbour exchange, checked by the library
        t = eng.comm_selftest(nbytes)
        assert t["allgather_us"] > 0 and t["exchange_us"] > 0
    r, d, info = _run(eng, a0, a1, fields, streams, n, mode)
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), totals=r["totals"], events=r["events"], n_events=r["n_events"],
             np_used=r["np_used"], i0=eng.i0, i1=eng.i1, info=np.array(info, dtype=np.float64), **d)
    eng.close()
    dist.barrier()
    if rank == 0:       # the single-process engine on the same inputs
        ref = cetkmc.Engine(L, impurity_c=0.2)
        r, d, info = _run(ref, 0, L, fields, streams, n, mode)
        np.savez(os.path.join(out_dir, "ref.npz"), totals=r["totals"], events=r["events"], n_events=r["n_events"],
                 np_used=r["np_used"], info=np.array(info, dtype=np.float64), **d)
        ref.close()
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,L,mode", [(2, 16, "full"), (3, 24, "full"), (2, 20, "incremental"), (2, 16, "laser"),
                                          (4, 64, "laser"), (4, 128, "laser"), (4, 128, "incremental")])
def test_ranks_sharing_one_gpu_match_single_process(world, L, mode, tmp_path):
    """(4, 128, ...): BASELINE config 4's rank count on a lattice of config-2 size -- four real processes, each with a
    32-plane slab (+halo), all three collectives per step."""
    import torch.multiprocessing as mp
    n = 70
    mp.spawn(_worker, args=(world, _free_port(), L, n, 31, mode, str(tmp_path)), nprocs=world, join=True)
    ref = np.load(tmp_path / "ref.npz")
    for rank in range(world):
        z = np.load(tmp_path / f"rank{rank}.npz")
        # every rank logs the same totals / events / counts / stream position as the undivided run
        assert np.array_equal(z["totals"], ref["totals"])
        assert z["events"].tobytes() == ref["events"].tobytes()
        assert np.array_equal(z["n_events"], ref["n_events"]) and z["np_used"] == ref["np_used"]
        assert np.array_equal(z["info"], ref["info"])
        i0, i1 = int(z["i0"]), int(z["i1"])
        for k in ("state", "theta", "phi", "T", "defects"):
            assert np.array_equal(z[k], ref[k][i0:i1]), (rank, k)


# ---- Mode B (super-steps) across ranks: boxes sharded with the slabs, boundary-layer events exchanged ----------------
def _run_b(eng, a0, a1, fields, n, box, thermal_mode, null_events=False):
    from cetkmc import synthetic
    state, theta, phi, T, defects = fields
    eng.upload_planes(a0, a1, state[a0:a1], theta[a0:a1], phi[a0:a1], T[a0:a1], defects[a0:a1])
    eng.set_prev_state(None)
    q = synthetic.laser_planes(eng.L, 0, n) if thermal_mode == 2 else None
    r = eng.run_supersteps(0, n, box, 0.05, seed=9, thermal_mode=thermal_mode, q_planes=q, want_events=True,
                           null_events=null_events)
    assert r["done"] == n and r["status"] == 0, r
    d = eng.download_planes(eng.i0, eng.i1, state=True, theta=True, phi=True, T=True, defects=True)
    info = eng.rate_sweep()
    # a Mode A batch afterwards: the interface list / rate table Mode B leaves behind must be complete
    rs = np.random.RandomState(5)
    m = 12
    ra = eng.run_steps(n, m, 0.0, rs.random_sample(m), None, rs.random_sample(2 * m + 2), rng_mode=1, seed=3, thermal_mode=1)
    return r, d, info, ra


def _worker_b(rank, world, port, L, n, box, thermal_mode, out_dir, null_events=False):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    root = os.path.dirname(HERE)
    sys.path.insert(0, os.path.join(root, "cet-driven-simulation-for-3d-printing-am-kmc-approach_amd"))
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import cetkmc
    from cetkmc import host_transport
    fields, _ = _inputs(L, 1, 77)
    eng = cetkmc.Engine(L, impurity_c=0.2, device=0, rank=rank, nranks=world, host_comm=host_transport.torch_callbacks())
    a0, a1 = max(0, eng.i0 - 2), min(L, eng.i1 + 2)
    r, d, info, ra = _run_b(eng, a0, a1, fields, n, box, thermal_mode, null_events)
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), totals=r["totals"], events=r["events"], n_exec=r["n_exec"],
             nuc=r["nucleation_count"], i0=eng.i0, i1=eng.i1, info=np.array(info, dtype=np.float64),
             a_totals=ra["totals"], a_events=ra["events"], dt_event=r["dt_event"], **d)
    eng.close()
    dist.barrier()
    if rank == 0:
        ref = cetkmc.Engine(L, impurity_c=0.2)
        r, d, info, ra = _run_b(ref, 0, L, fields, n, box, thermal_mode, null_events)
        np.savez(os.path.join(out_dir, "ref.npz"), totals=r["totals"], events=r["events"], n_exec=r["n_exec"],
                 nuc=r["nucleation_count"], info=np.array(info, dtype=np.float64), a_totals=ra["totals"], a_events=ra["events"],
                 dt_event=r["dt_event"], **d)
        ref.close()
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,L,box,thermal_mode,null_events", [(2, 32, 8, 1, False), (4, 64, 8, 2, False), (2, 64, 16, 2, False),
                                                                  (3, 48, 8, 1, False), (2, 32, 8, 1, True), (3, 48, 8, 2, True),
                                                                  (4, 128, 8, 0, True)])
def test_mode_b_ranks_sharing_one_gpu_match_single_process(world, L, box, thermal_mode, null_events, tmp_path):
    """cetkmc_run_supersteps with one slab per rank (boxes sharded with the slabs, block sums all-gathered, the events
    of the boundary box layers exchanged with the neighbour ranks every super-step) reproduces the single-process
    run bit for bit: per-box events, executed counts, totals, every field, and the Mode A batch that follows.
    null_events: the acceptance test needs the largest window total over ALL ranks' boxes (one more 8-byte all-gather
    per super-step); the per-event time increments are the same on every rank."""
    import torch.multiprocessing as mp
    n = 30
    mp.spawn(_worker_b, args=(world, _free_port(), L, n, box, thermal_mode, str(tmp_path), null_events), nprocs=world, join=True)
    ref = np.load(tmp_path / "ref.npz")
    zs = [np.load(tmp_path / f"rank{rank}.npz") for rank in range(world)]
    ev = np.concatenate([z["events"] for z in zs], axis=1)          # global box order = rank order
    assert ev.shape == ref["events"].shape and ev.tobytes() == ref["events"].tobytes()
    assert np.array_equal(sum(z["n_exec"] for z in zs), ref["n_exec"])
    assert sum(int(z["nuc"]) for z in zs) == int(ref["nuc"])
    if null_events:
        assert (ref["events"]["type"] == -2).sum() > 0
    for z in zs:
        assert np.array_equal(z["totals"], ref["totals"]) and np.array_equal(z["info"], ref["info"])
        assert np.array_equal(z["dt_event"], ref["dt_event"])
        assert np.array_equal(z["a_totals"], ref["a_totals"]) and z["a_events"].tobytes() == ref["a_events"].tobytes()
        i0, i1 = int(z["i0"]), int(z["i1"])
        for k in ("state", "theta", "phi", "T", "defects"):
            assert np.array_equal(z[k], ref[k][i0:i1]), k


# ---- the same with the RCCL communicator, one GPU per rank: collected everywhere, runs where >= 2 GPUs are visible ----
def _worker_rccl(rank, world, port, L, n, seed, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    root = os.path.dirname(HERE)
    sys.path.insert(0, os.path.join(root, "cet-driven-simulation-for-3d-printing-am-kmc-approach_amd"))
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import cetkmc
    box = [cetkmc.Engine.unique_id() if rank == 0 else None]
    dist.broadcast_object_list(box, src=0)
    fields, streams = _inputs(L, n, seed)
    eng = cetkmc.Engine(L, impurity_c=0.2, device=rank, rank=rank, nranks=world, unique_id=box[0])
    a0, a1 = max(0, eng.i0 - 2), min(L, eng.i1 + 2)
    eng.comm_selftest(64)
    eng.comm_selftest(2 * L * L * 8)
    r, d, info = _run(eng, a0, a1, fields, streams, n, "full")
    rb = eng.run_supersteps(n, 16, 8, 0.05, seed=9, thermal_mode=1, want_events=True)
    db = eng.download_planes(eng.i0, eng.i1, state=True, theta=True)
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), totals=r["totals"], events=r["events"], i0=eng.i0, i1=eng.i1,
             b_events=rb["events"], b_state=db["state"], b_theta=db["theta"], **d)
    eng.close()
    dist.barrier()
    if rank == 0:
        ref = cetkmc.Engine(L, impurity_c=0.2, device=0)
        r, d, info = _run(ref, 0, L, fields, streams, n, "full")
        rb = ref.run_supersteps(n, 16, 8, 0.05, seed=9, thermal_mode=1, want_events=True)
        db = ref.download_planes(0, L, state=True, theta=True)
        np.savez(os.path.join(out_dir, "ref.npz"), totals=r["totals"], events=r["events"], b_events=rb["events"],
                 b_state=db["state"], b_theta=db["theta"], **d)
        ref.close()
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_over_rccl_match_single_process(tmp_path):
    """Mode A batch + Mode B super-steps with a real 2-rank RCCL communicator (one GPU per rank: ncclAllGather of the
    block sums / event records, ncclSend/ncclRecv of the temperature halo and of the boundary box layers' events):
    bit-identical to the single-process engine.  The build box and the round's test box have ONE GPU: skipped there
    (N > 1 over RCCL has not been executed by the builder; the same per-rank code runs above over the host relay)."""
    import cetkmc
    if cetkmc.device_count() < 2:
        pytest.skip("needs >= 2 GPUs")
    import torch.multiprocessing as mp
    world, L, n = 2, 32, 40
    mp.spawn(_worker_rccl, args=(world, _free_port(), L, n, 31, str(tmp_path)), nprocs=world, join=True)
    ref = np.load(tmp_path / "ref.npz")
    zs = [np.load(tmp_path / f"rank{rank}.npz") for rank in range(world)]
    assert np.concatenate([z["b_events"] for z in zs], axis=1).tobytes() == ref["b_events"].tobytes()
    for z in zs:
        assert np.array_equal(z["totals"], ref["totals"]) and z["events"].tobytes() == ref["events"].tobytes()
        i0, i1 = int(z["i0"]), int(z["i1"])
        for k in ("state", "theta", "phi", "T", "defects"):
            assert np.array_equal(z[k], ref[k][i0:i1]), k
        assert np.array_equal(z["b_state"], ref["b_state"][i0:i1]) and np.array_equal(z["b_theta"], ref["b_theta"][i0:i1])


# ---- defects.track_defects across ranks (config 5's refresh): every rank draws the same seeded stream and keeps its slice ----
def _worker_defects(rank, world, port, L, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    root = os.path.dirname(HERE)
    sys.path.insert(0, os.path.join(root, "cet-driven-simulation-for-3d-printing-am-kmc-approach_amd"))
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import cetkmc
    import defects as defects_mod
    from cetkmc import host_transport
    fields, _ = _inputs(L, 1, 13)
    state, theta, phi, T, dmask = fields

    def rank_counts(n_mine):
        box = [None] * world
        dist.all_gather_object(box, int(n_mine))
        return box

    eng = cetkmc.Engine(L, impurity_c=0.2, device=0, rank=rank, nranks=world, host_comm=host_transport.torch_callbacks())
    a0, a1 = max(0, eng.i0 - 2), min(L, eng.i1 + 2)
    eng.upload_planes(a0, a1, state[a0:a1], theta[a0:a1], phi[a0:a1], T[a0:a1], dmask[a0:a1])
    np.random.seed(99)
    n_flag, _ = defects_mod.refresh_defects_device(eng, rank_counts=rank_counts, rank=rank)
    after = np.random.random()                       # the stream position afterwards is the same on every rank
    d = eng.download_planes(eng.i0, eng.i1, state=False, defects=True)["defects"]
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), defects=d, n_flag=n_flag, after=after, i0=eng.i0, i1=eng.i1)
    eng.close()
    dist.barrier()
    if rank == 0:       # host reference: defects.track_defects on the whole lattice, same seed
        np.random.seed(99)
        mask = defects_mod.track_defects(state, state, L, T)
        np.savez(os.path.join(out_dir, "ref.npz"), defects=mask, after=np.random.random())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,L", [(2, 16), (3, 24)])
def test_defect_refresh_across_ranks_matches_host(world, L, tmp_path):
    """refresh_defects_device(rank_counts=...) on slabs == defects.track_defects on the whole lattice (defects.py:4-19): same
    flagged sites, same NumPy stream position afterwards on every rank."""
    import torch.multiprocessing as mp
    mp.spawn(_worker_defects, args=(world, _free_port(), L, str(tmp_path)), nprocs=world, join=True)
    ref = np.load(tmp_path / "ref.npz")
    total = 0
    for rank in range(world):
        z = np.load(tmp_path / f"rank{rank}.npz")
        i0, i1 = int(z["i0"]), int(z["i1"])
        assert np.array_equal(z["defects"] != 0, ref["defects"][i0:i1] != 0), rank
        assert float(z["after"]) == float(ref["after"])
        total += int(z["n_flag"])
    assert total == int((ref["defects"] != 0).sum())
