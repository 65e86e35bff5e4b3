"""World-size-2 (gloo, CPU) rehearsal of the multi-GPU protocol of DESIGN.md "Multi-GPU":
axis-0 slabs with a 2-plane halo; per step an all-gather of the owned (plane,category) block
sums, every rank forms the same canonical total and picks the owner, the owner selects, the
chosen event record is all-gathered and applied by every rank whose slab+halo holds the touched
voxels; after each thermal update the two boundary T planes are exchanged.  Each rank's copy of
the lattice is POISONED outside its slab+halo, so any read beyond the halo changes the result.
The per-slab compute engine here is the CPU oracle (test infrastructure); the HIP library runs
the same protocol with RCCL (cetkmc_hip.hip: launch_sweep / launch_select / exchange_T_halo).
The distributed run must reproduce the single-domain oracle run bit for bit.
"""
import os
import socket

import numpy as np
import pytest

from helpers import random_lattice

torch = pytest.importorskip("torch")
import torch.distributed as dist  # noqa: E402
import torch.multiprocessing as mp  # noqa: E402


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _slab(L, rank, world):
    n = L // world
    return rank * n, (rank + 1) * n


def _worker(rank, world, port, L, n_steps, seed, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import oracle
    state, theta, phi, T, defects = random_lattice(L, seed, fill=0.25)
    i0, i1 = _slab(L, rank, world)
    a, b = max(0, i0 - 2), min(L, i1 + 2)
    rs = np.random.RandomState(1000 + rank)
    poison = np.ones(L, bool)
    poison[a:b] = False
    state = state.copy(); theta = theta.copy(); phi = phi.copy(); T = T.copy()
    state[poison] = rs.randint(0, 5, state[poison].shape)
    theta[poison] = 99.0
    phi[poison] = -99.0
    T[poison] = np.nan
    lat = oracle.Lattice(state, theta, phi, T, defects, impurity_c=0.2)
    streams = np.random.RandomState(7)
    u_pick, u_def = streams.random_sample(n_steps), streams.random_sample(n_steps)
    u_np = streams.random_sample(n_steps * (L * L + 2))
    pos = 0
    rowsum = np.zeros((L, 3, L)); rowcnt = np.zeros((L, 3, L), np.int32)
    log = []
    for step in range(n_steps):
        if step % 20 == 0:
            lat.T[poison] = np.nan                                   # keep the poison alive
            full = lat.thermal_cet(1e-6, scrub_nan=False)            # planes next to poison are garbage ...
            # ... so only the owned planes are kept, and the halo comes from the neighbours
            newT = np.full_like(full, np.nan)
            newT[i0:i1] = full[i0:i1]
            reqs = []
            if rank > 0:
                reqs.append(dist.isend(torch.from_numpy(np.ascontiguousarray(newT[i0:i0 + 2])), rank - 1))
                lo = torch.empty((2, L, L), dtype=torch.float64); reqs.append(dist.irecv(lo, rank - 1))
            if rank < world - 1:
                reqs.append(dist.isend(torch.from_numpy(np.ascontiguousarray(newT[i1 - 2:i1])), rank + 1))
                hi = torch.empty((2, L, L), dtype=torch.float64); reqs.append(dist.irecv(hi, rank + 1))
            for r_ in reqs:
                r_.wait()
            if rank > 0:
                newT[i0 - 2:i0] = lo.numpy()
            if rank < world - 1:
                newT[i1:i1 + 2] = hi.numpy()
            lat.T = newT
        lat.row_sums(i0, i1, rowsum, rowcnt)
        bs = np.zeros(3 * L); bc = np.zeros(3 * L, np.int64)
        lat.block_sums(rowsum, rowcnt, i0, i1, bs, bc)
        mine = torch.from_numpy(np.concatenate([bs[3 * i0:3 * i1], bc[3 * i0:3 * i1].astype(np.float64)]))
        gathered = [torch.empty_like(mine) for _ in range(world)]
        dist.all_gather(gathered, mine)                              # == ncclAllGather of the BlockEnt slices
        for r_, g in enumerate(gathered):
            j0, j1 = _slab(L, r_, world)
            g = g.numpy(); n = 3 * (j1 - j0)
            bs[3 * j0:3 * j1] = g[:n]; bc[3 * j0:3 * j1] = g[n:].astype(np.int64)
        total, n_events, n_dep = lat.total(bs, bc)
        assert n_events > 0
        r = u_pick[step] * total
        # every rank walks the block level identically; only the owner can finish the descent
        rec = np.zeros(14)
        ev = None
        probe = lat.select_tree  # owner test: the chosen block's plane
        import ctypes
        # find the owning plane by descending the block tree with zero row information: reuse the
        # oracle on the owner only -- ownership follows from the gathered block sums alone
        cum, owner_plane = 0.0, None
        # (canonical block descent, restated with numpy for the ownership decision)
        PB = 1
        while PB < 3 * L:
            PB *= 2
        lo_, n_, base = 0, PB, 0.0
        def tsum(lo__, n__):
            if lo__ >= 3 * L:
                return 0.0
            if n__ == 1:
                return bs[lo__]
            return tsum(lo__, n__ // 2) + tsum(lo__ + n__ // 2, n__ // 2)
        def tcnt(lo__, n__):
            return int(bc[lo__:min(lo__ + n__, 3 * L)].sum()) if lo__ < 3 * L else 0
        while n_ > 1:
            h_ = n_ // 2
            sl, cl, cr = tsum(lo_, h_), tcnt(lo_, h_), tcnt(lo_ + h_, h_)
            if cr == 0 or (cl > 0 and base + sl >= r):
                n_ = h_
            else:
                base += sl; lo_ += h_; n_ = h_
        owner_plane = lo_ // 3
        if i0 <= owner_plane < i1:
            ev = lat.select_tree(bs, bc, rowsum, rowcnt, r)
            assert ev.pos[0] == owner_plane
            # payload carried by the record (cetkmc_event.theta/phi): diff moves the source's
            # orientation, att copies the neighbour's -- the receiver may not hold that voxel
            src = tuple(ev.pos) if ev.type == 1 else (tuple(ev.target) if ev.type == 3 else None)
            pth, pph = (lat.theta[src], lat.phi[src]) if src else (0.0, 0.0)
            rec[:] = [1, ev.type, ev.pos[0], ev.pos[1], ev.pos[2], ev.target[0], ev.target[1], ev.target[2],
                      ev.atom, ev.rate, ev.dep_rank, 0, pth, pph]
        recs = [torch.empty(14, dtype=torch.float64) for _ in range(world)]
        dist.all_gather(recs, torch.from_numpy(rec))                 # == ncclAllGather of the 64-byte event records
        valid = [x.numpy() for x in recs if x[0] == 1]
        assert len(valid) == 1
        v = valid[0]
        e = oracle.Event()
        e.type = int(v[1]); e.pos[:] = [int(v[2]), int(v[3]), int(v[4])]
        e.target[:] = [int(v[5]), int(v[6]), int(v[7])]; e.atom = int(v[8]); e.rate = v[9]; e.dep_rank = int(v[10])
        if e.type == 0:
            u = u_np[pos + e.dep_rank]
            e.atom = 3 if u < 0.2 else (2 if u < 0.2 + 0.10 else 1)
        pos += n_dep
        th = ph = 0.0
        if e.type in (0, 2):
            th, ph = np.pi * u_np[pos], 2 * np.pi * u_np[pos + 1]
            pos += 2
        mk = u_def[step] < 0.05
        if e.type in (1, 3):
            th, ph = v[12], v[13]

        def put(ijk, st, t_, p_):                                    # write_site(): only inside slab+halo
            if a <= ijk[0] < b:
                lat.state[ijk] = st; lat.theta[ijk] = t_; lat.phi[ijk] = p_
        upd = tuple(e.pos)
        if e.type in (0, 2, 3):
            put(upd, e.atom, th, ph)
        else:
            put(tuple(e.target), e.atom, th, ph)
            put(upd, 0, 0.0, 0.0)
            upd = tuple(e.target)
        if mk:
            put(upd, 4, 0.0, 0.0)
        log.append((e.type, tuple(e.pos), tuple(e.target), e.atom, total))
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), state=lat.state[i0:i1], theta=lat.theta[i0:i1], phi=lat.phi[i0:i1],
             T=lat.T[i0:i1], log=np.array([(t, *p, *g, a_, tot) for t, p, g, a_, tot in log]))
    dist.destroy_process_group()


@pytest.mark.parametrize("L,world", [(12, 2), (12, 3)])
def test_slab_protocol_world(oracle_mod, tmp_path, L, world):
    n_steps, seed = 45, 5
    mp.spawn(_worker, args=(world, _free_port(), L, n_steps, seed, str(tmp_path)), nprocs=world, join=True)
    # single-domain reference run
    state, theta, phi, T, defects = random_lattice(L, seed, fill=0.25)
    lat = oracle_mod.Lattice(state, theta, phi, T, defects, impurity_c=0.2)
    streams = np.random.RandomState(7)
    u_pick, u_def = streams.random_sample(n_steps), streams.random_sample(n_steps)
    u_np = streams.random_sample(n_steps * (L * L + 2))
    res = lat.run_steps(0, n_steps, 0.05, u_pick, u_def, u_np, rng_mode=0, thermal_mode=1)
    assert res["done"] == n_steps
    for rank in range(world):
        z = np.load(os.path.join(str(tmp_path), f"rank{rank}.npz"))
        i0, i1 = _slab(L, rank, world)
        assert np.array_equal(z["state"], lat.state[i0:i1])
        assert np.array_equal(z["theta"], lat.theta[i0:i1]) and np.array_equal(z["phi"], lat.phi[i0:i1])
        assert np.array_equal(z["T"], lat.T[i0:i1])
        assert np.array_equal(z["log"][:, 0], res["events"]["type"])
        assert np.array_equal(z["log"][:, 1:4], res["events"]["pos"])
        assert np.array_equal(z["log"][:, -1], res["totals"])      # bit-identical totals on every rank


# ---- Mode B (super-steps) across ranks: boxes sharded with the slabs, boundary box layers' events exchanged ----------
KEY_PICK, KEY_THETA, KEY_PHI, KEY_DEFECT = 1 << 40, 2 << 40, 3 << 40, 4 << 40


def _worker_b(rank, world, port, L, box, n_steps, seed, df, out_dir, null_events=False):
    """Protocol of cetkmc_run_supersteps across ranks (cetkmc_hip.hip), with the CPU oracle as the per-slab engine:
    per super-step (1) own row/block sums, all-gather, global total; (2) every rank picks for ITS boxes on its
    slab + halo copy; (3) the events of the bottom / top box layer go to the rank below / above (neighbour
    send/recv), records carry species and orientation; (4) own and received events are applied, writes clipped
    to slab + halo.  Everything outside slab + halo is poisoned."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import oracle
    state, theta, phi, T, defects = random_lattice(L, seed, fill=0.25)
    i0, i1 = _slab(L, rank, world)
    a, b = max(0, i0 - 2), min(L, i1 + 2)
    rs = np.random.RandomState(1000 + rank)
    poison = np.ones(L, bool)
    poison[a:b] = False
    state = state.copy(); theta = theta.copy(); phi = phi.copy(); T = T.copy()
    state[poison] = rs.randint(0, 5, state[poison].shape)
    theta[poison] = 99.0
    phi[poison] = -99.0
    T[poison] = np.nan
    lat = oracle.Lattice(state, theta, phi, T, defects, impurity_c=0.2)
    nb, H = L // box, box // 2
    nb2 = nb * nb
    d0, D = (i0 // box) * nb2, ((i1 - i0) // box) * nb2
    rowsum = np.zeros((L, 3, L)); rowcnt = np.zeros((L, 3, L), np.int32)
    log, totals, n_exec = [], [], []
    for g in range(n_steps):
        lat.row_sums(i0, i1, rowsum, rowcnt)
        bs = np.zeros(3 * L); bc = np.zeros(3 * L, np.int64)
        lat.block_sums(rowsum, rowcnt, i0, i1, bs, bc)
        mine = torch.from_numpy(np.concatenate([bs[3 * i0:3 * i1], bc[3 * i0:3 * i1].astype(np.float64)]))
        gathered = [torch.empty_like(mine) for _ in range(world)]
        dist.all_gather(gathered, mine)
        for r_, gt in enumerate(gathered):
            j0_, j1_ = _slab(L, r_, world)
            gt = gt.numpy(); n = 3 * (j1_ - j0_)
            bs[3 * j0_:3 * j1_] = gt[:n]; bc[3 * j0_:3 * j1_] = gt[n:].astype(np.int64)
        total, n_events, _ = lat.total(bs, bc)
        totals.append(total)
        assert n_events > 0
        sec = g % 8
        si, sj, sk = (sec >> 2) & 1, (sec >> 1) & 1, sec & 1
        recs = np.zeros((D, 12))                       # valid, type, pos3, target3, atom, theta, phi, make_defect
        picks = []
        for dl in range(D):
            d = d0 + dl
            di, dj, dk = d // nb2, (d // nb) % nb, d % nb
            picks.append(lat.window_pick(di * box + si * H, dj * box + sj * H, dk * box + sk * H, H,
                                         oracle.counter_uniform(seed, g, KEY_PICK | d), with_total=True))
        r_max = 0.0
        if null_events:     # R_max over ALL ranks' boxes: one more small collective per super-step (the engine: 8-byte all-gather)
            mine_max = torch.tensor([max([R for ev, R in picks if ev is not None], default=0.0)], dtype=torch.float64)
            dist.all_reduce(mine_max, op=dist.ReduceOp.MAX)
            r_max = float(mine_max[0])
        for dl in range(D):
            d = d0 + dl
            ev, R_d = picks[dl]
            if ev is None:
                continue
            if null_events and not (oracle.counter_uniform(seed, g, oracle.KEY_ACCEPT | d) * r_max < R_d):
                recs[dl, 0] = -2                       # null event: nothing drawn, nothing applied, nothing sent as an event
                continue
            atom, th, ph = ev.atom, 0.0, 0.0
            if ev.type == 0:
                u = oracle.counter_uniform(seed, g, ev.pos[1] * L + ev.pos[2])
                atom = 3 if u < 0.2 else (2 if u < 0.2 + 0.10 else 1)
            if ev.type in (0, 2):
                th = np.pi * oracle.counter_uniform(seed, g, KEY_THETA | d)
                ph = 2 * np.pi * oracle.counter_uniform(seed, g, KEY_PHI | d)
            else:               # diff moves the source's orientation, att copies the neighbour's
                src = tuple(ev.pos) if ev.type == 1 else tuple(ev.target)
                th, ph = lat.theta[src], lat.phi[src]
            mk = df > 0.0 and oracle.counter_uniform(seed, g, KEY_DEFECT | d) < df
            recs[dl] = [1, ev.type, *ev.pos, *ev.target, atom, th, ph, mk]
        # boundary box layers to the neighbours (== ncclSend/ncclRecv of (L/box)^2 event records each way)
        reqs, lo, hi = [], None, None
        if rank > 0:
            reqs.append(dist.isend(torch.from_numpy(recs[:nb2].copy()), rank - 1))
            lo = torch.empty((nb2, 12), dtype=torch.float64); reqs.append(dist.irecv(lo, rank - 1))
        if rank < world - 1:
            reqs.append(dist.isend(torch.from_numpy(recs[D - nb2:].copy()), rank + 1))
            hi = torch.empty((nb2, 12), dtype=torch.float64); reqs.append(dist.irecv(hi, rank + 1))
        for r_ in reqs:
            r_.wait()
        todo = [recs] + [x.numpy() for x in (lo, hi) if x is not None]

        def put(ijk, st, t_, p_):                                    # write_site(): only inside slab+halo
            if a <= ijk[0] < b:
                lat.state[ijk] = st; lat.theta[ijk] = t_; lat.phi[ijk] = p_
        for block in todo:
            for v in block:
                if v[0] != 1:
                    continue
                typ, pos, tgt, atom = int(v[1]), tuple(int(x) for x in v[2:5]), tuple(int(x) for x in v[5:8]), int(v[8])
                upd = pos
                if typ in (0, 2, 3):
                    put(pos, atom, v[9], v[10])
                else:
                    put(tgt, atom, v[9], v[10])
                    put(pos, 0, 0.0, 0.0)
                    upd = tgt
                if v[11]:
                    put(upd, 4, 0.0, 0.0)
        n_exec.append(int((recs[:, 0] == 1).sum()))
        log.append(recs[:, :9].copy())
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), state=lat.state[i0:i1], theta=lat.theta[i0:i1], phi=lat.phi[i0:i1],
             log=np.array(log), totals=np.array(totals), n_exec=np.array(n_exec))
    dist.destroy_process_group()


@pytest.mark.parametrize("L,world,box,null_events", [(16, 2, 8, False), (24, 3, 8, False), (16, 2, 8, True), (24, 3, 8, True)])
def test_mode_b_slab_protocol_world(oracle_mod, tmp_path, L, world, box, null_events):
    n_steps, seed, df = 20, 11, 0.05
    mp.spawn(_worker_b, args=(world, _free_port(), L, box, n_steps, seed, df, str(tmp_path), null_events), nprocs=world, join=True)
    state, theta, phi, T, defects = random_lattice(L, seed, fill=0.25)
    lat = oracle_mod.Lattice(state, theta, phi, T, defects, impurity_c=0.2)
    res = lat.run_supersteps(0, n_steps, box, df, seed, thermal_mode=0, null_events=null_events)
    assert res["done"] == n_steps
    zs = [np.load(os.path.join(str(tmp_path), f"rank{r}.npz")) for r in range(world)]
    log = np.concatenate([z["log"] for z in zs], axis=1)             # [n][D global][9]
    live = log[:, :, 0] == 1
    assert np.array_equal(live, res["events"]["type"] >= 0)
    assert np.array_equal(log[:, :, 0] == -2, res["events"]["type"] == -2) and ((log[:, :, 0] == -2).sum() > 0) == null_events
    assert np.array_equal(log[:, :, 1][live], res["events"]["type"][live])
    assert np.array_equal(log[:, :, 2:5][live], res["events"]["pos"][live])
    assert np.array_equal(log[:, :, 5:8][live], res["events"]["target"][live])
    assert np.array_equal(log[:, :, 8][live], res["events"]["atom"][live])
    assert np.array_equal(sum(z["n_exec"] for z in zs), res["n_exec"])
    for rank, z in enumerate(zs):
        i0, i1 = _slab(L, rank, world)
        assert np.array_equal(z["totals"], res["totals"])
        assert np.array_equal(z["state"], lat.state[i0:i1])
        assert np.array_equal(z["theta"], lat.theta[i0:i1]) and np.array_equal(z["phi"], lat.phi[i0:i1])
