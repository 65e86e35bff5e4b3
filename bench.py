#!/usr/bin/env python3
"""Benchmark of the KMC stepping hot path (BASELINE.json metric: KMC events/sec on a 256^3
lattice; achieved HBM GB/s).

  python bench.py --gpus 1 --steps K --warmup W          (N=1: config 3, 256^3, one MI355X)
  python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...   (one rank per GPU)

A "step" is ONE exact KMC step (Mode A): full-lattice rate sweep (every candidate event's
Arrhenius rate evaluated and reduced), cumulative-rate event pick, lattice update, plus the
melt-pool temperature update every 20 steps.  `value` = candidate events evaluated+scanned per
second (sum over the timed steps of len(events)) / time -- the reference's unit of work per
sweep; the executed-event rate (= steps/s, one event per sweep exactly like the reference) is
reported next to it and never conflated with it.

N>1 is weak scaling: L = 256/320/408/512 for 1/2/4/8 GPUs (about 1.68e7 voxels per GPU),
axis-0 slabs, RCCL all-gathers of the block sums and of the chosen event every step, T halo
send/recv after every thermal update.  torch is imported only for the multi-process
rendezvous/barrier; the data path is HIP + RCCL inside libcetkmc_hip.so.
"""
import argparse
import json
import os
import random
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.join(ROOT, "cet-driven-simulation-for-3d-printing-am-kmc-approach_amd")
for p in (PKG, ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)

L_FOR_GPUS = {1: 256, 2: 320, 4: 408, 8: 512}
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: 8.0 TB/s spec
B_ALG_SWEEP = 10.0             # bytes/voxel/sweep: census class u16 + T f64, each read once
IMPURITY_C, DEFECT_FRACTION, SEED = 0.2, 3e-3, 42


def streams(n, seed):
    rs = random.Random(seed)
    u_pick = np.array([rs.random() for _ in range(n)])
    u_def = np.array([rs.random() for _ in range(n)])
    u_np = np.random.RandomState(seed).random_sample(2 * n + 2)
    return u_pick, u_def, u_np


def cpu_baseline(L, fields, n_gpu_events, budget_s=15.0):
    """Oracle (C restatement) on the first steps of the same workload: first as the scalar port (1 core, half the
    budget), then with its row / thermal loops on the box's CPU share (OpenMP); also checks the GPU's chosen
    events for those steps (full-size parity)."""
    from cetkmc import synthetic
    from oracle import oracle
    state, theta, phi, T, defects = fields
    lat = oracle.Lattice(state, theta, phi, T, defects, impurity_c=IMPURITY_C)
    done, ev_all, nev_all = 0, [], []
    n_max = len(n_gpu_events)
    n_threads = max(1, min(16, os.cpu_count() or 1))          # a 1-GPU box's CPU share
    legs = {"single": [1, 0, 0.0], "multi": [n_threads, 0, 0.0]}     # threads, steps, seconds
    for leg in ("single", "multi"):
        oracle.set_threads(legs[leg][0])
        while done < n_max:
            n = 1
            u_pick, u_def, u_np = streams(n_max, SEED)
            q = synthetic.laser_planes(L, done, n)
            t0 = time.perf_counter()
            # rng_mode 1 consumes only orientation draws; replay the cursor from the GPU log
            res = lat.run_steps(done, n, DEFECT_FRACTION, u_pick[done:done + n], u_def[done:done + n],
                                u_np[cpu_baseline.np_pos:], rng_mode=1, seed=SEED, thermal_mode=2, q_planes=q)
            legs[leg][2] += time.perf_counter() - t0
            legs[leg][1] += res["done"]
            cpu_baseline.np_pos += res["np_used"]
            ev_all.append(res["events"])
            nev_all.append(res["n_events"])
            done += res["done"]
            if res["done"] < n or legs[leg][2] > budget_s / 2 or (leg == "single" and legs[leg][1] >= n_max // 2):
                break
    oracle.set_threads(1)
    ev = np.concatenate(ev_all)
    nev = np.concatenate(nev_all)
    return dict(steps=done, events=ev, n_events=nev, legs=legs)


cpu_baseline.np_pos = 0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=100)
    ap.add_argument("--L", type=int, default=0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-incremental", action="store_true", help="skip the extra exact-incremental-mode run")
    ap.add_argument("--no-mode-b", action="store_true", help="skip the extra Mode B (super-step) run")
    ap.add_argument("--no-phases", action="store_true", help="skip the extra per-phase timing run")
    ap.add_argument("--transport", choices=["rccl", "host"], default="rccl",
                    help="N > 1: RCCL over xGMI (default) or the host-relay transport (ranks may share one GPU; rehearsal only)")
    ap.add_argument("--set-option", action="append", default=[], metavar="KEY=INT", help="cetkmc_set_option before the run")
    ap.add_argument("--overlap-interface", type=int, default=-1,
                    help="speculative interface evaluation of the next step during select/collectives: 1 on, 0 off (default: engine default = off)")
    ap.add_argument("--extras-multi", action="store_true",
                    help="also run the exact-incremental extra at N > 1 (default: N = 1 only, the scaling runs time the full-sweep loop alone)")
    ap.add_argument("--cpu-budget", type=float, default=15.0)
    ap.add_argument("--force-dist", action="store_true",
                    help="take the multi-process code path (gloo rendezvous + RCCL communicator) even with one rank")
    a = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    N = a.gpus
    if world != N and world > 1:
        raise SystemExit(f"--gpus {N} but WORLD_SIZE={world}")
    L = a.L or L_FOR_GPUS.get(N, 256)
    if L % N:
        raise SystemExit(f"L={L} not divisible by {N} ranks")

    # Libraries chat on stdout (gloo's "[Gloo] Rank ..." line, RCCL's version banner under NCCL_DEBUG): keep fd 1 for
    # the ONE JSON line -- everything else goes to stderr until the result is printed.
    sys.stdout.flush()
    stdout_fd = os.dup(1)
    os.dup2(2, 1)

    import cetkmc
    from cetkmc import synthetic

    dist = None
    uid = None
    if N > 1 or a.force_dist:
        import torch
        import torch.distributed as dist
        if "RANK" not in os.environ:             # plain `python bench.py --force-dist`
            os.environ.update(RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29533")
        dist.init_process_group("gloo")          # control plane only (id exchange, barrier, max)
        if a.transport == "rccl":
            box = [cetkmc.Engine.unique_id() if rank == 0 else None]
            dist.broadcast_object_list(box, src=0)
            uid = box[0]
        # one GPU per rank; if the launcher narrowed device visibility to one GPU per process, that GPU is index 0
        dev = local_rank % max(1, cetkmc.device_count())
        if a.transport == "host":     # rehearsal on a one-GPU box: ranks share the GPU, collectives relayed through gloo
            from cetkmc import host_transport
            eng = cetkmc.Engine(L, impurity_c=IMPURITY_C, device=dev, rank=rank, nranks=N, host_comm=host_transport.torch_callbacks())
        else:
            eng = cetkmc.Engine(L, impurity_c=IMPURITY_C, device=dev, rank=rank, nranks=N, unique_id=uid)
    else:
        eng = cetkmc.Engine(L, impurity_c=IMPURITY_C, device=0)

    if a.overlap_interface >= 0:
        eng.set_option("overlap_interface", a.overlap_interface)
    for kv in a.set_option:                       # engine options for A/B runs, e.g. --set-option fused_reduce=0
        k, v = kv.split("=")
        eng.set_option(k, int(v))

    # ---- synthetic input, resident in HBM before anything is timed -----------------------
    a0, a1 = max(0, eng.i0 - 2), min(L, eng.i1 + 2)
    state, theta, phi, T, defects = synthetic.planes(L, a0, a1, seed=SEED)
    eng.upload_planes(a0, a1, state, theta, phi, T, defects)
    eng.set_prev_state(None)

    def prepare(step0, n):
        """Host-side inputs of a batch (random streams, laser source planes); built BEFORE timing."""
        u_pick, u_def, u_np = streams(step0 + n, SEED)
        return dict(step0=step0, n=n, u_pick=u_pick[step0:], u_def=u_def[step0:], u_np=u_np,
                    q=synthetic.laser_planes(L, step0, n))

    def run(step0, n, profile=False, logs=False, prep=None, incremental=False):
        b = prep or prepare(step0, n)
        return eng.run_steps(step0, n, DEFECT_FRACTION, b["u_pick"], b["u_def"], b["u_np"][run.np_pos:],
                             rng_mode=1, seed=SEED, thermal_mode=2, q_planes=b["q"], use_latent=True,
                             profile=profile, want_logs=True, incremental=incremental)
    run.np_pos = 0

    def barrier():
        eng.sync()
        if dist is not None:
            dist.barrier()

    # ---- parity sample + CPU baseline (rank 0, N=1): GPU and oracle run the same first steps
    step = 0
    base = None
    if N == 1 and not a.no_cpu_baseline:
        n_chk = 40
        r = run(0, n_chk, logs=True)
        run.np_pos += r["np_used"]
        step = r["done"]
        base = cpu_baseline(L, (state, theta, phi, T, defects), r["n_events"], a.cpu_budget)
        k = base["steps"]
        same = all(np.array_equal(r["events"][f][:k], base["events"][f][:k]) for f in ("type", "pos", "target", "atom"))
        same = same and np.array_equal(r["n_events"][:k], base["n_events"][:k])
        base["parity"] = bool(same)
        if not same:
            print("WARNING: GPU and CPU oracle disagree on the first steps", file=sys.stderr)
        del state, theta, phi, T, defects

    # ---- warmup, then EXACTLY K timed steps ------------------------------------------------
    if a.warmup:
        r = run(step, a.warmup)
        run.np_pos += r["np_used"]
        step += r["done"]
    timed_inputs = prepare(step, a.steps)
    barrier()
    t0 = time.perf_counter()
    r = run(step, a.steps, profile=3, prep=timed_inputs)      # sweep kernel timed by hipEvents on every 8th launch
    barrier()
    dt = time.perf_counter() - t0
    assert r["done"] == a.steps and r["status"] == 0, r
    if dist is not None:
        import torch
        t = torch.tensor([dt], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t[0])

    # ---- after the timed region: per-phase device time (hipEvents at every phase boundary, cetkmc_get_counters)
    run.np_pos += r["np_used"]
    phases = None
    if not a.no_phases and (N == 1 or a.extras_multi):
        n_ph = min(200, a.steps)
        ph_inputs = prepare(step + a.steps, n_ph)
        eng.counters(reset=True)
        rp = run(step + a.steps, n_ph, profile=2, prep=ph_inputs)
        c = eng.counters()
        run.np_pos += rp["np_used"]
        if rp["done"] == n_ph and c["profiled_steps"] == n_ph:
            phases = {k[3:] + "_us_per_step": 1e3 * c[k] / n_ph for k in ("ms_thermal", "ms_interface", "ms_sweep", "ms_reduce",
                                                                           "ms_select_apply")}
            phases.update(steps=n_ph, device_us_per_step=1e3 * rp["wall_ms"] / n_ph,
                          alg_bytes_per_step=(c["alg_bytes_sweep"] + c["alg_bytes_thermal"]) / n_ph,
                          note="full-sweep loop, hipEvents at every phase boundary (slower than the timed run by the event "
                               "records); thermal averaged over its 1-in-20 cadence")

    # ---- the same loop in exact incremental mode (reported beside `value`, never as it)
    inc = None
    if not a.no_incremental and (N == 1 or a.extras_multi):
        inc_inputs = prepare(step + 2 * a.steps, a.steps)
        barrier()
        t1 = time.perf_counter()
        ri = run(step + 2 * a.steps, a.steps, prep=inc_inputs, incremental=True)
        barrier()
        dti = time.perf_counter() - t1
        if dist is not None:
            import torch
            t = torch.tensor([dti], dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dti = float(t[0])
        if ri["done"] == a.steps:
            inc = {"steps_per_s": a.steps / dti, "ms_per_step": 1e3 * dti / a.steps, "full_sweeps": ri["full_sweeps"],
                   "steps": a.steps,
                   "note": "cetkmc_run_steps incremental=1 on the following K steps: between temperature updates only the "
                           "rows the previous event made stale are re-evaluated; bit-identical to full sweeps "
                           "(tests/test_gpu_parity.py::test_incremental_mode_bit_identical); not part of `value`"}

    # ---- Mode B (super-steps over 8^3 boxes; not the reference's trajectory, own CPU comparator): executed events/s
    mode_b = None
    if N == 1 and not a.no_mode_b and L % 8 == 0:
        sb, nb_steps = step + 3 * a.steps, 40
        qb = synthetic.laser_planes(L, sb, nb_steps)
        eng.sync()
        t2 = time.perf_counter()
        rb = eng.run_supersteps(sb, nb_steps, 8, DEFECT_FRACTION, seed=SEED, thermal_mode=2, q_planes=qb)
        eng.sync()
        dtb = time.perf_counter() - t2
        if rb["done"] == nb_steps:
            mode_b = {"executed_events_per_s": float(rb["n_exec"].sum()) / dtb, "ms_per_superstep": 1e3 * dtb / nb_steps,
                      "events_per_superstep": float(rb["n_exec"].mean()), "boxes": rb["domains"], "supersteps": nb_steps,
                      "note": "cetkmc_run_supersteps box=8 on the lattice left by the runs above: one full rate sweep per "
                              "super-step, every box executes <= 1 event from its active octant; bit-identical to "
                              "oracle orc_run_supersteps (tests/test_gpu_mode_b.py); not part of `value`"}

    cand = float(np.sum(r["n_events"].astype(np.float64)))     # identical on every rank (global counts)
    steps_per_s = a.steps / dt
    sweep_ms = r["sweep_ms_total"] / max(r["sweep_launches"], 1)
    n_own = (eng.i1 - eng.i0) * L * L
    achieved = B_ALG_SWEEP * n_own / (sweep_ms * 1e-3) / 1e9 if sweep_ms > 0 else 0.0
    traffic, traffic_src = None, None
    for summary_path in sorted(__import__("glob").glob(os.path.join(ROOT, "profiles", "r*_summary.json")), reverse=True):
        try:
            kk = json.load(open(summary_path))["kernels"]["k_sweep_stream"]
            if N == 1 and L == 256 and "hbm_bytes_per_launch" in kk:
                traffic, traffic_src = kk["hbm_bytes_per_launch"], os.path.relpath(summary_path, ROOT)
                break
        except Exception:
            pass
    out = {
        "metric": "kmc_events_per_sec", "value": cand / dt, "unit": "events/s",
        "n_gpus": N, "steps": a.steps, "warmup": a.warmup, "ms_per_step": 1e3 * dt / a.steps,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {
            "workload": f"config3: {L}^3 voxel lattice ({N} axis-0 slab(s)), k<L/4 pre-filled W/Re/C, moving Gaussian "
                        "melt-pool T-field (update_temperature every 20 steps), exact Mode A: 1 executed event per "
                        "full rate sweep",
            "L": L, "impurity_c": IMPURITY_C, "defect_fraction": DEFECT_FRACTION, "rng_mode": "counter",
            "event_definition": "value counts CANDIDATE events (entries of get_event_rates' list: rates evaluated, "
                                "reduced and scanned per sweep); executed events/s = steps_per_s",
        },
        "steps_per_s": steps_per_s, "executed_events_per_s": steps_per_s,
        "candidate_events_per_step": cand / a.steps, "voxel_updates_per_s": float(L) ** 3 * steps_per_s,
        "device_ms_per_step": r["wall_ms"] / a.steps,
        "roofline": {"bound": "hbm", "kernel": "k_sweep_stream", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src,
                     "avg_launch_ms": sweep_ms, "launches_timed": int(r["sweep_launches"]),   # every 8th launch of the timed region
                     "alg_bytes_per_voxel": B_ALG_SWEEP, "voxels_per_launch": n_own,
                     # BASELINE.md section 4 / SURVEY 8(d) pre-build accounting: 26.8 B per voxel per step (state u8 + T +
                     # theta + phi f64 + defects u8 streamed every sweep, thermal 16 B / 20) x steps/s.  The build does
                     # NOT stream theta/phi/defects (touched at interface voxels only), so this over-counts moved bytes;
                     # shown only because BASELINE.md's 40 % target (7.1 k steps/s at 256^3) is phrased in it.
                     "baseline_md_accounting": {"bytes_per_voxel_step": 26.8,
                                                "achieved": 26.8 * float(L) ** 3 * steps_per_s / 1e9, "unit": "GB/s",
                                                "frac": 26.8 * float(L) ** 3 * steps_per_s / 1e9 / HBM_PEAK_GBS}},
    }
    if base is not None:
        nev = base["n_events"].astype(np.float64)
        (t1, s1, sec1), (tm, sm, secm) = base["legs"]["single"], base["legs"]["multi"]
        leg = ("multi", tm, sm, secm, float(nev[s1:s1 + sm].sum())) if sm > 0 else ("single", t1, s1, sec1, float(nev[:s1].sum()))
        out["cpu_baseline"] = {
            "value": leg[4] / leg[3], "unit": "events/s", "cores": leg[1], "kind": "port",
            "sample": f"steps {s1}..{s1 + sm} of the same {L}^3 workload on the C oracle with its row/thermal loops on "
                      f"{leg[1]} host threads ({leg[3]:.1f} s, incl. thermal updates)" if sm > 0 else
                      f"first {s1} steps of the same {L}^3 workload on the C oracle, 1 thread ({sec1:.1f} s)",
            "steps_per_s": leg[2] / leg[3],
            "single_core": {"value": float(nev[:s1].sum()) / sec1 if s1 else None, "steps_per_s": s1 / sec1 if s1 else None,
                            "cores": 1, "sample": f"first {s1} steps, scalar port ({sec1:.1f} s)"},
            "host_cores_available": os.cpu_count(), "parity_first_steps": base["parity"],
        }
    if phases is not None:
        out["phases"] = phases
    if inc is not None:
        out["incremental_exact"] = inc
    if mode_b is not None:
        out["mode_b"] = mode_b
    sys.stdout.flush()
    os.dup2(stdout_fd, 1)
    os.close(stdout_fd)
    if rank == 0:
        print(json.dumps(out), flush=True)
    os.dup2(2, 1)                # teardown messages (communicator destruction) stay off stdout as well
    eng.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
