#!/usr/bin/env python3
"""Benchmark of the KMC stepping hot path (BASELINE.json metric: KMC events/sec on a 256^3
lattice; achieved HBM GB/s).

  python bench.py --gpus 1 --steps K --warmup W          (N=1: config 3, 256^3, one MI355X)
  python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...   (one rank per GPU)

A "step" is ONE exact KMC step (Mode A): full-lattice rate sweep (every candidate event's rate
looked up / evaluated and reduced), cumulative-rate event pick, lattice update, plus the melt-pool
temperature update, rate-table refresh and interface-list evaluation every 20 steps.
`value` = EXECUTED events per second of that loop (= steps/s: one event per sweep, exactly like the
reference).  The candidate-event rate (entries of get_event_rates' list per second), the exact
incremental loop and Mode B (several events per sweep) are reported beside it under their own names.

`--gpus N` runs BASELINE.json's configurations: N=1 config 3 (256^3), N=2/4 config 4 (the SAME 256^3
lattice in 2/4 axis-0 slabs: strong scaling), N=8 config 5 (512^3, impurity_c=0.2, defect mask refreshed
every 200 steps).  `--scaling weak` keeps the per-GPU voxel count fixed instead (L = 256/320/408/512).
Across ranks: RCCL all-gathers of the block sums and of the chosen event every step, T halo send/recv after
every thermal update.  torch is imported only for the multi-process rendezvous/barrier; the data path is HIP
+ RCCL inside libcetkmc_hip.so.
"""
import argparse
import glob
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

L_BASELINE = {1: 256, 2: 256, 4: 256, 8: 512}      # BASELINE.json configs 3, 4, 4, 5
L_WEAK = {1: 256, 2: 320, 4: 408, 8: 512}          # ~1.68e7 voxels per GPU
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: 8.0 TB/s spec
B_ALG_SWEEP = 9.0              # bytes/voxel/sweep: census class u8 + rate-table entry f64, each read once
IMPURITY_C, DEFECT_FRACTION, SEED = 0.2, 3e-3, 42
DEFECT_REFRESH_EVERY = 200     # kmc_simulation.py:335-338 (METRIC_UPDATE_STEP)


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
    np_pos = 0
    for leg in ("single", "multi"):
        oracle.set_threads(legs[leg][0])
        while done < n_max:
            n = 1
            u_pick, u_def, u_np = streams(n_max, SEED)
            q = synthetic.laser_planes(L, done, n)
            t0 = time.perf_counter()
            # rng_mode 1 consumes only orientation draws; replay the cursor from the GPU log
            res = lat.run_steps(done, n, DEFECT_FRACTION, u_pick[done:done + n], u_def[done:done + n],
                                u_np[np_pos:], rng_mode=1, seed=SEED, thermal_mode=2, q_planes=q)
            legs[leg][2] += time.perf_counter() - t0
            legs[leg][1] += res["done"]
            np_pos += res["np_used"]
            ev_all.append(res["events"])
            nev_all.append(res["n_events"])
            done += res["done"]
            if res["done"] < n or legs[leg][2] > budget_s / 2 or (leg == "single" and legs[leg][1] >= n_max // 2):
                break
    oracle.set_threads(1)
    ev = np.concatenate(ev_all)
    nev = np.concatenate(nev_all)
    return dict(steps=done, events=ev, n_events=nev, legs=legs)


def live_traffic(L, steps=20, timeout=90):
    """The sweep kernel's memory-side traffic per launch, measured in THIS run: two child runs of this script (the same
    full-sweep loop, a few steps) under `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` -- separate passes, counters only,
    the program itself right behind `--` (MI355X_MICROARCH.md, HBM section).  Called BEFORE this process touches the GPU.
    bytes = 2 x FETCH_SIZE + WRITE_SIZE (KB -> B): on gfx950 FETCH_SIZE reports half the bytes of wide coalesced reads, which is
    what the streaming sweep issues (tools/summarize_profiles.py applies the same correction to the committed profiles).
    Returns (bytes per launch or None, provenance / reason)."""
    import csv
    import shutil
    import statistics
    import subprocess
    import tempfile
    if any(k.startswith(("ROCPROF", "ROCP_")) for k in os.environ) or "rocprofiler" in os.environ.get("LD_PRELOAD", ""):
        return None, "already running under a profiler"
    exe = shutil.which("rocprofv3")
    if not exe:
        return None, "rocprofv3 not on PATH"
    med = {}
    for counter in ("FETCH_SIZE", "WRITE_SIZE"):
        d = tempfile.mkdtemp(prefix="cetkmc_pmc_", dir="/tmp")
        try:
            cmd = [exe, "--pmc", counter, "--output-format", "csv", "-d", d, "--", "python3", os.path.abspath(__file__),
                   "--L", str(L), "--steps", str(steps), "--warmup", "2", "--no-cpu-baseline", "--no-incremental", "--no-mode-b",
                   "--no-phases", "--no-recompute", "--no-512", "--no-live-traffic"]
            # its own process group: a pass that does not end in time is removed together with the program it started
            proc = subprocess.Popen(cmd, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"), stdout=subprocess.DEVNULL,
                                    stderr=subprocess.DEVNULL, start_new_session=True)
            try:
                rc = proc.wait(timeout=timeout)
            except subprocess.TimeoutExpired:
                import signal
                os.killpg(proc.pid, signal.SIGKILL)
                proc.wait()
                return None, f"rocprofv3 --pmc {counter} child run did not end within {timeout} s"
            files = glob.glob(os.path.join(d, "*", "*counter_collection.csv"))
            if rc != 0 or not files:
                return None, f"rocprofv3 --pmc {counter} child run failed (rc {rc})"
            vals = []
            for row in csv.DictReader(open(files[0])):
                head = row["Kernel_Name"].split("(")[0]
                if "k_sweep_stream<true" in head:               # the rate-table sweep of the timed loop
                    vals.append(float(row["Counter_Value"]))
            if not vals:
                return None, f"no k_sweep_stream<table> launch in the {counter} pass"
            med[counter] = statistics.median(vals)
        except Exception as ex:                                    # noqa: BLE001 -- a failed side measurement must not cost the run
            return None, f"rocprofv3 --pmc {counter} child run: {ex!r}"
        finally:
            shutil.rmtree(d, ignore_errors=True)
    return (2.0 * med["FETCH_SIZE"] + med["WRITE_SIZE"]) * 1024.0, \
        (f"this run: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE child passes of the same loop ({steps} steps), median over the sweep "
         "launches, 2 x FETCH_SIZE + WRITE_SIZE (KB); Infinity-Cache hits are included (memory-side requests of the L2)")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=100)
    ap.add_argument("--L", type=int, default=0)
    ap.add_argument("--scaling", choices=["baseline", "weak"], default="baseline",
                    help="baseline: BASELINE.json configs (N=2/4: 256^3 split, N=8: 512^3); weak: ~1.68e7 voxels per GPU")
    ap.add_argument("--config5", action="store_true", help="force config 5's workload (defect refresh every 200 steps) at any N / L")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-incremental", action="store_true", help="skip the extra exact-incremental-mode run")
    ap.add_argument("--no-mode-b", action="store_true", help="skip the extra Mode B (super-step) run")
    ap.add_argument("--no-phases", action="store_true", help="skip the extra per-phase timing run")
    ap.add_argument("--no-recompute", action="store_true", help="skip the extra run of the recompute sweep variant")
    ap.add_argument("--no-live-traffic", action="store_true",
                    help="skip the two rocprofv3 --pmc child runs that measure the sweep kernel's memory traffic in this run (N = 1)")
    ap.add_argument("--no-512", action="store_true",
                    help="skip the extra 512^3 single-GPU sweep measurement (working set 1.2 GB: no Infinity-Cache residency)")
    ap.add_argument("--transport", choices=["rccl", "host"], default="rccl",
                    help="N > 1: RCCL over xGMI (default) or the host-relay transport (ranks may share one GPU; rehearsal only)")
    ap.add_argument("--set-option", action="append", default=[], metavar="KEY=INT", help="cetkmc_set_option before the run")
    ap.add_argument("--extras-multi", action="store_true",
                    help="also run the phase / incremental extras at N > 1 (default: N = 1 only; Mode B always runs)")
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
    L = a.L or (L_BASELINE if a.scaling == "baseline" else L_WEAK).get(N, 256)
    if L % N:
        raise SystemExit(f"L={L} not divisible by {N} ranks")
    config5 = a.config5 or (a.scaling == "baseline" and N == 8 and not a.L)
    if config5:
        cfg_name, scaling = "config5", "weak"          # 8x the voxels of config 3 on 8 GPUs
    elif N == 1:
        cfg_name, scaling = "config3", "weak"
    elif a.scaling == "baseline" and not a.L:
        cfg_name, scaling = "config4", "strong"        # the 256^3 lattice of config 3, split
    else:
        cfg_name, scaling = "config3-weak", "weak"

    # memory traffic of the dominant kernel, measured now (child runs under rocprofv3 --pmc) -- before this process owns the GPU
    live_bytes, live_src = (None, "not requested")
    if N == 1 and world == 1 and not a.no_live_traffic and not config5 and not a.force_dist:
        live_bytes, live_src = live_traffic(L)
        if live_bytes is None:
            print(f"bench: live traffic measurement skipped ({live_src})", file=sys.stderr)
    live512_bytes, live512_src = (None, "not requested")
    if live_bytes is not None and L == 256 and not a.no_512:      # the cache-free leg's traffic too (hbm_512 below)
        live512_bytes, live512_src = live_traffic(512, steps=6)

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
        # this pool's host driver supports dmabuf IPC only: RCCL's buffer sharing across processes needs it (the environment
        # normally exports it already)
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
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

    # N > 1: the transport proves itself before anything is timed (patterned all-gather + neighbour exchange at the two
    # message sizes of the stepping loop: the 64-byte event record / BlockEnt slices, one temperature-halo plane pair)
    comm_check = None
    if dist is not None:
        comm_check = [eng.comm_selftest(64), eng.comm_selftest(2 * L * L * 8)]

    for kv in a.set_option:                       # engine options for A/B runs, e.g. --set-option sweep_variant=2
        k, v = kv.split("=")
        eng.set_option(k, int(v))

    def allreduce(x, op):
        if dist is None:
            return x
        import torch
        t = torch.tensor([x], dtype=torch.float64)
        dist.all_reduce(t, op=getattr(dist.ReduceOp, op))
        return float(t[0])

    def rank_counts(n_mine):
        if dist is None:
            return [int(n_mine)]
        box = [None] * N
        dist.all_gather_object(box, int(n_mine))
        return box

    # ---- synthetic input, resident in HBM before anything is timed -----------------------
    a0, a1 = max(0, eng.i0 - 2), min(L, eng.i1 + 2)
    state, theta, phi, T, defects = synthetic.planes(L, a0, a1, seed=SEED)
    eng.sync()
    t_up = time.perf_counter()
    eng.upload_planes(a0, a1, state, theta, phi, T, defects)
    eng.set_prev_state(None)
    eng.sync()
    t_up = time.perf_counter() - t_up            # host -> HBM over PCIe (+ device-side packing); never part of `value`
    upload_bytes = sum(x.nbytes for x in (state, theta, phi, T, defects))

    def prepare(step0, n):
        """Host-side inputs of a batch (random streams, laser source planes); built BEFORE timing."""
        u_pick, u_def, u_np = streams(step0 + n, SEED)
        return dict(step0=step0, n=n, u_pick=u_pick[step0:], u_def=u_def[step0:], u_np=u_np,
                    q=synthetic.laser_planes(L, step0, n))

    refresh_ms = []

    def refresh_defects():
        """config 5: defects.track_defects after every step with step % 200 == 0 (kmc_simulation.py:335-338), without
        moving the lattice: carbon sites (index + T) of the owned planes -> host draws in global row-major site
        order -> flagged indices back.  Every rank draws the same seeded stream and uses its own slice."""
        import defects as defects_mod
        t0 = time.perf_counter()
        defects_mod.refresh_defects_device(eng, rank_counts=rank_counts, rank=rank)
        eng.sync()
        refresh_ms.append(1e3 * (time.perf_counter() - t0))

    def segment_end(step0, n, s):
        """config 5 splits the batches after every step % 200 == 0 (the defect refresh needs the host)."""
        e = step0 + n
        if config5:
            nxt = -(-s // DEFECT_REFRESH_EVERY) * DEFECT_REFRESH_EVERY         # first step >= s that is a multiple of 200
            e = min(e, nxt + 1)
        return e

    def seg_args(b, s, e, profile, incremental):
        o0 = s - b["step0"]
        q_lo = sum(1 for g in range(b["step0"], s) if g % 20 == 0)
        return (s, e - s, DEFECT_FRACTION, b["u_pick"][o0:], b["u_def"][o0:], b["u_np"][run.np_pos:]), \
            dict(rng_mode=1, seed=SEED, thermal_mode=2, q_planes=b["q"][q_lo:], use_latent=True, profile=profile, incremental=incremental)

    def stage(b, profile=False, incremental=False):
        """Inputs resident in HBM before the timed region: the batch's uniform streams and laser source planes are copied to
        the device now (cetkmc_stage_inputs); only possible when the batch is ONE device call (config 5 splits at the
        defect refreshes and hands its later segments over inside the timed region)."""
        if segment_end(b["step0"], b["n"], b["step0"]) != b["step0"] + b["n"]:
            return False
        args, kw = seg_args(b, b["step0"], b["step0"] + b["n"], profile, incremental)
        eng.stage_inputs(*args, **kw)
        b["staged"] = True
        return True

    def run(step0, n, profile=False, prep=None, incremental=False):
        """Steps step0 .. step0+n-1 as device batches; config 5 splits the batches after every step % 200 == 0."""
        b = prep or prepare(step0, n)
        out, s = None, step0
        while s < step0 + n:
            e = segment_end(step0, n, s)
            args, kw = seg_args(b, s, e, profile, incremental)
            r = eng.run_steps(*args, want_logs=True, staged=bool(b.get("staged")), **kw)
            run.np_pos += r["np_used"]
            if out is None:
                out = r
            else:
                for k in ("done", "sweep_ms_total", "sweep_launches", "wall_ms", "full_sweeps", "np_used"):
                    out[k] += r[k]
                out["status"] = r["status"]
                for k in ("n_events", "totals", "events"):
                    out[k] = np.concatenate([out[k], r[k]])
            if r["done"] < e - s:
                break
            if config5 and (e - 1) % DEFECT_REFRESH_EVERY == 0:
                refresh_defects()
            s = e
        return out
    run.np_pos = 0

    def barrier():
        eng.sync()
        if dist is not None:
            dist.barrier()

    # ---- parity sample + CPU baseline (rank 0, N=1): GPU and oracle run the same first steps
    step = 0
    base = None
    if N == 1 and not a.no_cpu_baseline and not config5:
        n_chk = 40
        r = run(0, n_chk)
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
        step += r["done"]
    timed_inputs = prepare(step, a.steps)
    # very short runs (<= 16 steps): a hipEvent pair on EVERY sweep launch; otherwise every 4th (<= 64 steps: at least 5 timed
    # launches in the driver's 20-step run) or every 8th.  The pair rides on the
    # launch (hipExtLaunchKernelGGL start / stop events: the dispatch's own timestamps, on the engine's stream), which still
    # costs the timed loop ~4 us per timed launch
    prof_mode = 1 if a.steps <= 16 else 3
    eng.set_option("reserve_batch", a.steps)       # batch buffers / hipEvents: hipMalloc and hipEventCreate stay out of the timed region
    inputs_staged = stage(timed_inputs, profile=prof_mode)
    barrier()
    t0 = time.perf_counter()
    r = run(step, a.steps, profile=prof_mode, prep=timed_inputs)
    barrier()
    dt = allreduce(time.perf_counter() - t0, "MAX")
    assert r["done"] == a.steps and r["status"] == 0, r
    step += a.steps
    # every rank applies every event: the logs of the timed steps must be the same bytes on all ranks
    ranks_agree = None
    if dist is not None:
        import hashlib
        digest = hashlib.sha256(r["events"].tobytes() + r["totals"].tobytes() + r["n_events"].tobytes()).hexdigest()
        box = [None] * max(N, 1)
        dist.all_gather_object(box, digest)
        ranks_agree = all(x == box[0] for x in box)
        assert ranks_agree, f"ranks disagree on the timed steps' event log: {box}"

    # ---- after the timed region: per-phase device time (hipEvents at every phase boundary, cetkmc_get_counters)
    extras_errors = {}

    def guarded(name, fn):
        """The extras run AFTER the timed region; one of them failing (e.g. a transport error at N > 1) must not cost the
        headline line.  A failure is recorded in the JSON line, never hidden."""
        try:
            return fn()
        except Exception as exc:                     # noqa: BLE001
            extras_errors[name] = f"{type(exc).__name__}: {exc}"
            print(f"bench extra '{name}' failed: {exc}", file=sys.stderr)
            return None

    st = {"step": step}          # the extras advance the trajectory one after another

    def do_phases():
        n_ph = min(200, a.steps)
        ph_inputs = prepare(st["step"], n_ph)
        eng.counters(reset=True)
        rp = run(st["step"], n_ph, profile=2, prep=ph_inputs)
        c = eng.counters()
        st["step"] += rp["done"]
        phases = None
        if rp["done"] == n_ph and c["profiled_steps"] == n_ph:
            phases = {k[3:] + "_us_per_step": 1e3 * c[k] / n_ph for k in ("ms_thermal", "ms_interface", "ms_sweep", "ms_reduce",
                                                                           "ms_select_apply")}
            phases["table_interface_us_per_step"] = phases.pop("interface_us_per_step")
            if dist is not None:
                # hipEvent pairs around every collective of the profiled steps (block-sum / event all-gathers, temperature halo)
                phases.update(comm_us_per_step=1e3 * c["ms_comm"] / n_ph, comm_calls_per_step=c["comm_calls"] / n_ph)
            phases.update(steps=n_ph, device_us_per_step=1e3 * rp["wall_ms"] / n_ph,
                          thermal_updates=c["thermal_updates"], table_updates=c["table_updates"],
                          interface_launches=c["interface_launches"],
                          alg_bytes_per_step=(c["alg_bytes_sweep"] + c["alg_bytes_thermal"] + c["alg_bytes_table"]) / n_ph,
                          note="full-sweep loop, hipEvents at every phase boundary (slower than the timed run by the event "
                               "records); thermal / rate table / interface list averaged over their 1-in-20 cadence")
        return phases

    phases = guarded("phases", do_phases) if not a.no_phases else None       # at N > 1 this is where comm_us_per_step comes from

    # ---- the recompute variant of the sweep kernel (nucleation rates evaluated in every sweep instead of looked up)
    def do_recompute():
        n_rc = min(200, a.steps)
        rc_inputs = prepare(st["step"], n_rc)
        eng.set_option("sweep_variant", 2)
        try:
            eng.sync()
            t3 = time.perf_counter()
            rr = run(st["step"], n_rc, profile=1, prep=rc_inputs)
            eng.sync()
            dtr = time.perf_counter() - t3
        finally:
            eng.set_option("sweep_variant", 1)
        st["step"] += rr["done"]
        recompute = None
        if rr["done"] == n_rc and rr["sweep_launches"]:
            ms = rr["sweep_ms_total"] / rr["sweep_launches"]
            ach = B_ALG_SWEEP * (eng.i1 - eng.i0) * L * L / (ms * 1e-3) / 1e9
            recompute = {"kernel": "k_sweep_stream<recompute>", "avg_launch_ms": ms, "launches_timed": int(rr["sweep_launches"]),
                         "achieved": ach, "frac": ach / HBM_PEAK_GBS, "steps_per_s": n_rc / dtr,
                         "note": "sweep_variant 2: streams T instead of the rate table and evaluates the nucleation rate of "
                                 "every bulk empty voxel in the sweep (same bits; tests/test_gpu_parity.py::"
                                 "test_sweep_variants_bit_identical); steps_per_s includes the hipEvent records"}
        return recompute

    recompute = guarded("sweep_recompute", do_recompute) if (not a.no_recompute and N == 1) else None

    # ---- the same loop in exact incremental mode (reported beside `value`, never as it)
    def do_incremental():
        inc_inputs = prepare(st["step"], a.steps)
        stage(inc_inputs, incremental=True)
        barrier()
        t1 = time.perf_counter()
        ri = run(st["step"], a.steps, prep=inc_inputs, incremental=True)
        barrier()
        dti = allreduce(time.perf_counter() - t1, "MAX")
        st["step"] += ri["done"]
        inc = None
        if ri["done"] == a.steps:
            inc = {"steps_per_s": a.steps / dti, "ms_per_step": 1e3 * dti / a.steps, "full_sweeps": int(ri["full_sweeps"]),
                   "steps": a.steps,
                   "note": "cetkmc_run_steps incremental=1 on the following K steps: between temperature updates only the "
                           "rows the previous event made stale are re-evaluated; bit-identical to full sweeps "
                           "(tests/test_gpu_parity.py::test_incremental_mode_bit_identical); not part of `value`"}
        return inc

    inc = guarded("incremental_exact", do_incremental) if (not a.no_incremental and (N == 1 or a.extras_multi)) else None

    # ---- Mode B (super-steps over 8^3 boxes; not the reference's trajectory, own CPU comparator): executed events/s
    def do_mode_b(null_events, const_T=None):
        nb_steps, nb_warm = 40, 8
        tm = 2
        if const_T is not None:
            # the same lattice under a STATIONARY, uniform temperature field (config 2's field; no melt pool, no updates): the
            # regime Mode B's null events are meant for -- every rate within a few decades, acceptance of the order of one half
            a0_, a1_ = max(0, eng.i0 - 2), min(L, eng.i1 + 2)
            eng.upload_planes(a0_, a1_, T=np.full((a1_ - a0_, L, L), float(const_T)))
            tm = 0
        # untimed warm-up (first launch of the Mode B kernels loads their code objects: ~10 ms once per process)
        rw = eng.run_supersteps(st["step"], nb_warm, 8, DEFECT_FRACTION, seed=SEED, thermal_mode=tm,
                                q_planes=synthetic.laser_planes(L, st["step"], nb_warm) if tm == 2 else None, null_events=null_events)
        st["step"] += rw["done"]
        qb = synthetic.laser_planes(L, st["step"], nb_steps) if tm == 2 else None
        barrier()
        t2 = time.perf_counter()
        rb = eng.run_supersteps(st["step"], nb_steps, 8, DEFECT_FRACTION, seed=SEED, thermal_mode=tm, q_planes=qb,
                                null_events=null_events)
        barrier()
        dtb = allreduce(time.perf_counter() - t2, "MAX")
        st["step"] += rb["done"]
        n_exec = allreduce(float(rb["n_exec"].sum()), "SUM")          # every rank counts the events of its own boxes
        mode_b = None
        if rb["done"] == nb_steps:
            # algorithmic bytes of one super-step on this rank: the ordinary rate sweep (9 B per owned voxel) + the box picks,
            # which read the table entry and class byte of the active octant's voxels (1/8 of the owned voxels) once more;
            # apply / touched-voxel upkeep are scattered ~0.2 KB-per-voxel gathers (profiles/: measured traffic), not counted
            alg = B_ALG_SWEEP * n_own_b * (1.0 + 1.0 / 8.0)
            ach = alg / (dtb / nb_steps) / 1e9
            mode_b = {"executed_events_per_s": n_exec / dtb, "ms_per_superstep": 1e3 * dtb / nb_steps,
                      "events_per_superstep": n_exec / nb_steps, "boxes": (L // 8) ** 3, "supersteps": nb_steps,
                      "warmup_supersteps": nb_warm, "null_events": bool(null_events),
                      "temperature_field": "moving melt pool (update_temperature every 20 super-steps)" if const_T is None else
                                           f"stationary, uniform {float(const_T):.0f} K (no updates)",
                      "acceptance": n_exec / nb_steps / max(1.0, allreduce(float(n_own_b) / 512.0, "SUM")),
                      "simulated_time_per_superstep_s": float(np.mean(rb["dt_event"] * rb["n_exec"])) if len(rb["n_exec"]) else None,
                      "roofline": {"bound": "hbm", "unit": "GB/s", "peak": HBM_PEAK_GBS, "achieved": ach, "frac": ach / HBM_PEAK_GBS,
                                   "alg_bytes_per_superstep": alg,
                                   "alg_bytes_definition": "9 B x owned voxels (rate sweep: class u8 + table f64) + 9 B x owned voxels / 8 "
                                                           "(box picks re-read the active octant's leaves); this rank",
                                   "timing": "wall time of the timed super-steps / their number (all launches of a super-step)",
                                   "kernel_avg_us": mode_b_kernels, "kernel_avg_us_source": mode_b_src,
                                   "note": "Mode B's super-step is a chain of seven launches of which only the sweep streams; the "
                                           "fraction is whole-super-step bytes over whole-super-step time, the per-kernel "
                                           "durations come from the committed rocprofv3 trace of this command"},
                      "note": "cetkmc_run_supersteps box=8 on the lattice left by the runs above: one full rate sweep per "
                              "super-step, every box picks <= 1 event from its active octant (boxes sharded with the slabs; "
                              "boundary-layer events exchanged with the neighbour ranks every super-step)" +
                              ("; null events: a box executes its pick with probability R_box / R_max (every event fires in "
                               "proportion to its rate, DESIGN.md section 12) -- run_kmc(mode='B') default" if null_events else
                               "; every non-idle box executes (no null events: boxes with few events are over-sampled)") +
                              "; bit-identical to oracle orc_run_supersteps (tests/test_gpu_mode_b.py); not part of `value`"}
        return mode_b

    n_own_b = (eng.i1 - eng.i0) * L * L
    mode_b_kernels, mode_b_src = None, None
    try:
        import re as _re
        cand_files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_summary.json")),
                            key=lambda f: int(_re.search(r"r(\d+)", os.path.basename(f)).group(1)), reverse=True)
        for f in cand_files:
            km = json.load(open(f)).get("kernels_modes")
            if km and "k_domain_touch" in km:
                mode_b_kernels = {k: round(km[k]["avg_us"], 2) for k in ("k_sweep_stream", "k_plane_reduce", "k_select", "k_domain_pick8",
                                                                          "k_domain_rmax", "k_domain_apply", "k_domain_touch",
                                                                          "k_super_commit", "k_ifc_relist", "k_interface_part") if k in km}
                mode_b_src = os.path.relpath(f, ROOT)
                break
    except Exception:
        pass
    mode_b_ok = not a.no_mode_b and L % 8 == 0 and (L // N) % 8 == 0
    mode_b = guarded("mode_b", lambda: do_mode_b(False)) if mode_b_ok else None
    mode_b_null = guarded("mode_b_null_events", lambda: do_mode_b(True)) if mode_b_ok else None
    # (last: it replaces the temperature field)
    mode_b_null_ct = guarded("mode_b_null_events_stationary_T", lambda: do_mode_b(True, const_T=3000.0)) if mode_b_ok else None

    ev_over_ms = guarded("event_overhead", lambda: eng.event_overhead(50))

    # ---- what the boundary costs when it hands over host buffers: the lattice up before and down after a job
    def do_pcie():
        eng.sync()
        t3 = time.perf_counter()
        d = eng.download_planes(eng.i0, eng.i1, state=True, theta=True, phi=True, T=True, defects=True)
        t_dn = time.perf_counter() - t3
        dn_bytes = sum(x.nbytes for x in d.values())
        job = 2000
        return {"upload_ms": 1e3 * t_up, "upload_GBps": upload_bytes / t_up / 1e9, "download_ms": 1e3 * t_dn,
                "download_GBps": dn_bytes / t_dn / 1e9, "bytes_up": int(upload_bytes), "bytes_down": int(dn_bytes),
                "executed_events_per_s_timed_steps": a.steps / (dt + t_up + t_dn),
                "executed_events_per_s_2000_step_job": job / (job * dt / a.steps + t_up + t_dn),
                "note": "rank 0's slab (+halo on the way up) in the plane API's types (u8 state/defects, f64 theta/phi/T) from and to "
                        "pageable NumPy arrays; the timed region itself starts with everything resident in HBM, so these rates "
                        "are context, never `value`"}
    pcie = guarded("pcie_inclusive", do_pcie)
    cand = float(np.sum(r["n_events"].astype(np.float64)))     # identical on every rank (global counts)
    steps_per_s = a.steps / dt
    sweep_ms = r["sweep_ms_total"] / max(r["sweep_launches"], 1)
    n_own = (eng.i1 - eng.i0) * L * L
    achieved = B_ALG_SWEEP * n_own / (sweep_ms * 1e-3) / 1e9 if sweep_ms > 0 else 0.0
    # from the committed rocprofv3 summary of the same command (profiles/rNN_summary.json, latest round): the kernel's own
    # average duration in the kernel trace and the PMC traffic per launch -- not measured in this run
    traffic, traffic_src, rocprof_ms = None, None, None
    import re
    rounds = [(int(m.group(1)), f) for f in glob.glob(os.path.join(ROOT, "profiles", "r*_summary.json"))
              for m in [re.fullmatch(r"r(\d+)_summary\.json", os.path.basename(f))] if m]
    rocprof_kernels = None
    for _, summary_path in sorted(rounds, reverse=True):
        try:
            allk = json.load(open(summary_path))["kernels"]
            kk = allk["k_sweep_stream"]
            if N == 1 and L == 256 and "hbm_bytes_per_launch" in kk:
                traffic, traffic_src, rocprof_ms = kk["hbm_bytes_per_launch"], os.path.relpath(summary_path, ROOT), kk["avg_us"] * 1e-3
                rocprof_kernels = {k: round(allk[k]["avg_us"], 2) for k in ("k_sweep_stream", "k_plane_reduce", "k_select_apply",
                                                                            "k_thermal_tiles16", "k_thermal_tiles", "k_thermal_march", "k_rate_table",
                                                                            "k_interface", "k_clear_row_flags") if k in allk and "avg_us" in allk[k]}
                break
        except Exception:
            pass
    summary_src = traffic_src                 # the committed profile (kernel averages; traffic too unless measured in this run)
    traffic_live = live_bytes is not None
    if traffic_live:
        traffic, traffic_src = live_bytes, live_src
    if phases is not None and rocprof_kernels:
        per_update = sum(rocprof_kernels.get(k, 0.0) for k in ("k_thermal_tiles16", "k_thermal_tiles", "k_thermal_march", "k_rate_table", "k_interface", "k_clear_row_flags"))
        phases["rocprof"] = {
            "kernel_avg_us": rocprof_kernels, "source": summary_src,
            "select_apply_plus_reduce_plus_interface_us_per_step":
                rocprof_kernels.get("k_select_apply", 0.0) + rocprof_kernels.get("k_plane_reduce", 0.0) + rocprof_kernels.get("k_interface", 0.0) / 20,
            "per_update_work_us_per_step": per_update / 20,
            "note": "the same loop in the committed rocprofv3 kernel trace (kernel durations without the hipEvent records: every "
                    "phase boundary above carries one record, ~2 us); interface / thermal / table kernels run once per 20 steps"}
    def do_hbm_512():
        """The same sweep kernel where the Infinity Cache cannot help: config 5's lattice edge (512^3, working set 1.2 GB) on
        this one GPU.  Lattice by the same fill rule, 12 steps of the exact loop (melt-pool update at step 0), the sweep
        launches timed by their own start / stop events."""
        L5 = 512
        e5 = cetkmc.Engine(L5, impurity_c=IMPURITY_C, device=0)
        try:
            f5 = synthetic.planes(L5, 0, L5, seed=SEED)
            e5.upload_planes(0, L5, *f5)
            del f5
            e5.set_prev_state(None)
            n5 = 12
            u_pick, u_def, u_np = streams(n5 + 4, SEED)
            r5 = e5.run_steps(0, 4, DEFECT_FRACTION, u_pick, u_def, u_np, rng_mode=1, seed=SEED, thermal_mode=2,
                              q_planes=synthetic.laser_planes(L5, 0, 4), use_latent=True)
            rr = e5.run_steps(4, n5, DEFECT_FRACTION, u_pick[4:], u_def[4:], u_np[r5["np_used"]:], rng_mode=1, seed=SEED, thermal_mode=2,
                              q_planes=synthetic.laser_planes(L5, 4, n5), use_latent=True, profile=1)
            ms5 = rr["sweep_ms_total"] / max(rr["sweep_launches"], 1)
            nv5 = float(L5) ** 3
            ach5 = B_ALG_SWEEP * nv5 / (ms5 * 1e-3) / 1e9
            return {"L": L5, "kernel": "k_sweep_stream<table> (full-wave rows)", "avg_launch_ms": ms5, "launches_timed": int(rr["sweep_launches"]),
                    "alg_bytes_per_launch": B_ALG_SWEEP * nv5, "achieved": ach5, "unit": "GB/s", "peak": HBM_PEAK_GBS, "frac": ach5 / HBM_PEAK_GBS,
                    "working_set_bytes": int(B_ALG_SWEEP * nv5), "fits_infinity_cache": False,
                    "device_ms_per_step": rr["wall_ms"] / n5,
                    "traffic": live512_bytes, "traffic_measured_in_run": live512_bytes is not None, "traffic_source": live512_src,
                    "traffic_over_algorithmic": (live512_bytes / (B_ALG_SWEEP * nv5)) if live512_bytes else None,
                    "note": "measured live in this run on the same GPU; committed trace + PMC of the same workload: profiles/r03_512_*"}
        finally:
            e5.close()

    hbm_512 = guarded("hbm_512", do_hbm_512) if (N == 1 and L == 256 and not a.no_512 and not config5) else None

    workloads = {
        "config3": f"config 3: {L}^3 voxel lattice, one MI355X",
        "config4": f"config 4: the {L}^3 lattice of config 3 split into {N} axis-0 slabs, one rank per MI355X (strong scaling)",
        "config5": f"config 5: {L}^3 voxel lattice in {N} axis-0 slab(s), impurity_c={IMPURITY_C}, defects.track_defects "
                   f"refreshed after every step % {DEFECT_REFRESH_EVERY} == 0",
        "config3-weak": f"config 3 workload at {L}^3 in {N} axis-0 slabs (~1.68e7 voxels per GPU, --scaling weak)",
    }
    out = {
        "metric": "kmc_events_per_sec", "value": steps_per_s, "unit": "executed events/s",
        "n_gpus": N, "steps": a.steps, "warmup": a.warmup, "ms_per_step": 1e3 * dt / a.steps,
        "higher_is_better": True, "scaling": scaling, "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {
            "workload": workloads[cfg_name] + "; k<L/4 pre-filled W/Re/C, moving Gaussian melt-pool T-field "
                        "(update_temperature every 20 steps), exact Mode A: 1 executed event per full rate sweep",
            "baseline_config": cfg_name, "L": L, "impurity_c": IMPURITY_C, "defect_fraction": DEFECT_FRACTION,
            "rng_mode": "counter", "transport": a.transport if N > 1 else "none",
            "inputs": ("uniform streams and laser source planes of the timed batch copied to HBM before the timed region "
                       "(cetkmc_stage_inputs)") if inputs_staged else
                      "host buffers handed over inside the timed region (batch split by the defect refreshes)",
            "event_definition": "value = EXECUTED events/s of the exact loop (one per sweep, = steps/s); "
                                "candidate_events_per_s = entries of get_event_rates' list evaluated, reduced and scanned per "
                                "second (round 1 reported that number as value)",
        },
        "steps_per_s": steps_per_s,
        "executed_events_per_s_mode_a": steps_per_s,
        "executed_events_per_s_incremental": inc["steps_per_s"] if inc else None,
        "executed_events_per_s_mode_b": mode_b["executed_events_per_s"] if mode_b else None,
        "executed_events_per_s_mode_b_null_events": mode_b_null["executed_events_per_s"] if mode_b_null else None,
        "executed_events_per_s_mode_b_null_events_stationary_T": mode_b_null_ct["executed_events_per_s"] if mode_b_null_ct else None,
        "candidate_events_per_s": cand / dt,
        "candidate_events_per_step": cand / a.steps, "voxel_updates_per_s": float(L) ** 3 * steps_per_s,
        "device_ms_per_step": r["wall_ms"] / a.steps,
        "roofline": {"bound": "hbm", "kernel": "k_sweep_stream<table>", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src,
                     # the sweep's working set (rate table f64 + class u8 of the owned planes) is re-read every step and only
                     # rewritten by a temperature update: when it fits the 256 MiB Infinity Cache the bytes above are served
                     # from it, not from DRAM -- `achieved` is then fabric / cache bandwidth against the HBM peak
                     "working_set_bytes": int(B_ALG_SWEEP * n_own), "infinity_cache_bytes": 256 * 2 ** 20,
                     "fits_infinity_cache": bool(B_ALG_SWEEP * n_own <= 256 * 2 ** 20),
                     "memory_level": ("Infinity-Cache resident between temperature updates: achieved is cache / fabric bandwidth, NOT "
                                      "DRAM; see hbm_512 for the cache-free figure") if B_ALG_SWEEP * n_own <= 256 * 2 ** 20 else
                                     "HBM (working set exceeds the Infinity Cache)",
                     "hbm_512": hbm_512,
                     # the whole step against the same peak: algorithmic bytes of everything a step issues (sweep 9 B/voxel every
                     # step; temperature update + rate table in one launch: T read, T and table entry written = 24 B/voxel every 20 steps) over the device time
                     # of a step -- the loop is latency-bound between its three launches (selection + application on one block)
                     "step_level": {"alg_bytes_per_step": (B_ALG_SWEEP + 24.0 / 20.0) * n_own,
                                    "achieved": (B_ALG_SWEEP + 24.0 / 20.0) * n_own / (r["wall_ms"] / a.steps * 1e-3) / 1e9,
                                    "frac": (B_ALG_SWEEP + 24.0 / 20.0) * n_own / (r["wall_ms"] / a.steps * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                    "unit": "GB/s"},
                     "traffic_measured_in_run": traffic_live,
                     "traffic_over_algorithmic": (traffic / (B_ALG_SWEEP * n_own)) if traffic else None,
                     "avg_launch_ms": sweep_ms, "launches_timed": int(r["sweep_launches"]),
                     "timing": "hipEvent pairs attached to the sweep launches of the timed region (hipExtLaunchKernelGGL start / "
                               "stop events on the engine's stream; several slabs per process: hipEventRecord around the launches)",
                     # what two hipEventRecord calls add around a launch (measured in this run around an empty kernel): the
                     # phase table's boundaries and multi-slab handles are timed that way, the sweep launches above are not
                     "event_pair_overhead_ms": ev_over_ms,       # two records around an EMPTY kernel (its ~3.5 us included)
                     "rocprof_avg_launch_ms": rocprof_ms,        # committed kernel trace of the same command
                     "rocprof_source": summary_src,
                     "frac_rocprof": (B_ALG_SWEEP * n_own / (rocprof_ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if rocprof_ms else None,
                     "launches_in_timed_region": int(r["full_sweeps"]),
                     "alg_bytes_per_voxel": B_ALG_SWEEP, "voxels_per_launch": n_own,
                     # BASELINE.md section 4 / SURVEY 8(d) pre-build accounting: 26.8 B per voxel per step (state u8 + T +
                     # theta + phi f64 + defects u8 streamed every sweep, thermal 16 B / 20) x steps/s.  The build does
                     # NOT stream theta/phi/defects (touched at interface voxels only), so this over-counts moved bytes;
                     # shown only because BASELINE.md's 40 % target (7.1 k steps/s at 256^3) is phrased in it.
                     "baseline_md_accounting": {"note": "SURVEY 8(d)'s pre-build byte count (state + T + theta + phi + defects streamed every "
                                                        "sweep) x steps/s; the build does not stream theta / phi / defects, so this counts "
                                                        "bytes that are not moved -- shown only because BASELINE.md's 40 % target is phrased "
                                                        "in it; `frac` above is the figure of merit",
                                                "bytes_per_voxel_step": 26.8,
                                                "achieved": 26.8 * float(L) ** 3 * steps_per_s / 1e9, "unit": "GB/s",
                                                "frac": 26.8 * float(L) ** 3 * steps_per_s / 1e9 / HBM_PEAK_GBS}},
    }
    if base is not None:
        nev = base["n_events"].astype(np.float64)
        (t1c, s1, sec1), (tm, sm, secm) = base["legs"]["single"], base["legs"]["multi"]
        multi = sm > 0
        out["cpu_baseline"] = {
            "value": (sm / secm) if multi else (s1 / sec1), "unit": "executed events/s", "cores": tm if multi else t1c, "kind": "port",
            "sample": (f"steps {s1}..{s1 + sm} of the same {L}^3 workload on the C oracle with its row/thermal loops on "
                       f"{tm} host threads ({secm:.1f} s, incl. thermal updates)") if multi else
                      f"first {s1} steps of the same {L}^3 workload on the C oracle, 1 thread ({sec1:.1f} s)",
            "candidate_events_per_s": (float(nev[s1:s1 + sm].sum()) / secm) if multi else (float(nev[:s1].sum()) / sec1),
            "single_core": {"value": s1 / sec1 if s1 else None, "unit": "executed events/s",
                            "candidate_events_per_s": float(nev[:s1].sum()) / sec1 if s1 else None,
                            "cores": 1, "sample": f"first {s1} steps, scalar port ({sec1:.1f} s)"},
            "host_cores_available": os.cpu_count(), "parity_first_steps": base["parity"],
            "rng_mode": "counter species draw (cetkmc_run_args.rng_mode = 1) on both sides: the reference's per-candidate NumPy draw "
                        "(kmc_event_rates.py:65; 65 536 MT draws per step at 256^2) is pinned against the reference at L <= 32 only",
            # SURVEY 8(d) / BASELINE.md section 2: the reference itself cannot travel to this box; its own speed was measured
            # once, in the survey container (reference source unmodified, identity-jit stand-in for the missing numba, 1 core)
            "reference_python": {"provenance": "BASELINE.md section 2: reference source, pure-Python mode (numba absent), 1 of 8 x86 "
                                               "cores of the survey container, python 3.10 / numpy 2.2 -- NOT measured in this run",
                                 "sweep_s_16": 0.217, "sweep_s_32": 1.33, "sweep_s_48": 4.95, "us_per_voxel_per_step": 42.0,
                                 "run_kmc_step_s_30": 1.64, "executed_events_per_s_30": 0.61,
                                 "sweep_s_256_extrapolated": 700.0, "executed_events_per_s_256_extrapolated": 1.0 / 700.0},
        }
    if config5:
        out["defect_refresh"] = {"every_steps": DEFECT_REFRESH_EVERY, "calls": len(refresh_ms),
                                 "ms_per_call": float(np.mean(refresh_ms)) if refresh_ms else None,
                                 "note": "gather of the carbon sites + host Bernoulli draws + sparse mask upload; inside the timed "
                                         "region whenever a timed step is a multiple of 200"}
    if pcie is not None:
        out["pcie_inclusive"] = pcie
    if phases is not None:
        out["phases"] = phases
    if recompute is not None:
        out["sweep_recompute"] = recompute
    if inc is not None:
        out["incremental_exact"] = inc
    if mode_b is not None:
        out["mode_b"] = mode_b
    if mode_b_null is not None:
        out["mode_b_null_events"] = mode_b_null
    if mode_b_null_ct is not None:
        out["mode_b_null_events_stationary_T"] = mode_b_null_ct
    if dist is not None:
        box = [None] * N
        dist.all_gather_object(box, {"rank": rank, "owned_planes": [eng.i0, eng.i1], "sweep_avg_launch_ms": sweep_ms,
                                     "achieved": achieved, "frac": achieved / HBM_PEAK_GBS, "unit": "GB/s",
                                     "working_set_bytes": int(B_ALG_SWEEP * n_own)})
        out["roofline"]["per_rank"] = box
        out["multi_rank_checks"] = {"transport_selftest": comm_check, "ranks_agree_on_event_log": ranks_agree,
                                    "note": "self-test: patterned all-gather + neighbour exchange verified before timing (wall "
                                            "microseconds per call incl. the stream synchronisation); event log: sha256 of the "
                                            "timed steps' events / totals / counts equal on every rank"}
    if extras_errors:
        out["extras_errors"] = extras_errors
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
