"""Rejection-free KMC stepping loop on the GPU (drop-in for the reference ``kmc_simulation.py``).

``run_kmc`` keeps the reference's signature, return tuple, prints and ``metrics.csv`` contract
(kmc_simulation.py:203-398).  The hot loop -- thermal update every 20 steps, full-lattice rate
sweep, cumulative-rate event pick, lattice update -- runs on the device in batches through
``cetkmc_run_steps`` with NO host round trip per step; the host only

  * pre-draws the random streams from the very generators the reference uses (CPython
    ``random`` for the pick / defect / time draws, NumPy's legacy global stream for the
    deposition-species and orientation draws) and afterwards rewinds them to exactly the
    position the reference would have reached, and
  * runs the analysis that is not part of the hot path (defect-mask refresh and metrics every
    ``METRIC_UPDATE_STEP`` steps).

``run_kmc(mode="B", box=8)`` runs the same model through the synchronous super-step engine (``cetkmc_run_supersteps``:
thousands of events per rate sweep, NOT the reference's trajectory -- DESIGN.md "Mode B"), with the same prints and the same
18-column ``metrics.csv``.

There is no CPU fallback; without libcetkmc_hip.so + a GPU this raises.
"""
import os
import random

import numpy as np
import pandas as pd

from constants import (ATOMIC_SPACING_W, CET_CHECK_INTERVAL, DEFECT_ID, LATTICE_SIZE,  # noqa: F401
                       METRIC_UPDATE_STEP, N_STEPS, NU_DEP, RANDOM_SEED, RATE_THRESHOLD, T_MELT, T_SUB,
                       VOXEL_SIZE)
from defects import introduce_defects, refresh_defects_device
from kmc_event_rates import get_event_rates  # noqa: F401  (re-exported like the reference)
from lattice_init import initialize_lattice
from metrics import compute_CET, compute_metrics, compute_metrics_device, detect_CET_transition  # noqa: F401
from constants import CET_AR_THRESHOLD, CET_EQ_THRESHOLD
from thermal_solver import update_temperature_cet as update_temperature  # noqa: F401

# What the last run_kmc call observed besides its return tuple (the reference's signature has no room for it):
#   min_margin  smallest selection margin of the run (cetkmc_run_result.min_margin: distance of r = u * total from the nearer
#               end of the chosen event's interval of the cumulative rate sum, relative to the total).  The device sums rates by
#               a balanced tree, the reference left to right (kmc_simulation.py:259,265-274); the two sums differ by <= ~1e-13
#               relative, so only a pick with a margin below MARGIN_WARN could have gone to the neighbouring event there.
last_run_info = {}
MARGIN_WARN = 1e-12

THERMAL_EVERY = 20          # kmc_simulation.py:248
THERMAL_DT = 1e-6           # kmc_simulation.py:250
_MAX_STREAM_DOUBLES = 1 << 25   # host staging cap for the pre-drawn NumPy stream (256 MiB)


def _advance_to(engine, first, last, L, defect_fraction, rng_mode=0, seed=0, incremental=True, thermal_mode=1):
    """Run steps first..last (inclusive) on the device.  Returns (steps_done, terminated,
    last_total, dt_sum_increments) with both host generators left where the reference's would be."""
    dts = []
    _advance_to.min_margin = 1.0
    step = first
    terminated = False
    last_total = 0.0
    per = 3 if defect_fraction > 0.0 else 2
    slack = 8
    while step <= last:
        n = last - step + 1
        if rng_mode == 0:
            # One deposition-species draw per finite-rate deposition candidate per step (kmc_event_rates.py:63-65) + 2
            # orientation draws.  The candidate count is asked from the device (one sweep) instead of assuming the
            # whole top plane (L*L): at L = 256 that assumption means ~13 M doubles drawn, uploaded and re-drawn per
            # 200-step batch.  A batch that still runs short stops with status 2 and is continued from there.
            n_dep_now = int(engine.rate_sweep()[2])
            per_step = min(L * L, n_dep_now + slack) + 2
        else:
            per_step = 2
        n = max(1, min(n, _MAX_STREAM_DOUBLES // per_step))
        py_state = random.getstate()
        draws = np.array([random.random() for _ in range(per * n)], dtype=np.float64).reshape(n, per)
        np_state = np.random.get_state()
        u_np = np.random.random(n * per_step)
        res = engine.run_steps(step, n, defect_fraction, draws[:, 0], draws[:, 1] if per == 3 else None, u_np,
                               rng_mode=rng_mode, seed=seed, thermal_mode=thermal_mode, thermal_dt=THERMAL_DT,
                               incremental=incremental)
        done = res["done"]
        _advance_to.min_margin = min(_advance_to.min_margin, res["min_margin"])
        # rewind both generators to what the executed steps consumed
        np.random.set_state(np_state)
        if res["np_used"]:
            np.random.random(res["np_used"])
        if done < n:
            random.setstate(py_state)
            for _ in range(per * done):
                random.random()
        for s in range(done):
            total = res["totals"][s]
            dts.append(max(-np.log(max(1e-12, draws[s, per - 1])) / total, 1e-12))   # kmc_simulation.py:331
        step += done
        if res["status"] == 1:
            terminated = True
            last_total = float(res["totals"][done]) if len(res["totals"]) > done else 0.0
            break
        if res["status"] == 2 and done == 0:
            if slack >= L * L:
                raise RuntimeError("pre-drawn NumPy stream too small for a single step")
            slack = min(L * L, 4 * slack + 64)       # the estimate was short (candidates appeared): widen and retry
    return step - first, terminated, last_total, dts


def save_checkpoint(path, fields, defects_mask, next_step, total_time, nucleation_count, metrics_data, cet_detected, extra=None):
    """Everything needed to continue a run bit-identically: the five lattice fields, the defect mask,
    the loop counters, the metrics rows so far and the state of BOTH host generators (CPython
    ``random`` and NumPy's legacy global stream).  The reference has no resume."""
    import json
    py = random.getstate()
    npst = np.random.get_state()
    tmp = path + ".tmp.npz"        # written beside the target and renamed into place: a crash mid-write keeps the old file
    np.savez_compressed(
        tmp, state=fields["state"].astype(np.int8), theta=fields["theta"], phi=fields["phi"], T=fields["T"],
        defects=np.asarray(defects_mask).astype(np.int8), next_step=next_step, total_time=total_time,
        nucleation_count=nucleation_count, cet_detected=bool(cet_detected),
        py_version=py[0], py_mt=np.array(py[1], dtype=np.uint64), py_gauss=np.array([np.nan if py[2] is None else py[2]]),
        np_mt=npst[1], np_pos=npst[2], np_has_gauss=npst[3], np_cached=npst[4],
        metrics_json=np.array(json.dumps(metrics_data, default=lambda o: o.item() if hasattr(o, "item") else str(o))),
        extra_json=np.array(json.dumps(extra or {})))          # mode "B": super-step index, temperature updates applied, seed, box
    os.replace(tmp, path)


def load_checkpoint(path):
    """Inverse of save_checkpoint; also restores both host generators."""
    import json
    z = np.load(path, allow_pickle=False)
    g = float(z["py_gauss"][0])
    random.setstate((int(z["py_version"]), tuple(int(x) for x in z["py_mt"]), None if np.isnan(g) else g))
    np.random.set_state(("MT19937", z["np_mt"], int(z["np_pos"]), int(z["np_has_gauss"]), float(z["np_cached"])))
    return dict(state=z["state"].astype(np.int64), theta=z["theta"], phi=z["phi"], T=z["T"],
                defects=z["defects"].astype(np.int64), next_step=int(z["next_step"]), total_time=float(z["total_time"]),
                nucleation_count=int(z["nucleation_count"]), cet_detected=bool(z["cet_detected"]),
                metrics_data=json.loads(str(z["metrics_json"])),
                extra=json.loads(str(z["extra_json"])) if "extra_json" in z.files else {})


def run_kmc(
    L: int = LATTICE_SIZE,
    n_steps: int = N_STEPS,
    temp: float = T_SUB,
    defect_fraction: float = 0.0,
    n_seeds: int = 5,
    impurity_c: float = 0.0,
    output_prefix: str = "cet_run",
    *,
    checkpoint_every: int = 0,
    resume_from: str = None,
    incremental: bool = True,
    nu_dep: float = None,
    mode: str = "A",
    box: int = 8,
    null_events: bool = True,
    thermal_cadence: str = "events",
    seed: int = None,
    metrics_every: int = METRIC_UPDATE_STEP,
    thermal_updates: bool = True,
):
    """KMC microstructure evolution with natural defect injection (same contract as the
    reference).  ``defect_fraction`` is the per-event probability that the just-updated voxel
    becomes a defect.

    Extensions (keyword-only, not in the reference): ``checkpoint_every=k`` writes
    ``outputs/<prefix>/checkpoint.npz`` every k steps (mode "B": after the super-step that reaches each multiple of k
    executed events); ``resume_from=path`` continues such a run --
    the continued run is bit-identical to an uninterrupted one (lattice, time, CSV, RNG streams), in both modes;
    ``incremental=False`` re-evaluates the whole lattice on every step like get_event_rates does (the
    default re-evaluates only the rows an event made stale between temperature updates -- same results);
    ``nu_dep`` overrides constants.NU_DEP (deposition attempt frequency = growth velocity V of the G-V sweep
    driver gv_sweep.py; the reference can only change it by editing constants.py);
    ``seed`` replaces constants.RANDOM_SEED for both host generators (and keys Mode B's counter uniforms);
    ``metrics_every`` replaces constants.METRIC_UPDATE_STEP (cadence of the metrics rows and of the defect-mask refresh);
    ``thermal_updates=False`` keeps the initial temperature field (no update_temperature_cet calls: a stationary
    environment, used to compare the two stepping modes without the reference's event-count thermal clock).

    ``mode="B"``: synchronous super-steps over ``(L/box)**3`` boxes (cetkmc_run_supersteps; ``box == L`` is the
    single-domain case = the exact loop with counter uniforms).  ``n_steps`` keeps its meaning -- the number of EXECUTED
    events -- and the run ends with the first super-step that reaches it (it may overshoot by less than one super-step;
    the last row's ``Step`` says by how much).  ``Step`` of a row = index of the last executed event, rows are written by
    the first super-step that reaches each multiple of ``metrics_every`` (and by the last one), ``Time`` advances per
    executed event as kmc_simulation.py:331-332 does.  ``null_events`` (default on): boxes execute with probability
    R_box / R_max, which makes every event's frequency proportional to its rate as in the reference's global pick.
    ``thermal_cadence="events"`` (default) keeps the reference's cadence of one temperature update per 20 executed events
    (kmc_simulation.py:248-250): before every super-step the field is brought to ``executed // 20 + 1`` updates;
    ``"supersteps"`` updates once per 20 super-steps inside the engine (throughput setting for large lattices: the
    temperature history per executed event then differs from the reference's)."""
    import cetkmc
    if mode not in ("A", "B"):
        raise ValueError("mode must be 'A' (exact, one event per sweep) or 'B' (super-steps)")
    if thermal_cadence not in ("events", "supersteps"):
        raise ValueError("thermal_cadence must be 'events' or 'supersteps'")
    run_seed = RANDOM_SEED if seed is None else int(seed)
    metrics_every = int(metrics_every)

    output_dir = f"outputs/{output_prefix}"
    os.makedirs(output_dir, exist_ok=True)
    ckpt = None
    if resume_from:
        ckpt = load_checkpoint(resume_from)
        state, theta, phi, T = ckpt["state"], ckpt["theta"], ckpt["phi"], ckpt["T"]
        atom_type = state.copy()
        defects_mask = ckpt["defects"]
    else:
        np.random.seed(run_seed)
        random.seed(run_seed)
        state, theta, phi, T, atom_type = initialize_lattice(
            lattice_size=L, n_seeds=n_seeds, T_sub=temp, impurity_c=impurity_c)
        defects_mask, defect_density = introduce_defects(state, atom_type, T, apply_to_state=False)

    nu_dep_eff = NU_DEP if nu_dep is None else float(nu_dep)
    G = (T_MELT - T_SUB) / (L * VOXEL_SIZE)
    R = nu_dep_eff * 2.74e-10 / VOXEL_SIZE
    R_phys = nu_dep_eff * ATOMIC_SPACING_W
    G_over_R_phys = G / R_phys

    params = cetkmc.default_params(impurity_c)
    params.nu_dep = nu_dep_eff
    engine = cetkmc.Engine(L, impurity_c=impurity_c, params=params)
    engine.upload(state, theta, phi, T, defects_mask)
    n_flagged = int(np.sum(defects_mask))
    if mode == "B" and not (box == L or (box in (8, 10, 12, 14, 16) and L % box == 0)):
        engine.close()
        raise ValueError("mode 'B': box must be even, 8..16, and divide L (or equal L: single domain)")

    total_time = 0.0
    min_margin = 1.0
    metrics_data = []
    cet_detected = False
    step = -1
    next_step = 0
    nuc_offset = 0
    if ckpt:
        total_time, metrics_data, cet_detected = ckpt["total_time"], ckpt["metrics_data"], ckpt["cet_detected"]
        next_step, nuc_offset = ckpt["next_step"], ckpt["nucleation_count"]
        step = next_step - 1
    def metrics_row(step, refresh_defects):
        """kmc_simulation.py:335-389 for the lattice as it stands after event index `step` -- WITHOUT moving the lattice: the
        defect mask is refreshed from the carbon sites only, grains are clustered and species counted on the GPU."""
        nonlocal n_flagged, cet_detected
        if refresh_defects:
            n_flagged, _ = refresh_defects_device(engine)       # kmc_simulation.py:335-338
        m = compute_metrics_device(engine, L ** 3, defects_count=n_flagged, voxel_size=VOXEL_SIZE)
        counts = engine.species_counts()
        defect_voxels = int(counts[DEFECT_ID])
        m["Defect_voxel_count"] = defect_voxels
        m["DefectDensity"] = float(defect_voxels / (L ** 3))
        if (not cet_detected) and detect_CET_transition(m):
            cet_detected = True
            print(f"CET detected at step {step} (G/R={G / R:.2e})")
        row = {
            "Step": step,
            "Time": total_time,
            "AspectRatio": m["AspectRatio"],
            "EquiaxedFraction": m["EquiaxedFraction"],
            "NucleationDensity": m["NucleationDensity"],
            "DefectDensity": m["DefectDensity"],
            "AvgGrainSize": m["AvgGrainSize"],
            "GrainCount": m["GrainCount"],
            "W_Count": int(counts[1]),
            "Re_Count": int(counts[2]),
            "C_Count": int(counts[3]),
            "NucleationCount": nuc_offset + engine.nucleation_count(),
            "G_over_R": (G / R) if R > 0 else np.inf,
            "G_phys": G,
            "R_phys": R_phys,
            "G_over_R_phys": G_over_R_phys,
            # compute_CET re-clusters the same lattice (metrics.py:99-101): same AR / equiaxed fraction
            "CET_Class": "Equiaxed" if (m["AspectRatio"] < CET_AR_THRESHOLD and m["EquiaxedFraction"] > CET_EQ_THRESHOLD) else "Columnar",
            "CET_Detected": cet_detected,
        }
        metrics_data.append(row)
        print(
            f"Step {step}: AR={row['AspectRatio']:.2f}, "
            f"EqFrac={row['EquiaxedFraction']:.2f}, "
            f"NucDens={row['NucleationDensity']:.3e}, "
            f"DefectDens={row['DefectDensity']:.3e}, "
            f"CET={row['CET_Class']}, "
            f"Detected={row['CET_Detected']}, "
            f"Time={row['Time']:.2e}s"
        )

    while mode == "A" and next_step < n_steps:
        # next step after which the host has work: metrics (and, on multiples of
        # metrics_every, the defect-mask refresh) -- kmc_simulation.py:335-341
        stop = next_step if next_step % metrics_every == 0 else \
            min((next_step // metrics_every + 1) * metrics_every, n_steps - 1)
        stop = min(stop, n_steps - 1)
        if checkpoint_every > 0:      # also stop right before every checkpoint boundary
            stop = min(stop, (next_step // checkpoint_every + 1) * checkpoint_every - 1)
        done, terminated, last_total, dts = _advance_to(engine, next_step, stop, L, defect_fraction, incremental=incremental,
                                                         thermal_mode=1 if thermal_updates else 0)
        for dt in dts:
            total_time += dt
        min_margin = min(min_margin, _advance_to.min_margin)
        if terminated:
            step = next_step + done
            print(f"Terminating at step {step}: no valid events (rate={last_total:.2e})")
            break
        step = stop
        next_step = stop + 1

        is_metric_step = (step % metrics_every == 0) or (step == n_steps - 1)
        if not is_metric_step:        # a pure checkpoint stop
            fields = engine.download()
            save_checkpoint(os.path.join(output_dir, "checkpoint.npz"), fields, engine.download(state=False, theta=False,
                            phi=False, T=False, defects=True)["defects"], next_step, total_time,
                            nuc_offset + engine.nucleation_count(), metrics_data, cet_detected)
            continue
        metrics_row(step, step % metrics_every == 0)
        if checkpoint_every > 0 and next_step % checkpoint_every == 0:
            save_checkpoint(os.path.join(output_dir, "checkpoint.npz"), engine.download(),
                            engine.download(state=False, theta=False, phi=False, T=False, defects=True)["defects"],
                            next_step, total_time, nuc_offset + engine.nucleation_count(), metrics_data, cet_detected)

    if mode == "B":
        # Super-step loop.  executed = events executed so far = index of the next event; g = super-step index (octant
        # g % 8, keys the counter uniforms together with run_seed).
        nbx = L // box if (box and L % box == 0) else 1
        d_max = 1 if box == L else nbx ** 3                       # events per super-step at most
        executed, g, thermal_done = 0, 0, 0
        mb_cfg = dict(mode="B", box=int(box), seed=run_seed, null_events=bool(null_events), thermal_cadence=thermal_cadence,
                      thermal_updates=bool(thermal_updates))
        if ckpt:
            ex = ckpt["extra"]
            if {k: ex.get(k) for k in mb_cfg} != mb_cfg:
                engine.close()
                raise ValueError(f"checkpoint was written by a run with {ex}, this call asks for {mb_cfg}")
            executed, g, thermal_done = next_step, int(ex["superstep"]), int(ex["thermal_done"])
        by_events = thermal_cadence == "events" and thermal_updates

        def checkpoint_b():
            save_checkpoint(os.path.join(output_dir, "checkpoint.npz"), engine.download(),
                            engine.download(state=False, theta=False, phi=False, T=False, defects=True)["defects"],
                            executed, total_time, nuc_offset + engine.nucleation_count(), metrics_data, cet_detected,
                            extra=dict(mb_cfg, superstep=g, thermal_done=thermal_done))
        while executed < n_steps:
            # super-steps until (at the earliest) the next metrics / checkpoint boundary or the end: a super-step executes
            # <= d_max events (a row is due once executed - 1 reaches the next multiple of metrics_every)
            boundary = min(((executed - 1) // metrics_every + 1) * metrics_every + 1, n_steps)
            if checkpoint_every > 0:
                boundary = min(boundary, (executed // checkpoint_every + 1) * checkpoint_every)
            nb = max(1, (boundary - executed) // d_max)
            if by_events:
                # kmc_simulation.py:248-250: event index e is preceded by e // 20 + 1 temperature updates; the super-step's
                # events all see the field of its first event (the lattice and T are frozen within a super-step)
                nb = 1
                due = executed // THERMAL_EVERY + 1
                for _ in range(due - thermal_done):
                    engine.thermal_cet(THERMAL_DT, scrub_nan=True)
                thermal_done = due
            r = engine.run_supersteps(g, nb, box, defect_fraction, run_seed, thermal_mode=0 if (by_events or not thermal_updates) else 1,
                                      thermal_dt=THERMAL_DT, null_events=null_events)
            before = executed
            for s in range(r["done"]):
                executed += int(r["n_exec"][s])
                total_time += int(r["n_exec"][s]) * float(r["dt_event"][s])
            g += r["done"]
            step = executed - 1
            if r["status"] == 1:
                print(f"Terminating at step {executed}: no valid events (rate={float(r['totals'][r['done']]):.2e})")
                break
            crossed = (executed - 1) // metrics_every > (before - 1) // metrics_every
            if crossed or executed >= n_steps:
                metrics_row(step, crossed)
            if checkpoint_every > 0 and executed // checkpoint_every > before // checkpoint_every:
                checkpoint_b()          # after the row of this super-step: a resumed run continues with the next super-step
        next_step = executed

    if metrics_data:
        df = pd.DataFrame(metrics_data)
        output_path = os.path.join(output_dir, "metrics.csv")
        df.to_csv(output_path, index=False)
        # plot_cet.py globs outputs/impurity_c_*/metrics_*.csv (plot_cet.py:26): also emit that name
        tag = output_prefix.split("_")[-1]
        df.to_csv(os.path.join(output_dir, f"metrics_{tag}.csv"), index=False)
        print(f"Metrics saved to {output_path}")

    fields = engine.download()
    state, theta, phi = fields["state"], fields["theta"], fields["phi"]
    atom_type = state.copy()
    engine.close()
    last_run_info.clear()
    last_run_info.update(mode=mode, min_margin=min_margin if mode == "A" else None, executed_events=step + 1)
    if mode == "A" and min_margin < MARGIN_WARN:
        print(f"Note: smallest selection margin {min_margin:.2e} < {MARGIN_WARN:.0e} of the total rate -- a pick this close to an "
              "event boundary may differ from the reference's sequential scan")
    print(f"Completed {step + 1} steps in {total_time:.2e} s")
    return state, atom_type, total_time, theta, phi
