/*
 * cetkmc.h -- C ABI of the MI355X-native rejection-free KMC stepping engine
 *             (libcetkmc_hip.so; hand-written HIP kernels for gfx950).
 *
 * The reference (codebits1001/CET-driven-simulation-for-3D-printing-AM-KMC-Approach)
 * is pure Python and has no FFI layer: its boundary for this path is the plain Python
 * call surface of kmc_simulation.run_kmc / kmc_event_rates.get_event_rates /
 * thermal_solver.update_temperature*.  Each entry point below names the reference
 * code it replaces (file:line relative to the reference root).  The Python modules of
 * the same names in this repository bind these symbols with ctypes.
 *
 * Conventions
 *   - every function returns 0 on success, non-zero on failure; cetkmc_last_error()
 *     returns a thread-local message.  Nothing throws across the boundary.
 *   - the caller owns all host buffers; the library owns all device memory behind the
 *     opaque handle.  A handle is not thread-safe; use one handle per host thread.
 *   - host arrays are C-contiguous (L,L,L), axis 0 (i) slowest -- the reference layout.
 *   - there is NO CPU fallback: without a usable gfx950 device cetkmc_create fails.
 */
#ifndef CETKMC_H
#define CETKMC_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CETKMC_ABI_VERSION 1

/* Model constants (defaults = reference constants.py:34-148, thermal_solver.py:6-9).
 * Passed by the host so that derived values carry the host's IEEE rounding. */
typedef struct cetkmc_params {
    double nu;             /* constants.py:70  NU            */
    double nu_dep;         /* constants.py:71  NU_DEP        */
    double E_b[3];         /* constants.py:74,80,84  W,Re,C  */
    double E_diff[3];      /* constants.py:75,81,85          */
    double kT;             /* constants.py:63  K_T           */
    double T_melt;         /* constants.py:64                */
    double I0;             /* constants.py:129               */
    double delta_T_c;      /* constants.py:123               */
    double K_nuc;          /* constants.py:130               */
    double beta_imp_nuc;   /* constants.py:131               */
    double max_imp_frac;   /* constants.py:90                */
    double rate_threshold; /* constants.py:145               */
    double anisotropy;     /* constants.py:102               */
    double impurity_re;    /* constants.py:82                */
    double impurity_c;     /* run_kmc / get_event_rates argument */
    double alpha;          /* thermal_solver.py:9   K/(RHO*CP)   */
    double inv_dx2;        /* thermal_solver.py:79,114  1/dx^2   */
    double T_clip_lo;      /* thermal_solver.py:105,117  T_SUB   */
    double T_clip_hi;      /* thermal_solver.py:105,117  1.1*T_MELT */
    double T_nan;          /* kmc_simulation.py:249  nan_to_num(nan=T_SUB) */
    double rho_cp;         /* thermal_solver.py:102  RHO*CP      */
    double latent_coef;    /* thermal_solver.py:102  200e3/CP    */
} cetkmc_params;

/* One KMC event (an element of the list get_event_rates returns,
 * kmc_event_rates.py:72,109,132,158).  64 bytes. */
typedef struct cetkmc_event {
    int32_t type;       /* 0 'dep', 1 'diff', 2 'nuc', 3 'att'; -1 = none            */
    int32_t pos[3];     /* (i,j,k) of the event site                                   */
    int32_t target[3];  /* diff: destination, att: source neighbour, else (-1,-1,-1)   */
    int32_t atom;       /* species the event writes (dep: 0 until the species is drawn)*/
    double  rate;       /* [1/s]                                                       */
    int64_t dep_rank;   /* dep: index among the deposition candidates in row-major
                           (j,k) order (selects the reference's per-candidate species
                           draw, kmc_event_rates.py:65); else -1                       */
    double  theta;      /* orientation written to the updated site (set by apply)     */
    double  phi;
} cetkmc_event;

enum { CETKMC_DEP = 0, CETKMC_DIFF = 1, CETKMC_NUC = 2, CETKMC_ATT = 3 };

/* Result of one full-lattice rate sweep (kmc_simulation.py:253-259). */
typedef struct cetkmc_sweep_info {
    double  total;      /* canonical-tree sum of all event rates           */
    int64_t n_events;   /* len(events)                                                 */
    int64_t n_dep;      /* number of deposition candidates (= RNG draws of the sweep)  */
} cetkmc_sweep_info;

/* Arguments of the batched stepping loop (kmc_simulation.py:246-332, n iterations).
 * All pointers are HOST pointers; arrays are copied to the device once per call -- or ahead of the call by
 * cetkmc_stage_inputs(), after which cetkmc_run_steps() takes the SAME arguments with all four input pointers NULL. */
typedef struct cetkmc_run_args {
    int64_t step0;            /* global index of the first step (thermal cadence step%20) */
    int64_t n_steps;
    double  defect_fraction;  /* kmc_simulation.py:323                                  */
    const double* u_pick;     /* [n] random.random() for r            (:265)            */
    const double* u_defect;   /* [n] random.random() defect draws (:323) or NULL if defect_fraction==0 */
    const double* u_np;       /* NumPy global stream, consumed in order                 */
    int64_t np_cap;           /* number of doubles in u_np                              */
    int32_t rng_mode;         /* 0: reference stream (n_dep species draws per step + 2 orientation
                                    draws per dep/nuc event);
                                 1: counter-based species draw, stream holds only the orientation draws;
                                 2: every uniform counter based -- u(seed, step, key), the keys of Mode B's box 0 (pick
                                    1<<40, theta 2<<40, phi 3<<40, defect 4<<40, species j*L+k): u_pick / u_defect / u_np
                                    are not read and may be NULL.  This is what cetkmc_run_supersteps(box == L) runs */
    uint64_t seed;            /* rng_mode 1 key                                         */
    int32_t thermal_mode;     /* 0 none, 1 update_temperature_cet every 20 steps (:248-250),
                                 2 update_temperature (laser) every 20 steps             */
    double  thermal_dt;       /* dt of the thermal update (run_kmc uses 1e-6)           */
    const double* q_planes;   /* thermal_mode 2: [n_q][L*L] source planes, consumed in order */
    int64_t n_q;
    int32_t use_latent;       /* thermal_mode 2: latent-heat term on/off                */
    int32_t profile;          /* 1: time every rate-sweep launch with hipEvents; 3: every 8th launch, every 4th in batches of <= 64
                                 steps (an event pair costs ~4 us of stream time); 2: time every phase (cetkmc_get_counters) */
    int32_t incremental;      /* 0: every step evaluates the whole lattice (get_event_rates, kmc_event_rates.py:162);
                                 1: exact incremental mode -- between temperature updates only the rows whose
                                    rates the previous event can have changed are re-evaluated (identical
                                    results: the canonical summation tree is rebuilt from the same row sums) */
} cetkmc_run_args;

typedef struct cetkmc_run_result {
    int64_t steps_done;
    int32_t status;           /* 0 ok, 1 terminated (no valid events, :260-262), 2 u_np exhausted */
    int64_t np_used;          /* doubles of u_np consumed                               */
    int64_t q_used;
    int64_t nucleation_count; /* running total on the handle (:310)                     */
    double  sweep_ms_total;   /* profile: sum of rate-sweep kernel durations            */
    int64_t sweep_launches;
    double  wall_ms;          /* device time of the whole call (hipEvents on the stream) */
    int64_t full_sweeps;      /* steps that evaluated the whole lattice (= steps issued unless incremental) */
    double  min_margin;       /* selection margin of the batch: min over its steps of the distance of r = u * total from the
                                 nearer end of the chosen event's interval of the cumulative sum, divided by the total
                                 (this rank's picks; 1.0 if it made none).  The canonical tree sum and the reference's
                                 sequential sum (kmc_simulation.py:259,265-274) differ by <= ~1e-13 relative: a margin
                                 below that means the reference's scan could have stopped at the neighbouring event. */
} cetkmc_run_result;

/* Mode B -- synchronous super-steps (NOT in the reference; SURVEY.md section 8(f)4, DESIGN.md "Mode B").
 * The lattice is tiled by (L/box)^3 cubic boxes; every super-step runs the ordinary full rate sweep and then
 * every box executes at most one event picked from its active octant (octant = step % 8), so several thousand
 * events are executed per sweep.  box: even, 8..16, divides L -- or box == L: ONE domain without octants, which is
 * exactly a Mode A step (kmc_simulation.py:253-327) driven by the counter uniforms of box 0 (single process).
 * All uniforms are counter based (seed, step, key).
 * null_events: with R_max = the largest window total of the super-step, box d executes its pick only with probability
 * R_d / R_max (else a null event, logged as type -2): every event of an active window then fires with probability
 * rate / R_max per visit -- proportional to its rate everywhere, like the reference's global pick
 * (kmc_simulation.py:265-274).  0: every non-idle box executes its pick (boxes with few events are over-sampled).
 * The trajectory is not the reference's; its bit-exact comparator is the oracle's orc_run_supersteps. */
typedef struct cetkmc_super_args {
    int64_t step0;            /* global index of the first super-step (thermal cadence step%20, octant step%8) */
    int64_t n_steps;
    int32_t box;
    double  defect_fraction;
    uint64_t seed;
    int32_t thermal_mode;     /* as cetkmc_run_args */
    double  thermal_dt;
    const double* q_planes;
    int64_t n_q;
    int32_t use_latent;
    int32_t null_events;      /* see above */
} cetkmc_super_args;

/* Work issued, bytes moved and (while cetkmc_run_args.profile == 2) device time per phase, accumulated on the
 * handle since creation / the last reset.  Algorithmic bytes: 9 B per owned voxel per rate sweep (class u8 + rate
 * table f64; the recompute variant streams T f64 instead), 16 B per owned voxel per temperature update and 16 B per
 * owned voxel per rate-table refresh (T read, entry written). */
typedef struct cetkmc_counters {
    int64_t steps;              /* batched Mode A steps executed (cetkmc_run_steps)               */
    int64_t sweeps;             /* full rate sweeps launched (any entry point)                     */
    int64_t incremental_steps;  /* steps that re-evaluated dirty rows only                         */
    int64_t thermal_updates;
    int64_t supersteps;         /* Mode B super-steps executed                                     */
    int64_t bytes_h2d, bytes_d2h;               /* host <-> device bytes moved by this handle       */
    int64_t alg_bytes_sweep, alg_bytes_thermal; /* algorithmic bytes of the work issued             */
    int64_t profiled_steps;     /* steps the ms_* fields below cover                               */
    double  ms_thermal, ms_interface /* rate table + interface list kernels */, ms_sweep, ms_dirty_rows, ms_reduce, ms_select_apply;
    int64_t alg_bytes_table;    /* rate-table refreshes (k_rate_table)                             */
    int64_t table_updates;      /* rate-table refreshes launched                                   */
    int64_t interface_launches; /* full interface-list evaluations launched                        */
    double  ms_comm;            /* profile 2, multi-rank handles: device time between the hipEvent pairs around every
                                   collective of the profiled steps (block-sum / event all-gathers, temperature-halo and
                                   boundary-layer exchanges); part of ms_reduce / ms_select_apply / ms_thermal above   */
    int64_t comm_calls;         /* collectives those pairs bracketed                               */
} cetkmc_counters;

const char* cetkmc_last_error(void);
int cetkmc_abi_version(void);
/* hash of the sources the library was compiled from (-DCETKMC_SRC_HASH, set by cetkmc/_lib.build_library; "unknown" for a
 * hand build): the binding compares it with the sources beside it and rebuilds / refuses a stale library */
const char* cetkmc_source_hash(void);
/* sizeof of an ABI struct by name ("params", "event", "sweep_info", "run_args", "run_result", "super_args", "counters",
 * "host_comm"); -1 for an unknown name.  Lets a binding check its mirrors against the library it loaded. */
int cetkmc_struct_size(const char* name);
int cetkmc_device_count(int* n);

/* Lifetime.  n_slabs > 1 with all device_ids equal splits the lattice into axis-0 slabs
 * inside one process on one GPU (decomposition check mode).  Multi-GPU runs use one
 * process per GPU: cetkmc_create_rank + an RCCL unique id shared by the launcher. */
int cetkmc_create(const cetkmc_params* p, int L, int n_slabs, const int* device_ids, void** handle);
int cetkmc_get_unique_id(char out[128]);
int cetkmc_create_rank(const cetkmc_params* p, int L, int rank, int nranks, int device_id,
                       const char unique_id[128], void** handle);
/* Bring-up / test transport: the same per-rank device code path as cetkmc_create_rank, with the three collectives
 * relayed through host callbacks (e.g. torch.distributed over gloo) instead of RCCL; the call synchronises the
 * stream around every exchange, and several ranks may share one GPU.  Callbacks return 0 on success.
 *   allgather(user, send, recv, nbytes): every rank contributes nbytes at `send`; recv gets nranks*nbytes in rank order
 *   exchange (user, lo, hi, send_lo, recv_lo, send_hi, recv_hi, nbytes): swap nbytes with rank lo and with rank hi
 *                                                                       (-1: no such neighbour)                    */
typedef int (*cetkmc_allgather_fn)(void* user, const void* send, void* recv, int64_t nbytes);
typedef int (*cetkmc_exchange_fn)(void* user, int lo, int hi, const void* send_lo, void* recv_lo, const void* send_hi,
                                  void* recv_hi, int64_t nbytes);
typedef struct cetkmc_host_comm {
    cetkmc_allgather_fn allgather;
    cetkmc_exchange_fn exchange;
    void* user;
} cetkmc_host_comm;
int cetkmc_create_rank_host(const cetkmc_params* p, int L, int rank, int nranks, int device_id, const cetkmc_host_comm* hc,
                            void** handle);
int cetkmc_destroy(void* handle);
int cetkmc_set_params(void* handle, const cetkmc_params* p);
int cetkmc_sync(void* handle);
/* tuning / A-B switches: "sweep_variant" 0 = simple kernel, 1 = streaming kernel (LDS census) + per-voxel rate table (default),
 * 2 = streaming kernel that recomputes the nucleation rates in every sweep, 3 = census-free table sweep (class bytes + rate table
 * streamed once, no LDS; same bits, measured no faster -- DESIGN.md section 13), 4 = the same with one block per plane and the block
 * sums folded in the sweep launch (no k_plane_reduce; faster at L <= 128 only -- which is why handles on which this option was
 * never set use it for L <= 128 and variant 1 above); "interface_every_step" 1 = evaluate the
 * whole interface list before every full sweep instead of only after a temperature update; "thermal_lookahead" 1 = the next
 * temperature update of a batch and its rate table are computed ahead on a second stream (single process; same bits;
 * default 0: measured slower); "thermal_table" 1 (default) = the default temperature tiles also write
 * the new field's rate table and deposition rates (no k_rate_table launch after an update; same bits), 0 = separate launch;
 * "thermal_variant" 0 = one thread per voxel, 1 = plane marching (default:
 * k_thermal_tiles16 -- 16 x 256 tiles, 1024 threads with 2 rows each, 16 planes per block -- where the tiles cover the lattice
 * exactly, else k_thermal_march), 2 = k_thermal_march everywhere, 3 = 16-row tiles with 4 rows per thread, 4 = the 8-row
 * k_thermal_tiles (round-2 kernel), 5 = 16 x 128 tiles (A/B variants);
 * "thermal_planes_per_block" (march / 8-row tiles), "thermal_planes_per_block16" (16-row tiles); "reserve_batch" n = allocate the
 * device buffers and hipEvents of a batch of n steps now (a bench keeps hipMalloc / hipEventCreate out of its timed region) */
int cetkmc_set_option(void* handle, const char* key, int64_t value);
/* planes [*i0,*i1) owned by this handle (whole lattice unless created with create_rank) */
int cetkmc_owned_planes(void* handle, int* i0, int* i1);

/* Lattice transfer in the reference's dtypes (run_kmc's arrays, kmc_simulation.py:226-233).
 * NULL pointers leave the field unchanged / are not written. */
int cetkmc_upload(void* handle, const int64_t* state, const double* theta, const double* phi,
                  const double* T, const int64_t* defects);
int cetkmc_download(void* handle, int64_t* state, double* theta, double* phi, double* T,
                    int64_t* defects);
/* Narrow transfer of planes [i_begin,i_end) only (arrays shaped (i_end-i_begin, L, L)); the
 * range must cover the owned planes plus the 2-plane halo (clipped to the lattice). */
int cetkmc_upload_planes(void* handle, int i_begin, int i_end, const uint8_t* state,
                         const double* theta, const double* phi, const double* T,
                         const uint8_t* defects);
int cetkmc_download_planes(void* handle, int i_begin, int i_end, uint8_t* state, double* theta,
                           double* phi, double* T, uint8_t* defects);
/* defects.track_defects result (defects.py:4-19), full lattice, 0/1 bytes */
int cetkmc_set_defects(void* handle, const uint8_t* mask);
/* latent-heat reference state (thermal_solver.py:98); NULL = snapshot the current state */
int cetkmc_set_prev_state(void* handle, const int64_t* prev_state);

/* thermal_solver.update_temperature_cet (thermal_solver.py:107-117); scrub_nan applies
 * np.nan_to_num(T, nan=T_SUB) first (kmc_simulation.py:249). */
int cetkmc_thermal_cet(void* handle, double dt, int scrub_nan);
/* thermal_solver.update_temperature (thermal_solver.py:36-105).  q_top = (L,L) volumetric
 * source of plane i=L-1, i.e. I_surface/VOXEL_SIZE built by the host as the reference does
 * (:82-95); the latent term uses the state set by cetkmc_set_prev_state. */
int cetkmc_thermal_laser(void* handle, double dt, const double* q_top, int use_latent, int scrub_nan);

/* kmc_event_rates.get_event_rates as a reduction (kmc_event_rates.py:162-176 +
 * kmc_simulation.py:259): evaluates every event rate, leaves row/block sums on the device. */
int cetkmc_rate_sweep(void* handle, cetkmc_sweep_info* info);
/* kmc_simulation.py:265-274: pick the event whose cumulative rate first reaches r
 * (canonical-tree order).  Needs a preceding cetkmc_rate_sweep. */
int cetkmc_select(void* handle, double r, cetkmc_event* out);
/* kmc_simulation.py:276-327: apply one event; theta/phi are the two np.random.uniform draws
 * (dep/nuc); make_defect is the outcome of the defect-injection draw (:323). */
int cetkmc_apply(void* handle, const cetkmc_event* ev, double theta_new, double phi_new, int make_defect);
/* kmc_event_rates.get_event_rates in list form (parity / small L).  *n receives len(events);
 * at most cap are written, in the reference's order. */
int cetkmc_enumerate_events(void* handle, cetkmc_event* buf, int64_t cap, int64_t* n);
/* Row sums/counts of the last sweep, shape (L,3,L) = [i][category dep/diff/empty][j]. */
int cetkmc_row_sums(void* handle, double* rowsum, int32_t* rowcnt);

/* kmc_simulation.py:246-332 for n steps without host round trips. totals/events/n_events
 * (each [n_steps], may be NULL) receive the per-step total rate, chosen event and len(events). */
int cetkmc_run_steps(void* handle, const cetkmc_run_args* args, cetkmc_run_result* res,
                     double* totals, cetkmc_event* events, int64_t* n_events);
/* Puts the host inputs of ONE coming batch (u_pick, u_defect, u_np, q_planes of `args`) into the handle's device
 * buffers and waits for the copies.  The next cetkmc_run_steps() call may then pass the same args with all four input
 * pointers NULL: it checks step0 / n_steps / np_cap / n_q / thermal_mode against the staged batch and copies nothing
 * (bench.py: inputs resident in HBM before the timed region).  Any other stepping call drops the staged batch. */
int cetkmc_stage_inputs(void* handle, const cetkmc_run_args* args);

/* totals[n] (Mode A total of every super-step's sweep), events[n][D] or NULL (type -1: idle box, -2: null event),
 * n_executed[n] events applied per super-step, dt_event[n] (or NULL) the time increment PER EXECUTED EVENT of every
 * super-step: kmc_simulation.py:331-332 restated with the super-step's total and one counter uniform (key 6<<40),
 * dt_event[g] = max(-ln(max(1e-12, u_g)) / totals[g], 1e-12); simulated time advances by n_executed[g] * dt_event[g]
 * (n_executed summed over ranks; dt_event is the same on every rank).  res->np_used is 0.
 * Across ranks (cetkmc_create_rank*; needs (L / nranks) % box == 0): the boxes are sharded with the slabs -- D is the
 * number of THIS rank's boxes ((L / nranks / box) * (L / box)^2, global box order = rank order), events / n_executed /
 * res->nucleation_count cover this rank's boxes only (the caller sums them), totals are global.  Per super-step the
 * ranks exchange the block sums (all-gather, as Mode A) and the events of their boundary box layers (neighbour
 * send/recv, (L/box)^2 records each way): every rank applies its neighbours' boundary events to its own copy of the
 * halo planes and of the owned planes a diffusion target reached. */
int cetkmc_run_supersteps(void* handle, const cetkmc_super_args* a, cetkmc_run_result* res, double* totals,
                          cetkmc_event* events, int64_t* n_executed, double* dt_event);

int cetkmc_get_counters(void* handle, cetkmc_counters* out, int reset);

/* Grain clustering on the device (utils.get_clusters / dfs_cluster, utils.py:28-84): connected
 * components of occupied 14-stencil neighbours with misorientation < threshold, numbered 1.. in the
 * order of their first voxel (row-major), exactly like the reference.  Whole lattice in one slab only.
 * _stats: per cluster first voxel (i,j,k), voxel count, bounding box (imin,jmin,kmin,imax,jmax,kmax);
 * _labels: the (L,L,L) int32 label volume the reference returns as `visited`. */
int cetkmc_cluster(void* handle, double threshold, int64_t* n_clusters);
int cetkmc_cluster_stats(void* handle, int64_t cap, int32_t* first_voxel, int64_t* size, int32_t* bbox);
int cetkmc_cluster_labels(void* handle, int32_t* labels);

/* Sparse site queries so that the host-side defect model (defects.track_defects, defects.py:4-19) and the
 * species counts of the metrics row (kmc_simulation.py:355-357) need no full-lattice transfer:
 * counts[s] = owned voxels in state s (0..4; [5] = other); gather: global linear index ((i*L+j)*L+k) and T of
 * every owned voxel of the given species (unordered, *n = total found; at most cap are written);
 * set_defects_sparse: defects := 0, then 1 at the listed voxels. */
int cetkmc_species_counts(void* handle, int64_t counts[6]);
int cetkmc_gather_species(void* handle, int species, int64_t* lin_idx, double* T_vals, int64_t cap, int64_t* n);
int cetkmc_set_defects_sparse(void* handle, const int64_t* lin_idx, int64_t n);

int64_t cetkmc_nucleation_count(void* handle);
int cetkmc_reset_counters(void* handle);

/* Measurement helpers (bench.py): n back-to-back rate sweeps timed with hipEvents on the
 * engine's stream; returns total milliseconds. */
int cetkmc_time_sweeps(void* handle, int n, double* ms_total);
/* bench helper: average elapsed time between two hipEvents recorded back to back on the handle's stream, with one empty
 * kernel between them (n pairs) -- what a hipEvent-bracketed kernel duration includes besides the kernel's own work */
int cetkmc_event_overhead(void* handle, int n, double* ms_avg);
/* Transport self-test of a multi-rank handle (collective: every rank calls it with the same `bytes`): patterned buffers
 * through the all-gather and the neighbour exchange the stepping loops use, verified on the host; a single-rank RCCL
 * communicator sends to itself.  times_us (may be NULL) receives the average wall time in microseconds of 20 further
 * all-gathers [0] and neighbour exchanges [1] of that size, stream synchronisation included.  Single-process handle: no-op. */
int cetkmc_comm_selftest(void* handle, int64_t bytes, double* times_us);

#ifdef __cplusplus
}
#endif
#endif /* CETKMC_H */
