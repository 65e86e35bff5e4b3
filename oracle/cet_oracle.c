/*
 * cet_oracle.c -- CPU restatement of the reference's KMC hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product: only
 * tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this
 * library, and only as the checker / reported CPU baseline.  The product path
 * (libcetkmc_hip.so) never links or calls it.
 *
 * Parity status: PINNED.  Every function below is checked in tests/test_oracle_*.py
 * against fixtures under tests/golden/ that were produced by running the reference
 * itself (tests/golden/make_golden.py; reference in identity-jit mode because numba
 * is not installed in the image -- see DESIGN.md "Oracle").
 *
 * Each function cites the reference file:line it restates (paths relative to the
 * reference repository root).  All arithmetic is IEEE-754 binary64 with no FMA
 * contraction (build with -ffp-contract=off) and keeps the reference's evaluation
 * order, so that +,-,*,/,max,min are bit-identical to NumPy/CPython; exp/sin/cos/acos
 * come from libm and may differ from NumPy's in the last ulp.
 *
 * Lattice layout: C-contiguous (L,L,L), axis 0 (i) slowest -- same as the reference.
 * state/defects are passed as int8 (values are 0..4 / 0..1 in the reference).
 */
#include <math.h>
#ifdef _OPENMP
#include <omp.h>
#endif
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef struct {
    double nu;            /* constants.py:70  NU */
    double nu_dep;        /* constants.py:71  NU_DEP */
    double E_b[3];        /* constants.py:74,80,84  W, Re, C */
    double E_diff[3];     /* constants.py:75,81,85 */
    double kT;            /* constants.py:63  K_T */
    double T_melt;        /* constants.py:64 */
    double I0;            /* constants.py:129 */
    double delta_T_c;     /* constants.py:123 */
    double K_nuc;         /* constants.py:130 */
    double beta_imp_nuc;  /* constants.py:131 */
    double max_imp_frac;  /* constants.py:90 */
    double rate_threshold;/* constants.py:145 */
    double anisotropy;    /* constants.py:102 */
    double impurity_re;   /* constants.py:82 */
    double impurity_c;    /* run_kmc / get_event_rates argument */
    /* thermal (thermal_solver.py:6-9, constants.py:54-55,65) */
    double alpha;         /* K/(RHO*CP) */
    double inv_dx2;       /* 1/(VOXEL_SIZE*VOXEL_SIZE) */
    double T_clip_lo;     /* T_SUB */
    double T_clip_hi;     /* T_MELT*1.1 */
    double T_nan;         /* nan_to_num(nan=T_SUB), kmc_simulation.py:249 */
    double rho_cp;        /* RHO*CP */
    double latent_coef;   /* 200e3/CP, thermal_solver.py:102 */
} orc_params;

typedef struct {
    int32_t type;      /* 0 dep, 1 diff, 2 nuc, 3 att */
    int32_t pos[3];
    int32_t target[3]; /* -1,-1,-1 when none */
    int32_t atom;      /* species written by the event */
    double  rate;
    int64_t dep_rank;  /* index among deposition candidates (type 0), else -1 */
} orc_event;

enum { EV_DEP = 0, EV_DIFF = 1, EV_NUC = 2, EV_ATT = 3 };
enum { CAT_DEP = 0, CAT_DIFF = 1, CAT_EMPTY = 2 };

/* kmc_event_rates.py:29-30,35 -- offsets in the reference's order */
static const int NB[14][3] = {
    {1, 1, 0}, {1, -1, 0}, {-1, 1, 0}, {-1, -1, 0},
    {0, 1, 1}, {0, 1, -1}, {0, -1, 1}, {0, -1, -1},
    {2, 0, 0}, {-2, 0, 0}, {0, 2, 0}, {0, -2, 0}, {0, 0, 2}, {0, 0, -2}};

#define IDX(L, i, j, k) (((int64_t)(i) * (L) + (j)) * (L) + (k))

/* Python's max(a, b) for two floats: "b if b > a else a" (keeps a when a is NaN). */
static inline double pymax(double a, double b) { return (b > a) ? b : a; }
/* Python's min(a, b): "b if b < a else a". */
static inline double pymin(double a, double b) { return (b < a) ? b : a; }

/* kmc_event_rates.py:25-40 get_bcc_neighbors: in-bounds survivors, order preserved. */
int orc_neighbors(int i, int j, int k, int L, int32_t out[14][3])
{
    int n = 0;
    for (int m = 0; m < 14; ++m) {
        int ni = i + NB[m][0], nj = j + NB[m][1], nk = k + NB[m][2];
        if (0 <= ni && ni < L && 0 <= nj && nj < L && 0 <= nk && nk < L) {
            out[n][0] = ni; out[n][1] = nj; out[n][2] = nk;
            ++n;
        }
    }
    return n;
}

/* kmc_event_rates.py:9-23 compute_misorientation */
double orc_misorientation(double t1, double p1, double t2, double p2)
{
    double v1x = sin(t1) * cos(p1), v1y = sin(t1) * sin(p1), v1z = cos(t1);
    double v2x = sin(t2) * cos(p2), v2y = sin(t2) * sin(p2), v2z = cos(t2);
    double dot = v1x * v2x + v1y * v2y + v1z * v2z;   /* left-to-right, :21 */
    dot = pymax(pymin(dot, 1.0), -1.0);               /* :22 */
    return acos(dot);
}

/* ---- per-voxel slot evaluation ------------------------------------------------
 * A voxel contributes to up to three "categories" in the reference's event order
 * (kmc_event_rates.py:47-160): DEP (only plane i==L-1), DIFF (occupied, not 4),
 * EMPTY (nuc then att...).  voxel_slots() evaluates one category of one voxel and
 * returns its valid events in reference order.
 */
typedef struct {
    int n;
    int32_t type[15];
    int32_t target[15][3];
    int32_t atom[15];
    double rate[15];
} slots_t;

static void voxel_slots(const orc_params *P, int L, const int8_t *state, const double *theta,
                        const double *phi, const double *T, const int8_t *defects,
                        int i, int j, int k, int cat, slots_t *S)
{
    S->n = 0;
    const int64_t c = IDX(L, i, j, k);
    const int st = state[c];
    if (cat == CAT_DEP) {
        /* kmc_event_rates.py:55-72 (species drawn by the caller) */
        if (i != L - 1 || st != 0) return;
        double local_T = pymax(T[c], 1.0);
        double thermal = exp(-(P->T_melt - local_T) / (P->kT * local_T));
        double rate = P->nu_dep * thermal;
        if (!isfinite(rate)) return;             /* :63 -- no RATE_THRESHOLD test here */
        S->type[0] = EV_DEP; S->rate[0] = rate; S->atom[0] = 0;
        S->target[0][0] = S->target[0][1] = S->target[0][2] = -1;
        S->n = 1;
        return;
    }
    int32_t nb[14][3];
    if (cat == CAT_DIFF) {
        /* kmc_event_rates.py:75-109 */
        if (st == 0 || st == 4) return;
        int ia = (st == 1) ? 0 : (st == 2) ? 1 : 2;      /* :83-91, "else" -> C */
        double E_b_atom = P->E_b[ia], E_diff_atom = P->E_diff[ia];
        double local_T = pymax(T[c], 1.0);
        double defect_factor = 1.0 + (double)defects[c];
        int nn = orc_neighbors(i, j, k, L, nb);
        int n_bonds = 0;
        for (int m = 0; m < nn; ++m)
            if (state[IDX(L, nb[m][0], nb[m][1], nb[m][2])] != 0) ++n_bonds;
        for (int m = 0; m < nn; ++m) {
            int64_t q = IDX(L, nb[m][0], nb[m][1], nb[m][2]);
            if (state[q] != 0) continue;
            double neighbor_T = pymax(T[q], 1.0);
            double dT = fabs(local_T - neighbor_T);
            double denom = pymax(P->T_melt - neighbor_T, 1.0);
            double grad_factor = 1.0 + 0.1 * dT / denom;
            double E_tot = pymax(E_diff_atom + 0.1 * (double)n_bonds * E_b_atom, 0.0);
            double rate = P->nu * grad_factor * exp(-defect_factor * E_tot / (P->kT * local_T));
            if (rate > P->rate_threshold && isfinite(rate)) {
                int s = S->n++;
                S->type[s] = EV_DIFF; S->rate[s] = rate; S->atom[s] = st;
                S->target[s][0] = nb[m][0]; S->target[s][1] = nb[m][1]; S->target[s][2] = nb[m][2];
            }
        }
        return;
    }
    /* CAT_EMPTY: kmc_event_rates.py:112-158 */
    if (st != 0) return;
    double local_T = pymax(T[c], 1.0);
    double dT = P->T_melt - local_T;
    int nn = orc_neighbors(i, j, k, L, nb);
    if (dT > P->delta_T_c) {
        int n_imp = 0;
        for (int m = 0; m < nn; ++m) {
            int s2 = state[IDX(L, nb[m][0], nb[m][1], nb[m][2])];
            if (s2 == 2 || s2 == 3) ++n_imp;
        }
        int den = nn > 1 ? nn : 1;
        double f_imp = pymin(P->max_imp_frac, (double)n_imp / (double)den);
        double K_eff = P->K_nuc * (1.0 - P->beta_imp_nuc * f_imp);
        K_eff = pymax(0.1 * P->K_nuc, pymin(P->K_nuc, K_eff));
        double barrier = K_eff / pymax((dT + 1e-6) * (dT + 1e-6), 1e-6);
        double rate = P->I0 * exp(-barrier / (P->kT * local_T));
        if (rate > P->rate_threshold && isfinite(rate)) {
            S->type[0] = EV_NUC; S->rate[0] = rate; S->atom[0] = 1;
            S->target[0][0] = S->target[0][1] = S->target[0][2] = -1;
            S->n = 1;
        }
    }
    for (int m = 0; m < nn; ++m) {
        int64_t q = IDX(L, nb[m][0], nb[m][1], nb[m][2]);
        int na = state[q];
        if (na == 0) continue;
        int ia;
        if (na == 1) ia = 0; else if (na == 2) ia = 1; else if (na == 3) ia = 2; else continue;
        double mis = orc_misorientation(theta[c], phi[c], theta[q], phi[q]);
        int km = k - 1 > 0 ? k - 1 : 0;
        int kp = k + 1 < L - 1 ? k + 1 : L - 1;
        double grad_z = (T[IDX(L, i, j, kp)] - T[IDX(L, i, j, km)]) * 0.5;
        double grad_factor = pymax(0.0, grad_z) / pymax(P->T_melt - local_T, 1.0);
        double E_att = 0.5 * P->E_b[ia] * (1.0 - cos(mis));
        double rate = P->nu * exp(-E_att / (P->kT * local_T)) * (1.0 + P->anisotropy * grad_factor);
        if (rate > P->rate_threshold && isfinite(rate)) {
            int s = S->n++;
            S->type[s] = EV_ATT; S->rate[s] = rate; S->atom[s] = na;
            S->target[s][0] = nb[m][0]; S->target[s][1] = nb[m][1]; S->target[s][2] = nb[m][2];
        }
    }
}

/* kmc_event_rates.py:66-71: deposition species from one uniform draw */
static inline int dep_species(const orc_params *P, double u)
{
    if (u < P->impurity_c) return 3;
    if (u < P->impurity_c + P->impurity_re) return 2;
    return 1;
}

/* kmc_event_rates.py:162-176 get_event_rates: the materialised, ordered event list.
 * u_dep: one uniform per deposition candidate in candidate order (the reference draws
 * np.random.random() per candidate, :65); may be NULL (species then reported as 0).
 * Returns the number of events (may exceed cap; only the first cap are stored). */
int64_t orc_enumerate(const orc_params *P, int L, const int8_t *state, const double *theta,
                      const double *phi, const double *T, const int8_t *defects,
                      const double *u_dep, int64_t n_u_dep, orc_event *out, int64_t cap,
                      int64_t *n_dep_out)
{
    int64_t n = 0, n_dep = 0;
    slots_t S;
    for (int i = 0; i < L; ++i)
        for (int cat = 0; cat < 3; ++cat)
            for (int j = 0; j < L; ++j)
                for (int k = 0; k < L; ++k) {
                    voxel_slots(P, L, state, theta, phi, T, defects, i, j, k, cat, &S);
                    for (int s = 0; s < S.n; ++s) {
                        if (n < cap) {
                            orc_event *e = &out[n];
                            e->type = S.type[s];
                            e->pos[0] = i; e->pos[1] = j; e->pos[2] = k;
                            memcpy(e->target, S.target[s], sizeof e->target);
                            e->rate = S.rate[s];
                            e->atom = S.atom[s];
                            e->dep_rank = -1;
                            if (S.type[s] == EV_DEP) {
                                e->dep_rank = n_dep;
                                e->atom = (u_dep && n_dep < n_u_dep) ? dep_species(P, u_dep[n_dep]) : 0;
                            }
                        }
                        if (S.type[s] == EV_DEP) ++n_dep;
                        ++n;
                    }
                }
    if (n_dep_out) *n_dep_out = n_dep;
    return n;
}

/* kmc_simulation.py:259-274: sequential total, r = u*total, first cumulative >= r,
 * fallback last.  Operates on a materialised list.  Returns chosen index or -1. */
int64_t orc_select_sequential(const orc_event *ev, int64_t n, double u, double *total_out)
{
    double total = 0.0;
    for (int64_t m = 0; m < n; ++m) total += ev[m].rate;
    if (total_out) *total_out = total;
    if (n == 0) return -1;
    double r = u * total, cum = 0.0;
    for (int64_t m = 0; m < n; ++m) {
        cum += ev[m].rate;
        if (cum >= r) return m;
    }
    return n - 1;
}

/* ---- canonical reduction tree (the build's parallel-friendly summation order) ----
 * The reference sums the event list left to right (kmc_simulation.py:259,269); a
 * parallel machine cannot reproduce that rounding, so the build fixes ONE summation
 * shape, used identically here and on the GPU (DESIGN.md "Canonical tree"):
 *   voxel-category sum : sequential over the voxel's valid slots (reference order)
 *   row sum    (c,i,j) : balanced binary tree over k in [0,P), zeros beyond L
 *   block sum  (c,i)   : balanced binary tree over j in [0,P)
 *   total              : balanced binary tree over blocks b = 3*i + c in [0,PB)
 * with P = next_pow2(L), PB = next_pow2(3L).  Event ORDER is unchanged.
 */
static int next_pow2(int n) { int p = 1; while (p < n) p <<= 1; return p; }

static double tree_sum(const double *x, int64_t lo, int64_t n, int64_t valid_hi)
{
    /* balanced tree over [lo, lo+n), entries at index >= valid_hi are 0 */
    if (lo >= valid_hi) return 0.0;
    if (n == 1) return x[lo];
    return tree_sum(x, lo, n / 2, valid_hi) + tree_sum(x, lo + n / 2, n / 2, valid_hi);
}
static int64_t range_count(const int32_t *c, int64_t lo, int64_t n, int64_t valid_hi)
{
    int64_t s = 0;
    for (int64_t m = lo; m < lo + n && m < valid_hi; ++m) s += c[m];
    return s;
}

/* Row sums / counts for planes [i0,i1): rowsum[(i*3+c)*L + j], rowcnt likewise
 * (arrays are full size 3*L*L; only planes in range are written). */
void orc_row_sums(const orc_params *P, int L, const int8_t *state, const double *theta,
                  const double *phi, const double *T, const int8_t *defects,
                  int i0, int i1, double *rowsum, int32_t *rowcnt)
{
    const int Pk = next_pow2(L);
    /* planes are independent: threads (orc_set_threads) only change who computes a row, not its value */
#pragma omp parallel
    {
        double *vs = (double *)malloc(sizeof(double) * (size_t)Pk);
        slots_t S;
#pragma omp for schedule(dynamic, 1)
        for (int i = i0; i < i1; ++i)
            for (int c = 0; c < 3; ++c)
                for (int j = 0; j < L; ++j) {
                    int32_t cnt = 0;
                    for (int k = 0; k < L; ++k) {
                        voxel_slots(P, L, state, theta, phi, T, defects, i, j, k, c, &S);
                        double s = 0.0;
                        for (int m = 0; m < S.n; ++m) s += S.rate[m];
                        vs[k] = s;
                        cnt += S.n;
                    }
                    rowsum[((int64_t)i * 3 + c) * L + j] = tree_sum(vs, 0, Pk, L);
                    rowcnt[((int64_t)i * 3 + c) * L + j] = cnt;
                }
        free(vs);
    }
}

/* worker threads of the row / thermal loops (1 = the scalar port); returns the count in effect */
int orc_set_threads(int n)
{
#ifdef _OPENMP
    if (n > 0) omp_set_num_threads(n);
    return omp_get_max_threads();
#else
    (void)n;
    return 1;
#endif
}

/* Block sums from row sums for planes [i0,i1): blocksum[i*3+c], blockcnt. */
void orc_block_sums(int L, const double *rowsum, const int32_t *rowcnt, int i0, int i1,
                    double *blocksum, int64_t *blockcnt)
{
    const int Pj = next_pow2(L);
    for (int b = 3 * i0; b < 3 * i1; ++b) {
        blocksum[b] = tree_sum(rowsum + (int64_t)b * L, 0, Pj, L);
        blockcnt[b] = range_count(rowcnt + (int64_t)b * L, 0, L, L);
    }
}

double orc_total(int L, const double *blocksum, const int64_t *blockcnt, int64_t *n_events,
                 int64_t *n_dep)
{
    const int NBk = 3 * L, PB = next_pow2(NBk);
    int64_t n = 0;
    for (int b = 0; b < NBk; ++b) n += blockcnt[b];
    if (n_events) *n_events = n;
    if (n_dep) *n_dep = blockcnt[3 * (L - 1) + CAT_DEP];
    return tree_sum(blocksum, 0, PB, NBk);
}

/* Generic descent: at node [lo,lo+n) go left iff the right half holds no events, or
 * the left half holds events and base+sum(left) >= r; otherwise base += sum(left). */
static int64_t descend(const double *sum, const int64_t *cnt64, const int32_t *cnt32,
                       int64_t P2, int64_t valid, double *base, double r)
{
    int64_t lo = 0, n = P2;
    while (n > 1) {
        int64_t h = n / 2;
        double sl = tree_sum(sum, lo, h, valid);
        int64_t cl = 0, cr = 0;
        for (int64_t m = lo; m < lo + h && m < valid; ++m) cl += cnt64 ? cnt64[m] : cnt32[m];
        for (int64_t m = lo + h; m < lo + n && m < valid; ++m) cr += cnt64 ? cnt64[m] : cnt32[m];
        if (cr == 0 || (cl > 0 && *base + sl >= r)) {
            n = h;
        } else {
            *base += sl;
            lo += h;
            n = h;
        }
    }
    return lo;
}

/* Canonical selection.  Needs all block sums/counts (global) and the row sums of the
 * chosen plane (local).  Returns 0 and fills *ev; returns 1 if there are no events. */
int orc_select_tree(const orc_params *P, int L, const int8_t *state, const double *theta,
                    const double *phi, const double *T, const int8_t *defects,
                    const double *blocksum, const int64_t *blockcnt, const double *rowsum,
                    const int32_t *rowcnt, double r, orc_event *ev)
{
    const int NBk = 3 * L, PB = next_pow2(NBk), Pj = next_pow2(L), Pk = next_pow2(L);
    int64_t n = 0;
    for (int b = 0; b < NBk; ++b) n += blockcnt[b];
    if (n == 0) return 1;
    double base = 0.0;
    int64_t b = descend(blocksum, blockcnt, NULL, PB, NBk, &base, r);
    int i = (int)(b / 3), c = (int)(b % 3);
    int64_t j = descend(rowsum + b * L, NULL, rowcnt + b * L, Pj, L, &base, r);
    /* recompute the row's voxel sums */
    double *vs = (double *)calloc((size_t)Pk, sizeof(double));
    int32_t *vc = (int32_t *)calloc((size_t)Pk, sizeof(int32_t));
    slots_t S;
    for (int k = 0; k < L; ++k) {
        voxel_slots(P, L, state, theta, phi, T, defects, i, (int)j, k, c, &S);
        double s = 0.0;
        for (int m = 0; m < S.n; ++m) s += S.rate[m];
        vs[k] = s; vc[k] = S.n;
    }
    int64_t k = descend(vs, NULL, vc, Pk, L, &base, r);
    voxel_slots(P, L, state, theta, phi, T, defects, i, (int)j, (int)k, c, &S);
    int pick = S.n - 1;
    double cum = base;
    for (int m = 0; m < S.n; ++m) {
        cum += S.rate[m];
        if (cum >= r) { pick = m; break; }
    }
    ev->type = S.type[pick];
    ev->pos[0] = i; ev->pos[1] = (int)j; ev->pos[2] = (int)k;
    memcpy(ev->target, S.target[pick], sizeof ev->target);
    ev->rate = S.rate[pick];
    ev->atom = S.atom[pick];
    ev->dep_rank = -1;
    if (ev->type == EV_DEP) {
        int64_t rank = 0;
        for (int jj = 0; jj < j; ++jj) rank += rowcnt[b * L + jj];
        for (int kk = 0; kk < k; ++kk) rank += vc[kk];
        ev->dep_rank = rank;
    }
    free(vs); free(vc);
    return 0;
}

/* kmc_simulation.py:276-327: apply one event (+ optional defect injection).
 * theta_new/phi_new: the two np.random.uniform draws (used by dep/nuc only).
 * Returns 1 if the event was a nucleation (nucleation_count += 1), else 0. */
int orc_apply(int L, int8_t *state, double *theta, double *phi, const orc_event *ev,
              double theta_new, double phi_new, int make_defect)
{
    int i = ev->pos[0], j = ev->pos[1], k = ev->pos[2];
    int64_t c = IDX(L, i, j, k);
    int nuc = 0;
    if (ev->type == EV_DEP || ev->type == EV_NUC) {
        state[c] = (int8_t)ev->atom; theta[c] = theta_new; phi[c] = phi_new;
        nuc = (ev->type == EV_NUC);
    } else if (ev->type == EV_DIFF && ev->target[0] != -1) {
        int64_t t = IDX(L, ev->target[0], ev->target[1], ev->target[2]);
        state[t] = state[c]; theta[t] = theta[c]; phi[t] = phi[c];
        state[c] = 0; theta[c] = 0.0; phi[c] = 0.0;
        c = t;                                      /* :303 updated site = target */
    } else if (ev->type == EV_ATT && ev->target[0] != -1) {
        int64_t t = IDX(L, ev->target[0], ev->target[1], ev->target[2]);
        state[c] = (int8_t)ev->atom; theta[c] = theta[t]; phi[c] = phi[t];
    }
    if (make_defect) { state[c] = 4; theta[c] = 0.0; phi[c] = 0.0; }   /* :323-327 */
    return nuc;
}

/* ---- thermal ------------------------------------------------------------------ */
static inline double scrub(const orc_params *P, double x)
{   /* np.nan_to_num(T, nan=T_SUB), kmc_simulation.py:249 */
    if (isnan(x)) return P->T_nan;
    if (isinf(x)) return x > 0 ? 1.7976931348623157e308 : -1.7976931348623157e308;
    return x;
}
static inline double clipT(const orc_params *P, double x)
{   /* np.clip(new_T, T_SUB, T_MELT*1.1), thermal_solver.py:105,117 */
    double v = x < P->T_clip_lo ? P->T_clip_lo : x;
    return v > P->T_clip_hi ? P->T_clip_hi : v;
}
/* scipy.ndimage.laplace(T) (mode='reflect'): per axis (-2*T) + (T[-1] + T[+1]) with
 * edge replication, accumulated axis0, +axis1, +axis2 (thermal_solver.py:91,115). */
static inline double lap_at(const double *T, int L, int i, int j, int k)
{
    int im = i > 0 ? i - 1 : 0, ip = i < L - 1 ? i + 1 : L - 1;
    int jm = j > 0 ? j - 1 : 0, jp = j < L - 1 ? j + 1 : L - 1;
    int km = k > 0 ? k - 1 : 0, kp = k < L - 1 ? k + 1 : L - 1;
    double c = T[IDX(L, i, j, k)];
    double d0 = c * -2.0 + (T[IDX(L, im, j, k)] + T[IDX(L, ip, j, k)]);
    double d1 = c * -2.0 + (T[IDX(L, i, jm, k)] + T[IDX(L, i, jp, k)]);
    double d2 = c * -2.0 + (T[IDX(L, i, j, km)] + T[IDX(L, i, j, kp)]);
    return (d0 + d1) + d2;
}

/* thermal_solver.py:107-117 update_temperature_cet; scrub_nan!=0 applies the run_kmc
 * pre-pass (kmc_simulation.py:249) to the input first. */
void orc_thermal_cet(const orc_params *P, int L, const double *Tin, double dt, int scrub_nan,
                     double *Tout)
{
    const int64_t n = (int64_t)L * L * L;
    double *Ts = (double *)malloc(sizeof(double) * (size_t)n);
    for (int64_t m = 0; m < n; ++m) Ts[m] = scrub_nan ? scrub(P, Tin[m]) : Tin[m];
    const double dta = dt * P->alpha;
#pragma omp parallel for schedule(static)
    for (int i = 0; i < L; ++i)
        for (int j = 0; j < L; ++j)
            for (int k = 0; k < L; ++k) {
                double lap = lap_at(Ts, L, i, j, k) * P->inv_dx2;
                Tout[IDX(L, i, j, k)] = clipT(P, Ts[IDX(L, i, j, k)] + dta * lap);
            }
    free(Ts);
}

/* thermal_solver.py:36-105 update_temperature.  q_top is the (L,L) volumetric source of
 * plane i=L-1, i.e. I_surface / VOXEL_SIZE (:86-95), built by the caller exactly as the
 * reference does (NumPy); prev/cur are lattice states for the latent-heat term (:98-99). */
void orc_thermal_laser(const orc_params *P, int L, const double *Tin, const int8_t *cur,
                       const int8_t *prev, double dt, const double *q_top, int scrub_nan,
                       double *Tout)
{
    const int64_t n = (int64_t)L * L * L;
    double *Ts = (double *)malloc(sizeof(double) * (size_t)n);
    for (int64_t m = 0; m < n; ++m) Ts[m] = scrub_nan ? scrub(P, Tin[m]) : Tin[m];
    const double dtm = dt > 1e-12 ? dt : 1e-12;           /* max(dt, 1e-12) :99 */
#pragma omp parallel for schedule(static)
    for (int i = 0; i < L; ++i)
        for (int j = 0; j < L; ++j)
            for (int k = 0; k < L; ++k) {
                int64_t c = IDX(L, i, j, k);
                double lap = lap_at(Ts, L, i, j, k) * P->inv_dx2;
                double q = (i == L - 1) ? q_top[(int64_t)j * L + k] : 0.0;
                double dF = ((prev[c] == 0 && cur[c] != 0) ? 1.0 : 0.0) / dtm;
                double dTdt = P->alpha * lap + q / P->rho_cp + P->latent_coef * dF;   /* :102 */
                Tout[c] = clipT(P, Ts[c] + dt * dTdt);
            }
    free(Ts);
}

/* ---- batched stepping loop (mirror of the device run_steps protocol) ------------
 * Executes up to n steps of kmc_simulation.py:246-332 with pre-drawn uniforms:
 *   u_pick[s]   : random.random() for r                       (:265)
 *   u_defect[s] : random.random() for the defect draw (:323), used iff defect_fraction>0
 *   u_np[...]   : NumPy global stream; rng_mode 0 consumes n_dep draws per step for the
 *                 deposition species (kmc_event_rates.py:65) then 2 for theta/phi of a
 *                 dep/nuc event (:283-284,308-309); rng_mode 1 takes the dep species from
 *                 a counter hash and consumes only the 2 orientation draws.
 * thermal_mode 0 none, 1 cet every 20 steps (:248-250), 2 laser every 20 steps with
 * q_planes[n_thermal_calls][L*L] consumed in order and prev_state kept by the caller.
 * Outputs per executed step: totals[s], events[s].  Returns steps executed; *status =
 * 0 ok, 1 terminated (no valid events, :260-262), 2 np stream exhausted (refill).
 */
static inline uint64_t mix64(uint64_t x)
{
    x ^= x >> 30; x *= 0xBF58476D1CE4E5B9ULL;
    x ^= x >> 27; x *= 0x94D049BB133111EBULL;
    x ^= x >> 31;
    return x;
}
double orc_counter_uniform(uint64_t seed, uint64_t step, uint64_t site)
{
    uint64_t x = mix64(seed + 0x9E3779B97F4A7C15ULL * (step + 1));
    x = mix64(x ^ (site * 0xD6E8FEB86659FD93ULL + 0x2545F4914F6CDD1DULL));
    return (double)(x >> 11) * (1.0 / 9007199254740992.0);
}

int64_t orc_run_steps(const orc_params *P, int L, int8_t *state, double *theta, double *phi,
                      double *T, const int8_t *defects, int8_t *prev_state,
                      int64_t step0, int64_t n, double defect_fraction,
                      const double *u_pick, const double *u_defect,
                      const double *u_np, int64_t np_cap, int64_t *np_used,
                      int rng_mode, uint64_t seed,
                      int thermal_mode, double thermal_dt, const double *q_planes,
                      int64_t *q_used,
                      double *totals, orc_event *events, int64_t *n_events_out,
                      int64_t *nuc_count, int *status)
{
    const int64_t nv = (int64_t)L * L * L;
    double *rowsum = (double *)malloc(sizeof(double) * 3 * (size_t)L * L);
    int32_t *rowcnt = (int32_t *)malloc(sizeof(int32_t) * 3 * (size_t)L * L);
    double *blocksum = (double *)malloc(sizeof(double) * 3 * (size_t)L);
    int64_t *blockcnt = (int64_t *)malloc(sizeof(int64_t) * 3 * (size_t)L);
    double *Tn = (double *)malloc(sizeof(double) * (size_t)nv);
    int64_t pos = 0, qpos = 0, s = 0;
    *status = 0;
    for (; s < n; ++s) {
        int64_t g = step0 + s;
        if (thermal_mode && g % 20 == 0) {
            if (thermal_mode == 1) {
                orc_thermal_cet(P, L, T, thermal_dt, 1, Tn);
            } else {
                orc_thermal_laser(P, L, T, state, prev_state, thermal_dt,
                                  q_planes + qpos * (int64_t)L * L, 1, Tn);
                memcpy(prev_state, state, (size_t)nv);
                ++qpos;
            }
            memcpy(T, Tn, sizeof(double) * (size_t)nv);
        }
        orc_row_sums(P, L, state, theta, phi, T, defects, 0, L, rowsum, rowcnt);
        orc_block_sums(L, rowsum, rowcnt, 0, L, blocksum, blockcnt);
        int64_t ne = 0, nd = 0;
        double total = orc_total(L, blocksum, blockcnt, &ne, &nd);
        /* get_event_rates has already drawn one species uniform per deposition candidate when run_kmc looks at the
         * total (kmc_event_rates.py:65 precedes kmc_simulation.py:259-262): also on the terminating step.  Parity
         * unpinned for that step: run_kmc cannot reach a termination with candidates left (T >= T_SUB after the first
         * update keeps every empty site's nucleation rate large), so no reference fixture covers it. */
        const int64_t dep_draws = (rng_mode == 0) ? nd : 0;
        if (pos + dep_draws > np_cap) { *status = 2; break; }
        if (ne == 0 || total < 1e-25 || !isfinite(total)) { *status = 1; totals[s] = total; pos += dep_draws; break; }
        if (pos + dep_draws + 2 > np_cap) { *status = 2; break; }
        orc_event ev;
        orc_select_tree(P, L, state, theta, phi, T, defects, blocksum, blockcnt, rowsum, rowcnt,
                        u_pick[s] * total, &ev);
        if (ev.type == EV_DEP) {
            double u = (rng_mode == 0) ? u_np[pos + ev.dep_rank]
                                       : orc_counter_uniform(seed, (uint64_t)g,
                                             (uint64_t)ev.pos[1] * (uint64_t)L + (uint64_t)ev.pos[2]);
            ev.atom = dep_species(P, u);
        }
        if (rng_mode == 0) pos += nd;
        double th = 0.0, ph = 0.0;
        if (ev.type == EV_DEP || ev.type == EV_NUC) {
            th = 0.0 + (3.141592653589793 - 0.0) * u_np[pos];          /* np.random.uniform(0, pi) */
            ph = 0.0 + (6.283185307179586 - 0.0) * u_np[pos + 1];      /* np.random.uniform(0, 2*pi) */
            pos += 2;
        }
        int mk = (defect_fraction > 0.0 && u_defect[s] < defect_fraction) ? 1 : 0;
        *nuc_count += orc_apply(L, state, theta, phi, &ev, th, ph, mk);
        totals[s] = total;
        if (events) events[s] = ev;
        if (n_events_out) n_events_out[s] = ne;
    }
    *np_used = pos;
    if (q_used) *q_used = qpos;
    free(rowsum); free(rowcnt); free(blocksum); free(blockcnt); free(Tn);
    return s;
}

/* ---- Mode B: synchronous super-steps over spatial domains ---------------------------
 * NOT in the reference (which executes one event per full-lattice sweep, kmc_simulation.py:246-332).
 * Definition (DESIGN.md "Mode B"); this function is its bit-exact comparator.
 *   - the lattice is tiled by nb^3 cubic boxes of edge `box` (even, >= 8, divides L); box d =
 *     (di*nb + dj)*nb + dk.  Super-step g activates one octant of every box: sector = g % 8,
 *     (si,sj,sk) = bits 2,1,0; window of box d = [di*box + si*H, +H) x ... with H = box/2.  Active
 *     windows of different boxes are >= H+1 >= 5 voxels apart on some axis, events write within +-2
 *     of their voxel, so the events of one super-step never touch the same voxel.
 *     box == L is the single-domain case: window = whole lattice, no sectors, identical to one
 *     Mode A step driven by the same uniforms.
 *   - all rates of a super-step are evaluated on the lattice as it was at the start (after the thermal
 *     update, same step%20 cadence as kmc_simulation.py:248-250); totals[g] is the Mode A total.
 *   - every box picks at most one event from its window: the window's events in the reference order
 *     (plane i; dep | diff | empty block; j; k; slot) summed by the canonical tree restricted to the
 *     window (k padded to pow2(H), j likewise, 3H blocks padded to a power of two), r = u_pick * R_d, same
 *     descent rule and slot scan as Mode A.  A box whose window holds no events (or R_d < 1e-25 / not
 *     finite) is idle (event type -1).
 *   - uniforms are counter based: u(seed, g, key): pick KEY 1<<40 | d, theta 2<<40 | d, phi 3<<40 | d,
 *     defect 4<<40 | d, deposition species j*L + k (as Mode A rng_mode 1).
 *   - null events (null_events != 0; Martinez et al., J. Comput. Phys. 227 (2008) 3804 "synchronous parallel KMC",
 *     combined with the octant sublattices of Shim & Amar, Phys. Rev. B 71 (2005) 125432): "one event per box" alone
 *     gives every non-idle box the same event frequency whatever its total rate R_d.  With R_max = the largest window
 *     total of the super-step, box d executes its pick only if u_accept * R_max < R_d (u_accept = u(seed, g,
 *     5<<40 | d)); otherwise the box performs a null event (logged as type -2, nothing applied).  An event e of an
 *     active window then fires with probability r_e / R_max per visit -- proportional to its rate everywhere, as in
 *     the reference's global pick (kmc_simulation.py:265-274).
 *   - all accepted events are applied (kmc_simulation.py:276-327 each), box order.
 *   - time advance: kmc_simulation.py:331-332 restated PER EXECUTED EVENT with the super-step's global total:
 *     dt_event[g] = max(-ln(max(1e-12, u(seed, g, 6<<40))) / totals[g], 1e-12); simulated time advances by
 *     n_exec[g] * dt_event[g].  (The reference advances time once per executed event by the same expression, and
 *     its 1e-12 floor always binds at the rates of this model, so both modes report n_executed * 1e-12; where the
 *     floor does not bind the mean is n_exec / total, the expected time for n_exec events of a process of rate
 *     `total` -- and, with null events, 1 / (8 R_max) per super-step, the uniformised step of one octant visit.)
 * events: [n][D] (type -1 idle box, -2 null event) or NULL; n_exec[n] = events applied per super-step. */
#define KEY_PICK   (1ULL << 40)
#define KEY_THETA  (2ULL << 40)
#define KEY_PHI    (3ULL << 40)
#define KEY_DEFECT (4ULL << 40)
#define KEY_ACCEPT (5ULL << 40)
#define KEY_DT     (6ULL << 40)

static int window_select(const orc_params *P, int L, const int8_t *state, const double *theta,
                         const double *phi, const double *T, const int8_t *defects,
                         int i0, int j0, int k0, int H, double u, double *leaf, int32_t *lcnt,
                         orc_event *ev, double *R_out)
{
    const int PH = next_pow2(H), PT = next_pow2(3 * H);
    const int64_t NL = (int64_t)PT * PH * PH;
    memset(leaf, 0, sizeof(double) * (size_t)NL);
    memset(lcnt, 0, sizeof(int32_t) * (size_t)NL);
    slots_t S;
    int64_t n = 0;
    for (int ii = 0; ii < H; ++ii)
        for (int c = 0; c < 3; ++c)
            for (int jj = 0; jj < H; ++jj)
                for (int kk = 0; kk < H; ++kk) {
                    voxel_slots(P, L, state, theta, phi, T, defects, i0 + ii, j0 + jj, k0 + kk, c, &S);
                    double s = 0.0;
                    for (int m = 0; m < S.n; ++m) s += S.rate[m];
                    const int64_t q = ((int64_t)(3 * ii + c) * PH + jj) * PH + kk;
                    leaf[q] = s; lcnt[q] = S.n;
                    n += S.n;
                }
    memset(ev, 0, sizeof *ev);
    ev->type = -1;                                   /* idle box: pos 0, target -1, rate 0 */
    ev->target[0] = ev->target[1] = ev->target[2] = -1;
    ev->dep_rank = -1;
    if (R_out) *R_out = 0.0;
    if (n == 0) return 1;
    const double R = tree_sum(leaf, 0, NL, NL);
    if (R < 1e-25 || !isfinite(R)) return 1;
    if (R_out) *R_out = R;
    const double r = u * R;
    double base = 0.0;
    const int64_t q = descend(leaf, NULL, lcnt, NL, NL, &base, r);
    const int kk = (int)(q % PH), jj = (int)((q / PH) % PH), b = (int)(q / ((int64_t)PH * PH));
    const int i = i0 + b / 3, c = b % 3, j = j0 + jj, k = k0 + kk;
    voxel_slots(P, L, state, theta, phi, T, defects, i, j, k, c, &S);
    int pick = S.n - 1;
    double cum = base;
    for (int m = 0; m < S.n; ++m) {
        cum += S.rate[m];
        if (cum >= r) { pick = m; break; }
    }
    ev->type = S.type[pick];
    ev->pos[0] = i; ev->pos[1] = j; ev->pos[2] = k;
    memcpy(ev->target, S.target[pick], sizeof ev->target);
    ev->rate = S.rate[pick];
    ev->atom = S.atom[pick];
    ev->dep_rank = -1;
    return 0;
}

/* one box's pick on the given lattice copy (window origin i0,j0,k0, edge H, uniform u): exported for the multi-rank
 * protocol rehearsal (tests/test_dist_gloo.py), where every rank picks for its own boxes on its slab + halo copy.
 * Returns 1 for an idle box (ev->type == -1). */
int orc_window_pick_r(const orc_params *P, int L, const int8_t *state, const double *theta, const double *phi,
                      const double *T, const int8_t *defects, int i0, int j0, int k0, int H, double u, orc_event *ev,
                      double *R_out /* window total (0 for an idle box); may be NULL */)
{
    const int PH = next_pow2(H), PT = next_pow2(3 * H);
    const int64_t NL = (int64_t)PT * PH * PH;
    double *leaf = (double *)malloc(sizeof(double) * (size_t)NL);
    int32_t *lcnt = (int32_t *)malloc(sizeof(int32_t) * (size_t)NL);
    const int idle = window_select(P, L, state, theta, phi, T, defects, i0, j0, k0, H, u, leaf, lcnt, ev, R_out);
    free(leaf); free(lcnt);
    return idle;
}
int orc_window_pick(const orc_params *P, int L, const int8_t *state, const double *theta, const double *phi,
                    const double *T, const int8_t *defects, int i0, int j0, int k0, int H, double u, orc_event *ev)
{
    return orc_window_pick_r(P, L, state, theta, phi, T, defects, i0, j0, k0, H, u, ev, NULL);
}

int64_t orc_run_supersteps(const orc_params *P, int L, int8_t *state, double *theta, double *phi,
                           double *T, const int8_t *defects, int8_t *prev_state,
                           int64_t step0, int64_t n, int box, double defect_fraction, uint64_t seed,
                           int thermal_mode, double thermal_dt, const double *q_planes, int64_t *q_used,
                           double *totals, orc_event *events, int64_t *n_exec,
                           int64_t *nuc_count, int *status, int null_events, double *dt_event)
{
    const int64_t nv = (int64_t)L * L * L;
    const int nb = L / box, H = (box == L) ? L : box / 2;
    const int64_t D = (int64_t)nb * nb * nb;
    const int PH = next_pow2(H), PT = next_pow2(3 * H);
    const int64_t NL = (int64_t)PT * PH * PH;
    double *rowsum = (double *)malloc(sizeof(double) * 3 * (size_t)L * L);
    int32_t *rowcnt = (int32_t *)malloc(sizeof(int32_t) * 3 * (size_t)L * L);
    double *blocksum = (double *)malloc(sizeof(double) * 3 * (size_t)L);
    int64_t *blockcnt = (int64_t *)malloc(sizeof(int64_t) * 3 * (size_t)L);
    double *Tn = (double *)malloc(sizeof(double) * (size_t)nv);
    double *leaf = (double *)malloc(sizeof(double) * (size_t)NL);
    int32_t *lcnt = (int32_t *)malloc(sizeof(int32_t) * (size_t)NL);
    orc_event *picked = (orc_event *)malloc(sizeof(orc_event) * (size_t)D);
    double *Rd = (double *)malloc(sizeof(double) * (size_t)D);
    int64_t qpos = 0, s = 0;
    *status = 0;
    for (; s < n; ++s) {
        const int64_t g = step0 + s;
        if (thermal_mode && g % 20 == 0) {
            if (thermal_mode == 1) {
                orc_thermal_cet(P, L, T, thermal_dt, 1, Tn);
            } else {
                orc_thermal_laser(P, L, T, state, prev_state, thermal_dt, q_planes + qpos * (int64_t)L * L, 1, Tn);
                memcpy(prev_state, state, (size_t)nv);
                ++qpos;
            }
            memcpy(T, Tn, sizeof(double) * (size_t)nv);
        }
        orc_row_sums(P, L, state, theta, phi, T, defects, 0, L, rowsum, rowcnt);
        orc_block_sums(L, rowsum, rowcnt, 0, L, blocksum, blockcnt);
        int64_t ne = 0, nd = 0;
        const double total = orc_total(L, blocksum, blockcnt, &ne, &nd);
        totals[s] = total;
        if (ne == 0 || total < 1e-25 || !isfinite(total)) { *status = 1; break; }
        if (dt_event) {     /* kmc_simulation.py:331-332 per executed event, one draw per super-step */
            const double u = orc_counter_uniform(seed, (uint64_t)g, KEY_DT);
            dt_event[s] = pymax(-log(pymax(1e-12, u)) / total, 1e-12);
        }
        const int sec = (int)(g % 8), si = (sec >> 2) & 1, sj = (sec >> 1) & 1, sk = sec & 1;
        /* select every box's event on the frozen lattice ... */
        for (int64_t d = 0; d < D; ++d) {
            const int di = (int)(d / ((int64_t)nb * nb)), dj = (int)((d / nb) % nb), dk = (int)(d % nb);
            const int i0 = (box == L) ? 0 : di * box + si * H;
            const int j0 = (box == L) ? 0 : dj * box + sj * H;
            const int k0 = (box == L) ? 0 : dk * box + sk * H;
            const double u = orc_counter_uniform(seed, (uint64_t)g, KEY_PICK | (uint64_t)d);
            window_select(P, L, state, theta, phi, T, defects, i0, j0, k0, H, u, leaf, lcnt, &picked[d], &Rd[d]);
        }
        double Rmax = 0.0;
        for (int64_t d = 0; d < D; ++d) if (picked[d].type >= 0 && Rd[d] > Rmax) Rmax = Rd[d];
        /* ... then apply them all (null events: only the accepted ones) */
        int64_t ex = 0;
        for (int64_t d = 0; d < D; ++d) {
            orc_event *ev = &picked[d];
            if (ev->type >= 0 && null_events) {
                const double ua = orc_counter_uniform(seed, (uint64_t)g, KEY_ACCEPT | (uint64_t)d);
                if (!(ua * Rmax < Rd[d])) ev->type = -2;      /* null event: the pick is logged, nothing is applied */
            }
            if (events) events[s * D + d] = *ev;
            if (ev->type < 0) continue;
            if (ev->type == EV_DEP) {
                const double u = orc_counter_uniform(seed, (uint64_t)g, (uint64_t)ev->pos[1] * (uint64_t)L + (uint64_t)ev->pos[2]);
                ev->atom = dep_species(P, u);
                if (events) events[s * D + d].atom = ev->atom;
            }
            double th = 0.0, ph = 0.0;
            if (ev->type == EV_DEP || ev->type == EV_NUC) {
                th = 0.0 + (3.141592653589793 - 0.0) * orc_counter_uniform(seed, (uint64_t)g, KEY_THETA | (uint64_t)d);
                ph = 0.0 + (6.283185307179586 - 0.0) * orc_counter_uniform(seed, (uint64_t)g, KEY_PHI | (uint64_t)d);
            }
            const int mk = (defect_fraction > 0.0 &&
                            orc_counter_uniform(seed, (uint64_t)g, KEY_DEFECT | (uint64_t)d) < defect_fraction) ? 1 : 0;
            *nuc_count += orc_apply(L, state, theta, phi, ev, th, ph, mk);
            ++ex;
        }
        if (n_exec) n_exec[s] = ex;
    }
    if (q_used) *q_used = qpos;
    free(rowsum); free(rowcnt); free(blocksum); free(blockcnt); free(Tn); free(leaf); free(lcnt); free(picked); free(Rd);
    return s;
}

int orc_sizeof_params(void) { return (int)sizeof(orc_params); }
int orc_sizeof_event(void) { return (int)sizeof(orc_event); }
