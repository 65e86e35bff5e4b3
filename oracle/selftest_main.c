/* selftest_main.c -- sanitizer self-test of the CPU oracle (TEST INFRASTRUCTURE, like cet_oracle.c itself).
 *
 * `make -C oracle asan` compiles cet_oracle.c together with this driver under -fsanitize=address,undefined (CPU only;
 * GPU sanitizers are not available on the pool) and runs it: a small lattice goes through every exported entry point --
 * enumeration, canonical sums, selection, the batched Mode A loop in all three stream modes, both thermal updates and
 * Mode B with and without null events, incl. the single-box case -- so that an out-of-bounds access or undefined
 * arithmetic in the restatement aborts the run.  Exit code 0 = clean.
 */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "cet_oracle.c"      /* one translation unit: the restatement has no header of its own */

static uint64_t rs = 0x9E3779B97F4A7C15ULL;
static double urand(void)
{
    rs ^= rs << 13; rs ^= rs >> 7; rs ^= rs << 17;
    return (double)(rs >> 11) * (1.0 / 9007199254740992.0);
}

int main(void)
{
    const int L = 16;
    const int64_t nv = (int64_t)L * L * L;
    orc_params P;
    memset(&P, 0, sizeof P);
    P.nu = 1e13; P.nu_dep = 2e13;
    P.E_b[0] = 3.8; P.E_b[1] = 4.2; P.E_b[2] = 3.2;
    P.E_diff[0] = 0.35; P.E_diff[1] = 0.50; P.E_diff[2] = 0.30;
    P.kT = 8.617333262e-5; P.T_melt = 3695.0; P.I0 = 5e13; P.delta_T_c = 10.0;
    P.K_nuc = 500.0; P.beta_imp_nuc = 0.4; P.max_imp_frac = 1.0; P.rate_threshold = 1e-30;
    P.anisotropy = 0.25; P.impurity_re = 0.10; P.impurity_c = 0.2;
    P.alpha = 173.0 / (19300.0 * 132.0); P.inv_dx2 = 1.0 / (5e-6 * 5e-6);
    P.T_clip_lo = 2800.0; P.T_clip_hi = 3695.0 * 1.1; P.T_nan = 2800.0; P.rho_cp = 19300.0 * 132.0; P.latent_coef = 200e3 / 132.0;

    int8_t *state = calloc((size_t)nv, 1), *defects = calloc((size_t)nv, 1), *prev = calloc((size_t)nv, 1);
    double *theta = calloc((size_t)nv, 8), *phi = calloc((size_t)nv, 8), *T = calloc((size_t)nv, 8);
    for (int64_t v = 0; v < nv; ++v) {
        const int k = (int)(v % L);
        T[v] = 2800.0 + (895.0 / L) * k;
        if (k < 4 || urand() < 0.05) {
            state[v] = (int8_t)(1 + (int)(urand() * 4.0));          /* 1..4 (4 = defect) */
            if (state[v] != 4) { theta[v] = urand() * 3.14159; phi[v] = urand() * 6.28318; }
            if (state[v] == 3 && urand() < 0.1) defects[v] = 1;
        }
    }
    T[5] = NAN; T[77] = 0.5; T[300] = 3690.0; T[301] = 4000.0;       /* scrub / T<1 / dT<10 / T>T_melt paths */
    memcpy(prev, state, (size_t)nv);

    /* enumeration + sequential selection (kmc_event_rates.py:162-176, kmc_simulation.py:259-274) */
    const int64_t cap = 16 * nv + 16;
    orc_event *ev = malloc(sizeof(orc_event) * (size_t)cap);
    double *u_dep = malloc(sizeof(double) * (size_t)L * L);
    for (int q = 0; q < L * L; ++q) u_dep[q] = urand();
    int64_t n_dep = 0;
    const int64_t n = orc_enumerate(&P, L, state, theta, phi, T, defects, u_dep, (int64_t)L * L, ev, cap, &n_dep);
    double total = 0.0;
    const int64_t pick = orc_select_sequential(ev, n, 0.37, &total);
    if (n <= 0 || pick < 0 || pick >= n || !(total > 0.0)) { fprintf(stderr, "enumerate/select failed\n"); return 1; }

    /* the batched loop in the three stream modes, both thermal modes */
    const int NS = 45;
    double *u_pick = malloc(8 * NS), *u_def = malloc(8 * NS), *u_np = malloc(8 * (size_t)NS * (L * L + 2));
    double *totals = malloc(8 * (NS + 1)), *q = calloc((size_t)3 * L * L, 8);
    orc_event *log = malloc(sizeof(orc_event) * NS);
    int64_t *nev = malloc(8 * NS);
    for (int s = 0; s < NS; ++s) { u_pick[s] = urand(); u_def[s] = urand(); }
    for (int s = 0; s < NS * (L * L + 2); ++s) u_np[s] = urand();
    for (int s = 0; s < 3 * L * L; ++s) q[s] = 1e12 * urand();
    for (int mode = 0; mode < 2; ++mode)
        for (int th = 1; th <= 2; ++th) {
            int64_t np_used = 0, q_used = 0, nuc = 0;
            int status = 0;
            const int64_t done = orc_run_steps(&P, L, state, theta, phi, T, defects, prev, 3, NS, 0.05, u_pick, u_def, u_np,
                                               (int64_t)NS * (L * L + 2), &np_used, mode, 42, th, 1e-6, q, &q_used, totals, log,
                                               nev, &nuc, &status);
            if (done != NS || status != 0) { fprintf(stderr, "run_steps mode %d thermal %d: done %lld status %d\n", mode, th, (long long)done, status); return 1; }
        }

    /* Mode B: boxes 8 and 16 (= L: single domain), with and without null events */
    const int boxes[2] = {8, 16};
    for (int b = 0; b < 2; ++b)
        for (int ne = 0; ne < 2; ++ne) {
            const int D = (L / boxes[b]) * (L / boxes[b]) * (L / boxes[b]);
            const int NB = 20;
            orc_event *evb = malloc(sizeof(orc_event) * (size_t)NB * D);
            int64_t *nex = malloc(8 * NB), q_used = 0, nuc = 0;
            double *tot = malloc(8 * (NB + 1)), *dte = malloc(8 * NB);
            int status = 0;
            const int64_t done = orc_run_supersteps(&P, L, state, theta, phi, T, defects, prev, 5, NB, boxes[b], 0.02, 7, 1, 1e-6,
                                                    NULL, &q_used, tot, evb, nex, &nuc, &status, ne, dte);
            if (done != NB || !(dte[0] >= 1e-12)) { fprintf(stderr, "run_supersteps box %d null %d failed\n", boxes[b], ne); return 1; }
            free(evb); free(nex); free(tot); free(dte);
        }
    printf("oracle sanitizer self-test ok: %lld events enumerated, total %.6e\n", (long long)n, total);
    free(state); free(defects); free(prev); free(theta); free(phi); free(T); free(ev); free(u_dep);
    free(u_pick); free(u_def); free(u_np); free(totals); free(q); free(log); free(nev);
    return 0;
}
