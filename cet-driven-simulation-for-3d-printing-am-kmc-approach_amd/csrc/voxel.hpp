// voxel.hpp -- per-voxel Arrhenius rate evaluation (device side), gfx950.
//
// Restates kmc_event_rates.py:42-160 (compute_row_events) of the reference for ONE voxel
// and ONE pass over its events, in the reference's slot order.  All arithmetic is IEEE
// binary64 in the reference's evaluation order; the translation unit is compiled with
// -ffp-contract=off so no multiply-add is fused.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace cetkmc {

constexpr int KOFF = 4;           // byte offset of k=0 inside a padded state row
constexpr uint8_t OOB = 255;      // sentinel state outside the lattice
constexpr int CAT_DEP = 0, CAT_DIFF = 1, CAT_EMPTY = 2;
constexpr int EV_DEP = 0, EV_DIFF = 1, EV_NUC = 2, EV_ATT = 3;

// kmc_event_rates.py:29-30,35 -- neighbour offsets in the reference's order.  constexpr
// lookups so that fully unrolled loops fold them into immediates.
__host__ __device__ constexpr int nbi(int m) { constexpr int t[14] = {1, 1, -1, -1, 0, 0, 0, 0, 2, -2, 0, 0, 0, 0}; return t[m]; }
__host__ __device__ constexpr int nbj(int m) { constexpr int t[14] = {1, -1, 1, -1, 1, 1, -1, -1, 0, 0, 2, -2, 0, 0}; return t[m]; }
__host__ __device__ constexpr int nbk(int m) { constexpr int t[14] = {0, 0, 0, 0, 1, -1, 1, -1, 0, 0, 0, 0, 2, -2}; return t[m]; }

// Kernel-side copy of the rate constants (cetkmc_params, include/cetkmc.h).
struct KParams {
    double nu, nu_dep;
    double E_b[3], E_diff[3];
    double kT, T_melt, I0, delta_T_c, rate_threshold, anisotropy;
    double impurity_re, impurity_c;
    double K_nuc, beta_imp_nuc, max_imp_frac;
};

// Geometry + field pointers of one axis-0 slab as the kernels see it.
// Local plane index li = i - (gi0 - 2): two halo planes on each side are always allocated
// (sentinel state when they fall outside the lattice).
struct SlabView {
    int L;        // lattice edge
    int gi0;      // first owned global plane
    int nloc;     // owned planes
    int RJ;       // padded rows per plane (>= L+4, j=-2 is row 0)
    int pitchS;   // bytes per padded state row (k=0 at KOFF)
    int pitchT;   // doubles per T/theta/phi row
    int Pk;       // next_pow2(L)
    uint8_t* state;     // [(nloc+4)][RJ][pitchS]
    uint8_t* defects;   // same layout
    double* T;          // [(nloc+4)][L][pitchT]  (current buffer)
    double* theta;
    double* phi;
    double* rowsum;     // [nloc*3][L]  row sums of the last sweep, index (lp*3+cat)*L + j
    int32_t* rowcnt;
    __device__ __forceinline__ int64_t sidx(int li, int j, int k) const {
        return ((int64_t)li * RJ + (j + 2)) * pitchS + KOFF + k;
    }
    __device__ __forceinline__ int64_t tidx(int li, int j, int k) const {
        return ((int64_t)li * L + j) * pitchT + k;
    }
};

// CPython's max(a,b)/min(a,b) on floats: "b if b > a else a" (keeps a when a is NaN)
__device__ __forceinline__ double pymax(double a, double b) { return (b > a) ? b : a; }
__device__ __forceinline__ double pymin(double a, double b) { return (b < a) ? b : a; }
__device__ __forceinline__ bool finite_d(double x) { return __builtin_isfinite(x); }

// kmc_event_rates.py:122-128: K_eff as a function of (#in-bounds neighbours, #Re/C neighbours).
// Tabulated once per launch (15x15) with the reference's expression order.
__device__ __forceinline__ double k_eff(const KParams& P, int n_nb, int n_imp)
{
    int den = n_nb > 1 ? n_nb : 1;
    double f_imp = pymin(P.max_imp_frac, (double)n_imp / (double)den);
    double K = P.K_nuc * (1.0 - P.beta_imp_nuc * f_imp);
    return pymax(0.1 * P.K_nuc, pymin(P.K_nuc, K));
}

// kmc_event_rates.py:9-23
__device__ __forceinline__ double misorientation(double t1, double p1, double t2, double p2)
{
    double s1 = sin(t1), c1 = cos(t1), s2 = sin(t2), c2 = cos(t2);
    double v1x = s1 * cos(p1), v1y = s1 * sin(p1), v1z = c1;
    double v2x = s2 * cos(p2), v2y = s2 * sin(p2), v2z = c2;
    double dot = v1x * v2x + v1y * v2y + v1z * v2z;
    dot = pymax(pymin(dot, 1.0), -1.0);
    return acos(dot);
}

// Evaluate the events of voxel (i,j,k) (local plane li) whose own state is `st` and raw
// temperature `Traw`.  `nb(m)` returns the state of neighbour slot m (OOB outside the
// lattice).  `ktab` is the 15x15 K_eff table.  For every valid event, in reference order,
// calls emit(category, type, rate, m, atom) with m the neighbour slot (or -1).
//   dep  : kmc_event_rates.py:55-72   (only plane i == L-1, st == 0)
//   diff : kmc_event_rates.py:75-109  (st != 0, st != 4)
//   nuc  : kmc_event_rates.py:116-132 (st == 0)
//   att  : kmc_event_rates.py:135-158 (st == 0)
template <class NB, class EMIT>
__device__ __forceinline__ void eval_voxel(const KParams& P, const SlabView& S, const double* ktab,
                                           int li, int i, int j, int k, int st, double Traw,
                                           NB nb, EMIT emit)
{
    if (st >= 128) return;   // sentinel / padding
    const double Tc = pymax(Traw, 1.0);
    if (st == 0) {
        if (i == S.L - 1) {
            double rate = P.nu_dep * exp(-(P.T_melt - Tc) / (P.kT * Tc));
            if (finite_d(rate)) emit(CAT_DEP, EV_DEP, rate, -1, 0);
        }
        int s[14];
        int n_nb = 0, n_imp = 0, n_src = 0;
#pragma unroll
        for (int m = 0; m < 14; ++m) {
            s[m] = nb(m);
            n_nb += (s[m] != OOB);
            n_imp += (s[m] == 2 || s[m] == 3);
            n_src += (s[m] >= 1 && s[m] <= 3);
        }
        const double dT = P.T_melt - Tc;
        const double kTT = P.kT * Tc;
        if (dT > P.delta_T_c) {
            double K = ktab[n_nb * 15 + n_imp];
            double a = dT + 1e-6;
            double barrier = K / pymax(a * a, 1e-6);
            double rate = P.I0 * exp(-barrier / kTT);
            if (rate > P.rate_threshold && finite_d(rate)) emit(CAT_EMPTY, EV_NUC, rate, -1, 1);
        }
        if (n_src > 0) {
            const int L = S.L;
            const int64_t c = S.tidx(li, j, k);
            const double th0 = S.theta[c], ph0 = S.phi[c];
            const int km = k - 1 > 0 ? k - 1 : 0;
            const int kp = k + 1 < L - 1 ? k + 1 : L - 1;
            const double grad_z = (S.T[S.tidx(li, j, kp)] - S.T[S.tidx(li, j, km)]) * 0.5;
            const double gf = pymax(0.0, grad_z) / pymax(dT, 1.0);
            const double aniso = 1.0 + P.anisotropy * gf;
#pragma unroll
            for (int m = 0; m < 14; ++m) {
                if (s[m] >= 1 && s[m] <= 3) {
                    int64_t q = S.tidx(li + nbi(m), j + nbj(m), k + nbk(m));
                    double mis = misorientation(th0, ph0, S.theta[q], S.phi[q]);
                    double E_att = 0.5 * P.E_b[s[m] - 1] * (1.0 - cos(mis));
                    double rate = P.nu * exp(-E_att / kTT) * aniso;
                    if (rate > P.rate_threshold && finite_d(rate)) emit(CAT_EMPTY, EV_ATT, rate, m, s[m]);
                }
            }
        }
    } else if (st != 4) {
        int s[14];
        int n_bonds = 0, n_empty = 0;
#pragma unroll
        for (int m = 0; m < 14; ++m) {
            s[m] = nb(m);
            n_bonds += (s[m] != 0 && s[m] != OOB);
            n_empty += (s[m] == 0);
        }
        if (n_empty > 0) {
            const int ia = (st == 1) ? 0 : (st == 2) ? 1 : 2;
            const double defect_factor = 1.0 + (double)S.defects[S.sidx(li, j, k)];
            const double E_tot = pymax(P.E_diff[ia] + 0.1 * (double)n_bonds * P.E_b[ia], 0.0);
            const double arr = exp(-defect_factor * E_tot / (P.kT * Tc));
#pragma unroll
            for (int m = 0; m < 14; ++m) {
                if (s[m] == 0) {
                    double Tn = pymax(S.T[S.tidx(li + nbi(m), j + nbj(m), k + nbk(m))], 1.0);
                    double dTn = fabs(Tc - Tn);
                    double denom = pymax(P.T_melt - Tn, 1.0);
                    double grad = 1.0 + 0.1 * dTn / denom;
                    double rate = P.nu * grad * arr;
                    if (rate > P.rate_threshold && finite_d(rate)) emit(CAT_DIFF, EV_DIFF, rate, m, st);
                }
            }
        }
    }
}

}  // namespace cetkmc
