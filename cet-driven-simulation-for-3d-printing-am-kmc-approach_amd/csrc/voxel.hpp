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
constexpr int KOFFC = 8;          // byte offset of k=0 inside a padded class row (8: a lane's 8 voxels are one aligned b64)

// Census class of a lattice state, one byte per voxel: bit 0 = empty, bit 1 = W/Re/C (an attachment source /
// a diffusing atom); defects (4) and everything outside the lattice are 0.  OR-ing the class bytes of the 14
// neighbours answers the two questions the rate sweep asks of a neighbourhood -- "any W/Re/C neighbour?" (the voxel
// owns attachment events: it is an interface voxel) and "any empty neighbour?" (an atom owns diffusion events) --
// without a compare; the neighbour COUNTS of the rate formulas (kmc_event_rates.py:95-101,121-125) are only needed at
// interface voxels, whose packed neighbourhood word (ifc_encode) carries them.
__host__ __device__ inline uint8_t class8(int st)
{
    if (st == 0) return 0x01;
    if (st >= 1 && st <= 3) return 0x02;
    return 0;
}
constexpr int CAT_DEP = 0, CAT_DIFF = 1, CAT_EMPTY = 2;
constexpr int EV_DEP = 0, EV_DIFF = 1, EV_NUC = 2, EV_ATT = 3;

// kmc_event_rates.py:29-30,35 -- the 14 neighbour offsets, in the reference's order:
//   (1,1,0),(1,-1,0),(-1,1,0),(-1,-1,0),(0,1,1),(0,1,-1),(0,-1,1),(0,-1,-1),
//   (2,0,0),(-2,0,0),(0,2,0),(0,-2,0),(0,0,2),(0,0,-2)   -- see nbi_rt/nbj_rt/nbk_rt below.

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
    int pitchC;   // bytes per padded class row (k=0 at KOFFC; >= KOFFC + L + 12 so that a lane may read 4 bytes past its 8)
    uint8_t* state;     // [(nloc+4)][RJ][pitchS]
    uint8_t* defects;   // same layout
    uint8_t* row_chg;   // [(nloc+4)][L]: 1 if a voxel of row (li, j) was written since prev_state was last brought level with
                        // state (the latent-heat test of the temperature update only looks at such rows)
    uint8_t* cls;       // [(nloc+4)][RJ][pitchC] neighbour-census class of every voxel: bits 1:0 class8(), bits 7:2 the event
                        // count of a listed (interface) voxel (ifc_store(); 0 until its first evaluation)
    double* T;          // [(nloc+4)][L][pitchT]  (current buffer)
    double* theta;
    double* phi;
    double* ovec;       // [(nloc+4)][L][pitchT][3] orientation unit vectors (sin t cos p, sin t sin p, cos t)
    double* rowsum;     // [nloc*3][L]  row sums of the last sweep, index (lp*3+cat)*L + j
    int32_t* rowcnt;
    // interface voxels (voxels owning attachment / diffusion events), see k_interface:
    double* vval;       // [(nloc+4)][L][pitchT] (tidx) per-voxel rate table: the EMPTY-category sum of an empty voxel / the
                        // DIFF-category sum of an atom.  Listed (interface) voxels: written by k_interface / ifc_touch /
                        // k_domain_touch whenever their neighbourhood or T changes.  All other voxels: the nucleation rate
                        // of an empty voxel WITHOUT W/Re/C neighbours at its temperature (K_eff = K_nuc for every neighbour
                        // count, kmc_event_rates.py:122-130), written by k_rate_table after every temperature change.
    double* dep_val;    // [L][pitchT] deposition rate of every (j,k) of plane L-1 from its T alone (kmc_event_rates.py:59-63;
                        // emptiness is tested by the reader); refreshed by k_rate_table; used iff the slab owns plane L-1
    uint8_t* ifc_in;    // same indexing: 1 if the voxel is in ifc_list
    uint32_t* ifc_code; // same indexing: packed neighbourhood of a listed voxel (ifc_encode), kept current by apply
    uint32_t* ifc_list; // packed (lp << 20 | j << 10 | k) of the listed voxels (append-only, superset)
    int* ifc_n;         // number of listed voxels
    __device__ __forceinline__ int64_t sidx(int li, int j, int k) const {
        return ((int64_t)li * RJ + (j + 2)) * pitchS + KOFF + k;
    }
    __device__ __forceinline__ int64_t cidx(int li, int j, int k) const {
        return ((int64_t)li * RJ + (j + 2)) * pitchC + KOFFC + k;
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

// kmc_event_rates.py:11-15: unit vector of an orientation (theta, phi)
__device__ __forceinline__ void orient_vec(double t, double p, double* out)
{
    double st, ct, sp, cp;
    sincos(t, &st, &ct);
    sincos(p, &sp, &cp);
    out[0] = st * cp; out[1] = st * sp; out[2] = ct;
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

// ---- rate primitives shared by every kernel (one expression order everywhere) ------------
// neighbour offsets of a RUNTIME slot m (bit-packed, value+2 in 3 bits per slot)
__device__ __forceinline__ int nbi_rt(int m) { return (int)((0x1248449225bULL >> (3 * m)) & 7ULL) - 2; }
__device__ __forceinline__ int nbj_rt(int m) { return (int)((0x1211225b2cbULL >> (3 * m)) & 7ULL) - 2; }
__device__ __forceinline__ int nbk_rt(int m) { return (int)((0x44922cb492ULL >> (3 * m)) & 7ULL) - 2; }

// kmc_event_rates.py:60-63
__device__ __forceinline__ double dep_rate_s(double nu_dep, double T_melt, double kT, double Tc)
{
    return nu_dep * exp(-(T_melt - Tc) / (kT * Tc));
}
__device__ __forceinline__ double dep_rate(const KParams& P, double Tc)
{
    return dep_rate_s(P.nu_dep, P.T_melt, P.kT, Tc);
}
// exp(x) for x <= 0 (NaN -> 0): round-to-nearest reduction x = n*ln2 + r, |r| <= ln2/2, degree-13
// Taylor polynomial in Horner form (truncation 4e-18), scaled by 2^n.  About 1 ulp; no overflow path
// (x <= 0) and underflow falls out of v_ldexp_f64.  Used for the nucleation, attachment and diffusion rates, whose
// arguments -E/(kT*T) are <= 0 by construction (the deposition rate's can be positive: libm exp there).
// d = a*b + c as ONE v_fma_f64 (hipcc otherwise expands a Horner step with a constant addend into
// v_mov_b64 + v_fmac_f64, doubling the instruction count of the polynomial)
__device__ __forceinline__ double fma1(double a, double b, double c)
{
    double d;
#ifdef CETKMC_EXP_SGPR
    asm("v_fma_f64 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "s"(c));
#else
    asm("v_fma_f64 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
#endif
    return d;
}
__device__ __forceinline__ double exp_nonpos(double x)
{
    x = fmax(x, -1000.0);
    const double n = rint(x * 1.4426950408889634074);
    double r = fma(n, -6.93147180369123816490e-01, x);
    r = fma(n, -1.90821492927058770002e-10, r);
    double p = 1.6059043836821613e-10;                 // 1/13!
    p = fma1(p, r, 2.08767569878681e-09);              // 1/12!
    p = fma1(p, r, 2.505210838544172e-08);             // 1/11!
    p = fma1(p, r, 2.755731922398589e-07);             // 1/10!
    p = fma1(p, r, 2.7557319223985893e-06);            // 1/9!
    p = fma1(p, r, 2.48015873015873e-05);              // 1/8!
    p = fma1(p, r, 1.984126984126984e-04);             // 1/7!
    p = fma1(p, r, 1.388888888888889e-03);             // 1/6!
    p = fma1(p, r, 8.333333333333333e-03);             // 1/5!
    p = fma1(p, r, 4.1666666666666664e-02);            // 1/4!
    p = fma1(p, r, 1.6666666666666666e-01);            // 1/3!
    p = fma(p, r, 0.5);
    p = fma(p, r, 1.0);
    p = fma(p, r, 1.0);
    return ldexp(p, (int)n);
}

// kmc_event_rates.py:126-130 with K_eff from the table.  The reference's two divisions
// barrier = K/den; x = -barrier/kTT are evaluated as x = -K * (1/(den*kTT)) with a Newton-refined
// reciprocal (differs from the reference's rounding by a few ulp of x, i.e. <= ~1e-15 relative in
// the rate), and the
// exponential by exp_nonpos().  Every kernel uses this one function, so GPU-side sums are
// mutually bit-consistent.
__device__ __forceinline__ double rcp_nr(double d)
{   // 1/d for normal positive d: hardware estimate + two Newton steps (~1 ulp, 5 instructions)
    double r = __builtin_amdgcn_rcp(d);
    double e = fma(-d, r, 1.0);
    r = fma(r, e, r);
    e = fma(-d, r, 1.0);
    return fma(r, e, r);
}
__device__ __forceinline__ double nuc_rate_s(double I0, double K, double dT, double kTT)
{
    const double a = dT + 1e-6;
    // max((dT+1e-6)^2, 1e-6), kmc_event_rates.py:127.  Callers only get here with dT > delta_T_c (never NaN), where
    // IEEE max and the reference's Python max agree; one v_max_f64 instead of compare + two selects.
    const double den = fmax(a * a, 1e-6);
    return I0 * exp_nonpos(-K * rcp_nr(den * kTT));
}
__device__ __forceinline__ double nuc_rate(const KParams& P, double K, double dT, double kTT)
{
    return nuc_rate_s(P.I0, K, dT, kTT);
}
// kmc_event_rates.py:147-156: attachment to the empty voxel (li,j,k) from its W/Re/C neighbours.
// The orientation unit vectors (compute_misorientation, :11-20) are kept per voxel in S.ovec
// (written whenever theta/phi change), so the hot kernels need no sin/cos; and since
// cos(arccos(d)) == d to 1 ulp, E_att = 0.5*E_b*(1 - cos(mis)) is evaluated as 0.5*E_b*(1 - d)
// with d the clamped dot product (:21-22).  Rates agree with the reference to ~1e-15 relative.
// Split into a per-voxel context and a per-neighbour item so that every kernel (row re-evaluation,
// interface list, event enumeration) performs the identical arithmetic.
struct AttCtx { double a0, a1, a2, aniso, kTT; };
__device__ __forceinline__ AttCtx att_ctx(const KParams& P, const SlabView& S, int li, int j, int k, double Tc)
{
    const int L = S.L;
    const double* a = S.ovec + 3 * S.tidx(li, j, k);
    const int km = k - 1 > 0 ? k - 1 : 0;
    const int kp = k + 1 < L - 1 ? k + 1 : L - 1;
    const double grad_z = (S.T[S.tidx(li, j, kp)] - S.T[S.tidx(li, j, km)]) * 0.5;
    const double gf = pymax(0.0, grad_z) / pymax(P.T_melt - Tc, 1.0);
    AttCtx c;
    c.a0 = a[0]; c.a1 = a[1]; c.a2 = a[2];
    c.aniso = 1.0 + P.anisotropy * gf;
    c.kTT = P.kT * Tc;
    return c;
}
__device__ __forceinline__ double att_item(const KParams& P, const AttCtx& c, double b0, double b1, double b2, int sn)
{
    double dot = c.a0 * b0 + c.a1 * b1 + c.a2 * b2;
    dot = pymax(pymin(dot, 1.0), -1.0);
    const double E_att = 0.5 * P.E_b[sn - 1] * (1.0 - dot);
    return P.nu * exp_nonpos(-E_att / c.kTT) * c.aniso;      // E_att >= 0: the argument is never positive
}
__device__ __forceinline__ double att_rate(const KParams& P, const SlabView& S, int li, int j, int k,
                                           int di, int dj, int dk, int sn, double Tc)
{
    const AttCtx c = att_ctx(P, S, li, j, k, Tc);
    const double* b = S.ovec + 3 * S.tidx(li + di, j + dj, k + dk);
    return att_item(P, c, b[0], b[1], b[2], sn);
}
// kmc_event_rates.py:93-107: diffusion of the atom st at (li,j,k) into its empty neighbours
struct DiffCtx { double arr, Tc; };
__device__ __forceinline__ DiffCtx diff_ctx(const KParams& P, const SlabView& S, int li, int j, int k, int st, int n_bonds, double Tc)
{
    const int ia = (st == 1) ? 0 : (st == 2) ? 1 : 2;
    const double defect_factor = 1.0 + (double)S.defects[S.sidx(li, j, k)];
    const double E_tot = pymax(P.E_diff[ia] + 0.1 * (double)n_bonds * P.E_b[ia], 0.0);
    DiffCtx c;
    c.arr = exp_nonpos(-defect_factor * E_tot / (P.kT * Tc));  // defect_factor >= 1, E_tot >= 0
    c.Tc = Tc;
    return c;
}
__device__ __forceinline__ double diff_item(const KParams& P, const DiffCtx& c, double Tn_raw)
{
    const double Tn = pymax(Tn_raw, 1.0);
    const double dTn = fabs(c.Tc - Tn);
    const double denom = pymax(P.T_melt - Tn, 1.0);
    const double grad = 1.0 + 0.1 * dTn / denom;
    return P.nu * grad * c.arr;
}
__device__ __forceinline__ double diff_rate(const KParams& P, const SlabView& S, int li, int j, int k,
                                            int di, int dj, int dk, int st, int n_bonds, double Tc)
{
    const DiffCtx c = diff_ctx(P, S, li, j, k, st, n_bonds, Tc);
    return diff_item(P, c, S.T[S.tidx(li + di, j + dj, k + dk)]);
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
            double rate = dep_rate(P, Tc);
            if (finite_d(rate)) emit(CAT_DEP, EV_DEP, rate, -1, 0);
        }
        int n_nb = 0, n_imp = 0;
        unsigned m_src = 0;
#pragma unroll
        for (int m = 0; m < 14; ++m) {
            const int sm = nb(m);
            n_nb += (sm != OOB);
            n_imp += (sm == 2 || sm == 3);
            if (sm >= 1 && sm <= 3) m_src |= 1u << m;
        }
        const double dT = P.T_melt - Tc;
        if (dT > P.delta_T_c) {
            double rate = nuc_rate(P, ktab[n_nb * 15 + n_imp], dT, P.kT * Tc);
            if (rate > P.rate_threshold && finite_d(rate)) emit(CAT_EMPTY, EV_NUC, rate, -1, 1);
        }
        if (m_src) {
            const AttCtx c = att_ctx(P, S, li, j, k, Tc);
            while (m_src) {
                const int m = __builtin_ctz(m_src);
                m_src &= m_src - 1;
                const int sm = nb(m);
                const double* b = S.ovec + 3 * S.tidx(li + nbi_rt(m), j + nbj_rt(m), k + nbk_rt(m));
                double rate = att_item(P, c, b[0], b[1], b[2], sm);
                if (rate > P.rate_threshold && finite_d(rate)) emit(CAT_EMPTY, EV_ATT, rate, m, sm);
            }
        }
    } else if (st != 4) {
        int n_bonds = 0;
        unsigned m_empty = 0;
#pragma unroll
        for (int m = 0; m < 14; ++m) {
            const int sm = nb(m);
            n_bonds += (sm != 0 && sm != OOB);
            if (sm == 0) m_empty |= 1u << m;
        }
        if (m_empty) {
            const DiffCtx c = diff_ctx(P, S, li, j, k, st, n_bonds, Tc);
            while (m_empty) {
                const int m = __builtin_ctz(m_empty);
                m_empty &= m_empty - 1;
                double rate = diff_item(P, c, S.T[S.tidx(li + nbi_rt(m), j + nbj_rt(m), k + nbk_rt(m))]);
                if (rate > P.rate_threshold && finite_d(rate)) emit(CAT_DIFF, EV_DIFF, rate, m, st);
            }
        }
    }
}

}  // namespace cetkmc
