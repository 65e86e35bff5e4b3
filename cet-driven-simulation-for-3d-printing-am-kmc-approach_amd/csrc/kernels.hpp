// kernels.hpp -- HIP kernels of the KMC stepping engine (gfx950, wave64).
//
//   k_sweep         per-voxel rate evaluation + row reduction       (kmc_event_rates.py:162-176)
//   k_plane_reduce  row sums -> block sums                          (kmc_simulation.py:259)
//   k_select        canonical-tree descent to the chosen event      (kmc_simulation.py:265-274)
//   k_apply*        lattice update + RNG bookkeeping                (kmc_simulation.py:276-327)
//   k_thermal       7-point explicit Euler + clip                   (thermal_solver.py:36-117)
//   k_enumerate     event list materialisation (parity / small L)   (kmc_event_rates.py:162-176)
//   k_pack/k_unpack reference layout <-> padded device layout
//
// Canonical summation shape: voxel-category sums are sequential over the voxel's
// slots; rows reduce over k, blocks over j and the total over blocks b=3i+c as balanced
// binary trees on power-of-two padded index ranges.
#pragma once
#include "voxel.hpp"
#include "../../include/cetkmc.h"

namespace cetkmc {

struct BlockEnt { double sum; int64_t cnt; };   // one (plane, category) block, 16 B

// Device-resident stepping state shared by select/apply (one per handle).
struct StepState {
    int64_t cur;        // steps executed in the current batch
    int32_t status;     // 0 ok, 1 terminated, 2 numpy stream exhausted
    int32_t pad0;
    int64_t np_pos;     // cursor into u_np
    int64_t nuc_count;
    double  total;      // last sweep
    int64_t n_events;
    int64_t n_dep;
    int64_t q_pos;
};

struct BatchCfg {
    int64_t step0;
    int64_t np_cap;
    double defect_fraction;
    uint64_t seed;
    int32_t rng_mode;
    int32_t batch;      // 1: r = u_pick[cur]*total and RNG bookkeeping; 0: direct r
};

constexpr int SWEEP_TJ = 8;        // rows of one plane per sweep block
constexpr int PMAX = 2048;         // max leaves of an LDS heap tree (3L <= PMAX)

__device__ __forceinline__ double wave_tree_sum(double v)
{
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) v = v + __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ int wave_sum_i(int v)
{
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) v = v + __shfl_xor(v, o, 64);
    return v;
}
// Binary-counter merge of aligned power-of-two chunks: returns the merged value so far;
// after the last chunk (m = nchunks-1) the return value is the full balanced-tree sum.
template <int LV>
__device__ __forceinline__ double stack_push(double (&stk)[LV], double t, int m)
{
    bool done = false;
#pragma unroll
    for (int b = 0; b < LV; ++b) {
        if (!done) {
            if ((m >> b) & 1) t = stk[b] + t;
            else { stk[b] = t; done = true; }
        }
    }
    return t;
}

// ----------------------------------------------------------------------------------------
// k_sweep_simple (variant 0, straightforward reference form kept for A/B and cross-checks):
// one block = one owned plane x SWEEP_TJ rows.  The 14-neighbour state stencil is
// staged in LDS (5 planes x (TJ+4) rows of the padded u8 state array, 16-B loads); T is
// streamed once, 16 B per lane; theta/phi/defects/T-neighbours are gathered only at
// interface voxels.  Each wave reduces its rows with xor-butterflies (balanced tree over k).
// ----------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_sweep_simple(KParams P, SlabView S, const double* __restrict__ ktab_g,
                                                      const StepState* __restrict__ ss)
{
    if (ss && ss->status) return;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int TJ = SWEEP_TJ, TR = TJ + 4;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int njt = (S.L + TJ - 1) / TJ;
    const int nblk = S.nloc * njt;
    int b = blockIdx.x;
    if ((nblk & 7) == 0) b = (b & 7) * (nblk >> 3) + (b >> 3);   // contiguous plane ranges per XCD
    const int lp = b / njt, jt = b - lp * njt;
    const int j0 = jt * TJ, li = lp + 2, i = S.gi0 + lp;
    const int pitchS = S.pitchS;
    const int tile_bytes = 5 * TR * pitchS;
    double* ktab = reinterpret_cast<double*>(smem + ((tile_bytes + 15) & ~15));

    {   // stage the state tile: planes li-2..li+2, padded rows j0..j0+TR-1
        const int cpr = pitchS >> 4;   // 16-B chunks per row
        const int nchunk = 5 * TR * cpr;
        for (int idx = tid; idx < nchunk; idx += 256) {
            int row = idx / cpr, ch = idx - row * cpr;
            int p = row / TR, rr = row - p * TR;
            const uint4* src = reinterpret_cast<const uint4*>(S.state + ((int64_t)(li - 2 + p) * S.RJ + (j0 + rr)) * pitchS) + ch;
            reinterpret_cast<uint4*>(smem + (int64_t)row * pitchS)[ch] = *src;
        }
        if (tid < 225) ktab[tid] = ktab_g[tid];
    }
    __syncthreads();

    const int nch = S.Pk > 128 ? (S.Pk >> 7) : 1;
    for (int r = w; r < TJ; r += 4) {
        const int j = j0 + r;
        if (j >= S.L) break;
        const unsigned char* own_row = smem + (int64_t)(2 * TR + r + 2) * pitchS + KOFF;
        double stk0[4], stk1[4], stk2[4];
        double row0 = 0.0, row1 = 0.0, row2 = 0.0;
        int c0 = 0, c1 = 0, c2 = 0;
        for (int m = 0; m < nch; ++m) {
            const int k0 = (m << 7) + 2 * lane;
            double s0 = 0.0, s1 = 0.0, s2 = 0.0;
            int cpack = 0;
            if (k0 < S.L) {
                const double2 Tv = *reinterpret_cast<const double2*>(S.T + S.tidx(li, j, k0));
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const int k = k0 + h;
                    const int st = (k < S.L) ? own_row[k] : OOB;
                    double a0 = 0.0, a1 = 0.0, a2 = 0.0;
                    auto nb = [&](int mm) -> int {
                        return smem[(int64_t)((nbi_rt(mm) + 2) * TR + (r + 2 + nbj_rt(mm))) * pitchS + KOFF + k + nbk_rt(mm)];
                    };
                    auto emit = [&](int cat, int, double rate, int, int) {
                        if (cat == CAT_DEP) { a0 += rate; cpack += 1; }
                        else if (cat == CAT_DIFF) { a1 += rate; cpack += (1 << 8); }
                        else { a2 += rate; cpack += (1 << 19); }
                    };
                    eval_voxel(P, S, ktab, li, i, j, k, st, h ? Tv.y : Tv.x, nb, emit);
                    if (h == 0) { s0 = a0; s1 = a1; s2 = a2; }
                    else { s0 = s0 + a0; s1 = s1 + a1; s2 = s2 + a2; }
                }
            }
            if (i == S.L - 1) s0 = wave_tree_sum(s0);
            s1 = wave_tree_sum(s1);
            s2 = wave_tree_sum(s2);
            cpack = wave_sum_i(cpack);
            c0 += cpack & 0xFF; c1 += (cpack >> 8) & 0x7FF; c2 += (cpack >> 19) & 0x7FF;
            row0 = stack_push(stk0, s0, m);
            row1 = stack_push(stk1, s1, m);
            row2 = stack_push(stk2, s2, m);
        }
        if (lane == 0) {
            const int64_t o = (int64_t)lp * 3 * S.L + j;
            S.rowsum[o] = row0; S.rowsum[o + S.L] = row1; S.rowsum[o + 2 * S.L] = row2;
            S.rowcnt[o] = c0; S.rowcnt[o + S.L] = c1; S.rowcnt[o + 2 * S.L] = c2;
        }
    }
}

// ----------------------------------------------------------------------------------------
// k_sweep (variant 1): same tiling, but the rare transcendental-heavy events (attachment,
// diffusion: only at occupied/empty interfaces) are NOT evaluated by the lane that owns the
// voxel.  Phase A evaluates the cheap per-voxel part (deposition, nucleation, neighbour census)
// for 2 voxels per lane; every interface event becomes a 4-byte work item in an LDS queue;
// phase B spreads the items over consecutive lanes (full waves instead of 1-2 busy lanes);
// phase C lets each owner add its items' rates in slot order, so the voxel sums -- and hence
// the canonical tree -- are bit-identical to the simple form.
// ----------------------------------------------------------------------------------------
constexpr int SWEEP_Q = 2048;   // item queue capacity per round

__device__ __forceinline__ double sweep_item_rate(const KParams& P, const SlabView& S, const unsigned char* smem,
                                                  int li, int j0, unsigned desc)
{
    constexpr int TR = SWEEP_TJ + 4;
    const int r = desc & 7, k = (desc >> 3) & 1023, m = (desc >> 13) & 15, n_bonds = (desc >> 17) & 15;
    const int j = j0 + r;
    const int di = nbi_rt(m), dj = nbj_rt(m), dk = nbk_rt(m);
    const int st = smem[(int64_t)(2 * TR + r + 2) * S.pitchS + KOFF + k];
    const double Tc = pymax(S.T[S.tidx(li, j, k)], 1.0);
    double rate;
    if (st == 0) {
        const int sn = smem[(int64_t)((di + 2) * TR + (r + 2 + dj)) * S.pitchS + KOFF + k + dk];
        rate = att_rate(P, S, li, j, k, di, dj, dk, sn, Tc);
    } else {
        rate = diff_rate(P, S, li, j, k, di, dj, dk, st, n_bonds, Tc);
    }
    return (rate > P.rate_threshold && finite_d(rate)) ? rate : -1.0;
}

__global__ __launch_bounds__(256) void k_sweep(KParams P, SlabView S, const double* __restrict__ ktab_g,
                                               const StepState* __restrict__ ss)
{
    if (ss && ss->status) return;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int TJ = SWEEP_TJ, TR = TJ + 4;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int njt = (S.L + TJ - 1) / TJ;
    const int nblk = S.nloc * njt;
    int b = blockIdx.x;
    if ((nblk & 7) == 0) b = (b & 7) * (nblk >> 3) + (b >> 3);   // contiguous plane ranges per XCD
    const int lp = b / njt, jt = b - lp * njt;
    const int j0 = jt * TJ, li = lp + 2, i = S.gi0 + lp;
    const int pitchS = S.pitchS;
    const int tile_bytes = (5 * TR * pitchS + 15) & ~15;
    double* ktab = reinterpret_cast<double*>(smem + tile_bytes);
    double* qres = ktab + 226;                                       // [SWEEP_Q]
    unsigned* qdesc = reinterpret_cast<unsigned*>(qres + SWEEP_Q);   // [SWEEP_Q]
    int* wave_tot = reinterpret_cast<int*>(qdesc + SWEEP_Q);         // [4]

    {
        const int cpr = pitchS >> 4;
        const int nchunk = 5 * TR * cpr;
        for (int idx = tid; idx < nchunk; idx += 256) {
            int row = idx / cpr, ch = idx - row * cpr;
            int p = row / TR, rr = row - p * TR;
            const uint4* src = reinterpret_cast<const uint4*>(S.state + ((int64_t)(li - 2 + p) * S.RJ + (j0 + rr)) * pitchS) + ch;
            reinterpret_cast<uint4*>(smem + (int64_t)row * pitchS)[ch] = *src;
        }
        if (tid < 225) ktab[tid] = ktab_g[tid];
    }
    __syncthreads();

    const bool top = (i == S.L - 1);
    const int nch = S.Pk > 128 ? (S.Pk >> 7) : 1;
    double stk0[2][4], stk1[2][4], stk2[2][4];
    double row0[2] = {0.0, 0.0}, row1[2] = {0.0, 0.0}, row2[2] = {0.0, 0.0};
    int c0[2] = {0, 0}, c1[2] = {0, 0}, c2[2] = {0, 0};
#pragma unroll
    for (int rr = 0; rr < 2; ++rr) {
        const int r = w + 4 * rr;
        const int j = j0 + r;
        const unsigned char* own_row = smem + (int64_t)(2 * TR + r + 2) * pitchS + KOFF;
        for (int m = 0; m < nch; ++m) {
            const int k0 = (m << 7) + 2 * lane;
            // ---- phase A: cheap part of both voxels ------------------------------------------
            double mainv[2] = {0.0, 0.0}, depv[2] = {0.0, 0.0};
            unsigned info[2] = {0u, 0u};   // [0:13] item mask  [14:17] n_bonds  [18] main valid  [19] dep valid  [20:21] 1 diff / 2 empty
            if (j < S.L && k0 < S.L) {
                const double2 Tv = *reinterpret_cast<const double2*>(S.T + S.tidx(li, j, k0));
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const int k = k0 + h;
                    const int st = (k < S.L) ? own_row[k] : OOB;
                    if (st < 128 && st != 4) {
                        int n_nb = 0, n_imp = 0, n_occ = 0;
                        unsigned m_src = 0, m_empty = 0;
#pragma unroll
                        for (int mm = 0; mm < 14; ++mm) {
                            const int sm = smem[(int64_t)((nbi_rt(mm) + 2) * TR + (r + 2 + nbj_rt(mm))) * pitchS + KOFF + k + nbk_rt(mm)];
                            n_nb += (sm != OOB);
                            n_imp += (sm == 2 || sm == 3);
                            n_occ += (sm != 0 && sm != OOB);
                            if (sm >= 1 && sm <= 3) m_src |= 1u << mm;
                            if (sm == 0) m_empty |= 1u << mm;
                        }
                        const double Tc = pymax(h ? Tv.y : Tv.x, 1.0);
                        if (st == 0) {
                            unsigned inf = m_src | (2u << 20);
                            if (top) {
                                const double rate = dep_rate(P, Tc);
                                if (finite_d(rate)) { depv[h] = rate; inf |= 1u << 19; }
                            }
                            const double dT = P.T_melt - Tc;
                            if (dT > P.delta_T_c) {
                                const double rate = nuc_rate(P, ktab[n_nb * 15 + n_imp], dT, P.kT * Tc);
                                if (rate > P.rate_threshold && finite_d(rate)) { mainv[h] = rate; inf |= 1u << 18; }
                            }
                            info[h] = inf;
                        } else {
                            info[h] = m_empty | ((unsigned)n_occ << 14) | (1u << 20);
                        }
                    }
                }
            }
            // ---- phase B: interface events through the block-wide item queue ----------------------
            const int n0 = __builtin_popcount(info[0] & 0x3FFF), n1 = __builtin_popcount(info[1] & 0x3FFF);
            int cnt[2] = {(int)((info[0] >> 18) & 1), (int)((info[1] >> 18) & 1)};
            if (__syncthreads_or(n0 + n1)) {
                int incl = n0 + n1;
#pragma unroll
                for (int o = 1; o < 64; o <<= 1) { int t = __shfl_up(incl, o, 64); if (lane >= o) incl += t; }
                if (lane == 63) wave_tot[w] = incl;
                __syncthreads();
                int base = incl - (n0 + n1), total = 0;
#pragma unroll
                for (int ww = 0; ww < 4; ++ww) { const int t = wave_tot[ww]; if (ww < w) base += t; total += t; }
                const int nrounds = (total + SWEEP_Q - 1) / SWEEP_Q;
                for (int rho = 0; rho < nrounds; ++rho) {
                    if (rho) __syncthreads();
                    int g = base;
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        unsigned mk = info[h] & 0x3FFF;
                        while (mk) {
                            const int mm = __builtin_ctz(mk);
                            mk &= mk - 1;
                            if (g / SWEEP_Q == rho)
                                qdesc[g % SWEEP_Q] = (unsigned)r | ((unsigned)(k0 + h) << 3) | ((unsigned)mm << 13) | (((info[h] >> 14) & 15u) << 17);
                            ++g;
                        }
                    }
                    __syncthreads();
                    const int nq = min(SWEEP_Q, total - rho * SWEEP_Q);
                    for (int q = tid; q < nq; q += 256) qres[q] = sweep_item_rate(P, S, smem, li, j0, qdesc[q]);
                    __syncthreads();
                    g = base;
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        unsigned mk = info[h] & 0x3FFF;
                        while (mk) {
                            mk &= mk - 1;
                            if (g / SWEEP_Q == rho) {
                                const double rt = qres[g % SWEEP_Q];
                                if (rt >= 0.0) { mainv[h] = mainv[h] + rt; ++cnt[h]; }
                            }
                            ++g;
                        }
                    }
                }
            }
            // ---- phase C: canonical reduction (pair, wave butterfly, chunk merge) ------------------
            const int kind0 = (info[0] >> 20) & 3, kind1 = (info[1] >> 20) & 3;
            double s1 = (kind0 == 1 ? mainv[0] : 0.0) + (kind1 == 1 ? mainv[1] : 0.0);
            double s2 = (kind0 == 2 ? mainv[0] : 0.0) + (kind1 == 2 ? mainv[1] : 0.0);
            double s0 = depv[0] + depv[1];
            int cpack = (int)((info[0] >> 19) & 1) + (int)((info[1] >> 19) & 1)
                      + (((kind0 == 1 ? cnt[0] : 0) + (kind1 == 1 ? cnt[1] : 0)) << 8)
                      + (((kind0 == 2 ? cnt[0] : 0) + (kind1 == 2 ? cnt[1] : 0)) << 19);
            if (top) s0 = wave_tree_sum(s0);
            s1 = wave_tree_sum(s1);
            s2 = wave_tree_sum(s2);
            cpack = wave_sum_i(cpack);
            c0[rr] += cpack & 0xFF; c1[rr] += (cpack >> 8) & 0x7FF; c2[rr] += (cpack >> 19) & 0x7FF;
            row0[rr] = stack_push(stk0[rr], s0, m);
            row1[rr] = stack_push(stk1[rr], s1, m);
            row2[rr] = stack_push(stk2[rr], s2, m);
        }
        if (lane == 0 && j < S.L) {
            const int64_t o = (int64_t)lp * 3 * S.L + j;
            S.rowsum[o] = row0[rr]; S.rowsum[o + S.L] = row1[rr]; S.rowsum[o + 2 * S.L] = row2[rr];
            S.rowcnt[o] = c0[rr]; S.rowcnt[o + S.L] = c1[rr]; S.rowcnt[o + 2 * S.L] = c2[rr];
        }
    }
}

// ----------------------------------------------------------------------------------------
// k_sweep_march (variant 2, default): 2.5-D blocking.  One block owns SWEEP_TJ rows and
// marches over MARCH_NI consecutive planes, keeping a 5-plane ring of the u16 census-class
// array in LDS (each class word is fetched ~1.5x instead of 7.5x).  A lane handles 4
// consecutive voxels of a row: the 14-neighbour census (#in-bounds, #empty, #W/Re/C, #Re/C)
// of all 4 voxels is 17 LDS loads + SWAR adds on packed 4-bit counters -- no compares.
// T is streamed once (32 B per lane).  Interface events (attachment/diffusion) go through
// the block-wide item queue as in variant 1; voxel sums are accumulated in slot order, so
// row sums are bit-identical to the simple kernel and to the oracle's canonical tree.
// ----------------------------------------------------------------------------------------
constexpr int MARCH_NI = 8;     // planes per block
constexpr int MARCH_Q = 1024;   // item-queue capacity per round
constexpr int MARCH_MAXCH = 4;  // chunks of 256 voxels per row (L <= 1024)

// Everything the hot part of k_sweep_march needs, by value (stays in SGPRs); the rare
// interface-event path reads the full KParams/SlabView through pointers instead.
struct MarchArgs {
    double T_melt, delta_T_c, kT, I0, rate_threshold, nu_dep;
    int L, gi0, nloc, RJ, pitchC, pitchT, Pk;
    const uint16_t* cls;
    const double* T;
    double* rowsum;
    int32_t* rowcnt;
};

__device__ __noinline__ double march_item_rate(const KParams* __restrict__ Pg, const SlabView* __restrict__ Sg,
                                               int li, int j0, unsigned desc)
{
    const KParams P = *Pg;
    const SlabView S = *Sg;
    const int r = desc & 7, k = (desc >> 3) & 1023, m = (desc >> 13) & 15, n_bonds = (desc >> 17) & 15;
    const int j = j0 + r;
    const int di = nbi_rt(m), dj = nbj_rt(m), dk = nbk_rt(m);
    const int st = S.state[S.sidx(li, j, k)];
    const double Tc = pymax(S.T[S.tidx(li, j, k)], 1.0);
    double rate;
    if (st == 0) rate = att_rate(P, S, li, j, k, di, dj, dk, S.state[S.sidx(li + di, j + dj, k + dk)], Tc);
    else rate = diff_rate(P, S, li, j, k, di, dj, dk, st, n_bonds, Tc);
    return (rate > P.rate_threshold && finite_d(rate)) ? rate : -1.0;
}

// item descriptors of one voxel (rare path): which neighbour slots are sources / free sites
__device__ __noinline__ int march_push_items(const SlabView* __restrict__ Sg, unsigned* qdesc, int li, int j, int r, int k,
                                             bool want_src, unsigned n_bonds, int g, int rho)
{
    const SlabView S = *Sg;
#pragma unroll 1
    for (int mm = 0; mm < 14; ++mm) {
        const int sm = S.state[S.sidx(li + nbi_rt(mm), j + nbj_rt(mm), k + nbk_rt(mm))];
        const bool hit = want_src ? (sm >= 1 && sm <= 3) : (sm == 0);
        if (hit) {
            if (g / MARCH_Q == rho)
                qdesc[g % MARCH_Q] = (unsigned)r | ((unsigned)k << 3) | ((unsigned)mm << 13) | (n_bonds << 17);
            ++g;
        }
    }
    return g;
}

__device__ __forceinline__ unsigned alignbit16(unsigned hi, unsigned lo) { return __builtin_amdgcn_alignbit(hi, lo, 16); }

__global__ __launch_bounds__(256) void k_sweep_march(MarchArgs A, const KParams* __restrict__ Pg,
                                                     const SlabView* __restrict__ Sg, const double* __restrict__ ktab_g,
                                                     const StepState* __restrict__ ss)
{
    if (ss && ss->status) return;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int TJ = SWEEP_TJ, TR = TJ + 4;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int L = A.L;
    const int njt = (L + TJ - 1) / TJ;
    const int nib = (A.nloc + MARCH_NI - 1) / MARCH_NI;
    const int nblk = njt * nib;
    int b = blockIdx.x;
    if ((nblk & 7) == 0) b = (b & 7) * (nblk >> 3) + (b >> 3);   // contiguous block ranges per XCD
    const int ib = b / njt, jt = b - ib * njt;
    const int j0 = jt * TJ;
    const int lp0 = ib * MARCH_NI, lp1 = min(lp0 + MARCH_NI, A.nloc);
    const int pitchC = A.pitchC;
    const int slab = TR * pitchC;                                     // u16 per plane slab
    uint16_t* ring = reinterpret_cast<uint16_t*>(smem);
    const int ring_bytes = (5 * slab * 2 + 15) & ~15;
    double* ktab = reinterpret_cast<double*>(smem + ring_bytes);      // [226]
    double* qres = ktab + 226;                                        // [MARCH_Q]
    double* rowpart = qres + MARCH_Q;                                 // [TJ][3][MARCH_MAXCH]
    unsigned* qdesc = reinterpret_cast<unsigned*>(rowpart + TJ * 3 * MARCH_MAXCH);   // [MARCH_Q]
    int* wave_tot = reinterpret_cast<int*>(qdesc + MARCH_Q);          // [4]

    const int cpr = pitchC >> 3;                                      // 16-B chunks per class row
    auto load_slab = [&](int lsrc) {
        const uint4* src = reinterpret_cast<const uint4*>(A.cls + ((int64_t)lsrc * A.RJ + j0) * pitchC);
        uint4* dst = reinterpret_cast<uint4*>(ring + (lsrc % 5) * slab);
        for (int idx = tid; idx < TR * cpr; idx += 256) dst[idx] = src[idx];
    };
    for (int d = 0; d < 4; ++d) load_slab(lp0 + d);                   // planes li-2 .. li+1 of the first plane
    if (tid < 225) ktab[tid] = ktab_g[tid];

    const int nch = A.Pk > 256 ? (A.Pk >> 8) : 1;
#pragma unroll 1
    for (int lp = lp0; lp < lp1; ++lp) {
        const int li = lp + 2, i = A.gi0 + lp;
        const bool top = (i == L - 1);
        load_slab(li + 2);
        __syncthreads();
        int so[5];
#pragma unroll
        for (int d = 0; d < 5; ++d) so[d] = ((li - 2 + d) % 5) * slab;
#pragma unroll 1
        for (int rr = 0; rr < 2; ++rr) {
            const int r = w + 4 * rr, j = j0 + r;
            int cdep = 0, cdiff = 0, cemp = 0;
#pragma unroll 1
            for (int m = 0; m < nch; ++m) {
                const int k0 = (m << 8) + 4 * lane;
                double mainv[4] = {0.0, 0.0, 0.0, 0.0};
                unsigned info[4] = {0u, 0u, 0u, 0u};   // [1:0] kind 1 diff / 2 empty, [2] main valid, [6:3] #items, [10:7] n_bonds
                double deps = 0.0;
                int nitems = 0;
                if (j < L && k0 < L) {
                    const double2 Ta = *reinterpret_cast<const double2*>(A.T + ((int64_t)li * L + j) * A.pitchT + k0);
                    const double2 Tb = (k0 + 2 < L) ? *reinterpret_cast<const double2*>(A.T + ((int64_t)li * L + j) * A.pitchT + k0 + 2)
                                                    : make_double2(0.0, 0.0);
                    // ---- SWAR census of the 4 voxels' 14 neighbours ------------------------------
                    auto rowp = [&](int d, int row) { return ring + so[d + 2] + row * pitchC + KOFFC + k0; };
                    auto ld2 = [&](const uint16_t* p) { return *reinterpret_cast<const uint2*>(p); };
                    auto ld1 = [&](const uint16_t* p) { return *reinterpret_cast<const unsigned*>(p); };
                    uint2 acc = ld2(rowp(1, r + 3));
                    uint2 t;
                    t = ld2(rowp(1, r + 1)); acc.x += t.x; acc.y += t.y;
                    t = ld2(rowp(-1, r + 3)); acc.x += t.x; acc.y += t.y;
                    t = ld2(rowp(-1, r + 1)); acc.x += t.x; acc.y += t.y;
                    t = ld2(rowp(2, r + 2)); acc.x += t.x; acc.y += t.y;
                    t = ld2(rowp(-2, r + 2)); acc.x += t.x; acc.y += t.y;
                    t = ld2(rowp(0, r + 4)); acc.x += t.x; acc.y += t.y;
                    t = ld2(rowp(0, r + 0)); acc.x += t.x; acc.y += t.y;
#pragma unroll
                    for (int dj = -1; dj <= 1; dj += 2) {          // (0,dj,+-1): neighbours k-1 and k+1
                        const uint16_t* p = rowp(0, r + 2 + dj);
                        const unsigned Aw = ld1(p - 2), Cw = ld1(p + 4);
                        const uint2 B = ld2(p);
                        const unsigned m1 = alignbit16(B.x, Aw), m2 = alignbit16(B.y, B.x), m3 = alignbit16(Cw, B.y);
                        acc.x += m1 + m2; acc.y += m2 + m3;
                    }
                    const uint16_t* po = rowp(0, r + 2);             // own row: neighbours k-2 and k+2
                    const unsigned Aw = ld1(po - 2), Cw = ld1(po + 4);
                    const uint2 own = ld2(po);
                    acc.x += Aw + own.y; acc.y += own.x + Cw;
                    // ---- cheap per-voxel part ----------------------------------------------------
                    double dsum[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
                    for (int h = 0; h < 4; ++h) {
                        const unsigned f = ((h < 2 ? acc.x : acc.y) >> (16 * (h & 1))) & 0xFFFFu;
                        const unsigned oc = ((h < 2 ? own.x : own.y) >> (16 * (h & 1))) & 0xFFFFu;
                        const int n_nb = f & 15, n_empty = (f >> 4) & 15, n_src = (f >> 8) & 15, n_imp = (f >> 12) & 15;
                        if (oc & 0x10u) {                            // empty voxel
                            const double Tc = pymax(h == 0 ? Ta.x : h == 1 ? Ta.y : h == 2 ? Tb.x : Tb.y, 1.0);
                            unsigned inf = 2u | ((unsigned)n_src << 3);
                            if (top) {
                                const double rate = A.nu_dep * exp(-(A.T_melt - Tc) / (A.kT * Tc));
                                if (finite_d(rate)) { dsum[h] = rate; ++cdep; }
                            }
                            const double dT = A.T_melt - Tc;
                            if (dT > A.delta_T_c) {
                                const double aa = dT + 1e-6;
                                const double barrier = ktab[n_nb * 15 + n_imp] / pymax(aa * aa, 1e-6);
                                const double rate = A.I0 * exp(-barrier / (A.kT * Tc));
                                if (rate > A.rate_threshold && finite_d(rate)) { mainv[h] = rate; inf |= 4u; }
                            }
                            info[h] = inf;
                            nitems += n_src;
                        } else if (oc & 0x100u) {                    // W / Re / C atom
                            info[h] = 1u | ((unsigned)n_empty << 3) | ((unsigned)(n_nb - n_empty) << 7);
                            nitems += n_empty;
                        }
                        __builtin_amdgcn_sched_barrier(0);
                    }
                    deps = (dsum[0] + dsum[1]) + (dsum[2] + dsum[3]);
                }
                // ---- interface events through the block-wide item queue ----------------------------
                int cnt[4];
#pragma unroll
                for (int h = 0; h < 4; ++h) cnt[h] = (info[h] >> 2) & 1;
                if (__syncthreads_or(nitems)) {
                    int incl = nitems;
#pragma unroll
                    for (int o = 1; o < 64; o <<= 1) { int t = __shfl_up(incl, o, 64); if (lane >= o) incl += t; }
                    if (lane == 63) wave_tot[w] = incl;
                    __syncthreads();
                    int base = incl - nitems, total = 0;
#pragma unroll
                    for (int ww = 0; ww < 4; ++ww) { const int t = wave_tot[ww]; if (ww < w) base += t; total += t; }
                    const int nrounds = (total + MARCH_Q - 1) / MARCH_Q;
#pragma unroll 1
                    for (int rho = 0; rho < nrounds; ++rho) {
                        if (rho) __syncthreads();
                        if (nitems) {
                            int g = base;
#pragma unroll
                            for (int h = 0; h < 4; ++h)
                                if ((info[h] >> 3) & 15u)
                                    g = march_push_items(Sg, qdesc, li, j, r, k0 + h, (info[h] & 3u) == 2u, (info[h] >> 7) & 15u, g, rho);
                        }
                        __syncthreads();
                        const int nq = min(MARCH_Q, total - rho * MARCH_Q);
#pragma unroll 1
                        for (int q = tid; q < nq; q += 256) qres[q] = march_item_rate(Pg, Sg, li, j0, qdesc[q]);
                        __syncthreads();
                        if (nitems) {
                            int g = base;
#pragma unroll
                            for (int h = 0; h < 4; ++h) {
                                const int ni = (info[h] >> 3) & 15;
                                for (int t = 0; t < ni; ++t, ++g) {
                                    if (g / MARCH_Q == rho) {
                                        const double rt = qres[g % MARCH_Q];
                                        if (rt >= 0.0) { mainv[h] = mainv[h] + rt; ++cnt[h]; }
                                    }
                                }
                            }
                        }
                    }
                }
                // ---- canonical reduction: 4-voxel tree, wave butterfly, chunk partial to LDS --------
                double e[4], d[4];
#pragma unroll
                for (int h = 0; h < 4; ++h) {
                    const unsigned kind = info[h] & 3u;
                    e[h] = (kind == 2u) ? mainv[h] : 0.0;
                    d[h] = (kind == 1u) ? mainv[h] : 0.0;
                    cemp += (kind == 2u) ? cnt[h] : 0;
                    cdiff += (kind == 1u) ? cnt[h] : 0;
                }
                double s2 = (e[0] + e[1]) + (e[2] + e[3]);
                double s1 = (d[0] + d[1]) + (d[2] + d[3]);
                double s0 = deps;
                if (__any(s2 != 0.0)) s2 = wave_tree_sum(s2);
                if (__any(s1 != 0.0)) s1 = wave_tree_sum(s1);
                if (top && __any(s0 != 0.0)) s0 = wave_tree_sum(s0);
                if (lane == 0) {
                    rowpart[(r * 3 + 0) * MARCH_MAXCH + m] = s0;
                    rowpart[(r * 3 + 1) * MARCH_MAXCH + m] = s1;
                    rowpart[(r * 3 + 2) * MARCH_MAXCH + m] = s2;
                }
            }
            // ---- row totals: balanced tree over the chunk partials; counts reduced once ------------
            const int n0 = wave_sum_i(cdep), n1 = wave_sum_i(cdiff), n2 = wave_sum_i(cemp);
            if (lane == 0 && j < L) {
                const int64_t o = (int64_t)lp * 3 * L + j;
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    double* p = rowpart + (r * 3 + c) * MARCH_MAXCH;
                    for (int n = nch; n > 1; n >>= 1)
                        for (int t = 0; t < (n >> 1); ++t) p[t] = p[2 * t] + p[2 * t + 1];
                    A.rowsum[o + (int64_t)c * L] = p[0];
                }
                A.rowcnt[o] = n0; A.rowcnt[o + L] = n1; A.rowcnt[o + 2 * L] = n2;
            }
        }
        __syncthreads();   // ring slot (li-2)%5 is overwritten by the next plane's load
    }
}

// k_plane_reduce: one wave per (owned plane, category): balanced tree over j of the row sums.
__global__ __launch_bounds__(64) void k_plane_reduce(SlabView S, BlockEnt* __restrict__ blocks,
                                                     const StepState* __restrict__ ss)
{
    if (ss && ss->status) return;
    const int b = blockIdx.x, lane = threadIdx.x;
    const int lp = b / 3, c = b - lp * 3;
    const int nch = S.Pk > 64 ? (S.Pk >> 6) : 1;
    double stk[5];
    double tot = 0.0;
    int64_t cnt = 0;
    for (int m = 0; m < nch; ++m) {
        const int j = (m << 6) + lane;
        double v = 0.0;
        int cv = 0;
        if (j < S.L) { v = S.rowsum[(int64_t)b * S.L + j]; cv = S.rowcnt[(int64_t)b * S.L + j]; }
        v = wave_tree_sum(v);
        cnt += wave_sum_i(cv);
        tot = stack_push(stk, v, m);
    }
    if (lane == 0) { blocks[3 * (S.gi0 + lp) + c].sum = tot; blocks[3 * (S.gi0 + lp) + c].cnt = cnt; }
}

// ---- LDS heap tree helpers (leaves at [P,2P), node n has children 2n, 2n+1) -------------
__device__ __forceinline__ void heap_build(double* hs, int* hf, int P, int tid)
{
    for (int n = P >> 1; n >= 1; n >>= 1) {
        for (int idx = tid; idx < n; idx += 256) {
            int node = n + idx;
            hs[node] = hs[2 * node] + hs[2 * node + 1];
            hf[node] = hf[2 * node] | hf[2 * node + 1];
        }
        __syncthreads();
    }
}
// go left iff the right half holds no events, or the left holds events and base+sum(left) >= r
__device__ __forceinline__ int heap_descend(const double* hs, const int* hf, int P, double& base, double r)
{
    int n = 1;
    while (n < P) {
        int l = 2 * n;
        if (hf[l + 1] == 0 || (hf[l] != 0 && base + hs[l] >= r)) n = l;
        else { base += hs[l]; n = l + 1; }
    }
    return n - P;
}

// k_select: single block.  (1) total + termination checks, (2) block descent, (3) row descent
// in the owning slab, (4) voxel descent with the row's rates re-evaluated, (5) slot scan.
__global__ __launch_bounds__(256) void k_select(KParams P, const SlabView* __restrict__ slabs, int nslabs, int L,
                                                int PB, const BlockEnt* __restrict__ blocks, StepState* ss,
                                                BatchCfg cfg, const double* __restrict__ u_pick, double r_direct,
                                                const double* __restrict__ ktab_g, cetkmc_event* my_event, int info_only)
{
    __shared__ double hs[2 * PMAX];
    __shared__ int hf[2 * PMAX];
    __shared__ int leafcnt[PMAX];
    __shared__ double ktab[225];
    __shared__ long long red[256];
    __shared__ double sh_base, sh_r;
    __shared__ int sh_go, sh_b, sh_slab, sh_j, sh_k;
    const int tid = threadIdx.x;
    if (cfg.batch && ss->status) return;
    const int NBk = 3 * L;
    long long csum = 0;
    for (int idx = tid; idx < PB; idx += 256) {
        double v = 0.0; int f = 0;
        if (idx < NBk) { v = blocks[idx].sum; long long c = blocks[idx].cnt; f = c > 0; csum += c; }
        hs[PB + idx] = v; hf[PB + idx] = f;
    }
    if (tid < 225) ktab[tid] = ktab_g[tid];
    red[tid] = csum;
    __syncthreads();
    for (int o = 128; o >= 1; o >>= 1) { if (tid < o) red[tid] += red[tid + o]; __syncthreads(); }
    heap_build(hs, hf, PB, tid);
    if (tid == 0) {
        const double total = hs[1];
        const long long n_events = red[0];
        const long long n_dep = blocks[3 * (L - 1) + CAT_DEP].cnt;
        ss->total = total; ss->n_events = n_events; ss->n_dep = n_dep;
        int go = info_only ? 0 : 1;
        if (!info_only) my_event->type = -1;
        if (n_events == 0 || total < 1e-25 || !finite_d(total)) {
            if (cfg.batch) ss->status = 1;
            go = 0;
        } else if (cfg.batch && go) {
            long long need = (cfg.rng_mode == 0 ? n_dep : 0) + 2;
            if (ss->np_pos + need > cfg.np_cap) { ss->status = 2; go = 0; }
        }
        if (go) {
            const double r = cfg.batch ? u_pick[ss->cur] * total : r_direct;
            double base = 0.0;
            const int b = heap_descend(hs, hf, PB, base, r);
            const int i = b / 3;
            int sl = -1;
            for (int s = 0; s < nslabs; ++s)
                if (i >= slabs[s].gi0 && i < slabs[s].gi0 + slabs[s].nloc) sl = s;
            sh_b = b; sh_slab = sl; sh_base = base; sh_r = r;
            if (sl < 0) go = 0;     // owned by another rank
        }
        sh_go = go;
    }
    __syncthreads();
    if (!sh_go) return;
    const SlabView S = slabs[sh_slab];
    const int b = sh_b, i = b / 3, c = b - 3 * i;
    const int lp = i - S.gi0, li = lp + 2;
    const double r = sh_r;
    const int Pk = S.Pk;
    // rows
    for (int idx = tid; idx < Pk; idx += 256) {
        double v = 0.0; int cv = 0;
        if (idx < L) { v = S.rowsum[((int64_t)lp * 3 + c) * L + idx]; cv = S.rowcnt[((int64_t)lp * 3 + c) * L + idx]; }
        hs[Pk + idx] = v; hf[Pk + idx] = cv > 0; leafcnt[idx] = cv;
    }
    __syncthreads();
    heap_build(hs, hf, Pk, tid);
    if (tid == 0) {
        double base = sh_base;
        const int j = heap_descend(hs, hf, Pk, base, r);
        long long rank = 0;
        if (c == CAT_DEP) for (int jj = 0; jj < j; ++jj) rank += leafcnt[jj];
        red[0] = rank;
        sh_j = j; sh_base = base;
    }
    __syncthreads();
    const int j = sh_j;
    // voxels of row (i, c, j): re-evaluate
    for (int k = tid; k < Pk; k += 256) {
        double sum = 0.0; int cnt = 0;
        if (k < L) {
            const int st = S.state[S.sidx(li, j, k)];
            auto nb = [&](int mm) -> int { return S.state[S.sidx(li + nbi_rt(mm), j + nbj_rt(mm), k + nbk_rt(mm))]; };
            auto emit = [&](int cat, int, double rate, int, int) { if (cat == c) { sum += rate; ++cnt; } };
            eval_voxel(P, S, ktab, li, i, j, k, st, S.T[S.tidx(li, j, k)], nb, emit);
        }
        hs[Pk + k] = sum; hf[Pk + k] = cnt > 0; leafcnt[k] = cnt;
    }
    __syncthreads();
    heap_build(hs, hf, Pk, tid);
    if (tid == 0) {
        double base = sh_base;
        const int k = heap_descend(hs, hf, Pk, base, r);
        long long rank = red[0];
        if (c == CAT_DEP) for (int kk = 0; kk < k; ++kk) rank += leafcnt[kk];
        // slot scan (kmc_simulation.py:268-274 restricted to this voxel's slots)
        const int st = S.state[S.sidx(li, j, k)];
        double cum = base;
        bool found = false;
        int p_type = -1, p_m = -1, p_atom = 0;
        double p_rate = 0.0;
        auto nb = [&](int mm) -> int { return S.state[S.sidx(li + nbi_rt(mm), j + nbj_rt(mm), k + nbk_rt(mm))]; };
        auto emit = [&](int cat, int type, double rate, int m, int atom) {
            if (cat != c || found) return;
            cum += rate;
            p_type = type; p_m = m; p_atom = atom; p_rate = rate;   // remembers the last valid slot
            if (cum >= r) found = true;
        };
        eval_voxel(P, S, ktab, li, i, j, k, st, S.T[S.tidx(li, j, k)], nb, emit);
        cetkmc_event ev;
        ev.type = p_type;
        ev.pos[0] = i; ev.pos[1] = j; ev.pos[2] = k;
        ev.target[0] = ev.target[1] = ev.target[2] = -1;
        ev.atom = p_atom; ev.rate = p_rate;
        ev.dep_rank = (p_type == EV_DEP) ? rank : -1;
        ev.theta = 0.0; ev.phi = 0.0;
        if (p_m >= 0) {
            const int di = nbi_rt(p_m), dj = nbj_rt(p_m), dk = nbk_rt(p_m);
            ev.target[0] = i + di; ev.target[1] = j + dj; ev.target[2] = k + dk;
            // orientation carried by the event: diff moves the source's, att copies the neighbour's
            const int64_t q = (p_type == EV_DIFF) ? S.tidx(li, j, k) : S.tidx(li + di, j + dj, k + dk);
            ev.theta = S.theta[q]; ev.phi = S.phi[q];
        }
        *my_event = ev;
    }
}

// ---- apply -------------------------------------------------------------------------------
__device__ __forceinline__ void write_site(const SlabView& S, int i, int j, int k, int st, double th, double ph)
{
    const int li = i - (S.gi0 - 2);
    if (li < 0 || li >= S.nloc + 4) return;
    S.state[S.sidx(li, j, k)] = (uint8_t)st;
    S.cls[S.cidx(li, j, k)] = class16(st);
    const int64_t q = S.tidx(li, j, k);
    S.theta[q] = th; S.phi[q] = ph;
    orient_vec(th, ph, S.ovec + 3 * q);
}
// kmc_simulation.py:276-327 on every local slab whose extended range holds the voxel(s)
__device__ __forceinline__ void apply_event(const SlabView* slabs, int nslabs, const cetkmc_event& ev, int make_defect)
{
    for (int s = 0; s < nslabs; ++s) {
        const SlabView S = slabs[s];
        int ui = ev.pos[0], uj = ev.pos[1], uk = ev.pos[2];
        if (ev.type == EV_DEP || ev.type == EV_NUC || ev.type == EV_ATT) {
            write_site(S, ui, uj, uk, ev.atom, ev.theta, ev.phi);
        } else if (ev.type == EV_DIFF) {
            write_site(S, ev.target[0], ev.target[1], ev.target[2], ev.atom, ev.theta, ev.phi);
            write_site(S, ui, uj, uk, 0, 0.0, 0.0);
            ui = ev.target[0]; uj = ev.target[1]; uk = ev.target[2];   // :303
        }
        if (make_defect) write_site(S, ui, uj, uk, 4, 0.0, 0.0);       // :323-327
    }
}

__device__ __forceinline__ uint64_t mix64(uint64_t x)
{
    x ^= x >> 30; x *= 0xBF58476D1CE4E5B9ULL;
    x ^= x >> 27; x *= 0x94D049BB133111EBULL;
    x ^= x >> 31;
    return x;
}
// counter-based uniform for the deposition species in rng_mode 1 (DESIGN.md "RNG")
__device__ __forceinline__ double counter_uniform(uint64_t seed, uint64_t step, uint64_t site)
{
    uint64_t x = mix64(seed + 0x9E3779B97F4A7C15ULL * (step + 1));
    x = mix64(x ^ (site * 0xD6E8FEB86659FD93ULL + 0x2545F4914F6CDD1DULL));
    return (double)(x >> 11) * (1.0 / 9007199254740992.0);
}
__device__ __forceinline__ int dep_species(const KParams& P, double u)
{   // kmc_event_rates.py:66-71
    if (u < P.impurity_c) return 3;
    if (u < P.impurity_c + P.impurity_re) return 2;
    return 1;
}

// Batched apply: RNG bookkeeping of one step + lattice update + per-step logs.
__global__ void k_apply_batch(KParams P, const SlabView* __restrict__ slabs, int nslabs, int L,
                              const cetkmc_event* __restrict__ events_all, int G, StepState* ss, BatchCfg cfg,
                              const double* __restrict__ u_defect, const double* __restrict__ u_np,
                              double* log_total, cetkmc_event* log_event, int64_t* log_nev)
{
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    if (ss->status) return;
    cetkmc_event ev;
    ev.type = -1;
    for (int g = 0; g < G; ++g) if (events_all[g].type >= 0) ev = events_all[g];
    const int64_t s = ss->cur;
    if (ev.type < 0) { ss->status = 1; return; }
    int64_t pos = ss->np_pos;
    if (ev.type == EV_DEP) {
        const double u = (cfg.rng_mode == 0)
            ? u_np[pos + ev.dep_rank]
            : counter_uniform(cfg.seed, (uint64_t)(cfg.step0 + s), (uint64_t)ev.pos[1] * (uint64_t)L + (uint64_t)ev.pos[2]);
        ev.atom = dep_species(P, u);
    }
    if (cfg.rng_mode == 0) pos += ss->n_dep;
    if (ev.type == EV_DEP || ev.type == EV_NUC) {
        ev.theta = 0.0 + (3.141592653589793 - 0.0) * u_np[pos];       // np.random.uniform(0, pi)
        ev.phi = 0.0 + (6.283185307179586 - 0.0) * u_np[pos + 1];     // np.random.uniform(0, 2*pi)
        pos += 2;
        if (ev.type == EV_NUC) ss->nuc_count += 1;
    }
    const int mk = (cfg.defect_fraction > 0.0 && u_defect[s] < cfg.defect_fraction) ? 1 : 0;
    apply_event(slabs, nslabs, ev, mk);
    ss->np_pos = pos;
    if (log_total) log_total[s] = ss->total;
    if (log_event) log_event[s] = ev;
    if (log_nev) log_nev[s] = ss->n_events;
    ss->cur = s + 1;
}

// Direct apply (cetkmc_apply): everything decided by the host.
__global__ void k_apply_direct(const SlabView* __restrict__ slabs, int nslabs, cetkmc_event ev, int make_defect,
                               StepState* ss)
{
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    if (ev.type == EV_NUC) ss->nuc_count += 1;
    apply_event(slabs, nslabs, ev, make_defect);
}

// ---- thermal -------------------------------------------------------------------------------
struct ThermalCfg {
    double dt, alpha, inv_dx2, clip_lo, clip_hi, T_nan, rho_cp, latent_coef;
    int laser, use_latent, scrub;
};
__device__ __forceinline__ double scrub_T(double x, double T_nan, int on)
{   // np.nan_to_num(T, nan=T_SUB), kmc_simulation.py:249
    if (!on) return x;
    if (x != x) return T_nan;
    if (__builtin_isinf(x)) return x > 0 ? 1.7976931348623157e308 : -1.7976931348623157e308;
    return x;
}
// thermal_solver.py:107-117 (laser=0) / :36-105 (laser=1).  One thread per owned voxel.
// Laplacian = scipy.ndimage.laplace mode='reflect': per axis (-2*T) + (T[-1]+T[+1]) with edge
// replication, accumulated axis0 + axis1 + axis2.
__global__ __launch_bounds__(256) void k_thermal(SlabView S, const double* __restrict__ Tin, double* __restrict__ Tout,
                                                 const uint8_t* __restrict__ prev_state, const double* __restrict__ q_top,
                                                 ThermalCfg C, const StepState* __restrict__ ss)
{
    const int L = S.L;
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    const int j = blockIdx.y, lp = blockIdx.z;
    if (k >= L) return;
    const int li = lp + 2, i = S.gi0 + lp;
    const int64_t c = S.tidx(li, j, k);
    if (ss && ss->status) { Tout[c] = Tin[c]; return; }
    const int im = (i > 0 ? i - 1 : 0) - (S.gi0 - 2), ip = (i < L - 1 ? i + 1 : L - 1) - (S.gi0 - 2);
    const int jm = j > 0 ? j - 1 : 0, jp = j < L - 1 ? j + 1 : L - 1;
    const int km = k > 0 ? k - 1 : 0, kp = k < L - 1 ? k + 1 : L - 1;
    const double tc = scrub_T(Tin[c], C.T_nan, C.scrub);
    const double d0 = tc * -2.0 + (scrub_T(Tin[S.tidx(im, j, k)], C.T_nan, C.scrub) + scrub_T(Tin[S.tidx(ip, j, k)], C.T_nan, C.scrub));
    const double d1 = tc * -2.0 + (scrub_T(Tin[S.tidx(li, jm, k)], C.T_nan, C.scrub) + scrub_T(Tin[S.tidx(li, jp, k)], C.T_nan, C.scrub));
    const double d2 = tc * -2.0 + (scrub_T(Tin[S.tidx(li, j, km)], C.T_nan, C.scrub) + scrub_T(Tin[S.tidx(li, j, kp)], C.T_nan, C.scrub));
    const double lap = ((d0 + d1) + d2) * C.inv_dx2;
    double nt;
    if (!C.laser) {
        nt = tc + (C.dt * C.alpha) * lap;                                   // thermal_solver.py:116
    } else {
        const double q = (i == L - 1) ? q_top[(int64_t)j * L + k] : 0.0;
        double dF = 0.0;
        if (C.use_latent) {
            const int64_t sc = S.sidx(li, j, k);
            dF = (prev_state[sc] == 0 && S.state[sc] != 0) ? 1.0 : 0.0;
        }
        const double dtm = C.dt > 1e-12 ? C.dt : 1e-12;
        dF = dF / dtm;                                                      // :99
        const double dTdt = C.alpha * lap + q / C.rho_cp + C.latent_coef * dF;   // :102
        nt = tc + C.dt * dTdt;                                              // :103
    }
    double v = nt < C.clip_lo ? C.clip_lo : nt;                             // np.clip :105,117
    v = v > C.clip_hi ? C.clip_hi : v;
    Tout[c] = v;
}

// ---- layout conversion -----------------------------------------------------------------------
// src: contiguous (nplanes, L, L) covering global planes [i_begin, i_begin+nplanes)
template <class SRC>
__global__ void k_pack_u8(SlabView S, uint8_t* dst, const SRC* __restrict__ src, int i_begin, int nplanes, int with_cls)
{
    const int L = S.L;
    const int64_t n = (int64_t)nplanes * L * L;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < n; idx += (int64_t)gridDim.x * blockDim.x) {
        int k = (int)(idx % L); int64_t t = idx / L; int j = (int)(t % L); int i = i_begin + (int)(t / L);
        int li = i - (S.gi0 - 2);
        if (li < 0 || li >= S.nloc + 4) continue;
        dst[S.sidx(li, j, k)] = (uint8_t)src[idx];
        if (with_cls) S.cls[S.cidx(li, j, k)] = class16((int)(uint8_t)src[idx]);
    }
}
template <class DST>
__global__ void k_unpack_u8(SlabView S, const uint8_t* __restrict__ srcp, DST* dst, int i_begin, int nplanes)
{
    const int L = S.L;
    const int64_t n = (int64_t)nplanes * L * L;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < n; idx += (int64_t)gridDim.x * blockDim.x) {
        int k = (int)(idx % L); int64_t t = idx / L; int j = (int)(t % L); int i = i_begin + (int)(t / L);
        int li = i - (S.gi0 - 2);
        if (li < 0 || li >= S.nloc + 4) continue;
        dst[idx] = (DST)srcp[S.sidx(li, j, k)];
    }
}
__global__ void k_check_range(const int64_t* __restrict__ src, int64_t n, int lo, int hi, int* bad)
{
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < n; idx += (int64_t)gridDim.x * blockDim.x)
        if (src[idx] < lo || src[idx] > hi) *bad = 1;
}

// orientation unit vectors of every voxel of the slab (after theta/phi uploads)
__global__ void k_orient(SlabView S)
{
    const int64_t n = (int64_t)(S.nloc + 4) * S.L * S.pitchT;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < n; idx += (int64_t)gridDim.x * blockDim.x)
        orient_vec(S.theta[idx], S.phi[idx], S.ovec + 3 * idx);
}

// ---- event list materialisation ----------------------------------------------------------------
// One thread per row (owned plane, category, j); offsets[] is the exclusive prefix of the row
// counts in canonical order, computed on the host from the last sweep.
__global__ void k_enumerate(KParams P, SlabView S, const double* __restrict__ ktab_g, const int64_t* __restrict__ offsets,
                            cetkmc_event* out, int64_t cap)
{
    __shared__ double ktab[225];
    for (int t = threadIdx.x; t < 225; t += blockDim.x) ktab[t] = ktab_g[t];
    __syncthreads();
    const int L = S.L;
    const int64_t row = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (row >= (int64_t)S.nloc * 3 * L) return;
    const int j = (int)(row % L);
    const int bc = (int)(row / L);
    const int lp = bc / 3, c = bc - 3 * lp, li = lp + 2, i = S.gi0 + lp;
    int64_t o = offsets[row];
    for (int k = 0; k < L; ++k) {
        const int st = S.state[S.sidx(li, j, k)];
        auto nb = [&](int mm) -> int { return S.state[S.sidx(li + nbi_rt(mm), j + nbj_rt(mm), k + nbk_rt(mm))]; };
        auto emit = [&](int cat, int type, double rate, int m, int atom) {
            if (cat != c) return;
            if (o < cap) {
                cetkmc_event ev;
                ev.type = type; ev.pos[0] = i; ev.pos[1] = j; ev.pos[2] = k;
                ev.target[0] = ev.target[1] = ev.target[2] = -1;
                if (m >= 0) {
                    ev.target[0] = i + nbi_rt(m); ev.target[1] = j + nbj_rt(m); ev.target[2] = k + nbk_rt(m);
                }
                ev.atom = atom; ev.rate = rate; ev.dep_rank = -1; ev.theta = 0.0; ev.phi = 0.0;
                out[o] = ev;
            }
            ++o;
        };
        eval_voxel(P, S, ktab, li, i, j, k, st, S.T[S.tidx(li, j, k)], nb, emit);
    }
}

}  // namespace cetkmc
