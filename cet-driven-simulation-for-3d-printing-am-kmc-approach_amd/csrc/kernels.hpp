// kernels.hpp -- HIP kernels of the KMC stepping engine (gfx950, wave64).
//
//   k_rate_table    per-voxel rate table from the temperature field      (kmc_event_rates.py:59-63,116-131)
//   k_interface*    category sums of the listed interface voxels         (kmc_event_rates.py:93-107,134-158)
//   k_sweep_stream  census + table lookup + canonical row reduction      (kmc_event_rates.py:162-176)
//   k_sweep_simple  straightforward per-voxel evaluation (cross-check)   (kmc_event_rates.py:42-160)
//   k_rows_eval     the same row reduction for the rows an event made stale (incremental mode)
//   k_plane_reduce  row sums -> block sums                               (kmc_simulation.py:259)
//   k_select*       canonical-tree descent to the chosen event           (kmc_simulation.py:265-274)
//   k_apply*        lattice update + RNG bookkeeping + interface upkeep  (kmc_simulation.py:276-327)
//   k_thermal*      7-point explicit Euler + clip                        (thermal_solver.py:36-117)
//   k_enumerate     event list materialisation (parity / small L)        (kmc_event_rates.py:162-176)
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
    double  min_margin; // batch: smallest distance of a pick r from the boundaries of the chosen event's interval, / total
};

struct BatchCfg {
    int64_t step0;
    int64_t np_cap;
    double defect_fraction;
    uint64_t seed;
    int32_t rng_mode;   // 0 reference stream, 1 counter species draw, 2 every uniform counter based (no host streams)
    int32_t batch;      // 1: r = u_pick[cur]*total and RNG bookkeeping; 0: direct r
};

#ifndef CETKMC_SWEEP_ATTR
#define CETKMC_SWEEP_ATTR          // e.g. __attribute__((amdgpu_waves_per_eu(4, 4))) for A/B builds
#endif
constexpr int SWEEP_TJ = 8;        // rows of one plane per sweep block
constexpr int L_INACTIVE = 1 << 20; // row index of an idle half-wave (>= any L)
constexpr int PMAX = 2048;         // max leaves of an LDS heap tree (3L <= PMAX)

// Balanced binary tree over the 64 lanes, result in every lane.  Levels 1,2 use quad_perm DPP,
// levels 4,8 row_half_mirror / row_mirror DPP (the partner lane holds the sibling subtree's sum),
// levels 16,32 the gfx950 v_permlane16_swap / v_permlane32_swap -- no LDS round trips.
// (update_dpp with bound_ctrl and old = src: every lane of these permutations has a valid source, so the result never
// depends on `old` and the compiler does not have to zero a register pair in front of every v_mov_b32_dpp)
__device__ __forceinline__ double dpp_f64(double v, int ctrl_sel)
{
    const unsigned lo = __double2loint(v), hi = __double2hiint(v);
    unsigned a, b;
    if (ctrl_sel == 0) { a = __builtin_amdgcn_update_dpp(lo, lo, 0xB1, 0xF, 0xF, true); b = __builtin_amdgcn_update_dpp(hi, hi, 0xB1, 0xF, 0xF, true); }
    else if (ctrl_sel == 1) { a = __builtin_amdgcn_update_dpp(lo, lo, 0x4E, 0xF, 0xF, true); b = __builtin_amdgcn_update_dpp(hi, hi, 0x4E, 0xF, 0xF, true); }
    else if (ctrl_sel == 2) { a = __builtin_amdgcn_update_dpp(lo, lo, 0x141, 0xF, 0xF, true); b = __builtin_amdgcn_update_dpp(hi, hi, 0x141, 0xF, 0xF, true); }
    else { a = __builtin_amdgcn_update_dpp(lo, lo, 0x140, 0xF, 0xF, true); b = __builtin_amdgcn_update_dpp(hi, hi, 0x140, 0xF, 0xF, true); }
    return __hiloint2double((int)b, (int)a);
}
__device__ __forceinline__ double wave_tree_sum(double v)
{
    v = v + dpp_f64(v, 0);
    v = v + dpp_f64(v, 1);
    v = v + dpp_f64(v, 2);
    v = v + dpp_f64(v, 3);
    {
        const unsigned lo = __double2loint(v), hi = __double2hiint(v);
        const auto r0 = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
        const auto r1 = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
        v = __hiloint2double((int)r1[0], (int)r0[0]) + __hiloint2double((int)r1[1], (int)r0[1]);
    }
    {
        const unsigned lo = __double2loint(v), hi = __double2hiint(v);
        const auto r0 = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
        const auto r1 = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
        v = __hiloint2double((int)r1[0], (int)r0[0]) + __hiloint2double((int)r1[1], (int)r0[1]);
    }
    return v;
}
__device__ __forceinline__ int wave_sum_i(int v)
{
    v += (int)__builtin_amdgcn_update_dpp((unsigned)v, (unsigned)v, 0xB1, 0xF, 0xF, true);
    v += (int)__builtin_amdgcn_update_dpp((unsigned)v, (unsigned)v, 0x4E, 0xF, 0xF, true);
    v += (int)__builtin_amdgcn_update_dpp((unsigned)v, (unsigned)v, 0x141, 0xF, 0xF, true);
    v += (int)__builtin_amdgcn_update_dpp((unsigned)v, (unsigned)v, 0x140, 0xF, 0xF, true);
    { const auto r = __builtin_amdgcn_permlane16_swap((unsigned)v, (unsigned)v, false, false); v = (int)(r[0] + r[1]); }
    { const auto r = __builtin_amdgcn_permlane32_swap((unsigned)v, (unsigned)v, false, false); v = (int)(r[0] + r[1]); }
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
// k_sweep_simple (variant 0, straightforward form kept for A/B and cross-checks):
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
// k_sweep_stream (variants 1 and 2): 2.5-D blocking.  One block owns SWEEP_TJ rows and marches over
// STREAM_NI consecutive planes, keeping a 6-slot ring of the u8 census-class array in LDS (5 planes of the
// stencil + the next plane's slab in flight).  A lane handles 8 consecutive voxels of a row: the 14-neighbour
// census of all 8 is 17 LDS loads + OR / v_alignbyte on packed bytes -- no compares, no counters.
//   TAB (variant 1, default): the per-voxel rate table vval is streamed (64 B per lane): an empty voxel's
//     EMPTY-category sum / an atom's DIFF-category sum is the table entry, the census decides which voxels
//     count (empty; atom with an empty neighbour) and which are interface voxels (event count in bits 7:2 of the class byte).
//   !TAB (variant 2, "recompute"): T is streamed instead and the nucleation rate of every empty voxel without
//     W/Re/C neighbours is evaluated in the sweep (Newton reciprocal + 14-FMA exp per voxel); interface voxels
//     take their sums from vval.  Same bits as variant 1 (k_rate_table evaluates the same nuc_bulk()).
// Plane L-1 adds the deposition rates (dep_val, by temperature).  Row sums follow the canonical tree: 8-voxel
// tree in the lane, DPP / v_permlane*_swap butterfly, chunk tree; event counts are scalar popcounts of ballots.
// HW: rows of <= 256 voxels occupy half a wave, a wave then carries two rows.
// ----------------------------------------------------------------------------------------
#ifndef CETKMC_STREAM_NI
#define CETKMC_STREAM_NI 8
#endif
constexpr int STREAM_NI = CETKMC_STREAM_NI;      // planes per block (A/B builds: -DCETKMC_STREAM_NI=16)
constexpr int STREAM_SLOTS = 6;   // ring slots
constexpr int STREAM_MAXPF = 3;   // 16-B chunks of a class slab per thread (L <= 682: 12 rows x 704 B = 528 chunks)

// Block sum of (owned plane lp, category c) by one wave: balanced tree over j of the row sums.  COH: some row sums
// were written by other blocks of the SAME launch (k_rows_eval's in-launch reduction) -> agent-scope loads.
template <bool COH>
__device__ __forceinline__ void plane_reduce_wave(const double* rowsum, const int32_t* rowcnt, BlockEnt* blocks,
                                                  int L, int Pk, int gi0, int lp, int c, int lane)
{
    const int b = lp * 3 + c;
    const int nch = Pk > 64 ? (Pk >> 6) : 1;
    double stk[5];
    double tot = 0.0;
    int64_t cnt = 0;
    for (int m = 0; m < nch; ++m) {
        const int j = (m << 6) + lane;
        double v = 0.0;
        int cv = 0;
        if (j < L) {
            if (COH) {
                v = __hip_atomic_load(rowsum + (int64_t)b * L + j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                cv = __hip_atomic_load(rowcnt + (int64_t)b * L + j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            } else {
                v = rowsum[(int64_t)b * L + j]; cv = rowcnt[(int64_t)b * L + j];
            }
        }
        v = wave_tree_sum(v);
        cnt += wave_sum_i(cv);
        tot = stack_push(stk, v, m);
    }
    if (lane == 0) { blocks[3 * (gi0 + lp) + c].sum = tot; blocks[3 * (gi0 + lp) + c].cnt = cnt; }
}

struct StreamArgs {
    double T_melt, delta_T_c, kT, I0, rate_threshold, K0;    // K0 = K_eff without W/Re/C neighbours (ktab[0])
    int L, gi0, nloc, RJ, pitchC, pitchT, Pk, group_first, group_count;
    const uint8_t* cls;
    const double* T;          // streamed by the recompute variant
    const double* vval;       // streamed by the table variant; gathered at interface voxels by the recompute variant
    const double* dep_val;    // plane L-1 deposition rates by temperature (SlabView::dep_val)
    double* rowsum;
    int32_t* rowcnt;
};

// kmc_event_rates.py:116-131 for an empty voxel WITHOUT W/Re/C neighbours: n_imp = 0, hence f_imp = 0 and
// K_eff = clamp(K_nuc) = K0 whatever the neighbour count; the rate is a function of the temperature alone.  0 when
// the voxel owns no nucleation event (dT <= delta_T_c, rate <= threshold or not finite).  Same arithmetic as
// eval_voxel()/ifc_eval_empty() with n_imp = 0, so every kernel agrees to the bit.
__device__ __forceinline__ double nuc_bulk(double T_melt, double delta_T_c, double kT, double I0, double thr, double K0, double Traw)
{
    const double Tc = pymax(Traw, 1.0);
    const double dT = T_melt - Tc;
    double r = 0.0;
    if (dT > delta_T_c) {
        r = nuc_rate_s(I0, K0, dT, kT * Tc);
        if (!(r > thr && finite_d(r))) r = 0.0;
    }
    return r;
}

// half == true: butterfly over the 32 lanes of each half-wave only
template <bool HALF>
__device__ __forceinline__ double wave_tree_sum_h(double v)
{
    v = v + dpp_f64(v, 0);
    v = v + dpp_f64(v, 1);
    v = v + dpp_f64(v, 2);
    v = v + dpp_f64(v, 3);
    {
        const unsigned lo = __double2loint(v), hi = __double2hiint(v);
        const auto r0 = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
        const auto r1 = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
        v = __hiloint2double((int)r1[0], (int)r0[0]) + __hiloint2double((int)r1[1], (int)r0[1]);
    }
    if (!HALF) {
        const unsigned lo = __double2loint(v), hi = __double2hiint(v);
        const auto r0 = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
        const auto r1 = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
        v = __hiloint2double((int)r1[0], (int)r0[0]) + __hiloint2double((int)r1[1], (int)r0[1]);
    }
    return v;
}
template <bool HALF>
__device__ __forceinline__ int wave_sum_i_h(int v)
{
    v += (int)__builtin_amdgcn_update_dpp((unsigned)v, (unsigned)v, 0xB1, 0xF, 0xF, true);
    v += (int)__builtin_amdgcn_update_dpp((unsigned)v, (unsigned)v, 0x4E, 0xF, 0xF, true);
    v += (int)__builtin_amdgcn_update_dpp((unsigned)v, (unsigned)v, 0x141, 0xF, 0xF, true);
    v += (int)__builtin_amdgcn_update_dpp((unsigned)v, (unsigned)v, 0x140, 0xF, 0xF, true);
    { const auto r = __builtin_amdgcn_permlane16_swap((unsigned)v, (unsigned)v, false, false); v = (int)(r[0] + r[1]); }
    if (!HALF) { const auto r = __builtin_amdgcn_permlane32_swap((unsigned)v, (unsigned)v, false, false); v = (int)(r[0] + r[1]); }
    return v;
}

__device__ __forceinline__ unsigned alignbyte(unsigned hi, unsigned lo, unsigned n) { return __builtin_amdgcn_alignbyte(hi, lo, n); }
// v & (bit b of w ? ~0 : 0): one v_bfe_i32 + two v_and_b32
__device__ __forceinline__ double mask_f64(double v, unsigned w, int b)
{
    const int m = __builtin_amdgcn_sbfe(w, b, 1);
    return __hiloint2double(__double2hiint(v) & m, __double2loint(v) & m);
}
__device__ __forceinline__ double tree8(const double (&x)[8])
{
    return ((x[0] + x[1]) + (x[2] + x[3])) + ((x[4] + x[5]) + (x[6] + x[7]));
}

// The lane's 8 streamed values (rate-table entries, or temperatures for the recompute variant) of chunk m of row
// jrow of local plane li: 64 contiguous bytes (rows are padded to a multiple of 8 doubles).  Issued one work item
// ahead by the streaming kernel, so that the memory latency runs under the previous row's arithmetic.
template <bool TAB, bool HW>
__device__ __forceinline__ void load_vals(const StreamArgs& A, int li, int jrow, int lane, int m, double (&v)[8])
{
    // unconditional (straight-line code lets the compiler keep these loads in flight across the previous row's
    // arithmetic): lanes outside the lattice read a clamped, valid address; their class bytes are zero, which masks
    // whatever they load
    const int k0 = min((m << 9) + 8 * (HW ? (lane & 31) : lane), A.pitchT - 8);
    const int jr = min(jrow, A.L - 1);
#ifdef CETKMC_COALESCED_TIMING
    // TIMING ONLY (results wrong): every load instruction reads whole lines -- lane l takes 16 B at 16 l + (HW ? 512 : 1024) q
    (void)k0;
    const double2* src = reinterpret_cast<const double2*>((TAB ? A.vval : A.T) + ((int64_t)li * A.L + jr) * A.pitchT + (m << 9));
    const int nl = HW ? 32 : 64, sl_ = HW ? (lane & 31) : lane;
#pragma unroll
    for (int q = 0; q < 4; ++q) { const double2 t = src[min(q * nl + sl_, A.pitchT / 2 - 1)]; v[2 * q] = t.x; v[2 * q + 1] = t.y; }
#else
    const double2* src = reinterpret_cast<const double2*>((TAB ? A.vval : A.T) + ((int64_t)li * A.L + jr) * A.pitchT + k0);
#pragma unroll
    for (int q = 0; q < 4; ++q) { const double2 t = src[q]; v[2 * q] = t.x; v[2 * q + 1] = t.y; }
#endif
}

// One lattice row per wave (HW: per half-wave; `jrow` is then the lane's own row) of plane li: census + rates +
// canonical row reduction, results to rowsum/rowcnt.  `rowp(d, dj)` returns the class-row pointer (at k = 0) of plane
// li+d, row jrow+dj -- an LDS ring slot in the streaming kernel, the global class array in the dirty-row kernel;
// everything else is shared, so both produce bit-identical row sums.  v0 = load_vals() of chunk 0.
// CH2: rows of 513..1024 voxels (two 512-voxel chunks per row; only without HW).
template <bool TAB, bool HW, bool CH2, class ROWP>
__device__ __forceinline__ void sweep_row(const StreamArgs& A, ROWP rowp, int li, int lp, int jrow, bool top, int lane,
                                          const double (&v0)[8])
{
    static_assert(!(HW && CH2), "half-wave rows have one chunk");
    const int L = A.L;
    const int sl = HW ? (lane & 31) : lane;
    constexpr int nch = CH2 ? 2 : 1;                         // chunks of 512 voxels
    double r0 = 0.0, r1 = 0.0, r2 = 0.0;                     // row totals (chunk tree: nch <= 2)
    int nE_a = 0, nE_b = 0, nD_a = 0, nD_b = 0;              // scalar event counts: whole wave / half a, half b
    int cI = 0;                                              // lane: interface counts, EMPTY | DIFF << 16
    bool any_ifc = false;
#pragma unroll
    for (int m = 0; m < nch; ++m) {
        const int k0 = (m << 9) + 8 * sl;
        const bool active = jrow < L && k0 < L;
        const int64_t trow = ((int64_t)li * L + jrow) * A.pitchT + k0;
        double v[8], ev[8];
        uint2 own = make_uint2(0u, 0u), acc = make_uint2(0u, 0u);
        if (m == 0) {
#pragma unroll
            for (int q = 0; q < 8; ++q) v[q] = v0[q];
        } else {
            load_vals<TAB, HW>(A, li, jrow, lane, m, v);
        }
        if (active) {
            // ---- census of the 8 voxels' 14 neighbours: OR of class bytes ---------------------------
            auto ld8 = [&](const uint8_t* p) { return *reinterpret_cast<const uint2*>(p + k0); };
            auto ld4 = [&](const uint8_t* p, int off) { return *reinterpret_cast<const unsigned*>(p + k0 + off); };
            uint2 t;
            acc = ld8(rowp(1, 1));
            t = ld8(rowp(1, -1)); acc.x |= t.x; acc.y |= t.y;
            t = ld8(rowp(-1, 1)); acc.x |= t.x; acc.y |= t.y;
            t = ld8(rowp(-1, -1)); acc.x |= t.x; acc.y |= t.y;
            t = ld8(rowp(2, 0)); acc.x |= t.x; acc.y |= t.y;
            t = ld8(rowp(-2, 0)); acc.x |= t.x; acc.y |= t.y;
            t = ld8(rowp(0, 2)); acc.x |= t.x; acc.y |= t.y;
            t = ld8(rowp(0, -2)); acc.x |= t.x; acc.y |= t.y;
#pragma unroll
            for (int dj = -1; dj <= 1; dj += 2) {                  // (0,dj,+-1): neighbours k-1 and k+1
                const uint8_t* p = rowp(0, dj);
                const unsigned Aw = ld4(p, -4), Cw = ld4(p, 8);
                const uint2 B = ld8(p);
                const unsigned mid = alignbyte(B.y, B.x, 1);       // bytes k0+1 .. k0+4
                acc.x |= alignbyte(B.x, Aw, 3) | mid;
                acc.y |= alignbyte(B.y, B.x, 3) | alignbyte(Cw, B.y, 1);
            }
            const uint8_t* po = rowp(0, 0);                        // own row: neighbours k-2 and k+2
            const unsigned Aw = ld4(po, -4), Cw = ld4(po, 8);
            own = ld8(po);
            const unsigned mid = alignbyte(own.y, own.x, 2);       // bytes k0+2 .. k0+5
            acc.x |= alignbyte(own.x, Aw, 2) | mid;
            acc.y |= mid | alignbyte(Cw, own.y, 2);
        }
        // bit 0 of every byte: E own voxel empty; ifE empty with a W/Re/C neighbour (attachment: interface voxel);
        // ifA W/Re/C atom with an empty neighbour (diffusion: interface voxel)
        const uint2 E = make_uint2(own.x & 0x01010101u, own.y & 0x01010101u);
        const uint2 ifE = make_uint2(E.x & (acc.x >> 1), E.y & (acc.y >> 1));
        const uint2 ifA = make_uint2((own.x >> 1) & acc.x & 0x01010101u, (own.y >> 1) & acc.y & 0x01010101u);
        // ---- EMPTY category ---------------------------------------------------------------------------
        if (TAB) {
#pragma unroll
            for (int h = 0; h < 8; ++h) ev[h] = mask_f64(v[h], h < 4 ? E.x : E.y, 8 * (h & 3));
        } else {
            const uint2 bulk = make_uint2(E.x & ~ifE.x, E.y & ~ifE.y);
#pragma unroll
            for (int h = 0; h < 8; ++h) {
                ev[h] = 0.0;
                if (((h < 4 ? bulk.x : bulk.y) >> (8 * (h & 3))) & 1u)
                    ev[h] = nuc_bulk(A.T_melt, A.delta_T_c, A.kT, A.I0, A.rate_threshold, A.K0, v[h]);
            }
        }
        // ---- interface voxels of the lane: event counts from bits 7:2 of their own class byte (kept by ifc_store()),
        // no memory access.  The ballots below count an empty voxel with a non-zero EMPTY-category sum once; an
        // interface voxel's sum is non-zero iff it owns events (every kept rate is > rate_threshold): add count - 1.
        const uint2 iany = make_uint2(ifE.x | ifA.x, ifE.y | ifA.y);
        double dv[8];
#pragma unroll
        for (int h = 0; h < 8; ++h) dv[h] = 0.0;
        const bool diff_here = __any((ifA.x | ifA.y) != 0u);
        if (__any((iany.x | iany.y) != 0u)) {
            any_ifc = true;
            const uint2 c6 = make_uint2((own.x >> 2) & 0x3F3F3F3Fu, (own.y >> 2) & 0x3F3F3F3Fu);
            const uint2 cE = make_uint2(c6.x & (ifE.x * 255u), c6.y & (ifE.y * 255u));
            const uint2 cA = make_uint2(c6.x & (ifA.x * 255u), c6.y & (ifA.y * 255u));
            const unsigned nz = __popc(((cE.x + 0x3F3F3F3Fu) >> 6) & 0x01010101u) + __popc(((cE.y + 0x3F3F3F3Fu) >> 6) & 0x01010101u);
            const unsigned nE = __builtin_amdgcn_sad_u8(cE.x, 0u, __builtin_amdgcn_sad_u8(cE.y, 0u, 0u)) - nz;
            const unsigned nA = __builtin_amdgcn_sad_u8(cA.x, 0u, __builtin_amdgcn_sad_u8(cA.y, 0u, 0u));
            cI += (int)(nE | (nA << 16));
            if (!TAB) {         // recompute variant: the interface voxels' sums come from the table
                unsigned long long bits = ((unsigned long long)iany.y << 32) | iany.x;
                const unsigned long long ebits = ((unsigned long long)ifE.y << 32) | ifE.x;
                while (bits) {
                    const int b = __builtin_ctzll(bits);
                    bits &= bits - 1;
                    const int h = b >> 3;
                    const bool isE = (ebits >> b) & 1ull;
                    const double x = A.vval[trow + h];
#pragma unroll
                    for (int q = 0; q < 8; ++q) {
                        if (q == h && isE) ev[q] = x;
                        if (q == h && !isE) dv[q] = x;
                    }
                }
            }
        }
        if (TAB && diff_here) {
#pragma unroll
            for (int h = 0; h < 8; ++h) dv[h] = mask_f64(v[h], h < 4 ? ifA.x : ifA.y, 8 * (h & 3));
        }
        // ---- counts: scalar popcounts of ballots (bulk voxels count iff their rate is kept) -------------
#pragma unroll
        for (int h = 0; h < 8; ++h) {
            const unsigned long long bm = __ballot(ev[h] != 0.0);
            if (HW) { nE_a += __popc((unsigned)bm); nE_b += __popc((unsigned)(bm >> 32)); }
            else nE_a += __popcll(bm);
        }
        // ---- sums: 8-voxel tree, wave butterfly, chunk tree ------------------------------------------------
        double s2 = wave_tree_sum_h<HW>(tree8(ev));
        r2 = (m == 0) ? s2 : r2 + s2;
        double s1 = 0.0;                                         // a chunk without diffusing atoms contributes +0.0
        if (diff_here) s1 = wave_tree_sum_h<HW>(tree8(dv));
        r1 = (m == 0) ? s1 : r1 + s1;
        if (top) {          // nu_dep * exp(-(T_melt - T')/(kT T')) of plane L-1 by temperature (k_rate_table): kept iff finite
            double dp[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) dp[q] = 0.0;
            if (active) {
                const double2* src = reinterpret_cast<const double2*>(A.dep_val + (int64_t)jrow * A.pitchT + k0);
#pragma unroll
                for (int q = 0; q < 4; ++q) { const double2 t = src[q]; dp[2 * q] = t.x; dp[2 * q + 1] = t.y; }
            }
#pragma unroll
            for (int h = 0; h < 8; ++h) {
                const bool keep = (((h < 4 ? E.x : E.y) >> (8 * (h & 3))) & 1u) && finite_d(dp[h]);
                dp[h] = keep ? dp[h] : 0.0;
                const unsigned long long bm = __ballot(keep);
                if (HW) { nD_a += __popc((unsigned)bm); nD_b += __popc((unsigned)(bm >> 32)); }
                else nD_a += __popcll(bm);
            }
            const double s0 = wave_tree_sum_h<HW>(tree8(dp));
            r0 = (m == 0) ? s0 : r0 + s0;
        }
    }
    int n2 = HW ? ((lane & 32) ? nE_b : nE_a) : nE_a;
    const int n0 = HW ? ((lane & 32) ? nD_b : nD_a) : nD_a;
    int n1 = 0;
    if (any_ifc) {
        const int packed = wave_sum_i_h<HW>(cI);
        n2 += packed & 0xFFFF;
        n1 = packed >> 16;
    }
    if (sl == 0 && jrow < L) {
        const int64_t o = (int64_t)lp * 3 * L + jrow;
        A.rowsum[o] = r0; A.rowsum[o + L] = r1; A.rowsum[o + 2 * (int64_t)L] = r2;
        A.rowcnt[o] = n0; A.rowcnt[o + L] = n1; A.rowcnt[o + 2 * (int64_t)L] = n2;
    }
}

// NPF = 16-B chunks of a class slab per thread (1 for L <= 256, up to 3 for L <= 682)
template <bool TAB, bool HW, int NPF, bool CH2>
__global__ __launch_bounds__(256) CETKMC_SWEEP_ATTR void k_sweep_stream(StreamArgs A, const StepState* __restrict__ ss)
{
    if (ss && ss->status) return;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int TJ = SWEEP_TJ, TR = TJ + 4;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int L = A.L;
    const int njt = (L + TJ - 1) / TJ;
    const int nblk = njt * A.group_count;
    int b = blockIdx.x;
    if ((nblk & 7) == 0) b = (b & 7) * (nblk >> 3) + (b >> 3);   // contiguous block ranges per XCD
    const int ibr = b / njt, jt = b - ibr * njt;
    const int j0 = jt * TJ;
    const int lp0 = (A.group_first + ibr) * STREAM_NI, lp1 = min(lp0 + STREAM_NI, A.nloc);
    const int pitchC = A.pitchC;
    const int slab = TR * pitchC;                                     // bytes per plane slab
    const int nchunk = slab >> 4;                                     // 16-B chunks per slab (<= 256 * NPF)
    const int64_t pstride = (int64_t)A.RJ * pitchC / 16;              // uint4 per class plane
    const uint4* cls4 = reinterpret_cast<const uint4*>(A.cls + (int64_t)j0 * pitchC);
    // class slab of local plane lsrc (rows j0-2 .. j0+TJ+1) -> registers -> ring slot lsrc % STREAM_SLOTS.  Unconditional,
    // clamped loads: straight-line code, so the compiler can count what is in flight.
    // (three named registers, not an array: an array captured by the lambdas below is not promoted out of private memory for
    // NPF > 1 -- the 512^3 sweep then wrote and re-read its prefetched slab through SCRATCH, 256 B per thread and launch:
    // 256 MiB of writes + as many reads on top of 1.2 GB of algorithmic traffic, profiles/r03_512_*)
    static_assert(NPF >= 1 && NPF <= 3, "class-slab prefetch registers");
    uint4 pf0 = make_uint4(0u, 0u, 0u, 0u), pf1 = pf0, pf2 = pf0;
    auto fetch_slab = [&](int lsrc) {
        const uint4* src = cls4 + min(lsrc, A.nloc + 3) * pstride;
        pf0 = src[min(tid, nchunk - 1)];
        if (NPF > 1) pf1 = src[min(tid + 256, nchunk - 1)];
        if (NPF > 2) pf2 = src[min(tid + 512, nchunk - 1)];
    };
    auto store_slab = [&](int lsrc) {
        uint4* dst = reinterpret_cast<uint4*>(smem + (lsrc % STREAM_SLOTS) * slab);
        if (tid < nchunk) dst[tid] = pf0;
        if (NPF > 1) { if (tid + 256 < nchunk) dst[tid + 256] = pf1; }
        if (NPF > 2) { if (tid + 512 < nchunk) dst[tid + 512] = pf2; }
    };
    // work items: (plane, row pass); a wave's row of pass rr is j0 + r(rr).  The values of item n+1 and the class slab of
    // the next plane are requested before item n is computed; two value buffers alternate (no register copies).
    constexpr int NP = HW ? 1 : TJ / 4;
    auto row_of = [&](int rr) { return HW ? (2 * w + (lane >> 5)) : (w + 4 * rr); };
    double va[8], vb[8];
    load_vals<TAB, HW>(A, lp0 + 2, j0 + row_of(0), lane, 0, va);
    {   // planes li-2 .. li+2 of the first plane: all five slabs requested before the first is stored
#pragma unroll
        for (int q = 0; q < NPF; ++q) {
            const int idx = min(tid + 256 * q, nchunk - 1);
            uint4 t[5];
#pragma unroll
            for (int d = 0; d < 5; ++d) t[d] = cls4[(lp0 + d) * pstride + idx];
            if (tid + 256 * q < nchunk) {
#pragma unroll
                for (int d = 0; d < 5; ++d) reinterpret_cast<uint4*>(smem + ((lp0 + d) % STREAM_SLOTS) * slab)[idx] = t[d];
            }
        }
    }
    __syncthreads();
    // one plane: its NP row passes; `cur` holds the first pass's values on entry, `nxt` the next plane's first pass on exit
    auto plane = [&](int lp, double (&cur)[8], double (&nxt)[8]) {
        const int li = lp + 2;
        const bool top = (A.gi0 + lp == L - 1);
        int so[5];
#pragma unroll
        for (int d = 0; d < 5; ++d) so[d] = ((li - 2 + d) % STREAM_SLOTS) * slab;
        // next plane's new slab: requested first, stored at the end of this plane while the value loads are still in flight.
        // After the block's last plane the (unconditional) requests go to what the block has just read -- its last slab, its
        // last plane's values: cache hits, not another block's data dragged across the fabric for nothing
#ifdef CETKMC_NO_TAIL_PREFETCH
        const bool more = lp + 1 < lp1;                                 // A/B: no requests beyond the block's own planes
        if (more) fetch_slab(li + 3);
#else
        fetch_slab(min(li + 3, lp1 + 3));
#endif
        if (NP == 1) {
#ifdef CETKMC_NO_TAIL_PREFETCH
            if (more)
#endif
            load_vals<TAB, HW>(A, min(li + 1, lp1 + 1), j0 + row_of(0), lane, 0, nxt);
            const int r = row_of(0);
            auto rowp = [&](int d, int dj) { return (const uint8_t*)smem + so[d + 2] + (r + 2 + dj) * pitchC + KOFFC; };
            sweep_row<TAB, HW, CH2>(A, rowp, li, lp, j0 + r, top, lane, cur);
        } else {
            {
                load_vals<TAB, HW>(A, li, j0 + row_of(1), lane, 0, nxt);
                const int r = row_of(0);
                auto rowp = [&](int d, int dj) { return (const uint8_t*)smem + so[d + 2] + (r + 2 + dj) * pitchC + KOFFC; };
                sweep_row<TAB, HW, CH2>(A, rowp, li, lp, j0 + r, top, lane, cur);
            }
            {
#ifdef CETKMC_NO_TAIL_PREFETCH
                if (more)
#endif
                load_vals<TAB, HW>(A, min(li + 1, lp1 + 1), j0 + row_of(0), lane, 0, cur);
                const int r = row_of(1);
                auto rowp = [&](int d, int dj) { return (const uint8_t*)smem + so[d + 2] + (r + 2 + dj) * pitchC + KOFFC; };
                sweep_row<TAB, HW, CH2>(A, rowp, li, lp, j0 + r, top, lane, nxt);
            }
        }
#ifdef CETKMC_NO_TAIL_PREFETCH
        if (more)
#endif
        store_slab(li + 3);                  // the sixth slot: not among the five this plane's stencil reads
        __syncthreads();
    };
    if (NP == 1) {
#pragma unroll 1
        for (int lp = lp0; lp < lp1; lp += 2) {
            plane(lp, va, vb);
            if (lp + 1 < lp1) plane(lp + 1, vb, va);
        }
    } else {
#pragma unroll 1
        for (int lp = lp0; lp < lp1; ++lp) plane(lp, va, vb);     // two passes: va -> vb -> va
    }
}

// ----------------------------------------------------------------------------------------
// k_sweep_table (variant 3): the rate sweep WITHOUT a neighbour census.  What the census of k_sweep_stream decides --
// which atoms own diffusion events, which empty voxels are interface voxels with more than one event -- is already
// written in every voxel's OWN class byte: bits 7:2 hold the event count of a listed voxel (ifc_store(), kept current by
// k_interface / ifc_touch / k_domain_touch), and every interface voxel is listed.  So a row reduces from two streams
// alone, the class bytes (1 B per voxel) and the rate table (8 B per voxel), each read exactly once:
//   EMPTY sum  = sum over empty voxels (class bit 0) of their table entry; its count = non-zero entries (+ count - 1 of a
//                listed empty voxel with a count);  DIFF sum = sum over atoms with a non-zero count of their entry;
//   DEP (plane L-1) as before from dep_val.
// No LDS ring, no halo planes or rows, no barriers: 9 B per voxel of traffic for 9 B algorithmic.  The sums, counts and
// the summation tree are those of sweep_row() to the bit (an atom with a non-zero count is an atom with an empty neighbour
// whose diffusion rates were kept; a listed empty voxel that lost its W/Re/C neighbours holds a count of 0 or 1).
// A lane handles 8 consecutive voxels; HW: rows of <= 256 voxels occupy half a wave, a wave item is a pair of rows.  A wave
// walks IPW items 4 apart (the block's four waves read 4 consecutive items at a time), values and class bytes of the next
// item requested before the current one is reduced.
// MEASURED: bit-identical, but not faster than k_sweep_stream -- 32.3 vs 31.3 us (sweep + reduce) at 256^3,
// 229 vs 222 at 512^3; requesting two items ahead or fully coalesced loads make it slower.  The sweep is bound by the memory
// system (5.6 TB/s of HBM traffic at 512^3), not by the census arithmetic.  Kept as an option and as the row kernel of the
// incremental mode (k_rows_eval), where it saves the five-plane gather of class rows.
// ----------------------------------------------------------------------------------------
__device__ __forceinline__ uint2 load_cls8(const StreamArgs& A, int li, int jrow, int sl, int m)
{
    const int k0 = min((m << 9) + 8 * sl, A.pitchT - 8);
    const int jr = min(jrow, A.L - 1);
    return *reinterpret_cast<const uint2*>(A.cls + ((int64_t)li * A.RJ + (jr + 2)) * A.pitchC + KOFFC + k0);
}
template <bool HW, bool CH2>
__device__ __forceinline__ void table_row(const StreamArgs& A, int li, int lp, int jrow, bool top, int lane, const double (&v0)[8],
                                          uint2 own0, double* lds_sum = nullptr, int* lds_cnt = nullptr /* [3][L] of the block's plane */)
{
    static_assert(!(HW && CH2), "half-wave rows have one chunk");
    const int L = A.L;
    const int sl = HW ? (lane & 31) : lane;
    constexpr int nch = CH2 ? 2 : 1;                         // chunks of 512 voxels
    double r0 = 0.0, r1 = 0.0, r2 = 0.0;
    int nE_a = 0, nE_b = 0, nD_a = 0, nD_b = 0;
    int cI = 0;
    bool any_ifc = false;
#pragma unroll
    for (int m = 0; m < nch; ++m) {
        const int k0 = (m << 9) + 8 * sl;
        const bool active = jrow < L && k0 < L;
        double v[8], ev[8];
        uint2 own = own0;
        if (m == 0) {
#pragma unroll
            for (int q = 0; q < 8; ++q) v[q] = v0[q];
        } else {
            load_vals<true, HW>(A, li, jrow, lane, m, v);
            own = load_cls8(A, li, jrow, sl, m);
        }
        if (!active) own = make_uint2(0u, 0u);              // (clamped addresses: whatever such a lane loaded is masked)
        // bit 0 of every byte: E own voxel empty, Aat own voxel a W/Re/C atom; c6 the count field
        const uint2 E = make_uint2(own.x & 0x01010101u, own.y & 0x01010101u);
        const uint2 Aat = make_uint2((own.x >> 1) & 0x01010101u, (own.y >> 1) & 0x01010101u);
        const uint2 c6 = make_uint2((own.x >> 2) & 0x3F3F3F3Fu, (own.y >> 2) & 0x3F3F3F3Fu);
#pragma unroll
        for (int h = 0; h < 8; ++h) ev[h] = mask_f64(v[h], h < 4 ? E.x : E.y, 8 * (h & 3));
        const uint2 cE = make_uint2(c6.x & (E.x * 255u), c6.y & (E.y * 255u));
        const uint2 cA = make_uint2(c6.x & (Aat.x * 255u), c6.y & (Aat.y * 255u));
        const bool diff_here = __any((cA.x | cA.y) != 0u);
        double dv[8];
#pragma unroll
        for (int h = 0; h < 8; ++h) dv[h] = 0.0;
        if (__any((cE.x | cE.y | cA.x | cA.y) != 0u)) {
            any_ifc = true;
            // a listed empty voxel with events is counted once by the ballots below (its sum is non-zero): add count - 1
            const unsigned nz = __popc(((cE.x + 0x3F3F3F3Fu) >> 6) & 0x01010101u) + __popc(((cE.y + 0x3F3F3F3Fu) >> 6) & 0x01010101u);
            const unsigned nE = __builtin_amdgcn_sad_u8(cE.x, 0u, __builtin_amdgcn_sad_u8(cE.y, 0u, 0u)) - nz;
            const unsigned nA = __builtin_amdgcn_sad_u8(cA.x, 0u, __builtin_amdgcn_sad_u8(cA.y, 0u, 0u));
            cI += (int)(nE | (nA << 16));
        }
        if (diff_here) {
            const uint2 hasA = make_uint2(((cA.x + 0x3F3F3F3Fu) >> 6) & 0x01010101u, ((cA.y + 0x3F3F3F3Fu) >> 6) & 0x01010101u);
#pragma unroll
            for (int h = 0; h < 8; ++h) dv[h] = mask_f64(v[h], h < 4 ? hasA.x : hasA.y, 8 * (h & 3));
        }
#pragma unroll
        for (int h = 0; h < 8; ++h) {
            const unsigned long long bm = __ballot(ev[h] != 0.0);
            if (HW) { nE_a += __popc((unsigned)bm); nE_b += __popc((unsigned)(bm >> 32)); }
            else nE_a += __popcll(bm);
        }
        double s2 = wave_tree_sum_h<HW>(tree8(ev));
        r2 = (m == 0) ? s2 : r2 + s2;
        double s1 = 0.0;                                         // a chunk without diffusing atoms contributes +0.0
        if (diff_here) s1 = wave_tree_sum_h<HW>(tree8(dv));
        r1 = (m == 0) ? s1 : r1 + s1;
        if (top) {          // nu_dep * exp(-(T_melt - T')/(kT T')) of plane L-1 by temperature (k_rate_table): kept iff finite
            double dp[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) dp[q] = 0.0;
            if (active) {
                const double2* src = reinterpret_cast<const double2*>(A.dep_val + (int64_t)jrow * A.pitchT + k0);
#pragma unroll
                for (int q = 0; q < 4; ++q) { const double2 t = src[q]; dp[2 * q] = t.x; dp[2 * q + 1] = t.y; }
            }
#pragma unroll
            for (int h = 0; h < 8; ++h) {
                const bool keep = (((h < 4 ? E.x : E.y) >> (8 * (h & 3))) & 1u) && finite_d(dp[h]);
                dp[h] = keep ? dp[h] : 0.0;
                const unsigned long long bm = __ballot(keep);
                if (HW) { nD_a += __popc((unsigned)bm); nD_b += __popc((unsigned)(bm >> 32)); }
                else nD_a += __popcll(bm);
            }
            const double s0 = wave_tree_sum_h<HW>(tree8(dp));
            r0 = (m == 0) ? s0 : r0 + s0;
        }
    }
    int n2 = HW ? ((lane & 32) ? nE_b : nE_a) : nE_a;
    const int n0 = HW ? ((lane & 32) ? nD_b : nD_a) : nD_a;
    int n1 = 0;
    if (any_ifc) {
        const int packed = wave_sum_i_h<HW>(cI);
        n2 += packed & 0xFFFF;
        n1 = packed >> 16;
    }
    if (sl == 0 && jrow < L) {
        const int64_t o = (int64_t)lp * 3 * L + jrow;
        A.rowsum[o] = r0; A.rowsum[o + L] = r1; A.rowsum[o + 2 * (int64_t)L] = r2;
        A.rowcnt[o] = n0; A.rowcnt[o + L] = n1; A.rowcnt[o + 2 * (int64_t)L] = n2;
        if (lds_sum) {
            lds_sum[jrow] = r0; lds_sum[L + jrow] = r1; lds_sum[2 * L + jrow] = r2;
            lds_cnt[jrow] = n0; lds_cnt[L + jrow] = n1; lds_cnt[2 * L + jrow] = n2;
        }
    }
}

#ifndef CETKMC_TABLE_IPW
#define CETKMC_TABLE_IPW 8
#endif
constexpr int TABLE_IPW = CETKMC_TABLE_IPW;       // items per wave
template <bool HW, bool CH2>
__global__ __launch_bounds__(256) CETKMC_SWEEP_ATTR void k_sweep_table(StreamArgs A, const StepState* __restrict__ ss)
{
    if (ss && ss->status) return;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int L = A.L;
    const int ipp = HW ? (L + 1) >> 1 : L;                       // items per plane (HW: pairs of rows)
    const int n_items = ipp * A.nloc;
    int b = blockIdx.x;
    if ((gridDim.x & 7) == 0) b = (b & 7) * (int)(gridDim.x >> 3) + (b >> 3);      // contiguous item ranges per XCD
    int item = b * 4 * TABLE_IPW + w;
    if (item >= n_items) return;
    int lp = item / ipp, jp = item - lp * ipp;
    const int sl = HW ? (lane & 31) : lane;
    auto jrow_of = [&](int jpair) { return HW ? 2 * jpair + (lane >> 5) : jpair; };
#ifdef CETKMC_TABLE_DEPTH2
    // A/B: requests TWO items ahead (three rotating buffers)
    double va[8], vb[8], vc[8];
    uint2 ca, cb, cc;
    auto advance = [&](int& lpx, int& jpx) { jpx += 4; while (jpx >= ipp) { jpx -= ipp; ++lpx; } };
    int lp1 = lp, jp1 = jp;
    advance(lp1, jp1);
    const bool has1 = item + 4 < n_items;
    load_vals<true, HW>(A, lp + 2, jrow_of(jp), lane, 0, va);
    ca = load_cls8(A, lp + 2, jrow_of(jp), sl, 0);
    load_vals<true, HW>(A, (has1 ? lp1 : lp) + 2, jrow_of(has1 ? jp1 : jp), lane, 0, vb);
    cb = load_cls8(A, (has1 ? lp1 : lp) + 2, jrow_of(has1 ? jp1 : jp), sl, 0);
    int done = 0;
    auto step = [&](double (&cur)[8], uint2& ccur, double (&far)[8], uint2& cfar) {
        // request item + 8 into `far`, reduce `cur`
        int lp2 = lp, jp2 = jp;
        advance(lp2, jp2);
        int lp3 = lp2, jp3 = jp2;
        advance(lp3, jp3);
        const bool more = (done + 2 < TABLE_IPW) && (item + 8 < n_items);
        load_vals<true, HW>(A, (more ? lp3 : lp) + 2, jrow_of(more ? jp3 : jp), lane, 0, far);
        cfar = load_cls8(A, (more ? lp3 : lp) + 2, jrow_of(more ? jp3 : jp), sl, 0);
        table_row<HW, CH2>(A, lp + 2, lp, jrow_of(jp), A.gi0 + lp == L - 1, lane, cur, ccur);
        item += 4; lp = lp2; jp = jp2; ++done;
    };
#pragma unroll 1
    while (true) {
        step(va, ca, vc, cc);
        if (done >= TABLE_IPW || item >= n_items) break;
        step(vb, cb, va, ca);
        if (done >= TABLE_IPW || item >= n_items) break;
        step(vc, cc, vb, cb);
        if (done >= TABLE_IPW || item >= n_items) break;
    }
#else
    double va[8], vb[8];
    uint2 ca, cb;
    load_vals<true, HW>(A, lp + 2, jrow_of(jp), lane, 0, va);
    ca = load_cls8(A, lp + 2, jrow_of(jp), sl, 0);
    // one item: `cur` holds its values on entry; the next item's are requested into `nxt` first (past the wave's last item
    // the request repeats the current one: a cache hit)
    auto step = [&](double (&cur)[8], uint2& ccur, double (&nxt)[8], uint2& cnxt, bool last) {
        int lp2 = lp, jp2 = jp + 4;
        while (jp2 >= ipp) { jp2 -= ipp; ++lp2; }
        const bool more = !last && (item + 4 < n_items);
        const int lpn = more ? lp2 : lp, jpn = more ? jp2 : jp;
        load_vals<true, HW>(A, lpn + 2, jrow_of(jpn), lane, 0, nxt);
        cnxt = load_cls8(A, lpn + 2, jrow_of(jpn), sl, 0);
        table_row<HW, CH2>(A, lp + 2, lp, jrow_of(jp), A.gi0 + lp == L - 1, lane, cur, ccur);
        item += 4; lp = lp2; jp = jp2;
    };
#pragma unroll 1
    for (int t = 0; t < TABLE_IPW; t += 2) {
        step(va, ca, vb, cb, false);
        if (item >= n_items) break;
        step(vb, cb, va, ca, t + 2 >= TABLE_IPW);
        if (item >= n_items) break;
    }
#endif
}

// k_sweep_plane (variant 4): k_sweep_table with one block per owned plane (16 waves; wave w takes the plane's items w, w + 16, ...),
// which therefore holds all of the plane's row sums when its last row is done: the three block sums (balanced tree over j,
// exactly plane_reduce_wave()) are folded from LDS by three of its waves -- no k_plane_reduce launch, no hand-off between
// workgroups.  Same bits as variants 1 / 3 (+ k_plane_reduce).
template <bool HW, bool CH2>
__global__ __launch_bounds__(1024) CETKMC_SWEEP_ATTR void k_sweep_plane(StreamArgs A, const StepState* __restrict__ ss, BlockEnt* __restrict__ blocks)
{
    if (ss && ss->status) return;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int L = A.L;
    double* lds_sum = reinterpret_cast<double*>(smem);                  // [3][L]
    int* lds_cnt = reinterpret_cast<int*>(lds_sum + 3 * L);             // [3][L]
    const int ipp = HW ? (L + 1) >> 1 : L;                              // items per plane (HW: pairs of rows)
    int lp = blockIdx.x;
    if ((gridDim.x & 7) == 0) lp = (lp & 7) * (int)(gridDim.x >> 3) + (lp >> 3);      // contiguous plane ranges per XCD
    const int sl = HW ? (lane & 31) : lane;
    auto jrow_of = [&](int jpair) { return HW ? 2 * jpair + (lane >> 5) : jpair; };
    const bool top = A.gi0 + lp == L - 1;
    int jp = w;
    if (jp < ipp) {
        double va[8], vb[8];
        uint2 ca, cb;
        load_vals<true, HW>(A, lp + 2, jrow_of(jp), lane, 0, va);
        ca = load_cls8(A, lp + 2, jrow_of(jp), sl, 0);
        auto step = [&](double (&cur)[8], uint2& ccur, double (&nxt)[8], uint2& cnxt) {
            const int jpn = (jp + 16 < ipp) ? jp + 16 : jp;             // past the wave's last item: the current one again (cache hit)
            load_vals<true, HW>(A, lp + 2, jrow_of(jpn), lane, 0, nxt);
            cnxt = load_cls8(A, lp + 2, jrow_of(jpn), sl, 0);
            table_row<HW, CH2>(A, lp + 2, lp, jrow_of(jp), top, lane, cur, ccur, lds_sum, lds_cnt);
            jp += 16;
        };
#pragma unroll 1
        while (true) {
            step(va, ca, vb, cb);
            if (jp >= ipp) break;
            step(vb, cb, va, ca);
            if (jp >= ipp) break;
        }
    }
    __syncthreads();
    if (w < 3) {        // block sum of (plane lp, category w): balanced tree over j, zeros beyond L -- plane_reduce_wave() on the LDS copy
        const int nch = A.Pk > 64 ? (A.Pk >> 6) : 1;
        double stk[5];
        double tot = 0.0;
        int64_t cnt = 0;
        for (int m = 0; m < nch; ++m) {
            const int j = (m << 6) + lane;
            double v = 0.0;
            int cv = 0;
            if (j < L) { v = lds_sum[w * L + j]; cv = lds_cnt[w * L + j]; }
            v = wave_tree_sum(v);
            cnt += wave_sum_i(cv);
            tot = stack_push(stk, v, m);
        }
        if (lane == 0) { blocks[3 * (A.gi0 + lp) + w].sum = tot; blocks[3 * (A.gi0 + lp) + w].cnt = cnt; }
    }
}

// Exact incremental stepping: between two temperature updates an event changes the rates of a few rows
// only (those holding the changed voxel(s) or one of their 14 neighbours).  k_apply_batch records those
// rows; this kernel re-evaluates just them -- one wave per row, class bytes read straight from global
// memory -- with the same sweep_row() as the streaming kernel, so every row sum equals what a full sweep
// would have produced.  dirty[0] = count, dirty[1..] = (global plane << 16) | row.
constexpr int DIRTY_MAX = 63;
template <bool TAB, bool HW, bool CH2>
__global__ __launch_bounds__(256) void k_rows_eval(StreamArgs A, const int* __restrict__ dirty,
                                                   const StepState* __restrict__ ss, BlockEnt* __restrict__ blocks, int* plane_cnt)
{
    if (ss && ss->status) return;
    __shared__ int sh_last;
    const int n_dirty = dirty[0];
    if ((int)blockIdx.x >= n_dirty) return;
    const int e = dirty[1 + blockIdx.x];
    const int i = e >> 16, j = e & 0xFFFF;
    const int lp = i - A.gi0;
    if (lp < 0 || lp >= A.nloc) return;                     // another slab's row
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    if (w == 0) {
        const int li = lp + 2;
        const int jrow = (HW && lane >= 32) ? L_INACTIVE : j;       // the second half-wave idles
        double v0[8];
        load_vals<TAB, HW>(A, li, jrow, lane, 0, v0);
        if (TAB) {          // the census-free row reduction of k_sweep_table: same bits
            const uint2 c0 = load_cls8(A, li, jrow, HW ? (lane & 31) : lane, 0);
            table_row<HW, CH2>(A, li, lp, jrow, A.gi0 + lp == A.L - 1, lane, v0, c0);
        } else {
            auto rowp = [&](int d, int dj) { return A.cls + ((int64_t)(li + d) * A.RJ + (j + 2 + dj)) * A.pitchC + KOFFC; };
            sweep_row<TAB, HW, CH2>(A, rowp, li, lp, jrow, A.gi0 + lp == A.L - 1, lane, v0);
        }
    }
    __syncthreads();
    // the block that finishes a plane's last dirty row reduces that plane's three category blocks (waves 0..2);
    // the plane's counter is the slot of its first entry in the list
    if (tid == 0) {
        int first = -1, n_plane = 0;
        for (int q = 0; q < n_dirty; ++q)
            if ((dirty[1 + q] >> 16) == i) { if (first < 0) first = q; ++n_plane; }
        __threadfence();
        const int prev = __hip_atomic_fetch_add(plane_cnt + first, 1, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
        sh_last = (prev == n_plane - 1);
        if (sh_last) plane_cnt[first] = 0;
    }
    __syncthreads();
    if (sh_last && w < 3) plane_reduce_wave<true>(A.rowsum, A.rowcnt, blocks, A.L, A.Pk, A.gi0, lp, w, lane);
}

// Per-voxel rate table (SlabView::vval) + plane L-1 deposition rates (dep_val) from the temperature field: run after
// every temperature change (upload, thermal update) and parameter change, BEFORE k_interface (which then overwrites
// the entries of the listed voxels with their interface sums).  Two voxels per thread; 16 B/voxel.
__global__ __launch_bounds__(256) void k_rate_table(KParams P, SlabView S, double K0, const StepState* __restrict__ ss)
{
    if (ss && ss->status) return;          // terminated batch: T was passed through unchanged, the table stands
    const int L = S.L, half = S.pitchT >> 1;
    const int64_t n = (int64_t)S.nloc * L * half;
    const int64_t gtid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x, nthr = (int64_t)gridDim.x * blockDim.x;
    for (int64_t idx = gtid; idx < n; idx += nthr) {
        const int kp = (int)(idx % half);
        const int64_t row = idx / half;                    // lp * L + j
        const int64_t t = ((int64_t)2 * L + row) * S.pitchT + 2 * kp;       // local plane lp + 2
        const double2 Tv = *reinterpret_cast<const double2*>(S.T + t);
        double2 r;
        r.x = (2 * kp < L) ? nuc_bulk(P.T_melt, P.delta_T_c, P.kT, P.I0, P.rate_threshold, K0, Tv.x) : 0.0;      // row padding: 0
        r.y = (2 * kp + 1 < L) ? nuc_bulk(P.T_melt, P.delta_T_c, P.kT, P.I0, P.rate_threshold, K0, Tv.y) : 0.0;
        *reinterpret_cast<double2*>(S.vval + t) = r;
    }
    const int lp = L - 1 - S.gi0;                          // kmc_event_rates.py:59-63 for plane L-1
    if (lp >= 0 && lp < S.nloc)
        for (int64_t idx = gtid; idx < (int64_t)L * L; idx += nthr) {
            const int j = (int)(idx / L), k = (int)(idx - (int64_t)j * L);
            S.dep_val[(int64_t)j * S.pitchT + k] = dep_rate(P, pymax(S.T[S.tidx(lp + 2, j, k)], 1.0));
        }
}

// k_plane_reduce: one wave per (owned plane, category): balanced tree over j of the row sums.
__global__ __launch_bounds__(64) void k_plane_reduce(SlabView S, BlockEnt* __restrict__ blocks,
                                                     const StepState* __restrict__ ss)
{
    if (ss && ss->status) return;
    const int b = blockIdx.x;
    plane_reduce_wave<false>(S.rowsum, S.rowcnt, blocks, S.L, S.Pk, S.gi0, b / 3, b % 3, (int)threadIdx.x);
}

// ---- counter-based uniforms (DESIGN.md "RNG"): u(seed, step, key) in [0, 1); the oracle's orc_counter_uniform ----------
__host__ __device__ __forceinline__ uint64_t mix64(uint64_t x)
{
    x ^= x >> 30; x *= 0xBF58476D1CE4E5B9ULL;
    x ^= x >> 27; x *= 0x94D049BB133111EBULL;
    x ^= x >> 31;
    return x;
}
__host__ __device__ __forceinline__ double counter_uniform(uint64_t seed, uint64_t step, uint64_t site)
{
    uint64_t x = mix64(seed + 0x9E3779B97F4A7C15ULL * (step + 1));
    x = mix64(x ^ (site * 0xD6E8FEB86659FD93ULL + 0x2545F4914F6CDD1DULL));
    return (double)(x >> 11) * (1.0 / 9007199254740992.0);
}
// keys of the per-box uniforms of Mode B (| box index) -- and, with box index 0, of the all-counter Mode A stream (rng_mode 2);
// the deposition species is keyed by j * L + k in both (as rng_mode 1)
constexpr uint64_t KEY_PICK = 1ull << 40, KEY_THETA = 2ull << 40, KEY_PHI = 3ull << 40, KEY_DEFECT = 4ull << 40,
                   KEY_ACCEPT = 5ull << 40, KEY_DT = 6ull << 40;
__device__ __forceinline__ int dep_species(const KParams& P, double u)
{   // kmc_event_rates.py:66-71
    if (u < P.impurity_c) return 3;
    if (u < P.impurity_c + P.impurity_re) return 2;
    return 1;
}

// A/B instrumentation (tools/sel_stamps.py, alternative build with -DCETKMC_SEL_STAMPS): thread 0 records the 100 MHz
// wall clock at the phase boundaries of the fused selection + application kernel
#ifdef CETKMC_SEL_STAMPS
__device__ long long g_sel_stamps[16];
#define SEL_STAMP(q) do { if (threadIdx.x == 0) g_sel_stamps[q] = wall_clock64(); } while (0)
#else
#define SEL_STAMP(q) do { } while (0)
#endif

// What the selection already holds when the application starts (fused launch: same block, thread 0): the step state
// and the step's uniforms, requested behind the selection's own loads instead of in a chain of their own.
struct SelCarry {
    double total, u_def, u_th, u_ph;
    double ov[3];          // orientation unit vector of (pi u_th, 2 pi u_ph): computed by an idle wave during the selection
    long long n_events, n_dep, np_pos, cur, nuc_count;
    int ready;
};

// ---- the selection trees, in registers ------------------------------------------------------------------------------
// One canonical tree (leaves = 256 * npt, npt = 1, 2, 4 or 8 consecutive leaves per thread; a shorter tree is padded with
// empty leaves, which changes neither a sum nor a descent) built and descended without an LDS heap: the thread's own
// leaves fold in registers, the six levels of a wave by DPP / v_permlane*_swap with EVERY level's block sum kept (a lane
// holds the sum of each aligned block it belongs to), the four wave sums meet in LDS (barrier 1).  Every thread then walks
// the two top levels; the chosen wave continues with v_readlane of the kept block sums (scalar work, no memory) down to a
// lane, every lane descends its own leaves, the chosen lane's answer is published (barrier 2).  "Holds events" flags are
// ballot bits.  Descent rule: go left iff the right half holds no events, or the left holds events and base + sum(left) >= r
// (DESIGN.md section 3); pairs and order of additions are those of the balanced tree over the leaf index.
struct TreeScratch { double ws[4]; unsigned long long wm[4]; double base; int leaf; };
struct TreeSel { int leaf; double base, total; };
__device__ __forceinline__ double readlane_f64(double v, int lane_uniform)
{
    const int l = __builtin_amdgcn_readfirstlane(lane_uniform);
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), l), __builtin_amdgcn_readlane(__double2loint(v), l));
}
template <class RFN>
__device__ __forceinline__ TreeSel tree_select(const double (&lf)[8], unsigned fm, int npt, double base0, RFN rfn, TreeScratch& X, int tid)
{
    const int lane = tid & 63, w = tid >> 6;
    // the thread's 8 leaf slots (npt used, the rest empty)
    const double p0 = lf[0] + lf[1], p1 = lf[2] + lf[3], p2 = lf[4] + lf[5], p3 = lf[6] + lf[7];
    const double q0 = p0 + p1, q1 = p2 + p3;
    double lv[7];
    lv[0] = q0 + q1;
    lv[1] = lv[0] + dpp_f64(lv[0], 0);
    lv[2] = lv[1] + dpp_f64(lv[1], 1);
    lv[3] = lv[2] + dpp_f64(lv[2], 2);
    lv[4] = lv[3] + dpp_f64(lv[3], 3);
    {
        const unsigned lo = __double2loint(lv[4]), hi = __double2hiint(lv[4]);
        const auto r0 = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
        const auto r1 = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
        lv[5] = __hiloint2double((int)r1[0], (int)r0[0]) + __hiloint2double((int)r1[1], (int)r0[1]);
    }
    {
        const unsigned lo = __double2loint(lv[5]), hi = __double2hiint(lv[5]);
        const auto r0 = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
        const auto r1 = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
        lv[6] = __hiloint2double((int)r1[0], (int)r0[0]) + __hiloint2double((int)r1[1], (int)r0[1]);
    }
    const unsigned long long mask = __ballot(fm != 0u);
    if (lane == 0) { X.ws[w] = lv[6]; X.wm[w] = mask; }
    __syncthreads();
    const double w0 = X.ws[0], w1 = X.ws[1], w2 = X.ws[2], w3 = X.ws[3];
    const unsigned long long m0 = X.wm[0], m1 = X.wm[1], m2 = X.wm[2], m3 = X.wm[3];
    const double h0 = w0 + w1, h1 = w2 + w3;
    TreeSel out;
    out.total = h0 + h1;
    const double r = rfn(out.total);
    double base = base0;
    int wsel;
    {   // go left iff the right half holds no events, or the left holds events and base + sum(left) >= r
        int half;
        if ((m2 | m3) == 0ull || ((m0 | m1) != 0ull && base + h0 >= r)) half = 0;
        else { base += h0; half = 1; }
        const double sl = half ? w2 : w0;
        const unsigned long long ml = half ? m2 : m0, mr = half ? m3 : m1;
        if (mr == 0ull || (ml != 0ull && base + sl >= r)) wsel = 2 * half;
        else { base += sl; wsel = 2 * half + 1; }
    }
    if (w == wsel) {
        int a = 0;
#pragma unroll
        for (int l = 5; l >= 0; --l) {
            const int hw = 1 << l;
            const double sl = readlane_f64(lv[l], a);
            const unsigned long long bits = (hw == 32) ? 0xFFFFFFFFull : ((1ull << hw) - 1ull);
            const bool fl = ((mask >> a) & bits) != 0ull, fr = ((mask >> (a + hw)) & bits) != 0ull;
            if (!(!fr || (fl && base + sl >= r))) { base += sl; a += hw; }
        }
        // inside the lane: every lane walks its own 8 slots from the same base; the chosen lane's walk counts
        double bl = base;
        int slot;
        {
            const bool fq0 = (fm & 0x0Fu) != 0u, fq1 = (fm & 0xF0u) != 0u;
            int hq;
            if (!fq1 || (fq0 && bl + q0 >= r)) hq = 0; else { bl += q0; hq = 1; }
            const double pl = hq ? p2 : p0;
            const unsigned f4 = (fm >> (4 * hq)) & 0xFu;
            int hp;
            if ((f4 & 0xCu) == 0u || ((f4 & 0x3u) != 0u && bl + pl >= r)) hp = 0; else { bl += pl; hp = 1; }
            const int s2 = 4 * hq + 2 * hp;
            const double ll = (s2 == 0) ? lf[0] : (s2 == 2) ? lf[2] : (s2 == 4) ? lf[4] : lf[6];
            const unsigned f2 = (fm >> s2) & 0x3u;
            int hl;
            if ((f2 & 0x2u) == 0u || ((f2 & 0x1u) != 0u && bl + ll >= r)) hl = 0; else { bl += ll; hl = 1; }
            slot = s2 + hl;
        }
        const int la = __builtin_amdgcn_readfirstlane(a);
        const int slot_a = __builtin_amdgcn_readlane(slot, la);
        const double base_a = readlane_f64(bl, la);
        if (lane == 0) { X.leaf = (wsel * 64 + la) * npt + slot_a; X.base = base_a; }
    }
    __syncthreads();
    out.leaf = X.leaf; out.base = X.base;
    return out;
}

// k_select: single block.  (1) total + termination checks, (2) block descent, (3) row descent
// in the owning slab, (4) voxel descent (leaves looked up in the rate table), (5) slot scan.
// cur_hint >= 0: the batch step index as the host knows it (== ss->cur while status == 0): saves a dependent load.
template <bool IFC>
__device__ __forceinline__ void select_body(const KParams& P, const SlabView* __restrict__ slabs, int nslabs, int L,
                                            int PB, const BlockEnt* __restrict__ blocks, StepState* ss,
                                            const BatchCfg& cfg, const double* __restrict__ u_pick, double r_direct,
                                            const double* __restrict__ ktab_g, cetkmc_event* my_event, int info_only, int ifc_ready,
                                            long long cur_hint = -1, SelCarry* carry = nullptr,
                                            const double* __restrict__ u_defect = nullptr, const double* __restrict__ u_np = nullptr)
{
    __shared__ double leafval[PMAX];           // the voxel tree's leaves (the chosen one is read back)
    __shared__ int rowcnt_l[PMAX], voxcnt_l[PMAX];     // event counts of the chosen block's rows / the chosen row's voxels (deposition rank)
    __shared__ double ktab[225];
    __shared__ long long red[4];
    __shared__ TreeScratch X;
    __shared__ double sh_u0, sh_base, sh_r;
    __shared__ int sh_go, sh_b, sh_slab;
    __shared__ int sh_gi0[64], sh_nloc[64];
    __shared__ long long sh_ndep;
    const int tid = threadIdx.x;
    SEL_STAMP(0);
    // everything thread 0 will need is requested now, behind the block loads: the batch status, the step's uniforms, the
    // stream cursor and the slabs' plane ranges; the first slab's view (the only one in most runs) comes with them
    double u0 = 0.0, u_def = 0.0, u_th = 0.0, u_ph = 0.0;
    long long np_pos0 = 0, cur = 0, nuc0 = 0;
    int status0 = 0;
    double margin0 = 1.0;
    if (tid == 0 && cfg.batch) {
        status0 = ss->status;
        margin0 = ss->min_margin;
        np_pos0 = ss->np_pos;
        nuc0 = ss->nuc_count;
        cur = cur_hint >= 0 ? cur_hint : ss->cur;
        if (cfg.rng_mode == 2) {            // the single-domain super-step's uniforms (oracle orc_run_supersteps, box == L)
            const uint64_t g = (uint64_t)(cfg.step0 + cur);
            u0 = counter_uniform(cfg.seed, g, KEY_PICK);
            if (carry) {
                if (cfg.defect_fraction > 0.0) u_def = counter_uniform(cfg.seed, g, KEY_DEFECT);
                u_th = counter_uniform(cfg.seed, g, KEY_THETA); u_ph = counter_uniform(cfg.seed, g, KEY_PHI);
            }
        } else {
            u0 = u_pick[cur];
            if (carry && u_defect && cfg.defect_fraction > 0.0) u_def = u_defect[cur];
            if (carry && u_np && cfg.rng_mode == 1 && np_pos0 + 2 <= cfg.np_cap) { u_th = u_np[np_pos0]; u_ph = u_np[np_pos0 + 1]; }
        }
    }
    if (tid == 0 && carry) carry->ready = 0;
    const SlabView S0 = slabs[0];
    if (tid < nslabs && tid < 64) { sh_gi0[tid] = slabs[tid].gi0; sh_nloc[tid] = slabs[tid].nloc; }
    const int NBk = 3 * L;
    // ---- (1)+(2) blocks: thread t owns blocks [t*npb, (t+1)*npb)
    const int npb = PB >= 256 ? (PB >> 8) : 1;
    double lf[8];
    unsigned fm = 0u;
    long long csum = 0;
#pragma unroll
    for (int u = 0; u < 8; ++u) {
        lf[u] = 0.0;
        const int idx = tid * npb + u;
        if (u < npb && idx < NBk) {
            lf[u] = blocks[idx].sum;
            const long long c = blocks[idx].cnt;
            csum += c;
            if (c > 0) fm |= 1u << u;
            if (idx == 3 * (L - 1) + CAT_DEP) sh_ndep = c;
        }
    }
    if (tid < 225) ktab[tid] = ktab_g[tid];
#pragma unroll
    for (int l = 0; l < 6; ++l) csum += __shfl_xor(csum, 1 << l);       // per-wave event count
    if ((tid & 63) == 0) red[tid >> 6] = csum;
    if (tid == 0) sh_u0 = u0;
    SEL_STAMP(1);
    const TreeSel ta = tree_select(lf, fm, npb, 0.0, [&](double total) { return cfg.batch ? sh_u0 * total : r_direct; }, X, tid);
    SEL_STAMP(2);
    if (tid == 0 && status0) sh_go = 0;                 // terminated / exhausted batch: nothing is read or written
    else if (tid == 0) {
        const double total = ta.total;
        const long long n_events = red[0] + red[1] + red[2] + red[3];
        const long long n_dep = sh_ndep;
        ss->total = total; ss->n_events = n_events; ss->n_dep = n_dep;
        int go = info_only ? 0 : 1;
        if (!info_only) my_event->type = -1;
        // reference stream: get_event_rates has drawn one species uniform per deposition candidate BEFORE run_kmc looks
        // at the total (kmc_event_rates.py:65 precedes kmc_simulation.py:259-262), also on the step that terminates
        const long long dep_draws = (cfg.batch && cfg.rng_mode == 0) ? n_dep : 0;
        if (cfg.batch && go && np_pos0 + dep_draws > cfg.np_cap) {
            ss->status = 2; go = 0;
        } else if (n_events == 0 || total < 1e-25 || !finite_d(total)) {
            if (cfg.batch) { ss->status = 1; ss->np_pos = np_pos0 + dep_draws; }
            go = 0;
        } else if (cfg.batch && go) {
            if (np_pos0 + dep_draws + 2 > cfg.np_cap) { ss->status = 2; go = 0; }
        }
        if (go) {
            const double r = cfg.batch ? u0 * total : r_direct;
            const int b = ta.leaf;
            const int i = b / 3;
            int sl = -1;
            for (int s = 0; s < nslabs; ++s) {
                const int g0 = s < 64 ? sh_gi0[s] : slabs[s].gi0, nl = s < 64 ? sh_nloc[s] : slabs[s].nloc;
                if (i >= g0 && i < g0 + nl) sl = s;
            }
            sh_b = b; sh_slab = sl; sh_base = ta.base; sh_r = r;
            if (sl < 0) go = 0;     // owned by another rank
            if (go && carry) {
                // reference stream: the orientation draws follow this sweep's n_dep species draws
                if (u_np && cfg.rng_mode == 0) { u_th = u_np[np_pos0 + n_dep]; u_ph = u_np[np_pos0 + n_dep + 1]; }
                carry->total = total; carry->n_events = n_events; carry->n_dep = n_dep; carry->np_pos = np_pos0; carry->cur = cur;
                carry->u_def = u_def; carry->u_th = u_th; carry->u_ph = u_ph; carry->nuc_count = nuc0;
                carry->ready = 1;
            }
        }
        sh_go = go;
    }
    __syncthreads();
    SEL_STAMP(3);
    if (!sh_go) return;
    // a deposition / nucleation will write the orientation (pi u_th, 2 pi u_ph) (kmc_simulation.py:283-284,308-309): its unit
    // vector (two sincos) is worked out now by a thread of the otherwise idle second wave, same expressions as the application
    if (carry && tid == 64)
        orient_vec(0.0 + (3.141592653589793 - 0.0) * carry->u_th, 0.0 + (6.283185307179586 - 0.0) * carry->u_ph, carry->ov);
    const SlabView S = (sh_slab == 0) ? S0 : slabs[sh_slab];
    const int b = sh_b, i = b / 3, c = b - 3 * i;
    const int lp = i - S.gi0, li = lp + 2;
    const double r = sh_r;
    const int Pk = S.Pk;
    const int npk = Pk >= 256 ? (Pk >> 8) : 1;
    // ---- (3) rows of block (i, c): thread t owns rows [t*npk, (t+1)*npk)
    fm = 0u;
#pragma unroll
    for (int u = 0; u < 8; ++u) {
        lf[u] = 0.0;
        const int idx = tid * npk + u;
        if (u < npk && idx < Pk) {
            int cv = 0;
            if (idx < L) { lf[u] = S.rowsum[((int64_t)lp * 3 + c) * L + idx]; cv = S.rowcnt[((int64_t)lp * 3 + c) * L + idx]; }
            if (cv > 0) fm |= 1u << u;
            rowcnt_l[idx] = cv;
        }
    }
    SEL_STAMP(4);
    const TreeSel tb = tree_select(lf, fm, npk, sh_base, [&](double) { return r; }, X, tid);
    const int j = tb.leaf;
    SEL_STAMP(5);
    // ---- (4) voxels of row (i, c, j).  With the rate table (ifc_ready): a leaf is a lookup -- listed voxels hold their full
    // EMPTY/DIFF category sum and count (k_interface / ifc_touch; every interface voxel is listed), every other empty
    // voxel its nucleation rate by temperature (k_rate_table), plane L-1 its deposition rates (dep_val).  Without it
    // (simple kernel): re-evaluated here.
    fm = 0u;
#pragma unroll
    for (int u = 0; u < 8; ++u) {
        lf[u] = 0.0;
        const int k = tid * npk + u;
        if (u < npk && k < Pk) {
            double sum = 0.0; int cnt = 0;
            int maybe_ifc = 1;      // 0: certainly no interface voxel (its EMPTY category can only hold a nucleation)
            if (k < L) {
                const int64_t t = S.tidx(li, j, k);
                if (IFC) {
                    // table entry and class byte (one round trip): the byte says whether the voxel is empty (bit 0) and how
                    // many events a LISTED voxel owns (bits 7:2, ifc_store(); zero for every voxel that is not listed --
                    // and a listed voxel without events holds a zero table entry, so it reads like an unlisted one)
                    const double v_tab = S.vval[t];
                    const unsigned cb = S.cls[S.cidx(li, j, k)];
                    const int c_ifc = (int)(cb >> 2);
                    const bool empty = (cb & 1u) != 0u;
                    maybe_ifc = c_ifc > 0 ? 1 : 0;
                    if (c == CAT_DEP) {
                        const double rate = S.dep_val[(int64_t)j * S.pitchT + k];
                        if (empty && finite_d(rate)) { sum = rate; cnt = 1; }
                    } else if (c == CAT_EMPTY) {
                        if (empty) { sum = v_tab; cnt = c_ifc > 0 ? c_ifc : ((v_tab != 0.0) ? 1 : 0); }
                    } else if (!empty && c_ifc > 0) {
                        sum = v_tab; cnt = c_ifc;
                    }
                } else {
                    const int st = S.state[S.sidx(li, j, k)];
                    const double Traw = S.T[t];
                    auto nb = [&](int mm) -> int { return S.state[S.sidx(li + nbi_rt(mm), j + nbj_rt(mm), k + nbk_rt(mm))]; };
                    auto emit = [&](int cat, int, double rate, int, int) { if (cat == c) { sum += rate; ++cnt; } };
                    eval_voxel(P, S, ktab, li, i, j, k, st, Traw, nb, emit);
                }
            }
            lf[u] = sum;
            if (cnt > 0) fm |= 1u << u;
            leafval[k] = sum;
            voxcnt_l[k] = cnt | (maybe_ifc << 8);
        }
    }
    SEL_STAMP(6);
    const TreeSel tc = tree_select(lf, fm, npk, tb.base, [&](double) { return r; }, X, tid);
    SEL_STAMP(7);
    if (tid == 0) {
        const int k = tc.leaf;
        const double base = tc.base;
        long long rank = 0;
        if (c == CAT_DEP) {
            for (int jj = 0; jj < j; ++jj) rank += rowcnt_l[jj];
            for (int kk = 0; kk < k; ++kk) rank += voxcnt_l[kk] & 255;
        }
        // slot scan (kmc_simulation.py:268-274 restricted to this voxel's slots)
        int p_type = -1, p_m = -1, p_atom = 0;
        double p_rate = 0.0, p_cum = 0.0;          // p_cum: running sum including the chosen event
        const int lc = voxcnt_l[k];
        if ((lc & 255) == 1 && (c == CAT_DEP || (c == CAT_EMPTY && !(lc >> 8)))) {
            // a single event whose kind is known without looking again: the deposition of this voxel, or the nucleation
            // of an empty voxel without W/Re/C neighbours; its rate is the leaf itself
            p_type = (c == CAT_DEP) ? EV_DEP : EV_NUC;
            p_atom = (c == CAT_DEP) ? 0 : 1;
            p_rate = leafval[k];
            p_cum = base + p_rate;
        } else {
            const int st = S.state[S.sidx(li, j, k)];
            double cum = base;
            bool found = false;
            auto nb = [&](int mm) -> int { return S.state[S.sidx(li + nbi_rt(mm), j + nbj_rt(mm), k + nbk_rt(mm))]; };
            auto emit = [&](int cat, int type, double rate, int m, int atom) {
                if (cat != c || found) return;
                cum += rate;
                p_type = type; p_m = m; p_atom = atom; p_rate = rate;   // remembers the last valid slot
                p_cum = cum;
                if (cum >= r) found = true;
            };
            eval_voxel(P, S, ktab, li, i, j, k, st, S.T[S.tidx(li, j, k)], nb, emit);
        }
        if (cfg.batch) {
            // Selection margin (hard part 3): r lies in the chosen event's interval (p_cum - rate, p_cum] of
            // the canonical cumulative sum; the reference scans a sequentially rounded sum (kmc_simulation.py:265-274) that
            // differs by ~1e-13 relative, so a pick closer than that to either end of the interval could be the neighbouring
            // event there.  The batch reports the smallest such distance relative to the total.
            const double lo = r - (p_cum - p_rate), hi = p_cum - r;
            const double m = (fabs(lo) < fabs(hi) ? fabs(lo) : fabs(hi)) / ta.total;
            if (m < margin0) ss->min_margin = m;
        }
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
    SEL_STAMP(8);
}

__global__ __launch_bounds__(256) void k_select(KParams P, const SlabView* __restrict__ slabs, int nslabs, int L,
                                                int PB, const BlockEnt* __restrict__ blocks, StepState* ss,
                                                BatchCfg cfg, const double* __restrict__ u_pick, double r_direct,
                                                const double* __restrict__ ktab_g, cetkmc_event* my_event, int info_only, int ifc_ready)
{
    // (two instantiations: the table path carries none of the per-voxel evaluation code of the simple-kernel path)
    if (ifc_ready) select_body<true>(P, slabs, nslabs, L, PB, blocks, ss, cfg, u_pick, r_direct, ktab_g, my_event, info_only, 1);
    else select_body<false>(P, slabs, nslabs, L, PB, blocks, ss, cfg, u_pick, r_direct, ktab_g, my_event, info_only, 0);
}

// ---- interface voxels --------------------------------------------------------------------
// A voxel "owns interface events" iff it is empty with a W/Re/C neighbour (attachment) or a
// W/Re/C atom with an empty neighbour (diffusion).  These are rare (a few per lattice row) and
// expensive, so they are kept in a per-slab list (append-only superset, rebuilt at upload),
// evaluated one voxel per lane by k_interface into vval / the class byte's count, and merely
// looked up by the streaming sweep kernel.
// Packed neighbourhood of a voxel: bits [2m+1:2m] describe neighbour slot m, bits [29:28] the voxel.
//   empty voxel (own = 0): slot = species of a W/Re/C neighbour (1,2,3), 0 otherwise
//   atom voxel  (own = 1,2,3 = species): slot = 1 empty, 2 occupied (any non-empty in-lattice state), 0 outside
// k_interface evaluates a listed voxel from this word alone (no neighbour-state gathers).
__device__ __forceinline__ unsigned ifc_encode(const SlabView& S, int li, int j, int k, bool* interface_out)
{
    // all 15 states are requested at once (the padding makes every address valid), then decoded
    const int st = S.state[S.sidx(li, j, k)];
    int sm[14];
#pragma unroll
    for (int m = 0; m < 14; ++m) sm[m] = S.state[S.sidx(li + nbi_rt(m), j + nbj_rt(m), k + nbk_rt(m))];
    unsigned code = 0;
    bool hit = false;
    if (st < 128 && st != 4) {
        code = (unsigned)(st & 3) << 28;
#pragma unroll
        for (int m = 0; m < 14; ++m) {
            unsigned c;
            if (st == 0) { c = (sm[m] >= 1 && sm[m] <= 3) ? (unsigned)sm[m] : 0u; hit |= (c != 0u); }
            else { c = (sm[m] == 0) ? 1u : ((sm[m] != OOB) ? 2u : 0u); hit |= (sm[m] == 0); }
            code |= c << (2 * m);
        }
    } else {
        code = 3u << 30;                        // defect / outside: no events
    }
    *interface_out = hit;
    return code;
}
__device__ __forceinline__ void ifc_append(const SlabView& S, int lp, int j, int k)
{
    const int64_t t = S.tidx(lp + 2, j, k);
    // test-and-set on the byte through its 32-bit word (two lanes of an event may touch the same voxel); callers have
    // already seen the flag clear
    unsigned* w = reinterpret_cast<unsigned*>(S.ifc_in + (t & ~(int64_t)3));
    const unsigned bit = 1u << (8 * (int)(t & 3));
    if (atomicOr(w, bit) & bit) return;
    const int pos = atomicAdd(S.ifc_n, 1);
    S.ifc_list[pos] = ((unsigned)lp << 20) | ((unsigned)j << 10) | (unsigned)k;
}
// full rebuild (upload): membership flags and neighbourhood words of all owned voxels; the list itself is then
// written in address order by k_ifc_relist (one atomic per 32 rows instead of one per entry)
__global__ void k_ifc_rebuild(SlabView S)
{
    const int L = S.L;
    const int64_t n = (int64_t)S.nloc * L * L;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < n; idx += (int64_t)gridDim.x * blockDim.x) {
        const int k = (int)(idx % L);
        const int64_t t = idx / L;
        const int j = (int)(t % L), lp = (int)(t / L);
        bool hit;
        const unsigned code = ifc_encode(S, lp + 2, j, k, &hit);
        if (hit) { const int64_t t = S.tidx(lp + 2, j, k); S.ifc_in[t] = 1; S.ifc_code[t] = code; }     // list: k_ifc_relist
    }
}
// EMPTY-category sum of a listed empty voxel (nuc + attachments) / DIFF-category sum of a listed atom, from its
// packed neighbourhood word alone
template <int BATCH = 4>
__device__ __forceinline__ void ifc_eval_empty(const KParams& P, const SlabView& S, const double* ktab, int lp, int j, int k,
                                               int64_t t, unsigned code, double Tc, double& sum, int& cnt)
{
    const int li = lp + 2;
    // in-lattice neighbour count from the coordinates (kmc_event_rates.py:32,37)
    const int i = S.gi0 + lp, L = S.L;
    int n_nb = 0, n_imp = 0;
    unsigned mask = 0;
#pragma unroll
    for (int m = 0; m < 14; ++m) {
        const int ni = i + nbi_rt(m), nj = j + nbj_rt(m), nk = k + nbk_rt(m);
        n_nb += (ni >= 0 && ni < L && nj >= 0 && nj < L && nk >= 0 && nk < L);
        const unsigned c = (code >> (2 * m)) & 3u;
        n_imp += (c >= 2u);
        if (c) mask |= 1u << m;
    }
    const double dT = P.T_melt - Tc;
    if (dT > P.delta_T_c) {
        const double rate = nuc_rate(P, ktab[n_nb * 15 + n_imp], dT, P.kT * Tc);
        if (rate > P.rate_threshold && finite_d(rate)) { sum = rate; cnt = 1; }
    }
    if (mask) {
        const AttCtx c = att_ctx(P, S, li, j, k, Tc);
        while (mask) {                                   // batches of BATCH attachment sources: one memory round trip each
            int ms[BATCH];
            double b[BATCH][3];
#pragma unroll
            for (int u = 0; u < BATCH; ++u) {
                ms[u] = mask ? __builtin_ctz(mask) : -1;
                mask &= mask - 1;
                const int m = ms[u] < 0 ? 0 : ms[u];
                const double* p = S.ovec + 3 * (ms[u] < 0 ? t : S.tidx(li + nbi_rt(m), j + nbj_rt(m), k + nbk_rt(m)));
                b[u][0] = p[0]; b[u][1] = p[1]; b[u][2] = p[2];
            }
#pragma unroll
            for (int u = 0; u < BATCH; ++u) {
                if (ms[u] >= 0) {
                    const int sn = (int)((code >> (2 * ms[u])) & 3u);
                    const double rate = att_item(P, c, b[u][0], b[u][1], b[u][2], sn);
                    if (rate > P.rate_threshold && finite_d(rate)) { sum = sum + rate; ++cnt; }
                }
            }
        }
    }
}
__device__ __forceinline__ void ifc_eval_atom(const KParams& P, const SlabView& S, int lp, int j, int k, int64_t t,
                                              unsigned code, int st, double Tc, double& sum, int& cnt)
{
    const int li = lp + 2;
    int n_bonds = 0;
    unsigned mask = 0;
#pragma unroll
    for (int m = 0; m < 14; ++m) {
        const unsigned c = (code >> (2 * m)) & 3u;
        n_bonds += (c == 2u);
        if (c == 1u) mask |= 1u << m;
    }
    if (mask) {
        const DiffCtx c = diff_ctx(P, S, li, j, k, st, n_bonds, Tc);
        double Tn[14];
#pragma unroll
        for (int m = 0; m < 14; ++m)                       // all neighbour temperatures in one round trip
            Tn[m] = S.T[(mask >> m) & 1u ? S.tidx(li + nbi_rt(m), j + nbj_rt(m), k + nbk_rt(m)) : t];
#pragma unroll
        for (int m = 0; m < 14; ++m) {
            if ((mask >> m) & 1u) {
                const double rate = diff_item(P, c, Tn[m]);
                if (rate > P.rate_threshold && finite_d(rate)) { sum = sum + rate; ++cnt; }
            }
        }
    }
}

// result of a listed voxel's evaluation: table entry, event count, and the count mirrored into bits 7:2 of the voxel's
// class byte (the sweep reads it there, from LDS, no separate count array); bits 1:0 = class8(state), which the
// packed neighbourhood word carries (write_site keeps cls and state level), so the byte is written without being read
__device__ __forceinline__ int code_state(unsigned code) { return (code >> 30) ? 4 : (int)((code >> 28) & 3u); }
__device__ __forceinline__ void ifc_store(const SlabView& S, int li, int j, int k, int64_t t, double sum, int cnt, unsigned code)
{
    S.vval[t] = sum;
    S.cls[S.cidx(li, j, k)] = (uint8_t)(class8(code_state(code)) | ((unsigned)cnt << 2));
}

// after an event changed voxel (i,j,k): it and its 14 neighbours may have become interface voxels,
// and the category sums of those already listed are stale (their neighbour states / orientations
// changed).  Called by a full wave: lane l < 15 handles one of the 15 voxels: append if needed, then
// re-evaluate if listed -- this keeps vval / the count exact when k_interface ran BEFORE the event
// (the speculative, overlapped launch of the batched loop).
// dedupe: the wave touches TWO neighbourhoods (a diffusion: site and target), so two lanes may hold the same voxel and
// the list append needs the atomic test-and-set; otherwise the <= 15 voxels are distinct and the appends of the wave
// share one atomic, whose result is only consumed after the evaluation.
// defer_centre (single neighbourhood only): the centre voxel (lane 14), if it is an atom, is NOT evaluated here:
// centre_atom_eval() spreads its <= 14 diffusion items over the lanes of a wave (another wave of the block, where
// there is one, so that it runs beside this function instead of after it).
__device__ __forceinline__ void ifc_touch(const KParams& P, const SlabView& S, const double* ktab, int i, int j, int k, int lane,
                                          int eval, bool dedupe, bool defer_centre)
{
    if (lane >= 15) return;
    int ai = i, aj = j, ak = k;
    if (lane < 14) { ai += nbi_rt(lane); aj += nbj_rt(lane); ak += nbk_rt(lane); }
    const int L = S.L;
    if (ai < 0 || ai >= L || aj < 0 || aj >= L || ak < 0 || ak >= L) return;
    const int lp = ai - S.gi0;
    if (lp < 0 || lp >= S.nloc) return;                      // owned planes only
    const int li = lp + 2;
    const int64_t t = S.tidx(li, aj, ak);
    // one memory round trip: membership flag and temperature go out together with the 15 states of ifc_encode
    bool listed = S.ifc_in[t] != 0;
    const double Traw = eval ? S.T[t] : 0.0;
    bool hit;
    const unsigned code = ifc_encode(S, li, aj, ak, &hit);
    SEL_STAMP(12);
    bool want = false;
    unsigned long long m = 0ull;
    int base = 0, leader = 0;
    if (dedupe) {
        if (hit && !listed) { ifc_append(S, lp, aj, ak); listed = true; }
    } else {
        want = hit && !listed;
        m = __ballot(want);                                          // the lanes still here (same slab, inside the lattice)
        if (m) {
            leader = __builtin_ctzll(m);
            if (lane == leader) base = atomicAdd(S.ifc_n, __popcll(m));      // consumed at the end of the function
            if (want) { S.ifc_in[t] = 1; listed = true; }
        }
    }
    if (listed) S.ifc_code[t] = code;
    SEL_STAMP(13);
    if (eval && listed) {            // the same evaluation from the packed word as k_interface
        const int st = code_state(code);
        const double Tc = pymax(Traw, 1.0);
        if (!(defer_centre && lane == 14 && st >= 1 && st <= 3)) {
            double sum = 0.0;
            int cnt = 0;
            if (st == 0) ifc_eval_empty<8>(P, S, ktab, lp, aj, ak, t, code, Tc, sum, cnt);     // one wave: registers are free
            else if (st != 4) ifc_eval_atom(P, S, lp, aj, ak, t, code, st, Tc, sum, cnt);
            ifc_store(S, li, aj, ak, t, sum, cnt, code);
        }
    }
    SEL_STAMP(14);
    if (m) {
        base = __shfl(base, leader);
        if (want) S.ifc_list[base + __popcll(m & ((1ull << lane) - 1ull))] = ((unsigned)lp << 20) | ((unsigned)aj << 10) | (unsigned)ak;
    }
}
// every step: EMPTY/DIFF category sum + count of every listed voxel, one voxel per lane.
// Latency-bound gather kernel: all loads of a phase are issued together (own fields + 14
// neighbour states, then 7 neighbours' vectors / temperatures at a time from clamped-safe
// addresses), so a voxel costs ~3 memory round trips instead of one per event.
__global__ __launch_bounds__(256) void k_interface(KParams P, SlabView S, const double* __restrict__ ktab_g,
                                                   const StepState* __restrict__ ss)
{
    if (ss && ss->status) return;
    __shared__ double ktab[225];
    for (int t = threadIdx.x; t < 225; t += blockDim.x) ktab[t] = ktab_g[t];
    __syncthreads();
    const int n = *S.ifc_n;
    for (int q = blockIdx.x * blockDim.x + threadIdx.x; q < n; q += gridDim.x * blockDim.x) {
        const unsigned v = S.ifc_list[q];
        const int lp = v >> 20, j = (v >> 10) & 1023, k = v & 1023;
        const int64_t t = S.tidx(lp + 2, j, k);
        const unsigned code = S.ifc_code[t];
        const int st = (code >> 30) ? 4 : (int)((code >> 28) & 3u);     // 4: no events
        const double Tc = pymax(S.T[t], 1.0);
        double sum = 0.0;
        int cnt = 0;
        if (st == 0) ifc_eval_empty(P, S, ktab, lp, j, k, t, code, Tc, sum, cnt);
        else if (st != 4) ifc_eval_atom(P, S, lp, j, k, t, code, st, Tc, sum, cnt);
        ifc_store(S, lp + 2, j, k, t, sum, cnt, code);
    }
}

// Same results for long lists (Mode B, where most of the lattice becomes interface): a block takes tiles of
// 1024 entries, sorts them by kind (empty / atom) into two LDS queues and evaluates each queue with full
// waves, so that a wave runs only one of the two (expensive, differently shaped) rate loops.
template <int IFC_TILE>
__global__ __launch_bounds__(256) void k_interface_part(KParams P, SlabView S, const double* __restrict__ ktab_g,
                                                        const StepState* __restrict__ ss)
{
    if (ss && ss->status) return;
    __shared__ double ktab[225];
    __shared__ unsigned qe[IFC_TILE], qa[IFC_TILE];
    __shared__ int ne, na;
    const int tid = threadIdx.x;
    for (int t = tid; t < 225; t += 256) ktab[t] = ktab_g[t];
    const int n = *S.ifc_n;
    for (int tile = blockIdx.x * IFC_TILE; tile < n; tile += gridDim.x * IFC_TILE) {
        __syncthreads();
        if (tid == 0) { ne = 0; na = 0; }
        __syncthreads();
#pragma unroll
        for (int u = 0; u < IFC_TILE / 256; ++u) {
            const int q = tile + u * 256 + tid;
            if (q < n) {
                const unsigned v = S.ifc_list[q];
                const unsigned code = S.ifc_code[S.tidx((int)(v >> 20) + 2, (v >> 10) & 1023, v & 1023)];
                if (code >> 30) {                                       // no events: result is zero
                    const int64_t t = S.tidx((int)(v >> 20) + 2, (v >> 10) & 1023, v & 1023);
                    ifc_store(S, (int)(v >> 20) + 2, (v >> 10) & 1023, v & 1023, t, 0.0, 0, code);
                } else if (((code >> 28) & 3u) == 0) qe[atomicAdd(&ne, 1)] = v;
                else qa[atomicAdd(&na, 1)] = v;
            }
        }
        __syncthreads();
        const int ne_pad = (ne + 63) & ~63, work = ne_pad + na;
        for (int w = tid; w < work; w += 256) {
            const bool is_e = w < ne_pad;
            if (is_e && w >= ne) continue;
            const unsigned v = is_e ? qe[w] : qa[w - ne_pad];
            const int lp = v >> 20, j = (v >> 10) & 1023, k = v & 1023;
            const int64_t t = S.tidx(lp + 2, j, k);
            const unsigned code = S.ifc_code[t];
            const double Tc = pymax(S.T[t], 1.0);
            double sum = 0.0;
            int cnt = 0;
            if (is_e) ifc_eval_empty(P, S, ktab, lp, j, k, t, code, Tc, sum, cnt);
            else ifc_eval_atom(P, S, lp, j, k, t, code, (int)((code >> 28) & 3u), Tc, sum, cnt);
            ifc_store(S, lp + 2, j, k, t, sum, cnt, code);
        }
    }
}

// ---- apply -------------------------------------------------------------------------------
// ov: the unit vector of (th, ph) if the caller already holds it (same orient_vec() bits), else null
__device__ __forceinline__ void write_site(const SlabView& S, int i, int j, int k, int st, double th, double ph, const double* ov = nullptr)
{
    const int li = i - (S.gi0 - 2);
    if (li < 0 || li >= S.nloc + 4) return;
    S.state[S.sidx(li, j, k)] = (uint8_t)st;
    S.row_chg[(int64_t)li * S.L + j] = 1;
    S.cls[S.cidx(li, j, k)] = class8(st);
    const int64_t q = S.tidx(li, j, k);
    S.theta[q] = th; S.phi[q] = ph;
    double* o = S.ovec + 3 * q;
    if (ov) { o[0] = ov[0]; o[1] = ov[1]; o[2] = ov[2]; }
    else if (__double_as_longlong(th) == 0 && __double_as_longlong(ph) == 0) { o[0] = 0.0; o[1] = 0.0; o[2] = 1.0; }   // sin 0 = 0, cos 0 = 1
    else orient_vec(th, ph, o);
}
// kmc_simulation.py:276-327 on every local slab whose extended range holds the voxel(s)
// ov: unit vector of (ev.theta, ev.phi) if already known (the fused selection computes it for dep / nuc events)
__device__ __forceinline__ void apply_event(const SlabView* slabs, int nslabs, const cetkmc_event& ev, int make_defect, const double* ov = nullptr)
{
    for (int s = 0; s < nslabs; ++s) {
        const SlabView& S = slabs[s];
        int ui = ev.pos[0], uj = ev.pos[1], uk = ev.pos[2];
        if (ev.type == EV_DEP || ev.type == EV_NUC || ev.type == EV_ATT) {
            write_site(S, ui, uj, uk, ev.atom, ev.theta, ev.phi, ov);
        } else if (ev.type == EV_DIFF) {
            write_site(S, ev.target[0], ev.target[1], ev.target[2], ev.atom, ev.theta, ev.phi);
            write_site(S, ui, uj, uk, 0, 0.0, 0.0);
            ui = ev.target[0]; uj = ev.target[1]; uk = ev.target[2];   // :303
        }
        if (make_defect) write_site(S, ui, uj, uk, 4, 0.0, 0.0);       // :323-327
    }
}

// interface-list update for the voxels an event touched (all lanes of the first wave)
// kmc_event_rates.py:93-107 for the atom an event has just placed at (i,j,k) -- the common case of every dep / nuc /
// att event: its diffusion items one per lane (lane m = neighbour slot m), summed by lane 0 in slot order exactly like
// ifc_eval_atom() does serially.  Called by all 64 lanes of ONE wave; every lane derives the (identical) packed word,
// membership and context itself -- the same cache lines for all of them.  Counterpart of ifc_touch(defer_centre).
__device__ __forceinline__ void centre_atom_eval(const KParams& P, const SlabView& S, int i, int j, int k, int lane)
{
    const int L = S.L;
    const int lp = i - S.gi0;
    if (i < 0 || i >= L || j < 0 || j >= L || k < 0 || k >= L || lp < 0 || lp >= S.nloc) return;
    const int li = lp + 2;
    const int64_t t = S.tidx(li, j, k);
    const bool listed0 = S.ifc_in[t] != 0;
    const double Tc = pymax(S.T[t], 1.0);
    bool hit;
    const unsigned code = ifc_encode(S, li, j, k, &hit);
    const int st = code_state(code);
    if (!(listed0 || hit) || st < 1 || st > 3) return;               // not listed, or evaluated by ifc_touch
    int n_bonds = 0;
    unsigned mask = 0;
#pragma unroll
    for (int m = 0; m < 14; ++m) {
        const unsigned c = (code >> (2 * m)) & 3u;
        n_bonds += (c == 2u);
        if (c == 1u) mask |= 1u << m;
    }
    double rate = 0.0;
    int ok = 0;
    if (mask) {
        const bool mine_item = lane < 14 && ((mask >> lane) & 1u);
        const int m = lane < 14 ? lane : 0;
        const double Tn = S.T[mine_item ? S.tidx(li + nbi_rt(m), j + nbj_rt(m), k + nbk_rt(m)) : t];
        const DiffCtx c = diff_ctx(P, S, li, j, k, st, n_bonds, Tc);
        if (mine_item) {
            rate = diff_item(P, c, Tn);
            ok = (rate > P.rate_threshold && finite_d(rate)) ? 1 : 0;
        }
    }
    double sum = 0.0;
    int cnt = 0;
#pragma unroll
    for (int m = 0; m < 14; ++m) {
        const double r = __shfl(rate, m);
        const int o = __shfl(ok, m);
        if (o) { sum = sum + r; ++cnt; }
    }
    if (lane == 0) ifc_store(S, li, j, k, t, sum, cnt, code);
}
// tid / nthreads: the calling block's thread index and size (whole waves).  Wave 0: lanes 0..14 the neighbourhood of the
// event site, lanes 16..30 the neighbourhood of a diffusion target; the atom a dep / nuc / att event placed is
// evaluated by wave 1 (wave 0 in a one-wave block).
__device__ __forceinline__ void apply_touch(const KParams& P, const SlabView* slabs, int nslabs, const double* ktab,
                                            const cetkmc_event& ev, int tid, int eval, int nthreads)
{
    const int wave = tid >> 6, lane = tid & 63;
    const bool two = ev.type == EV_DIFF;
    const bool defer = !two && eval;
    const int cw = nthreads >= 128 ? 1 : 0;
    for (int s = 0; s < nslabs; ++s) {
        const SlabView& S = slabs[s];
        if (wave == 0) {
            if (lane < 16) ifc_touch(P, S, ktab, ev.pos[0], ev.pos[1], ev.pos[2], lane, eval, two, defer);
            else if (two) ifc_touch(P, S, ktab, ev.target[0], ev.target[1], ev.target[2], lane - 16, eval, two, false);
        }
        if (defer && wave == cw) centre_atom_eval(P, S, ev.pos[0], ev.pos[1], ev.pos[2], lane);
    }
}

// Batched apply: RNG bookkeeping of one step + lattice update + per-step logs (lane 0), then the
// interface-list update for the touched voxels (whole wave).  Launched with ONE 64-thread block.
__device__ __forceinline__ void apply_batch_body(const KParams& P, const SlabView* __restrict__ slabs, int nslabs, int L,
                                                 const cetkmc_event* events_all, int G, StepState* ss,
                                                 const BatchCfg& cfg, const double* __restrict__ u_defect,
                                                 const double* __restrict__ u_np, double* log_total,
                                                 cetkmc_event* log_event, int64_t* log_nev,
                                                 const double* __restrict__ ktab_g, int eval_touched, int* dirty,
                                                 const SelCarry* carry = nullptr)
{
    __shared__ cetkmc_event sh_ev;
    __shared__ int sh_ok;
    SEL_STAMP(9);
    if (threadIdx.x == 0) {
        sh_ok = 0;
        const bool have = carry && carry->ready;            // fused launch: the selection succeeded in this block
        if (have || (!carry && !ss->status)) {
            cetkmc_event ev;
            ev.type = -1;
            for (int g = 0; g < G; ++g) if (events_all[g].type >= 0) ev = events_all[g];
            const int64_t s = have ? carry->cur : ss->cur;
            if (ev.type < 0) {
                ss->status = 1;
            } else {
                int64_t pos = have ? carry->np_pos : ss->np_pos;
                const int64_t n_dep = have ? carry->n_dep : ss->n_dep;
                if (ev.type == EV_DEP) {
                    const double u = (cfg.rng_mode == 0)
                        ? u_np[pos + ev.dep_rank]
                        : counter_uniform(cfg.seed, (uint64_t)(cfg.step0 + s), (uint64_t)ev.pos[1] * (uint64_t)L + (uint64_t)ev.pos[2]);
                    ev.atom = dep_species(P, u);
                }
                if (cfg.rng_mode == 0) pos += n_dep;
                if (ev.type == EV_DEP || ev.type == EV_NUC) {
                    const uint64_t gs = (uint64_t)(cfg.step0 + s);
                    const double ut = have ? carry->u_th : (cfg.rng_mode == 2 ? counter_uniform(cfg.seed, gs, KEY_THETA) : u_np[pos]);
                    const double up = have ? carry->u_ph : (cfg.rng_mode == 2 ? counter_uniform(cfg.seed, gs, KEY_PHI) : u_np[pos + 1]);
                    ev.theta = 0.0 + (3.141592653589793 - 0.0) * ut;       // np.random.uniform(0, pi)
                    ev.phi = 0.0 + (6.283185307179586 - 0.0) * up;         // np.random.uniform(0, 2*pi)
                    if (cfg.rng_mode != 2) pos += 2;
                    if (ev.type == EV_NUC) { if (have) ss->nuc_count = carry->nuc_count + 1; else ss->nuc_count += 1; }
                }
                const int mk = (cfg.defect_fraction > 0.0 &&
                                (have ? carry->u_def
                                      : (cfg.rng_mode == 2 ? counter_uniform(cfg.seed, (uint64_t)(cfg.step0 + s), KEY_DEFECT) : u_defect[s])) < cfg.defect_fraction) ? 1 : 0;
                apply_event(slabs, nslabs, ev, mk, (have && (ev.type == EV_DEP || ev.type == EV_NUC)) ? carry->ov : nullptr);
                ss->np_pos = pos;
                if (log_total) log_total[s] = have ? carry->total : ss->total;
                if (log_event) log_event[s] = ev;
                if (log_nev) log_nev[s] = have ? carry->n_events : ss->n_events;
                ss->cur = s + 1;
                sh_ev = ev;
                sh_ok = 1;
            }
        }
    }
    SEL_STAMP(10);
    __syncthreads();
    if (sh_ok) apply_touch(P, slabs, nslabs, ktab_g, sh_ev, threadIdx.x, eval_touched, blockDim.x);
    SEL_STAMP(11);
    if (dirty && threadIdx.x == 0) {
        // rows whose rates may have changed: the rows of the changed voxel(s) and of their 14 neighbours
        int n = 0;
        if (sh_ok) {
            const int di[11] = {0, 1, 1, -1, -1, 0, 0, 2, -2, 0, 0}, dj[11] = {0, 1, -1, 1, -1, 1, -1, 0, 0, 2, -2};
            for (int v = 0; v < (sh_ev.type == EV_DIFF ? 2 : 1); ++v) {
                const int ci = v ? sh_ev.target[0] : sh_ev.pos[0], cj = v ? sh_ev.target[1] : sh_ev.pos[1];
                for (int q = 0; q < 11; ++q) {
                    const int i = ci + di[q], j = cj + dj[q];
                    if (i < 0 || i >= L || j < 0 || j >= L) continue;
                    const int e = (i << 16) | j;
                    bool dup = false;
                    for (int t = 1; t <= n; ++t) dup |= (dirty[t] == e);
                    if (!dup && n < DIRTY_MAX) dirty[++n] = e;
                }
            }
        }
        dirty[0] = n;
    }
}

__global__ __launch_bounds__(64) void k_apply_batch(KParams P, const SlabView* __restrict__ slabs, int nslabs, int L,
                                                    const cetkmc_event* __restrict__ events_all, int G, StepState* ss,
                                                    BatchCfg cfg, const double* __restrict__ u_defect,
                                                    const double* __restrict__ u_np, double* log_total,
                                                    cetkmc_event* log_event, int64_t* log_nev,
                                                    const double* __restrict__ ktab_g, int eval_touched, int* dirty)
{
    apply_batch_body(P, slabs, nslabs, L, events_all, G, ss, cfg, u_defect, u_np, log_total, log_event, log_nev, ktab_g,
                     eval_touched, dirty);
}
// Single-process batched loop: selection and application in one launch (the event record never leaves the block's
// view of memory; threads >= 64 only take part in the barriers of the apply part).
template <bool IFC>
__global__ __launch_bounds__(256) void k_select_apply(KParams P, const SlabView* __restrict__ slabs, int nslabs, int L,
                                                      int PB, const BlockEnt* __restrict__ blocks, StepState* ss,
                                                      BatchCfg cfg, const double* __restrict__ u_pick,
                                                      const double* __restrict__ ktab_g, cetkmc_event* my_event, int ifc_ready,
                                                      const double* __restrict__ u_defect, const double* __restrict__ u_np,
                                                      double* log_total, cetkmc_event* log_event, int64_t* log_nev,
                                                      int eval_touched, int* dirty, long long cur_hint)
{
    __shared__ cetkmc_event sh_sel;            // the chosen event goes from the selection to the application through LDS,
    __shared__ SelCarry sh_carry;              // and so do the step state and the uniforms the selection requested early
    (void)my_event;
    if (threadIdx.x == 0) sh_sel.type = -1;
    __syncthreads();
    select_body<IFC>(P, slabs, nslabs, L, PB, blocks, ss, cfg, u_pick, 0.0, ktab_g, &sh_sel, 0, ifc_ready, cur_hint, &sh_carry, u_defect, u_np);
    __syncthreads();
    apply_batch_body(P, slabs, nslabs, L, &sh_sel, 1, ss, cfg, u_defect, u_np, log_total, log_event, log_nev, ktab_g,
                     eval_touched, dirty, &sh_carry);
}

// Direct apply (cetkmc_apply): everything decided by the host.  ONE 64-thread block.
__global__ __launch_bounds__(64) void k_apply_direct(KParams P, const SlabView* __restrict__ slabs, int nslabs, cetkmc_event ev,
                                                     int make_defect, StepState* ss, const double* __restrict__ ktab_g)
{
    if (threadIdx.x == 0) {
        if (ev.type == EV_NUC) ss->nuc_count += 1;
        apply_event(slabs, nslabs, ev, make_defect);
    }
    __syncthreads();
    apply_touch(P, slabs, nslabs, ktab_g, ev, threadIdx.x, 0, blockDim.x);
}

__global__ void k_empty() {}

// start of a batch: the batch part of the step state (nucleation_count persists) -- on the stream, no host round trip
__global__ void k_batch_reset(StepState* ss)
{
    ss->cur = 0; ss->status = 0; ss->np_pos = 0; ss->min_margin = 1.0;
}

// ---- thermal -------------------------------------------------------------------------------
struct ThermalCfg {
    double dt, alpha, inv_dx2, clip_lo, clip_hi, T_nan, rho_cp, latent_coef;
    int laser, use_latent, scrub, ni;
};
// the rate table's inputs, for the temperature kernel that writes the table entry of every voxel it updates (k_thermal_tiles16<.., TABLE>)
struct TableCfg {
    double T_melt, delta_T_c, kT, I0, rate_threshold, K0, nu_dep;
    double* vval;       // the NEW field's table (the buffer pair is flipped with T)
    double* dep_val;
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

// after a latent-heat temperature update prev_state equals state again: drop the row flags (not when the batch has
// already terminated -- the update was then a pass-through and prev_state was left alone)
__global__ void k_clear_row_flags(SlabView S, const StepState* __restrict__ ss)
{
    if (ss && ss->status) return;
    const int64_t n = (int64_t)(S.nloc + 4) * S.L;
    for (int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; q < n; q += (int64_t)gridDim.x * blockDim.x) S.row_chg[q] = 0;
}

// thermal_solver.py:36-105 for ONE voxel (same expression order as k_thermal / k_thermal_march), with the latent-heat
// indicator dF01 (0 or 1) given by the caller
__device__ __forceinline__ double thermal_voxel(const SlabView& S, const double* __restrict__ Tin, const double* __restrict__ q_top,
                                                const ThermalCfg& C, int li, int i, int j, int k, double dF01)
{
    const int L = S.L;
    const int im = (i > 0 ? i - 1 : 0) - (S.gi0 - 2), ip = (i < L - 1 ? i + 1 : L - 1) - (S.gi0 - 2);
    const int jm = j > 0 ? j - 1 : 0, jp = j < L - 1 ? j + 1 : L - 1;
    const int km = k > 0 ? k - 1 : 0, kp = k < L - 1 ? k + 1 : L - 1;
    const double tc = scrub_T(Tin[S.tidx(li, j, k)], C.T_nan, C.scrub);
    const double d0 = tc * -2.0 + (scrub_T(Tin[S.tidx(im, j, k)], C.T_nan, C.scrub) + scrub_T(Tin[S.tidx(ip, j, k)], C.T_nan, C.scrub));
    const double d1 = tc * -2.0 + (scrub_T(Tin[S.tidx(li, jm, k)], C.T_nan, C.scrub) + scrub_T(Tin[S.tidx(li, jp, k)], C.T_nan, C.scrub));
    const double d2 = tc * -2.0 + (scrub_T(Tin[S.tidx(li, j, km)], C.T_nan, C.scrub) + scrub_T(Tin[S.tidx(li, j, kp)], C.T_nan, C.scrub));
    const double lap = ((d0 + d1) + d2) * C.inv_dx2;
    const double qv = (i == L - 1) ? q_top[(int64_t)j * L + k] : 0.0;
    const double qterm = (qv != 0.0) ? qv / C.rho_cp : 0.0;
    const double dtm = C.dt > 1e-12 ? C.dt : 1e-12;
    const double dF = (dF01 != 0.0) ? 1.0 / dtm : 0.0;
    const double dTdt = C.alpha * lap + qterm + C.latent_coef * dF;
    const double nt = tc + C.dt * dTdt;
    double v = nt < C.clip_lo ? C.clip_lo : nt;
    v = v > C.clip_hi ? C.clip_hi : v;
    return v;
}

// Look-ahead temperature update (laser mode with the latent-heat term): T(n+1) and its rate table were computed ahead of
// time on a second stream WITHOUT the latent-heat term, which only differs at voxels that turned from empty to occupied
// since the previous update (thermal_solver.py:98-99) -- a handful per 20 steps, all in rows flagged by write_site().
// This kernel, launched at the update itself, recomputes exactly those voxels (temperature, table entry, deposition
// rate) from the old field, brings prev_state level with state on the flagged rows and clears the flags.  A batch that
// has terminated leaves T alone (kmc_simulation.py:260-262 breaks before the next update): the old field, table and
// deposition rates are copied over the speculative ones.
__global__ __launch_bounds__(256) void k_thermal_fix(KParams P, SlabView S, const double* __restrict__ Tin, double* __restrict__ Tout,
                                                     uint8_t* __restrict__ prev_state, const double* __restrict__ q_top, ThermalCfg C,
                                                     const StepState* __restrict__ ss, const double* __restrict__ vv_old,
                                                     double* __restrict__ vv_new, const double* __restrict__ dep_old,
                                                     double* __restrict__ dep_new, double K0)
{
    const int L = S.L;
    const int64_t gtid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x, nthr = (int64_t)gridDim.x * blockDim.x;
    if (ss && ss->status) {
        const int64_t nT = (int64_t)(S.nloc + 4) * L * S.pitchT;
        for (int64_t q = gtid; q < nT; q += nthr) { Tout[q] = Tin[q]; vv_new[q] = vv_old[q]; }
        for (int64_t q = gtid; q < (int64_t)L * S.pitchT; q += nthr) dep_new[q] = dep_old[q];
        return;
    }
    if (!(C.laser && C.use_latent)) return;                // no latent-heat term: the look-ahead field is already exact
    const int lp_top = L - 1 - S.gi0;
    // one wave per row (the flag test is wave-uniform), lanes over k
    const int lane = threadIdx.x & 63;
    const int64_t wave = gtid >> 6, nwaves = nthr >> 6;
    for (int64_t r = wave; r < (int64_t)S.nloc * L; r += nwaves) {
        const int lp = (int)(r / L), j = (int)(r - (int64_t)lp * L), li = lp + 2;
        if (!S.row_chg[(int64_t)li * L + j]) continue;
        for (int k = lane; k < L; k += 64) {
            const int64_t sc = S.sidx(li, j, k);
            const int st = S.state[sc], pv = prev_state[sc];
            if (pv == 0 && st != 0) {
                const double v = thermal_voxel(S, Tin, q_top, C, li, S.gi0 + lp, j, k, 1.0);
                const int64_t c = S.tidx(li, j, k);
                Tout[c] = v;
                vv_new[c] = nuc_bulk(P.T_melt, P.delta_T_c, P.kT, P.I0, P.rate_threshold, K0, v);
                if (lp == lp_top) dep_new[(int64_t)j * S.pitchT + k] = dep_rate(P, pymax(v, 1.0));
            }
            if (pv != st) prev_state[sc] = (uint8_t)st;
        }
    }
    // every flag of the slab (halo rows included) is dropped by k_clear_row_flags, launched behind this kernel
}

// k_thermal_march: same arithmetic as k_thermal, 2.5-D blocked.  One block owns THERM_TJ rows x 256
// columns and marches over THERM_NI planes: the planes i-1, i, i+1 of its own voxels live in registers
// (8 voxels per thread), plane i additionally in an LDS tile with a one-voxel rim for the j+-1 / k+-1
// neighbours.  Every T value is read ~1.4x and written once (k_thermal: 7 reads through L2).
constexpr int THERM_TJ = 8, THERM_NI = 4, THERM_KT = 256;
__global__ __launch_bounds__(256) void k_thermal_march(SlabView S, const double* __restrict__ Tin, double* __restrict__ Tout,
                                                       uint8_t* __restrict__ prev_state, const double* __restrict__ q_top,
                                                       ThermalCfg C, const StepState* __restrict__ ss)
{
    constexpr int TJ = THERM_TJ, KT = THERM_KT, LW = KT + 2;
    __shared__ double tile[(TJ + 2) * LW];
    const int L = S.L;
    const int tid = threadIdx.x;
    const int kc = blockIdx.x * KT, j0 = blockIdx.y * TJ;
    const int lp0 = blockIdx.z * C.ni, lp1 = min(lp0 + C.ni, S.nloc);
    const bool passthrough = ss && ss->status;
    const int col = 2 * (tid & 127), rbase = tid >> 7;          // thread: columns kc+col, kc+col+1 of rows rbase+2q
    const int k0 = kc + col;
    auto ldT = [&](int li, int j, int k) { return scrub_T(Tin[S.tidx(li, j, k)], C.T_nan, C.scrub && !passthrough); };
    auto lplane = [&](int i) { return (i < 0 ? 0 : (i > L - 1 ? L - 1 : i)) - (S.gi0 - 2); };   // clamped global plane -> local
    double prv[4][2], cur[4][2], nxt[4][2];
    auto load_own = [&](int li, double (&dst)[4][2]) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int j = j0 + rbase + 2 * q;
            dst[q][0] = dst[q][1] = 0.0;
            if (j < L && k0 < L) {
                const double2 v = *reinterpret_cast<const double2*>(Tin + S.tidx(li, j, k0));
                dst[q][0] = scrub_T(v.x, C.T_nan, C.scrub && !passthrough);
                dst[q][1] = scrub_T(v.y, C.T_nan, C.scrub && !passthrough);
            }
        }
    };
    load_own(lplane(S.gi0 + lp0 - 1), prv);
    load_own(lp0 + 2, cur);
    const double dtm = C.dt > 1e-12 ? C.dt : 1e-12;
#pragma unroll 1
    for (int lp = lp0; lp < lp1; ++lp) {
        const int li = lp + 2, i = S.gi0 + lp;
        load_own(lplane(i + 1), nxt);
        // plane i into LDS: own values, then the rim (rows j0-1 / j0+TJ, columns kc-1 / kc+KT), edge-replicated
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            double* row = tile + (rbase + 2 * q + 1) * LW + 1 + col;
            row[0] = cur[q][0]; row[1] = cur[q][1];
        }
        for (int e = tid; e < 2 * KT + 2 * (TJ + 2); e += 256) {
            int tr, tc;                                          // tile coordinates of a rim cell
            if (e < 2 * KT) { tr = (e < KT) ? 0 : TJ + 1; tc = 1 + (e % KT); }
            else { const int f = e - 2 * KT; tr = f >> 1; tc = (f & 1) ? KT + 1 : 0; }
            int j = j0 + tr - 1, kk = kc + tc - 1;
            j = j < 0 ? 0 : (j > L - 1 ? L - 1 : j);
            kk = kk < 0 ? 0 : (kk > L - 1 ? L - 1 : kk);
            tile[tr * LW + tc] = ldT(li, j, kk);
        }
        __syncthreads();
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int r = rbase + 2 * q, j = j0 + r;
            // latent-heat term: states of the thread's two columns now and at the previous update, one 16-bit load each
            // (k0 is even, rows are padded); prev_state of the next update is written back the same way (:100).  Only rows
            // written since the last update can differ from prev_state (row_chg, set by write_site)
            unsigned st2 = 0, pv2 = 0;
            if (C.laser && C.use_latent && !passthrough && j < L && k0 < L && S.row_chg[(int64_t)li * L + j]) {
                const int64_t sc = S.sidx(li, j, k0);
                st2 = *reinterpret_cast<const uint16_t*>(S.state + sc);
                pv2 = *reinterpret_cast<const uint16_t*>(prev_state + sc);
                *reinterpret_cast<uint16_t*>(prev_state + sc) = (uint16_t)st2;
            }
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int k = k0 + h;
                if (j < L && k < L) {
                    const int64_t c = S.tidx(li, j, k);
                    if (passthrough) { Tout[c] = Tin[c]; continue; }
                    const double tc = cur[q][h];
                    // edge replication inside the lattice: the rim already holds clamped values, but an
                    // interior tile edge that coincides with the lattice edge must replicate too
                    const double* cell = tile + (r + 1) * LW + 1 + col + h;
                    const double jm = (j > 0) ? cell[-LW] : tc, jp = (j < L - 1) ? cell[LW] : tc;
                    const double km = (k > 0) ? cell[-1] : tc, kp = (k < L - 1) ? cell[1] : tc;
                    const double d0 = tc * -2.0 + (prv[q][h] + nxt[q][h]);
                    const double d1 = tc * -2.0 + (jm + jp);
                    const double d2 = tc * -2.0 + (km + kp);
                    const double lap = ((d0 + d1) + d2) * C.inv_dx2;
                    double nt;
                    if (!C.laser) {
                        nt = tc + (C.dt * C.alpha) * lap;
                    } else {
                        // q/(rho cp) and dF/dt are zero for almost every voxel: 0/x is +0 exactly, so the two fp64
                        // divisions are only performed where the numerator is not zero (same bits as thermal_solver.py:99-103)
                        const double qv = (i == L - 1) ? q_top[(int64_t)j * L + k] : 0.0;
                        const double qterm = (qv != 0.0) ? qv / C.rho_cp : 0.0;
                        double dF = 0.0;
                        if (C.use_latent && ((pv2 >> (8 * h)) & 255u) == 0 && ((st2 >> (8 * h)) & 255u) != 0) dF = 1.0 / dtm;
                        const double dTdt = C.alpha * lap + qterm + C.latent_coef * dF;
                        nt = tc + C.dt * dTdt;
                    }
                    double v = nt < C.clip_lo ? C.clip_lo : nt;
                    v = v > C.clip_hi ? C.clip_hi : v;
                    Tout[c] = v;
                }
            }
        }
        __syncthreads();
#pragma unroll
        for (int q = 0; q < 4; ++q) { prv[q][0] = cur[q][0]; prv[q][1] = cur[q][1]; cur[q][0] = nxt[q][0]; cur[q][1] = nxt[q][1]; }
    }
}

// k_thermal_tiles: k_thermal_march for lattices that the tiles cover exactly (L a multiple of THERM_TJ and of THERM_KT:
// 256, 512, ...), with everything a partial tile needs taken out of the instruction stream: no per-voxel range tests, no
// edge selects (the rim is loaded from clamped coordinates, which IS the edge replication: a tile edge is a lattice edge or
// an interior boundary, never in between), row offsets computed once, the k-neighbours inside the thread's own column pair
// taken from registers.  Same loads, same expression order, same bits (test_thermal_kernel_variants_identical).
#ifndef CETKMC_THERM_ATTR
#define CETKMC_THERM_ATTR __attribute__((amdgpu_waves_per_eu(4, 4)))   // latent-heat instantiation: 150 VGPRs otherwise (3 waves/SIMD); 40 B of scratch at 4
#endif
template <bool LASER, bool LATENT>
__global__ __launch_bounds__(256) CETKMC_THERM_ATTR void k_thermal_tiles(SlabView S, const double* __restrict__ Tin, double* __restrict__ Tout,
                                                       uint8_t* __restrict__ prev_state, const double* __restrict__ q_top,
                                                       ThermalCfg C, const StepState* __restrict__ ss)
{
    constexpr int TJ = THERM_TJ, KT = THERM_KT, LW = KT + 2;
    __shared__ double tile[(TJ + 2) * LW];
    const int L = S.L, pitchT = S.pitchT;
    const int tid = threadIdx.x;
    const int kc = blockIdx.x * KT, j0 = blockIdx.y * TJ;
    const int lp0 = blockIdx.z * C.ni, lp1 = min(lp0 + C.ni, S.nloc);
    const int col = 2 * (tid & 127), rbase = tid >> 7;          // thread: columns kc+col, kc+col+1 of rows rbase+2q
    const int k0 = kc + col;
    const int64_t pstride = (int64_t)L * pitchT;
    int off[4];                                                 // in-plane offsets of the thread's four column pairs
#pragma unroll
    for (int q = 0; q < 4; ++q) off[q] = (j0 + rbase + 2 * q) * pitchT + k0;
    if (ss && ss->status) {                                     // terminated batch: the field is passed through unchanged
        for (int lp = lp0; lp < lp1; ++lp)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int64_t c = (int64_t)(lp + 2) * pstride + off[q];
                *reinterpret_cast<double2*>(Tout + c) = *reinterpret_cast<const double2*>(Tin + c);
            }
        return;
    }
    const int scrub = C.scrub;
    auto lplane = [&](int i) { return (i < 0 ? 0 : (i > L - 1 ? L - 1 : i)) - (S.gi0 - 2); };   // clamped global plane -> local
    // rim cells of a plane: thread t loads (row j0-1, column kc+t) and (row j0+TJ, column kc+t); threads < 2 (TJ+2) the two
    // side columns -- all from clamped coordinates
    const int rim_top = max(j0 - 1, 0) * pitchT + kc + tid, rim_bot = min(j0 + TJ, L - 1) * pitchT + kc + tid;
    const int side_row = tid >> 1, side_right = tid & 1;
    const int side_off = min(max(j0 + side_row - 1, 0), L - 1) * pitchT + (side_right ? min(kc + KT, L - 1) : max(kc - 1, 0));
    double prv[4][2], cur[4][2], nxt[4][2];
    auto load_own = [&](int li, double (&dst)[4][2]) {
        const double* base = Tin + (int64_t)li * pstride;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const double2 v = *reinterpret_cast<const double2*>(base + off[q]);
            dst[q][0] = scrub_T(v.x, C.T_nan, scrub);
            dst[q][1] = scrub_T(v.y, C.T_nan, scrub);
        }
    };
    load_own(lplane(S.gi0 + lp0 - 1), prv);
    load_own(lp0 + 2, cur);
    // rim of the first plane; every later plane's rim is requested one plane ahead, like its own values
    double rt, rb, rs = 0.0;
    {
        const double* plane = Tin + (int64_t)(lp0 + 2) * pstride;
        rt = plane[rim_top]; rb = plane[rim_bot];
        if (tid < 2 * (TJ + 2)) rs = plane[side_off];
    }
    const double dtm = C.dt > 1e-12 ? C.dt : 1e-12;
    const double dt_alpha = C.dt * C.alpha;
#pragma unroll 1
    for (int lp = lp0; lp < lp1; ++lp) {
        const int li = lp + 2, i = S.gi0 + lp;
        // next plane's rim (after the block's last plane: the same plane again, cache hits) and own values are requested
        // now and used in the next iteration; plane i itself goes to LDS from registers
        const double* plane_n = Tin + (int64_t)min(li + 1, lp1 + 1) * pstride;
        const double rt_n = plane_n[rim_top], rb_n = plane_n[rim_bot];
        double rs_n = 0.0;
        if (tid < 2 * (TJ + 2)) rs_n = plane_n[side_off];
        load_own(lplane(i + 1), nxt);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            double* row = tile + (rbase + 2 * q + 1) * LW + 1 + col;
            row[0] = cur[q][0]; row[1] = cur[q][1];
        }
        tile[1 + tid] = scrub_T(rt, C.T_nan, scrub);
        tile[(TJ + 1) * LW + 1 + tid] = scrub_T(rb, C.T_nan, scrub);
        if (tid < 2 * (TJ + 2)) tile[side_row * LW + (side_right ? KT + 1 : 0)] = scrub_T(rs, C.T_nan, scrub);
        __syncthreads();
        const bool top = LASER && (i == L - 1);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int r = rbase + 2 * q, j = j0 + r;
            unsigned st2 = 0, pv2 = 0;
            if (LATENT && S.row_chg[(int64_t)li * L + j]) {      // wave-uniform (a wave's rows differ by q only)
                const int64_t sc = S.sidx(li, j, k0);
                st2 = *reinterpret_cast<const uint16_t*>(S.state + sc);
                pv2 = *reinterpret_cast<const uint16_t*>(prev_state + sc);
                *reinterpret_cast<uint16_t*>(prev_state + sc) = (uint16_t)st2;
            }
            const double* cell = tile + (r + 1) * LW + 1 + col;
            const double jm0 = cell[-LW], jm1 = cell[-LW + 1], jp0 = cell[LW], jp1 = cell[LW + 1];
            const double km0 = cell[-1], kp1 = cell[2];
            double out[2];
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const double tc = cur[q][h];
                const double jm = h ? jm1 : jm0, jp = h ? jp1 : jp0;
                const double km = h ? cur[q][0] : km0, kp = h ? kp1 : cur[q][1];
                const double d0 = tc * -2.0 + (prv[q][h] + nxt[q][h]);
                const double d1 = tc * -2.0 + (jm + jp);
                const double d2 = tc * -2.0 + (km + kp);
                const double lap = ((d0 + d1) + d2) * C.inv_dx2;
                double nt;
                if (!LASER) {
                    nt = tc + dt_alpha * lap;
                } else {
                    // q/(rho cp) and dF/dt are zero for almost every voxel: 0/x is +0 exactly, so the two fp64 divisions are
                    // only performed where the numerator is not zero (same bits as thermal_solver.py:99-103)
                    double qterm = 0.0;
                    if (top) {
                        const double qv = q_top[(int64_t)j * L + k0 + h];
                        qterm = (qv != 0.0) ? qv / C.rho_cp : 0.0;
                    }
                    double dF = 0.0;
                    if (LATENT && ((pv2 >> (8 * h)) & 255u) == 0 && ((st2 >> (8 * h)) & 255u) != 0) dF = 1.0 / dtm;
                    const double dTdt = C.alpha * lap + qterm + C.latent_coef * dF;
                    nt = tc + C.dt * dTdt;
                }
                double v = nt < C.clip_lo ? C.clip_lo : nt;
                out[h] = v > C.clip_hi ? C.clip_hi : v;
            }
            *reinterpret_cast<double2*>(Tout + (int64_t)li * pstride + off[q]) = make_double2(out[0], out[1]);
        }
        __syncthreads();
#pragma unroll
        for (int q = 0; q < 4; ++q) { prv[q][0] = cur[q][0]; prv[q][1] = cur[q][1]; cur[q][0] = nxt[q][0]; cur[q][1] = nxt[q][1]; }
        rt = rt_n; rb = rb_n; rs = rs_n;
    }
}

// k_thermal_tiles16: the same update with 16-row x 256-column tiles and 512 threads per block (thread: two columns of the
// rows rbase + 4 q, q = 0..3), marching over C.ni planes (default 16).  Against k_thermal_tiles (8 rows, 4 planes):
//   * the rim of a plane's tile is 2 rows + 2 columns for 16 rows (13 % extra reads instead of 26 %), and the two planes a
//     march needs beyond its own are amortised over 16 planes instead of 4 (12.5 % instead of 50 %): T is read ~1.28 x and
//     written once -- 18.2 B per voxel, 1.14 x the algorithmic 16 B (k_thermal_tiles: 23 B, 1.44 x);
//   * a plane's own values are requested TWO planes ahead (registers prv | cur | nxt | nx2), its rim one plane ahead: the
//     stencil of plane i needs plane i+1's values, so with a one-plane distance the loads issued at the top of an iteration
//     were awaited a few instructions later; now a whole iteration's arithmetic and barriers lie in between.
// Same loads per voxel, same expression order, same bits (test_thermal_kernel_variants_identical).
// MEASURED in the real loop (laser + latent heat, 256^3, rocprofv3): with FOUR rows per thread (512 threads,
// 128 VGPRs, 60 B of scratch in the latent instantiation) 84.6 us at 8 planes per block and 105 at 16 against 79.7 for the 8-row
// k_thermal_tiles -- fewer bytes, but half the waves; with TWO rows per thread (1024 threads: half the plane registers, 96 VGPRs,
// no scratch, one block = 16 waves per CU) 71.0 us at 16 planes per block: the default (thermal_variant 1).
constexpr int THERM16_TJ = 16, THERM16_NI = 16;
#ifndef CETKMC_THERM16_ATTR
#define CETKMC_THERM16_ATTR __attribute__((amdgpu_waves_per_eu(4, 4)))
#endif
// RPT = rows per thread (4 or 2: half the plane registers per thread), KT = columns per tile (256 or 128); threads per block =
// (KT / 2) * (16 / RPT): 512 (4, 256), 1024 (2, 256) or 512 (2, 128: two blocks per CU at <= 128 VGPRs)
// TABLE: the kernel also writes what k_rate_table would compute from the new field -- the rate table entry of every voxel it
// updates (and plane L-1's deposition rates) -- while the new temperature is still in a register: the field is not read a
// second time (16 of the pair's 32 B per voxel saved).  Same functions on the same values: same bits.
template <bool LASER, bool LATENT, int RPT, int KT, bool TABLE = false>
__global__ __launch_bounds__((KT / 2) * (16 / RPT)) CETKMC_THERM16_ATTR void k_thermal_tiles16(SlabView S, const double* __restrict__ Tin, double* __restrict__ Tout,
                                                         uint8_t* __restrict__ prev_state, const double* __restrict__ q_top,
                                                         ThermalCfg C, const StepState* __restrict__ ss, TableCfg TB = TableCfg{})
{
    constexpr int TJ = THERM16_TJ, LW = KT + 2, HC = KT / 2, NTHR = HC * (16 / RPT);
    static_assert(NTHR >= 2 * KT, "one rim cell per thread");
    __shared__ double tile[(TJ + 2) * LW];
    const int L = S.L, pitchT = S.pitchT;
    const int tid = threadIdx.x;
    const int kc = blockIdx.x * KT, j0 = blockIdx.y * TJ;
    const int lp0 = blockIdx.z * C.ni, lp1 = min(lp0 + C.ni, S.nloc);
    constexpr int RS = TJ / RPT;                                // row stride of a thread's rows (= tid >> 7 range)
    const int col = 2 * (tid % HC), rbase = tid / HC;           // thread: columns kc+col, kc+col+1 of rows rbase + RS q
    const int k0 = kc + col;
    const int64_t pstride = (int64_t)L * pitchT;
    int off[RPT];                                               // in-plane offsets of the thread's column pairs
#pragma unroll
    for (int q = 0; q < RPT; ++q) off[q] = (j0 + rbase + RS * q) * pitchT + k0;
    if (ss && ss->status) {                                     // terminated batch: the field is passed through unchanged
        for (int lp = lp0; lp < lp1; ++lp)
#pragma unroll
            for (int q = 0; q < RPT; ++q) {
                const int64_t c = (int64_t)(lp + 2) * pstride + off[q];
                *reinterpret_cast<double2*>(Tout + c) = *reinterpret_cast<const double2*>(Tin + c);
            }
        return;
    }
    const int scrub = C.scrub;
    auto lplane = [&](int i) { return (i < 0 ? 0 : (i > L - 1 ? L - 1 : i)) - (S.gi0 - 2); };   // clamped global plane -> local
    // rim cells of a plane, one load per thread: threads 0..255 the row above the tile (j0-1), 256..511 the row below (j0+TJ);
    // 2 (TJ+2) threads (the first ones; with 1024 threads: 512..547) one cell of the two side columns -- all from clamped
    // coordinates (= edge replication)
    const int rim_col = tid % KT, rim_low = (tid / KT) & 1;
    const bool has_rim = tid < 2 * KT;
    const int rim_off = (rim_low ? min(j0 + TJ, L - 1) : max(j0 - 1, 0)) * pitchT + kc + rim_col;
    const int stid = (NTHR >= 2 * KT + 2 * (TJ + 2)) ? tid - 2 * KT : tid;
    const int side_row = stid >> 1, side_right = stid & 1;
    const bool has_side = stid >= 0 && stid < 2 * (TJ + 2);
    const int side_off = min(max(j0 + side_row - 1, 0), L - 1) * pitchT + (side_right ? min(kc + KT, L - 1) : max(kc - 1, 0));
    double prv[RPT][2], cur[RPT][2], nxt[RPT][2], nx2[RPT][2];
    auto load_own = [&](int li, double (&dst)[RPT][2]) {
        const double* base = Tin + (int64_t)li * pstride;
#pragma unroll
        for (int q = 0; q < RPT; ++q) {
            const double2 v = *reinterpret_cast<const double2*>(base + off[q]);
            dst[q][0] = scrub_T(v.x, C.T_nan, scrub);
            dst[q][1] = scrub_T(v.y, C.T_nan, scrub);
        }
    };
    load_own(lplane(S.gi0 + lp0 - 1), prv);
    load_own(lp0 + 2, cur);
    load_own(lplane(S.gi0 + lp0 + 1), nxt);
    double rr = 0.0, rs = 0.0;
    {
        const double* plane = Tin + (int64_t)(lp0 + 2) * pstride;
        if (has_rim) rr = plane[rim_off];
        if (has_side) rs = plane[side_off];
    }
    const double dtm = C.dt > 1e-12 ? C.dt : 1e-12;
    const double dt_alpha = C.dt * C.alpha;
#pragma unroll 1
    for (int lp = lp0; lp < lp1; ++lp) {
        const int li = lp + 2, i = S.gi0 + lp;
        // requests of this iteration: plane i+2's own values (used two iterations on), plane i+1's rim (used in the next).
        // After the block's last planes the (unconditional) requests repeat a plane the block has just read: cache hits
        load_own(lplane(min(i + 2, S.gi0 + lp1)), nx2);
        const double* plane_n = Tin + (int64_t)min(li + 1, lp1 + 1) * pstride;
        double rr_n = 0.0, rs_n = 0.0;
        if (has_rim) rr_n = plane_n[rim_off];
        if (has_side) rs_n = plane_n[side_off];
#pragma unroll
        for (int q = 0; q < RPT; ++q) {
            double* row = tile + (rbase + RS * q + 1) * LW + 1 + col;
            row[0] = cur[q][0]; row[1] = cur[q][1];
        }
        if (has_rim) tile[(rim_low ? (TJ + 1) * LW : 0) + 1 + rim_col] = scrub_T(rr, C.T_nan, scrub);
        if (has_side) tile[side_row * LW + (side_right ? KT + 1 : 0)] = scrub_T(rs, C.T_nan, scrub);
        __syncthreads();
        const bool top = LASER && (i == L - 1);
#pragma unroll
        for (int q = 0; q < RPT; ++q) {
            const int r = rbase + RS * q, j = j0 + r;
            unsigned st2 = 0, pv2 = 0;
            if (LATENT && S.row_chg[(int64_t)li * L + j]) {      // wave-uniform (a wave's rows differ by q only)
                const int64_t sc = S.sidx(li, j, k0);
                st2 = *reinterpret_cast<const uint16_t*>(S.state + sc);
                pv2 = *reinterpret_cast<const uint16_t*>(prev_state + sc);
                *reinterpret_cast<uint16_t*>(prev_state + sc) = (uint16_t)st2;
            }
            const double* cell = tile + (r + 1) * LW + 1 + col;
            const double jm0 = cell[-LW], jm1 = cell[-LW + 1], jp0 = cell[LW], jp1 = cell[LW + 1];
            const double km0 = cell[-1], kp1 = cell[2];
            double out[2];
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const double tc = cur[q][h];
                const double jm = h ? jm1 : jm0, jp = h ? jp1 : jp0;
                const double km = h ? cur[q][0] : km0, kp = h ? kp1 : cur[q][1];
                const double d0 = tc * -2.0 + (prv[q][h] + nxt[q][h]);
                const double d1 = tc * -2.0 + (jm + jp);
                const double d2 = tc * -2.0 + (km + kp);
                const double lap = ((d0 + d1) + d2) * C.inv_dx2;
                double nt;
                if (!LASER) {
                    nt = tc + dt_alpha * lap;
                } else {
                    double qterm = 0.0;
                    if (top) {
                        const double qv = q_top[(int64_t)j * L + k0 + h];
                        qterm = (qv != 0.0) ? qv / C.rho_cp : 0.0;
                    }
                    double dF = 0.0;
                    if (LATENT && ((pv2 >> (8 * h)) & 255u) == 0 && ((st2 >> (8 * h)) & 255u) != 0) dF = 1.0 / dtm;
                    const double dTdt = C.alpha * lap + qterm + C.latent_coef * dF;
                    nt = tc + C.dt * dTdt;
                }
                double v = nt < C.clip_lo ? C.clip_lo : nt;
                out[h] = v > C.clip_hi ? C.clip_hi : v;
            }
            *reinterpret_cast<double2*>(Tout + (int64_t)li * pstride + off[q]) = make_double2(out[0], out[1]);
            if (TABLE) {
                const double r0 = nuc_bulk(TB.T_melt, TB.delta_T_c, TB.kT, TB.I0, TB.rate_threshold, TB.K0, out[0]);
                const double r1 = nuc_bulk(TB.T_melt, TB.delta_T_c, TB.kT, TB.I0, TB.rate_threshold, TB.K0, out[1]);
                *reinterpret_cast<double2*>(TB.vval + (int64_t)li * pstride + off[q]) = make_double2(r0, r1);
                if (i == L - 1) {                  // kmc_event_rates.py:59-63 for plane L-1 (block-uniform)
                    double* dp = TB.dep_val + (int64_t)j * pitchT + k0;
                    dp[0] = dep_rate_s(TB.nu_dep, TB.T_melt, TB.kT, pymax(out[0], 1.0));
                    dp[1] = dep_rate_s(TB.nu_dep, TB.T_melt, TB.kT, pymax(out[1], 1.0));
                }
            }
        }
        __syncthreads();
#pragma unroll
        for (int q = 0; q < RPT; ++q) {
            prv[q][0] = cur[q][0]; prv[q][1] = cur[q][1]; cur[q][0] = nxt[q][0]; cur[q][1] = nxt[q][1];
            nxt[q][0] = nx2[q][0]; nxt[q][1] = nx2[q][1];
        }
        rr = rr_n; rs = rs_n;
    }
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
        if (with_cls) S.cls[S.cidx(li, j, k)] = class8((int)(uint8_t)src[idx]);
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
