// superstep.hpp -- Mode B: synchronous super-steps over spatial boxes (several executed events per rate sweep).
//
// NOT in the reference, whose loop executes one event per full-lattice get_event_rates call
// (kmc_simulation.py:246-332).  Definition and its CPU comparator: DESIGN.md "Mode B" / oracle
// orc_run_supersteps.  Per super-step the ordinary sweep kernels run first (global total, interface sums);
// then every box picks one event from its active octant by the canonical tree restricted to that window,
// and all picked events are applied -- they cannot touch the same voxel by construction of the windows.
#pragma once
#include "kernels.hpp"

namespace cetkmc {

constexpr int SUPER_CNT_SLOTS = 256, SUPER_CNT_STRIDE = 16;     // per-slot {executed, nucleations} counters, 128 B apart

struct SuperCfg {
    int64_t step0;
    double defect_fraction;
    uint64_t seed;
    int32_t box, H, PH, PT;     // box edge, window edge (box/2), pow2(H), pow2(3H)
    int32_t nb;                 // boxes per axis
    int32_t d0, D_loc;          // this handle's boxes: global indices [d0, d0 + D_loc) (box layers of its owned planes)
    int32_t null_events;        // 1: a box executes its pick with probability R_d / R_max, else a null event (type -2)
    int32_t nranks;             // entries of the gathered per-rank maxima (1 in a single process)
};

// The slot scan of kmc_simulation.py:268-274 inside ONE chosen voxel by a whole wave: lane m < 14 evaluates the rate of
// neighbour slot m (attachment from a W/Re/C neighbour, or diffusion into an empty one), lane 14 the nucleation; lane 0
// then walks them in the reference's order (nucleation, then slots 0..13) with the running sum starting at `base`.
// Same arithmetic as eval_voxel() (att_ctx/att_item, diff_ctx/diff_item, nuc_rate with the K_eff table).  Result
// (lane 0): type, slot, species, rate of the first event with cum >= r, else of the last valid one.
struct SlotPick { int type, m, atom; double rate; };
__device__ __forceinline__ SlotPick slot_scan_wave(const KParams& P, const SlabView& S, const double* ktab, int li, int j, int k,
                                                   int c, double base, double r, int lane)
{
    const int64_t t = S.tidx(li, j, k);
    const int st = S.state[S.sidx(li, j, k)];
    const double Tc = pymax(S.T[t], 1.0);
    const int m = lane < 14 ? lane : 0;
    const int di = nbi_rt(m), dj = nbj_rt(m), dk = nbk_rt(m);
    const int sm = lane < 14 ? (int)S.state[S.sidx(li + di, j + dj, k + dk)] : (int)OOB;
    const unsigned long long M14 = 0x3FFFull;
    const int n_nb = __popcll(__ballot(sm != OOB) & M14);
    const int n_imp = __popcll(__ballot(sm == 2 || sm == 3) & M14);
    const int n_bonds = __popcll(__ballot(sm != 0 && sm != OOB) & M14);
    double rate = 0.0;
    int ok = 0, atom = 0;
    if (c == CAT_EMPTY && st == 0) {
        if (lane == 14) {
            const double dT = P.T_melt - Tc;
            if (dT > P.delta_T_c) {
                rate = nuc_rate(P, ktab[n_nb * 15 + n_imp], dT, P.kT * Tc);
                ok = (rate > P.rate_threshold && finite_d(rate)) ? 1 : 0;
                atom = 1;
            }
        } else if (lane < 14 && sm >= 1 && sm <= 3) {
            const AttCtx cx = att_ctx(P, S, li, j, k, Tc);
            const double* b = S.ovec + 3 * S.tidx(li + di, j + dj, k + dk);
            rate = att_item(P, cx, b[0], b[1], b[2], sm);
            ok = (rate > P.rate_threshold && finite_d(rate)) ? 1 : 0;
            atom = sm;
        }
    } else if (c == CAT_DIFF && st >= 1 && st <= 3) {
        if (lane < 14 && sm == 0) {
            const DiffCtx cx = diff_ctx(P, S, li, j, k, st, n_bonds, Tc);
            rate = diff_item(P, cx, S.T[S.tidx(li + di, j + dj, k + dk)]);
            ok = (rate > P.rate_threshold && finite_d(rate)) ? 1 : 0;
            atom = st;
        }
    }
    SlotPick pk;
    pk.type = -1; pk.m = -1; pk.atom = 0; pk.rate = 0.0;
    double cum = base;
    bool found = false;
#pragma unroll
    for (int q = 0; q < 15; ++q) {
        const int src = (q == 0) ? 14 : q - 1;                   // nucleation first, then the slots in order
        const double rq = __shfl(rate, src);
        const int oq = __shfl(ok, src), aq = __shfl(atom, src);
        if (oq && !found) {
            cum += rq;
            pk.type = (src == 14) ? EV_NUC : (c == CAT_DIFF ? EV_DIFF : EV_ATT);
            pk.m = (src == 14) ? -1 : src; pk.atom = aq; pk.rate = rq;      // remembers the last valid slot
            if (cum >= r) found = true;
        }
    }
    return pk;
}

struct DomPick {          // the chosen event of one box, before the uniforms are drawn
    int32_t i, j, k;       // chosen voxel
    int32_t type;          // EV_*; < 0: idle box
    int32_t m, atom;       // neighbour slot (or -1), species
    double rate;
    double R;              // total rate of the box's window (canonical window tree); 0 for an idle box
};

// One wave per box: leaves = category sums of the window's voxels, LDS heap of NL = PT*PH*PH leaves built by the wave
// (lane-local levels + shuffles, no block barriers), descent, slot scan inside the chosen voxel (slot_scan_wave).
// The uniforms and the lattice write follow in k_domain_apply, one THREAD per box (fused into this kernel's lane 0 they
// cost more: 32 768 waves each dragging a serial sincos / store tail, 97 vs 37 + 14 us).
// The per-voxel rate table (SlabView::vval) holds the EMPTY- or
// DIFF-category sum of EVERY owned voxel that owns events (listed voxels: interface sums + the count in the class byte; other empty voxels:
// the nucleation rate by temperature), so a leaf is a few loads (dep leaves: one exp, top plane only).
__global__ __launch_bounds__(64) void k_domain_pick(KParams P, const SlabView* __restrict__ slabs, int nslabs, int L,
                                                    SuperCfg C, const StepState* __restrict__ ss, const double* __restrict__ ktab_g,
                                                    DomPick* __restrict__ picks)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x;
    // boxes that are neighbours along k share cache lines (a window row is 32 B of a 128-B line): consecutive boxes go to
    // the same XCD (blocks are dealt round-robin over the 8 XCDs, each with its own L2)
    int dl = blockIdx.x;
    if ((gridDim.x & 7) == 0) dl = (dl & 7) * (int)(gridDim.x >> 3) + (dl >> 3);
    const int d = C.d0 + dl;                    // local / global box index (the global one keys the uniforms)
    if (ss->status) return;
    const int NL = C.PT * C.PH * C.PH;
    double* hs = reinterpret_cast<double*>(smem);             // [2*NL]
    uint8_t* hf = reinterpret_cast<uint8_t*>(hs + 2 * NL);    // [2*NL]
    uint8_t* lc = hf + 2 * NL;                                // [NL] leaf: event count | 128 if the voxel is a listed interface voxel
    for (int q = lane; q < NL; q += 64) { hs[NL + q] = 0.0; hf[NL + q] = 0; lc[q] = 0; }
    const int64_t cur = ss->cur;
    const int64_t g = C.step0 + cur;
    const int sec = (int)(g & 7);
    const int H = C.H, nb = C.nb;
    const int di = d / (nb * nb), dj = (d / nb) % nb, dk = d % nb;
    const int i0 = di * C.box + ((sec >> 2) & 1) * H, j0 = dj * C.box + ((sec >> 1) & 1) * H, k0 = dk * C.box + (sec & 1) * H;
    __builtin_amdgcn_wave_barrier();
    for (int v = lane; v < H * H * H; v += 64) {
        const int kk = v % H, jj = (v / H) % H, ii = v / (H * H);
        const int i = i0 + ii, j = j0 + jj, k = k0 + kk;
        int sl = 0;
        for (int s = 0; s < nslabs; ++s) if (i >= slabs[s].gi0 && i < slabs[s].gi0 + slabs[s].nloc) sl = s;
        const SlabView& S = slabs[sl];
        const int li = i - S.gi0 + 2;
        const int64_t t = S.tidx(li, j, k);
        // table entry + class byte: bit 0 empty, bit 1 W/Re/C atom (neither: defect / outside), bits 7:2 the event count of a
        // LISTED voxel (zero for every other voxel; a listed voxel without events holds a zero table entry)
        double val = S.vval[t];
        const unsigned cb = S.cls[S.cidx(li, j, k)];
        if ((cb & 3u) == 0u) continue;
        const bool empty = (cb & 1u) != 0u;
        const int listed = (cb >> 2) != 0u;
        int n_ev = (int)(cb >> 2);
        if (!listed) {          // no interface events: an empty voxel owns at most its nucleation (the table entry), an atom nothing
            if (empty) n_ev = (val != 0.0) ? 1 : 0;
            else { val = 0.0; n_ev = 0; }
        }
        const int c = empty ? CAT_EMPTY : CAT_DIFF;
        const int q = ((3 * ii + c) * C.PH + jj) * C.PH + kk;
        hs[NL + q] = val; hf[NL + q] = n_ev > 0; lc[q] = (uint8_t)(n_ev | (listed << 7));
        if (empty && i == L - 1) {
            const double rate = dep_rate(P, pymax(S.T[t], 1.0));
            if (finite_d(rate)) {
                const int qd = ((3 * ii + CAT_DEP) * C.PH + jj) * C.PH + kk;
                hs[NL + qd] = rate; hf[NL + qd] = 1; lc[qd] = 1;
            }
        }
    }
    __builtin_amdgcn_wave_barrier();
    // heap: the lane's NL/64 consecutive leaves fold locally, the six levels above by shuffles (a + b and b + a are the same
    // double, so both lanes of a pair hold the node's value).  One wave: LDS operations complete in program order.
    const int per = NL >> 6;                                   // >= 4
    for (int w = per >> 1, lvl = NL >> 1; w >= 1; w >>= 1, lvl >>= 1)
        for (int q = 0; q < w; ++q) {
            const int node = lvl + w * lane + q;
            hs[node] = hs[2 * node] + hs[2 * node + 1];
            hf[node] = hf[2 * node] | hf[2 * node + 1];
        }
    {
        double v = hs[64 + lane];
        int f = hf[64 + lane];
#pragma unroll
        for (int l = 0; l < 6; ++l) {
            v = v + __shfl_xor(v, 1 << l);
            f |= __shfl_xor(f, 1 << l);
            if ((lane & ((2 << l) - 1)) == 0) {
                const int node = (64 >> (l + 1)) + (lane >> (l + 1));
                hs[node] = v; hf[node] = (uint8_t)f;
            }
        }
    }
    __builtin_amdgcn_wave_barrier();
    // descent (every lane walks the same path: LDS broadcasts)
    int pi = 0, pj = 0, pkk = 0, cat = -1, simple = 0;
    double base = 0.0, r = 0.0, leaf_rate = 0.0;
    const double R = hs[1];
    if (hf[1] && !(R < 1e-25) && finite_d(R)) {
        r = counter_uniform(C.seed, (uint64_t)g, KEY_PICK | (uint64_t)d) * R;
        int n = 1;
        while (n < NL) {
            const int l = 2 * n;
            if (hf[l + 1] == 0 || (hf[l] != 0 && base + hs[l] >= r)) n = l;
            else { base += hs[l]; n = l + 1; }
        }
        const int q = n - NL;
        const int b = q / (C.PH * C.PH);
        pi = i0 + b / 3; cat = b % 3; pj = j0 + (q / C.PH) % C.PH; pkk = k0 + q % C.PH;
        leaf_rate = hs[NL + q];
        simple = (lc[q] == 1 && cat != CAT_DIFF) ? 1 : 0;       // one event, voxel not listed (or a deposition leaf)
    }
    DomPick pk;
    pk.i = pi; pk.j = pj; pk.k = pkk; pk.type = -1; pk.m = -1; pk.atom = 0; pk.rate = 0.0;
    pk.R = (cat >= 0) ? R : 0.0;
    if (cat >= 0) {                                              // wave-uniform
        if (simple) {            // the deposition of this voxel / the nucleation of a bulk empty voxel: nothing to scan
            pk.type = (cat == CAT_DEP) ? EV_DEP : EV_NUC;
            pk.atom = (cat == CAT_DEP) ? 0 : 1;
            pk.rate = leaf_rate;
        } else {
            int sl = 0;
            for (int s = 0; s < nslabs; ++s) if (pi >= slabs[s].gi0 && pi < slabs[s].gi0 + slabs[s].nloc) sl = s;
            const SlabView& S = slabs[sl];
            const SlotPick sp = slot_scan_wave(P, S, ktab_g, pi - S.gi0 + 2, pj, pkk, cat, base, r, lane);     // one table entry: from global
            pk.type = sp.type; pk.m = sp.m; pk.atom = sp.atom; pk.rate = sp.rate;
        }
    }
    if (lane == 0) picks[dl] = pk;
}

// The same pick for box = 8 (window 4 x 4 x 4, NL = 256 leaves) with the window tree in REGISTERS: lane = (block b = 3 ii + c,
// row jj) owns the row's four leaves kk = 0..3 -- the canonical tree's first two levels fold inside the lane, the six above
// by DPP / v_permlane*_swap with every level's block sum kept in the lane (as tree_select() does for Mode A), the descent
// reads them back with v_readlane.  No LDS heap, no index divisions; the three category lanes of a row issue the same
// row loads (32 B of the table, one word of class bytes).  Same leaves, pairs, descent rule and slot scan
// as k_domain_pick (tests: Mode B parity against the oracle runs box 8 through this kernel, the other boxes through that one).
__global__ __launch_bounds__(64) void k_domain_pick8(KParams P, const SlabView* __restrict__ slabs, int nslabs, int L,
                                                     SuperCfg C, const StepState* __restrict__ ss, const double* __restrict__ ktab_g,
                                                     DomPick* __restrict__ picks, long long cur_hint)
{
    const int lane = threadIdx.x;
    int dl = blockIdx.x;
    if ((gridDim.x & 7) == 0) dl = (dl & 7) * (int)(gridDim.x >> 3) + (dl >> 3);      // consecutive boxes on one XCD
    const int d = C.d0 + dl;
    // the batch status is only needed for the one store at the end (a terminated batch writes nothing; everything read
    // on the way is valid memory either way), the super-step index comes from the host (== ss->cur while status == 0):
    // no load stands between the launch and the window's loads
    const int status = ss->status;
    const int64_t g = C.step0 + cur_hint;
    const int sec = (int)(g & 7);
    const int nb = C.nb;
    const int di = d / (nb * nb), dj = (d / nb) % nb, dk = d % nb;
    const int i0 = di * 8 + ((sec >> 2) & 1) * 4, j0 = dj * 8 + ((sec >> 1) & 1) * 4, k0 = dk * 8 + (sec & 1) * 4;
    const int b = lane >> 2, jj = lane & 3;
    const int ii = (b * 11) >> 5, c = b - 3 * ii;            // b / 3, b % 3 for b < 16 (b >= 12: padding blocks, no leaves)
    double lf[4] = {0.0, 0.0, 0.0, 0.0};
    unsigned lcode = 0;                                       // byte kk: event count | 128 if the voxel is a listed interface voxel
    if (b < 12) {
        const int i = i0 + ii, j = j0 + jj;
        int sl = 0;
        for (int s = 1; s < nslabs; ++s) if (i >= slabs[s].gi0 && i < slabs[s].gi0 + slabs[s].nloc) sl = s;
        const SlabView& S = slabs[sl];
        const int li = i - S.gi0 + 2;
        const int64_t t = S.tidx(li, j, k0);
        // the row's four voxels: everything is requested together (k0 is a multiple of 4: the words are aligned)
        const unsigned cl4 = *reinterpret_cast<const unsigned*>(S.cls + S.cidx(li, j, k0));
        const double2 va = *reinterpret_cast<const double2*>(S.vval + t), vb = *reinterpret_cast<const double2*>(S.vval + t + 2);
        const double vv[4] = {va.x, va.y, vb.x, vb.y};
        const bool top = (i == L - 1);
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            // class byte: bit 0 empty, bit 1 W/Re/C atom (neither: defect / outside), bits 7:2 the event count of a LISTED
            // voxel (zero for every other voxel; a listed voxel without events holds a zero table entry)
            const unsigned cb = (cl4 >> (8 * kk)) & 255u;
            if ((cb & 3u) == 0u) continue;
            const bool empty = (cb & 1u) != 0u;
            if (c == CAT_DEP) {
                if (top && empty) {
                    const double rate = dep_rate(P, pymax(S.T[t + kk], 1.0));
                    if (finite_d(rate)) { lf[kk] = rate; lcode |= 1u << (8 * kk); }
                }
                continue;
            }
            const bool listed = (cb >> 2) != 0u;
            double val = vv[kk];
            int n_ev = (int)(cb >> 2);
            if (!listed) {      // an empty voxel owns at most its nucleation (the table entry), an atom nothing
                if (empty) n_ev = (val != 0.0) ? 1 : 0;
                else { val = 0.0; n_ev = 0; }
            }
            if (c == (empty ? CAT_EMPTY : CAT_DIFF)) { lf[kk] = val; lcode |= ((unsigned)n_ev | (listed ? 128u : 0u)) << (8 * kk); }
        }
    }
    // "holds events" flags of the four leaves, then the tree
    unsigned fm = 0u;
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) if ((lcode >> (8 * kk)) & 127u) fm |= 1u << kk;
    const double p0 = lf[0] + lf[1], p1 = lf[2] + lf[3];
    double lv[7];
    lv[0] = p0 + p1;
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
    const double R = lv[6];
    DomPick pk;
    pk.i = 0; pk.j = 0; pk.k = 0; pk.type = -1; pk.m = -1; pk.atom = 0; pk.rate = 0.0; pk.R = 0.0;
    if (mask != 0ull && !(R < 1e-25) && finite_d(R)) {       // wave-uniform
        pk.R = R;
        const double r = counter_uniform(C.seed, (uint64_t)g, KEY_PICK | (uint64_t)d) * R;
        double base = 0.0;
        int a = 0;
#pragma unroll
        for (int l = 5; l >= 0; --l) {          // go left iff the right half holds no events, or the left holds events and base + sum(left) >= r
            const int hw = 1 << l;
            const double sl = readlane_f64(lv[l], a);
            const unsigned long long bits = (hw == 32) ? 0xFFFFFFFFull : ((1ull << hw) - 1ull);
            const bool fl = ((mask >> a) & bits) != 0ull, fr = ((mask >> (a + hw)) & bits) != 0ull;
            if (!(!fr || (fl && base + sl >= r))) { base += sl; a += hw; }
        }
        // inside the lane: every lane walks its own four leaves from the same base; the chosen lane's walk counts
        double bl = base;
        int hp;
        if ((fm & 0xCu) == 0u || ((fm & 0x3u) != 0u && bl + p0 >= r)) hp = 0; else { bl += p0; hp = 1; }
        const double ll = hp ? lf[2] : lf[0];
        const unsigned f2 = (fm >> (2 * hp)) & 0x3u;
        int hl;
        if ((f2 & 0x2u) == 0u || ((f2 & 0x1u) != 0u && bl + ll >= r)) hl = 0; else { bl += ll; hl = 1; }
        const int slot = 2 * hp + hl;
        const double leaf = (slot == 0) ? lf[0] : (slot == 1) ? lf[1] : (slot == 2) ? lf[2] : lf[3];
        const int code = (int)((lcode >> (8 * slot)) & 255u);
        const int la = __builtin_amdgcn_readfirstlane(a);
        const int kk_a = __builtin_amdgcn_readlane(slot, la), code_a = __builtin_amdgcn_readlane(code, la);
        const double base_a = readlane_f64(bl, la), leaf_a = readlane_f64(leaf, la);
        const int b_a = la >> 2, ii_a = (b_a * 11) >> 5, cat = b_a - 3 * ii_a;
        pk.i = i0 + ii_a; pk.j = j0 + (la & 3); pk.k = k0 + kk_a;
        if (code_a == 1 && cat != CAT_DIFF) {    // one event, voxel not listed (or a deposition leaf): nothing to scan
            pk.type = (cat == CAT_DEP) ? EV_DEP : EV_NUC;
            pk.atom = (cat == CAT_DEP) ? 0 : 1;
            pk.rate = leaf_a;
        } else {
            int sl = 0;
            for (int s = 1; s < nslabs; ++s) if (pk.i >= slabs[s].gi0 && pk.i < slabs[s].gi0 + slabs[s].nloc) sl = s;
            const SlabView& S = slabs[sl];
            const SlotPick sp = slot_scan_wave(P, S, ktab_g, pk.i - S.gi0 + 2, pk.j, pk.k, cat, base_a, r, lane);
            pk.type = sp.type; pk.m = sp.m; pk.atom = sp.atom; pk.rate = sp.rate;
        }
    }
    if (lane == 0 && !status) picks[dl] = pk;
}

// Null events: R_max of this rank's boxes into rmax[rank] (zero at launch: k_super_commit clears it); across ranks the
// entries are all-gathered and k_domain_apply takes the largest.  One block per 1024 boxes, the block maxima meet in one
// 64-bit atomic max (positive doubles order like their bit patterns; a maximum does not depend on the order of arrival).
__global__ __launch_bounds__(1024) void k_domain_rmax(const DomPick* __restrict__ picks, int D, const StepState* __restrict__ ss,
                                                       double* __restrict__ rmax, int rank)
{
    __shared__ double wm[16];
    if (ss->status) return;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int d = blockIdx.x * 1024 + tid;
    double m = 0.0;
    if (d < D) { const double R = picks[d].R; if (picks[d].type >= 0 && R > m) m = R; }
#pragma unroll
    for (int l = 0; l < 6; ++l) { const double o = __shfl_xor(m, 1 << l); m = o > m ? o : m; }
    if (lane == 0) wm[w] = m;
    __syncthreads();
    if (tid == 0) {
        for (int q = 1; q < 16; ++q) m = wm[q] > m ? wm[q] : m;
        if (m > 0.0) atomicMax(reinterpret_cast<unsigned long long*>(rmax + rank), (unsigned long long)__double_as_longlong(m));
    }
}

// One thread per box: event record, uniforms, lattice write (kmc_simulation.py:276-327).  Reads stay within +-2 of the
// chosen voxel and so do the writes of every other box's event (>= 5 away on some axis): no box reads what another writes.
__global__ __launch_bounds__(64) void k_domain_apply(KParams P, const SlabView* __restrict__ slabs, int nslabs, int L, int D,
                                                     SuperCfg C, StepState* ss, const DomPick* __restrict__ picks,
                                                     cetkmc_event* __restrict__ dom_events,
                                                     unsigned long long* counters /* per slot: [0] executed, [1] nucleations */,
                                                     cetkmc_event* log_events /* [n][D] or null */,
                                                     const double* __restrict__ rmax /* [C.nranks] window-total maxima (null events) */)
{
    if (ss->status) return;
    const int dl = blockIdx.x * 64 + threadIdx.x;       // local box
    if (dl >= D) return;
    const int d = C.d0 + dl;                            // global box index (keys the uniforms)
    const int64_t cur = ss->cur;
    const int64_t g = C.step0 + cur;
    const DomPick pk = picks[dl];
    cetkmc_event ev;
    ev.type = pk.type;
    ev.pos[0] = ev.pos[1] = ev.pos[2] = 0;
    ev.target[0] = ev.target[1] = ev.target[2] = -1;
    ev.atom = 0; ev.rate = 0.0; ev.dep_rank = -1; ev.theta = 0.0; ev.phi = 0.0;
    bool accepted = true;
    if (pk.type >= 0 && C.null_events) {
        double Rmax = rmax[0];
        for (int q = 1; q < C.nranks; ++q) Rmax = rmax[q] > Rmax ? rmax[q] : Rmax;
        accepted = counter_uniform(C.seed, (uint64_t)g, KEY_ACCEPT | (uint64_t)d) * Rmax < pk.R;
    }
    if (pk.type >= 0) {
        const int i = pk.i, j = pk.j, k = pk.k;
        ev.pos[0] = i; ev.pos[1] = j; ev.pos[2] = k;
        ev.atom = pk.atom; ev.rate = pk.rate;
        if (pk.m >= 0) {
            ev.target[0] = i + nbi_rt(pk.m); ev.target[1] = j + nbj_rt(pk.m); ev.target[2] = k + nbk_rt(pk.m);
        }
    }
    if (!accepted) ev.type = -2;        // null event: the pick is logged, nothing is drawn or applied
    if (ev.type >= 0) {
        const int i = pk.i, j = pk.j, k = pk.k;
        if (pk.m >= 0) {
            int sl = 0;
            for (int s = 0; s < nslabs; ++s) if (i >= slabs[s].gi0 && i < slabs[s].gi0 + slabs[s].nloc) sl = s;
            const SlabView& S = slabs[sl];
            const int li = i - S.gi0 + 2;
            const int ai = nbi_rt(pk.m), aj = nbj_rt(pk.m), ak = nbk_rt(pk.m);
            const int64_t src = (pk.type == EV_DIFF) ? S.tidx(li, j, k) : S.tidx(li + ai, j + aj, k + ak);
            ev.theta = S.theta[src]; ev.phi = S.phi[src];
        }
        if (pk.type == EV_DEP)
            ev.atom = dep_species(P, counter_uniform(C.seed, (uint64_t)g, (uint64_t)j * (uint64_t)L + (uint64_t)k));
        if (pk.type == EV_DEP || pk.type == EV_NUC) {
            ev.theta = 0.0 + (3.141592653589793 - 0.0) * counter_uniform(C.seed, (uint64_t)g, KEY_THETA | (uint64_t)d);
            ev.phi = 0.0 + (6.283185307179586 - 0.0) * counter_uniform(C.seed, (uint64_t)g, KEY_PHI | (uint64_t)d);
        }
    }
    dom_events[dl] = ev;
    if (log_events) log_events[cur * (int64_t)D + dl] = ev;
    if (ev.type < 0) return;
    const int mk = (C.defect_fraction > 0.0 && counter_uniform(C.seed, (uint64_t)g, KEY_DEFECT | (uint64_t)d) < C.defect_fraction) ? 1 : 0;
    apply_event(slabs, nslabs, ev, mk);
    // (a block's 64 lanes share a slot: the compiler folds them into one atomic per wave)
    unsigned long long* slot = counters + (size_t)(blockIdx.x % SUPER_CNT_SLOTS) * SUPER_CNT_STRIDE;
    atomicAdd(&slot[0], 1ull);
    if (ev.type == EV_NUC) atomicAdd(&slot[1], 1ull);
}

// Across ranks: the events of the neighbour ranks' boundary box layers (nb^2 boxes each, received after their own
// k_domain_apply) applied to this rank's copy -- write_site() clips to the slab + halo, so owned planes receive
// the diffusion targets that crossed the slab boundary and the halo planes stay a faithful copy of the neighbour.
// recv[0 .. nb^2): layer below (global boxes d0 - nb^2 + q), recv[nb^2 .. 2 nb^2): layer above (d0 + D_loc + q').
__global__ __launch_bounds__(64) void k_domain_apply_remote(const SlabView* __restrict__ slabs, int nslabs, SuperCfg C,
                                                            const StepState* __restrict__ ss, const cetkmc_event* __restrict__ recv)
{
    if (ss->status) return;
    const int nb2 = C.nb * C.nb;
    const int q = blockIdx.x * 64 + threadIdx.x;
    if (q >= 2 * nb2) return;
    const cetkmc_event ev = recv[q];
    if (ev.type < 0) return;
    const int d = (q < nb2) ? (C.d0 - nb2 + q) : (C.d0 + C.D_loc + (q - nb2));
    const int64_t g = C.step0 + ss->cur;
    const int mk = (C.defect_fraction > 0.0 && counter_uniform(C.seed, (uint64_t)g, KEY_DEFECT | (uint64_t)d) < C.defect_fraction) ? 1 : 0;
    apply_event(slabs, nslabs, ev, mk);
}

// Interface upkeep for the touched voxels, after ALL lattice writes of the super-step.  8 boxes per 256-thread
// block, 32 lanes per event (0..14 neighbourhood of the site, 16..30 of a diffusion target): membership flag,
// packed neighbourhood word, and the voxel's EMPTY/DIFF category sum re-evaluated from that word exactly as
// k_interface does.  These are the only listed voxels whose sums an event can change while T stands still, so the
// interface kernel itself runs only on super-steps that follow a temperature update (and on the first of a
// batch); the list is rebuilt in address order by k_ifc_relist right before it (k_interface is bound by the
// locality of its gathers, Mode B lists are long).
#ifndef CETKMC_TOUCH_ATTR
#define CETKMC_TOUCH_ATTR __attribute__((amdgpu_waves_per_eu(4, 4)))    // 145 VGPRs otherwise (3 waves/SIMD): -10 us per super-step
#endif
__global__ __launch_bounds__(256) CETKMC_TOUCH_ATTR void k_domain_touch(KParams P, const SlabView* __restrict__ slabs, int nslabs, int D,
                                                      const cetkmc_event* __restrict__ dom_events, const StepState* __restrict__ ss,
                                                      const double* __restrict__ ktab_g)
{
    __shared__ double ktab[225];
    if (ss->status) return;
    const int tid = threadIdx.x;
    if (tid < 225) ktab[tid] = ktab_g[tid];
    __syncthreads();
    // 16 lanes per (event, half): the event's voxel and its 14 neighbours; the first grid half takes the events' own voxels,
    // the second half the diffusion targets (whole waves of it leave at once where no event of theirs is a diffusion)
    const int nb_first = (D + 15) / 16;
    const bool second = (int)blockIdx.x >= nb_first;
    int bh = (int)blockIdx.x - (second ? nb_first : 0);
    if ((nb_first & 7) == 0) bh = (bh & 7) * (nb_first >> 3) + (bh >> 3);       // consecutive events on one XCD (shared lines)
    const int d = bh * 16 + (tid >> 4), m = tid & 15;
    if (d >= D || m == 15) return;
    const cetkmc_event ev = dom_events[d];
    if (ev.type < 0 || (second && ev.type != EV_DIFF)) return;
    int ai = second ? ev.target[0] : ev.pos[0], aj = second ? ev.target[1] : ev.pos[1], ak = second ? ev.target[2] : ev.pos[2];
    if (m < 14) { ai += nbi_rt(m); aj += nbj_rt(m); ak += nbk_rt(m); }
    for (int s = 0; s < nslabs; ++s) {
        const SlabView& S = slabs[s];
        const int lp = ai - S.gi0;
        if (ai < 0 || ai >= S.L || aj < 0 || aj >= S.L || ak < 0 || ak >= S.L || lp < 0 || lp >= S.nloc) continue;
        bool hit;
        const unsigned code = ifc_encode(S, lp + 2, aj, ak, &hit);
        const int64_t t = S.tidx(lp + 2, aj, ak);
        if (hit) S.ifc_in[t] = 1;
        if (!hit && !S.ifc_in[t]) continue;
        S.ifc_code[t] = code;
        const int st = code_state(code);
        const double Tc = pymax(S.T[t], 1.0);
        double sum = 0.0;
        int cnt = 0;
#ifndef CETKMC_TOUCH_BATCH
#define CETKMC_TOUCH_BATCH 4
#endif
        if (st == 0) ifc_eval_empty<CETKMC_TOUCH_BATCH>(P, S, ktab, lp, aj, ak, t, code, Tc, sum, cnt);
        else if (st != 4) ifc_eval_atom(P, S, lp, aj, ak, t, code, st, Tc, sum, cnt);
        ifc_store(S, lp + 2, aj, ak, t, sum, cnt, code);
    }
}

// Rebuild a slab's interface list from the membership flags, in address order inside each block's 32 rows
// (one global atomic per block).  *S.ifc_n must be zero at launch.
constexpr int RELIST_ROWS = 32;
__global__ __launch_bounds__(256) void k_ifc_relist(SlabView S, const StepState* __restrict__ ss)
{
    __shared__ int wsum[4], wbase[4];
    __shared__ int base;
    if (ss && ss->status) return;
    const int L = S.L, nrows = S.nloc * L;
    const int r0 = blockIdx.x * RELIST_ROWS, r1 = min(r0 + RELIST_ROWS, nrows);
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    int cnt = 0;
    for (int r = r0 + w; r < r1; r += 4) {
        const int lp = r / L, j = r - lp * L;
        const uint8_t* row = S.ifc_in + S.tidx(lp + 2, j, 0);
        for (int k = lane; k < L; k += 64) cnt += row[k] != 0;
    }
    cnt = wave_sum_i(cnt);
    if (lane == 0) wsum[w] = cnt;
    __syncthreads();
    if (tid == 0) {
        int tot = 0;
        for (int q = 0; q < 4; ++q) { wbase[q] = tot; tot += wsum[q]; }
        base = tot ? atomicAdd(S.ifc_n, tot) : 0;
    }
    __syncthreads();
    int off = base + wbase[w];
    for (int r = r0 + w; r < r1; r += 4) {
        const int lp = r / L, j = r - lp * L;
        const uint8_t* row = S.ifc_in + S.tidx(lp + 2, j, 0);
        for (int k0 = 0; k0 < L; k0 += 64) {
            const int k = k0 + lane;
            const bool f = k < L && row[k] != 0;
            const unsigned long long mask = __ballot(f);
            if (f) S.ifc_list[off + __popcll(mask & ((1ull << lane) - 1ull))] = ((unsigned)lp << 20) | ((unsigned)j << 10) | (unsigned)k;
            off += __popcll(mask);
        }
    }
}

__global__ __launch_bounds__(64) void k_super_commit(StepState* ss, unsigned long long* counters, double* log_total, int64_t* log_exec,
                                                     double* rmax, int nranks)
{
    if (ss->status) return;
    const int lane = threadIdx.x;
    if (rmax && lane < nranks) rmax[lane] = 0.0;            // the next super-step's window-total maxima start from zero
    long long ex = 0, nu = 0;
    for (int q = lane; q < SUPER_CNT_SLOTS; q += 64) {
        unsigned long long* slot = counters + (size_t)q * SUPER_CNT_STRIDE;
        ex += (long long)slot[0]; nu += (long long)slot[1];
        slot[0] = 0; slot[1] = 0;
    }
#pragma unroll
    for (int l = 0; l < 6; ++l) { ex += __shfl_xor(ex, 1 << l); nu += __shfl_xor(nu, 1 << l); }
    if (lane != 0) return;
    const int64_t s = ss->cur;
    if (log_total) log_total[s] = ss->total;
    if (log_exec) log_exec[s] = ex;
    ss->nuc_count += nu;
    ss->cur = s + 1;
}

}  // namespace cetkmc
