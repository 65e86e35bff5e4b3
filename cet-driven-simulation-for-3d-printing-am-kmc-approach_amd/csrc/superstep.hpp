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

constexpr uint64_t KEY_PICK = 1ull << 40, KEY_THETA = 2ull << 40, KEY_PHI = 3ull << 40, KEY_DEFECT = 4ull << 40;

struct SuperCfg {
    int64_t step0;
    double defect_fraction;
    uint64_t seed;
    int32_t box, H, PH, PT;     // box edge, window edge (box/2), pow2(H), pow2(3H)
    int32_t nb;                 // boxes per axis
    int32_t d0, D_loc;          // this handle's boxes: global indices [d0, d0 + D_loc) (box layers of its owned planes)
};

struct DomPick {          // result of the tree descent of one box
    int32_t i, j, k, cat;  // cat < 0: idle box
    double base, r;
    double rate;           // simple: the chosen leaf itself (its single event's rate)
    int32_t simple, pad;   // 1: a single event whose kind is known without evaluating the voxel again (deposition, or the
                           // nucleation of an empty voxel that is not an interface voxel)
};

// One wave per box: leaves = category sums of the window's voxels, LDS heap of NL = PT*PH*PH leaves, descent.
// The per-voxel rate table (SlabView::vval) holds the EMPTY- or DIFF-category sum of EVERY owned voxel that owns events
// (listed voxels: interface sums + ifc_cnt; other empty voxels: the nucleation rate by temperature), so a leaf is a
// few loads (dep leaves: one exp, top plane only).
__global__ __launch_bounds__(64) void k_domain_pick(KParams P, const SlabView* __restrict__ slabs, int nslabs, int L,
                                                    SuperCfg C, const StepState* __restrict__ ss, DomPick* __restrict__ picks)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x;
    const int d = C.d0 + blockIdx.x;            // global box index (keys the uniforms)
    if (ss->status) return;
    const int NL = C.PT * C.PH * C.PH;
    double* hs = reinterpret_cast<double*>(smem);             // [2*NL]
    uint8_t* hf = reinterpret_cast<uint8_t*>(hs + 2 * NL);    // [2*NL]
    uint8_t* lc = hf + 2 * NL;                                // [NL] leaf: event count | 128 if the voxel is a listed interface voxel
    for (int q = lane; q < NL; q += 64) { hs[NL + q] = 0.0; hf[NL + q] = 0; lc[q] = 0; }
    __syncthreads();
    const int64_t g = C.step0 + ss->cur;
    const int sec = (int)(g & 7);
    const int H = C.H, nb = C.nb;
    const int di = d / (nb * nb), dj = (d / nb) % nb, dk = d % nb;
    const int i0 = di * C.box + ((sec >> 2) & 1) * H, j0 = dj * C.box + ((sec >> 1) & 1) * H, k0 = dk * C.box + (sec & 1) * H;
    for (int v = lane; v < H * H * H; v += 64) {
        const int kk = v % H, jj = (v / H) % H, ii = v / (H * H);
        const int i = i0 + ii, j = j0 + jj, k = k0 + kk;
        int sl = 0;
        for (int s = 0; s < nslabs; ++s) if (i >= slabs[s].gi0 && i < slabs[s].gi0 + slabs[s].nloc) sl = s;
        const SlabView& S = slabs[sl];
        const int li = i - S.gi0 + 2;
        const int st = S.state[S.sidx(li, j, k)];
        const int64_t t = S.tidx(li, j, k);
        // value, count and membership flag are requested together with the state
        double val = S.vval[t];
        int n_ev = S.ifc_cnt[t];
        const int listed = S.ifc_in[t] != 0;
        if (st >= 128 || st == 4) continue;
        if (!listed) {          // not an interface voxel: an empty voxel owns at most its nucleation (the table entry), an atom nothing
            if (st == 0) n_ev = (val != 0.0) ? 1 : 0;
            else { val = 0.0; n_ev = 0; }
        }
        const int c = (st == 0) ? CAT_EMPTY : CAT_DIFF;
        const int q = ((3 * ii + c) * C.PH + jj) * C.PH + kk;
        hs[NL + q] = val; hf[NL + q] = n_ev > 0; lc[q] = (uint8_t)(n_ev | (listed << 7));
        if (st == 0 && i == L - 1) {
            const double rate = dep_rate(P, pymax(S.T[t], 1.0));
            if (finite_d(rate)) {
                const int qd = ((3 * ii + CAT_DEP) * C.PH + jj) * C.PH + kk;
                hs[NL + qd] = rate; hf[NL + qd] = 1; lc[qd] = 1;
            }
        }
    }
    __syncthreads();
    for (int n = NL >> 1; n >= 1; n >>= 1) {
        for (int idx = lane; idx < n; idx += 64) {
            const int node = n + idx;
            hs[node] = hs[2 * node] + hs[2 * node + 1];
            hf[node] = hf[2 * node] | hf[2 * node + 1];
        }
        __syncthreads();
    }
    if (lane != 0) return;
    DomPick pk;
    pk.i = pk.j = pk.k = 0; pk.cat = -1; pk.base = 0.0; pk.r = 0.0; pk.rate = 0.0; pk.simple = 0; pk.pad = 0;
    const double R = hs[1];
    if (hf[1] && !(R < 1e-25) && finite_d(R)) {
        const double r = counter_uniform(C.seed, (uint64_t)g, KEY_PICK | (uint64_t)d) * R;
        double base = 0.0;
        int n = 1;
        while (n < NL) {
            const int l = 2 * n;
            if (hf[l + 1] == 0 || (hf[l] != 0 && base + hs[l] >= r)) n = l;
            else { base += hs[l]; n = l + 1; }
        }
        const int q = n - NL;
        const int b = q / (C.PH * C.PH);
        pk.i = i0 + b / 3; pk.cat = b % 3; pk.j = j0 + (q / C.PH) % C.PH; pk.k = k0 + q % C.PH;
        pk.base = base; pk.r = r;
        pk.rate = hs[NL + q];
        pk.simple = (lc[q] == 1 && pk.cat != CAT_DIFF) ? 1 : 0;     // one event, voxel not listed (or a deposition leaf)
    }
    picks[blockIdx.x] = pk;
}

// One thread per box: slot scan inside the chosen voxel, uniforms, lattice write (kmc_simulation.py:276-327).
// Reads stay within +-2 of the chosen voxel and so do the writes of every other box's event (>= 5 away on
// some axis): no box reads what another one writes, selection and application can share a kernel.
__global__ __launch_bounds__(64) void k_domain_slot_apply(KParams P, const SlabView* __restrict__ slabs, int nslabs, int L, int D,
                                                          SuperCfg C, StepState* ss, const DomPick* __restrict__ picks,
                                                          const double* __restrict__ ktab_g, cetkmc_event* __restrict__ dom_events,
                                                          unsigned long long* counters /* [0] executed, [1] nucleations */,
                                                          cetkmc_event* log_events /* [n][D] or null */)
{
    __shared__ double ktab[226];
    for (int t = threadIdx.x; t < 225; t += 64) ktab[t] = ktab_g[t];
    __syncthreads();
    if (ss->status) return;
    const int dl = blockIdx.x * 64 + threadIdx.x;       // local box
    if (dl >= D) return;
    const int d = C.d0 + dl;                            // global box index (keys the uniforms)
    const int64_t g = C.step0 + ss->cur;
    cetkmc_event ev;
    ev.type = -1;
    ev.pos[0] = ev.pos[1] = ev.pos[2] = 0;
    ev.target[0] = ev.target[1] = ev.target[2] = -1;
    ev.atom = 0; ev.rate = 0.0; ev.dep_rank = -1; ev.theta = 0.0; ev.phi = 0.0;
    const DomPick pk = picks[dl];
    if (pk.cat >= 0) {
        const int i = pk.i, j = pk.j, k = pk.k, c = pk.cat;
        const double r = pk.r;
        int sl = 0;
        for (int s = 0; s < nslabs; ++s) if (i >= slabs[s].gi0 && i < slabs[s].gi0 + slabs[s].nloc) sl = s;
        const SlabView& S = slabs[sl];
        const int li = i - S.gi0 + 2;
        int p_type = -1, p_m = -1, p_atom = 0;
        double p_rate = 0.0;
        if (pk.simple) {            // the deposition of this voxel / the nucleation of a bulk empty voxel: nothing to scan
            p_type = (c == CAT_DEP) ? EV_DEP : EV_NUC;
            p_atom = (c == CAT_DEP) ? 0 : 1;
            p_rate = pk.rate;
        } else {
            const int st = S.state[S.sidx(li, j, k)];
            double cum = pk.base;
            bool found = false;
            auto nbs = [&](int mm) -> int { return S.state[S.sidx(li + nbi_rt(mm), j + nbj_rt(mm), k + nbk_rt(mm))]; };
            auto emit = [&](int cat, int type, double rate, int m, int atom) {
                if (cat != c || found) return;
                cum += rate;
                p_type = type; p_m = m; p_atom = atom; p_rate = rate;   // remembers the last valid slot
                if (cum >= r) found = true;
            };
            eval_voxel(P, S, ktab, li, i, j, k, st, S.T[S.tidx(li, j, k)], nbs, emit);
        }
        ev.type = p_type;
        ev.pos[0] = i; ev.pos[1] = j; ev.pos[2] = k;
        ev.atom = p_atom; ev.rate = p_rate;
        if (p_m >= 0) {
            const int ai = nbi_rt(p_m), aj = nbj_rt(p_m), ak = nbk_rt(p_m);
            ev.target[0] = i + ai; ev.target[1] = j + aj; ev.target[2] = k + ak;
            const int64_t src = (p_type == EV_DIFF) ? S.tidx(li, j, k) : S.tidx(li + ai, j + aj, k + ak);
            ev.theta = S.theta[src]; ev.phi = S.phi[src];
        }
        if (p_type == EV_DEP)
            ev.atom = dep_species(P, counter_uniform(C.seed, (uint64_t)g, (uint64_t)j * (uint64_t)L + (uint64_t)k));
        if (p_type == EV_DEP || p_type == EV_NUC) {
            ev.theta = 0.0 + (3.141592653589793 - 0.0) * counter_uniform(C.seed, (uint64_t)g, KEY_THETA | (uint64_t)d);
            ev.phi = 0.0 + (6.283185307179586 - 0.0) * counter_uniform(C.seed, (uint64_t)g, KEY_PHI | (uint64_t)d);
        }
    }
    dom_events[dl] = ev;
    if (log_events) log_events[ss->cur * (int64_t)D + dl] = ev;
    if (ev.type < 0) return;
    const int mk = (C.defect_fraction > 0.0 && counter_uniform(C.seed, (uint64_t)g, KEY_DEFECT | (uint64_t)d) < C.defect_fraction) ? 1 : 0;
    apply_event(slabs, nslabs, ev, mk);
    atomicAdd(&counters[0], 1ull);
    if (ev.type == EV_NUC) atomicAdd(&counters[1], 1ull);
}

// Across ranks: the events of the neighbour ranks' boundary box layers (nb^2 boxes each, received after their own
// k_domain_slot_apply) applied to this rank's copy -- write_site() clips to the slab + halo, so owned planes receive
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
    const int d = blockIdx.x * 8 + (tid >> 5), l = tid & 31;
    if (d >= D || l == 15 || l == 31) return;
    const cetkmc_event ev = dom_events[d];
    const bool second = l >= 16;
    if (ev.type < 0 || (second && ev.type != EV_DIFF)) return;
    int ai = second ? ev.target[0] : ev.pos[0], aj = second ? ev.target[1] : ev.pos[1], ak = second ? ev.target[2] : ev.pos[2];
    const int m = l & 15;
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
        const int st = (code >> 30) ? 4 : (int)((code >> 28) & 3u);
        const double Tc = pymax(S.T[t], 1.0);
        double sum = 0.0;
        int cnt = 0;
        if (st == 0) ifc_eval_empty(P, S, ktab, lp, aj, ak, t, code, Tc, sum, cnt);
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

__global__ void k_super_commit(StepState* ss, unsigned long long* counters, double* log_total, int64_t* log_exec)
{
    if (ss->status) return;
    const int64_t s = ss->cur;
    if (log_total) log_total[s] = ss->total;
    if (log_exec) log_exec[s] = (int64_t)counters[0];
    ss->nuc_count += (int64_t)counters[1];
    counters[0] = 0; counters[1] = 0;
    ss->cur = s + 1;
}

}  // namespace cetkmc
