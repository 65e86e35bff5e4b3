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
};

// One wave per box.  LDS: heap of NL = PT*PH*PH leaves (sums + has-events flags).
__global__ __launch_bounds__(64) void k_domain_select(KParams P, const SlabView* __restrict__ slabs, int nslabs, int L,
                                                      SuperCfg C, const StepState* __restrict__ ss,
                                                      const double* __restrict__ ktab_g, cetkmc_event* __restrict__ dom_events,
                                                      int* __restrict__ dom_defect)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    __shared__ double ktab[226];
    const int lane = threadIdx.x;
    const int d = blockIdx.x;
    if (ss->status) return;
    const int NL = C.PT * C.PH * C.PH;
    double* hs = reinterpret_cast<double*>(smem);             // [2*NL]
    uint8_t* hf = reinterpret_cast<uint8_t*>(hs + 2 * NL);    // [2*NL]
    for (int t = lane; t < 225; t += 64) ktab[t] = ktab_g[t];
    for (int q = lane; q < NL; q += 64) { hs[NL + q] = 0.0; hf[NL + q] = 0; }
    __syncthreads();
    const int64_t g = C.step0 + ss->cur;
    const int sec = (int)(g & 7);
    const int H = C.H, nb = C.nb;
    const int di = d / (nb * nb), dj = (d / nb) % nb, dk = d % nb;
    const int i0 = di * C.box + ((sec >> 2) & 1) * H, j0 = dj * C.box + ((sec >> 1) & 1) * H, k0 = dk * C.box + (sec & 1) * H;

    auto slab_of = [&](int i) { int sl = 0; for (int s = 0; s < nslabs; ++s) if (i >= slabs[s].gi0 && i < slabs[s].gi0 + slabs[s].nloc) sl = s; return sl; };

    // ---- leaves: category sums of the window's voxels (frozen lattice) ----------------------
    for (int v = lane; v < H * H * H; v += 64) {
        const int kk = v % H, jj = (v / H) % H, ii = v / (H * H);
        const int i = i0 + ii, j = j0 + jj, k = k0 + kk;
        const SlabView& S = slabs[slab_of(i)];
        const int li = i - S.gi0 + 2;
        const int st = S.state[S.sidx(li, j, k)];
        const int64_t t = S.tidx(li, j, k);
        double sum[3] = {0.0, 0.0, 0.0};
        int cnt[3] = {0, 0, 0};
        if (S.ifc_in[t] && st < 128 && st != 4) {
            // listed interface voxel: k_interface left its EMPTY- or DIFF-category sum in ifc_val
            const int c = (st == 0) ? CAT_EMPTY : CAT_DIFF;
            sum[c] = S.ifc_val[t]; cnt[c] = S.ifc_cnt[t];
            if (st == 0 && i == L - 1) {
                const double rate = dep_rate(P, pymax(S.T[t], 1.0));
                if (finite_d(rate)) { sum[CAT_DEP] = rate; cnt[CAT_DEP] = 1; }
            }
        } else {
            auto nbs = [&](int mm) -> int { return S.state[S.sidx(li + nbi_rt(mm), j + nbj_rt(mm), k + nbk_rt(mm))]; };
            auto emit = [&](int cat, int, double rate, int, int) {
                if (cat == CAT_DEP) { sum[0] += rate; ++cnt[0]; }
                else if (cat == CAT_DIFF) { sum[1] += rate; ++cnt[1]; }
                else { sum[2] += rate; ++cnt[2]; }
            };
            eval_voxel(P, S, ktab, li, i, j, k, st, S.T[t], nbs, emit);
        }
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const int q = ((3 * ii + c) * C.PH + jj) * C.PH + kk;
            hs[NL + q] = sum[c]; hf[NL + q] = cnt[c] > 0;
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
    // ---- pick (lane 0): tree descent, slot scan, uniforms ------------------------------------
    cetkmc_event ev;
    ev.type = -1;
    ev.pos[0] = ev.pos[1] = ev.pos[2] = 0;
    ev.target[0] = ev.target[1] = ev.target[2] = -1;
    ev.atom = 0; ev.rate = 0.0; ev.dep_rank = -1; ev.theta = 0.0; ev.phi = 0.0;
    int mk = 0;
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
        const int kk = q % C.PH, jj = (q / C.PH) % C.PH, b = q / (C.PH * C.PH);
        const int i = i0 + b / 3, c = b % 3, j = j0 + jj, k = k0 + kk;
        const SlabView& S = slabs[slab_of(i)];
        const int li = i - S.gi0 + 2;
        const int st = S.state[S.sidx(li, j, k)];
        double cum = base;
        bool found = false;
        int p_type = -1, p_m = -1, p_atom = 0;
        double p_rate = 0.0;
        auto nbs = [&](int mm) -> int { return S.state[S.sidx(li + nbi_rt(mm), j + nbj_rt(mm), k + nbk_rt(mm))]; };
        auto emit = [&](int cat, int type, double rate, int m, int atom) {
            if (cat != c || found) return;
            cum += rate;
            p_type = type; p_m = m; p_atom = atom; p_rate = rate;   // remembers the last valid slot
            if (cum >= r) found = true;
        };
        eval_voxel(P, S, ktab, li, i, j, k, st, S.T[S.tidx(li, j, k)], nbs, emit);
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
        if (p_type >= 0)
            mk = (C.defect_fraction > 0.0 && counter_uniform(C.seed, (uint64_t)g, KEY_DEFECT | (uint64_t)d) < C.defect_fraction) ? 1 : 0;
    }
    dom_events[d] = ev;
    dom_defect[d] = mk;
}

// lattice writes of all picked events (one thread per box; the written voxels are pairwise distinct)
__global__ __launch_bounds__(256) void k_domain_apply(const SlabView* __restrict__ slabs, int nslabs, int D,
                                                      const cetkmc_event* __restrict__ dom_events, const int* __restrict__ dom_defect,
                                                      StepState* ss, unsigned long long* counters /* [0] executed, [1] nucleations */,
                                                      cetkmc_event* log_events /* [n][D] or null */)
{
    if (ss->status) return;
    const int d = blockIdx.x * blockDim.x + threadIdx.x;
    if (d >= D) return;
    const cetkmc_event ev = dom_events[d];
    if (log_events) log_events[ss->cur * (int64_t)D + d] = ev;
    if (ev.type < 0) return;
    apply_event(slabs, nslabs, ev, dom_defect[d]);
    atomicAdd(&counters[0], 1ull);
    if (ev.type == EV_NUC) atomicAdd(&counters[1], 1ull);
}

// interface-list upkeep for the touched voxels, after ALL lattice writes of the super-step (one wave per box)
__global__ __launch_bounds__(64) void k_domain_touch(KParams P, const SlabView* __restrict__ slabs, int nslabs,
                                                     const cetkmc_event* __restrict__ dom_events, const StepState* __restrict__ ss,
                                                     const double* __restrict__ ktab_g)
{
    if (ss->status) return;
    const cetkmc_event ev = dom_events[blockIdx.x];
    if (ev.type < 0) return;
    apply_touch(P, slabs, nslabs, ktab_g, ev, threadIdx.x, 0);
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
