// cluster.hpp -- grain clustering on the device (reference utils.py:28-84).
//
// The reference's dfs_cluster joins a voxel to the cluster of the CURRENT stack voxel when both are
// occupied (state != 0) 14-stencil neighbours whose misorientation (kmc_event_rates.py:9-23) is below
// the threshold.  The join predicate is symmetric, so the clusters are exactly the connected components
// of that undirected graph, independent of traversal order; the reference numbers them 1,2,... in the
// order of their first voxel in row-major order.  Here: lock-free union-find that always links the
// larger root under the smaller linear index (so a component's root IS its first row-major voxel),
// then per-cluster size and bounding box by atomics.  Single slab only (whole lattice on one GPU).
#pragma once
#include "voxel.hpp"

namespace cetkmc {

__device__ __forceinline__ int cc_find(int* parent, int x)
{
    int p = __atomic_load_n(parent + x, __ATOMIC_RELAXED);
    while (p != x) {
        const int gp = __atomic_load_n(parent + p, __ATOMIC_RELAXED);
        if (gp != p) atomicMin(parent + x, gp);          // path halving (monotone: only ever decreases)
        x = p;
        p = gp;
    }
    return x;
}
__device__ __forceinline__ void cc_unite(int* parent, int a, int b)
{
    while (true) {
        a = cc_find(parent, a);
        b = cc_find(parent, b);
        if (a == b) return;
        if (a > b) { const int t = a; a = b; b = t; }     // a < b: link root b under a
        const int old = atomicMin(parent + b, a);
        if (old == b) return;                               // b was still a root: linked
        b = old;                                            // somebody re-parented b meanwhile: retry from there
    }
}

// parent[v] = v for occupied voxels, -1 for empty ones; v = (lp*L + j)*L + k over the owned planes
__global__ void k_cc_init(SlabView S, int* parent)
{
    const int L = S.L;
    const int64_t n = (int64_t)S.nloc * L * L;
    for (int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; v < n; v += (int64_t)gridDim.x * blockDim.x) {
        const int k = (int)(v % L); const int64_t t = v / L; const int j = (int)(t % L), lp = (int)(t / L);
        parent[v] = (S.state[S.sidx(lp + 2, j, k)] != 0) ? (int)v : -1;
    }
}
// one thread per voxel: unite with the 7 "forward" neighbours (the other 7 are covered from the other side)
__global__ void k_cc_hook(SlabView S, int* parent, double threshold)
{
    const int L = S.L;
    const int64_t n = (int64_t)S.nloc * L * L;
    for (int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; v < n; v += (int64_t)gridDim.x * blockDim.x) {
        const int k = (int)(v % L); const int64_t t = v / L; const int j = (int)(t % L), lp = (int)(t / L);
        const int li = lp + 2;
        if (S.state[S.sidx(li, j, k)] == 0) continue;
        const double* a = S.ovec + 3 * S.tidx(li, j, k);
        const double a0 = a[0], a1 = a[1], a2 = a[2];
        // forward half of the stencil (positive linear offset): slots 0,1,4,5,8,10,12 of kmc_event_rates.py:29-35
        const int fwd[7] = {0, 1, 4, 5, 8, 10, 12};
#pragma unroll
        for (int q = 0; q < 7; ++q) {
            const int m = fwd[q];
            const int nlp = lp + nbi_rt(m), nj = j + nbj_rt(m), nk = k + nbk_rt(m);
            if (nlp < 0 || nlp >= S.nloc || nj < 0 || nj >= L || nk < 0 || nk >= L) continue;
            if (S.state[S.sidx(nlp + 2, nj, nk)] == 0) continue;
            const double* b = S.ovec + 3 * S.tidx(nlp + 2, nj, nk);
            double dot = a0 * b[0] + a1 * b[1] + a2 * b[2];
            dot = pymax(pymin(dot, 1.0), -1.0);
            if (acos(dot) < threshold) cc_unite(parent, (int)v, (int)(((int64_t)nlp * L + nj) * L + nk));
        }
    }
}
// flatten + collect roots
__global__ void k_cc_compress(int64_t n, int* parent, int* roots, int* n_roots)
{
    for (int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; v < n; v += (int64_t)gridDim.x * blockDim.x) {
        if (parent[v] < 0) continue;
        const int r = cc_find(parent, (int)v);
        parent[v] = r;
        if (r == (int)v) roots[atomicAdd(n_roots, 1)] = r;
    }
}
// cid[root] = 1-based cluster id (roots sorted ascending by the host)
__global__ void k_cc_ids(const int* __restrict__ sorted_roots, int n_roots, int* cid)
{
    const int q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q < n_roots) cid[sorted_roots[q]] = q + 1;
}
// labels + per-cluster size / bounding box.  stats[id-1] = {size, imin,jmin,kmin, imax,jmax,kmax, pad}
__global__ void k_cc_stats(SlabView S, const int* __restrict__ parent, const int* __restrict__ cid, int* labels, int* stats)
{
    const int L = S.L;
    const int64_t n = (int64_t)S.nloc * L * L;
    for (int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; v < n; v += (int64_t)gridDim.x * blockDim.x) {
        const int p = parent[v];
        if (p < 0) { labels[v] = 0; continue; }
        const int id = cid[cc_find(const_cast<int*>(parent), p)];
        labels[v] = id;
        const int k = (int)(v % L); const int64_t t = v / L; const int j = (int)(t % L), i = S.gi0 + (int)(t / L);
        int* s = stats + 8 * (int64_t)(id - 1);
        atomicAdd(s + 0, 1);
        atomicMin(s + 1, i); atomicMin(s + 2, j); atomicMin(s + 3, k);
        atomicMax(s + 4, i); atomicMax(s + 5, j); atomicMax(s + 6, k);
    }
}
__global__ void k_cc_stats_init(int n_roots, int* stats)
{
    const int q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q < n_roots) {
        int* s = stats + 8 * (int64_t)q;
        s[0] = 0; s[1] = s[2] = s[3] = 0x7fffffff; s[4] = s[5] = s[6] = -1; s[7] = 0;
    }
}

// ---- sparse site queries for the host-side defect model (defects.track_defects, defects.py:4-19) ----
// counts[s] = number of owned voxels in state s (0..4), counts[5] = anything else
__global__ void k_species_counts(SlabView S, unsigned long long* counts)
{
    __shared__ unsigned int hist[6];
    if (threadIdx.x < 6) hist[threadIdx.x] = 0;
    __syncthreads();
    const int L = S.L;
    const int64_t n = (int64_t)S.nloc * L * L;
    for (int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; v < n; v += (int64_t)gridDim.x * blockDim.x) {
        const int k = (int)(v % L); const int64_t t = v / L; const int j = (int)(t % L), lp = (int)(t / L);
        const int st = S.state[S.sidx(lp + 2, j, k)];
        atomicAdd(&hist[st < 5 ? st : 5], 1u);
    }
    __syncthreads();
    if (threadIdx.x < 6) atomicAdd(counts + threadIdx.x, (unsigned long long)hist[threadIdx.x]);
}
// (global linear index, T) of every owned voxel in state `species` (unordered; the host sorts)
__global__ void k_gather_species(SlabView S, int species, long long* idx, double* Tv, unsigned long long cap, unsigned long long* n_out)
{
    const int L = S.L;
    const int64_t n = (int64_t)S.nloc * L * L;
    for (int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; v < n; v += (int64_t)gridDim.x * blockDim.x) {
        const int k = (int)(v % L); const int64_t t = v / L; const int j = (int)(t % L), lp = (int)(t / L);
        if (S.state[S.sidx(lp + 2, j, k)] != species) continue;
        const unsigned long long pos = atomicAdd(n_out, 1ULL);
        if (pos < cap) { idx[pos] = ((long long)(S.gi0 + lp) * L + j) * L + k; Tv[pos] = S.T[S.tidx(lp + 2, j, k)]; }
    }
}
__global__ void k_scatter_defects(SlabView S, const long long* __restrict__ idx, long long n)
{
    const int L = S.L;
    for (long long q = (long long)blockIdx.x * blockDim.x + threadIdx.x; q < n; q += (long long)gridDim.x * blockDim.x) {
        const long long g = idx[q];
        const int k = (int)(g % L); const long long t = g / L; const int j = (int)(t % L), i = (int)(t / L);
        const int li = i - (S.gi0 - 2);
        if (li >= 0 && li < S.nloc + 4) S.defects[S.sidx(li, j, k)] = 1;
    }
}

}  // namespace cetkmc
