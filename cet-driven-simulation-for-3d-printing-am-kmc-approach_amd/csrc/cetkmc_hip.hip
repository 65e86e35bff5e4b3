// cetkmc_hip.hip -- C ABI (include/cetkmc.h) + host orchestration of the HIP kernels.
//
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -ffp-contract=off
//        -fno-fast-math cetkmc_hip.hip -o libcetkmc_hip.so -ldl
// RCCL is dlopen()ed lazily (only multi-process runs need it).  There is no CPU fallback:
// every compute entry point fails if no HIP device is usable.
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdint>
#include <climits>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "kernels.hpp"
#include "cluster.hpp"
#include "superstep.hpp"

using namespace cetkmc;

namespace {

thread_local std::string g_err;
int fail(const std::string& m) { g_err = m; return 1; }

#define HIPCHK(expr)                                                                        \
    do {                                                                                    \
        hipError_t e_ = (expr);                                                             \
        if (e_ != hipSuccess)                                                               \
            return fail(std::string(#expr) + ": " + hipGetErrorString(e_) + " (" + __FILE__ + \
                        ":" + std::to_string(__LINE__) + ")");                              \
    } while (0)
#define CHK(expr) do { if (int rc_ = (expr)) return rc_; } while (0)

int next_pow2(int n) { int p = 1; while (p < n) p <<= 1; return p; }
int round_up(int n, int m) { return (n + m - 1) / m * m; }

// ---- RCCL, loaded on demand -----------------------------------------------------------
struct Rccl {
    void* so = nullptr;
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclAllGather) AllGather = nullptr;
    decltype(&ncclSend) Send = nullptr;
    decltype(&ncclRecv) Recv = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
} g_rccl;

int load_rccl()
{
    if (g_rccl.so) return 0;
    const char* names[] = {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so"};
    void* so = nullptr;
    for (const char* n : names) if ((so = dlopen(n, RTLD_NOW | RTLD_GLOBAL))) break;
    if (!so) return fail(std::string("cannot load librccl.so: ") + dlerror());
#define SYM(field, name)                                                         \
    g_rccl.field = reinterpret_cast<decltype(g_rccl.field)>(dlsym(so, name));    \
    if (!g_rccl.field) return fail(std::string("librccl.so lacks ") + name);
    SYM(GetUniqueId, "ncclGetUniqueId")
    SYM(CommInitRank, "ncclCommInitRank")
    SYM(CommDestroy, "ncclCommDestroy")
    SYM(AllGather, "ncclAllGather")
    SYM(Send, "ncclSend")
    SYM(Recv, "ncclRecv")
    SYM(GroupStart, "ncclGroupStart")
    SYM(GroupEnd, "ncclGroupEnd")
    SYM(GetErrorString, "ncclGetErrorString")
#undef SYM
    g_rccl.so = so;
    return 0;
}
#define NCCLCHK(expr)                                                                             \
    do {                                                                                          \
        ncclResult_t r_ = (expr);                                                                 \
        if (r_ != ncclSuccess)                                                                    \
            return fail(std::string(#expr) + ": " + g_rccl.GetErrorString(r_));                   \
    } while (0)

// ---- handle ---------------------------------------------------------------------------
struct Slab {
    SlabView v{};
    uint8_t* prev = nullptr;
    double* Tbuf[2] = {nullptr, nullptr};
    double* vvalbuf[2] = {nullptr, nullptr};     // rate table of Tbuf[b] (the pair is flipped together)
    double* depbuf[2] = {nullptr, nullptr};      // plane L-1 deposition rates of Tbuf[b]
    size_t nS = 0, nT = 0, nC = 0;   // bytes of a u8 array, doubles of an f64 array, bytes of the class array
};

struct Handle {
    cetkmc_params p{};
    KParams kp{};
    int L = 0, Pk = 1, PB = 1, RJ = 0, pitchS = 0, pitchT = 0, pitchC = 0;
    int dev = 0;
    hipStream_t stream = nullptr, stream2 = nullptr;
    std::vector<Slab> slabs;
    SlabView* d_views[2] = {nullptr, nullptr};   // per T-buffer parity
    int cur = 0;                                 // current T buffer
    int G = 1, my_first = 0;                     // global slab count, index of first local slab
    int own_i0 = 0, own_i1 = 0;
    BlockEnt* d_blocks = nullptr;
    cetkmc_event* d_events_all = nullptr;
    StepState* d_ss = nullptr;
    int* d_dirty = nullptr;      // [1 + DIRTY_MAX] rows made stale by the last applied event (incremental mode)
    double* d_ktab = nullptr;
    KParams* d_kp = nullptr;
    void* d_scratch = nullptr;
    size_t scratch_bytes = 0;
    int* d_flag = nullptr;
    double* d_qtop = nullptr;
    // batch buffers
    double *d_u_pick = nullptr, *d_u_defect = nullptr, *d_u_np = nullptr, *d_q = nullptr, *d_log_total = nullptr;
    cetkmc_event* d_log_event = nullptr;
    int64_t* d_log_nev = nullptr;
    size_t cap_steps = 0, cap_np = 0, cap_q = 0;
    // page-locked host staging for a batch's inputs and logs: copies to / from it are queued on the stream like kernels
    // (a copy to or from pageable memory blocks the caller once per array)
    char* pin_in = nullptr; size_t pin_in_cap = 0;
    char* pin_out = nullptr; size_t pin_out_cap = 0;
    // a batch whose host inputs are already in d_u_pick / d_u_defect / d_u_np / d_q (cetkmc_stage_inputs)
    struct { bool valid = false, has_defect = false; int64_t step0 = 0, n_steps = 0, np_cap = 0, n_q = 0; int thermal_mode = 0; } staged;
    // rccl
    ncclComm_t comm = nullptr;
    int rank = 0, nranks = 1;
    bool swept = false;
    int sweep_auto = 1;        // sweep_variant never set explicitly: lattices of L <= 128 take variant 4 (launch-bound there: one launch
                               // less per sweep; same bits), larger ones variant 1
    int sweep_variant = 1;     // 0 simple, 1 streaming + rate table with the LDS census (default), 2 streaming, nucleation rates
                               // recomputed per sweep, 3 census-free table sweep (same bits, measured no faster: DESIGN.md section 13)
    bool table_fresh = false;  // vval / dep_val match the current T and parameters (k_rate_table)
    bool ifc_fresh = false;    // vval / event count (class byte) of every listed voxel match the current lattice, T, defects and parameters
    int ifc_every_step = 0;    // 1: k_interface before every full sweep (round-1 behaviour, A/B); 0: only when stale --
                               // between temperature updates the apply kernel re-evaluates the <= 30 listed voxels an event touches
    int thermal_variant = 1;   // 1 = plane-marching LDS kernel, 0 = one thread per voxel
    int thermal_general = 0;   // 1: k_thermal_march also where k_thermal_tiles applies (A/B, tests)
    int therm_ni = THERM_NI;   // planes per block of the marching kernel
    int therm_ni16 = THERM16_NI;   // planes per block of k_thermal_tiles16
    int thermal_rpt = 2;       // rows per thread of k_thermal_tiles16 (2: 1024 threads per block at 256 columns (default); 4: 512)
    int thermal_kt = 256;      // columns per tile of k_thermal_tiles16 (256, or 128 with 2 rows per thread: 512 threads)
    int thermal_tiles16 = 1;   // 1: the 16-row k_thermal_tiles16 where tiles apply (default); 0: the 8-row k_thermal_tiles (round-2 kernel)
    int ifc_blocks = 64;       // grid of k_interface (grid-stride over the device-side list length)
    int ifc_block = 256;
    size_t shmem_stream = 0;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    // grain clustering (cetkmc_cluster): results kept until the next call
    int *d_cc_parent = nullptr, *d_cc_roots = nullptr, *d_cc_cid = nullptr, *d_cc_labels = nullptr, *d_cc_stats = nullptr, *d_cc_n = nullptr;
    int64_t cc_n_clusters = -1;
    std::vector<int> cc_roots_sorted;
    std::vector<hipEvent_t> prof;
    cetkmc_host_comm hc{};       // host-relay transport (cetkmc_create_rank_host); used when comm is null
    // profile 2: every collective of the batch is bracketed by a hipEvent pair on the stream (RCCL: the collective's own
    // device time; host relay: copies + callback + the synchronisations around them)
    bool time_comm = false;
    std::vector<hipEvent_t> comm_ev;
    size_t comm_ev_used = 0;
    std::vector<char> hc_stage;
    cetkmc_counters cnt{};       // cetkmc_get_counters: work issued / bytes moved / per-phase device time
    // look-ahead temperature update: T(n+1) + its rate table computed on stream2 while the 19 steps between two updates run
    struct Spec {
        bool valid = false;
        int64_t g = -1;          // global step of the update it stands for
        int laser = 0, use_latent = 0, scrub = 0;
        double dt = 0.0;
        const double* d_q = nullptr;
    } spec;
    // A batch that stops with status 2 (stream exhausted) ON a temperature-update step has already applied that step's update
    // (the thermal kernels run before the selection notices the shortage).  The continuation batch starts at the same
    // global step: it must not apply the update a second time (kmc_simulation.py:248-250 updates T once per 20 steps).
    int64_t therm_applied_g = -1;
    int thermal_table = 1;       // option "thermal_table": the default temperature tiles write the rate table too (no k_rate_table launch)
    int thermal_ahead = 0;       // option "thermal_lookahead": off by default -- measured slower on one GPU:
                                 // the look-ahead kernels run right behind the update, beside the next sweeps, which they slow down
                                 // by more than the update costs (both are memory bound, and they evict the sweep's working set)
    hipEvent_t ev_main = nullptr, ev_spec = nullptr;
    // cetkmc_run_supersteps' working buffers (grow-only: run_kmc(mode="B") under the event-count thermal clock makes one call
    // per super-step, which five hipMalloc / hipFree pairs per call would dominate)
    cetkmc_event* d_sup_dom = nullptr; size_t cap_sup_dom = 0;
    DomPick* d_sup_picks = nullptr; size_t cap_sup_picks = 0;
    unsigned long long* d_sup_cnt = nullptr; size_t cap_sup_cnt = 0;
    cetkmc_event* d_sup_log = nullptr; size_t cap_sup_log = 0;
    double* d_sup_rmax = nullptr; size_t cap_sup_rmax = 0;
};

KParams make_kparams(const cetkmc_params& p)
{
    KParams k{};
    k.nu = p.nu; k.nu_dep = p.nu_dep;
    for (int a = 0; a < 3; ++a) { k.E_b[a] = p.E_b[a]; k.E_diff[a] = p.E_diff[a]; }
    k.kT = p.kT; k.T_melt = p.T_melt; k.I0 = p.I0; k.delta_T_c = p.delta_T_c;
    k.rate_threshold = p.rate_threshold; k.anisotropy = p.anisotropy;
    k.impurity_re = p.impurity_re; k.impurity_c = p.impurity_c;
    k.K_nuc = p.K_nuc; k.beta_imp_nuc = p.beta_imp_nuc; k.max_imp_frac = p.max_imp_frac;
    return k;
}

// kmc_event_rates.py:126-128, same expression order as the device helper k_eff()
double host_k_eff(const cetkmc_params& p, int n_nb, int n_imp)
{
    int den = n_nb > 1 ? n_nb : 1;
    double x = (double)n_imp / (double)den;
    double f_imp = (x < p.max_imp_frac) ? x : p.max_imp_frac;
    double K = p.K_nuc * (1.0 - p.beta_imp_nuc * f_imp);
    double m = (K < p.K_nuc) ? K : p.K_nuc;
    double lo = 0.1 * p.K_nuc;
    return (m > lo) ? m : lo;
}

int upload_ktab(Handle* h)
{
    double tab[225];
    for (int a = 0; a < 15; ++a)
        for (int b = 0; b < 15; ++b) tab[a * 15 + b] = host_k_eff(h->p, a, b);
    HIPCHK(hipMemcpyAsync(h->d_ktab, tab, sizeof tab, hipMemcpyHostToDevice, h->stream));
    HIPCHK(hipMemcpyAsync(h->d_kp, &h->kp, sizeof(KParams), hipMemcpyHostToDevice, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    return 0;
}

int push_views(Handle* h)
{
    for (int par = 0; par < 2; ++par) {
        std::vector<SlabView> v;
        for (auto& s : h->slabs) { SlabView x = s.v; x.T = s.Tbuf[par]; x.vval = s.vvalbuf[par]; x.dep_val = s.depbuf[par]; v.push_back(x); }
        HIPCHK(hipMemcpyAsync(h->d_views[par], v.data(), v.size() * sizeof(SlabView), hipMemcpyHostToDevice, h->stream));
    }
    HIPCHK(hipStreamSynchronize(h->stream));
    return 0;
}

int ensure_scratch(Handle* h, size_t bytes)
{
    if (bytes <= h->scratch_bytes) return 0;
    if (h->d_scratch) HIPCHK(hipFree(h->d_scratch));
    h->d_scratch = nullptr; h->scratch_bytes = 0;
    HIPCHK(hipMalloc(&h->d_scratch, bytes));
    h->scratch_bytes = bytes;
    return 0;
}

int create_common(const cetkmc_params* p, int L, const std::vector<std::pair<int, int>>& ranges, int dev,
                  int G, int my_first, void** out)
{
    if (!p || !out) return fail("null argument");
    if (L < 1) return fail("L must be >= 1");
    if (3 * L > PMAX) return fail("L too large for this build (3L <= " + std::to_string(PMAX) + ")");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return fail("no usable HIP device: libcetkmc_hip has no CPU fallback");
    if (dev < 0 || dev >= ndev) return fail("device id out of range");
    HIPCHK(hipSetDevice(dev));
    Handle* h = new Handle();
    h->p = *p; h->kp = make_kparams(*p);
    h->L = L; h->Pk = next_pow2(L); h->PB = next_pow2(3 * L);
    h->RJ = round_up(L, SWEEP_TJ) + 4;
    h->pitchS = round_up(KOFF + L + 4, 16);
    h->pitchT = round_up(L, 8);      // rows of the f64 fields are 64-B multiples: a sweep lane reads 8 entries unconditionally
    if (const char* e = getenv("CETKMC_IFC_BLOCK")) h->ifc_block = atoi(e);
    h->pitchC = round_up(KOFFC + L + 12, 16);
    h->shmem_stream = (size_t)STREAM_SLOTS * (SWEEP_TJ + 4) * h->pitchC;
    if (h->shmem_stream > 64 * 1024 || (SWEEP_TJ + 4) * h->pitchC > 16 * 256 * STREAM_MAXPF) { delete h; return fail("L too large for the LDS ring"); }
    h->dev = dev; h->G = G; h->my_first = my_first;
    {   // the stepping loop's stream at the highest priority, the look-ahead stream at the lowest: its workgroups are
        // dispatched into what the main stream leaves free (the one-block selection kernels leave almost everything)
        int least = 0, greatest = 0;
        HIPCHK(hipDeviceGetStreamPriorityRange(&least, &greatest));
        HIPCHK(hipStreamCreateWithPriority(&h->stream, hipStreamNonBlocking, greatest));
        HIPCHK(hipStreamCreateWithPriority(&h->stream2, hipStreamNonBlocking, least));
    }
    HIPCHK(hipEventCreate(&h->ev0));
    HIPCHK(hipEventCreate(&h->ev1));
    HIPCHK(hipEventCreateWithFlags(&h->ev_main, hipEventDisableTiming));
    HIPCHK(hipEventCreateWithFlags(&h->ev_spec, hipEventDisableTiming));
    h->own_i0 = ranges.front().first;
    h->own_i1 = ranges.back().first + ranges.back().second;
    for (auto& r : ranges) {
        Slab s;
        s.v.L = L; s.v.gi0 = r.first; s.v.nloc = r.second;
        s.v.RJ = h->RJ; s.v.pitchS = h->pitchS; s.v.pitchT = h->pitchT; s.v.Pk = h->Pk; s.v.pitchC = h->pitchC;
        s.nS = (size_t)(r.second + 4) * h->RJ * h->pitchS;
        s.nT = (size_t)(r.second + 4) * L * h->pitchT;
        s.nC = (size_t)(r.second + 4) * h->RJ * h->pitchC;
        HIPCHK(hipMalloc((void**)&s.v.cls, s.nC));
        HIPCHK(hipMemsetAsync(s.v.cls, 0, s.nC, h->stream));
        HIPCHK(hipMalloc((void**)&s.v.state, s.nS));
        HIPCHK(hipMalloc((void**)&s.v.defects, s.nS));
        HIPCHK(hipMalloc((void**)&s.prev, s.nS));
        HIPCHK(hipMalloc((void**)&s.v.row_chg, (size_t)(s.v.nloc + 4) * L));
        HIPCHK(hipMemsetAsync(s.v.row_chg, 1, (size_t)(s.v.nloc + 4) * L, h->stream));
        HIPCHK(hipMemsetAsync(s.v.state, OOB, s.nS, h->stream));
        HIPCHK(hipMemsetAsync(s.v.defects, 0, s.nS, h->stream));
        HIPCHK(hipMemsetAsync(s.prev, OOB, s.nS, h->stream));
        for (int b = 0; b < 2; ++b) {
            HIPCHK(hipMalloc((void**)&s.Tbuf[b], s.nT * sizeof(double)));
            HIPCHK(hipMemsetAsync(s.Tbuf[b], 0, s.nT * sizeof(double), h->stream));
        }
        s.v.T = s.Tbuf[0];
        HIPCHK(hipMalloc((void**)&s.v.theta, s.nT * sizeof(double)));
        HIPCHK(hipMalloc((void**)&s.v.phi, s.nT * sizeof(double)));
        HIPCHK(hipMemsetAsync(s.v.theta, 0, s.nT * sizeof(double), h->stream));
        HIPCHK(hipMemsetAsync(s.v.phi, 0, s.nT * sizeof(double), h->stream));
        HIPCHK(hipMalloc((void**)&s.v.ovec, 3 * s.nT * sizeof(double)));
        for (int b = 0; b < 2; ++b) {
            HIPCHK(hipMalloc((void**)&s.vvalbuf[b], s.nT * sizeof(double)));
            HIPCHK(hipMemsetAsync(s.vvalbuf[b], 0, s.nT * sizeof(double), h->stream));
            HIPCHK(hipMalloc((void**)&s.depbuf[b], (size_t)L * h->pitchT * sizeof(double)));
            HIPCHK(hipMemsetAsync(s.depbuf[b], 0, (size_t)L * h->pitchT * sizeof(double), h->stream));
        }
        s.v.vval = s.vvalbuf[0]; s.v.dep_val = s.depbuf[0];
        HIPCHK(hipMalloc((void**)&s.v.ifc_in, s.nT));
        HIPCHK(hipMalloc((void**)&s.v.ifc_code, s.nT * sizeof(uint32_t)));
        HIPCHK(hipMemsetAsync(s.v.ifc_code, 0xFF, s.nT * sizeof(uint32_t), h->stream));
        HIPCHK(hipMalloc((void**)&s.v.ifc_list, (size_t)r.second * L * L * sizeof(uint32_t)));
        HIPCHK(hipMalloc((void**)&s.v.ifc_n, sizeof(int)));
        HIPCHK(hipMemsetAsync(s.v.ifc_in, 0, s.nT, h->stream));
        HIPCHK(hipMemsetAsync(s.v.ifc_n, 0, sizeof(int), h->stream));
        HIPCHK(hipMalloc((void**)&s.v.rowsum, (size_t)r.second * 3 * L * sizeof(double)));
        HIPCHK(hipMalloc((void**)&s.v.rowcnt, (size_t)r.second * 3 * L * sizeof(int32_t)));
        HIPCHK(hipMemsetAsync(s.v.rowsum, 0, (size_t)r.second * 3 * L * sizeof(double), h->stream));
        HIPCHK(hipMemsetAsync(s.v.rowcnt, 0, (size_t)r.second * 3 * L * sizeof(int32_t), h->stream));
        h->slabs.push_back(s);
    }
    for (int par = 0; par < 2; ++par) HIPCHK(hipMalloc((void**)&h->d_views[par], h->slabs.size() * sizeof(SlabView)));
    HIPCHK(hipMalloc((void**)&h->d_blocks, (size_t)PMAX * sizeof(BlockEnt)));
    HIPCHK(hipMemsetAsync(h->d_blocks, 0, (size_t)PMAX * sizeof(BlockEnt), h->stream));
    HIPCHK(hipMalloc((void**)&h->d_events_all, (size_t)G * sizeof(cetkmc_event)));
    HIPCHK(hipMemsetAsync(h->d_events_all, 0xFF, (size_t)G * sizeof(cetkmc_event), h->stream));   // type = -1
    HIPCHK(hipMalloc((void**)&h->d_ss, sizeof(StepState)));
    HIPCHK(hipMemsetAsync(h->d_ss, 0, sizeof(StepState), h->stream));
    HIPCHK(hipMalloc((void**)&h->d_dirty, (1 + 2 * DIRTY_MAX) * sizeof(int)));     // list + per-plane arrival counters
    HIPCHK(hipMemsetAsync(h->d_dirty, 0, (1 + 2 * DIRTY_MAX) * sizeof(int), h->stream));
    HIPCHK(hipMalloc((void**)&h->d_ktab, 225 * sizeof(double)));
    HIPCHK(hipMalloc((void**)&h->d_kp, sizeof(KParams)));
    HIPCHK(hipMalloc((void**)&h->d_flag, sizeof(int)));
    HIPCHK(hipMalloc((void**)&h->d_qtop, (size_t)L * L * sizeof(double)));
    CHK(upload_ktab(h));
    CHK(push_views(h));
    for (auto& s : h->slabs) hipLaunchKernelGGL(k_orient, dim3(1024), dim3(256), 0, h->stream, s.v);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(h->stream));
    *out = h;
    return 0;
}

SlabView view_of(Handle* h, int s, int par = -1)
{
    if (par < 0) par = h->cur;
    SlabView v = h->slabs[s].v;
    v.T = h->slabs[s].Tbuf[par]; v.vval = h->slabs[s].vvalbuf[par]; v.dep_val = h->slabs[s].depbuf[par];
    return v;
}

// per-call device allocation released on every return path
template <class T>
struct DevTmp {
    T* p = nullptr;
    DevTmp() = default;
    DevTmp(const DevTmp&) = delete;
    DevTmp& operator=(const DevTmp&) = delete;
    ~DevTmp() { if (p) (void)hipFree(p); }
    hipError_t alloc(size_t n) { return hipMalloc((void**)&p, n * sizeof(T)); }
    operator T*() const { return p; }
};

// interface list of one slab / every slab rebuilt from the membership flags, in address order
int relist_slab(Handle* h, int sl)
{
    SlabView v = view_of(h, sl);
    HIPCHK(hipMemsetAsync(v.ifc_n, 0, sizeof(int), h->stream));
    hipLaunchKernelGGL(k_ifc_relist, dim3((v.nloc * h->L + RELIST_ROWS - 1) / RELIST_ROWS), dim3(256), 0, h->stream, v, (const StepState*)nullptr);
    HIPCHK(hipGetLastError());
    return 0;
}
int relist(Handle* h)
{
    for (size_t sl = 0; sl < h->slabs.size(); ++sl) CHK(relist_slab(h, (int)sl));
    return 0;
}

// extended (owned + halo, clipped) global plane range of a slab
void ext_range(const Handle* h, const Slab& s, int* a, int* b)
{
    *a = std::max(0, s.v.gi0 - 2);
    *b = std::min(h->L, s.v.gi0 + s.v.nloc + 2);
}

// f64 field: host contiguous planes [i_begin,..) -> pitched device array, planes [a,b)
int h2d_f64(Handle* h, const Slab& s, double* dst, const double* src, int i_begin, int a, int b)
{
    const int L = h->L;
    const int li = a - (s.v.gi0 - 2);
    h->cnt.bytes_h2d += (int64_t)(b - a) * L * L * 8;
    HIPCHK(hipMemcpy2DAsync(dst + (size_t)li * L * h->pitchT, (size_t)h->pitchT * 8,
                            src + (size_t)(a - i_begin) * L * L, (size_t)L * 8, (size_t)L * 8,
                            (size_t)(b - a) * L, hipMemcpyHostToDevice, h->stream));
    return 0;
}
int d2h_f64(Handle* h, const Slab& s, const double* srcd, double* dst, int i_begin, int a, int b)
{
    const int L = h->L;
    const int li = a - (s.v.gi0 - 2);
    h->cnt.bytes_d2h += (int64_t)(b - a) * L * L * 8;
    HIPCHK(hipMemcpy2DAsync(dst + (size_t)(a - i_begin) * L * L, (size_t)L * 8,
                            srcd + (size_t)li * L * h->pitchT, (size_t)h->pitchT * 8, (size_t)L * 8,
                            (size_t)(b - a) * L, hipMemcpyDeviceToHost, h->stream));
    return 0;
}

template <class SRC>
int h2d_u8(Handle* h, const Slab& s, uint8_t* dst, const SRC* src, int i_begin, int a, int b, bool check, int with_cls = 0)
{
    const int L = h->L;
    const size_t n = (size_t)(b - a) * L * L;
    CHK(ensure_scratch(h, n * sizeof(SRC)));
    h->cnt.bytes_h2d += (int64_t)(n * sizeof(SRC));
    HIPCHK(hipMemcpyAsync(h->d_scratch, src + (size_t)(a - i_begin) * L * L, n * sizeof(SRC), hipMemcpyHostToDevice, h->stream));
    const int grid = (int)std::min<size_t>((n + 255) / 256, 4096);
    if (check) {
        if constexpr (sizeof(SRC) == 8) {
            HIPCHK(hipMemsetAsync(h->d_flag, 0, sizeof(int), h->stream));
            hipLaunchKernelGGL(k_check_range, dim3(grid), dim3(256), 0, h->stream, (const int64_t*)h->d_scratch, (int64_t)n, 0, 4, h->d_flag);
            int bad = 0;
            HIPCHK(hipMemcpyAsync(&bad, h->d_flag, sizeof(int), hipMemcpyDeviceToHost, h->stream));
            HIPCHK(hipStreamSynchronize(h->stream));
            if (bad) return fail("state/defects values must lie in 0..4 (constants.STATES)");
        }
    }
    hipLaunchKernelGGL(HIP_KERNEL_NAME(k_pack_u8<SRC>), dim3(grid), dim3(256), 0, h->stream, s.v, dst, (const SRC*)h->d_scratch, a, b - a, with_cls);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(h->stream));   // scratch is reused by the next field
    return 0;
}
template <class DST>
int d2h_u8(Handle* h, const Slab& s, const uint8_t* srcd, DST* dst, int i_begin, int a, int b)
{
    const int L = h->L;
    const size_t n = (size_t)(b - a) * L * L;
    CHK(ensure_scratch(h, n * sizeof(DST)));
    const int grid = (int)std::min<size_t>((n + 255) / 256, 4096);
    hipLaunchKernelGGL(HIP_KERNEL_NAME(k_unpack_u8<DST>), dim3(grid), dim3(256), 0, h->stream, s.v, srcd, (DST*)h->d_scratch, a, b - a);
    HIPCHK(hipGetLastError());
    h->cnt.bytes_d2h += (int64_t)(n * sizeof(DST));
    HIPCHK(hipMemcpyAsync(dst + (size_t)(a - i_begin) * L * L, h->d_scratch, n * sizeof(DST), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    return 0;
}

template <class I>
int upload_impl(Handle* h, int i_begin, int i_end, const I* state, const double* theta, const double* phi,
                const double* T, const I* defects)
{
    HIPCHK(hipSetDevice(h->dev));
    for (auto& s : h->slabs) {
        int a, b;
        ext_range(h, s, &a, &b);
        if (a < i_begin || b > i_end) return fail("upload range does not cover the slab's planes + halo");
        if (state) {
            CHK(h2d_u8<I>(h, s, s.v.state, state, i_begin, a, b, true, 1));
            HIPCHK(hipMemcpyAsync(s.prev, s.v.state, s.nS, hipMemcpyDeviceToDevice, h->stream));
            HIPCHK(hipMemsetAsync(s.v.row_chg, 0, (size_t)(s.v.nloc + 4) * h->L, h->stream));      // prev_state == state
            HIPCHK(hipMemsetAsync(s.v.ifc_in, 0, s.nT, h->stream));
            hipLaunchKernelGGL(k_ifc_rebuild, dim3(2048), dim3(256), 0, h->stream, s.v);
            CHK(relist_slab(h, (int)(&s - h->slabs.data())));
        }
        if (defects) CHK(h2d_u8<I>(h, s, s.v.defects, defects, i_begin, a, b, true));
        if (theta) CHK(h2d_f64(h, s, s.v.theta, theta, i_begin, a, b));
        if (phi) CHK(h2d_f64(h, s, s.v.phi, phi, i_begin, a, b));
        if (T) CHK(h2d_f64(h, s, s.Tbuf[h->cur], T, i_begin, a, b));
        if (theta || phi) hipLaunchKernelGGL(k_orient, dim3(1024), dim3(256), 0, h->stream, s.v);
        if (state) {
            int n = 0;
            HIPCHK(hipMemcpyAsync(&n, s.v.ifc_n, sizeof(int), hipMemcpyDeviceToHost, h->stream));
            HIPCHK(hipStreamSynchronize(h->stream));
            h->ifc_blocks = std::max(h->ifc_blocks, std::min(8192, (n + 255) / 256 + 64));
        }
    }
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(h->stream));
    h->swept = false;
    if (T) h->table_fresh = false;
    h->ifc_fresh = false;
    h->therm_applied_g = -1;
    return 0;
}
template <class I>
int download_impl(Handle* h, int i_begin, int i_end, I* state, double* theta, double* phi, double* T, I* defects)
{
    HIPCHK(hipSetDevice(h->dev));
    for (auto& s : h->slabs) {
        int a = std::max(s.v.gi0, i_begin), b = std::min(s.v.gi0 + s.v.nloc, i_end);
        if (a >= b) continue;
        if (state) CHK(d2h_u8<I>(h, s, s.v.state, state, i_begin, a, b));
        if (defects) CHK(d2h_u8<I>(h, s, s.v.defects, defects, i_begin, a, b));
        if (theta) CHK(d2h_f64(h, s, s.v.theta, theta, i_begin, a, b));
        if (phi) CHK(d2h_f64(h, s, s.v.phi, phi, i_begin, a, b));
        if (T) CHK(d2h_f64(h, s, s.Tbuf[h->cur], T, i_begin, a, b));
    }
    HIPCHK(hipStreamSynchronize(h->stream));
    return 0;
}

// ---- per-step launches (all asynchronous on h->stream) -----------------------------------
int launch_interface(Handle* h, bool batch, hipStream_t st, bool long_list = false)
{
    const StepState* ss = batch ? h->d_ss : nullptr;
    for (size_t s = 0; s < h->slabs.size(); ++s)
        if (long_list)
            hipLaunchKernelGGL(HIP_KERNEL_NAME(k_interface_part<1024>), dim3(2048), dim3(256), 0, st, h->kp, view_of(h, (int)s), h->d_ktab, ss);
        else
            hipLaunchKernelGGL(k_interface, dim3(h->ifc_blocks * (256 / h->ifc_block)), dim3(h->ifc_block), 0, st, h->kp,
                           view_of(h, (int)s), h->d_ktab, ss);
    HIPCHK(hipGetLastError());
    return 0;
}

// per-voxel rate table + plane L-1 deposition rates from the current temperature field (stale after an upload of T, a
// temperature update or a parameter change); it overwrites the listed voxels' entries too, so the interface kernel
// has to follow
int launch_table(Handle* h, hipStream_t st, const StepState* ss, int par)
{
    const double K0 = host_k_eff(h->p, 0, 0);
    for (size_t s = 0; s < h->slabs.size(); ++s) {
        SlabView v = view_of(h, (int)s, par);
        const int64_t pairs = (int64_t)v.nloc * v.L * (v.pitchT / 2);
        hipLaunchKernelGGL(k_rate_table, dim3((unsigned)std::min<int64_t>((pairs + 255) / 256, 8192)), dim3(256), 0, st, h->kp, v, K0, ss);
        h->cnt.alg_bytes_table += (int64_t)16 * v.nloc * v.L * v.L;     // T read, table entry written
    }
    HIPCHK(hipGetLastError());
    ++h->cnt.table_updates;
    return 0;
}
int ensure_table(Handle* h, hipStream_t st, const StepState* ss)
{
    if (h->table_fresh) return 0;
    CHK(launch_table(h, st, ss, h->cur));
    h->table_fresh = true;
    h->ifc_fresh = false;
    return 0;
}

StreamArgs stream_args(Handle* h, const SlabView& v)
{
    StreamArgs sa{};
    sa.T_melt = h->kp.T_melt; sa.delta_T_c = h->kp.delta_T_c; sa.kT = h->kp.kT; sa.I0 = h->kp.I0;
    sa.rate_threshold = h->kp.rate_threshold; sa.K0 = host_k_eff(h->p, 0, 0);
    sa.L = v.L; sa.gi0 = v.gi0; sa.nloc = v.nloc; sa.RJ = v.RJ; sa.pitchC = v.pitchC; sa.pitchT = v.pitchT; sa.Pk = v.Pk;
    sa.cls = v.cls; sa.T = v.T; sa.vval = v.vval; sa.dep_val = v.dep_val; sa.rowsum = v.rowsum; sa.rowcnt = v.rowcnt;
    sa.group_first = 0; sa.group_count = (v.nloc + STREAM_NI - 1) / STREAM_NI;
    return sa;
}

// the interface lists grow while stepping: keep the interface kernel's grid at one entry per thread (called where the
// host has just synchronised anyway)
int refresh_ifc_grid(Handle* h)
{
    for (auto& sl : h->slabs) {
        int n_list = 0;
        HIPCHK(hipMemcpy(&n_list, sl.v.ifc_n, sizeof(int), hipMemcpyDeviceToHost));
        h->ifc_blocks = std::max(h->ifc_blocks, std::min(8192, (n_list + 255) / 256 + 64));
    }
    return 0;
}

// ---- collectives: RCCL on the stream, or relayed through host callbacks (bring-up / test transport) ------
inline bool multi_rank(const Handle* h) { return h->comm != nullptr || h->hc.allgather != nullptr; }

// profile 2: one more hipEvent of the batch's collective timers, recorded on the stream now (nullptr when not timing)
int comm_stamp(Handle* h)
{
    if (!h->time_comm) return 0;
    if (h->comm_ev_used == h->comm_ev.size()) { hipEvent_t e; HIPCHK(hipEventCreate(&e)); h->comm_ev.push_back(e); }
    HIPCHK(hipEventRecord(h->comm_ev[h->comm_ev_used++], h->stream));
    return 0;
}

// in-place all-gather of `per` bytes per rank inside the device buffer `buf` (rank r's part at buf + r*per)
int comm_allgather_raw(Handle* h, void* buf, size_t per)
{
    if (h->comm) {
        NCCLCHK(g_rccl.AllGather((const char*)buf + per * h->rank, buf, per, ncclChar, h->comm, h->stream));
        return 0;
    }
    if (!h->hc.allgather) return 0;
    h->hc_stage.resize(std::max(h->hc_stage.size(), per * h->nranks));
    char* host = h->hc_stage.data();
    HIPCHK(hipMemcpyAsync(host + per * h->rank, (const char*)buf + per * h->rank, per, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    if (h->hc.allgather(h->hc.user, host + per * h->rank, host, (int64_t)per)) return fail("host all-gather callback failed");
    HIPCHK(hipMemcpyAsync(buf, host, per * h->nranks, hipMemcpyHostToDevice, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    return 0;
}
int comm_allgather(Handle* h, void* buf, size_t per)
{
    CHK(comm_stamp(h));
    CHK(comm_allgather_raw(h, buf, per));
    return comm_stamp(h);
}

// Incremental step (run_steps with incremental = 1, between temperature updates): only the rows the
// previous event made stale are re-evaluated; their planes' block sums follow.
int launch_dirty_rows(Handle* h, hipEvent_t ev_a, hipEvent_t ev_b)
{
    if (ev_a) HIPCHK(hipEventRecord(ev_a, h->stream));
    // dirty rows re-evaluated, their planes' block sums reduced by the last-arriving block of each plane (one launch;
    // the arrival counters live behind the dirty list and are zero at rest)
    for (size_t s = 0; s < h->slabs.size(); ++s) {
        SlabView v = view_of(h, (int)s);
        const StreamArgs sa = stream_args(h, v);
        const int* dl = h->d_dirty;
        int* pc = h->d_dirty + 1 + DIRTY_MAX;
        const StepState* ss = h->d_ss;
        const bool tab = h->sweep_variant == 1 || h->sweep_variant >= 3, hw = h->Pk <= 256, ch2 = h->Pk > 512;
#define CETKMC_LAUNCH_ROWS(TAB, HW, CH2) \
    hipLaunchKernelGGL(HIP_KERNEL_NAME(k_rows_eval<TAB, HW, CH2>), dim3(24), dim3(256), 0, h->stream, sa, dl, ss, h->d_blocks, pc)
        if (hw) { if (tab) CETKMC_LAUNCH_ROWS(true, true, false); else CETKMC_LAUNCH_ROWS(false, true, false); }
        else if (!ch2) { if (tab) CETKMC_LAUNCH_ROWS(true, false, false); else CETKMC_LAUNCH_ROWS(false, false, false); }
        else { if (tab) CETKMC_LAUNCH_ROWS(true, false, true); else CETKMC_LAUNCH_ROWS(false, false, true); }
#undef CETKMC_LAUNCH_ROWS
    }
    if (ev_b) HIPCHK(hipEventRecord(ev_b, h->stream));
    HIPCHK(hipGetLastError());
    if (multi_rank(h)) CHK(comm_allgather(h, h->d_blocks, (size_t)3 * (h->L / h->nranks) * sizeof(BlockEnt)));
    h->swept = true;
    return 0;
}

// One full-lattice rate sweep: (rate table, if T changed) -> (interface kernel, if the listed voxels' sums are stale)
// -> sweep kernel -> block sums (+ all-gather across ranks).  ev_pre | table + interface | ev_a | sweep | ev_b |
// reduce | ev_post are the phase boundaries of cetkmc_get_counters.
int launch_sweep(Handle* h, bool batch, hipEvent_t ev_a = nullptr, hipEvent_t ev_b = nullptr, bool long_list = false,
                 hipEvent_t ev_pre = nullptr, hipEvent_t ev_post = nullptr)
{
    ++h->cnt.sweeps;
    // streaming kernels: class u8 + one f64 (rate table / T) per voxel; simple kernel: state tile + T
    for (auto& sl : h->slabs) h->cnt.alg_bytes_sweep += (int64_t)9 * sl.v.nloc * h->L * h->L;
    if (ev_pre) HIPCHK(hipEventRecord(ev_pre, h->stream));
    const int TR = SWEEP_TJ + 4;
    const size_t shmem0 = (size_t)((5 * TR * h->pitchS + 15) & ~15) + 225 * sizeof(double);
    const StepState* ss = batch ? h->d_ss : nullptr;
    const int njt = (h->L + SWEEP_TJ - 1) / SWEEP_TJ;
    if (h->sweep_variant >= 1) {
        CHK(ensure_table(h, h->stream, ss));
        if (!h->ifc_fresh || h->ifc_every_step) {
            CHK(launch_interface(h, batch, h->stream, long_list));
            ++h->cnt.interface_launches;
            h->ifc_fresh = true;
        }
    }
    // sweep timing of a batch (profile 1 / 3: ev_a, ev_b without the phase events): the two hipEvents ride on the launch itself
    // (hipExtLaunchKernelGGL start / stop events = the dispatch's own begin / end timestamps) -- no barrier packets in the
    // stream, so the timed steps run like untimed ones.  Several slabs in one process / the phase table: plain records.
    const bool ext = ev_a && ev_b && !ev_pre && h->slabs.size() == 1 && h->sweep_variant >= 1;
    if (ev_a && !ext) HIPCHK(hipEventRecord(ev_a, h->stream));
    const int sv = (h->sweep_auto && h->sweep_variant == 1 && h->L <= 128) ? 4 : h->sweep_variant;      // effective sweep kernel
    for (size_t s = 0; s < h->slabs.size(); ++s) {
        SlabView v = view_of(h, (int)s);
        if (sv == 4) {
            // census-free table sweep, one 16-wave block per owned plane, block sums folded in the same launch
            const StreamArgs sa = stream_args(h, v);
            const bool hw = h->Pk <= 256, ch2 = h->Pk > 512;
            const dim3 g((unsigned)v.nloc);
            const uint32_t shm = (uint32_t)(3 * h->L * (sizeof(double) + sizeof(int)));
#define CETKMC_LAUNCH_PLANE(HW, CH2)                                                                                             \
    do {                                                                                                                         \
        if (ext) hipExtLaunchKernelGGL(HIP_KERNEL_NAME(k_sweep_plane<HW, CH2>), g, dim3(1024), shm, h->stream, ev_a, ev_b, 0, sa, ss, h->d_blocks); \
        else hipLaunchKernelGGL(HIP_KERNEL_NAME(k_sweep_plane<HW, CH2>), g, dim3(1024), shm, h->stream, sa, ss, h->d_blocks);     \
    } while (0)
            if (hw) CETKMC_LAUNCH_PLANE(true, false);
            else if (!ch2) CETKMC_LAUNCH_PLANE(false, false);
            else CETKMC_LAUNCH_PLANE(false, true);
#undef CETKMC_LAUNCH_PLANE
        } else if (sv == 3) {
            // census-free table sweep: class bytes + rate table streamed once, no LDS
            const StreamArgs sa = stream_args(h, v);
            const bool hw = h->Pk <= 256, ch2 = h->Pk > 512;
            const int ipp = hw ? (h->L + 1) / 2 : h->L;
            const int64_t n_items = (int64_t)ipp * v.nloc;
            const dim3 g((unsigned)((n_items + 4 * TABLE_IPW - 1) / (4 * TABLE_IPW)));
#define CETKMC_LAUNCH_TABLE(HW, CH2)                                                                                             \
    do {                                                                                                                         \
        if (ext) hipExtLaunchKernelGGL(HIP_KERNEL_NAME(k_sweep_table<HW, CH2>), g, dim3(256), 0, h->stream, ev_a, ev_b, 0, sa, ss); \
        else hipLaunchKernelGGL(HIP_KERNEL_NAME(k_sweep_table<HW, CH2>), g, dim3(256), 0, h->stream, sa, ss);                     \
    } while (0)
            if (hw) CETKMC_LAUNCH_TABLE(true, false);
            else if (!ch2) CETKMC_LAUNCH_TABLE(false, false);
            else CETKMC_LAUNCH_TABLE(false, true);
#undef CETKMC_LAUNCH_TABLE
        } else if (h->sweep_variant >= 1) {
            const StreamArgs sa = stream_args(h, v);
            const dim3 g(sa.group_count * njt);
            const bool tab = h->sweep_variant == 1, hw = h->Pk <= 256;
            const int npf = ((SWEEP_TJ + 4) * h->pitchC / 16 + 255) / 256;       // 16-B chunks of a class slab per thread
#define CETKMC_LAUNCH_STREAM(TAB, HW, NPF, CH2)                                                                                   \
    do {                                                                                                                         \
        if (ext) hipExtLaunchKernelGGL(HIP_KERNEL_NAME(k_sweep_stream<TAB, HW, NPF, CH2>), g, dim3(256), (uint32_t)h->shmem_stream, \
                                       h->stream, ev_a, ev_b, 0, sa, ss);                                                        \
        else hipLaunchKernelGGL(HIP_KERNEL_NAME(k_sweep_stream<TAB, HW, NPF, CH2>), g, dim3(256), h->shmem_stream, h->stream, sa, ss); \
    } while (0)
            const bool ch2 = h->Pk > 512;       // L > 512: pitchC >= 544, so npf >= 2
            if (hw) { if (tab) CETKMC_LAUNCH_STREAM(true, true, 1, false); else CETKMC_LAUNCH_STREAM(false, true, 1, false); }
            else if (npf == 1) { if (tab) CETKMC_LAUNCH_STREAM(true, false, 1, false); else CETKMC_LAUNCH_STREAM(false, false, 1, false); }
            else if (npf == 2 && !ch2) { if (tab) CETKMC_LAUNCH_STREAM(true, false, 2, false); else CETKMC_LAUNCH_STREAM(false, false, 2, false); }
            else if (npf == 2) { if (tab) CETKMC_LAUNCH_STREAM(true, false, 2, true); else CETKMC_LAUNCH_STREAM(false, false, 2, true); }
            else { if (tab) CETKMC_LAUNCH_STREAM(true, false, 3, true); else CETKMC_LAUNCH_STREAM(false, false, 3, true); }
#undef CETKMC_LAUNCH_STREAM
        } else {
            hipLaunchKernelGGL(k_sweep_simple, dim3(v.nloc * njt), dim3(256), shmem0, h->stream, h->kp, v, h->d_ktab, ss);
        }
    }
    if (ev_b && !ext) HIPCHK(hipEventRecord(ev_b, h->stream));
    if (sv != 4)          // (variant 4 folds the block sums in the sweep launch)
        for (size_t s = 0; s < h->slabs.size(); ++s) {
            SlabView v = view_of(h, (int)s);
            hipLaunchKernelGGL(k_plane_reduce, dim3(3 * v.nloc), dim3(64), 0, h->stream, v, h->d_blocks, ss);
        }
    HIPCHK(hipGetLastError());
    if (multi_rank(h)) CHK(comm_allgather(h, h->d_blocks, (size_t)3 * (h->L / h->nranks) * sizeof(BlockEnt)));
    if (ev_post) HIPCHK(hipEventRecord(ev_post, h->stream));
    h->swept = true;
    return 0;
}

int launch_select(Handle* h, const BatchCfg& cfg, double r_direct, int info_only)
{
    hipLaunchKernelGGL(k_select, dim3(1), dim3(256), 0, h->stream, h->kp, (const SlabView*)h->d_views[h->cur],
                       (int)h->slabs.size(), h->L, h->PB, (const BlockEnt*)h->d_blocks, h->d_ss, cfg,
                       (const double*)h->d_u_pick, r_direct, (const double*)h->d_ktab,
                       h->d_events_all + h->my_first, info_only, h->sweep_variant >= 1 ? 1 : 0);
    HIPCHK(hipGetLastError());
    if (multi_rank(h) && !info_only) CHK(comm_allgather(h, h->d_events_all, sizeof(cetkmc_event)));
    return 0;
}

// selection + application of one batched step: one fused launch in a single process, select / all-gather / apply
// across ranks
int launch_select_apply(Handle* h, const BatchCfg& cfg, int eval_touched, int* dirty, int64_t cur_hint)
{
    if (!multi_rank(h)) {
#define CETKMC_LAUNCH_SELAPPLY(IFC)                                                                                              \
    hipLaunchKernelGGL(HIP_KERNEL_NAME(k_select_apply<IFC>), dim3(1), dim3(256), 0, h->stream, h->kp, (const SlabView*)h->d_views[h->cur], \
                       (int)h->slabs.size(), h->L, h->PB, (const BlockEnt*)h->d_blocks, h->d_ss, cfg,                               \
                       (const double*)h->d_u_pick, (const double*)h->d_ktab, h->d_events_all + h->my_first,                         \
                       IFC ? 1 : 0, (const double*)h->d_u_defect, (const double*)h->d_u_np,                                         \
                       h->d_log_total, h->d_log_event, h->d_log_nev, eval_touched, dirty, (long long)cur_hint)
        if (h->sweep_variant >= 1) CETKMC_LAUNCH_SELAPPLY(true); else CETKMC_LAUNCH_SELAPPLY(false);
#undef CETKMC_LAUNCH_SELAPPLY
        HIPCHK(hipGetLastError());
        return 0;
    }
    CHK(launch_select(h, cfg, 0.0, 0));
    hipLaunchKernelGGL(k_apply_batch, dim3(1), dim3(64), 0, h->stream, h->kp, (const SlabView*)h->d_views[h->cur],
                       (int)h->slabs.size(), h->L, (const cetkmc_event*)h->d_events_all, h->G, h->d_ss, cfg,
                       (const double*)h->d_u_defect, (const double*)h->d_u_np, h->d_log_total, h->d_log_event,
                       h->d_log_nev, (const double*)h->d_ktab, eval_touched, dirty);
    HIPCHK(hipGetLastError());
    return 0;
}

// neighbour exchange along the slab axis: `bytes` from send_lo to rank-1 / send_hi to rank+1 (device pointers), the
// neighbours' counterparts into recv_lo / recv_hi.  End ranks skip the missing side (its recv buffer is left alone).
int comm_exchange_raw(Handle* h, const void* send_lo, void* recv_lo, const void* send_hi, void* recv_hi, size_t bytes);
int comm_exchange(Handle* h, const void* send_lo, void* recv_lo, const void* send_hi, void* recv_hi, size_t bytes)
{
    if (h->nranks <= 1) return 0;
    CHK(comm_stamp(h));
    CHK(comm_exchange_raw(h, send_lo, recv_lo, send_hi, recv_hi, bytes));
    return comm_stamp(h);
}
int comm_exchange_raw(Handle* h, const void* send_lo, void* recv_lo, const void* send_hi, void* recv_hi, size_t bytes)
{
    if (h->nranks <= 1) return 0;
    const bool lo = h->rank > 0, hi = h->rank < h->nranks - 1;
    if (h->comm) {
        NCCLCHK(g_rccl.GroupStart());
        if (lo) {
            NCCLCHK(g_rccl.Send(send_lo, bytes, ncclChar, h->rank - 1, h->comm, h->stream));
            NCCLCHK(g_rccl.Recv(recv_lo, bytes, ncclChar, h->rank - 1, h->comm, h->stream));
        }
        if (hi) {
            NCCLCHK(g_rccl.Send(send_hi, bytes, ncclChar, h->rank + 1, h->comm, h->stream));
            NCCLCHK(g_rccl.Recv(recv_hi, bytes, ncclChar, h->rank + 1, h->comm, h->stream));
        }
        NCCLCHK(g_rccl.GroupEnd());
    } else if (h->hc.exchange) {
        h->hc_stage.resize(std::max(h->hc_stage.size(), 4 * bytes));
        char* host = h->hc_stage.data();           // [send_lo | send_hi | recv_lo | recv_hi]
        if (lo) HIPCHK(hipMemcpyAsync(host, send_lo, bytes, hipMemcpyDeviceToHost, h->stream));
        if (hi) HIPCHK(hipMemcpyAsync(host + bytes, send_hi, bytes, hipMemcpyDeviceToHost, h->stream));
        HIPCHK(hipStreamSynchronize(h->stream));
        if (h->hc.exchange(h->hc.user, lo ? h->rank - 1 : -1, hi ? h->rank + 1 : -1, host, host + 2 * bytes, host + bytes, host + 3 * bytes,
                           (int64_t)bytes))
            return fail("host neighbour-exchange callback failed");
        if (lo) HIPCHK(hipMemcpyAsync(recv_lo, host + 2 * bytes, bytes, hipMemcpyHostToDevice, h->stream));
        if (hi) HIPCHK(hipMemcpyAsync(recv_hi, host + 3 * bytes, bytes, hipMemcpyHostToDevice, h->stream));
        HIPCHK(hipStreamSynchronize(h->stream));
    }
    return 0;
}

int exchange_T_halo(Handle* h, int buf, hipStream_t st = nullptr)
{
    if (!st) st = h->stream;
    const size_t plane = (size_t)h->L * h->pitchT;   // doubles
    for (size_t s = 0; s + 1 < h->slabs.size(); ++s) {
        Slab& a = h->slabs[s];
        Slab& b = h->slabs[s + 1];
        // a's top two owned planes -> b's lower halo; b's bottom two owned planes -> a's upper halo
        HIPCHK(hipMemcpyAsync(b.Tbuf[buf], a.Tbuf[buf] + plane * a.v.nloc, 2 * plane * 8, hipMemcpyDeviceToDevice, st));
        HIPCHK(hipMemcpyAsync(a.Tbuf[buf] + plane * (a.v.nloc + 2), b.Tbuf[buf] + plane * 2, 2 * plane * 8, hipMemcpyDeviceToDevice, st));
    }
    if (multi_rank(h) && h->nranks > 1) {
        // my bottom / top two owned planes -> the neighbours' halos; theirs -> mine (2 L^2 doubles each way, one xGMI link per side)
        Slab& s = h->slabs[0];
        double* T = s.Tbuf[buf];
        CHK(comm_exchange(h, T + plane * 2, T, T + plane * s.v.nloc, T + plane * (s.v.nloc + 2), 2 * plane * 8));
    }
    return 0;
}

int launch_thermal(Handle* h, double dt, int laser, const double* d_q, int use_latent, int scrub, bool batch)
{
    ThermalCfg C{};
    C.dt = dt; C.alpha = h->p.alpha; C.inv_dx2 = h->p.inv_dx2; C.clip_lo = h->p.T_clip_lo; C.clip_hi = h->p.T_clip_hi;
    C.T_nan = h->p.T_nan; C.rho_cp = h->p.rho_cp; C.latent_coef = h->p.latent_coef;
    C.laser = laser; C.use_latent = use_latent; C.scrub = scrub; C.ni = h->therm_ni;
    ++h->cnt.thermal_updates;
    for (auto& sl : h->slabs) h->cnt.alg_bytes_thermal += (int64_t)16 * sl.v.nloc * h->L * h->L;   // T read + written
    const int nxt = h->cur ^ 1;
    size_t n_fused = 0;
    for (size_t s = 0; s < h->slabs.size(); ++s) {
        SlabView v = view_of(h, (int)s);
        if (h->thermal_variant == 1) {
            dim3 grid((h->L + THERM_KT - 1) / THERM_KT, (h->L + THERM_TJ - 1) / THERM_TJ, (v.nloc + h->therm_ni - 1) / h->therm_ni);
            const StepState* ssp = batch ? (const StepState*)h->d_ss : nullptr;
            const double* Tin = h->slabs[s].Tbuf[h->cur];
            double* Tout = h->slabs[s].Tbuf[nxt];
            uint8_t* prev = h->slabs[s].prev;
            if (h->L % THERM_KT == 0 && h->L % THERM16_TJ == 0 && !h->thermal_general && h->thermal_tiles16) {   // 16-row tiles (option), covering the lattice exactly
                ThermalCfg C16 = C;
                C16.ni = h->therm_ni16;
                dim3 g16(h->L / h->thermal_kt, h->L / THERM16_TJ, (v.nloc + C16.ni - 1) / C16.ni);
                // the default tiles also write the new field's rate table (k_rate_table's work, without reading T again)
                const bool fuse = h->thermal_table && h->thermal_rpt == 2 && h->thermal_kt == 256 && h->sweep_variant >= 1;
                TableCfg TB{};
                TB.T_melt = h->kp.T_melt; TB.delta_T_c = h->kp.delta_T_c; TB.kT = h->kp.kT; TB.I0 = h->kp.I0;
                TB.rate_threshold = h->kp.rate_threshold; TB.K0 = host_k_eff(h->p, 0, 0); TB.nu_dep = h->kp.nu_dep;
                TB.vval = h->slabs[s].vvalbuf[nxt]; TB.dep_val = h->slabs[s].depbuf[nxt];
                if (fuse) { ++n_fused; h->cnt.alg_bytes_table += (int64_t)8 * v.nloc * v.L * v.L; }      // table entry written (T not re-read)
#define CETKMC_LAUNCH_T16(LA, LT)                                                                                                 \
    do {                                                                                                                          \
        if (fuse) hipLaunchKernelGGL((k_thermal_tiles16<LA, LT, 2, 256, true>), g16, dim3(1024), 0, h->stream, v, Tin, Tout, prev, d_q, C16, ssp, TB); \
        else if (h->thermal_rpt == 2 && h->thermal_kt == 128) hipLaunchKernelGGL((k_thermal_tiles16<LA, LT, 2, 128>), g16, dim3(512), 0, h->stream, v, Tin, Tout, prev, d_q, C16, ssp, TableCfg{}); \
        else if (h->thermal_rpt == 2) hipLaunchKernelGGL((k_thermal_tiles16<LA, LT, 2, 256>), g16, dim3(1024), 0, h->stream, v, Tin, Tout, prev, d_q, C16, ssp, TableCfg{}); \
        else hipLaunchKernelGGL((k_thermal_tiles16<LA, LT, 4, 256>), g16, dim3(512), 0, h->stream, v, Tin, Tout, prev, d_q, C16, ssp, TableCfg{});   \
    } while (0)
                if (laser && use_latent) CETKMC_LAUNCH_T16(true, true);
                else if (laser) CETKMC_LAUNCH_T16(true, false);
                else CETKMC_LAUNCH_T16(false, false);
#undef CETKMC_LAUNCH_T16
            } else if (h->L % THERM_KT == 0 && h->L % THERM_TJ == 0 && !h->thermal_general) {      // the 8-row tiles cover the lattice exactly
                if (laser && use_latent) hipLaunchKernelGGL((k_thermal_tiles<true, true>), grid, dim3(256), 0, h->stream, v, Tin, Tout, prev, d_q, C, ssp);
                else if (laser) hipLaunchKernelGGL((k_thermal_tiles<true, false>), grid, dim3(256), 0, h->stream, v, Tin, Tout, prev, d_q, C, ssp);
                else hipLaunchKernelGGL((k_thermal_tiles<false, false>), grid, dim3(256), 0, h->stream, v, Tin, Tout, prev, d_q, C, ssp);
            } else {
                hipLaunchKernelGGL(k_thermal_march, grid, dim3(256), 0, h->stream, v, Tin, Tout, prev, d_q, C, ssp);
            }
        } else {
            dim3 grid((h->L + 255) / 256, h->L, v.nloc);
            hipLaunchKernelGGL(k_thermal, grid, dim3(256), 0, h->stream, v, (const double*)h->slabs[s].Tbuf[h->cur],
                               h->slabs[s].Tbuf[nxt], (const uint8_t*)h->slabs[s].prev, d_q, C,
                               batch ? (const StepState*)h->d_ss : nullptr);
        }
    }
    HIPCHK(hipGetLastError());
    CHK(exchange_T_halo(h, nxt));
    if (laser && use_latent) {
        for (auto& s : h->slabs) {
            if (h->thermal_variant != 1)      // the marching kernel brings prev_state level with state itself (changed rows)
                HIPCHK(hipMemcpyAsync(s.prev, s.v.state, s.nS, hipMemcpyDeviceToDevice, h->stream));
            hipLaunchKernelGGL(k_clear_row_flags, dim3(64), dim3(256), 0, h->stream, s.v, batch ? (const StepState*)h->d_ss : nullptr);
        }
        HIPCHK(hipGetLastError());
    }
    h->cur = nxt;
    h->swept = false;
    h->table_fresh = n_fused == h->slabs.size();      // every slab's update wrote its table as well
    if (h->table_fresh) { h->ifc_fresh = false; ++h->cnt.table_updates; }       // the listed voxels' entries follow (k_interface)
    h->therm_applied_g = -1;
    return 0;
}

ThermalCfg thermal_cfg(Handle* h, double dt, int laser, int use_latent, int scrub)
{
    ThermalCfg C{};
    C.dt = dt; C.alpha = h->p.alpha; C.inv_dx2 = h->p.inv_dx2; C.clip_lo = h->p.T_clip_lo; C.clip_hi = h->p.T_clip_hi;
    C.T_nan = h->p.T_nan; C.rho_cp = h->p.rho_cp; C.latent_coef = h->p.latent_coef;
    C.laser = laser; C.use_latent = use_latent; C.scrub = scrub; C.ni = h->therm_ni;
    return C;
}

// Look-ahead: the temperature update that global step g will perform, and the rate table of its result, computed NOW on
// stream2 from the current field into the other buffer pair -- without the latent-heat term (k_thermal_fix adds it at
// the update, for the few voxels it concerns).  The current field does not change until then, so this is the update's
// own arithmetic, merely scheduled into the idle time of the 19 steps in between (the one-block selection kernels
// leave the chip almost empty).  Single process only (the T halo of a multi-rank run travels on the main stream).
int launch_thermal_ahead(Handle* h, int64_t g, double dt, int laser, const double* d_q, int use_latent, int scrub)
{
    h->spec.valid = false;
    if (!h->thermal_ahead || h->thermal_variant != 1 || (multi_rank(h) && h->nranks > 1)) return 0;
    const ThermalCfg C = thermal_cfg(h, dt, laser, 0, scrub);
    const int nxt = h->cur ^ 1;
    HIPCHK(hipEventRecord(h->ev_main, h->stream));
    HIPCHK(hipStreamWaitEvent(h->stream2, h->ev_main, 0));
    for (size_t s = 0; s < h->slabs.size(); ++s) {
        SlabView v = view_of(h, (int)s);
        dim3 grid((h->L + THERM_KT - 1) / THERM_KT, (h->L + THERM_TJ - 1) / THERM_TJ, (v.nloc + h->therm_ni - 1) / h->therm_ni);
        hipLaunchKernelGGL(k_thermal_march, grid, dim3(256), 0, h->stream2, v, (const double*)h->slabs[s].Tbuf[h->cur],
                           h->slabs[s].Tbuf[nxt], h->slabs[s].prev, d_q, C, (const StepState*)h->d_ss);
    }
    HIPCHK(hipGetLastError());
    CHK(exchange_T_halo(h, nxt, h->stream2));
    CHK(launch_table(h, h->stream2, (const StepState*)h->d_ss, nxt));
    HIPCHK(hipEventRecord(h->ev_spec, h->stream2));
    h->spec.valid = true; h->spec.g = g; h->spec.laser = laser; h->spec.use_latent = use_latent; h->spec.scrub = scrub;
    h->spec.dt = dt; h->spec.d_q = d_q;
    return 0;
}

// the temperature update of global step g inside a batch: the look-ahead result if it stands for exactly this update
// (then only k_thermal_fix runs on the main stream), else the synchronous kernels
int thermal_step(Handle* h, int64_t g, double dt, int laser, const double* d_q, int use_latent, int scrub)
{
    Handle::Spec& sp = h->spec;
    if (!(sp.valid && sp.g == g && sp.laser == laser && sp.use_latent == use_latent && sp.scrub == scrub && sp.dt == dt && sp.d_q == d_q)) {
        if (sp.valid) { HIPCHK(hipStreamSynchronize(h->stream2)); sp.valid = false; }     // a stale look-ahead still owns the other buffers
        return launch_thermal(h, dt, laser, d_q, use_latent, scrub, true);
    }
    sp.valid = false;
    const ThermalCfg C = thermal_cfg(h, dt, laser, use_latent, scrub);
    const int nxt = h->cur ^ 1;
    const double K0 = host_k_eff(h->p, 0, 0);
    ++h->cnt.thermal_updates;
    HIPCHK(hipStreamWaitEvent(h->stream, h->ev_spec, 0));
    for (size_t s = 0; s < h->slabs.size(); ++s) {
        Slab& sl = h->slabs[s];
        SlabView v = view_of(h, (int)s);
        h->cnt.alg_bytes_thermal += (int64_t)16 * v.nloc * h->L * h->L;
        hipLaunchKernelGGL(k_thermal_fix, dim3(1024), dim3(256), 0, h->stream, h->kp, v, (const double*)sl.Tbuf[h->cur], sl.Tbuf[nxt],
                           sl.prev, d_q, C, (const StepState*)h->d_ss, (const double*)sl.vvalbuf[h->cur], sl.vvalbuf[nxt],
                           (const double*)sl.depbuf[h->cur], sl.depbuf[nxt], K0);
        if (laser && use_latent)
            hipLaunchKernelGGL(k_clear_row_flags, dim3(64), dim3(256), 0, h->stream, v, (const StepState*)h->d_ss);
    }
    HIPCHK(hipGetLastError());
    if (h->slabs.size() > 1) CHK(exchange_T_halo(h, nxt));      // the recomputed voxels may sit in a neighbour slab's halo
    h->cur = nxt;
    h->swept = false;
    h->table_fresh = true;       // computed ahead with the field
    h->ifc_fresh = false;        // the listed voxels' entries are not: the interface kernel follows
    return 0;
}

template <class T>
int grow(T** p, size_t* cap, size_t need)
{
    if (need <= *cap && *p) return 0;
    if (*p) HIPCHK(hipFree(*p));
    *p = nullptr;
    size_t n = std::max<size_t>(need, 16);
    HIPCHK(hipMalloc((void**)p, n * sizeof(T)));
    *cap = n;
    return 0;
}

int grow_pinned(char** p, size_t* cap, size_t need)
{
    if (*cap >= need) return 0;
    if (*p) { HIPCHK(hipHostFree(*p)); *p = nullptr; *cap = 0; }
    const size_t c = std::max<size_t>(need + need / 2, 1 << 16);
    HIPCHK(hipHostMalloc((void**)p, c, hipHostMallocDefault));
    *cap = c;
    return 0;
}
constexpr size_t PIN_SMALL_MAX = 1 << 20;      // larger arrays (long u_np streams, many source planes) are copied directly

void destroy_impl(Handle* h)
{
    if (!h) return;
    (void)hipSetDevice(h->dev);
    if (h->stream) (void)hipStreamSynchronize(h->stream);
    if (h->stream2) (void)hipStreamSynchronize(h->stream2);
    if (h->comm && g_rccl.CommDestroy) g_rccl.CommDestroy(h->comm);
    for (auto& s : h->slabs) {
        (void)hipFree(s.v.state); (void)hipFree(s.v.defects); (void)hipFree(s.prev); (void)hipFree(s.v.row_chg); (void)hipFree(s.v.cls);
        (void)hipFree(s.Tbuf[0]); (void)hipFree(s.Tbuf[1]); (void)hipFree(s.v.theta); (void)hipFree(s.v.phi); (void)hipFree(s.v.ovec);
        (void)hipFree(s.v.rowsum); (void)hipFree(s.v.rowcnt);
        (void)hipFree(s.vvalbuf[0]); (void)hipFree(s.vvalbuf[1]); (void)hipFree(s.depbuf[0]); (void)hipFree(s.depbuf[1]); (void)hipFree(s.v.ifc_in); (void)hipFree(s.v.ifc_code); (void)hipFree(s.v.ifc_list); (void)hipFree(s.v.ifc_n);
    }
    void* ccp[] = {h->d_cc_parent, h->d_cc_roots, h->d_cc_cid, h->d_cc_labels, h->d_cc_stats, h->d_cc_n};
    for (void* p : ccp) if (p) (void)hipFree(p);
    void* ptrs[] = {h->d_views[0], h->d_views[1], h->d_blocks, h->d_events_all, h->d_ss, h->d_dirty, h->d_ktab, h->d_kp, h->d_scratch,
                    h->d_flag, h->d_qtop, h->d_u_pick, h->d_u_defect, h->d_u_np, h->d_q, h->d_log_total,
                    h->d_log_event, h->d_log_nev, h->d_sup_dom, h->d_sup_picks, h->d_sup_cnt, h->d_sup_log, h->d_sup_rmax};
    for (void* p : ptrs) if (p) (void)hipFree(p);
    if (h->pin_in) (void)hipHostFree(h->pin_in);
    if (h->pin_out) (void)hipHostFree(h->pin_out);
    for (auto e : h->prof) (void)hipEventDestroy(e);
    for (auto e : h->comm_ev) (void)hipEventDestroy(e);
    if (h->ev0) (void)hipEventDestroy(h->ev0);
    if (h->ev1) (void)hipEventDestroy(h->ev1);
    if (h->ev_main) (void)hipEventDestroy(h->ev_main);
    if (h->ev_spec) (void)hipEventDestroy(h->ev_spec);
    if (h->stream2) (void)hipStreamDestroy(h->stream2);
    if (h->stream) (void)hipStreamDestroy(h->stream);
    delete h;
}

}  // namespace

// =========================================================================================
extern "C" {

const char* cetkmc_last_error(void) { return g_err.c_str(); }
int cetkmc_abi_version(void) { return CETKMC_ABI_VERSION; }
#ifndef CETKMC_SRC_HASH
#define CETKMC_SRC_HASH "unknown"
#endif
// the hash follows a marker so that the binding can read it from the file without loading the library
static const char g_src_hash[] = "cetkmc-source-hash=" CETKMC_SRC_HASH;
const char* cetkmc_source_hash(void) { return g_src_hash + 19; }

int cetkmc_struct_size(const char* name)
{
    if (!name) return -1;
#define SZ(n, t) if (!strcmp(name, n)) return (int)sizeof(t)
    SZ("params", cetkmc_params); SZ("event", cetkmc_event); SZ("sweep_info", cetkmc_sweep_info);
    SZ("run_args", cetkmc_run_args); SZ("run_result", cetkmc_run_result); SZ("super_args", cetkmc_super_args);
    SZ("counters", cetkmc_counters); SZ("host_comm", cetkmc_host_comm);
#undef SZ
    return -1;
}

int cetkmc_device_count(int* n)
{
    if (!n) return fail("null argument");
    int c = 0;
    hipError_t e = hipGetDeviceCount(&c);
    if (e != hipSuccess) { *n = 0; return fail(std::string("hipGetDeviceCount: ") + hipGetErrorString(e)); }
    *n = c;
    return 0;
}

int cetkmc_create(const cetkmc_params* p, int L, int n_slabs, const int* device_ids, void** handle)
{
    if (n_slabs < 1) return fail("n_slabs must be >= 1");
    int dev = device_ids ? device_ids[0] : 0;
    for (int s = 1; s < n_slabs && device_ids; ++s)
        if (device_ids[s] != dev)
            return fail("in-process slabs must share one device; multi-GPU runs use one process per GPU (cetkmc_create_rank)");
    if (n_slabs > 1 && L / n_slabs < 2) return fail("each slab needs at least 2 planes");
    std::vector<std::pair<int, int>> ranges;
    int base = L / n_slabs, rem = L % n_slabs, at = 0;
    for (int s = 0; s < n_slabs; ++s) { int n = base + (s < rem ? 1 : 0); ranges.push_back({at, n}); at += n; }
    return create_common(p, L, ranges, dev, n_slabs, 0, handle);
}

int cetkmc_get_unique_id(char out[128])
{
    CHK(load_rccl());
    ncclUniqueId id;
    static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId size");
    NCCLCHK(g_rccl.GetUniqueId(&id));
    memcpy(out, &id, 128);
    return 0;
}

int cetkmc_create_rank(const cetkmc_params* p, int L, int rank, int nranks, int device_id, const char unique_id[128],
                       void** handle)
{
    if (nranks < 1 || rank < 0 || rank >= nranks) return fail("bad rank/nranks");
    if (L % nranks != 0) return fail("L must be divisible by the number of ranks");
    if (nranks > 1 && L / nranks < 2) return fail("each slab needs at least 2 planes");
    const int n = L / nranks;
    std::vector<std::pair<int, int>> ranges{{rank * n, n}};
    CHK(create_common(p, L, ranges, device_id, nranks, rank, handle));
    Handle* h = (Handle*)*handle;
    h->rank = rank; h->nranks = nranks;
    if (nranks > 1 || unique_id) {
        if (!unique_id) { destroy_impl(h); *handle = nullptr; return fail("unique_id required"); }
        if (load_rccl()) { destroy_impl(h); *handle = nullptr; return 1; }
        ncclUniqueId id;
        memcpy(&id, unique_id, 128);
        ncclResult_t r = g_rccl.CommInitRank(&h->comm, nranks, id, rank);
        if (r != ncclSuccess) {
            std::string m = std::string("ncclCommInitRank: ") + g_rccl.GetErrorString(r);
            destroy_impl(h); *handle = nullptr;
            return fail(m);
        }
    }
    return 0;
}

int cetkmc_create_rank_host(const cetkmc_params* p, int L, int rank, int nranks, int device_id, const cetkmc_host_comm* hc,
                            void** handle)
{
    if (!hc || !hc->allgather || !hc->exchange) return fail("host transport callbacks required");
    if (nranks < 1 || rank < 0 || rank >= nranks) return fail("bad rank/nranks");
    if (L % nranks != 0) return fail("L must be divisible by the number of ranks");
    if (nranks > 1 && L / nranks < 2) return fail("each slab needs at least 2 planes");
    const int n = L / nranks;
    std::vector<std::pair<int, int>> ranges{{rank * n, n}};
    CHK(create_common(p, L, ranges, device_id, nranks, rank, handle));
    Handle* h = (Handle*)*handle;
    h->rank = rank; h->nranks = nranks;
    h->hc = *hc;
    return 0;
}

int cetkmc_destroy(void* handle) { destroy_impl((Handle*)handle); return 0; }

int cetkmc_set_params(void* handle, const cetkmc_params* p)
{
    Handle* h = (Handle*)handle;
    if (!h || !p) return fail("null argument");
    HIPCHK(hipSetDevice(h->dev));
    h->p = *p; h->kp = make_kparams(*p);
    h->swept = false; h->table_fresh = false; h->ifc_fresh = false;
    return upload_ktab(h);
}

int cetkmc_set_option(void* handle, const char* key, int64_t value)
{
    Handle* h = (Handle*)handle;
    if (!h || !key) return fail("null argument");
    if (!strcmp(key, "sweep_variant")) {
        if (value < 0 || value > 4) return fail("sweep_variant must be 0 (simple), 1 (streaming + rate table, default), 2 (streaming, recompute), 3 (census-free table sweep) or 4 (census-free, one block per plane, block sums in the same launch)");
        h->sweep_variant = (int)value;
        h->sweep_auto = 0;          // an explicit choice stands at every lattice size
        h->swept = false; h->table_fresh = false; h->ifc_fresh = false;
        return 0;
    }
    if (!strcmp(key, "interface_every_step")) { h->ifc_every_step = value ? 1 : 0; h->ifc_fresh = false; return 0; }
    if (!strcmp(key, "thermal_planes_per_block")) {
        if (value < 1 || value > 64) return fail("thermal_planes_per_block must be 1..64");
        h->therm_ni = (int)value;
        return 0;
    }
    if (!strcmp(key, "thermal_planes_per_block16")) {
        if (value < 1 || value > 64) return fail("thermal_planes_per_block16 must be 1..64");
        h->therm_ni16 = (int)value;
        return 0;
    }
    if (!strcmp(key, "thermal_lookahead")) { h->thermal_ahead = value ? 1 : 0; return 0; }
    if (!strcmp(key, "thermal_table")) { h->thermal_table = value ? 1 : 0; return 0; }
    if (!strcmp(key, "reserve_batch")) {
        // device buffers (uniform streams, per-step logs, one laser source plane per temperature update) and hipEvents of a
        // batch of `value` steps, allocated ahead of it: a bench keeps hipMalloc / hipEventCreate out of its timed region
        if (value < 0 || value > (1 << 22)) return fail("reserve_batch out of range");
        HIPCHK(hipSetDevice(h->dev));
        h->staged.valid = false;           // growing a buffer moves it: a batch staged before this call is gone
        const size_t n = (size_t)value;
        size_t c1 = h->cap_steps, c2 = h->cap_steps, c3 = h->cap_steps, c4 = h->cap_steps, c5 = h->cap_steps;
        CHK(grow(&h->d_u_pick, &c1, n));
        CHK(grow(&h->d_u_defect, &c2, n));
        CHK(grow(&h->d_log_total, &c3, n));
        CHK(grow(&h->d_log_event, &c4, n));
        CHK(grow(&h->d_log_nev, &c5, n));
        h->cap_steps = std::min({c1, c2, c3, c4, c5});
        CHK(grow(&h->d_u_np, &h->cap_np, 2 * n + 2 + (size_t)h->L * h->L));
        CHK(grow(&h->d_q, &h->cap_q, (n / 20 + 2) * (size_t)h->L * h->L));
        CHK(grow_pinned(&h->pin_in, &h->pin_in_cap, 2 * n * 8 + (2 * n + 2) * 8 + std::min<size_t>(PIN_SMALL_MAX, (n / 20 + 2) * (size_t)h->L * h->L * 8)));
        CHK(grow_pinned(&h->pin_out, &h->pin_out_cap, 64 + n * (16 + sizeof(cetkmc_event)) + h->slabs.size() * sizeof(int)));
        // events: the per-phase mode (7 per step) is used on short batches only; the sampled modes need 2 per (8th) step
        const int64_t need = std::max<int64_t>(7 * std::min<int64_t>(value, 256), value <= 64 ? 2 * value : 2 * (value / 8 + 1));
        while ((int64_t)h->prof.size() < need) { hipEvent_t e; HIPCHK(hipEventCreate(&e)); h->prof.push_back(e); }
        return 0;
    }
    if (!strcmp(key, "thermal_variant")) {
        // 0: one thread per voxel; 1 (default): plane marching -- k_thermal_tiles16 (16 x 256 tiles, 1024 threads, 2 rows per
        // thread) where the tiles cover the lattice exactly (L a multiple of 256), else k_thermal_march; 2: k_thermal_march
        // everywhere; 3: 16-row tiles with 4 rows per thread (512 threads); 4: the 8-row k_thermal_tiles; 5: 16 x 128 tiles
        if (value < 0 || value > 5) return fail("thermal_variant must be 0 (simple), 1 (marching, default: 16-row tiles, 2 rows per thread), 2 (marching, general kernel only), 3 (16-row tiles, 4 rows per thread), 4 (8-row tiles) or 5 (16-row x 128-column tiles)");
        h->thermal_general = value == 2 ? 1 : 0;
        h->thermal_tiles16 = value == 4 ? 0 : 1;
        h->thermal_rpt = value == 3 ? 4 : 2;
        h->thermal_kt = value == 5 ? 128 : 256;
        if (value >= 2) value = 1;
        h->thermal_variant = (int)value;
        return 0;
    }
    return fail(std::string("unknown option ") + key);
}

int cetkmc_sync(void* handle)
{
    Handle* h = (Handle*)handle;
    if (!h) return fail("null handle");
    HIPCHK(hipStreamSynchronize(h->stream));
    return 0;
}

int cetkmc_owned_planes(void* handle, int* i0, int* i1)
{
    Handle* h = (Handle*)handle;
    if (!h) return fail("null handle");
    if (i0) *i0 = h->own_i0;
    if (i1) *i1 = h->own_i1;
    return 0;
}

int cetkmc_upload(void* handle, const int64_t* state, const double* theta, const double* phi, const double* T,
                  const int64_t* defects)
{
    Handle* h = (Handle*)handle;
    if (!h) return fail("null handle");
    return upload_impl<int64_t>(h, 0, h->L, state, theta, phi, T, defects);
}
int cetkmc_download(void* handle, int64_t* state, double* theta, double* phi, double* T, int64_t* defects)
{
    Handle* h = (Handle*)handle;
    if (!h) return fail("null handle");
    return download_impl<int64_t>(h, 0, h->L, state, theta, phi, T, defects);
}
int cetkmc_upload_planes(void* handle, int i_begin, int i_end, const uint8_t* state, const double* theta,
                         const double* phi, const double* T, const uint8_t* defects)
{
    Handle* h = (Handle*)handle;
    if (!h) return fail("null handle");
    if (i_begin < 0 || i_end > h->L || i_begin >= i_end) return fail("bad plane range");
    return upload_impl<uint8_t>(h, i_begin, i_end, state, theta, phi, T, defects);
}
int cetkmc_download_planes(void* handle, int i_begin, int i_end, uint8_t* state, double* theta, double* phi, double* T,
                           uint8_t* defects)
{
    Handle* h = (Handle*)handle;
    if (!h) return fail("null handle");
    if (i_begin < 0 || i_end > h->L || i_begin >= i_end) return fail("bad plane range");
    return download_impl<uint8_t>(h, i_begin, i_end, state, theta, phi, T, defects);
}

int cetkmc_set_defects(void* handle, const uint8_t* mask)
{
    Handle* h = (Handle*)handle;
    if (!h || !mask) return fail("null argument");
    HIPCHK(hipSetDevice(h->dev));
    for (auto& s : h->slabs) {
        int a, b;
        ext_range(h, s, &a, &b);
        CHK(h2d_u8<uint8_t>(h, s, s.v.defects, mask, 0, a, b, false));
    }
    h->swept = false; h->ifc_fresh = false;
    return 0;
}

int cetkmc_set_prev_state(void* handle, const int64_t* prev_state)
{
    Handle* h = (Handle*)handle;
    if (!h) return fail("null handle");
    HIPCHK(hipSetDevice(h->dev));
    for (auto& s : h->slabs) {
        if (prev_state) {
            int a, b;
            ext_range(h, s, &a, &b);
            CHK(h2d_u8<int64_t>(h, s, s.prev, prev_state, 0, a, b, true));
            HIPCHK(hipMemsetAsync(s.v.row_chg, 1, (size_t)(s.v.nloc + 4) * h->L, h->stream));      // may differ anywhere
        } else {
            HIPCHK(hipMemcpyAsync(s.prev, s.v.state, s.nS, hipMemcpyDeviceToDevice, h->stream));
            HIPCHK(hipMemsetAsync(s.v.row_chg, 0, (size_t)(s.v.nloc + 4) * h->L, h->stream));
            HIPCHK(hipMemsetAsync(s.v.ifc_in, 0, s.nT, h->stream));
            hipLaunchKernelGGL(k_ifc_rebuild, dim3(2048), dim3(256), 0, h->stream, s.v);
            CHK(relist_slab(h, (int)(&s - h->slabs.data())));
            h->ifc_fresh = false;
        }
    }
    HIPCHK(hipStreamSynchronize(h->stream));
    return 0;
}

int cetkmc_thermal_cet(void* handle, double dt, int scrub_nan)
{
    Handle* h = (Handle*)handle;
    if (!h) return fail("null handle");
    HIPCHK(hipSetDevice(h->dev));
    CHK(launch_thermal(h, dt, 0, nullptr, 0, scrub_nan, false));
    HIPCHK(hipStreamSynchronize(h->stream));
    return 0;
}

int cetkmc_thermal_laser(void* handle, double dt, const double* q_top, int use_latent, int scrub_nan)
{
    Handle* h = (Handle*)handle;
    if (!h || !q_top) return fail("null argument");
    HIPCHK(hipSetDevice(h->dev));
    HIPCHK(hipMemcpyAsync(h->d_qtop, q_top, (size_t)h->L * h->L * sizeof(double), hipMemcpyHostToDevice, h->stream));
    CHK(launch_thermal(h, dt, 1, h->d_qtop, use_latent, scrub_nan, false));
    HIPCHK(hipStreamSynchronize(h->stream));
    return 0;
}

int cetkmc_rate_sweep(void* handle, cetkmc_sweep_info* info)
{
    Handle* h = (Handle*)handle;
    if (!h) return fail("null handle");
    HIPCHK(hipSetDevice(h->dev));
    CHK(launch_sweep(h, false));
    BatchCfg cfg{};
    CHK(launch_select(h, cfg, 0.0, 1));
    StepState ss;
    HIPCHK(hipMemcpyAsync(&ss, h->d_ss, sizeof ss, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    if (info) { info->total = ss.total; info->n_events = ss.n_events; info->n_dep = ss.n_dep; }
    return 0;
}

int cetkmc_select(void* handle, double r, cetkmc_event* out)
{
    Handle* h = (Handle*)handle;
    if (!h || !out) return fail("null argument");
    if (!h->swept) return fail("cetkmc_select needs a preceding cetkmc_rate_sweep on the current lattice");
    HIPCHK(hipSetDevice(h->dev));
    BatchCfg cfg{};
    CHK(launch_select(h, cfg, r, 0));
    std::vector<cetkmc_event> all(h->G);
    HIPCHK(hipMemcpyAsync(all.data(), h->d_events_all, all.size() * sizeof(cetkmc_event), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    out->type = -1;
    for (auto& e : all) if (e.type >= 0) *out = e;
    if (out->type < 0) return fail("no valid events");
    return 0;
}

int cetkmc_apply(void* handle, const cetkmc_event* ev, double theta_new, double phi_new, int make_defect)
{
    Handle* h = (Handle*)handle;
    if (!h || !ev) return fail("null argument");
    if (ev->type < 0 || ev->type > 3) return fail("bad event type");
    for (int a = 0; a < 3; ++a) if (ev->pos[a] < 0 || ev->pos[a] >= h->L) return fail("event position out of range");
    if (ev->type == CETKMC_DIFF || ev->type == CETKMC_ATT)
        for (int a = 0; a < 3; ++a) if (ev->target[a] < 0 || ev->target[a] >= h->L) return fail("event target out of range");
    HIPCHK(hipSetDevice(h->dev));
    cetkmc_event e = *ev;
    if (e.type == CETKMC_DEP || e.type == CETKMC_NUC) { e.theta = theta_new; e.phi = phi_new; }
    hipLaunchKernelGGL(k_apply_direct, dim3(1), dim3(64), 0, h->stream, h->kp, (const SlabView*)h->d_views[h->cur],
                       (int)h->slabs.size(), e, make_defect, h->d_ss, (const double*)h->d_ktab);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(h->stream));
    h->swept = false; h->ifc_fresh = false;      // the touched voxels' codes are current, their sums are not
    return 0;
}

int cetkmc_row_sums(void* handle, double* rowsum, int32_t* rowcnt)
{
    Handle* h = (Handle*)handle;
    if (!h) return fail("null handle");
    if (!h->swept) return fail("cetkmc_row_sums needs a preceding cetkmc_rate_sweep");
    HIPCHK(hipSetDevice(h->dev));
    const size_t L = h->L;
    for (auto& s : h->slabs) {
        const size_t off = (size_t)s.v.gi0 * 3 * L, n = (size_t)s.v.nloc * 3 * L;
        if (rowsum) HIPCHK(hipMemcpyAsync(rowsum + off, s.v.rowsum, n * sizeof(double), hipMemcpyDeviceToHost, h->stream));
        if (rowcnt) HIPCHK(hipMemcpyAsync(rowcnt + off, s.v.rowcnt, n * sizeof(int32_t), hipMemcpyDeviceToHost, h->stream));
    }
    HIPCHK(hipStreamSynchronize(h->stream));
    return 0;
}

int cetkmc_enumerate_events(void* handle, cetkmc_event* buf, int64_t cap, int64_t* n)
{
    Handle* h = (Handle*)handle;
    if (!h || !n) return fail("null argument");
    HIPCHK(hipSetDevice(h->dev));
    CHK(launch_sweep(h, false));
    const size_t L = h->L;
    int64_t total = 0;
    std::vector<std::vector<int64_t>> offs(h->slabs.size());
    for (size_t s = 0; s < h->slabs.size(); ++s) {
        Slab& sl = h->slabs[s];
        const size_t rows = (size_t)sl.v.nloc * 3 * L;
        std::vector<int32_t> cnt(rows);
        HIPCHK(hipMemcpyAsync(cnt.data(), sl.v.rowcnt, rows * sizeof(int32_t), hipMemcpyDeviceToHost, h->stream));
        HIPCHK(hipStreamSynchronize(h->stream));
        offs[s].resize(rows);
        for (size_t r = 0; r < rows; ++r) { offs[s][r] = total; total += cnt[r]; }
    }
    *n = total;
    if (!buf || cap <= 0 || total == 0) return 0;
    const int64_t m = std::min<int64_t>(cap, total);
    DevTmp<cetkmc_event> d_out;
    HIPCHK(d_out.alloc((size_t)m));
    for (size_t s = 0; s < h->slabs.size(); ++s) {
        const size_t rows = offs[s].size();
        DevTmp<int64_t> d_off;
        HIPCHK(d_off.alloc(rows));
        HIPCHK(hipMemcpyAsync(d_off, offs[s].data(), rows * sizeof(int64_t), hipMemcpyHostToDevice, h->stream));
        hipLaunchKernelGGL(k_enumerate, dim3((unsigned)((rows + 63) / 64)), dim3(64), 0, h->stream, h->kp, view_of(h, (int)s),
                           (const double*)h->d_ktab, (const int64_t*)d_off.p, d_out.p, m);
        HIPCHK(hipGetLastError());
        HIPCHK(hipStreamSynchronize(h->stream));
    }
    HIPCHK(hipMemcpy(buf, d_out, (size_t)m * sizeof(cetkmc_event), hipMemcpyDeviceToHost));
    int64_t rank = 0;
    for (int64_t e = 0; e < m; ++e) if (buf[e].type == CETKMC_DEP) buf[e].dep_rank = rank++;
    return 0;
}

// host inputs of a batch -> the handle's device buffers (grown as needed); shared by cetkmc_stage_inputs / cetkmc_run_steps
static int check_run_args(Handle* h, const cetkmc_run_args* a, bool need_ptrs, int64_t* n_therm_out)
{
    const int64_t n = a->n_steps;
    if (n < 0) return fail("n_steps < 0");
    if (a->rng_mode < 0 || a->rng_mode > 2) return fail("rng_mode must be 0 (reference stream), 1 (counter species draw) or 2 (all counter based)");
    const bool streams = a->rng_mode != 2;          // rng_mode 2 draws every uniform from (seed, step, key): no host streams
    if (need_ptrs && streams && n > 0 && !a->u_pick) return fail("u_pick required");
    if (need_ptrs && streams && a->defect_fraction > 0.0 && n > 0 && !a->u_defect) return fail("u_defect required when defect_fraction > 0");
    if (need_ptrs && streams && a->np_cap > 0 && !a->u_np) return fail("u_np required");
    int64_t n_therm = 0;
    if (a->thermal_mode) for (int64_t s = 0; s < n; ++s) if ((a->step0 + s) % 20 == 0) ++n_therm;
    if (a->thermal_mode == 2 && (n_therm > a->n_q || (need_ptrs && n_therm > 0 && !a->q_planes)))
        return fail("thermal_mode 2 needs one q plane per thermal update in the batch");
    *n_therm_out = n_therm;
    (void)h;
    return 0;
}

static int upload_run_inputs(Handle* h, const cetkmc_run_args* a, int64_t n_therm)
{
    const int64_t n = a->n_steps;
    const size_t L2 = (size_t)h->L * h->L;
    size_t c1 = h->cap_steps, c2 = h->cap_steps, c3 = h->cap_steps, c4 = h->cap_steps, c5 = h->cap_steps;
    CHK(grow(&h->d_u_pick, &c1, (size_t)n));
    CHK(grow(&h->d_u_defect, &c2, (size_t)n));
    CHK(grow(&h->d_log_total, &c3, (size_t)n));
    CHK(grow(&h->d_log_event, &c4, (size_t)n));
    CHK(grow(&h->d_log_nev, &c5, (size_t)n));
    h->cap_steps = std::min({c1, c2, c3, c4, c5});
    CHK(grow(&h->d_u_np, &h->cap_np, (size_t)std::max<int64_t>(a->np_cap, 2)));
    if (a->thermal_mode == 2) CHK(grow(&h->d_q, &h->cap_q, (size_t)std::max<int64_t>(n_therm, 1) * L2));
    // the small arrays travel through the page-locked staging buffer (queued copies); a large one goes directly
    struct Part { void* dst; const void* src; size_t bytes; };
    const bool streams = a->rng_mode != 2;
    const Part parts[4] = {{h->d_u_pick, a->u_pick, (streams && n > 0 && a->u_pick) ? (size_t)n * 8 : 0},
                           {h->d_u_defect, a->u_defect, (streams && n > 0 && a->u_defect) ? (size_t)n * 8 : 0},
                           {h->d_u_np, a->u_np, (streams && a->np_cap > 0 && a->u_np) ? (size_t)a->np_cap * 8 : 0},
                           {h->d_q, a->q_planes, (a->thermal_mode == 2 && n_therm > 0) ? (size_t)n_therm * L2 * 8 : 0}};
    size_t small = 0;
    for (const Part& q : parts) if (q.bytes && q.bytes <= PIN_SMALL_MAX) small += q.bytes;
    // the staging buffer may still feed the copies of the previous call: they are complete (every stepping call ends with a
    // stream synchronisation), so it can be rewritten
    CHK(grow_pinned(&h->pin_in, &h->pin_in_cap, small));
    size_t off = 0;
    for (const Part& q : parts) {
        if (!q.bytes) continue;
        if (q.bytes <= PIN_SMALL_MAX) {
            memcpy(h->pin_in + off, q.src, q.bytes);
            HIPCHK(hipMemcpyAsync(q.dst, h->pin_in + off, q.bytes, hipMemcpyHostToDevice, h->stream));
            off += q.bytes;
        } else {
            HIPCHK(hipMemcpyAsync(q.dst, q.src, q.bytes, hipMemcpyHostToDevice, h->stream));
        }
    }
    h->cnt.bytes_h2d += (n > 0 ? n * 8 : 0) + (n > 0 && a->u_defect ? n * 8 : 0) + std::max<int64_t>(a->np_cap, 0) * 8 +
                        (a->thermal_mode == 2 ? n_therm * (int64_t)L2 * 8 : 0);
    return 0;
}

int cetkmc_stage_inputs(void* handle, const cetkmc_run_args* a)
{
    Handle* h = (Handle*)handle;
    if (!h || !a) return fail("null argument");
    int64_t n_therm = 0;
    h->staged.valid = false;
    CHK(check_run_args(h, a, true, &n_therm));
    HIPCHK(hipSetDevice(h->dev));
    CHK(upload_run_inputs(h, a, n_therm));
    HIPCHK(hipStreamSynchronize(h->stream));
    h->staged.valid = true;
    h->staged.step0 = a->step0; h->staged.n_steps = a->n_steps; h->staged.np_cap = a->np_cap; h->staged.n_q = a->n_q;
    h->staged.thermal_mode = a->thermal_mode; h->staged.has_defect = a->u_defect != nullptr;
    return 0;
}

int cetkmc_run_steps(void* handle, const cetkmc_run_args* a, cetkmc_run_result* res, double* totals,
                     cetkmc_event* events, int64_t* n_events)
{
    Handle* h = (Handle*)handle;
    if (!h || !a || !res) return fail("null argument");
    const int64_t n = a->n_steps;
    // all input pointers NULL: the batch's inputs were put on the device by cetkmc_stage_inputs (same batch shape)
    const bool staged = n > 0 && a->rng_mode != 2 && !a->u_pick && !a->u_defect && !a->u_np && !a->q_planes;
    int64_t n_therm = 0;
    CHK(check_run_args(h, a, !staged, &n_therm));
    const auto g = h->staged;
    h->staged.valid = false;           // a staged batch is consumed (or dropped) by the next stepping call
    if (staged) {
        if (!g.valid) return fail("no input pointers and no staged batch (cetkmc_stage_inputs)");
        if (g.step0 != a->step0 || g.n_steps != n || g.np_cap != a->np_cap || g.n_q != a->n_q || g.thermal_mode != a->thermal_mode ||
            (a->defect_fraction > 0.0 && !g.has_defect))
            return fail("the staged batch (cetkmc_stage_inputs) does not match this call's step0 / n_steps / np_cap / n_q / thermal_mode");
    }
    HIPCHK(hipSetDevice(h->dev));
    const size_t L2 = (size_t)h->L * h->L;
    if (!staged) CHK(upload_run_inputs(h, a, n_therm));
    // reset the batch part of the step state (nucleation_count persists): a one-thread kernel, no host round trip
    StepState ss;
    hipLaunchKernelGGL(k_batch_reset, dim3(1), dim3(1), 0, h->stream, h->d_ss);

    BatchCfg cfg{};
    cfg.step0 = a->step0; cfg.np_cap = a->np_cap; cfg.defect_fraction = a->defect_fraction; cfg.seed = a->seed;
    cfg.rng_mode = a->rng_mode; cfg.batch = 1;
    if (a->rng_mode == 2) cfg.np_cap = INT64_MAX / 2;       // no stream to run out of
    // profile 1: two events per step around the rate-sweep kernel (bench roofline); profile 2: seven per step
    // (thermal | interface | sweep | reduce(+all-gather) | select+apply boundaries) for cetkmc_get_counters
    const int EPS = a->profile == 2 ? 7 : 2;
    // profile 3: like 1 but only every 8th step (every 4th in batches of <= 64 steps) carries the two events (a pair costs
    // ~4 us of stream time)
    const int64_t pstride = a->profile == 3 ? (n <= 64 ? 4 : 8) : 1;
    auto sampled = [&](int64_t s) { return a->profile == 1 || (a->profile == 3 && s % pstride == 0); };
    if (a->profile) {
        const int64_t need = a->profile == 2 ? EPS * n : 2 * ((n + pstride - 1) / pstride);     // sampled steps only
        while ((int64_t)h->prof.size() < need) { hipEvent_t e; HIPCHK(hipEventCreate(&e)); h->prof.push_back(e); }
    }
    auto pev = [&](int64_t s, int q) -> hipEvent_t { return a->profile == 2 ? h->prof[EPS * s + q] : nullptr; };
    std::vector<char> was_thermal, was_full;
    if (a->profile == 2) { was_thermal.assign((size_t)n, 0); was_full.assign((size_t)n, 0); }
    h->time_comm = a->profile == 2 && multi_rank(h);
    h->comm_ev_used = 0;
    HIPCHK(hipEventRecord(h->ev0, h->stream));
    int64_t q_idx = 0;
    const bool incr = a->incremental && h->sweep_variant >= 1;
    // the apply kernel re-evaluates the <= 30 listed voxels an event touches, which keeps every listed voxel's sum
    // current between temperature updates (the interface kernel then runs only after one); off: the interface kernel
    // re-evaluates the whole list before every full sweep
    const int eval_touched = (h->sweep_variant >= 1 && !h->ifc_every_step) ? 1 : 0;
    res->full_sweeps = 0;
    // continuation of a batch that ran out of stream on a temperature-update step: that update is already in the field
    const int64_t therm_skip_g = h->therm_applied_g;
    h->therm_applied_g = -1;
    for (int64_t s = 0; s < n; ++s) {
        const int64_t g = a->step0 + s;
        const bool therm_due = a->thermal_mode && g % 20 == 0;
        const bool therm = therm_due && !(s == 0 && g == therm_skip_g);
        if (therm_due && !therm && a->thermal_mode == 2) ++q_idx;      // its source plane was consumed by the stopped batch
        if (incr && s > 0 && !therm) {
            // exact incremental step: rates can only have changed in the rows recorded by the last apply
            ++h->cnt.incremental_steps;
            if (a->profile == 2) {
                HIPCHK(hipEventRecord(pev(s, 2), h->stream));
                CHK(launch_dirty_rows(h, nullptr, pev(s, 3)));
                HIPCHK(hipEventRecord(pev(s, 4), h->stream));
                CHK(launch_select_apply(h, cfg, 1, h->d_dirty, s));
                HIPCHK(hipEventRecord(pev(s, 5), h->stream));
            } else {
                CHK(launch_dirty_rows(h, sampled(s) ? h->prof[2 * (s / pstride)] : nullptr, sampled(s) ? h->prof[2 * (s / pstride) + 1] : nullptr));
                CHK(launch_select_apply(h, cfg, 1, h->d_dirty, s));
            }
            h->swept = false;
            continue;
        }
        ++res->full_sweeps;
        if (a->profile == 2) { was_full[s] = 1; HIPCHK(hipEventRecord(pev(s, 0), h->stream)); }
        if (therm) {
            const int laser = a->thermal_mode == 2 ? 1 : 0, latent = laser ? a->use_latent : 0;
            CHK(thermal_step(h, g, a->thermal_dt, laser, laser ? h->d_q + (size_t)q_idx * L2 : nullptr, latent, 1));
            q_idx += laser;
            if (g + 20 < a->step0 + n)      // the next update lies inside this batch (its source plane is on the device): look ahead
                CHK(launch_thermal_ahead(h, g + 20, a->thermal_dt, laser, laser ? h->d_q + (size_t)q_idx * L2 : nullptr, latent, 1));
            if (a->profile == 2) was_thermal[s] = 1;
        }
        if (a->profile == 2) CHK(launch_sweep(h, true, pev(s, 2), pev(s, 3), false, pev(s, 1), pev(s, 4)));
        else if (sampled(s)) CHK(launch_sweep(h, true, h->prof[2 * (s / pstride)], h->prof[2 * (s / pstride) + 1]));
        else CHK(launch_sweep(h, true));
        CHK(launch_select_apply(h, cfg, (incr || eval_touched) ? 1 : 0, incr ? h->d_dirty : nullptr, s));
        if (!(incr || eval_touched)) h->ifc_fresh = false;
        if (a->profile == 2) HIPCHK(hipEventRecord(pev(s, 5), h->stream));
        h->swept = false;
    }
    HIPCHK(hipGetLastError());
    HIPCHK(hipEventRecord(h->ev1, h->stream));
    // everything the host wants back travels behind ONE synchronisation: step state, the per-step logs (all n entries;
    // only the first steps_done are meaningful) and the interface lists' lengths
    const size_t nsl = h->slabs.size();
    const size_t o_tot = 64, o_ev = o_tot + (size_t)n * 8, o_nev = o_ev + (size_t)n * sizeof(cetkmc_event), o_len = o_nev + (size_t)n * 8;
    static_assert(sizeof(StepState) <= 64, "StepState grew: move the log offsets");
    CHK(grow_pinned(&h->pin_out, &h->pin_out_cap, o_len + nsl * sizeof(int)));
    HIPCHK(hipMemcpyAsync(h->pin_out, h->d_ss, sizeof ss, hipMemcpyDeviceToHost, h->stream));
    if (totals && n > 0) HIPCHK(hipMemcpyAsync(h->pin_out + o_tot, h->d_log_total, (size_t)n * 8, hipMemcpyDeviceToHost, h->stream));
    if (events && n > 0) HIPCHK(hipMemcpyAsync(h->pin_out + o_ev, h->d_log_event, (size_t)n * sizeof(cetkmc_event), hipMemcpyDeviceToHost, h->stream));
    if (n_events && n > 0) HIPCHK(hipMemcpyAsync(h->pin_out + o_nev, h->d_log_nev, (size_t)n * 8, hipMemcpyDeviceToHost, h->stream));
    std::vector<int> list_len(nsl, 0);
    for (size_t sl = 0; sl < nsl; ++sl)
        HIPCHK(hipMemcpyAsync(h->pin_out + o_len + sl * sizeof(int), h->slabs[sl].v.ifc_n, sizeof(int), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    memcpy(&ss, h->pin_out, sizeof ss);
    if (totals && n > 0) memcpy(totals, h->pin_out + o_tot, (size_t)n * 8);
    if (events && n > 0) memcpy(events, h->pin_out + o_ev, (size_t)n * sizeof(cetkmc_event));
    if (n_events && n > 0) memcpy(n_events, h->pin_out + o_nev, (size_t)n * 8);
    if (nsl) memcpy(list_len.data(), h->pin_out + o_len, nsl * sizeof(int));
    if (h->spec.valid) { HIPCHK(hipStreamSynchronize(h->stream2)); h->spec.valid = false; }      // never outlives its batch
    if (h->time_comm) {
        for (size_t q = 0; q + 1 < h->comm_ev_used; q += 2) {
            float t = 0.f;
            HIPCHK(hipEventElapsedTime(&t, h->comm_ev[q], h->comm_ev[q + 1]));
            h->cnt.ms_comm += t;
            ++h->cnt.comm_calls;
        }
        h->time_comm = false; h->comm_ev_used = 0;
    }
    res->steps_done = ss.cur; res->status = ss.status; res->np_used = ss.np_pos; res->q_used = q_idx;
    res->nucleation_count = ss.nuc_count;
    res->min_margin = ss.min_margin;
    float ms = 0.f;
    HIPCHK(hipEventElapsedTime(&ms, h->ev0, h->ev1));
    res->wall_ms = ms;
    res->sweep_ms_total = 0.0; res->sweep_launches = 0;
    if (a->profile == 2) {
        auto el = [&](int64_t s, int qa, int qb, double* acc) -> int {
            float t = 0.f;
            HIPCHK(hipEventElapsedTime(&t, h->prof[EPS * s + qa], h->prof[EPS * s + qb]));
            *acc += t;
            return 0;
        };
        for (int64_t s = 0; s < n; ++s) {
            if (was_full[s]) {
                if (was_thermal[s]) CHK(el(s, 0, 1, &h->cnt.ms_thermal));
                CHK(el(s, 1, 2, &h->cnt.ms_interface));
                CHK(el(s, 2, 3, &h->cnt.ms_sweep));
                CHK(el(s, 3, 4, &h->cnt.ms_reduce));
                CHK(el(s, 4, 5, &h->cnt.ms_select_apply));
                double t = 0.0;
                CHK(el(s, 2, 3, &t));
                res->sweep_ms_total += t;
                ++res->sweep_launches;
            } else {
                CHK(el(s, 2, 3, &h->cnt.ms_dirty_rows));
                CHK(el(s, 3, 4, &h->cnt.ms_reduce));
                CHK(el(s, 4, 5, &h->cnt.ms_select_apply));
            }
        }
        h->cnt.profiled_steps += n;
    } else if (a->profile) {
        for (int64_t s = 0; s < n; ++s) {
            if (!sampled(s)) continue;
            float t = 0.f;
            HIPCHK(hipEventElapsedTime(&t, h->prof[2 * (s / pstride)], h->prof[2 * (s / pstride) + 1]));
            res->sweep_ms_total += t;
            ++res->sweep_launches;
        }
    }
    h->cnt.steps += ss.cur;
    const int64_t done = ss.cur;
    if (ss.status != 0) {
        // the batch stopped early: the steps still queued behind the stop ran as pass-throughs (temperature copied through,
        // buffer pair flipped, rate table / interface kernels returning at once), so the host's freshness flags describe
        // work the device skipped.  Recompute both from the field as it stands.
        h->table_fresh = false; h->ifc_fresh = false; h->swept = false;
        if (ss.status == 2 && a->thermal_mode && done < n && (a->step0 + done) % 20 == 0 &&
            !(done == 0 && a->step0 == therm_skip_g))
            h->therm_applied_g = a->step0 + done;          // that step's update ran before the shortage was noticed
        else if (ss.status == 2 && done == 0 && a->step0 == therm_skip_g)
            h->therm_applied_g = therm_skip_g;             // still the same pending step (nothing executed, nothing updated)
    }
    const int64_t nt = done + (ss.status == 1 ? 1 : 0);
    if (totals && ss.status == 1 && nt <= n) totals[done] = ss.total;
    h->cnt.bytes_d2h += (totals ? n * 8 : 0) + (events ? n * (int64_t)sizeof(cetkmc_event) : 0) + (n_events ? n * 8 : 0);
    // the interface lists grow while stepping: keep the interface kernel's grid at one entry per thread
    for (int n_list : list_len) h->ifc_blocks = std::max(h->ifc_blocks, std::min(8192, (n_list + 255) / 256 + 64));
    return 0;
}

int cetkmc_get_counters(void* handle, cetkmc_counters* out, int reset)
{
    Handle* h = (Handle*)handle;
    if (!h || !out) return fail("null argument");
    *out = h->cnt;
    if (reset) h->cnt = cetkmc_counters{};
    return 0;
}

// ---- Mode B: synchronous super-steps over spatial boxes (superstep.hpp) ------------------------------
// kmc_simulation.py:331-332 per executed event of super-step g (DESIGN.md "Mode B": time advance; oracle orc_run_supersteps)
static double superstep_dt_event(uint64_t seed, int64_t g, double total)
{
    const double u = counter_uniform(seed, (uint64_t)g, KEY_DT);
    const double um = (u > 1e-12) ? u : 1e-12;              // max(1e-12, u)
    const double dt = -std::log(um) / total;
    return (1e-12 > dt) ? 1e-12 : dt;                       // max(dt, 1e-12)
}

int cetkmc_run_supersteps(void* handle, const cetkmc_super_args* a, cetkmc_run_result* res, double* totals,
                          cetkmc_event* events, int64_t* n_executed, double* dt_event)
{
    Handle* h = (Handle*)handle;
    if (!h || !a || !res) return fail("null argument");
    const int64_t n = a->n_steps;
    if (n < 0) return fail("n_steps < 0");
    if (h->sweep_variant < 1) return fail("cetkmc_run_supersteps needs a streaming sweep_variant (1 or 2)");
    if (a->box == h->L) {
        // single domain (window = whole lattice, no octants): one event per super-step from the global canonical tree --
        // this IS a Mode A step whose uniforms are the counter-based ones of box 0, so it runs through the Mode A kernels
        if (multi_rank(h) && h->nranks > 1) return fail("the single-domain case (box == L) runs in one process");
        cetkmc_run_args ra{};
        ra.step0 = a->step0; ra.n_steps = n; ra.defect_fraction = a->defect_fraction;
        ra.rng_mode = 2; ra.seed = a->seed; ra.thermal_mode = a->thermal_mode; ra.thermal_dt = a->thermal_dt;
        ra.q_planes = a->q_planes; ra.n_q = a->n_q; ra.use_latent = a->use_latent;
        std::vector<double> tot((size_t)n + 1, 0.0);
        CHK(cetkmc_run_steps(handle, &ra, res, tot.data(), events, nullptr));
        const int64_t done = res->steps_done;
        for (int64_t s = 0; s < done; ++s) {
            if (totals) totals[s] = tot[(size_t)s];
            if (n_executed) n_executed[s] = 1;
            if (dt_event) dt_event[s] = superstep_dt_event(a->seed, a->step0 + s, tot[(size_t)s]);
        }
        if (totals && res->status == 1 && done < n) totals[done] = tot[(size_t)done];
        h->cnt.supersteps += done;
        return 0;
    }
    if (a->box < 8 || a->box > 16 || (a->box & 1) || h->L % a->box) return fail("box must be even, 8..16, and divide L (or equal L: single domain)");
    const bool ranks = multi_rank(h) && h->nranks > 1;
    if (ranks && h->nranks > 64) return fail("Mode B supports up to 64 ranks");
    if (ranks && (h->L / h->nranks) % a->box) return fail("across ranks the boxes must be aligned to the slabs: (L / nranks) % box == 0");
    int64_t n_therm = 0;
    if (a->thermal_mode) for (int64_t s = 0; s < n; ++s) if ((a->step0 + s) % 20 == 0) ++n_therm;
    if (a->thermal_mode == 2 && (n_therm > a->n_q || (n_therm > 0 && !a->q_planes)))
        return fail("thermal_mode 2 needs one q plane per thermal update in the batch");
    HIPCHK(hipSetDevice(h->dev));
    const size_t L2 = (size_t)h->L * h->L;
    SuperCfg C{};
    C.step0 = a->step0; C.defect_fraction = a->defect_fraction; C.seed = a->seed;
    C.box = a->box; C.H = a->box / 2; C.nb = h->L / a->box;
    C.PH = 1; while (C.PH < C.H) C.PH <<= 1;
    C.PT = 1; while (C.PT < 3 * C.H) C.PT <<= 1;
    // this handle's boxes: the box layers of its owned planes (all of them in a single process); global box index
    // d = (di * nb + dj) * nb + dk, so they are the contiguous range [d0, d0 + D)
    const int nb2 = C.nb * C.nb;
    C.d0 = ranks ? (h->own_i0 / a->box) * nb2 : 0;
    const int D = ranks ? ((h->own_i1 - h->own_i0) / a->box) * nb2 : C.nb * nb2;
    C.D_loc = D;
    C.null_events = a->null_events ? 1 : 0;
    C.nranks = ranks ? h->nranks : 1;
    const int NE = D + 2 * nb2;                               // own events | lower neighbour's top layer | upper neighbour's bottom layer
    const size_t shmem = (size_t)C.PT * C.PH * C.PH * 19;     // heap: 2*NL doubles + 2*NL flags, NL leaf codes
    h->staged.valid = false;            // the batch buffers are reused below
    {
        size_t c1 = h->cap_steps, c2 = h->cap_steps, c3 = h->cap_steps, c4 = h->cap_steps, c5 = h->cap_steps;
        CHK(grow(&h->d_u_pick, &c1, (size_t)n));
        CHK(grow(&h->d_u_defect, &c2, (size_t)n));
        CHK(grow(&h->d_log_total, &c3, (size_t)n));
        CHK(grow(&h->d_log_event, &c4, (size_t)n));
        CHK(grow(&h->d_log_nev, &c5, (size_t)n));
        h->cap_steps = std::min({c1, c2, c3, c4, c5});
        if (a->thermal_mode == 2) CHK(grow(&h->d_q, &h->cap_q, (size_t)std::max<int64_t>(n_therm, 1) * L2));
    }
    // working buffers of the handle (grow-only; the previous call ended with a synchronisation, nothing still reads them)
    CHK(grow(&h->d_sup_rmax, &h->cap_sup_rmax, (size_t)std::max(C.nranks, 1)));
    CHK(grow(&h->d_sup_dom, &h->cap_sup_dom, (size_t)NE));
    CHK(grow(&h->d_sup_picks, &h->cap_sup_picks, (size_t)D));
    CHK(grow(&h->d_sup_cnt, &h->cap_sup_cnt, (size_t)SUPER_CNT_SLOTS * SUPER_CNT_STRIDE));
    if (events && n > 0) CHK(grow(&h->d_sup_log, &h->cap_sup_log, (size_t)n * D));
    double* const d_rmax = h->d_sup_rmax;
    cetkmc_event* const d_dom = h->d_sup_dom;
    DomPick* const d_picks = h->d_sup_picks;
    unsigned long long* const d_cnt = h->d_sup_cnt;
    cetkmc_event* const d_log = (events && n > 0) ? h->d_sup_log : nullptr;
    HIPCHK(hipMemsetAsync(d_rmax, 0, (size_t)std::max(C.nranks, 1) * sizeof(double), h->stream));
    HIPCHK(hipMemsetAsync(d_dom, 0xFF, (size_t)NE * sizeof(cetkmc_event), h->stream));      // type -1: nothing received
    const size_t cnt_bytes = (size_t)SUPER_CNT_SLOTS * SUPER_CNT_STRIDE * sizeof(unsigned long long);
    HIPCHK(hipMemsetAsync(d_cnt, 0, cnt_bytes, h->stream));
    if (a->thermal_mode == 2 && n_therm > 0)
        HIPCHK(hipMemcpyAsync(h->d_q, a->q_planes, (size_t)n_therm * L2 * 8, hipMemcpyHostToDevice, h->stream));
    StepState ss;
    hipLaunchKernelGGL(k_batch_reset, dim3(1), dim3(1), 0, h->stream, h->d_ss);     // the batch part of the step state, on the stream
    BatchCfg cfg{};
    cfg.step0 = a->step0; cfg.np_cap = 0; cfg.defect_fraction = a->defect_fraction; cfg.seed = a->seed;
    cfg.rng_mode = 1; cfg.batch = 1;
    HIPCHK(hipEventRecord(h->ev0, h->stream));
    int64_t q_idx = 0;
    for (int64_t s = 0; s < n; ++s) {
        const int64_t g = a->step0 + s;
        const bool therm = a->thermal_mode && g % 20 == 0;
        if (therm) {
            const int laser = a->thermal_mode == 2 ? 1 : 0, latent = laser ? a->use_latent : 0;
            CHK(thermal_step(h, g, a->thermal_dt, laser, laser ? h->d_q + (size_t)q_idx * L2 : nullptr, latent, 1));
            q_idx += laser;
            if (g + 20 < a->step0 + n)      // the next update lies inside this batch (its source plane is on the device): look ahead
                CHK(launch_thermal_ahead(h, g + 20, a->thermal_dt, laser, laser ? h->d_q + (size_t)q_idx * L2 : nullptr, latent, 1));
        }
        // interface sums: the list kernel after a temperature update / when something else made them stale (list
        // rebuilt in address order first: k_domain_touch flags new interface voxels without appending them); otherwise
        // they are current -- k_domain_touch re-evaluated what the events changed
        if (s > 0 && (!h->table_fresh || !h->ifc_fresh)) CHK(relist(h));
        CHK(launch_sweep(h, true, nullptr, nullptr, true));
        CHK(launch_select(h, cfg, 0.0, 1));                              // total, counts, termination test
        if (C.H == 4)       // box 8: window tree in registers
            hipLaunchKernelGGL(k_domain_pick8, dim3(D), dim3(64), 0, h->stream, h->kp, (const SlabView*)h->d_views[h->cur],
                               (int)h->slabs.size(), h->L, C, (const StepState*)h->d_ss, (const double*)h->d_ktab, d_picks, (long long)s);
        else
            hipLaunchKernelGGL(k_domain_pick, dim3(D), dim3(64), shmem, h->stream, h->kp, (const SlabView*)h->d_views[h->cur],
                               (int)h->slabs.size(), h->L, C, (const StepState*)h->d_ss, (const double*)h->d_ktab, d_picks);
        if (C.null_events) {
            // R_max of the super-step: this rank's largest window total, then (across ranks) the largest of all ranks
            hipLaunchKernelGGL(k_domain_rmax, dim3((D + 1023) / 1024), dim3(1024), 0, h->stream, (const DomPick*)d_picks, D, (const StepState*)h->d_ss,
                               d_rmax, ranks ? h->rank : 0);
            if (ranks) CHK(comm_allgather(h, d_rmax, sizeof(double)));
        }
        hipLaunchKernelGGL(k_domain_apply, dim3((D + 63) / 64), dim3(64), 0, h->stream, h->kp, (const SlabView*)h->d_views[h->cur],
                           (int)h->slabs.size(), h->L, D, C, h->d_ss, (const DomPick*)d_picks, d_dom, d_cnt, d_log, (const double*)d_rmax);
        int n_touch = D;
        if (ranks) {
            // the events of my bottom / top box layer go to the ranks below / above, theirs come here: every rank applies
            // them to its own copy (owned planes: diffusion targets across the boundary; halo planes: the neighbour's state)
            CHK(comm_exchange(h, d_dom, d_dom + D, d_dom + (D - nb2), d_dom + D + nb2, (size_t)nb2 * sizeof(cetkmc_event)));
            hipLaunchKernelGGL(k_domain_apply_remote, dim3((2 * nb2 + 63) / 64), dim3(64), 0, h->stream, (const SlabView*)h->d_views[h->cur],
                               (int)h->slabs.size(), C, (const StepState*)h->d_ss, (const cetkmc_event*)(d_dom + D));
            n_touch = NE;
        }
        hipLaunchKernelGGL(k_domain_touch, dim3(2 * ((n_touch + 15) / 16)), dim3(256), 0, h->stream, h->kp, (const SlabView*)h->d_views[h->cur],
                           (int)h->slabs.size(), n_touch, (const cetkmc_event*)d_dom, (const StepState*)h->d_ss, (const double*)h->d_ktab);
        hipLaunchKernelGGL(k_super_commit, dim3(1), dim3(64), 0, h->stream, h->d_ss, d_cnt, h->d_log_total, h->d_log_nev,
                           C.null_events ? d_rmax : (double*)nullptr, std::min(C.nranks, 64));
        h->swept = false;
    }
    if (n > 0) CHK(relist(h));        // leave a complete, address-ordered interface list behind (Mode A reads it)
    HIPCHK(hipGetLastError());
    HIPCHK(hipEventRecord(h->ev1, h->stream));
    HIPCHK(hipMemcpyAsync(&ss, h->d_ss, sizeof ss, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    if (h->spec.valid) { HIPCHK(hipStreamSynchronize(h->stream2)); h->spec.valid = false; }
    res->steps_done = ss.cur; res->status = ss.status; res->np_used = 0; res->q_used = q_idx;
    res->nucleation_count = ss.nuc_count;
    float ms = 0.f;
    HIPCHK(hipEventElapsedTime(&ms, h->ev0, h->ev1));
    res->wall_ms = ms; res->sweep_ms_total = 0.0; res->sweep_launches = 0; res->full_sweeps = n; res->min_margin = 1.0;
    h->cnt.supersteps += ss.cur;
    if (ss.status != 0) { h->table_fresh = false; h->ifc_fresh = false; h->swept = false; }     // as in cetkmc_run_steps
    CHK(refresh_ifc_grid(h));
    const int64_t done = ss.cur;
    if ((totals || dt_event) && done > 0) {
        std::vector<double> tot((size_t)done);
        HIPCHK(hipMemcpy(tot.data(), h->d_log_total, (size_t)done * 8, hipMemcpyDeviceToHost));
        if (totals) memcpy(totals, tot.data(), (size_t)done * 8);
        if (dt_event) for (int64_t s = 0; s < done; ++s) dt_event[s] = superstep_dt_event(a->seed, a->step0 + s, tot[(size_t)s]);
    }
    if (totals && ss.status == 1 && done < n) totals[done] = ss.total;
    if (n_executed && done > 0) HIPCHK(hipMemcpy(n_executed, h->d_log_nev, (size_t)done * 8, hipMemcpyDeviceToHost));
    if (events && done > 0) HIPCHK(hipMemcpy(events, d_log, (size_t)done * D * sizeof(cetkmc_event), hipMemcpyDeviceToHost));
    return 0;
}

// ---- grain clustering (utils.get_clusters / dfs_cluster, utils.py:28-84) ---------------------------
int cetkmc_cluster(void* handle, double threshold, int64_t* n_clusters)
{
    Handle* h = (Handle*)handle;
    if (!h || !n_clusters) return fail("null argument");
    if (h->slabs.size() != 1 || h->nranks != 1) return fail("cetkmc_cluster needs the whole lattice in one slab");
    HIPCHK(hipSetDevice(h->dev));
    const SlabView v = view_of(h, 0);
    const int64_t n = (int64_t)h->L * h->L * h->L;
    if (!h->d_cc_parent) {
        HIPCHK(hipMalloc((void**)&h->d_cc_parent, n * sizeof(int)));
        HIPCHK(hipMalloc((void**)&h->d_cc_roots, n * sizeof(int)));
        HIPCHK(hipMalloc((void**)&h->d_cc_cid, n * sizeof(int)));
        HIPCHK(hipMalloc((void**)&h->d_cc_labels, n * sizeof(int)));
        HIPCHK(hipMalloc((void**)&h->d_cc_n, sizeof(int)));
    }
    const int grid = (int)std::min<int64_t>((n + 255) / 256, 8192);
    HIPCHK(hipMemsetAsync(h->d_cc_n, 0, sizeof(int), h->stream));
    hipLaunchKernelGGL(k_cc_init, dim3(grid), dim3(256), 0, h->stream, v, h->d_cc_parent);
    hipLaunchKernelGGL(k_cc_hook, dim3(grid), dim3(256), 0, h->stream, v, h->d_cc_parent, threshold);
    hipLaunchKernelGGL(k_cc_compress, dim3(grid), dim3(256), 0, h->stream, n, h->d_cc_parent, h->d_cc_roots, h->d_cc_n);
    HIPCHK(hipGetLastError());
    int nr = 0;
    HIPCHK(hipMemcpyAsync(&nr, h->d_cc_n, sizeof(int), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    h->cc_roots_sorted.resize((size_t)nr);
    if (nr > 0) {
        HIPCHK(hipMemcpy(h->cc_roots_sorted.data(), h->d_cc_roots, (size_t)nr * sizeof(int), hipMemcpyDeviceToHost));
        std::sort(h->cc_roots_sorted.begin(), h->cc_roots_sorted.end());      // reference order: first voxel, row-major
        HIPCHK(hipMemcpy(h->d_cc_roots, h->cc_roots_sorted.data(), (size_t)nr * sizeof(int), hipMemcpyHostToDevice));
        if (h->d_cc_stats) { HIPCHK(hipFree(h->d_cc_stats)); h->d_cc_stats = nullptr; }
        HIPCHK(hipMalloc((void**)&h->d_cc_stats, (size_t)nr * 8 * sizeof(int)));
        const int g2 = (nr + 255) / 256;
        hipLaunchKernelGGL(k_cc_ids, dim3(g2), dim3(256), 0, h->stream, (const int*)h->d_cc_roots, nr, h->d_cc_cid);
        hipLaunchKernelGGL(k_cc_stats_init, dim3(g2), dim3(256), 0, h->stream, nr, h->d_cc_stats);
    }
    hipLaunchKernelGGL(k_cc_stats, dim3(grid), dim3(256), 0, h->stream, v, (const int*)h->d_cc_parent, (const int*)h->d_cc_cid,
                       h->d_cc_labels, h->d_cc_stats);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(h->stream));
    h->cc_n_clusters = nr;
    *n_clusters = nr;
    return 0;
}

int cetkmc_cluster_stats(void* handle, int64_t cap, int32_t* first_voxel, int64_t* size, int32_t* bbox)
{
    Handle* h = (Handle*)handle;
    if (!h) return fail("null handle");
    if (h->cc_n_clusters < 0) return fail("cetkmc_cluster_stats needs a preceding cetkmc_cluster");
    const int64_t n = std::min<int64_t>(cap, h->cc_n_clusters);
    if (n <= 0) return 0;
    std::vector<int> st((size_t)n * 8);
    HIPCHK(hipMemcpy(st.data(), h->d_cc_stats, st.size() * sizeof(int), hipMemcpyDeviceToHost));
    const int L = h->L;
    for (int64_t q = 0; q < n; ++q) {
        const int r = h->cc_roots_sorted[(size_t)q];
        if (first_voxel) { first_voxel[3 * q] = r / (L * L); first_voxel[3 * q + 1] = (r / L) % L; first_voxel[3 * q + 2] = r % L; }
        if (size) size[q] = st[8 * q];
        if (bbox) for (int c = 0; c < 6; ++c) bbox[6 * q + c] = st[8 * q + 1 + c];
    }
    return 0;
}

int cetkmc_cluster_labels(void* handle, int32_t* labels)
{
    Handle* h = (Handle*)handle;
    if (!h || !labels) return fail("null argument");
    if (h->cc_n_clusters < 0) return fail("cetkmc_cluster_labels needs a preceding cetkmc_cluster");
    HIPCHK(hipMemcpy(labels, h->d_cc_labels, (size_t)h->L * h->L * h->L * sizeof(int), hipMemcpyDeviceToHost));
    return 0;
}

int cetkmc_species_counts(void* handle, int64_t counts[6])
{
    Handle* h = (Handle*)handle;
    if (!h || !counts) return fail("null argument");
    HIPCHK(hipSetDevice(h->dev));
    DevTmp<unsigned long long> d;
    HIPCHK(d.alloc(6));
    HIPCHK(hipMemsetAsync(d, 0, 6 * sizeof(unsigned long long), h->stream));
    for (size_t s = 0; s < h->slabs.size(); ++s)
        hipLaunchKernelGGL(k_species_counts, dim3(1024), dim3(256), 0, h->stream, view_of(h, (int)s), d.p);
    HIPCHK(hipGetLastError());
    unsigned long long out[6];
    HIPCHK(hipMemcpyAsync(out, d, sizeof out, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    for (int c = 0; c < 6; ++c) counts[c] = (int64_t)out[c];
    return 0;
}

int cetkmc_gather_species(void* handle, int species, int64_t* lin_idx, double* T_vals, int64_t cap, int64_t* n)
{
    Handle* h = (Handle*)handle;
    if (!h || !n) return fail("null argument");
    HIPCHK(hipSetDevice(h->dev));
    DevTmp<unsigned long long> d_n;
    DevTmp<long long> d_idx;
    DevTmp<double> d_T;
    const size_t c = (size_t)std::max<int64_t>(cap, 1);
    HIPCHK(d_n.alloc(1));
    HIPCHK(d_idx.alloc(c));
    HIPCHK(d_T.alloc(c));
    HIPCHK(hipMemsetAsync(d_n, 0, sizeof(unsigned long long), h->stream));
    for (size_t s = 0; s < h->slabs.size(); ++s)
        hipLaunchKernelGGL(k_gather_species, dim3(1024), dim3(256), 0, h->stream, view_of(h, (int)s), species, d_idx.p, d_T.p,
                           (unsigned long long)(lin_idx && T_vals ? cap : 0), d_n.p);
    HIPCHK(hipGetLastError());
    unsigned long long cnt = 0;
    HIPCHK(hipMemcpyAsync(&cnt, d_n, sizeof cnt, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    *n = (int64_t)cnt;
    const size_t m = (size_t)std::min<int64_t>((int64_t)cnt, cap);
    if (lin_idx && T_vals && m > 0) {
        HIPCHK(hipMemcpy(lin_idx, d_idx, m * sizeof(long long), hipMemcpyDeviceToHost));
        HIPCHK(hipMemcpy(T_vals, d_T, m * sizeof(double), hipMemcpyDeviceToHost));
    }
    return 0;
}

int cetkmc_set_defects_sparse(void* handle, const int64_t* lin_idx, int64_t n)
{
    Handle* h = (Handle*)handle;
    if (!h || (n > 0 && !lin_idx)) return fail("null argument");
    HIPCHK(hipSetDevice(h->dev));
    DevTmp<long long> d_idx;
    if (n > 0) {
        HIPCHK(d_idx.alloc((size_t)n));
        HIPCHK(hipMemcpyAsync(d_idx, lin_idx, (size_t)n * sizeof(long long), hipMemcpyHostToDevice, h->stream));
    }
    for (auto& s : h->slabs) {
        HIPCHK(hipMemsetAsync(s.v.defects, 0, s.nS, h->stream));
        if (n > 0) hipLaunchKernelGGL(k_scatter_defects, dim3((unsigned)std::min<int64_t>((n + 255) / 256, 4096)), dim3(256), 0,
                                      h->stream, s.v, (const long long*)d_idx.p, (long long)n);
    }
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(h->stream));
    h->swept = false; h->ifc_fresh = false;
    return 0;
}

int64_t cetkmc_nucleation_count(void* handle)
{
    Handle* h = (Handle*)handle;
    if (!h) return -1;
    StepState ss;
    if (hipSetDevice(h->dev) != hipSuccess) return -1;
    if (hipMemcpy(&ss, h->d_ss, sizeof ss, hipMemcpyDeviceToHost) != hipSuccess) return -1;
    return ss.nuc_count;
}

int cetkmc_reset_counters(void* handle)
{
    Handle* h = (Handle*)handle;
    if (!h) return fail("null handle");
    HIPCHK(hipSetDevice(h->dev));
    HIPCHK(hipMemsetAsync(h->d_ss, 0, sizeof(StepState), h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    return 0;
}

#ifdef CETKMC_SEL_STAMPS
int cetkmc_debug_sel_stamps(long long out[16])      /* alternative builds only (tools/sel_stamps.py); not part of the ABI */
{
    HIPCHK(hipDeviceSynchronize());
    HIPCHK(hipMemcpyFromSymbol(out, HIP_SYMBOL(cetkmc::g_sel_stamps), 16 * sizeof(long long)));
    return 0;
}
#endif

int cetkmc_time_sweeps(void* handle, int n, double* ms_total)
{
    Handle* h = (Handle*)handle;
    if (!h || !ms_total) return fail("null argument");
    HIPCHK(hipSetDevice(h->dev));
    HIPCHK(hipEventRecord(h->ev0, h->stream));
    for (int s = 0; s < n; ++s) CHK(launch_sweep(h, false));
    HIPCHK(hipEventRecord(h->ev1, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    float ms = 0.f;
    HIPCHK(hipEventElapsedTime(&ms, h->ev0, h->ev1));
    *ms_total = ms;
    return 0;
}

int cetkmc_event_overhead(void* handle, int n, double* ms_avg)
{
    Handle* h = (Handle*)handle;
    if (!h || !ms_avg || n < 1) return fail("bad argument");
    HIPCHK(hipSetDevice(h->dev));
    while ((int)h->prof.size() < 2 * n) { hipEvent_t e; HIPCHK(hipEventCreate(&e)); h->prof.push_back(e); }
    for (int q = 0; q < n; ++q) {
        // a kernel before the pair as well: the bracketed launches of a batch follow other kernels back to back
        hipLaunchKernelGGL(k_empty, dim3(1), dim3(64), 0, h->stream);
        HIPCHK(hipEventRecord(h->prof[2 * q], h->stream));
        hipLaunchKernelGGL(k_empty, dim3(1), dim3(64), 0, h->stream);
        HIPCHK(hipEventRecord(h->prof[2 * q + 1], h->stream));
    }
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(h->stream));
    double tot = 0.0;
    for (int q = 0; q < n; ++q) { float t = 0.f; HIPCHK(hipEventElapsedTime(&t, h->prof[2 * q], h->prof[2 * q + 1])); tot += t; }
    *ms_avg = tot / n;
    return 0;
}

// Transport self-test: every collective shape the stepping loops use, on patterned buffers, checked on the host.
//   all-gather of `bytes` per rank (Mode A: block sums / event records) and the neighbour exchange of `bytes` each way
//   (temperature halo, Mode B boundary layers).  A single-rank RCCL communicator sends to and receives from itself.
// times_us (optional, 2 doubles): average of 20 back-to-back all-gathers / exchanges after the checked one.
int cetkmc_comm_selftest(void* handle, int64_t bytes, double* times_us)
{
    Handle* h = (Handle*)handle;
    if (!h || bytes < 1 || bytes > (1 << 26)) return fail("bad argument");
    HIPCHK(hipSetDevice(h->dev));
    if (times_us) times_us[0] = times_us[1] = 0.0;
    if (!multi_rank(h)) return 0;
    const int R = h->nranks, me = h->rank;
    const size_t B = (size_t)bytes;
    auto pat = [](int from, int tag, size_t q) -> unsigned char { return (unsigned char)(from * 37 + tag * 101 + q * 7 + (q >> 8)); };
    char* d = nullptr;          // [gather: R*B | send_lo | send_hi | recv_lo | recv_hi]
    HIPCHK(hipMalloc(&d, (R + 4) * B));
    std::vector<unsigned char> host((R + 4) * B, 0xEE);
    for (size_t q = 0; q < B; ++q) {
        host[me * B + q] = pat(me, 0, q);
        host[(R + 0) * B + q] = pat(me, 1, q);      // to rank-1
        host[(R + 1) * B + q] = pat(me, 2, q);      // to rank+1
    }
    HIPCHK(hipMemcpy(d, host.data(), host.size(), hipMemcpyHostToDevice));
    int rc = comm_allgather(h, d, B);
    if (!rc) {
        if (h->comm && R == 1) {            // loopback: the Send/Recv signatures and group semantics on this build of RCCL
            NCCLCHK(g_rccl.GroupStart());
            NCCLCHK(g_rccl.Send(d + (R + 0) * B, B, ncclChar, 0, h->comm, h->stream));
            NCCLCHK(g_rccl.Recv(d + (R + 2) * B, B, ncclChar, 0, h->comm, h->stream));
            NCCLCHK(g_rccl.GroupEnd());
        } else {
            rc = comm_exchange(h, d + (R + 0) * B, d + (R + 2) * B, d + (R + 1) * B, d + (R + 3) * B, B);
        }
    }
    if (rc) { (void)hipFree(d); return rc; }
    HIPCHK(hipStreamSynchronize(h->stream));
    HIPCHK(hipMemcpy(host.data(), d, host.size(), hipMemcpyDeviceToHost));
    std::string bad;
    for (int r = 0; r < R && bad.empty(); ++r)
        for (size_t q = 0; q < B; ++q)
            if (host[r * B + q] != pat(r, 0, q)) { bad = "all-gather: part of rank " + std::to_string(r) + " wrong at byte " + std::to_string(q); break; }
    if (bad.empty() && h->comm && R == 1) {
        for (size_t q = 0; q < B; ++q) if (host[(R + 2) * B + q] != pat(0, 1, q)) { bad = "loopback send/recv wrong at byte " + std::to_string(q); break; }
    } else if (bad.empty()) {
        for (size_t q = 0; q < B && bad.empty(); ++q) {
            // from rank-1 comes what it sent "to rank+1" (tag 2); from rank+1 what it sent "to rank-1" (tag 1)
            if (me > 0 && host[(R + 2) * B + q] != pat(me - 1, 2, q)) bad = "exchange: data from rank-1 wrong at byte " + std::to_string(q);
            if (me < R - 1 && host[(R + 3) * B + q] != pat(me + 1, 1, q)) bad = "exchange: data from rank+1 wrong at byte " + std::to_string(q);
            if (me == 0 && host[(R + 2) * B + q] != 0xEE) bad = "exchange: end rank's unused receive buffer was written";
        }
    }
    if (!bad.empty()) { (void)hipFree(d); return fail("transport self-test (rank " + std::to_string(me) + "): " + bad); }
    if (times_us) {
        for (int leg = 0; leg < 2; ++leg) {
            HIPCHK(hipStreamSynchronize(h->stream));
            const auto t0 = std::chrono::steady_clock::now();
            for (int q = 0; q < 20 && !rc; ++q)
                rc = leg == 0 ? comm_allgather(h, d, B)
                              : comm_exchange(h, d + (R + 0) * B, d + (R + 2) * B, d + (R + 1) * B, d + (R + 3) * B, B);
            if (rc) { (void)hipFree(d); return rc; }
            HIPCHK(hipStreamSynchronize(h->stream));
            times_us[leg] = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / 20.0;
        }
    }
    HIPCHK(hipFree(d));
    return 0;
}

}  // extern "C"
