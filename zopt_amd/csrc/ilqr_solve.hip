// K8  ilqr_solve -- the iLQR / DDP outer loop as ONE C-ABI call: the host side of the iteration (launch sequence, compacted
// list of still-active trajectories, termination test) runs here in C++ instead of in Python.
//
// Replaces: zopt/ilqrUtils.py:290-327 (iterativeLqr) and :360-397 (differentialDynamicProgramming), per trajectory
//     policy = (uGuess, 0); traj = rollout(x0, policy, zeros); J = cost(traj)                                   (:293-298)
//     while not converged and it < maxIter:                                                                      (:301-303)
//         expansions along traj; PD-conditioned Hessians; backward pass; 16-way line search; converged = |J - Jn| <= tol
// on a batch: every kernel runs over the compacted id list of the trajectories that have not converged yet (rebuilt on the
// device every `sync_every` iterations, when the host also learns how many are left), converged trajectories drop out.
// The kernels are the ones behind the array-level entry points (linearize.hip, ilqr_backward.hip, rollout*.hip, psd.hip);
// nothing is allocated here: the caller provides the workspace (zm_ilqr_solve_workspace_f64 tells how much).
#include <utility>
#include <cstdint>
#include <cstdlib>
#include <mutex>
#include <vector>

#include "models.h"
#include "quad_derivs_gen.h"
#include "zm_common.h"

namespace zm {

// Stable compaction of the active ids (ascending trajectory order, as torch.nonzero gave the Python loop): one block, one
// ballot-scan pass per 1024 trajectories.  count[0] <- number of active trajectories.
// hostword (pinned host memory, may be null): <- (seq << 32 | count) as ONE 8-byte system-scope store -- the host side of the solve
// polls it instead of waiting for a copy kernel, an event and the interrupt behind hipEventSynchronize (~45 us per round trip).
__global__ __launch_bounds__(1024) void compact_active_kernel(const int* __restrict__ active, const long batch, int* __restrict__ list,
                                                              int* __restrict__ count, unsigned long long* hostword,
                                                              const unsigned seq) {
    __shared__ int wsum[16];
    __shared__ int base;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    if (tid == 0) base = 0;
    __syncthreads();
    for (long start = 0; start < batch; start += 1024) {
        const long i = start + tid;
        const bool a = i < batch && active[i] != 0;
        const unsigned long long m = __ballot(a);
        const int before = __builtin_popcountll(m & ((1ull << lane) - 1ull));
        if (lane == 0) wsum[w] = __builtin_popcountll(m);
        __syncthreads();
        int off = base;
        for (int j = 0; j < w; ++j) off += wsum[j];
        if (a) list[off + before] = (int)i;
        __syncthreads();
        if (tid == 0) {
            int s = 0;
            for (int j = 0; j < 16; ++j) s += wsum[j];
            base += s;
        }
        __syncthreads();
    }
    if (tid == 0) {
        count[0] = base;
        if (hostword) __hip_atomic_store(hostword, ((unsigned long long)seq << 32) | (unsigned)base, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

// Start of a solve in ONE launch (it was seven memset / memcpy / fill calls, each a runtime call of its own on the host):
// l <- uGuess, L <- 0 (policy0 = (uGuess, 0), ilqrUtils.py:293), previous trajectory <- 0 (:294), alphas <- 0.5 ** arange(16) (:145),
// active <- 1, converged <- 0, where <- 0.
__global__ __launch_bounds__(256) void ilqr_init_kernel(double* __restrict__ l, const double* __restrict__ uGuess, const long nl,
                                                        double* __restrict__ L, const long nL, double* __restrict__ xT2, const long nx,
                                                        double* __restrict__ uT2, double* __restrict__ alphas, int* __restrict__ active,
                                                        int* __restrict__ converged, int* __restrict__ where, const long batch) {
    const long stride = (long)gridDim.x * blockDim.x, t0 = blockIdx.x * (long)blockDim.x + threadIdx.x;
    for (long i = t0; i < nL; i += stride) L[i] = 0.0;
    for (long i = t0; i < nx; i += stride) xT2[i] = 0.0;
    for (long i = t0; i < nl; i += stride) {
        l[i] = uGuess[i];
        uT2[i] = 0.0;
    }
    for (long i = t0; i < batch; i += stride) {
        active[i] = 1;
        converged[i] = 0;
        if (where) where[i] = 0;
    }
    if (t0 < 16) alphas[t0] = 1.0 / (double)(1u << t0);   // exact powers of two
}

__global__ void fill_i32_kernel(int* __restrict__ p, const long n, const int v) {
    const long i = blockIdx.x * (long)blockDim.x + threadIdx.x;
    if (i < n) p[i] = v;
}

__global__ void fill_alphas_kernel(double* __restrict__ a) {
    a[threadIdx.x] = 1.0 / (double)(1u << threadIdx.x);   // exact powers of two
}

__global__ void fill_f64_kernel(double* __restrict__ p, const long n, const double v) {
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) p[i] = v;
}

struct IlqrWs {   // carve of the caller's workspace (doubles)
    long l, xT2, uT2, Jn, f_x, f_u, c_x, c_u, v_x, c_xx, c_ux, c_uu, v_xx, alphas, f_xx, f_ux, f_uu, idx, where, scratch, tail_slots, total;
};

// Once at most this many trajectories are left, the line search runs in all-store mode (rollout_fast.hip): its 16 rollouts per
// trajectory go to scratch rows and the accept step copies the winner, instead of re-rolling the winner in a second pass of T more
// dependent steps.  With few trajectories a launch lasts as long as one wave's chain, so this halves it; with many, the 16x stores
// (207 KB per trajectory at n=12, m=4, T=100) would cost more HBM time than the second pass costs ALU time.
// ZOPT_AMD_ILQR_TAIL=<n> overrides the threshold (0: never).
constexpr long ILQR_TAIL_DEFAULT = 2048;
static long ilqr_tail_slots(long batch) {
    static const long thr = [] {
        const char* e = zm::lab_env("ZOPT_AMD_ILQR_TAIL");
        return e ? atol(e) : ILQR_TAIL_DEFAULT;
    }();
    return thr < 0 ? 0 : (thr < batch ? thr : batch);
}

int expand_list(const zm_model_t* model, const zm_quadcost_t* cost, const double* xTraj, const double* uTraj, const int32_t* list,
                int64_t count, const int32_t* active, double* f_x, double* f_u, double* c_x, double* c_u, double* v_x, int64_t batch,
                int T, void* stream, int packed);
struct TrajList {
    const int* list;
    long count;
};
int sweep_packed_jacobians(const double* Fp, int njp, const unsigned char* pos, const double* Hpk, const PairTab& ptab, const double* c_x,
                           const double* c_u, const double* c_xx, const double* c_ux, const double* c_uu, const double* vf_x,
                           const double* vf_xx, const int* act, double* l, double* L, int64_t batch, int T, hipStream_t st, TrajList tl,
                           int nhs, const unsigned short* hdense, int nh);
int quad_hessian_sparse_list(const zm_model_t* model, const double* xTraj, const double* uTraj, const int32_t* list, int64_t count,
                             const int32_t* active, double* Hs, int64_t batch, int T, void* stream);
bool rollout_all_store_supported(const zm_model_t* model, const zm_quadcost_t* cost, int T);
bool rollout_all_store_model_ok(const zm_model_t* model);
int rollout_linesearch_all_store(const zm_model_t* model, const zm_quadcost_t* cost, const double* x0, const double* l, const double* L,
                                 const double* xPrev, const double* uPrev, const double* alphas, const int32_t* list, int64_t count,
                                 const int32_t* active, double* scratch, double* J, int32_t* alpha_idx, int64_t batch, int T,
                                 void* stream);
int ilqr_accept(const int32_t* list, int64_t count, double* J, const double* Jn, double* xTraj, const double* xTrajNew, double* uTraj,
                const double* uTrajNew, int32_t* converged, int32_t* active, double tol, int64_t batch, int T, int n, int m,
                void* stream, const double* scratch, const int32_t* idx, int32_t* where, int where_val);
int ilqr_collect(const int32_t* where, double* xTraj, const double* xAlt, double* uTraj, const double* uAlt, int64_t batch, int T, int n,
                 int m, void* stream);

static IlqrWs carve(const zm_model_t* model, long b, long T, int ddp, int need_ux, int npairs) {
    const long n = model->n, m = model->m;
    IlqrWs w;
    long o = 0;
    auto take = [&](long cnt) {
        const long at = o;
        o += (cnt + 1) & ~1L;   // 16-B aligned pieces (the sweep kernels' DMA path needs it)
        return at;
    };
    w.l = take(b * T * m);
    w.xT2 = take(b * (T + 1) * n);
    w.uT2 = take(b * T * m);
    w.Jn = take(b);
    w.f_x = take(b * T * n * n);
    w.f_u = take(b * T * n * m);
    w.c_x = take(b * T * n);
    w.c_u = take(b * T * m);
    w.v_x = take(b * n);
    w.c_xx = take(n * n);
    w.c_ux = take(m * n);
    w.c_uu = take(m * m);
    w.v_xx = take(n * n);
    w.alphas = take(16);
    // second derivatives: packed (pairs x n per point) when the model declares its nonzero pairs, else the full tensors
    w.f_xx = ddp ? take(npairs > 0 ? b * T * npairs * n : b * T * n * n * n) : 0;
    w.f_ux = (ddp && need_ux && npairs == 0) ? take(b * T * n * m * n) : 0;
    w.f_uu = (ddp && need_ux && npairs == 0) ? take(b * T * n * m * m) : 0;
    w.idx = take((b + 1) / 2);   // int32 winner index per trajectory
    w.where = take((b + 1) / 2); // int32 per trajectory: 0 = newest rows in the caller's xTraj / uTraj, 1 = in xT2 / uT2
    // all-store blocks (16 step sizes x 16 doubles per step and slot: 423 MB at T = 100 for 2048 slots) only for the models whose line
    // search can run in that form -- a windy quadcopter, another model or the generic rollout path never touch them
    w.tail_slots = rollout_all_store_model_ok(model) ? ilqr_tail_slots(b) : 0;
    w.scratch = take(w.tail_slots * (T + 1) * 256);
    w.total = o;
    return w;
}

// Pool of (pinned int32, event) pairs per device for the drivers' host round trips.
struct HostSlot {
    int dev;
    unsigned long long* word;   // pinned: (seq << 32 | count), written by compact_active_kernel
    hipEvent_t event;
    bool busy;
    unsigned seq;               // last sequence number handed out for this word
};
static std::mutex g_slot_mutex;
static std::vector<HostSlot> g_slots;

class HostSlotLease {
  public:
    explicit HostSlotLease(int dev) : idx_(-1) {
        std::lock_guard<std::mutex> lk(g_slot_mutex);
        for (size_t i = 0; i < g_slots.size(); ++i)
            if (g_slots[i].dev == dev && !g_slots[i].busy) {
                g_slots[i].busy = true;
                idx_ = (long)i;
                word_ = g_slots[i].word;
                event_ = g_slots[i].event;
                return;
            }
        HostSlot s{dev, nullptr, nullptr, true, 0u};
        if (hipHostMalloc((void**)&s.word, sizeof(unsigned long long), hipHostMallocCoherent | hipHostMallocMapped) != hipSuccess) return;
        *s.word = 0ull;
        if (hipEventCreateWithFlags(&s.event, hipEventDisableTiming) != hipSuccess) {
            (void)hipHostFree(s.word);
            return;
        }
        g_slots.push_back(s);
        idx_ = (long)g_slots.size() - 1;
        word_ = s.word;
        event_ = s.event;
    }
    ~HostSlotLease() {
        if (idx_ < 0) return;
        std::lock_guard<std::mutex> lk(g_slot_mutex);
        if ((size_t)idx_ < g_slots.size() && g_slots[idx_].word == word_) g_slots[idx_].busy = false;
    }
    bool ok() const { return idx_ >= 0; }
    unsigned long long* word() const { return word_; }
    hipEvent_t event() const { return event_; }
    unsigned next_seq() {   // a fresh tag per round trip (never 0: the word starts as 0)
        std::lock_guard<std::mutex> lk(g_slot_mutex);
        unsigned& q = g_slots[idx_].seq;
        q = (q == 0xffffffffu) ? 1u : q + 1u;
        return q;
    }

  private:
    long idx_;
    unsigned long long* word_ = nullptr;
    hipEvent_t event_ = nullptr;
};

// Waits until the device has published (seq, count) in the pinned word; returns count.  Busy-polls (the answer is a few
// microseconds away: one small kernel behind what is already queued); after `spin_limit` polls without it -- a slow or
// profiled run -- it falls back to the event behind the kernel and the device-side copy of the count.
static int wait_for_count(const unsigned long long* word, const unsigned seq, hipEvent_t ev, const int32_t* dcount, int32_t* out) {
    for (long spin = 0; spin < 20000000L; ++spin) {
        const unsigned long long v = __atomic_load_n(word, __ATOMIC_ACQUIRE);
        if ((unsigned)(v >> 32) == seq) {
            *out = (int32_t)(unsigned)(v & 0xffffffffull);
            return ZM_OK;
        }
        __builtin_ia32_pause();
    }
    ZM_HIP_CHECK(hipEventSynchronize(ev));
    ZM_HIP_CHECK(hipMemcpy(out, dcount, sizeof(int32_t), hipMemcpyDeviceToHost));
    return ZM_OK;
}

// zm_shutdown(): frees every pair that is not lent out; returns how many were released.
int release_host_slots() {
    std::lock_guard<std::mutex> lk(g_slot_mutex);
    int freed = 0;
    std::vector<HostSlot> keep;
    for (auto& s : g_slots) {
        if (s.busy) {
            keep.push_back(s);
            continue;
        }
        (void)hipEventDestroy(s.event);
        (void)hipHostFree(s.word);
        ++freed;
    }
    g_slots.swap(keep);
    return freed;
}

}  // namespace zm

extern "C" int64_t zm_ilqr_solve_workspace_f64(const zm_model_t* model, int64_t batch, int T, int ddp) {
    if (!model || batch < 0 || T < 1) return -1;
    uint32_t mask = 0;
    int32_t npairs = 0;
    if (ddp && (zm_model_nonlinear_mask(model, &mask) != ZM_OK || zm_model_hessian_pairs(model, nullptr, &npairs) != ZM_OK)) return -1;
    return zm::carve(model, batch, T, ddp, (mask >> model->n) != 0, npairs).total;
}

static int ilqr_solve_impl(const zm_model_t* model, const zm_quadcost_t* cost, const double* x0, const double* uGuess,
                           int ddp, int max_iter, double tol, int sync_every, double* workspace, int64_t workspace_doubles,
                           int32_t* iwork, double* xTraj, double* uTraj, double* L, double* J, int32_t* converged,
                           int32_t* iterations, int64_t batch, int T, void* stream, double* J_trace, int32_t* alpha_trace) {
    if (batch == 0) return ZM_OK;
    if (!model || !cost || !x0 || !uGuess || !workspace || !iwork || !xTraj || !uTraj || !L || !J || !converged)
        return zm::set_error(ZM_EINVAL, "zm_ilqr_solve_f64: null pointer");
    if (batch < 0 || T < 1 || max_iter < 0 || sync_every < 1) return zm::set_error(ZM_EINVAL, "zm_ilqr_solve_f64: bad size");
    const int n = model->n, m = model->m;
    uint32_t mask = 0;
    int32_t npairs = 0;
    if (ddp) {
        int rc = zm_model_nonlinear_mask(model, &mask);
        if (rc) return rc;
        rc = zm_model_hessian_pairs(model, nullptr, &npairs);
        if (rc) return rc;
    }
    const int need_ux = (mask >> n) != 0;   // a model affine in its controls has f_ux = f_uu = 0: neither written nor read
    const zm::IlqrWs w = zm::carve(model, batch, T, ddp, need_ux, npairs);
    if (workspace_doubles < w.total)
        return zm::set_error(ZM_EINVAL, "zm_ilqr_solve_f64: workspace of %lld doubles, %lld needed", (long long)workspace_doubles,
                             (long long)w.total);
    hipStream_t st = (hipStream_t)stream;
    double* ws = workspace;
    // iwork (int32): active (batch) | list (batch) | count (2)
    int32_t* active = iwork;
    int32_t* list = iwork + batch;
    int32_t* dcount = iwork + 2 * batch;
    const long xrow = (long)(T + 1) * n, urow = (long)T * m;

    // policy = (uGuess, 0); traj_prev = zeros; step sizes; masks                                     (ilqrUtils.py:293-294, :145)
    int32_t* where = (int32_t*)(ws + w.where);
    {
        const long nL = (long)batch * urow * n, nx = (long)batch * xrow, nl = (long)batch * urow;
        const long work = nL > nx ? nL : nx;
        const unsigned blocks = (unsigned)((work + 256L * 8 - 1) / (256L * 8) < 4096 ? (work + 256L * 8 - 1) / (256L * 8) : 4096);
        hipLaunchKernelGGL(zm::ilqr_init_kernel, dim3(blocks ? blocks : 1), dim3(256), 0, st, ws + w.l, uGuess, nl, L, nL, ws + w.xT2, nx,
                           ws + w.uT2, ws + w.alphas, (int*)active, (int*)converged, (int*)where, (long)batch);
    }
    // initial rollout (alpha = 1) from the zero trajectory and its cost                                (:297-298)
    int rc = zm_rollout_linesearch_f64(model, cost, x0, ws + w.l, L, ws + w.xT2, ws + w.uT2, ws + w.alphas, 1, nullptr, xTraj, uTraj, J,
                                       nullptr, batch, T, st);
    if (rc) return rc;
    // trajectory-independent Hessians of the quadratic cost, PD-conditioned once                         (:309-313)
    rc = zm_quadratize_cost_f64(cost, n, m, xTraj, uTraj, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, ws + w.c_xx,
                                ws + w.c_ux, ws + w.c_uu, ws + w.v_xx, 0, T, st);
    if (rc) return rc;
    rc = zm_condition_cost_f64(ws + w.c_xx, ws + w.c_ux, ws + w.c_uu, 1, n, m, 1e-3, st);
    if (rc) return rc;
    rc = zm_psd_project_f64(ws + w.v_xx, 1, n, 1e-3, st);
    if (rc) return rc;

    double* f_ux = (need_ux && npairs == 0) ? ws + w.f_ux : nullptr;
    double* f_uu = (need_ux && npairs == 0) ? ws + w.f_uu : nullptr;
    const bool can_all_store = w.tail_slots > 0 && zm::rollout_all_store_supported(model, cost, T);
    // Packed Jacobians (quadcopter): the expansion writes and the ring sweeps read only the structurally nonzero entries of
    // [f_x | f_u] -- 448 B (480 with wind) per point instead of 1 536 B.  ZOPT_AMD_JAC=full or ZOPT_AMD_ILQR_PATH=reg: the full matrices.
    static const bool packed_off = [] {
        const char* e = zm::lab_env("ZOPT_AMD_JAC");
        const char* r = zm::fallback_env("ZOPT_AMD_ILQR_PATH");
        return (e && e[0] == 'f') || (r && r[0] == 'r');
    }();
    const bool windy = model->wind_ned[0] != 0.0 || model->wind_ned[1] != 0.0 || model->wind_ned[2] != 0.0;
    const bool packed = !packed_off && model->kind == ZM_MODEL_QUADCOPTER && model->dt != 0.0 && (((uintptr_t)ws) & 15) == 0 &&
                        (!ddp || npairs == 28);
    const int njp = windy ? ((zm::QUAD_NJ_WIND + 1) & ~1) : ((zm::QUAD_NJ_STILL + 1) & ~1);
    const unsigned char* jpos = windy ? zm::QUAD_JPOS_WIND : zm::QUAD_JPOS_STILL;
    const zm::PairTab ptab = zm::model_pair_table(model->kind);
    // ... and the DDP path's second derivatives SPARSE (69 / 85 structurally nonzero entries of the 28 x 12 per point; ZOPT_AMD_HES=dense:
    // the dense pair rows)
    static const bool sparse_off = [] {
        const char* e = zm::lab_env("ZOPT_AMD_HES");
        return e && e[0] == 'd';
    }();
    const bool sparse_h = packed && ddp && !sparse_off;
    const int nh = windy ? zm::QUAD_NH_WIND : zm::QUAD_NH_STILL, nhs = (nh + 1) & ~1;
    const unsigned short* hdense = windy ? zm::QUAD_HDENSE_WIND : zm::QUAD_HDENSE_STILL;
    int32_t* widx = (int32_t*)(ws + w.idx);
    // pinned word + event for the active count, borrowed from a per-device pool for the duration of this call (a pinned allocation
    // per solve costs ~1 ms; an event belongs to the device that was current when it was created).  Concurrent solves on one device
    // get distinct pairs; zm_shutdown() releases the pool.
    int dev = 0;
    ZM_HIP_CHECK(hipGetDevice(&dev));
    zm::HostSlotLease lease(dev);
    if (!lease.ok()) return zm::set_error(ZM_EUNSUPPORTED, "zm_ilqr_solve_f64: cannot get a pinned word / event on device %d", dev);
    unsigned long long* const hword = lease.word();
    const hipEvent_t count_ready = lease.event();
    // The trajectories alternate between two buffers: the two-pass line search reads the current rows (xPrev, uPrev) and writes the
    // winner's into the other buffer, which then IS the current one -- the acceptance step copies nothing (it was 212 MB per
    // full-batch iteration).  Every active trajectory is rewritten in every iteration, so the current buffer is the same for all of
    // them; a trajectory that retires stays where it was last written (`where`), and the rows left in the workspace buffer move to
    // the caller's arrays once, at the end.  (ZOPT_AMD_ILQR_SWAP=0: the acceptance step copies, as before; same results.)
    static const bool swap_on = [] {
        const char* e = zm::lab_env("ZOPT_AMD_ILQR_SWAP");
        return !(e && e[0] == '0');
    }();
    double *curX = xTraj, *curU = uTraj, *altX = ws + w.xT2, *altU = ws + w.uT2;
    int it = 0;
    int64_t count = batch;
    for (; it < max_iter; ++it) {                                                                        // (:301-303)
        // Every `sync_every` iterations: rebuild the id list on the device and learn how many trajectories are left.  The expansion
        // launch does not wait for that answer: it goes out behind the compaction with the PREVIOUS count (list entries past the new
        // count are ids of the previous list: inactive ones are skipped by the mask, a still-active one is merely expanded twice --
        // the same values into the same rows), and the host round trip (~45 us) passes while it runs.
        // (Measured: a longer interval once the active set is small does not pay -- a stale list costs the DDP tail 1-3 %.)
        const bool sync_now = (it % sync_every == 0);
        unsigned seq = 0;
        if (sync_now) {
            seq = lease.next_seq();
            hipLaunchKernelGGL(zm::compact_active_kernel, dim3(1), dim3(1024), 0, st, (const int*)active, (long)batch, (int*)list,
                               (int*)dcount, hword, seq);
            ZM_HIP_CHECK(hipEventRecord(count_ready, st));   // only waited for if the pinned word does not show up (wait_for_count)
        }
        // expansions along the current trajectories: [f_x | f_u], c_x, c_u, v_x in one launch                  (:304-313)
        rc = zm::expand_list(model, cost, curX, curU, list, count, active, ws + w.f_x, ws + w.f_u, ws + w.c_x, ws + w.c_u,
                             ws + w.v_x, batch, T, st, packed ? 1 : 0);
        if (rc) return rc;
        if (sync_now) {
            int32_t c32 = 0;
            rc = zm::wait_for_count(hword, seq, count_ready, dcount, &c32);
            if (rc) return rc;
            count = c32;
            if (count == 0) break;
        }
        if (packed) {
            if (sparse_h) {
                rc = zm::quad_hessian_sparse_list(model, curX, curU, list, count, active, ws + w.f_xx, batch, T, st);
                if (rc) return rc;
            } else if (ddp) {
                rc = zm_quadratic_dynamics_pairs_list_f64(model, curX, curU, list, count, active, ws + w.f_xx, batch, T, st);
                if (rc) return rc;
            }
            rc = zm::sweep_packed_jacobians(ws + w.f_x, njp, jpos, ddp ? ws + w.f_xx : nullptr, ptab, ws + w.c_x, ws + w.c_u, ws + w.c_xx,
                                            ws + w.c_ux, ws + w.c_uu, ws + w.v_x, ws + w.v_xx, (const int*)active, ws + w.l, L, batch, T,
                                            (hipStream_t)st, zm::TrajList{(const int*)list, (long)count}, sparse_h ? nhs : 0, hdense, nh);
            if (rc == ZM_EUNSUPPORTED) return zm::set_error(ZM_EUNSUPPORTED, "zm_ilqr_solve_f64: packed sweep refused its operands");
        } else if (ddp && npairs > 0) {
            // packed second derivatives: 28 x 12 doubles per point for the quadcopter instead of the zero-filled (n,n,n) tensors
            // (2.7 KB instead of 13.8 KB written by the expansion and read back by the sweep, same arithmetic)
            rc = zm_quadratic_dynamics_pairs_list_f64(model, curX, curU, list, count, active, ws + w.f_xx, batch, T, st);
            if (rc) return rc;
            rc = zm_ddp_backward_pairs_list_f64(model, ws + w.f_x, ws + w.f_u, ws + w.f_xx, ws + w.c_x, ws + w.c_u, ws + w.c_xx,
                                                ws + w.c_ux, ws + w.c_uu, ws + w.v_x, ws + w.v_xx, list, count, active, 1, ws + w.l,
                                                L, batch, T, st);
        } else if (ddp) {
            rc = zm_quadratic_dynamics_list_f64(model, curX, curU, list, count, active, ws + w.f_xx, f_ux, f_uu, batch, T, st);
            if (rc) return rc;
            rc = zm_ddp_backward_list_f64(ws + w.f_x, ws + w.f_u, ws + w.f_xx, f_ux, f_uu, ws + w.c_x, ws + w.c_u, ws + w.c_xx,
                                          ws + w.c_ux, ws + w.c_uu, ws + w.v_x, ws + w.v_xx, list, count, active, 1, ws + w.l, L,
                                          batch, T, n, m, st);
        } else {
            rc = zm_ilqr_backward_list_f64(ws + w.f_x, ws + w.f_u, ws + w.c_x, ws + w.c_u, ws + w.c_xx, ws + w.c_ux, ws + w.c_uu,
                                           ws + w.v_x, ws + w.v_xx, list, count, active, 1, ws + w.l, L, batch, T, n, m, st);
        }
        if (rc) return rc;
        // 16-way line search (:116-150), then accept: converged = |J - J_new| <= tol, (traj, J) <- (traj_new, J_new) on the listed
        // rows (:316-320)
        if (can_all_store && count <= w.tail_slots) {
            rc = zm::rollout_linesearch_all_store(model, cost, x0, ws + w.l, L, curX, curU, ws + w.alphas, list, count, active,
                                                  ws + w.scratch, ws + w.Jn, widx, batch, T, st);
            if (rc) return rc;
            rc = zm::ilqr_accept(list, count, J, ws + w.Jn, curX, nullptr, curU, nullptr, converged, active, tol, batch, T, n, m, st,
                                 ws + w.scratch, widx, swap_on ? where : nullptr, curX == xTraj ? 0 : 1);
        } else if (swap_on) {
            rc = zm_rollout_linesearch_list_f64(model, cost, x0, ws + w.l, L, curX, curU, ws + w.alphas, 16, list, count, active, altX,
                                                altU, ws + w.Jn, widx, batch, T, st);
            if (rc) return rc;
            rc = zm::ilqr_accept(list, count, J, ws + w.Jn, altX, nullptr, altU, nullptr, converged, active, tol, batch, T, n, m, st,
                                 nullptr, nullptr, where, altX == xTraj ? 0 : 1);
            std::swap(curX, altX);
            std::swap(curU, altU);
        } else {
            rc = zm_rollout_linesearch_list_f64(model, cost, x0, ws + w.l, L, xTraj, uTraj, ws + w.alphas, 16, list, count, active,
                                                ws + w.xT2, ws + w.uT2, ws + w.Jn, widx, batch, T, st);
            if (rc) return rc;
            rc = zm_ilqr_accept_f64(list, count, J, ws + w.Jn, xTraj, ws + w.xT2, uTraj, ws + w.uT2, converged, active, tol, batch, T,
                                    n, m, st);
        }
        if (rc) return rc;
        // diagnostics (zm_ilqr_solve_trace_f64): row `it` <- every trajectory's cost after this iteration's acceptance step and the
        // line search's winning step-size index (meaningful for the trajectories that were active in this iteration)
        if (J_trace) ZM_HIP_CHECK(hipMemcpyAsync(J_trace + (long)it * batch, J, sizeof(double) * batch, hipMemcpyDeviceToDevice, st));
        if (alpha_trace)
            ZM_HIP_CHECK(hipMemcpyAsync(alpha_trace + (long)it * batch, widx, sizeof(int32_t) * batch, hipMemcpyDeviceToDevice, st));
    }
    if (swap_on) {
        rc = zm::ilqr_collect(where, xTraj, ws + w.xT2, uTraj, ws + w.uT2, batch, T, n, m, st);
        if (rc) return rc;
    }
    if (iterations) *iterations = it;
    return ZM_OK;
}

extern "C" int zm_ilqr_solve_f64(const zm_model_t* model, const zm_quadcost_t* cost, const double* x0, const double* uGuess,
                                 int ddp, int max_iter, double tol, int sync_every, double* workspace, int64_t workspace_doubles,
                                 int32_t* iwork, double* xTraj, double* uTraj, double* L, double* J, int32_t* converged,
                                 int32_t* iterations, int64_t batch, int T, void* stream) {
    return ilqr_solve_impl(model, cost, x0, uGuess, ddp, max_iter, tol, sync_every, workspace, workspace_doubles, iwork, xTraj, uTraj, L,
                           J, converged, iterations, batch, T, stream, nullptr, nullptr);
}

extern "C" int zm_ilqr_solve_trace_f64(const zm_model_t* model, const zm_quadcost_t* cost, const double* x0, const double* uGuess,
                                       int ddp, int max_iter, double tol, int sync_every, double* workspace,
                                       int64_t workspace_doubles, int32_t* iwork, double* xTraj, double* uTraj, double* L, double* J,
                                       int32_t* converged, int32_t* iterations, int64_t batch, int T, void* stream, double* J_trace,
                                       int32_t* alpha_trace) {
    return ilqr_solve_impl(model, cost, x0, uGuess, ddp, max_iter, tol, sync_every, workspace, workspace_doubles, iwork, xTraj, uTraj, L,
                           J, converged, iterations, batch, T, stream, J_trace, alpha_trace);
}

extern "C" int zm_shutdown(void) { return zm::release_host_slots(); }
