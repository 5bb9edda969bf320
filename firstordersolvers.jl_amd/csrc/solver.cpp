// libfoship: handle life cycle, the device-resident CG driver, the GAP/GAPA/FISTA/Dykstra step loops, the status
// check and the C ABI of include/foship.h.  Host C++ only orchestrates launches on ONE HIP stream; all data stays
// in HBM between fos_create and fos_destroy (the only transfers are N doubles at set/get-iterate and ~100 bytes of
// scalars per CG poll / convergence check).
#include <dlfcn.h>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <rccl/rccl.h>
#include <functional>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <memory>
#include <mutex>
#include <thread>

#include "fos_internal.hpp"

namespace fos {
const char* last_error_cstr();

typedef double2 d2;

// ------------------------------------------------------------------------------------------------ RCCL via dlopen
// (no link-time dependency: single-GPU users never load it; inside a torch process dlopen returns the copy torch
// already mapped, so both share one RCCL instance)
struct Rccl {
    void* lib = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
};
static Rccl g_rccl;

// ------------------------------------------------------------------------------------------------ roctx ranges
// Named ranges around the phases of an outer iteration (affine projection / cone projection / status check) for
// `rocprofv3 --marker-trace`.  Resolved by dlopen at first use and only when FOS_ROCTX=1: no link-time dependency, no cost
// otherwise (one predictable branch per phase).
struct Roctx {
    int state = 0;                         // 0: not tried, 1: on, -1: off
    int (*push)(const char*) = nullptr;
    int (*pop)() = nullptr;
};
static Roctx g_roctx;
static bool roctx_on() {
    if (g_roctx.state == 0) {
        g_roctx.state = -1;
        const char* e = getenv("FOS_ROCTX");
        if (e && atoi(e) != 0) {
            const char* names[] = {"librocprofiler-sdk-roctx.so", "librocprofiler-sdk-roctx.so.1", "libroctx64.so", "libroctx64.so.4"};
            for (const char* nm : names) {
                void* lib = dlopen(nm, RTLD_NOW | RTLD_GLOBAL);
                if (!lib) continue;
                g_roctx.push = (int (*)(const char*))dlsym(lib, "roctxRangePushA");
                g_roctx.pop = (int (*)())dlsym(lib, "roctxRangePop");
                if (g_roctx.push && g_roctx.pop) { g_roctx.state = 1; break; }
            }
        }
    }
    return g_roctx.state == 1;
}
struct RoctxRange {
    bool on;
    explicit RoctxRange(const char* name) : on(roctx_on()) { if (on) g_roctx.push(name); }
    ~RoctxRange() { if (on) g_roctx.pop(); }
};

static int rccl_load() {
    if (g_rccl.lib) return FOS_OK;
    // ONE copy of RCCL per process: a host that already carries one (PyTorch ships its own librccl.so) must be joined, not
    // doubled -- two copies interpose each other's globals and the process dies in their destructors at exit.  So: the copy that
    // is already loaded, if any; otherwise a private one (local scope, own symbols first) that a later-loaded copy cannot touch.
    const char* names[] = {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so.1"};
    void* lib = nullptr;
    for (const char* nm : names) {
        lib = dlopen(nm, RTLD_NOW | RTLD_NOLOAD);
        if (lib) break;
    }
    for (int k = 0; !lib && k < 3; ++k) lib = dlopen(names[k], RTLD_NOW | RTLD_LOCAL | RTLD_DEEPBIND);
    if (!lib) { set_error("cannot dlopen librccl.so: %s", dlerror()); return FOS_ECOMM; }
    g_rccl.GetUniqueId = (decltype(g_rccl.GetUniqueId))dlsym(lib, "ncclGetUniqueId");
    g_rccl.CommInitRank = (decltype(g_rccl.CommInitRank))dlsym(lib, "ncclCommInitRank");
    g_rccl.AllReduce = (decltype(g_rccl.AllReduce))dlsym(lib, "ncclAllReduce");
    g_rccl.CommDestroy = (decltype(g_rccl.CommDestroy))dlsym(lib, "ncclCommDestroy");
    g_rccl.GetErrorString = (decltype(g_rccl.GetErrorString))dlsym(lib, "ncclGetErrorString");
    if (!g_rccl.GetUniqueId || !g_rccl.CommInitRank || !g_rccl.AllReduce || !g_rccl.CommDestroy) {
        set_error("librccl.so lacks a required symbol");
        return FOS_ECOMM;
    }
    g_rccl.lib = lib;
    return FOS_OK;
}

#define FOS_NCCL(expr)                                                                                     \
    do {                                                                                                   \
        ncclResult_t _r = (expr);                                                                          \
        if (_r != ncclSuccess) {                                                                           \
            set_error("%s:%d: %s -> %s", __FILE__, __LINE__, #expr,                                        \
                      g_rccl.GetErrorString ? g_rccl.GetErrorString(_r) : "rccl error");                   \
            return FOS_ECOMM;                                                                              \
        }                                                                                                  \
    } while (0)

}  // namespace fos

using namespace fos;

// ------------------------------------------------------------------------------------------------ the handle
struct fos_solver {
    int device = 0;
    hipStream_t stream = nullptr;
    int64_t m = 0, n = 0, l = 0, nnz = 0;
    int64_t l_global = 0;                      // == l unless sharded (tolerance floor uses the global size)

    // operator
    HostBlkCsr hostS;                          // kept only for re-partitioning (indices freed after upload)
    int64_t win_stats[4] = {0, 0, 0, 0};       // window panels: panels, (panel, window) segments, 64-row slices, stored entries
    DevBlkCsr S{};
    std::vector<void*> owned;                  // every hipMalloc'd pointer
    std::vector<void*> pooled;                 // blocks of the process-wide pool of uncached memory (uncached_acquire): released, never freed
    double* cb = nullptr;
    double nb = 0.0, nc = 0.0;                 // ||b||, ||c|| (global)
    double nb_local = 0.0, nc_local = 0.0;     // this shard's ||b||, ||c||

    // vectors: l double2 each
    d2 *X = nullptr, *T1 = nullptr, *T2 = nullptr;          // iterate, tmp1, tmp2
    d2 *SOL = nullptr, *RHS = nullptr, *R = nullptr, *AP = nullptr;                 // CG: xinit/y, rhs, r, z
    d2 *PB[2] = {nullptr, nullptr};                         // CG direction, ping-pong: p_j lives in PB[j & 1]
    d2 *Y = nullptr, *XOLD = nullptr;                       // FISTA y / xold ; Dykstra p / q
    d2 *W = nullptr;                                        // scratch (Dykstra sums, test entries)
    d2 *SOL2 = nullptr;                                     // HSDEMatrix.cgdata.xinit
    double* plain = nullptr;                                // 2l doubles: ABI staging

    // cones
    uint8_t* ew_op = nullptr;
    ConeDesc* soc = nullptr; int nsoc = 0;
    ConeDesc* expc = nullptr; int nexp = 0;
    ConeDesc* psd = nullptr; int npsd = 0; int psd_kmax = 0, psd_kmin = 0;       // PSD cones of order <= 64 (psd.hip)
    PsdSign* psd_big = nullptr;                // ... of order > 64: projected by matrix products (psd_sign.hip)
    double* psd_scratch = nullptr;
    double* psd_V[2] = {nullptr, nullptr};     // warm-start eigenvector bases (ping-pong), orders <= 64
    int psd_cur = 0, psd_have_prev = 0;
    int* psd_stats = nullptr;                  // Jacobi sweeps of the last projection, per (cone, copy)  (fos_psd_debug)
    int psd_phase_limit = 0;                   // diagnostic: stop the PSD kernel after a phase (wrong results!)
    int cus = 256;                             // compute units of `device`
    int psd_wave = -1; bool psd_narrow = false, psd_wide = false; int psd_wide_threads = 512;      // FOS_PSD_* (read at fos_create)
    mutable bool psd_attr_set = false, psd_attr_set_r = false;
    int32_t* psd_redo = nullptr;               // per (cone, copy): 1 = the refinement kernel left the matrix to the Jacobi kernel
    int psd_refine = -1;                       // FOS_PSD_REFINE
    bool psd_extrapolate = true;               // FOS_PSD_EXTRAPOLATE
    double psd_theta = 0.0;                    // FOS_PSD_THETA
    bool peer_same_device = false;             // a peer rank's mailbox lives on THIS device (several ranks on one GPU: tests)

    // scalars
    DevState* st = nullptr;
    DevState* st_host = nullptr;               // pinned
    // speculation past a CG solve: the kernels that follow it are enqueued (gated on DevState.done) BEFORE the host learns the
    // iteration count, which it then reads from a record the CG kernels write into pinned host memory (wait_cg_mark)
    HostMark* mark = nullptr;                  // pinned + mapped; DevState.hostmark points at it
    double* pre_sums = nullptr;                // LaunchCtx::pre
    bool pre_on = true;
    bool speculate = true;
    double* partials = nullptr;
    double* reduced = nullptr;                 // 16 doubles
    int vec_blocks = 0;
    int cg_blocks = 0;
    uint32_t* def_mask = nullptr;              // bit i: row i of S is finished from partial slots (dual tiles)
    bool fuse_p = false;                       // the p update of CG rides on the next sweep (2 launches per iteration)
    int cg_variant = -1;                       // FOS_CG_*: -1 = the handle's default (sharded: merged reduction, closing in the update)
    // FOS_CG_RESIDENT (resident.hip): the plan (which workgroup holds which tiles), its device copy and the workgroups' record arrays
    ResPlan res_plan;
    ResLaunch res{};
    bool res_ok = false;                       // this handle's operator qualifies (under the current workgroup budget)
    bool res_all = false;                      // ... and so does every rank's (sharded: the vote of global_setup)
    int res_gmax = 0;                          // the budget the plan was made for

    // algorithm (gap.jl:6-21, gapa.jl:9-25, fista.jl:6-18, dykstra.jl:5-17)
    int alg = FOS_ALG_GAP;
    double alpha = 0.8, alpha1 = 1.8, alpha2 = 1.8, beta = 0.0;
    double fista_t = 1.0;

    // direct = true (HSDE.jl:12-15): S1 = IndAffine([Q -I], 0), an exact projection through a one-time dense factorisation
    bool direct = false;
    bool direct_cg = false;                    // direct = true on an operator too large for the dense inverse: the same projection by CG at its tolerance floor from the first call on
    double* Ginv = nullptr;                    // (I + Q Q')^-1, symmetric, column-major, leading dimension Gld (l padded to 64)
    int64_t Gld = 0;
    int direct_iters = 0;                      // Newton-Schulz iterations the set-up took
    double* dvec[2] = {nullptr, nullptr};      // two plain l-vectors
    // direct = true on a BLOCK-SEPARABLE operator: I + A'A is block diagonal with blocks of order <= BLKDIR_MAX (an SDP with few variables per
    // block: C4), so the exact projection costs three KKT sweeps -- no CG, no dense l x l inverse (prox_affine_direct_block)
    bool direct_blk = false;
    int blk_n = 0;                             // diagonal blocks of I + A'A
    int64_t* blk_goff = nullptr;               // [blk_n] start of block b's inverse (s_b x s_b, column-major) in blk_ginv
    int32_t* blk_ioff = nullptr;               // [blk_n + 1] start of block b's column list in blk_idx
    int32_t* blk_idx = nullptr;
    double* blk_ginv = nullptr;
    d2 *blk_phg = nullptr, *blk_qphg = nullptr;   // (D^-1 h, D^-1 M h) and their images under Q, per row
    double* blk_prm = nullptr;                 // the 3 x 3 inverse of the border system (9), delta = 1 + |[c; b]|^2
    double* blk_ctx = nullptr;                 // [blk_n] the blocks' shares of c'x^ (blkdir_solve_kernel)
    bool blk_skip_tail = true;                 // the third apply runs without its deferred-row and tau-row kernels (FOS_BLKDIR_FULL_APPLY=1: with them)
    bool blk_ready = false;                    // blkdir_setup ran to its end (a half-finished set-up must not pass for the block form)

    // S1 = AffinePlusLinear state (affinepluslinear.jl:58-69)
    int64_t prox_i = 1;
    bool firstrun = true;
    int64_t cgiter = 0;
    int hit_max_accum = 0;
    bool firstrun2 = true;                     // HSDEMatrix.cgdata.firstrun
    int last_cg_pred = 0;
    // LineSearchWrapper (wrappers/linesearch.jl): every ls_interval-th iteration is a 31-point step-length search
    bool shift_ready = false;                  // RHS already holds SOL - [0; X.y] (written by the step's last kernel): inside fos_step only
    bool shift_fuse = true;                    // FOS_SHIFT_FUSE=0: every projection runs its own shift pass
    bool in_step = false;
    int64_t ls_interval = 0;
    bool ls_now = false;                       // the iteration in flight is a line-search iteration (between step_once and step_finish)
    // GAPP ("projected GAP", solvers/gapproj.jl): GAP whose every gapp_iproj-th iteration is a 21-point projected search
    int64_t gapp_iproj = 0;
    bool gapp_now = false;
    double gapp_log[23] = {0};                 // 21 test norms, alpha_best, iteration
    double ls_log[34] = {0};                   // last search: ||res||, the 31 test residuals, the chosen alpha, the iteration
    // LongstepWrapper (wrappers/longstep.jl, saveplanes.jl): the last nsave + 1 iterations of every long_interval save the two half-planes of
    // their projections; the iterate is then projected onto the saved planes
    LongPlanes lp;                             // (fos_internal.hpp)
    int cg_same_run = 0;                       // consecutive solves that took exactly last_cg_pred iterations
    const d2* last_checked = nullptr;          // vector the last checkstatus was evaluated on

    // sharding: scalar sums cross GPUs either by an in-stream RCCL all-reduce (comm) or through peer mailboxes (peer_on)
    ncclComm_t comm = nullptr;
    int nranks = 1, rank = 0;
    unsigned long long* peer_mbox = nullptr;   // own mailbox (uncached device memory, exported through HIP IPC)
    std::vector<void*> peer_opened;            // IPC mappings of the peers' mailboxes
    PeerBox peer{};
    bool peer_on = false;
    // host-pinned mailboxes (fos_peer_open_host): the mapped + registered shm segment, its name (rank 0 unlinks it), the local relay
    void* host_seg = nullptr;
    int host_seg_fd = -1;                      // kept open: fos_peer_selftest asks it whether the mapped segment is still linked under its name
    size_t host_seg_bytes = 0;
    std::string host_seg_name;
    unsigned long long* peer_relay = nullptr;
    uint32_t cg_epoch = 0;                     // CG solves so far: the sequence space of the folded exchanges
    // ... or through the caller's own collective on host buffers (fos_comm_init_host: MPI.jl, gloo, ...)
    fos_allreduce_fn host_fn = nullptr;
    void* host_user = nullptr;
    double* host_buf = nullptr;                // pinned, max(2n, 16) doubles
    // row-sharded + peer mailboxes: the n-vector A'y crosses the ranks through peer-mapped memory too (fos_internal.hpp, VecBox)
    double* vec_buf = nullptr;                 // own exchange buffer: [2][nranks][2n] doubles, then [2][nranks] flags (uncached, IPC-exported)
    std::vector<void*> vec_opened;
    VecBox vec{};
    uint32_t vec_seq = 0;                      // exchanges enqueued so far (the same on every rank: all make the same calls)
    int vec_nranks = 0;
    bool sharded() const { return comm != nullptr || peer_on || host_fn != nullptr; }
    // row sharding of a non-block-diagonal A: the first n entries (and tau, kappa) of every vector are replicated,
    // the slots of the rows of A' are summed over the ranks (RCCL all-reduce of 2n doubles) between a sweep and its slot-list sums
    bool row_sharded = false;
    double* slots_rd = nullptr;

    // tuning / measurement
    int cg_chunk = 8;
    int nwg_target = 2048;
    bool prof = false;
    int prof_period = 1;                       // every prof_period-th launch of a class is bracketed by events (1: all)
    int64_t cg_total = 0;                      // CG iterations since fos_create
    int64_t direct_sweeps = 0;                 // sweeps of the block-direct projection since fos_create (profiling ordinal)
    struct ProfRec { hipEvent_t a, b; int cls; int j; };
    std::vector<ProfRec> prof_recs;            // event pairs, reused
    size_t prof_used = 0;
    int64_t prof_seen[FOS_PROF_CLASSES] = {0, 0, 0, 0, 0};
    int64_t prof_steps = 0, prof_steps_sampled = 0;   // outer iterations since fos_profile / of them sampled for FOS_PROF_OTHER
    bool prof_step_on = false;                 // the outer iteration in flight brackets its FOS_PROF_OTHER groups
    static constexpr size_t PROF_CAP = 16384;

    static int sum_slots_over_ranks(void* self);      // defined below (needs the RCCL table)
    // row-sharded WITH dual tiles: the sweep fills local slot lists (S.slots = slots_rd + 2n doubles); between sweep and consumers the
    // lists of the n rows of A' are added up (cmp_rec / cmp_idx -> cmp_local) and THAT n-vector crosses the ranks into slots_rd[0..n);
    // the consumers' records (S.def_rec) name slot j for row j < n and the local lists, shifted by n, for the rows of A
    DefRow* cmp_rec = nullptr;
    int32_t* cmp_idx = nullptr;
    double* cmp_local = nullptr;
    int cmp_lpr = 1;

    LaunchCtx ctx() const {
        LaunchCtx c;
        c.stream = stream; c.S = S; c.cb = cb; c.n = n; c.m = m; c.l = l; c.st = st;
        c.partials = partials; c.reduced = reduced; c.vec_blocks = vec_blocks; c.cg_blocks = cg_blocks;
        c.peer = peer_on ? &peer : nullptr;
        c.def_mask = def_mask;
        c.pre = pre_on ? pre_sums : nullptr;
        c.between = nullptr; c.between_arg = nullptr;
        c.cus = cus; c.psd_wave = psd_wave; c.psd_narrow = psd_narrow; c.psd_wide = psd_wide; c.psd_wide_threads = psd_wide_threads;
        c.psd_attr_set = &psd_attr_set; c.psd_attr_set_r = &psd_attr_set_r; c.psd_refine = psd_refine; c.psd_extrapolate = psd_extrapolate; c.psd_theta = psd_theta;
        c.psd_refine_max_mats = (peer_same_device && nranks > 1) ? std::max(1, cus / nranks) : 0;
        c.count_repl = (!row_sharded || rank == 0) ? 1 : 0;
        c.n_repl = row_sharded ? n : 0;
        if (row_sharded) { c.between = &fos_solver::sum_slots_over_ranks; c.between_arg = const_cast<fos_solver*>(this); }
        return c;
    }
};

// row-sharded operators: slots (this rank's partial sums of A'y, 2n doubles) -> slots_rd (their sum over the ranks), in stream
// the caller's collective: stage `count` doubles through the pinned host buffer (synchronises the stream; a slow path by design)
static int host_allreduce(fos_solver* h, const double* src, double* dst, size_t count) {
    if (hipMemcpyAsync(h->host_buf, src, sizeof(double) * count, hipMemcpyDeviceToHost, h->stream) != hipSuccess) return FOS_EHIP;
    if (hipStreamSynchronize(h->stream) != hipSuccess) return FOS_EHIP;
    if (h->host_fn(h->host_user, h->host_buf, (int64_t)count) != 0) { set_error("the caller's all-reduce callback failed"); return FOS_ECOMM; }
    if (hipMemcpyAsync(dst, h->host_buf, sizeof(double) * count, hipMemcpyHostToDevice, h->stream) != hipSuccess) return FOS_EHIP;
    return hipStreamSynchronize(h->stream) == hipSuccess ? FOS_OK : FOS_EHIP;          // the buffer is reused by the next call
}

int fos_solver::sum_slots_over_ranks(void* self) {
    fos_solver* h = static_cast<fos_solver*>(self);
    const double* src = h->S.slots;            // one slot per row of A' ...
    if (h->cmp_local) {                        // ... or, with dual tiles, the rows' local slot lists added up first
        launch_slots_compact(h->ctx(), (int)h->n, h->cmp_rec, h->cmp_idx, h->cmp_lpr, h->S.slots, h->cmp_local);
        src = h->cmp_local;
    }
    if (h->host_fn) return host_allreduce(h, src, h->slots_rd, (size_t)2 * (size_t)h->n);
    if (h->peer_on && h->vec.buf) {            // peer-mapped memory: push + sum, in stream, no library call
        launch_vec_exchange(h->ctx(), h->vec, ++h->vec_seq, src, h->slots_rd);
        return FOS_OK;
    }
    if (!h->comm) {            // no communicator yet (set-up calls before fos_comm_init, or a single process): the sum is the copy
        return hipMemcpyAsync(h->slots_rd, src, sizeof(double) * 2 * (size_t)h->n, hipMemcpyDeviceToDevice, h->stream) == hipSuccess ? FOS_OK : FOS_EHIP;
    }
    return g_rccl.AllReduce(src, h->slots_rd, (size_t)2 * (size_t)h->n, ncclDouble, ncclSum, h->comm, h->stream) == ncclSuccess ? FOS_OK : FOS_ECOMM;
}

namespace {

template <class T>
int dev_alloc(fos_solver* h, T** p, size_t count) {
    void* q = nullptr;
    size_t bytes = std::max<size_t>(count, 1) * sizeof(T);
    hipError_t e = hipMalloc(&q, bytes);
    if (e != hipSuccess) { set_error("hipMalloc(%zu bytes): %s", bytes, hipGetErrorString(e)); return FOS_ENOMEM; }
    h->owned.push_back(q);
    *p = reinterpret_cast<T*>(q);
    return FOS_OK;
}

template <class T>
void dev_release(fos_solver* h, T** p) {          // frees a dev_alloc'd buffer before the handle's end
    if (!*p) return;
    auto it = std::find(h->owned.begin(), h->owned.end(), (void*)*p);
    if (it != h->owned.end()) h->owned.erase(it);
    (void)hipFree(*p);
    *p = nullptr;
}

template <class T>
int dev_upload(fos_solver* h, T** p, const std::vector<T>& v) {
    FOS_TRY(dev_alloc(h, p, v.size()));
    if (!v.empty()) FOS_HIP(hipMemcpy(*p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice));
    return FOS_OK;
}

int psd_order(int64_t len) {
    int64_t k = (int64_t)std::llround(std::sqrt(0.25 + 2.0 * (double)len) - 0.5);
    if (k * (k + 1) / 2 != len) return -1;
    return (int)k;
}

// all-reduce of `count` doubles in h->reduced (in place, in stream) when sharded
int allreduce(fos_solver* h, int count) {
    if (h->peer_on) return FOS_OK;                             // peer mailboxes: launch_reduce1 already exchanged
    if (h->host_fn) return host_allreduce(h, h->reduced, h->reduced, (size_t)count);
    if (!h->comm) return FOS_OK;
    FOS_NCCL(g_rccl.AllReduce(h->reduced, h->reduced, (size_t)count, ncclDouble, ncclSum, h->comm, h->stream));
    return FOS_OK;
}

// partials[count][nacc] --(sharded: local reduce + all-reduce)--> returns from_reduced flag for the finalize kernel
int finish_reduce(fos_solver* h, const LaunchCtx& c, int count, int nacc, int gate, int* from_reduced, int off = 0) {
    if (!h->sharded()) { *from_reduced = 0; return FOS_OK; }
    launch_reduce1(c, count, nacc, gate, off);
    FOS_TRY(allreduce(h, nacc));
    *from_reduced = 1;
    return FOS_OK;
}

// a kernel launch that the runtime rejected (bad configuration, missing attribute) is only reported by hipGetLastError
int check_launch(const char* where) {
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) { set_error("%s: a kernel launch failed: %s", where, hipGetErrorString(e)); return FOS_EHIP; }
    return FOS_OK;
}

// profiling: a launch group whose ordinal (CG: the iteration's number counted over all solves since fos_create, so that every
// position inside a solve gets sampled; PSD: the call's number) is a multiple of prof_period is bracketed by an event pair
int prof_begin(fos_solver* h, int cls, int j, int64_t ordinal) {
    if (!h->prof) return -1;
    if ((ordinal % h->prof_period) != 0 || h->prof_used >= fos_solver::PROF_CAP) return -1;
    if (h->prof_recs.size() <= h->prof_used) {
        fos_solver::ProfRec r{};
        if (hipEventCreate(&r.a) != hipSuccess || hipEventCreate(&r.b) != hipSuccess) return -1;
        h->prof_recs.push_back(r);
    }
    fos_solver::ProfRec& r = h->prof_recs[h->prof_used];
    r.cls = cls; r.j = j;
    if (hipEventRecord(r.a, h->stream) != hipSuccess) return -1;
    return (int)h->prof_used++;
}
// FOS_PROF_OTHER: a launch group that is neither sweep, CG vector update nor PSD projection, in a sampled outer iteration; post = 1: the group
// belongs to the work enqueued (gated) behind a CG solve -- dropped with it when the gate stayed shut (affine_then)
int prof_begin_other(fos_solver* h, int post) {
    if (!h->prof || !h->prof_step_on) return -1;
    return prof_begin(h, FOS_PROF_OTHER, post, 0);
}
void prof_end(fos_solver* h, int idx) {
    if (idx >= 0) (void)hipEventRecord(h->prof_recs[idx].b, h->stream);
}

int poll_state(fos_solver* h) {
    FOS_TRY(check_launch("poll"));
    FOS_HIP(hipMemcpyAsync(h->st_host, h->st, sizeof(DevState), hipMemcpyDeviceToHost, h->stream));
    FOS_HIP(hipStreamSynchronize(h->stream));
    if (h->st_host->xchg_failed) {
        set_error("rank %d: a peer-mailbox exchange timed out (a peer rank stopped or is not running the same call sequence)", h->rank);
        return FOS_ECOMM;
    }
    if (h->st_host->bar_failed) {
        set_error("a wait between the workgroups of one launch timed out (cg_update_kernel's producer flags: FOS_CG_PRE=0 switches them off; "
                  "the resident CG solve's records: FOS_CG_VARIANT=3 runs the launch-per-iteration form)");
        return FOS_EHIP;
    }
    return FOS_OK;
}

// Has CG solve number `epoch` ended within its first batch of iterations?  No stream synchronisation and nothing in the stream:
// the kernel that ends a solve writes HostMark.seq in pinned host memory, the marked p update of a batch that runs out before
// convergence writes HostMark.batch; the host spins on the two (and looks at the stream now and then, in case neither comes).
int wait_cg_mark(fos_solver* h, uint32_t epoch, int32_t batch_id, bool* ended) {
    FOS_TRY(check_launch("poll"));
    volatile HostMark* m = h->mark;
    // a wall-clock bound (FOS_CG_WAIT_S, default 120 s): a kernel of the solve that never ends must not hang the host either
    static const double wait_s = getenv("FOS_CG_WAIT_S") ? atof(getenv("FOS_CG_WAIT_S")) : 120.0;
    const auto t0 = std::chrono::steady_clock::now();
    for (uint32_t spin = 1;; ++spin) {
        if (__atomic_load_n(&m->seq, __ATOMIC_ACQUIRE) == epoch) { *ended = true; break; }
        if (__atomic_load_n(&m->batch, __ATOMIC_ACQUIRE) == batch_id) { *ended = false; break; }
        __builtin_ia32_pause();
        if ((spin & 0xFFFFu) == 0 && hipStreamQuery(h->stream) == hipSuccess) {          // everything enqueued has run
            if (__atomic_load_n(&m->seq, __ATOMIC_ACQUIRE) == epoch) { *ended = true; break; }
            if (__atomic_load_n(&m->batch, __ATOMIC_ACQUIRE) == batch_id) { *ended = false; break; }
            // no mark: a kernel of the solve gave up (a peer exchange timed out, ...) -- the ordinary state read reports why
            FOS_TRY(poll_state(h));
            *ended = h->st_host->done != 0;
            return FOS_OK;
        }
        if ((spin & 0xFFFFFu) == 0 && std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > wait_s) {
            set_error("CG solve %u: neither its end mark nor its batch mark arrived within %.0f s (a kernel of the solve does not finish)", epoch, wait_s);
            return FOS_EHIP;
        }
    }
    if (*ended) { h->st_host->done = 1; h->st_host->iter = m->iter; h->st_host->hit_max = m->hit_max; h->st_host->rr = m->rr; }
    else h->st_host->done = 0;
    return FOS_OK;
}

// UNCACHED device memory (the mailboxes, the relay and vector-exchange buffers, the resident solve's record arrays) is kept for the life of the
// process and handed from handle to handle, never returned to the runtime: on this image (ROCm 7.2), device memory that had been allocated
// uncached, freed, and handed out again by a later hipMalloc gave a WRONG dense IndAffine set-up several handles later (deterministic;
// DESIGN_LOG.md "the recycled uncached pages") -- so these pages are never recycled as anything else.  A block serves a later request of
// the same device for at most its own size and at least half of it.
namespace {
struct UncachedBlock { int device; void* p; size_t bytes; bool uncached; };
std::mutex g_unc_mu;
std::vector<UncachedBlock> g_unc_free, g_unc_used;
}
// falls back to fine-grained memory (what the mailboxes accept), or with `allow_plain` (the records: cache-bypassing accesses work on any
// memory) to ordinary memory
static int uncached_acquire(int device, size_t bytes, bool allow_plain, void** out, const char* what) {
    UncachedBlock r{device, nullptr, 0, false};
    {
        std::lock_guard<std::mutex> lk(g_unc_mu);
        for (size_t i = 0; i < g_unc_free.size(); ++i) {
            const UncachedBlock& f = g_unc_free[i];
            if (f.device == device && f.bytes >= bytes && f.bytes / 2 <= bytes) { r = f; g_unc_free.erase(g_unc_free.begin() + (long)i); break; }
        }
    }
    if (!r.p) {
        static const bool unc = !(getenv("FOS_RES_UNCACHED") && atoi(getenv("FOS_RES_UNCACHED")) == 0);
        hipError_t e = (unc || !allow_plain) ? hipExtMallocWithFlags(&r.p, bytes, hipDeviceMallocUncached) : hipErrorUnknown;
        r.uncached = (e == hipSuccess);
        if (e != hipSuccess && !allow_plain) { (void)hipGetLastError(); e = hipExtMallocWithFlags(&r.p, bytes, hipDeviceMallocFinegrained); }
        if (e != hipSuccess && allow_plain) { (void)hipGetLastError(); e = hipMalloc(&r.p, bytes); }
        if (e != hipSuccess) { set_error("hipExtMallocWithFlags(%s, %zu bytes): %s", what, bytes, hipGetErrorString(e)); return FOS_ENOMEM; }
        r.bytes = bytes;
    }
    { std::lock_guard<std::mutex> lk(g_unc_mu); g_unc_used.push_back(r); }
    *out = r.p;
    return FOS_OK;
}
static void uncached_release(void* p) {
    if (!p) return;
    std::lock_guard<std::mutex> lk(g_unc_mu);
    for (size_t i = 0; i < g_unc_used.size(); ++i)
        if (g_unc_used[i].p == p) { g_unc_free.push_back(g_unc_used[i]); g_unc_used.erase(g_unc_used.begin() + (long)i); return; }
}

// FOS_CG_RESIDENT: (re)plan the resident solve for at most `gmax` workgroups -- every one of them must be on the device at once, so ranks that
// share ONE device (the tests' stand-in for a multi-GPU box) share its CUs
int resident_setup(fos_solver* h, int gmax) {
    if (const char* e = getenv("FOS_RESIDENT_GMAX")) gmax = std::max(1, std::min(gmax, atoi(e)));      // (tests: several tiles per workgroup on small problems)
    if (h->res_gmax == gmax) return FOS_OK;
    h->res_gmax = gmax;
    h->res_ok = false;
    ResPlan plan;
    if (!build_resident_plan(h->hostS, h->m, h->n, gmax, &plan)) { h->res_plan = plan; h->res_all = false; return FOS_OK; }
    FOS_HIP(hipStreamSynchronize(h->stream));
    if (!h->res.grec) {
        constexpr size_t b1 = sizeof(unsigned long long) * 2 * RES_GMAX * 8, b2 = sizeof(unsigned long long) * 2 * RES_GMAX * 64 * 4;
        void *q1 = nullptr, *q2 = nullptr;
        FOS_TRY(uncached_acquire(h->device, b1, true, &q1, "resident CG records"));
        h->pooled.push_back(q1);
        FOS_TRY(uncached_acquire(h->device, b2, true, &q2, "resident CG column records"));
        h->pooled.push_back(q2);
        FOS_HIP(hipMemset(q1, 0, b1));                           // sequence number 0 is never sent
        FOS_HIP(hipMemset(q2, 0, b2));
        h->res.grec = static_cast<unsigned long long*>(q1); h->res.crec = static_cast<unsigned long long*>(q2);
    }
    ResWG* dwg = const_cast<ResWG*>(h->res.wg);
    dev_release(h, &dwg);
    FOS_TRY(dev_upload(h, &dwg, plan.wg));
    h->res.wg = dwg; h->res.G = plan.G; h->res.nw = plan.nw; h->res.ncomm = plan.ncomm; h->res.rpt = plan.rpt; h->res.tmax = plan.tmax;
    h->res.stream = plan.stream; h->res.nt = plan.nt; h->res.tiles_wg_max = plan.tiles_wg_max;
    static const double res_wait_s = getenv("FOS_RESIDENT_WAIT_S") ? atof(getenv("FOS_RESIDENT_WAIT_S")) : 5.0;
    h->res.timeout_ticks = (int64_t)(res_wait_s * 1e8);
    h->res_plan = plan;
    h->res_ok = true;
    h->res_all = !h->sharded();
    return FOS_OK;
}

// sharded set-up: global problem size and norms (tolerance floor, status normalisation) from the shards'
int global_setup(fos_solver* h) {
    // the resident CG solve runs on a sharded handle only where EVERY rank's shard qualifies (all ranks must run the same exchanges) and
    // the sums cross the ranks through mailboxes (no collective call can sit inside a kernel): a vote through the handle's transport
    FOS_TRY(resident_setup(h, (h->peer_same_device && h->nranks > 1) ? std::max(1, h->cus / h->nranks) : h->cus));
    h->res_all = false;
    if (h->peer_on) {
        LaunchCtx cv = h->ctx();
        double q = h->res_ok ? 1.0 : 0.0;
        FOS_HIP(hipMemcpyAsync(h->partials, &q, sizeof(q), hipMemcpyHostToDevice, h->stream));
        launch_reduce1(cv, 1, 1, 0);
        FOS_TRY(allreduce(h, 1));
        FOS_HIP(hipMemcpyAsync(&q, h->reduced, sizeof(q), hipMemcpyDeviceToHost, h->stream));
        FOS_TRY(poll_state(h));
        h->res_all = q == (double)h->nranks;
    }
    LaunchCtx c = h->ctx();
    const bool cnt = !h->row_sharded || h->rank == 0;     // row-sharded: the n columns and c are replicated, counted by rank 0
    double loc[3] = {(double)(h->m + (cnt ? h->n : 0)), h->nb_local * h->nb_local, cnt ? h->nc_local * h->nc_local : 0.0};
    FOS_HIP(hipMemcpyAsync(h->partials, loc, sizeof(loc), hipMemcpyHostToDevice, h->stream));
    launch_reduce1(c, 1, 3, 0);
    FOS_TRY(allreduce(h, 3));
    FOS_HIP(hipMemcpyAsync(loc, h->reduced, sizeof(loc), hipMemcpyDeviceToHost, h->stream));
    FOS_TRY(poll_state(h));
    h->l_global = (int64_t)std::llround(loc[0]) + 1;
    h->nb = std::sqrt(loc[1]);
    h->nc = std::sqrt(loc[2]);
    return FOS_OK;
}

// out = [I Q'; Q -I] w, all l rows (sweep + tau-row finalize)
// deferred = false: the slot-spread rows of `out` (rows of A' under dual tiles) are not finished; tau_row = false: nor is the tau row -- for callers that
// need neither (the block-direct projection: single-GPU handles only)
int kkt_apply_full(fos_solver* h, const LaunchCtx& c, const d2* w, d2* out, bool deferred = true, bool tau_row = true) {
    launch_kkt2(c, w, out, 0, deferred || tau_row);          // (the tau-row kernel reads the records the deferred-row kernel leaves)
    if (!tau_row) return FOS_OK;
    int fr = 0;
    FOS_TRY(finish_reduce(h, c, c.S.npart, 3, 0, &fr, c.S.part_off));
    launch_kkt_finalize(c, w, out, 0, fr);
    return FOS_OK;
}

// conjugategradient!(x, KKTMatrix(Q), rhs, r, p, Ap; tol, max_iters)      conjugategradients.jl:31-55
// Device resident: the host enqueues iterations AHEAD (every CG kernel is gated on DevState.done) and polls once per batch.
// `apply_on` (default: x): the vector the start residual's operator product is taken of -- prox_affine passes x - (0, in.y)
// together with rhs = in, which is rhs - M x without ever forming rhs (see there).
// `post` (optional): what follows the solve on the stream (relaxation, cone projection, ...), taking a LaunchCtx whose `gate`
// it must hand to every launch.  It is enqueued right behind the FIRST batch of iterations, gated on DevState.done, and the host
// then reads the state at the end of the batch on a second stream: when the batch sufficed (the steady state) the GPU has
// gone straight on while the host was still finding that out, and *post_ran = true.  Otherwise the gated launches were
// no-ops, CG continues as usual and the caller runs `post` itself.
typedef std::function<int(const LaunchCtx&)> PostFn;
int cg_solve(fos_solver* h, d2* x, const d2* rhs, double tol, int maxit, int64_t* iters, const d2* apply_on = nullptr,
             const PostFn* post = nullptr, bool* post_ran = nullptr) {
    LaunchCtx c = h->ctx();
    int next_j = 1;                          // iteration number of the next enqueued launch group (if CG still runs)
    static const bool fold_env = !(getenv("FOS_PEER_FOLD") && atoi(getenv("FOS_PEER_FOLD")) == 0);
    const bool fold = h->peer_on && fold_env;
    const bool rccl = h->sharded() && !fold;  // sums cross the ranks between the kernels (reduce kernel + all-reduce, or unfolded mailboxes)
    // which recurrence: the reference's (three launches, two reduction points per iteration) or the merged-reduction form (two
    // launches, one reduction point); sharded handles take the latter by default and always close in the update kernel
    int32_t variant = FOS_CG_REFERENCE;
    FOS_TRY(fos_get_cg_variant(h, &variant));
    const bool resident = variant == FOS_CG_RESIDENT;
    const bool merged = variant == FOS_CG_MERGED_SWEEP || variant == FOS_CG_MERGED_UPDATE;
    const bool close_in_update = variant == FOS_CG_MERGED_UPDATE;
    const bool fuse_p = variant == FOS_CG_FUSED_P;
    // single GPU, reference recurrence: the solve starts like the merged one -- sweep, ONE launch for r_0 = rhs - M v, p_1 = r_0,
    // the tau row, the slot-spread rows and the r.r records (added by the sweep of iteration 1) -- instead of five launches
    static const bool start_env = !(getenv("FOS_CG_FUSED_START") && atoi(getenv("FOS_CG_FUSED_START")) == 0);
    const bool start_fused = !merged && !h->sharded() && !c.between && start_env;
    h->cg_epoch += 1;                        // the same on every rank: all ranks make the same calls
    const uint32_t seq_base = (uint32_t)(h->cg_epoch * 2048u);          // + 2 j + phase  (j <= 1000)
    auto iter_desc = [&](int j) {
        CgIter it;
        it.j = j; it.r = h->R; it.p_prev = h->PB[(j - 1) & 1]; it.p_cur = h->PB[j & 1];
        it.fuse_p = fuse_p; it.rr_from_reduced = rccl ? 1 : 0; it.fold = fold ? &h->peer : nullptr; it.seq_base = seq_base;
        it.start_fused = start_fused;
        return it;
    };
    auto merged_desc = [&](int j) {
        CgmIter it;
        it.j = j; it.x = x; it.r = h->R; it.p = h->PB[0]; it.s = h->PB[1]; it.w = h->AP;
        it.close_in_update = close_in_update; it.from_reduced = rccl ? 1 : 0; it.fold = fold ? &h->peer : nullptr; it.seq_base = seq_base;
        return it;
    };
    const size_t prof_start = h->prof_used;
    // (sharded: only with the exchanges folded into the CG kernels -- no collective call sits in the stream -- and only for GAP,
    //  whose post-solve kernels contain no reduction; every rank takes the same decisions, so all mispredict together.
    //  Solve number 0 mod 2^21 is the mark's initial value.)
    const bool spec = post && h->speculate && !c.between && !fuse_p && (seq_base >> 11) != 0u &&
                      (!h->sharded() || (fold && h->alg != FOS_ALG_GAPA));
    const int32_t batch_id = (int32_t)(((seq_base >> 11) & 0x7FFFFFu) << 8 | 1u);
    bool mark_last = spec;                   // the first batch ends with a marked launch
    // merged reduction, sharded without folded mailboxes: the sweep's three sums and the update's r.r cross the ranks together,
    // between the sweep of iteration j-1 and the update of iteration j -- ONE all-reduce of four doubles per iteration
    auto merged_reduce = [&](int j) -> int {
        if (c.between) FOS_TRY(c.between(c.between_arg));         // row-sharded: A'y partial sums over the ranks (n-vector)
        if (!rccl) return FOS_OK;
        launch_reduce1(c, c.S.nwg, 3, 1, 0);
        LaunchCtx c2 = c;
        c2.partials = c.partials + 3 * (size_t)PART_CAP + (size_t)((j - 1) & 1) * CGM_RR_STRIDE;
        c2.reduced = c.reduced + 3;
        launch_reduce1(c2, c.cg_blocks, 1, 1);
        return allreduce(h, 4);
    };
    if (resident) {
        // the whole solve is ONE launch (resident.hip): nothing to enqueue ahead, nothing to predict; what follows the solve is enqueued behind it
        // (gated on DevState.done, which the launch sets when it ends) and the host reads the iteration count from the mark the launch leaves
        const int pe = prof_begin(h, FOS_PROF_RESIDENT, 1, h->prof_seen[FOS_PROF_RESIDENT]++);
        launch_cg_resident(c, h->res, x, rhs, apply_on ? apply_on : x, tol, maxit, fold ? &h->peer : nullptr, seq_base);
        prof_end(h, pe);
        if (spec) {
            LaunchCtx cg = c;
            cg.gate = &h->st->done;
            FOS_TRY((*post)(cg));
            bool ended = false;
            FOS_TRY(wait_cg_mark(h, (uint32_t)(seq_base >> 11), batch_id, &ended));
            if (!ended) { set_error("resident CG solve %u ended without its mark", (unsigned)(seq_base >> 11)); return FOS_EHIP; }
            if (post_ran) *post_ran = true;
        } else {
            FOS_TRY(poll_state(h));
            if (!h->st_host->done) { set_error("resident CG solve %u did not finish", (unsigned)(seq_base >> 11)); return FOS_EHIP; }
        }
        *iters = h->st_host->iter;
        h->cg_same_run = (h->st_host->iter == h->last_cg_pred) ? h->cg_same_run + 1 : 0;
        h->last_cg_pred = h->st_host->iter;
        h->cg_total += h->st_host->iter;
        if (h->st_host->hit_max) h->hit_max_accum = 1;
        return FOS_OK;
    }
    if (merged) {
        // start: sweep M v, ONE launch for r = rhs - M v (+ r.r records, tau row, slot-spread rows, the solve's scalars), then
        // w_0 = M r_0, the sweep every iteration's update starts from (it also adds g_0 = r_0.r_0 unless the first update does)
        CgmIter it0 = merged_desc(0);
        const d2* v = apply_on ? apply_on : x;
        const int po = prof_begin_other(h, 0);
        launch_cgm_apply(c, it0, v);                                       // :32  mul!(Ap, A, x)
        if (c.between) FOS_TRY(c.between(c.between_arg));
        if (rccl) { launch_reduce1(c, c.S.nwg, 3, 0, 0); FOS_TRY(allreduce(h, 3)); }
        launch_cgm_start(c, it0, rhs, v, tol, maxit);                      // :33-36
        prof_end(h, po);
        const int pe = prof_begin(h, FOS_PROF_KKT, 0, h->cg_total);
        launch_cgm_sweep(c, it0, close_in_update ? -1 : 0);
        prof_end(h, pe);
    } else if (start_fused) {
        CgmIter it0 = merged_desc(0);
        const d2* v = apply_on ? apply_on : x;
        const int po = prof_begin_other(h, 0);
        launch_cgm_apply(c, it0, v);                                       // :32  mul!(Ap, A, x)
        launch_cgm_start(c, it0, rhs, v, tol, maxit, h->PB[1]);            // :33-36   (p_1 in buffer 1)
        prof_end(h, po);
    } else {
        int fr = 0;
        const int po = prof_begin_other(h, 0);
        FOS_TRY(kkt_apply_full(h, c, apply_on ? apply_on : x, h->AP));    // :32  mul!(Ap, A, x)
        launch_cg_init(c, rhs, h->AP, h->R, h->PB[1]);                     // :33-34   (p_1 in buffer 1)
        FOS_TRY(finish_reduce(h, c, c.vec_blocks, 1, 0, &fr));
        launch_cg_init_finalize(c, h->R, tol, maxit, fr);                  // :35-36
        prof_end(h, po);
    }
    auto enqueue = [&](int count) -> int {
        if (merged) {
            for (int q = 0; q < count && next_j <= maxit; ++q, ++next_j) {
                CgmIter it = merged_desc(next_j);
                const bool last = q == count - 1 || next_j == maxit;
                if (mark_last && last) it.batch_mark = batch_id;
                FOS_TRY(merged_reduce(next_j));
                // launch 1: scalars, p = r + beta p, s = w + beta s, x += alpha p, r -= alpha s, r.r partials             :39-41,:46-50
                int pe = prof_begin(h, FOS_PROF_CGVEC, next_j, (h->cg_total + next_j - 1) % 4 == 1 ? (h->cg_total + next_j - 1) / 4 : 1);
                launch_cgm_update(c, it, false);
                prof_end(h, pe);
                // launch 2: [close iteration j: r.r, stop test] sweep w = M r + partial sums                               :38,:42
                // (closing in the sweep: the sweep behind the LAST update returns in its prologue -- recorded as iteration j+1, which
                //  the profile drops)
                pe = prof_begin(h, FOS_PROF_KKT, next_j + (close_in_update ? 0 : 1), h->cg_total + next_j);
                launch_cgm_sweep(c, it, close_in_update ? -1 : next_j);
                prof_end(h, pe);
            }
            if (close_in_update) {           // the last enqueued iteration is closed by a one-workgroup launch of the update kernel
                CgmIter it = merged_desc(next_j);
                if (mark_last) it.batch_mark = batch_id;
                FOS_TRY(merged_reduce(next_j));
                launch_cgm_update(c, it, true);
            }
            return FOS_OK;
        }
        for (int q = 0; q < count && next_j <= maxit; ++q, ++next_j) {
            CgIter it = iter_desc(next_j);
            if (mark_last && (q == count - 1 || next_j == maxit)) it.batch_mark = batch_id;
            // launch 1: [close iteration j-1: r.r, stop test, beta; p_j on the fly] KKT sweep Ap = M p_j + partial sums   :38,:42-50
            int pe = prof_begin(h, FOS_PROF_KKT, next_j, h->cg_total + next_j - 1);
            launch_kkt2_cg(c, it, h->AP);
            prof_end(h, pe);
            int f1 = 0;
            if (c.between) FOS_TRY(c.between(c.between_arg));         // row-sharded: A'y partial sums over the ranks (n-vector)
            if (rccl) { launch_reduce1(c, c.S.nwg, 3, 1, 0); FOS_TRY(allreduce(h, 3)); f1 = 1; }
            // launch 2: alpha, tau rows, slot-spread rows, x += alpha p, r -= alpha Ap, r.r partials                       :39-41
            pe = prof_begin(h, FOS_PROF_CGVEC, next_j, (h->cg_total + next_j - 1) % 4 == 1 ? (h->cg_total + next_j - 1) / 4 : 1);   // a quarter of the sweeps' rate, other iterations
            launch_cg_update(c, it, x, h->R, h->AP, f1);
            if (rccl) {
                LaunchCtx c2 = c;
                c2.partials = c.partials + 3 * (size_t)PART_CAP;
                launch_reduce1(c2, c.cg_blocks, 1, 1);
                FOS_TRY(allreduce(h, 1));
            }
            // gather-bound operators: closing the iteration and p_{j+1} = r + beta p_j stay a launch of their own           :42-51
            if (!fuse_p) launch_cg_pupdate(c, it, x, h->PB[(next_j + 1) & 1]);
            prof_end(h, pe);
        }
        // the last update of the batch is closed by a one-workgroup launch (a following batch's sweep repeats it, same result)
        if (fuse_p) launch_cg_stop_check(c, iter_desc(next_j));
        return FOS_OK;
    };
    static const int pred_slack = getenv("FOS_CG_SLACK") ? atoi(getenv("FOS_CG_SLACK")) : 1;   // iterations enqueued beyond the last solve's count (each costs a few gated no-op launches when not needed; too few costs a poll)
    // (the slack iteration is dropped once three solves in a row took the same number of iterations: steady state of a
    // well-conditioned problem -- C4: 17, 17, 17, ... -- where it would be three gated no-op launches per solve)
    const int slack = (pred_slack == 1 && h->cg_same_run >= 3) ? 0 : pred_slack;
    int first = h->last_cg_pred > 0 ? h->last_cg_pred + slack : h->cg_chunk;
    first = std::max(1, std::min(first, maxit));
    FOS_TRY(enqueue(first));
    mark_last = false;
    if (spec) {
        LaunchCtx cg = c;
        cg.gate = &h->st->done;
        FOS_TRY((*post)(cg));
        bool ended = false;
        FOS_TRY(wait_cg_mark(h, (uint32_t)(seq_base >> 11), batch_id, &ended));
        if (post_ran) *post_ran = ended;
    } else {
        FOS_TRY(poll_state(h));
    }
    while (!h->st_host->done) {
        FOS_TRY(enqueue(h->cg_chunk));
        FOS_TRY(poll_state(h));
    }
    *iters = h->st_host->iter;
    // profiling: launches enqueued past convergence were gated no-ops -- drop their event pairs
    for (size_t k = prof_start; k < h->prof_used; ++k)
        if (h->prof_recs[k].cls != FOS_PROF_OTHER && h->prof_recs[k].j > h->st_host->iter) h->prof_recs[k].cls = -1;
    h->cg_same_run = (h->st_host->iter == h->last_cg_pred) ? h->cg_same_run + 1 : 0;
    h->last_cg_pred = h->st_host->iter;
    h->cg_total += h->st_host->iter;
    if (h->st_host->hit_max) h->hit_max_accum = 1;
    return FOS_OK;
}

// prox!(y, S1::IndAffine([Q -I], 0), x) on a block-separable operator: the exact projection in THREE KKT sweeps.
//   The projection of (u, v) onto {v = Q u} is (u^, Q u^) with (I - Q^2) u^ = u - Q v =: g   (Q' = -Q).  With h = [c; b], M = [0 A'; -A 0]:
//       I - Q^2 = [ P + h h', -M h ; -(M h)', delta ],   P = blkdiag(I + A'A, I + AA'),  delta = 1 + h'h
//               = D + W C W',   D = blkdiag(P, delta),  W = [ (h; 0), (M h; 0), e_tau ],  C = [1 0 0; 0 0 -1; 0 -1 0]
//   so by Woodbury  u^ = D^-1 g - [ph, pg, e_tau / delta] kappa,  kappa = (C^-1 + W' D^-1 W)^-1 [ph.g, pg.g, g_tau / delta],  ph = D^-1 (h; 0), pg = D^-1 (M h; 0)
//   (ph, pg, Q ph, Q pg and the 3 x 3 inverse are formed once).  D^-1 g: the x part is a product with the inverted diagonal blocks of I + A'A,
//   the y part (I + AA')^-1 g2 = g2 - A q, q = (I + A'A)^-1 A' g2.  Q D^-1 g needs A'(g2 - A q) = q (no sweep) and A x^.  Sweeps, all through the
//   dual-right-hand-side KKT apply out = (w1 - Q w2, Q w1 - w2):   (1) w = (u, v) -> g;   (2) w = (0, (0, g2, 0)) -> A' g2;   (3) w = ((q,0,0), (x^,0,0)) -> A q, A x^.
//   `from_T`: h->R already holds g in its first part (set-up: ph, pg); zero_kappa: no border correction (set-up).  Scratch: the CG vectors.
int prox_affine_direct_block(fos_solver* h, const d2* x, d2* out, bool from_T = false, bool zero_kappa = false) {
    RoctxRange range("fos:prox_affine_direct_block (3 KKT sweeps + block-diagonal solve)");
    LaunchCtx c = h->ctx();
    const LaunchCtx& cb = c;
    d2 *T = h->R, *W2 = h->PB[0], *Rr = h->AP, *W3 = h->PB[1], *V = h->RHS;
    double* p1 = h->partials + (size_t)4 * PART_CAP;
    double* p2 = h->partials + (size_t)5 * PART_CAP;
    // Cone-sharded handles: D is block diagonal, so everything above is local to a rank -- except the scalars: the tau row of the first apply (summed over
    // the ranks by kkt_apply_full), the two dots behind kappa and the tau row of the result (c'x^ + b'y^): three exchanges per projection instead of one per
    // CG iteration.  The sums go through the handle's transport (reduce kernel with the mailbox exchange inside, or reduce kernel + all-reduce).
    const bool sh = h->sharded();
    int fr = 0;
    // (profiling: the three sweeps -- each with the deferred-row kernel and the tau-row finalize of a stand-alone apply -- are the KKT class)
    int pe = -1;
    if (!from_T) { pe = prof_begin(h, FOS_PROF_KKT, 1, h->direct_sweeps++); FOS_TRY(kkt_apply_full(h, c, x, T)); prof_end(h, pe); }      // T.x = u - Q v = g
    int po = prof_begin_other(h, 0);
    launch_blkdir_prep(cb, T, h->blk_phg, W2, W3, p1);
    if (sh) { LaunchCtx c2 = c; c2.partials = p1; launch_reduce1(c2, c.vec_blocks, 3, 0); FOS_TRY(allreduce(h, 3)); fr = 1; }
    prof_end(h, po);
    pe = prof_begin(h, FOS_PROF_KKT, 1, h->direct_sweeps++);
    FOS_TRY(kkt_apply_full(h, c, W2, Rr, true, false));                // Rr.x = -Q (0, g2, 0): its x part is -A' g2 (rows of A': the deferred rows; the tau row is not needed)
    prof_end(h, pe);
    po = prof_begin_other(h, 0);
    launch_blkdir_solve(cb, h->blk_n, h->blk_goff, h->blk_ioff, h->blk_idx, h->blk_ginv, Rr, T, W3, h->blk_ctx);
    prof_end(h, po);
    pe = prof_begin(h, FOS_PROF_KKT, 1, h->direct_sweeps++);
    // V.y = -A q, V.x = A x^ on the rows of A -- rows the sweep finishes itself; with dual tiles neither the rows of A' nor the tau row (c'x^: the solve kernel's records) are needed.
    // The tau row of THIS apply is needed by nobody, and on a sharded handle its reduce would overwrite the prep sums in c.reduced that blkdir_combine reads
    // (from_reduced) and add an exchange that only some ranks make: never run there.  The deferred-row kernel runs where rows of A are slot-spread.
    FOS_TRY(kkt_apply_full(h, c, W3, V, !h->blk_skip_tail, !h->blk_skip_tail && !sh));
    prof_end(h, pe);
    po = prof_begin_other(h, 0);
    launch_blkdir_combine(cb, T, W3, V, h->blk_phg, h->blk_qphg, h->blk_prm, zero_kappa ? 1 : 0, out, p1, p2, fr);
    if (sh && h->peer_on && cb.peer) {
        // mailbox transports: the tau kernel sums the rank's record, exchanges it and forms the tau row itself (one launch instead of three)
        launch_blkdir_tau(cb, T, h->blk_qphg, h->blk_prm, zero_kappa ? 1 : 0, out, p1, p2, h->blk_ctx, h->blk_n, 2);
    } else {
        if (sh) {
            double* p3 = p1;                                           // (the prep records are spent)
            launch_blkdir_tausum(cb, p2, h->blk_ctx, h->blk_n, p3);
            LaunchCtx c3 = c; c3.partials = p3;
            launch_reduce1(c3, 1, 1, 0);
            FOS_TRY(allreduce(h, 1));
        }
        launch_blkdir_tau(cb, T, h->blk_qphg, h->blk_prm, zero_kappa ? 1 : 0, out, p1, p2, h->blk_ctx, h->blk_n, fr);
    }
    prof_end(h, po);
    h->cgiter = 0;
    return check_launch("block-direct affine projection");
}

// prox!(y, S1::IndAffine([Q -I], 0), x) with the result left in h->SOL        HSDE.jl:12-15 (direct = true)
int prox_affine_direct(fos_solver* h, const d2* x) {
    if (h->direct_blk) return prox_affine_direct_block(h, x, h->SOL);
    RoctxRange range("fos:prox_affine_direct (2 Q sweeps + dense symmetric matvec)");
    LaunchCtx c = h->ctx();
    int fr = 0;
    // (scratch: the CG vectors R, AP -- the input may be X, Y or W)
    launch_q1(c, Q_VFROMU, x, 0, 1.0, h->R);                           // R = (u, Q u)
    FOS_TRY(finish_reduce(h, c, c.S.npart, 1, 0, &fr, c.S.part_off));
    launch_q1_finalize(c, Q_VFROMU, x, 0, 1.0, h->R, fr);
    launch_direct_rhs(c, h->R, x, h->dvec[0]);                         // t = Q u - v
    launch_dense_symv(c, h->Gld, h->Ginv, h->dvec[0], h->dvec[1]);     // w = (I + Q Q')^-1 t
    launch_set_comp(c, h->AP, h->dvec[1], 0);                          // (w, 0)
    launch_q1(c, Q_VFROMU, h->AP, 0, 1.0, h->R);                       // R = (w, Q w)
    FOS_TRY(finish_reduce(h, c, c.S.npart, 1, 0, &fr, c.S.part_off));
    launch_q1_finalize(c, Q_VFROMU, h->AP, 0, 1.0, h->R, fr);
    launch_direct_finish(c, x, h->R, h->SOL);                          // (u + Q w, v + w)
    h->cgiter = 0;
    return check_launch("direct affine projection");
}

// prox!(y, S1::AffinePlusLinear, x) with the result left in h->SOL       affinepluslinear.jl:83-126
int prox_affine(fos_solver* h, const d2* x, const PostFn* post = nullptr, bool* post_ran = nullptr) {
    if (h->direct) return prox_affine_direct(h, x);
    RoctxRange range("fos:prox_affine (rhs build + warm-started CG over the KKT operator)");
    LaunchCtx c = h->ctx();
    // The right-hand side [x1 - Q x2; 0] (:94-95) costs a sweep over A, and CG starts with another one for rhs - M y (:32-33).
    // Since M [0; x2] = [-Q x2; -x2], rhs = x + M [0; x2] and the start residual is  x - M (y - [0; x2]) : ONE sweep, applied
    // to the warm start shifted by the input's second part, and rhs is never formed (FOS_FUSED_RHS=0: the two sweeps).
    static const bool fused_rhs = !(getenv("FOS_FUSED_RHS") && atoi(getenv("FOS_FUSED_RHS")) == 0);
    if (!fused_rhs) {
        int fr = 0;
        launch_q1(c, Q_RHS, x, 1, 1.0, h->RHS);                        // :94-95
        FOS_TRY(finish_reduce(h, c, c.S.npart, 1, 0, &fr, c.S.part_off));
        launch_q1_finalize(c, Q_RHS, x, 1, 1.0, h->RHS, fr);
    }
    if (h->firstrun) {                                                  // :101-104
        FOS_HIP(hipMemcpyAsync(h->SOL, x, sizeof(d2) * h->l, hipMemcpyDeviceToDevice, h->stream));
        h->firstrun = false;
    }
    if (fused_rhs && !(h->shift_ready && x == h->X)) {
        const int po = prof_begin_other(h, 0);
        launch_shift_part2(c, h->RHS, h->SOL, x);                       // RHS buffer := y - [0; x2]
        prof_end(h, po);
    }
    h->shift_ready = false;                                             // (SOL changes below)
    // :108-112   tol = max(0.2^sqrt(i), size(A,2)*eps())
    const double eps = 2.220446049250313e-16;
    double tol = std::max(h->direct_cg ? 0.0 : std::pow(0.2, std::sqrt((double)h->prox_i)), (double)h->l_global * eps);
    h->prox_i += 1;                                                     // :114
    int64_t it = 0;
    const int64_t cap = h->direct_cg ? 10000 : 1000;                   // (:115: 1000; the exact projection is given the default cap of conjugategradients.jl:31)
    if (fused_rhs) FOS_TRY(cg_solve(h, h->SOL, x, tol, cap, &it, h->RHS, post, post_ran));
    else FOS_TRY(cg_solve(h, h->SOL, h->RHS, tol, cap, &it, nullptr, post, post_ran));          // :115-117 ; y aliases xinit (:106,:122)
    h->cgiter = it;                                                     // :121
    return FOS_OK;                                                      // :124 y2 .*= beta with beta = 1
}

// prox!(y, S2::DualConeProduct, x)                                        cones.jl:122-142
// ew_done: the elementwise cones were projected by the kernel that wrote `in` (launch_relax_ew)
int prox_cones(fos_solver* h, d2* out, const d2* in, const int32_t* gate = nullptr, bool ew_done = false) {
    RoctxRange range("fos:prox_cones (elementwise + SOC + Exp + batched PSD)");
    LaunchCtx c = h->ctx();
    c.gate = gate;
    const int po = (!ew_done || h->nsoc > 0 || h->nexp > 0) ? prof_begin_other(h, gate ? 1 : 0) : -1;
    if (!ew_done) launch_cones_elementwise(c, out, in, h->ew_op);
    launch_cones_soc(c, out, in, h->soc, h->nsoc);
    launch_cones_exp(c, out, in, h->expc, h->nexp);
    prof_end(h, po);
    const int pe = (h->npsd > 0 || h->psd_big) ? prof_begin(h, FOS_PROF_PSD, 0, h->prof_seen[FOS_PROF_PSD]++) : -1;
    FOS_TRY(launch_cones_psd(c, out, in, h->psd, h->npsd, h->psd_kmin, h->psd_kmax, h->psd_scratch,
                             h->psd_V[h->psd_cur], h->psd_V[1 - h->psd_cur], h->psd_have_prev, h->psd_stats, h->psd_phase_limit, h->psd_redo));
    FOS_TRY(launch_cones_psd_sign(c, h->psd_big, out, in));                  // cones of order > 64
    prof_end(h, pe);
    // (a diagnostic launch truncated before the basis store leaves the previous basis current)
    if (h->npsd > 0 && h->psd_V[0] && (h->psd_phase_limit == 0 || (h->psd_phase_limit >= 4 && h->psd_phase_limit < 11))) { h->psd_cur = 1 - h->psd_cur; h->psd_have_prev = std::min(h->psd_have_prev + 1, 2); }
    return check_launch("cone projection");
}

// checkstatus(status, z, override=true) values + decision               HSDEStatus.jl:27-63
int status_check(fos_solver* h, const d2* z, double eps, fos_check_result* res) {
    RoctxRange range("fos:checkstatus");
    LaunchCtx c = h->ctx();
    int fr = 0;
    launch_q1(c, Q_STATUS, z, 0, 1.0, nullptr);
    FOS_TRY(finish_reduce(h, c, c.S.npart, 6, 0, &fr, c.S.part_off));
    launch_status_finalize(c, z, fr);
    FOS_TRY(poll_state(h));
    const double* s = h->st_host->stat;
    const double tau = s[ST_TAU], kappa = s[ST_KAPPA];
    const double nb = h->nb, nc = h->nc;
    const double ctx = s[ST_CTX], bty = s[ST_BTY];
    res->p = std::sqrt(s[ST_RP2]) / std::fabs(1 + nb);                                   // :34
    res->d = std::sqrt(s[ST_RD2]) / std::fabs(1 + nc);                                   // :35
    res->ctx = ctx;                                                                      // :36
    res->bty = bty;                                                                      // :37
    res->g = std::fabs(ctx / tau + bty / tau) / (1 + std::fabs(ctx / tau) + std::fabs(bty / tau));   // :38
    res->kappa = kappa;
    res->tau = tau;
    res->norm_axs = std::sqrt(s[ST_AXS2]);
    res->norm_aty = std::sqrt(s[ST_ATY2]);
    res->norm_b = nb;
    res->norm_c = nc;
    res->cgiter = h->cgiter;
    res->cg_maxiter_hit = h->hit_max_accum;
    h->hit_max_accum = 0;
    int status = FOS_STATUS_CONTINUE;                                                    // :53-63
    if (res->p <= eps * (1 + nb) && res->d <= eps * (1 + nc) &&
        res->g <= eps * (1 + std::fabs(ctx / tau) + std::fabs(bty / tau))) {
        status = FOS_STATUS_OPTIMAL;
    } else if (res->norm_axs <= eps * (-ctx / nc)) {
        status = FOS_STATUS_UNBOUNDED;
    } else if (res->norm_aty <= eps * (-bty / nb)) {
        status = FOS_STATUS_INFEASIBLE;
    }
    res->status = status;
    return FOS_OK;
}

// ---- LineSearchWrapper(GAP / GAPA)                                   wrappers/linesearch.jl:36-75
// S1!(y, x) = a1 prox_S1(x) + (1 - a1) x ,  S2!(y, x) = a2 prox_S2(x) + (1 - a2) x   (gap.jl:42-59; GAPA: a1 = a2 = alpha12, gapa.jl:61-79)
// scratch: Y = tmp1 (the iterate the search starts from), XOLD = res, W = tmp3 -- unused by these two algorithms
void ls_relax(fos_solver* h, const LaunchCtx& c, d2* out, const d2* y, const d2* x, int which) {
    if (h->alg == FOS_ALG_GAPA) { launch_relax_a12(c, out, y, x); return; }
    const double a = which == 1 ? h->alpha1 : h->alpha2;
    launch_axpby(c, out, a, y, 1 - a, x);
}
int ls_normdiff(fos_solver* h, const LaunchCtx& c, const d2* x, const d2* y, double* out) {
    launch_normdiff(c, x, y);
    std::vector<double> part((size_t)c.vec_blocks);
    FOS_HIP(hipMemcpyAsync(part.data(), c.partials, sizeof(double) * part.size(), hipMemcpyDeviceToHost, h->stream));
    FOS_HIP(hipStreamSynchronize(h->stream));
    double s = 0.0;
    for (double v : part) s += v;
    *out = std::sqrt(s);
    return FOS_OK;
}
int ls_begin(fos_solver* h, const d2** check_on) {
    LaunchCtx c = h->ctx();
    FOS_HIP(hipMemcpyAsync(h->Y, h->X, sizeof(d2) * h->l, hipMemcpyDeviceToDevice, h->stream));     // tmp1 .= x            :41
    FOS_TRY(prox_affine(h, h->X));                                                                  // S1!(tmp2, x)         :45
    ls_relax(h, c, h->T1, h->SOL, h->X, 1);
    FOS_TRY(prox_cones(h, h->T2, h->T1));                                                           // S2!(x, tmp2): prox + checkstatus :46
    *check_on = h->T2;
    return FOS_OK;
}
int ls_finish(fos_solver* h, int64_t i) {
    LaunchCtx c = h->ctx();
    ls_relax(h, c, h->X, h->T2, h->T1, 2);                                                          //   ... and its relaxation
    launch_axpby(c, h->XOLD, 1.0, h->X, -1.0, h->Y);                                                // res .= x .- tmp1     :49
    FOS_TRY(ls_normdiff(h, c, h->X, h->Y, &h->ls_log[0]));                                          // normres = norm(res)  :50
    double best = INFINITY, abest = 1.0, a = 0.1;                                                   // :53-55
    for (int k = 0; k <= 30; ++k) {                                                                 // :56
        a = a * 1.8;                                                                                // :57
        launch_axpby(c, h->X, 1.0, h->Y, a, h->XOLD);                                               // x .= tmp1 .+ a.*res  :58
        FOS_TRY(prox_affine(h, h->X));                                                              // S1!(tmp2, x, nostatus) :60
        ls_relax(h, c, h->T1, h->SOL, h->X, 1);
        FOS_TRY(prox_cones(h, h->T2, h->T1));                                                       // S2!(tmp3, tmp2, nostatus) :61
        ls_relax(h, c, h->W, h->T2, h->T1, 2);
        double tr = 0.0;
        FOS_TRY(ls_normdiff(h, c, h->X, h->W, &tr));                                                // testres = normdiff(x, tmp3) :62
        h->ls_log[1 + k] = tr;
        if (tr < best) { best = tr; abest = a; }                                                    // :64-67
    }
    launch_axpby(c, h->X, 1.0, h->Y, abest, h->XOLD);                                               // x .= tmp1 .+ abest.*res :70
    h->ls_log[32] = abest;
    h->ls_log[33] = (double)i;
    return FOS_OK;
}

// ---- GAPP, a search iteration                                       solvers/gapproj.jl:29-62
// scratch: Y = tmp1 = P_S1(x), XOLD = res, W = tmp3 / the new tmp1, T1 = tmp4 -- Y, XOLD, W are unused by GAP
int gapp_begin(fos_solver* h, int64_t i, const d2** check_on) {
    LaunchCtx c = h->ctx();
    FOS_TRY(prox_affine(h, h->X));                                                                  // prox!(tmp1, S1, x)        :33
    FOS_HIP(hipMemcpyAsync(h->Y, h->SOL, sizeof(d2) * h->l, hipMemcpyDeviceToDevice, h->stream));
    FOS_TRY(prox_cones(h, h->T2, h->Y));                                                            // prox!(tmp2, S2, tmp1)     :39
    FOS_TRY(prox_affine(h, h->T2));                                                                 // prox!(res, S1, tmp2)      :40
    launch_axpby(c, h->XOLD, 1.0, h->SOL, -1.0, h->Y);                                              // res .= res - tmp1         :41
    double normbest = INFINITY, abest = -1.0;                                                       // :44-45
    for (int k = 0; k <= 20; ++k) {                                                                 // :46
        const double at = std::ldexp(1.0, k);                                                       // 2.0^k
        launch_axpby(c, h->W, 1.0, h->Y, at, h->XOLD);                                              // tmp3 .= tmp1 .+ at.*res
        FOS_TRY(prox_cones(h, h->T1, h->W));                                                        // prox!(tmp4, S2, tmp3)
        double nt = 0.0;
        FOS_TRY(ls_normdiff(h, c, h->T1, h->W, &nt));                                               // norm(tmp4 - tmp3)
        h->gapp_log[k] = nt;
        if (nt < normbest) { abest = at; normbest = nt; }
    }
    h->gapp_log[21] = abest; h->gapp_log[22] = (double)i;
    launch_axpby(c, h->W, 1.0, h->Y, abest, h->XOLD);                                               // tmp1 .= tmp1 .+ abest.*res :58
    FOS_TRY(prox_cones(h, h->T2, h->W));                                                            // prox!(tmp2, S2, tmp1)     :59
    *check_on = h->T2;                                                                              // checkstatus(status, tmp2) :60
    return FOS_OK;
}
int gapp_finish(fos_solver* h) {
    LaunchCtx c = h->ctx();
    launch_axpby(c, h->X, h->alpha2, h->T2, 1 - h->alpha2, h->W);                                   // x .= a2 tmp2 + (1-a2) tmp1 :61-62
    return FOS_OK;
}

// ---- LongstepWrapper                                                 wrappers/longstep.jl:41-101, saveplanes.jl:13-35
int step_finish_launch(fos_solver* h, const LaunchCtx& c);
// addprojeq (which = 0) / addprojineq (which = 1): row i = (savepos - 1) (neq + nineq) + eqi (+ uneqi) + 1 with neq = nineq = 1     longstep.jl:69,88
void long_save(fos_solver* h, const LaunchCtx& c, int which, const d2* y, const d2* x) { fos::long_save_plane(c, h->lp, which, y, x); }
// a saving iteration: the step of the wrapped algorithm, unfused, with the two planes taken where the reference's step calls addprojeq /
// addprojineq (gap.jl:47,57  gapa.jl:66,76  fista.jl:36,42  dykstra.jl:30,34)
int long_begin(fos_solver* h, int64_t i, const d2** check_on) {
    LaunchCtx c = h->ctx();
    switch (h->alg) {
        case FOS_ALG_GAP:
        case FOS_ALG_GAPA:
            FOS_TRY(prox_affine(h, h->X));                                       // prox!(y, S1, x)
            long_save(h, c, 0, h->SOL, h->X);                                    // addprojeq(longstep, y, x)
            ls_relax(h, c, h->T1, h->SOL, h->X, 1);                              // y .= a1 y + (1 - a1) x
            FOS_TRY(prox_cones(h, h->T2, h->T1));                                // prox!(y, S2, x); checkstatus(status, y)
            *check_on = h->T2;
            return FOS_OK;
        case FOS_ALG_FISTA:
            if (i == 1) FOS_HIP(hipMemcpyAsync(h->Y, h->X, sizeof(d2) * h->l, hipMemcpyDeviceToDevice, h->stream));
            FOS_TRY(prox_affine(h, h->Y));                                       // prox!(tmp1, S1, y)
            long_save(h, c, 0, h->SOL, h->Y);                                    // addprojeq(longstep, tmp1, y)
            launch_axpby(c, h->T1, h->alpha, h->SOL, 1 - h->alpha, h->Y);
            launch_copy(c, h->XOLD, h->X);
            FOS_TRY(prox_cones(h, h->X, h->T1));
            *check_on = h->X;
            return FOS_OK;
        case FOS_ALG_DYKSTRA:
            launch_add(c, h->W, h->X, h->Y);                                     // x .+ p
            FOS_TRY(prox_affine(h, h->W));                                       // prox!(y, S1, x .+ p)
            long_save(h, c, 0, h->SOL, h->W);                                    // addprojeq(longstep, y, x .+ p)
            launch_dykstra_corr(c, h->Y, h->X, h->SOL);                          // p .= x .+ p .- y
            launch_add(c, h->W, h->SOL, h->XOLD);                                // y .+ q
            FOS_TRY(prox_cones(h, h->X, h->W));                                  // prox!(x, S2, y .+ q)
            *check_on = h->X;
            return FOS_OK;
    }
    return FOS_EINVAL;
}
// The projection of x onto {v : A v = b, C v >= d}, A = the first nsave + 1 saved rows, C = the others (saveplanes.jl:17-28 -- the rows are
// saved equality, inequality, equality, ... and split here into halves: the reference's behaviour, kept).  The reference hands the n-variable
// problem to QPDAS in BigFloat; the solution is v = x + P' nu with nu = (lambda, mu >= 0) the minimiser of the SMALL dual
//     1/2 nu' G nu - nu' (beta - P x),  G = P P'  (K = 2 (nsave + 1) <= 32 unknowns),
// solved on the host: the inequality multipliers by enumeration of their support (<= 2^16 candidate supports; each a K x K symmetric system
// by eigen-decomposition, singular ones -- parallel planes -- through the pseudo-inverse); a support is accepted when its multipliers are
// non-negative and the inequalities outside it hold.  Only G, P x and beta leave the device; x += P' nu runs there.
// 113-bit arithmetic for the small dual QP of the LongstepWrapper (the reference solves it in BigFloat, saveplanes.jl:24)
typedef __float128 qreal;
static inline qreal qabs(qreal x) { return x < 0 ? -x : x; }
static inline qreal qsqrt(qreal x) {
    if (!(x > 0)) return 0;
    qreal y = (qreal)sqrtl((long double)x);
    y = (y + x / y) / 2; y = (y + x / y) / 2;
    return y;
}
static void sym_eig_jacobi(int n, std::vector<qreal>& A, std::vector<qreal>& V) {      // A (n x n, row-major) -> eigenvalues on its diagonal, eigenvectors in the columns of V
    V.assign((size_t)n * n, (qreal)0);
    for (int i = 0; i < n; ++i) V[(size_t)i * n + i] = 1;
    for (int sweep = 0; sweep < 80; ++sweep) {
        qreal off = 0, diag = 0;
        for (int i = 0; i < n; ++i) { diag += A[(size_t)i * n + i] * A[(size_t)i * n + i]; for (int j = i + 1; j < n; ++j) off += A[(size_t)i * n + j] * A[(size_t)i * n + j]; }
        if (off <= (qreal)1e-64 * diag || off == 0) break;
        for (int p = 0; p < n; ++p)
            for (int q = p + 1; q < n; ++q) {
                const qreal apq = A[(size_t)p * n + q];
                if (apq == 0) continue;
                const qreal theta = (A[(size_t)q * n + q] - A[(size_t)p * n + p]) / (2 * apq);
                const qreal t = (theta >= 0 ? (qreal)1 : (qreal)-1) / (qabs(theta) + qsqrt(theta * theta + 1));
                const qreal cs = 1 / qsqrt(t * t + 1), sn = t * cs;
                for (int k = 0; k < n; ++k) {
                    const qreal akp = A[(size_t)k * n + p], akq = A[(size_t)k * n + q];
                    A[(size_t)k * n + p] = cs * akp - sn * akq; A[(size_t)k * n + q] = sn * akp + cs * akq;
                }
                for (int k = 0; k < n; ++k) {
                    const qreal apk = A[(size_t)p * n + k], aqk = A[(size_t)q * n + k];
                    A[(size_t)p * n + k] = cs * apk - sn * aqk; A[(size_t)q * n + k] = sn * apk + cs * aqk;
                }
                for (int k = 0; k < n; ++k) {
                    const qreal vkp = V[(size_t)k * n + p], vkq = V[(size_t)k * n + q];
                    V[(size_t)k * n + p] = cs * vkp - sn * vkq; V[(size_t)k * n + q] = sn * vkp + cs * vkq;
                }
            }
    }
}
// nu_F = pinv(G_FF) c_F on the index set F, zero elsewhere.  Cholesky first (a few thousand 113-bit operations); a pivot below 1e-26 of the
// trace = dependent planes -> the pseudo-inverse through the eigendecomposition (eigenvalues below that cut dropped)
static void long_solve_support(int K, const std::vector<qreal>& G, const std::vector<qreal>& cvec, const std::vector<int>& F, std::vector<qreal>& nu) {
    const int nf = (int)F.size();
    nu.assign((size_t)K, (qreal)0);
    if (nf == 0) return;
    std::vector<qreal> A((size_t)nf * nf), V;
    qreal tr = 0;
    for (int i = 0; i < nf; ++i) { for (int j = 0; j < nf; ++j) A[(size_t)i * nf + j] = G[(size_t)F[i] * K + F[j]]; tr += A[(size_t)i * nf + i]; }
    const qreal cut = (qreal)1e-26 * (tr > 0 ? tr : (qreal)1e-300);
    {
        std::vector<qreal> L(A);
        bool ok = true;
        for (int j = 0; j < nf && ok; ++j) {
            qreal dj = L[(size_t)j * nf + j];
            for (int k = 0; k < j; ++k) dj -= L[(size_t)j * nf + k] * L[(size_t)j * nf + k];
            if (!(dj > cut)) { ok = false; break; }
            const qreal ljj = qsqrt(dj);
            L[(size_t)j * nf + j] = ljj;
            for (int i = j + 1; i < nf; ++i) {
                qreal v = L[(size_t)i * nf + j];
                for (int k = 0; k < j; ++k) v -= L[(size_t)i * nf + k] * L[(size_t)j * nf + k];
                L[(size_t)i * nf + j] = v / ljj;
            }
        }
        if (ok) {
            std::vector<qreal> y((size_t)nf);
            for (int i = 0; i < nf; ++i) { qreal v = cvec[(size_t)F[i]]; for (int k = 0; k < i; ++k) v -= L[(size_t)i * nf + k] * y[(size_t)k]; y[(size_t)i] = v / L[(size_t)i * nf + i]; }
            for (int i = nf - 1; i >= 0; --i) { qreal v = y[(size_t)i]; for (int k = i + 1; k < nf; ++k) v -= L[(size_t)k * nf + i] * y[(size_t)k]; y[(size_t)i] = v / L[(size_t)i * nf + i]; }
            for (int i = 0; i < nf; ++i) nu[(size_t)F[i]] = y[(size_t)i];
            return;
        }
    }
    sym_eig_jacobi(nf, A, V);
    for (int e = 0; e < nf; ++e) {
        const qreal lam = A[(size_t)e * nf + e];
        if (!(lam > cut)) continue;
        qreal proj = 0;
        for (int i = 0; i < nf; ++i) proj += V[(size_t)i * nf + e] * cvec[(size_t)F[i]];
        proj /= lam;
        for (int i = 0; i < nf; ++i) nu[(size_t)F[i]] += V[(size_t)i * nf + e] * proj;
    }
}
int long_project(fos_solver* h, int64_t i) {
    FOS_TRY(fos::long_project_planes(h->ctx(), h->lp, h->X, i));
    h->shift_ready = false;                                                   // (the iterate changed behind the step's last kernel)
    return FOS_OK;
}
int long_finish(fos_solver* h, int64_t i) {
    LaunchCtx c = h->ctx();
    switch (h->alg) {
        case FOS_ALG_GAP:
        case FOS_ALG_GAPA: long_save(h, c, 1, h->T2, h->T1); break;           // addprojineq(longstep, y, x)
        case FOS_ALG_FISTA: long_save(h, c, 1, h->X, h->T1); break;           // addprojineq(longstep, x, tmp1)
        case FOS_ALG_DYKSTRA: long_save(h, c, 1, h->X, h->W); break;          // addprojineq(longstep, x, y .+ q)
    }
    FOS_TRY(step_finish_launch(h, c));
    if (h->lp.savepos == h->lp.nsave + 1) {                               // longstep.jl:53-58
        FOS_TRY(long_project(h, i));
        h->lp.savepos = -1;
    }
    return FOS_OK;
}

// one outer iteration; *check_on receives the vector checkstatus is evaluated on (the cone-feasible point)

// prox!(.., S1, in) followed by `post` -- everything of the step behind the CG solve (relaxation, cone projection and, when no
// status check sits in between, the step's last pass).  `post` is handed to the solve, which enqueues it behind the first CG
// batch, gated, so that the GPU does not wait for the host to learn the iteration count (cg_solve); if the batch did not
// suffice the gated launches were no-ops: the host-side state they advanced is rolled back and `post` runs again, ungated.
int affine_then(fos_solver* h, const LaunchCtx& c, const d2* in, const PostFn& post) {
    const int psd_cur = h->psd_cur, psd_have_prev = h->psd_have_prev;
    const size_t prof_used = h->prof_used;
    const int64_t psd_seen = h->prof_seen[FOS_PROF_PSD];
    const double fista_t = h->fista_t;
    bool ran = false;
    FOS_TRY(prox_affine(h, in, &post, &ran));
    if (!ran) {
        h->psd_cur = psd_cur; h->psd_have_prev = psd_have_prev; h->prof_seen[FOS_PROF_PSD] = psd_seen; h->fista_t = fista_t;
        // (event pairs recorded around no-op launches: forget them; the CG records of this solve stay)
        for (size_t k = prof_used; k < h->prof_used; ++k)
            if (h->prof_recs[k].cls == FOS_PROF_PSD || (h->prof_recs[k].cls == FOS_PROF_OTHER && h->prof_recs[k].j == 1)) h->prof_recs[k].cls = -1;
        FOS_TRY(post(c));
    }
    return FOS_OK;
}
int step_once(fos_solver* h, int64_t i, const d2** check_on, bool will_check, bool* finish_done) {
    LaunchCtx c = h->ctx();
    *finish_done = false;
    h->ls_now = h->ls_interval > 0 && (i % h->ls_interval) == 0;                                    // linesearch.jl:39
    if (h->ls_now) return ls_begin(h, check_on);
    h->gapp_now = h->gapp_iproj > 0 && (i % h->gapp_iproj) == 0;                                     // gapproj.jl:34
    if (h->gapp_now) return gapp_begin(h, i, check_on);
    h->lp.now = false;
    if (h->lp.interval > 0) {                                                                      // longstep.jl:44-49
        const int64_t savepos = (i - 1) % h->lp.interval - h->lp.interval + h->lp.nsave + 2;
        if (savepos > 0) h->lp.savepos = savepos;
        h->lp.now = h->lp.savepos > 0;
        if (h->lp.now) return long_begin(h, i, check_on);
    }
    switch (h->alg) {
        case FOS_ALG_GAP:                                                // gap.jl:61-80
        case FOS_ALG_GAPA: {                                             // gapa.jl:80-105
            // everything behind the CG solve of S1! -- and, when no status check sits in between, the final relaxation of
            // the step too -- is handed to the solve as its `post` work (cg_solve): enqueued behind the first CG batch,
            // gated, so that the GPU does not wait for the host to learn the iteration count
            const bool gapa = h->alg == FOS_ALG_GAPA;
            const PostFn post = [h, gapa, will_check](const LaunchCtx& cg) -> int {
                static const bool fuse_ew = !(getenv("FOS_RELAX_EW") && atoi(getenv("FOS_RELAX_EW")) == 0);
                static const bool fuse_psd = !(getenv("FOS_PSD_FUSE") && atoi(getenv("FOS_PSD_FUSE")) == 0);
                // GAP / DR, no status check in this step, every non-elementwise cone a PSD(64) cone with a basis from the last projection:
                // relaxation, projection and the step's last pass are ONE launch (PsdFuse, fos_internal.hpp)
                if (fuse_psd && !gapa && !will_check && h->nsoc == 0 && h->nexp == 0 && !h->psd_big && !h->ls_interval && !h->gapp_iproj && !h->lp.interval &&
                    psd_fuse_possible(cg, h->npsd, h->psd_kmin, h->psd_kmax, h->psd_V[h->psd_cur], h->psd_V[1 - h->psd_cur], h->psd_have_prev, h->psd_redo,
                                      h->psd_phase_limit)) {
                    RoctxRange range("fos:relaxation + PSD projection + final pass (one launch)");
                    const bool sh = h->in_step && h->shift_fuse && !h->direct;
                    PsdFuse fz{};
                    fz.sol = h->SOL; fz.xv = h->X; fz.shift = sh ? h->RHS : nullptr;
                    fz.a1 = h->alpha1; fz.alpha = h->alpha; fz.alpha2 = h->alpha2;
                    fz.ew_op = h->ew_op; fz.l = h->l;
                    const int pe = prof_begin(h, FOS_PROF_PSD, 0, h->prof_seen[FOS_PROF_PSD]++);
                    FOS_TRY(launch_cones_psd(cg, h->T2, h->T1, h->psd, h->npsd, h->psd_kmin, h->psd_kmax, h->psd_scratch, h->psd_V[h->psd_cur],
                                             h->psd_V[1 - h->psd_cur], h->psd_have_prev, h->psd_stats, 0, h->psd_redo, &fz));
                    prof_end(h, pe);
                    h->psd_cur = 1 - h->psd_cur; h->psd_have_prev = std::min(h->psd_have_prev + 1, 2);
                    h->shift_ready = sh;
                    return check_launch("fused relaxation + PSD projection + final pass");
                }
                const int pg = cg.gate ? 1 : 0;
                int po = prof_begin_other(h, pg);
                if (fuse_ew) {                       // the relaxation and the elementwise cones of S2! in one pass
                    launch_relax_ew(cg, h->T1, h->T2, h->SOL, h->X, h->alpha1, gapa, h->ew_op);
                    prof_end(h, po);
                    FOS_TRY(prox_cones(h, h->T2, h->T1, cg.gate, true));
                } else {
                if (gapa) launch_relax_a12(cg, h->T1, h->SOL, h->X);                          // gapa.jl:67
                else launch_axpby(cg, h->T1, h->alpha1, h->SOL, 1 - h->alpha1, h->X);          //   y = a1 y + (1-a1) x     :48
                prof_end(h, po);
                FOS_TRY(prox_cones(h, h->T2, h->T1, cg.gate));                                // S2!: prox!(y,S2,x)  :55
                }
                if (!will_check) { po = prof_begin_other(h, pg); FOS_TRY(step_finish_launch(h, cg)); prof_end(h, po); }
                return FOS_OK;
            };
            FOS_TRY(affine_then(h, c, h->X, post));                      // S1!: prox!(y,S1,x)          :45
            *check_on = h->T2;                                           //   checkstatus(status, y)    :56
            *finish_done = !will_check;
            return FOS_OK;
        }
        case FOS_ALG_FISTA: {                                            // fista.jl:28-48
            if (i == 1) FOS_HIP(hipMemcpyAsync(h->Y, h->X, sizeof(d2) * h->l, hipMemcpyDeviceToDevice, h->stream));   // :31-33
            const PostFn post = [h, will_check](const LaunchCtx& cg) -> int {
                const int pg = cg.gate ? 1 : 0;
                int po = prof_begin_other(h, pg);
                launch_axpby(cg, h->T1, h->alpha, h->SOL, 1 - h->alpha, h->Y);                 // :37
                launch_copy(cg, h->XOLD, h->X);                                               // :39  xold .= x
                prof_end(h, po);
                FOS_TRY(prox_cones(h, h->X, h->T1, cg.gate));                                 // :40
                if (!will_check) { po = prof_begin_other(h, pg); FOS_TRY(step_finish_launch(h, cg)); prof_end(h, po); }
                return FOS_OK;
            };
            FOS_TRY(affine_then(h, c, h->Y, post));                      // :35
            *check_on = h->X;                                            // :41
            *finish_done = !will_check;
            return FOS_OK;
        }
        case FOS_ALG_DYKSTRA: {                                          // dykstra.jl:25-36   (p = Y, q = XOLD)
            launch_add(c, h->W, h->X, h->Y);                             // x .+ p
            const PostFn post = [h, will_check](const LaunchCtx& cg) -> int {
                const int pg = cg.gate ? 1 : 0;
                int po = prof_begin_other(h, pg);
                launch_dykstra_corr(cg, h->Y, h->X, h->SOL);                                  // p .= x .+ p .- y            :30
                launch_add(cg, h->W, h->SOL, h->XOLD);                                        // y .+ q
                prof_end(h, po);
                FOS_TRY(prox_cones(h, h->X, h->W, cg.gate));                                  // prox!(x, S2, y .+ q)        :31
                if (!will_check) { po = prof_begin_other(h, pg); FOS_TRY(step_finish_launch(h, cg)); prof_end(h, po); }
                return FOS_OK;
            };
            FOS_TRY(affine_then(h, c, h->W, post));                      // prox!(y, S1, x .+ p)        :28
            *check_on = h->X;                                            // :32
            *finish_done = !will_check;
            return FOS_OK;
        }
    }
    set_error("unknown algorithm %d", h->alg);
    return FOS_EINVAL;
}

// the part of the step after checkstatus
int step_finish(fos_solver* h, int64_t i) {
    LaunchCtx c = h->ctx();
    if (h->ls_now) { h->ls_now = false; return ls_finish(h, i); }
    if (h->gapp_now) { h->gapp_now = false; return gapp_finish(h); }
    if (h->lp.now) { h->lp.now = false; return long_finish(h, i); }
    return step_finish_launch(h, c);
}
int step_finish_launch(fos_solver* h, const LaunchCtx& c) {
    switch (h->alg) {
        case FOS_ALG_GAP: {
            // (inside fos_step, CG projection: the kernel also leaves SOL - [0; X.y] in RHS for the next iteration's CG start)
            const bool sh = h->in_step && h->shift_fuse && !h->direct;
            launch_gap_final(c, h->X, h->T2, h->T1, h->alpha, h->alpha2, sh ? h->RHS : nullptr, h->SOL);          // gap.jl:58,78
            h->shift_ready = sh;
            return FOS_OK;
        }
        case FOS_ALG_GAPA: {
            const bool sh = h->in_step && h->shift_fuse && !h->direct;
            launch_gapa_final(c, h->X, h->T2, h->T1, h->alpha, sh ? h->RHS : nullptr, h->SOL);                    // gapa.jl:77,96,103
            h->shift_ready = sh;
            int fr = 0;
            FOS_TRY(finish_reduce(h, c, c.vec_blocks, 3, 0, &fr));
            launch_gapa_finalize(c, h->beta, fr);                                  // gapa.jl:96-101
            return FOS_OK;
        }
        case FOS_ALG_FISTA: {
            const double told = h->fista_t;                                        // fista.jl:44-46
            h->fista_t = (1 + std::sqrt(1 + 4 * told * told)) / 2;
            launch_fista_extrap(c, h->Y, h->X, h->XOLD, (told - 1) / h->fista_t);
            return FOS_OK;
        }
        case FOS_ALG_DYKSTRA:
            launch_dykstra_corr(c, h->XOLD, h->SOL, h->X);                         // q .= y .+ q .- x   dykstra.jl:34
            return FOS_OK;
    }
    return FOS_EINVAL;
}

int upload_plain(fos_solver* h, d2* dst, const double* z) {
    LaunchCtx c = h->ctx();
    FOS_HIP(hipMemcpyAsync(h->plain, z, sizeof(double) * 2 * h->l, hipMemcpyHostToDevice, h->stream));
    launch_interleave(c, dst, h->plain);
    FOS_HIP(hipStreamSynchronize(h->stream));
    return FOS_OK;
}
int download_plain(fos_solver* h, double* z, const d2* src) {
    LaunchCtx c = h->ctx();
    launch_deinterleave(c, h->plain, src);
    FOS_HIP(hipMemcpyAsync(z, h->plain, sizeof(double) * 2 * h->l, hipMemcpyDeviceToHost, h->stream));
    FOS_HIP(hipStreamSynchronize(h->stream));
    return FOS_OK;
}

int validate_cones(const char* which, int64_t total, int64_t nK, const int32_t* type, const int64_t* start, const int64_t* len) {
    int64_t prev_end = 0;                                   // cones.jl:66-72
    for (int64_t i = 0; i < nK; ++i) {
        if (start[i] != prev_end + 1) { set_error("%s cone %lld: range starts at %lld, expected %lld (ranges must be contiguous and ordered)", which, (long long)i + 1, (long long)start[i], (long long)prev_end + 1); return FOS_EINVAL; }
        if (len[i] < 1) { set_error("%s cone %lld: empty range", which, (long long)i + 1); return FOS_EINVAL; }
        if (len[i] > (int64_t)INT32_MAX) { set_error("%s cone %lld: %lld entries exceed the int32 cone descriptor", which, (long long)i + 1, (long long)len[i]); return FOS_EUNSUPPORTED; }
        prev_end = start[i] + len[i] - 1;
        switch (type[i]) {
            case FOS_CONE_FREE: case FOS_CONE_ZERO: case FOS_CONE_NONNEG: case FOS_CONE_NONPOS: case FOS_CONE_SOC: break;
            case FOS_CONE_SOCROT:
                if (len[i] < 2) { set_error("%s cone %lld: rotated SOC needs >= 2 entries", which, (long long)i + 1); return FOS_EINVAL; }
                break;
            case FOS_CONE_SDP:
                if (psd_order(len[i]) < 0) { set_error("%s cone %lld: SDP length %lld is not k(k+1)/2", which, (long long)i + 1, (long long)len[i]); return FOS_EINVAL; }
                break;
            case FOS_CONE_EXPPRIMAL: case FOS_CONE_EXPDUAL:
                if (len[i] != 3) { set_error("%s cone %lld: an exponential cone has exactly 3 entries (got %lld)", which, (long long)i + 1, (long long)len[i]); return FOS_EINVAL; }
                break;
            default:
                set_error("%s cone %lld: unknown cone code %d", which, (long long)i + 1, (int)type[i]);
                return FOS_EINVAL;
        }
    }
    if (prev_end != total) { set_error("%s cones cover 1..%lld, expected 1..%lld", which, (long long)prev_end, (long long)total); return FOS_EINVAL; }
    return FOS_OK;
}

void add_cones(int64_t offset, bool is_K1, int64_t nK, const int32_t* type, const int64_t* start, const int64_t* len,
               std::vector<uint8_t>& ew, std::vector<ConeDesc>& soc, std::vector<ConeDesc>& psd, std::vector<ConeDesc>& expc) {
    // K2 cone on [x | r]: part1 primal, part2 dual.   K1 cone on [y | s]: part1 dual, part2 primal.   cones.jl:136-140
    for (int64_t i = 0; i < nK; ++i) {
        const int64_t s0 = offset + start[i] - 1;
        uint8_t prim = 0, dual = 0;
        bool ewise = true;
        switch (type[i]) {
            case FOS_CONE_FREE: prim = EW_COPY; dual = EW_ZERO; break;      // cones.jl:100
            case FOS_CONE_ZERO: prim = EW_ZERO; dual = EW_COPY; break;      // cones.jl:98
            case FOS_CONE_NONNEG: prim = dual = EW_MAX0; break;             // cones.jl:101
            case FOS_CONE_NONPOS: prim = dual = EW_MIN0; break;             // cones.jl:102
            default: ewise = false;
        }
        if (ewise) {
            const uint8_t op = is_K1 ? (uint8_t)(dual | (prim << 2)) : (uint8_t)(prim | (dual << 2));
            for (int64_t k = 0; k < len[i]; ++k) ew[s0 + k] = op;
        } else {
            for (int64_t k = 0; k < len[i]; ++k) ew[s0 + k] = EW_SKIP;
            ConeDesc cd;
            cd.start = s0; cd.len = (int32_t)len[i]; cd.type = type[i];
            cd.dual_part = is_K1 ? 0 : 1;
            cd.k = type[i] == FOS_CONE_SDP ? psd_order(len[i]) : 0;
            if (type[i] == FOS_CONE_SDP) psd.push_back(cd);
            else if (type[i] == FOS_CONE_EXPPRIMAL || type[i] == FOS_CONE_EXPDUAL) expc.push_back(cd);
            else soc.push_back(cd);
        }
    }
}

// ---- direct = true, block-separable operators: the set-up.  Columns j, j' of A belong to one block when they share a row (connected components of the
// pattern of A'A); separable = every component has at most BLKDIR_MAX columns.  Per block G_b = I + A_b' A_b is formed on the host (row-wise outer
// products), inverted through its Cholesky factor and polished by one Newton step in extended precision; ph, pg and the 3 x 3 border system follow
// from two runs of the device path itself.  *ok = false: not separable (nothing allocated).
// v[0..2] <- their sums over the ranks of a sharded handle (set-up: through the handle's transport, one host round trip); no-op on one GPU
int global_sum3(fos_solver* h, double* v) {
    if (!h->sharded()) return FOS_OK;
    LaunchCtx c = h->ctx();
    FOS_HIP(hipMemcpyAsync(h->partials, v, sizeof(double) * 3, hipMemcpyHostToDevice, h->stream));
    launch_reduce1(c, 1, 3, 0);
    FOS_TRY(allreduce(h, 3));
    FOS_HIP(hipMemcpyAsync(v, h->reduced, sizeof(double) * 3, hipMemcpyDeviceToHost, h->stream));
    return poll_state(h);
}

int blkdir_setup(fos_solver* h, const int64_t* colptr, const int64_t* rowval, const double* nzval, bool* ok) {
    *ok = false;
    const int64_t n = h->n, m = h->m, l = h->l;
    // (sharded handles: every rank takes part in the vote below, whatever its own operator looks like)
    bool local_ok = !(n < 1 || n > (int64_t)INT32_MAX);
    if (!local_ok && !h->sharded()) return FOS_OK;
    std::vector<int32_t> parent((size_t)n);
    for (int64_t j = 0; j < n; ++j) parent[(size_t)j] = (int32_t)j;
    auto find = [&](int32_t a) { while (parent[(size_t)a] != a) { parent[(size_t)a] = parent[(size_t)parent[(size_t)a]]; a = parent[(size_t)a]; } return a; };
    bool bad_index = false;
    {
        std::vector<int32_t> first((size_t)m, -1);
        for (int64_t j = 0; j < n && !bad_index; ++j)
            for (int64_t p = colptr[j] - 1; p < colptr[j + 1] - 1; ++p) {
                const int64_t r = rowval[p] - 1;
                if (r < 0 || r >= m) { bad_index = true; local_ok = false; break; }          // (reported behind the vote: the peers of a sharded handle are waiting in it)
                if (first[(size_t)r] < 0) { first[(size_t)r] = (int32_t)j; continue; }
                const int32_t a = find((int32_t)j), b = find(first[(size_t)r]);
                if (a != b) parent[(size_t)std::max(a, b)] = std::min(a, b);       // the root of a component is its smallest column
            }
    }
    std::vector<int32_t> cnt((size_t)n, 0), blkid((size_t)n, -1);
    for (int64_t j = 0; j < n; ++j) cnt[(size_t)find((int32_t)j)] += 1;
    int nblk = 0;
    size_t gtotal = 0;
    for (int64_t j = 0; j < n && local_ok; ++j)
        if (cnt[(size_t)j] > 0) {
            if (cnt[(size_t)j] > BLKDIR_MAX) { local_ok = false; break; }          // a block of I + A'A too large to invert densely per wavefront
            blkid[(size_t)j] = nblk++;
            gtotal += (size_t)cnt[(size_t)j] * (size_t)cnt[(size_t)j];
        }
    if (gtotal * sizeof(double) > ((size_t)1 << 31)) local_ok = false;
    {
        // the form is taken by ALL ranks or by none (its reductions are collective)
        double vote[3] = {local_ok ? 1.0 : 0.0, 0.0, 0.0};
        FOS_TRY(global_sum3(h, vote));
        if (bad_index) { set_error("fos_enable_direct: row index out of range"); return FOS_EINVAL; }
        if (h->sharded() ? vote[0] != (double)h->nranks : !local_ok) return FOS_OK;
    }
    std::vector<int32_t> ioff((size_t)nblk + 1, 0), idx((size_t)n);
    std::vector<int64_t> goff((size_t)nblk, 0);
    for (int64_t j = 0; j < n; ++j) if (cnt[(size_t)j] > 0) ioff[(size_t)blkid[(size_t)j] + 1] = cnt[(size_t)j];
    for (int b = 0; b < nblk; ++b) { goff[(size_t)b] = b ? goff[(size_t)b - 1] + (int64_t)(ioff[(size_t)b] - ioff[(size_t)b - 1]) * (ioff[(size_t)b] - ioff[(size_t)b - 1]) : 0; ioff[(size_t)b + 1] += ioff[(size_t)b]; }
    {
        std::vector<int32_t> fill(ioff.begin(), ioff.end() - 1);
        for (int64_t j = 0; j < n; ++j) idx[(size_t)fill[(size_t)blkid[(size_t)find((int32_t)j)]]++] = (int32_t)j;      // ascending inside a block
    }
    std::vector<double> ginv(gtotal, 0.0);
    std::vector<int32_t> rowlocal((size_t)m, -1);            // (the blocks' row sets are disjoint: one shared map, no conflicts between threads)
    std::atomic<int> next{0}, bad{0};
    auto work = [&]() {
        std::vector<int32_t> rcount, rstart, ecol;
        std::vector<double> eval, G;
        std::vector<long double> Lc, X, Y;
        for (;;) {
            const int b = next.fetch_add(1);
            if (b >= nblk) break;
            const int32_t i0 = ioff[(size_t)b], sdim = ioff[(size_t)b + 1] - i0;
            // rows of the block, in order of first appearance; entries bucketed by row
            int32_t nrows = 0;
            int64_t nent = 0;
            rcount.clear();
            for (int32_t q = 0; q < sdim; ++q) {
                const int64_t j = idx[(size_t)i0 + q];
                for (int64_t p = colptr[j] - 1; p < colptr[j + 1] - 1; ++p) {
                    const int64_t r = rowval[p] - 1;
                    if (rowlocal[(size_t)r] < 0) { rowlocal[(size_t)r] = nrows++; rcount.push_back(0); }
                    rcount[(size_t)rowlocal[(size_t)r]] += 1;
                    ++nent;
                }
            }
            rstart.assign((size_t)nrows + 1, 0);
            for (int32_t r = 0; r < nrows; ++r) rstart[(size_t)r + 1] = rstart[(size_t)r] + rcount[(size_t)r];
            ecol.resize((size_t)nent); eval.resize((size_t)nent);
            std::fill(rcount.begin(), rcount.end(), 0);
            for (int32_t q = 0; q < sdim; ++q) {
                const int64_t j = idx[(size_t)i0 + q];
                for (int64_t p = colptr[j] - 1; p < colptr[j + 1] - 1; ++p) {
                    const int32_t r = rowlocal[(size_t)(rowval[p] - 1)];
                    const size_t at = (size_t)rstart[(size_t)r] + (size_t)rcount[(size_t)r]++;
                    ecol[at] = q; eval[at] = nzval[p];
                }
            }
            G.assign((size_t)sdim * sdim, 0.0);
            for (int32_t r = 0; r < nrows; ++r)
                for (int32_t a = rstart[(size_t)r]; a < rstart[(size_t)r + 1]; ++a)
                    for (int32_t c2 = a; c2 < rstart[(size_t)r + 1]; ++c2) G[(size_t)ecol[(size_t)a] * sdim + ecol[(size_t)c2]] += eval[(size_t)a] * eval[(size_t)c2];
            for (int32_t a = 0; a < sdim; ++a) {
                G[(size_t)a * sdim + a] += 1.0;
                for (int32_t c2 = a + 1; c2 < sdim; ++c2) {          // (entries of a row arrive in ascending column order: the upper triangle was filled)
                    const double v = G[(size_t)a * sdim + c2] + G[(size_t)c2 * sdim + a];
                    G[(size_t)a * sdim + c2] = G[(size_t)c2 * sdim + a] = v;
                }
            }
            // X = G^-1 through Cholesky (extended precision: the blocks are tiny), one Newton step X <- X + X (I - G X)
            Lc.assign((size_t)sdim * sdim, 0.0L);
            bool pd = true;
            for (int32_t j = 0; j < sdim && pd; ++j) {
                long double dj = G[(size_t)j * sdim + j];
                for (int32_t k = 0; k < j; ++k) dj -= Lc[(size_t)j * sdim + k] * Lc[(size_t)j * sdim + k];
                if (!(dj > 0.0L)) { pd = false; break; }
                const long double ljj = sqrtl(dj);
                Lc[(size_t)j * sdim + j] = ljj;
                for (int32_t i = j + 1; i < sdim; ++i) {
                    long double v = G[(size_t)i * sdim + j];
                    for (int32_t k = 0; k < j; ++k) v -= Lc[(size_t)i * sdim + k] * Lc[(size_t)j * sdim + k];
                    Lc[(size_t)i * sdim + j] = v / ljj;
                }
            }
            if (!pd) { bad.store(1); continue; }
            X.assign((size_t)sdim * sdim, 0.0L);
            Y.assign((size_t)sdim, 0.0L);
            for (int32_t e = 0; e < sdim; ++e) {
                for (int32_t i = 0; i < sdim; ++i) { long double v = (i == e) ? 1.0L : 0.0L; for (int32_t k = 0; k < i; ++k) v -= Lc[(size_t)i * sdim + k] * Y[(size_t)k]; Y[(size_t)i] = v / Lc[(size_t)i * sdim + i]; }
                for (int32_t i = sdim - 1; i >= 0; --i) { long double v = Y[(size_t)i]; for (int32_t k = i + 1; k < sdim; ++k) v -= Lc[(size_t)k * sdim + i] * X[(size_t)e * sdim + k]; X[(size_t)e * sdim + i] = v / Lc[(size_t)i * sdim + i]; }
            }
            double* out = ginv.data() + goff[(size_t)b];
            for (int32_t e = 0; e < sdim; ++e)
                for (int32_t i = 0; i < sdim; ++i) out[(size_t)e * sdim + i] = (double)((X[(size_t)e * sdim + i] + X[(size_t)i * sdim + e]) / 2);      // symmetric, column-major
        }
    };
    {
        const unsigned hw = std::max(1u, std::min(16u, std::thread::hardware_concurrency()));
        std::vector<std::thread> pool;
        for (unsigned t = 1; t < hw; ++t) pool.emplace_back(work);
        work();
        for (auto& t : pool) t.join();
    }
    {
        double vote[3] = {bad.load() ? 1.0 : 0.0, 0.0, 0.0};       // (collective again: a rank that failed here must not leave its peers in the sums below)
        FOS_TRY(global_sum3(h, vote));
        if (bad.load()) { set_error("fos_enable_direct: I + A'A has a block that is not positive definite (non-finite entries in A?)"); return FOS_EINVAL; }
        if (vote[0] != 0.0) { set_error("fos_enable_direct: I + A'A has a block that is not positive definite on another rank"); return FOS_EINVAL; }
    }
    // ---- device data
    FOS_TRY(dev_upload(h, &h->blk_goff, goff));
    FOS_TRY(dev_upload(h, &h->blk_ioff, ioff));
    FOS_TRY(dev_upload(h, &h->blk_idx, idx));
    FOS_TRY(dev_upload(h, &h->blk_ginv, ginv));
    h->blk_n = nblk;
    FOS_TRY(dev_alloc(h, &h->blk_phg, (size_t)l));
    FOS_TRY(dev_alloc(h, &h->blk_qphg, (size_t)l));
    FOS_TRY(dev_alloc(h, &h->blk_prm, 16));
    FOS_TRY(dev_alloc(h, &h->blk_ctx, (size_t)nblk));
    // (rows of A that the sweep does not finish itself -- rows wider than one tile chunk are slot-spread too -- need the deferred-row kernel behind the third apply)
    h->blk_skip_tail = !(getenv("FOS_BLKDIR_FULL_APPLY") && atoi(getenv("FOS_BLKDIR_FULL_APPLY")) != 0);
    for (int32_t r : h->hostS.def_rows) if (r >= n) { h->blk_skip_tail = false; break; }
    FOS_HIP(hipMemset(h->blk_phg, 0, sizeof(d2) * (size_t)l));
    FOS_HIP(hipMemset(h->blk_qphg, 0, sizeof(d2) * (size_t)l));
    // ---- the border: h = [c; b], M h = [A'b; -A c], ph = D^-1 (h; 0), pg = D^-1 (M h; 0) by two runs of the device path without the border terms
    std::vector<double> cbv((size_t)(n + m));
    FOS_HIP(hipMemcpy(cbv.data(), h->cb, sizeof(double) * (size_t)(n + m), hipMemcpyDeviceToHost));
    long double hh2 = 0.0L;
    for (double v : cbv) hh2 += (long double)v * v;
    {
        double g3[3] = {(double)hh2, 0.0, 0.0};                    // |[c; b]|^2 over all ranks
        FOS_TRY(global_sum3(h, g3));
        hh2 = g3[0];
    }
    const double delta = (double)(1.0L + hh2);
    std::vector<double> prm(16, 0.0);
    prm[9] = delta;
    FOS_HIP(hipMemcpy(h->blk_prm, prm.data(), sizeof(double) * 16, hipMemcpyHostToDevice));
    std::vector<double> mh((size_t)(n + m), 0.0);
    for (int64_t j = 0; j < n; ++j) {
        long double acc = 0.0L;
        for (int64_t p = colptr[j] - 1; p < colptr[j + 1] - 1; ++p) {
            const int64_t r = rowval[p] - 1;
            acc += (long double)nzval[p] * cbv[(size_t)(n + r)];
            mh[(size_t)(n + r)] -= nzval[p] * cbv[(size_t)j];
        }
        mh[(size_t)j] = (double)acc;
    }
    std::vector<d2> tv((size_t)l), res((size_t)l), phg((size_t)l), qphg((size_t)l);
    for (int pass = 0; pass < 2; ++pass) {
        const std::vector<double>& src = pass == 0 ? cbv : mh;
        for (int64_t i = 0; i < l - 1; ++i) tv[(size_t)i] = make_double2(src[(size_t)i], 0.0);
        tv[(size_t)l - 1] = make_double2(0.0, 0.0);
        FOS_HIP(hipMemcpy(h->R, tv.data(), sizeof(d2) * (size_t)l, hipMemcpyHostToDevice));
        FOS_TRY(prox_affine_direct_block(h, nullptr, h->W, true, true));
        FOS_HIP(hipStreamSynchronize(h->stream));
        FOS_HIP(hipMemcpy(res.data(), h->W, sizeof(d2) * (size_t)l, hipMemcpyDeviceToHost));
        for (int64_t i = 0; i < l; ++i) {
            if (pass == 0) { phg[(size_t)i].x = res[(size_t)i].x; qphg[(size_t)i].x = res[(size_t)i].y; }
            else { phg[(size_t)i].y = res[(size_t)i].x; qphg[(size_t)i].y = res[(size_t)i].y; }
        }
    }
    long double hph = 0, hpg = 0, gph = 0, gpg = 0;
    for (int64_t i = 0; i < l - 1; ++i) {
        hph += (long double)cbv[(size_t)i] * phg[(size_t)i].x; hpg += (long double)cbv[(size_t)i] * phg[(size_t)i].y;
        gph += (long double)mh[(size_t)i] * phg[(size_t)i].x; gpg += (long double)mh[(size_t)i] * phg[(size_t)i].y;
    }
    {
        double g3[3] = {(double)hph, (double)hpg, (double)gph}, g1[3] = {(double)gpg, 0.0, 0.0};       // the border's dots over all ranks
        FOS_TRY(global_sum3(h, g3));
        FOS_TRY(global_sum3(h, g1));
        if (h->sharded()) { hph = g3[0]; hpg = g3[1]; gph = g3[2]; gpg = g1[0]; }
    }
    // S3 = C^-1 + W' D^-1 W,  C^-1 = [1 0 0; 0 0 -1; 0 -1 0]
    long double S3[3][3] = {{1 + hph, hpg, 0}, {gph, gpg, -1}, {0, -1, 1 / (long double)delta}}, Inv[3][3];
    const long double det = S3[0][0] * (S3[1][1] * S3[2][2] - S3[1][2] * S3[2][1]) - S3[0][1] * (S3[1][0] * S3[2][2] - S3[1][2] * S3[2][0]) +
                            S3[0][2] * (S3[1][0] * S3[2][1] - S3[1][1] * S3[2][0]);
    if (!(fabsl(det) > 0.0L) || !std::isfinite((double)det)) { set_error("fos_enable_direct: the 3 x 3 border system of the block form is singular"); return FOS_EINVAL; }
    for (int a = 0; a < 3; ++a)
        for (int bq = 0; bq < 3; ++bq) {
            const int a1 = (a + 1) % 3, a2 = (a + 2) % 3, b1 = (bq + 1) % 3, b2 = (bq + 2) % 3;
            Inv[bq][a] = (S3[a1][b1] * S3[a2][b2] - S3[a1][b2] * S3[a2][b1]) / det;          // adjugate, transposed
        }
    for (int a = 0; a < 3; ++a) for (int bq = 0; bq < 3; ++bq) prm[(size_t)(3 * a + bq)] = (double)Inv[a][bq];
    FOS_HIP(hipMemcpy(h->blk_prm, prm.data(), sizeof(double) * 16, hipMemcpyHostToDevice));
    FOS_HIP(hipMemcpy(h->blk_phg, phg.data(), sizeof(d2) * (size_t)l, hipMemcpyHostToDevice));
    FOS_HIP(hipMemcpy(h->blk_qphg, qphg.data(), sizeof(d2) * (size_t)l, hipMemcpyHostToDevice));
    *ok = true;
    return FOS_OK;
}

}  // namespace

namespace fos {
int long_project_planes(const LaunchCtx& c, LongPlanes& lp, d2* X, int64_t i) {
    const int K = (int)(2 * (lp.nsave + 1)), neq = (int)(lp.nsave + 1), nin = K - neq;
    const int nb = c.vec_blocks;
    const size_t W = 2 * (LONG_KMAX_ROWS + 1);                                // (hi, lo) pairs per workgroup
    std::vector<qreal> G((size_t)K * K, (qreal)0), Px((size_t)K, (qreal)0), beta((size_t)K, (qreal)0);
    std::vector<double> part((size_t)nb * W), bp((size_t)K * nb);
    for (int a = 0; a < K; ++a) {
        launch_long_dots(c, lp.P, K, a, X, lp.dots);
        FOS_HIP(hipMemcpyAsync(part.data(), lp.dots, sizeof(double) * part.size(), hipMemcpyDeviceToHost, c.stream));
        FOS_HIP(hipStreamSynchronize(c.stream));
        for (int k = 0; a + k < K; ++k) {
            qreal sacc = 0;
            for (int b = 0; b < nb; ++b) sacc += (qreal)part[(size_t)b * W + 2 * k] + (qreal)part[(size_t)b * W + 2 * k + 1];
            G[(size_t)a * K + a + k] = G[(size_t)(a + k) * K + a] = sacc;
        }
        qreal sx = 0;
        for (int b = 0; b < nb; ++b) sx += (qreal)part[(size_t)b * W + 2 * LONG_KMAX_ROWS] + (qreal)part[(size_t)b * W + 2 * LONG_KMAX_ROWS + 1];
        Px[(size_t)a] = sx;
    }
    FOS_HIP(hipMemcpyAsync(bp.data(), lp.bpart, sizeof(double) * bp.size(), hipMemcpyDeviceToHost, c.stream));
    FOS_HIP(hipStreamSynchronize(c.stream));
    // (the offsets b = (x - y).y are float64 sums in the reference too, longstep.jl:71-75: data of the QP, not part of its solution)
    for (int a = 0; a < K; ++a) { double sacc = 0.0; for (int b = 0; b < nb; ++b) sacc += bp[(size_t)a * nb + b]; beta[(size_t)a] = (qreal)sacc; }
    std::vector<qreal> cvec((size_t)K);
    for (int a = 0; a < K; ++a) cvec[(size_t)a] = beta[(size_t)a] - Px[(size_t)a];
    qreal scale = 0;
    for (int a = 0; a < K; ++a) { const qreal v = qabs(cvec[(size_t)a]) + qsqrt(G[(size_t)a * K + a]); if (v > scale) scale = v; }
    if (!(scale > 0)) scale = (qreal)1e-300;
    std::vector<qreal> nu, best_nu((size_t)K, (qreal)0);
    qreal best_viol = (qreal)INFINITY;
    int best_active = 0;
    int64_t tried = 0;
    std::vector<int> F;
    std::vector<qreal> gres((size_t)K);
    // one candidate support: solve, residuals g = G nu - c, KKT violation (equality residuals, negative multipliers inside the support, violated
    // inequalities outside it); true when it is the solution
    auto try_support = [&](uint32_t mask) {
        F.clear();
        for (int a = 0; a < neq; ++a) F.push_back(a);
        for (int j = 0; j < nin; ++j) if (mask & (1u << j)) F.push_back(neq + j);
        long_solve_support(K, G, cvec, F, nu);
        ++tried;
        qreal viol = 0;
        for (int a = 0; a < K; ++a) {
            qreal g = -cvec[(size_t)a];                                       // (G nu - c)_a = P_a v - beta_a
            for (int b2 = 0; b2 < K; ++b2) g += G[(size_t)a * K + b2] * nu[(size_t)b2];
            gres[(size_t)a] = g;
            qreal va;
            if (a < neq) va = qabs(g);
            else if (mask & (1u << (a - neq))) { const qreal gd = G[(size_t)a * K + a]; const qreal m2 = -nu[(size_t)a] * qsqrt(gd > 0 ? gd : (qreal)1e-300); va = qabs(g) > m2 ? qabs(g) : m2; }
            else va = -g;
            if (va > viol) viol = va;
        }
        if (viol < best_viol) { best_viol = viol; best_nu = nu; best_active = __builtin_popcount(mask); }
        return best_viol <= (qreal)1e-12 * scale;
    };
    // (i) an active-set walk from the empty support: drop the most negative multiplier, else add the most violated inequality -- a handful of
    //     solves when it ends at the solution (it is accepted by the same KKT test); (ii) otherwise the enumeration of all supports
    bool found = false;
    // budget of candidate supports (FOS_LONG_MAX_SUPPORTS; 0 tries none: the failure path, tests)
    // (read once, at fos_set_longstep; a COUNT only -- a wall-clock cut-off made the iterate sequence depend on the load of the machine)
    const int64_t LONG_MAX_SUPPORTS = lp.max_supports;
    {
        uint32_t mask = 0;
        std::vector<uint32_t> seen;
        for (int it = 0; it < 4 * nin + 4 && !found && tried < LONG_MAX_SUPPORTS; ++it) {
            if (std::find(seen.begin(), seen.end(), mask) != seen.end()) break;
            seen.push_back(mask);
            if (try_support(mask)) { found = true; break; }
            int drop = -1, add = -1;
            qreal worst_nu = 0, worst_g = 0;
            for (int j = 0; j < nin; ++j) {
                const int a = neq + j;
                if (mask & (1u << j)) { if (nu[(size_t)a] < worst_nu) { worst_nu = nu[(size_t)a]; drop = j; } }
                else if (gres[(size_t)a] < worst_g) { worst_g = gres[(size_t)a]; add = j; }
            }
            if (drop >= 0) mask &= ~(1u << drop);
            else if (add >= 0) mask |= 1u << add;
            else break;
        }
    }
    // (ii) bounded: inconsistent or dependent planes have NO support that passes the KKT test, and 2^nin solves in 113-bit arithmetic (each a
    // Jacobi eigen-decomposition when Cholesky rejects the block) would hold fos_step for minutes -- at most LONG_MAX_SUPPORTS candidates, smallest
    // supports first in counting order (deterministic: the same problem projects, or does not, on every run)
    for (uint32_t mask = 0; !found && mask < (1u << nin) && tried < LONG_MAX_SUPPORTS; ++mask) found = try_support(mask);
    lp.log[6] = found ? 0.0 : 1.0;
    if (!found) lp.log[7] += 1.0;                              // projections given up since fos_set_longstep (the host mirrors warn when it grows)
    if (!found) {
        // no KKT point of the small dual within the budget: the planes do not describe a projection -- the iterate stays as the wrapped
        // algorithm left it (the reference's QP solver throws here; a step that is not the projection must not be applied silently)
        lp.log[0] = (double)i; lp.log[1] = (double)best_active; lp.log[2] = (double)(best_viol / scale); lp.log[3] = 0.0; lp.log[4] = (double)K; lp.log[5] = (double)tried;
        return FOS_OK;
    }
    qreal step2 = 0;                                                          // |P' nu|^2 = nu' G nu
    for (int a = 0; a < K; ++a) for (int b2 = 0; b2 < K; ++b2) step2 += best_nu[(size_t)a] * G[(size_t)a * K + b2] * best_nu[(size_t)b2];
    std::vector<double> nu2((size_t)2 * K);                                   // multipliers as (hi, lo) pairs for the double-double update
    for (int a = 0; a < K; ++a) { const double hi = (double)best_nu[(size_t)a]; nu2[(size_t)2 * a] = hi; nu2[(size_t)2 * a + 1] = (double)(best_nu[(size_t)a] - (qreal)hi); }
    FOS_HIP(hipMemcpyAsync(lp.nu, nu2.data(), sizeof(double) * 2 * K, hipMemcpyHostToDevice, c.stream));
    launch_long_apply(c, X, lp.P, K, lp.nu);                     // x .= longstep.tmp      longstep.jl:57
    FOS_HIP(hipStreamSynchronize(c.stream));                                 // (nu2 leaves scope)
    lp.log[0] = (double)i; lp.log[1] = (double)best_active; lp.log[2] = (double)best_viol; lp.log[3] = (double)qsqrt(step2 > 0 ? step2 : (qreal)0);
    lp.log[4] = (double)K; lp.log[5] = (double)tried;
    return FOS_OK;
}
void long_save_plane(const LaunchCtx& c, LongPlanes& lp, int which, const d2* y, const d2* x) {
    const int64_t row = 2 * (lp.savepos - 1) + which;
    launch_long_plane(c, lp.P + row * c.l, x, y, lp.bpart + row * c.vec_blocks);
}
}  // namespace fos

// ------------------------------------------------------------------------------------------------ C ABI
extern "C" {

int fos_abi_version(void) { return FOS_ABI_VERSION; }
const char* fos_last_error(void) { return fos::last_error_cstr(); }

int fos_device_count(int* count) {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) {
        if (count) *count = 0;
        set_error("no HIP device visible (%s): libfoship has no CPU fallback", e == hipSuccess ? "count = 0" : hipGetErrorString(e));
        return FOS_ENODEVICE;
    }
    if (count) *count = n;
    return FOS_OK;
}

int fos_device_name(int device, char* buf, int buflen) {
    hipDeviceProp_t prop;
    FOS_HIP(hipGetDeviceProperties(&prop, device));
    snprintf(buf, buflen, "%s (%s, %d CUs)", prop.name, prop.gcnArchName, prop.multiProcessorCount);
    return FOS_OK;
}

int fos_create(int64_t m, int64_t n, const int64_t* colptr, const int64_t* rowval, const double* nzval,
               const double* b, const double* c,
               int64_t nK1, const int32_t* K1type, const int64_t* K1start, const int64_t* K1len,
               int64_t nK2, const int32_t* K2type, const int64_t* K2start, const int64_t* K2len,
               int device, fos_handle* out) {
    return fos_create2(m, n, colptr, rowval, nzval, b, c, nK1, K1type, K1start, K1len, nK2, K2type, K2start, K2len, device, 0, out);
}

int fos_create2(int64_t m, int64_t n, const int64_t* colptr, const int64_t* rowval, const double* nzval,
                const double* b, const double* c,
                int64_t nK1, const int32_t* K1type, const int64_t* K1start, const int64_t* K1len,
                int64_t nK2, const int32_t* K2type, const int64_t* K2start, const int64_t* K2len,
                int device, int32_t flags, fos_handle* out) {
    if (!out) { set_error("out handle is NULL"); return FOS_EINVAL; }
    *out = nullptr;
    if (m < 0 || n < 0 || !colptr || (!rowval && colptr[n] > 1) || (!b && m) || (!c && n)) { set_error("NULL or negative argument"); return FOS_EINVAL; }
    int ndev = 0;
    FOS_TRY(fos_device_count(&ndev));
    if (device < 0 || device >= ndev) { set_error("device %d out of range (0..%d)", device, ndev - 1); return FOS_EINVAL; }
    FOS_TRY(validate_cones("K1", m, nK1, K1type, K1start, K1len));
    FOS_TRY(validate_cones("K2", n, nK2, K2type, K2start, K2len));
    FOS_HIP(hipSetDevice(device));

    struct Guard {                       // frees everything on an early error return
        fos_solver* h;
        ~Guard() { if (h) fos_destroy(h); }
    } guard{new fos_solver()};
    fos_solver* h = guard.h;
    h->device = device;
    h->m = m; h->n = n; h->l = n + m + 1; h->l_global = h->l;
    h->nnz = colptr[n] - 1;
    FOS_HIP(hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking));
    FOS_HIP(hipHostMalloc((void**)&h->mark, sizeof(HostMark), hipHostMallocMapped));
    memset(h->mark, 0, sizeof(HostMark));
    h->speculate = !(getenv("FOS_SPECULATE") && atoi(getenv("FOS_SPECULATE")) == 0);
    h->shift_fuse = !(getenv("FOS_SHIFT_FUSE") && atoi(getenv("FOS_SHIFT_FUSE")) == 0);

    hipDeviceProp_t prop;
    FOS_HIP(hipGetDeviceProperties(&prop, device));
    const int cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    h->cus = cus;
    if (const char* e = getenv("FOS_PSD_WAVE")) h->psd_wave = atoi(e);
    if (const char* e = getenv("FOS_PSD_REFINE")) h->psd_refine = atoi(e);
    if (const char* e = getenv("FOS_PSD_EXTRAPOLATE")) h->psd_extrapolate = atoi(e) != 0;
    if (const char* e = getenv("FOS_PSD_THETA")) h->psd_theta = atof(e);
    h->psd_narrow = getenv("FOS_PSD_NARROW") != nullptr;
    h->psd_wide = getenv("FOS_PSD_WIDE") != nullptr;
    if (const char* e = getenv("FOS_PSD_THREADS")) h->psd_wide_threads = atoi(e);
    if (const char* e = getenv("FOS_SPMV_WG")) h->nwg_target = std::max(1, std::min(16384, atoi(e)));
    else h->nwg_target = cus * 12;      // ~1.7x the resident workgroups (7/CU): measured best for the KKT sweep (dynamic balance)

    // ---- operator
    HostBlkCsr& hs = h->hostS;
    h->row_sharded = (flags & FOS_CREATE_ROW_SHARDED) != 0;
    FOS_TRY(build_stacked_csr(m, n, colptr, rowval, nzval, h->nwg_target, &hs, cus * 28, -1, h->row_sharded, cus * 16));
    const bool windowed = !hs.wpanel.empty();
    if (!windowed && !getenv("FOS_SPMV_WG") && hs.ntiles > hs.nblk / 2) {
        // tile-dominated operator: the blocks are (nearly) equal work units, so give every wavefront the SAME whole number of them
        // -- one, unless that needs more than 16384 workgroups (C4: 16896 tiles over 12288 wavefronts leaves 3/8 of them with
        // twice the work: 77 us; 4224 workgroups: 71 us; tall tiles: 8704 blocks over 4096 wavefronts 73 us, one each 64)
        const int64_t waves_target = (int64_t)h->nwg_target * SPMV_WAVES;
        int64_t per_wave = std::max<int64_t>(1, (hs.nblk + waves_target / 2) / waves_target);
        while ((hs.nblk + SPMV_WAVES * per_wave - 1) / (SPMV_WAVES * per_wave) > 16384) ++per_wave;
        h->nwg_target = (int)((hs.nblk + SPMV_WAVES * per_wave - 1) / (SPMV_WAVES * per_wave));
        partition_workgroups(&hs, h->nwg_target);
    } else if (!windowed && !getenv("FOS_SPMV_WG") && hs.nblk / SPMV_WAVES < h->nwg_target) {
        // small operators: one row block per wavefront up to ONE resident round of workgroups (4 per CU); beyond that a
        // wavefront walks several blocks rather than queueing a second round behind the first (C3, 11 736 blocks: 2 934
        // workgroups 38.3 us per CG iteration, 1 024: 32.6, 1 152: 40.7)
        h->nwg_target = std::min(cus * 4, std::max(cus, (hs.nblk + SPMV_WAVES - 1) / SPMV_WAVES));
        partition_workgroups(&hs, h->nwg_target);
    }
    double* dval; int32_t* dcol; BlkDesc* dblk; uint16_t* drr; int32_t* dwv;
    FOS_TRY(dev_upload(h, &dval, hs.val));
    FOS_TRY(dev_upload(h, &dcol, hs.col));
    FOS_TRY(dev_upload(h, &dblk, hs.blk));
    FOS_TRY(dev_upload(h, &drr, hs.row_rel));
    // room for re-partitioning up to 16384 workgroups
    {
        std::vector<int32_t> wv(std::max<size_t>(hs.wave_blk0.size(), (size_t)SPMV_WAVES * 16384 + 16), 0);
        std::copy(hs.wave_blk0.begin(), hs.wave_blk0.end(), wv.begin());
        FOS_TRY(dev_upload(h, &dwv, wv));
    }
    BlkDesc* dwf = nullptr;
    {
        std::vector<BlkDesc> wf(std::max<size_t>(hs.wave_first.size(), (size_t)SPMV_WAVES * 16384 + 16), BlkDesc{});
        std::copy(hs.wave_first.begin(), hs.wave_first.end(), wf.begin());
        FOS_TRY(dev_upload(h, &dwf, wf));
    }
    if (windowed) {
        // window panels: persistent workgroups, one per CU (96 KB of LDS each), walking the panels (measured on C5: 116 us per
        // sweep with 256 workgroups, 122 us with one workgroup per panel)
        int nwg = std::min<int>((int)hs.wpanel.size(), cus * hs.wgeom.wg_per_cu);
        if (const char* e = getenv("FOS_SPMV_WG")) nwg = std::max(1, std::min<int>(atoi(e), (int)hs.wpanel.size()));
        // (a multiple of 8 lets xcd_remap give every XCD a contiguous run of panels -- but never at the price of a workgroup walking two
        //  panels while others walk one: with fewer panels than slots every panel gets its own workgroup, remapped or not)
        if (nwg >= 8 && (int)hs.wpanel.size() > nwg) nwg -= nwg % 8;
        hs.nwg = nwg;
    }
    h->S.npanel = (int32_t)hs.wpanel.size();
    h->S.win_tall = hs.wgeom.rows == WinTall::ROWS ? 1 : 0;
    h->S.wpanel = nullptr; h->S.wwave = nullptr; h->S.wdesc = nullptr; h->S.wval = nullptr; h->S.wcol = nullptr; h->S.wrow = nullptr;
    if (h->S.npanel > 0) {
        WinPanel* dp; WinWave* dsg; WinDesc* dsl; double* dwv2; uint16_t *dwc, *dwr;
        FOS_TRY(dev_upload(h, &dp, hs.wpanel));
        FOS_TRY(dev_upload(h, &dsg, hs.wwave));
        FOS_TRY(dev_upload(h, &dsl, hs.wdesc));
        FOS_TRY(dev_upload(h, &dwv2, hs.wval));
        FOS_TRY(dev_upload(h, &dwc, hs.wcol));
        FOS_TRY(dev_upload(h, &dwr, hs.wrow));
        h->S.wpanel = dp; h->S.wwave = dsg; h->S.wdesc = dsl; h->S.wval = dwv2; h->S.wcol = dwc; h->S.wrow = dwr;
        h->win_stats[0] = (int64_t)hs.wpanel.size(); h->win_stats[1] = (int64_t)hs.wdesc.size() / hs.wgeom.waves; h->win_stats[2] = hs.wnslice;
        h->win_stats[3] = (int64_t)hs.wval.size();
        std::vector<double>().swap(hs.wval);
        std::vector<uint16_t>().swap(hs.wcol);
        std::vector<uint16_t>().swap(hs.wrow);
    }
    h->S.nrows = hs.nrows; h->S.nnz = hs.nnz; h->S.nnz_padded = hs.nnz_padded;
    h->S.val = dval; h->S.col = dcol; h->S.blk = dblk;
    h->S.dbg_flags = getenv("FOS_DBG_FLAGS") ? atoi(getenv("FOS_DBG_FLAGS")) : 0;
    // an operator whose stored form (with the dozen vectors of the solver beside it) fits the XCDs' L2s / the Infinity Cache is
    // read with ordinary loads: it then stays on-die from one sweep to the next (FOS_RESIDENT=0/1 forces)
    {
        const double op_bytes = 8.0 * (double)hs.nnz_padded + 4.0 * (double)hs.ncol_stored + 48.0 * (double)hs.nblk;
        // (measured: C3, 12 MB, sweep 22.0 -> 17.1 us; the 64-block shard of C4, 34 MB = 4.3 MB per XCD against 4 MB of L2 each,
        //  24.5 -> 26.2 us per CG iteration: an operator that does not fit the L2s only evicts the vectors)
        //  (C3's stored form is 34.4 MB too -- what differs is the access: row blocks that GATHER (explicit column indices, latency
        //  bound) gain from a cache-resident matrix, a streamed dual-tile operator of that size does not)
        h->S.resident = (!windowed && hs.ntiles == 0 && op_bytes <= 48e6) ? 1 : 0;
        if (const char* e = getenv("FOS_RESIDENT")) h->S.resident = atoi(e) != 0 ? 1 : 0;
    }
    h->S.row_rel = drr; h->S.wave_blk0 = dwv; h->S.wave_first = dwf; h->S.nblk = hs.nblk; h->S.nwg = hs.nwg; h->S.nwaves = hs.nwaves;
    // dual tiles: partial-sum slots and the deferred rows' slot lists
    h->S.slots = nullptr; h->S.slots_rd = nullptr; h->S.row_defer = nullptr; h->S.def_rows = nullptr; h->S.def_ptr = nullptr; h->S.def_idx = nullptr;
    h->S.def_rec = nullptr;
    h->S.ndef = (int32_t)hs.def_rows.size();
    h->S.nwg_def = 0; h->S.def_lpr = 1;
    if (h->S.ndef > 0) {
        int32_t *drd, *ddr, *ddp, *ddi;
        double* dsl;
        FOS_TRY(dev_upload(h, &drd, hs.row_defer));
        FOS_TRY(dev_upload(h, &ddr, hs.def_rows));
        FOS_TRY(dev_upload(h, &ddp, hs.def_ptr));
        FOS_TRY(dev_upload(h, &ddi, hs.def_idx));
        DefRow* ddrec = nullptr;
        double avg_list = (double)hs.def_ptr.back() / (double)h->S.ndef;
        const bool rs_tiles = h->row_sharded && hs.ntiles > 0;
        if (!rs_tiles) {
            FOS_TRY(dev_upload(h, &ddrec, hs.def_rec));
            h->S.def_rec = ddrec;
            FOS_TRY(dev_alloc(h, &dsl, (size_t)2 * hs.nslots));
            FOS_HIP(hipMemset(dsl, 0, sizeof(double) * 2 * hs.nslots));
            h->S.slots = dsl; h->S.slots_rd = dsl; h->S.row_defer = drd; h->S.def_rows = ddr; h->S.def_ptr = ddp; h->S.def_idx = ddi;
            if (h->row_sharded) {                      // the all-reduced copy the slot-list sums read
                FOS_TRY(dev_alloc(h, &h->slots_rd, (size_t)2 * hs.nslots));
                FOS_HIP(hipMemset(h->slots_rd, 0, sizeof(double) * 2 * hs.nslots));
                h->S.slots_rd = h->slots_rd;
            }
        } else {
            // Two views of the builder's lists (fos_solver::cmp_rec).  The builder lists the n rows of A' first (rows ascending).
            const size_t nn = (size_t)h->n;
            if (hs.def_rows.size() < nn || hs.def_rows[nn - 1] != (int32_t)(nn - 1)) { set_error("internal: row-sharded tile operator without all rows of A' deferred"); return FOS_EINVAL; }
            // (1) what slots_compact_kernel adds up: the records of rows 0..n-1 as built, over the sweep's slot array
            std::vector<DefRow> crec(hs.def_rec.begin(), hs.def_rec.begin() + nn);
            FOS_TRY(dev_upload(h, &h->cmp_rec, crec));
            h->cmp_idx = ddi;
            double lsum = 0.0;
            for (size_t q = 0; q < nn; ++q) lsum += (double)(hs.def_ptr[q + 1] - hs.def_ptr[q]);
            int clpr = 1;
            while (clpr < 64 && 4.0 * clpr < lsum / (double)nn) clpr <<= 1;
            h->cmp_lpr = clpr;
            FOS_TRY(dev_alloc(h, &h->cmp_local, 2 * nn));
            FOS_HIP(hipMemset(h->cmp_local, 0, sizeof(double) * 2 * nn));
            // (2) what the consumers read, over slots_rd = [n summed rows | the sweep's slot array]: row j < n -> the single slot j;
            //     rows of A spread over column chunks -> their local lists, shifted by n
            std::vector<DefRow> vrec(hs.def_rec);
            std::vector<int32_t> vidx(hs.def_idx.size() + nn);
            for (size_t k = 0; k < hs.def_idx.size(); ++k) vidx[k] = hs.def_idx[k] + (int32_t)nn;
            for (size_t j = 0; j < nn; ++j) vidx[hs.def_idx.size() + j] = (int32_t)j;
            for (size_t q = 0; q < vrec.size(); ++q) {
                DefRow& d = vrec[q];
                if (q < nn) { d.own = -1; d.count = 1; d.base = (int32_t)q; d.stride = getenv("FOS_DEF_EXPLICIT") ? DEF_EXPLICIT : 0; d.kidx = (int32_t)(hs.def_idx.size() + q); }
                else { if (d.own >= 0) d.own += (int32_t)nn; d.base += (int32_t)nn; }
            }
            FOS_TRY(dev_upload(h, &ddrec, vrec));
            h->S.def_rec = ddrec;
            int32_t* dvi = nullptr;
            FOS_TRY(dev_upload(h, &dvi, vidx));
            FOS_TRY(dev_alloc(h, &h->slots_rd, 2 * (nn + (size_t)hs.nslots)));
            FOS_HIP(hipMemset(h->slots_rd, 0, sizeof(double) * 2 * (nn + (size_t)hs.nslots)));
            h->S.slots_rd = h->slots_rd;
            h->S.slots = h->slots_rd + 2 * nn;
            h->S.row_defer = drd; h->S.def_rows = ddr; h->S.def_ptr = ddp; h->S.def_idx = dvi;
            // consumers: one slot per row of A', the lists of the rows of A
            avg_list = ((double)nn + ((double)hs.def_ptr.back() - lsum)) / (double)h->S.ndef;
        }
        // lanes per deferred row: about a quarter of the average slot-list length (C4: 33 slots -> 8 lanes, dense LP: 80 -> 16)
        int lpr = 1;
        while (lpr < 64 && 4.0 * lpr < avg_list) lpr <<= 1;
        if (getenv("FOS_DEF_LPR")) lpr = std::max(1, std::min(64, atoi(getenv("FOS_DEF_LPR"))));
        while (lpr & (lpr - 1)) lpr &= lpr - 1;
        h->S.def_lpr = lpr;
        h->S.nwg_def = (int32_t)std::min<int64_t>(DEF_MAX_WG, ((int64_t)h->S.ndef * lpr + DEF_THREADS - 1) / DEF_THREADS);
        // bit mask of the rows finished from partial slots (cg_update_kernel skips them in its streaming pass)
        std::vector<uint32_t> mask((size_t)(h->l + 31) / 32 + 1, 0u);
        for (int32_t r : hs.def_rows) mask[(size_t)r >> 5] |= 1u << (r & 31);
        FOS_TRY(dev_upload(h, &h->def_mask, mask));
        std::vector<int32_t>().swap(hs.row_defer);
        std::vector<int32_t>().swap(hs.def_idx);
    }
    // The p update of CG can ride on the next sweep (two launches per iteration, launch_kkt2_cg): built and measured -- on
    // MI355X it is SLOWER than the separate launch at every size tried (C4: sweep 59 -> 86 us against a 10 us p update kernel;
    // 64-block shard: 13 -> 20 us against 5 us), so it stays an option (fos_set_tuning / FOS_CG_FUSE_P)
    h->fuse_p = false;
    if (const char* e = getenv("FOS_CG_FUSE_P")) h->fuse_p = atoi(e) != 0;
    if (const char* e = getenv("FOS_CG_VARIANT")) h->cg_variant = std::max(-1, std::min((int)FOS_CG_RESIDENT, atoi(e)));
    h->S.npart = h->S.nwg_def > 0 ? h->S.nwg_def : h->S.nwg;
    h->S.part_off = h->S.nwg_def > 0 ? h->S.nwg : 0;
    // free the big host arrays (keep block table for re-partitioning)
    std::vector<double>().swap(hs.val);
    std::vector<int32_t>().swap(hs.col);
    std::vector<uint16_t>().swap(hs.row_rel);

    std::vector<double> cbv((size_t)(n + m));
    double nb2 = 0, nc2 = 0;
    for (int64_t j = 0; j < n; ++j) { cbv[j] = c[j]; }
    for (int64_t i = 0; i < m; ++i) { cbv[n + i] = b[i]; }
    // norm(b), norm(c): plain sums of squares (values are O(1)-scaled problem data)
    for (int64_t j = 0; j < n; ++j) nc2 += c[j] * c[j];
    for (int64_t i = 0; i < m; ++i) nb2 += b[i] * b[i];
    h->nb = std::sqrt(nb2); h->nc = std::sqrt(nc2);
    h->nb_local = h->nb; h->nc_local = h->nc;
    FOS_TRY(dev_upload(h, &h->cb, cbv));

    // ---- vectors
    const size_t l = (size_t)h->l;
    d2** vecs[] = {&h->X, &h->T1, &h->T2, &h->SOL, &h->RHS, &h->R, &h->PB[0], &h->PB[1], &h->AP, &h->Y, &h->XOLD, &h->W, &h->SOL2};
    for (d2** v : vecs) {
        FOS_TRY(dev_alloc(h, v, l));
        FOS_HIP(hipMemset(*v, 0, sizeof(d2) * l));
    }
    FOS_TRY(dev_alloc(h, &h->plain, 2 * l));
    h->vec_blocks = (int)std::max<int64_t>(1, std::min<int64_t>((h->l + 255) / 256, 1024));
    h->cg_blocks = std::min(h->vec_blocks, getenv("FOS_CG_BLOCKS") ? std::max(1, atoi(getenv("FOS_CG_BLOCKS"))) : 2 * cus);
    // the x,r update kernel also adds the slot lists of the rows spread over dual tiles (`def_lpr` lanes per row): its grid must
    // cover them in ONE pass even when the vectors are short (dense LP C2: 15 000 such rows x 16 lanes against l = 15 001 --
    // sized by l alone the kernel took 105 us instead of 17)
    if (h->S.ndef > 0)
        h->cg_blocks = std::max<int>(h->cg_blocks, (int)std::min<int64_t>(1024, ((int64_t)h->S.ndef * h->S.def_lpr + 255) / 256));
    h->cg_blocks = std::min(h->cg_blocks, 1024);         // (cg_close_iteration adds at most 1024 r.r records per wavefront)

    // ---- cones
    std::vector<uint8_t> ew(l, 0);
    std::vector<ConeDesc> soc, psd, expc;
    add_cones(0, false, nK2, K2type, K2start, K2len, ew, soc, psd, expc);
    add_cones(n, true, nK1, K1type, K1start, K1len, ew, soc, psd, expc);
    h->nexp = (int)expc.size();
    FOS_TRY(dev_upload(h, &h->expc, expc));
    ew[l - 1] = (uint8_t)(EW_MAX0 | (EW_MAX0 << 2));            // tau, kappa  cones.jl:138,141
    FOS_TRY(dev_upload(h, &h->ew_op, ew));
    h->nsoc = (int)soc.size();
    FOS_TRY(psd_sign_setup(psd, &h->psd_big));            // (orders > 64 leave the list: projected by matrix products)
    h->npsd = (int)psd.size();
    FOS_TRY(dev_upload(h, &h->soc, soc));
    for (auto& cd : psd) h->psd_kmax = std::max(h->psd_kmax, cd.k);
    h->psd_kmin = h->psd_kmax;
    for (auto& cd : psd) h->psd_kmin = std::min(h->psd_kmin, cd.k);
    FOS_TRY(dev_upload(h, &h->psd, psd));
    size_t sb = psd_scratch_bytes(h->psd_kmax, h->npsd);
    if (sb) FOS_TRY(dev_alloc(h, &h->psd_scratch, sb / sizeof(double)));
    if (!getenv("FOS_PSD_COLD")) {
        const size_t vb = psd_basis_doubles(h->psd_kmax, h->npsd);
        if (vb) { FOS_TRY(dev_alloc(h, &h->psd_V[0], vb)); FOS_TRY(dev_alloc(h, &h->psd_V[1], vb)); }
        if (vb && h->psd_kmin == 64 && h->psd_kmax == 64) {
            FOS_TRY(dev_alloc(h, &h->psd_redo, (size_t)8 * h->npsd + 16));     // code [2 npsd] + 64-bit column mask [2 npsd] + a diagnostics switch
            FOS_HIP(hipMemset(h->psd_redo, 0, sizeof(int32_t) * (8 * h->npsd + 16)));
            if (getenv("FOS_PSD_DEBUG_REC")) { const int32_t v = 77; FOS_HIP(hipMemcpy(h->psd_redo + 8 * h->npsd, &v, sizeof(v), hipMemcpyHostToDevice)); }
        }
    }

    // ---- scalars
    FOS_TRY(dev_alloc(h, &h->st, 1));
    FOS_HIP(hipMemset(h->st, 0, sizeof(DevState)));
    FOS_TRY(dev_alloc(h, &h->pre_sums, 128));           // 16 producers x 3 sums x 2 self-validating words (sweep_sums3)
    FOS_HIP(hipMemset(h->pre_sums, 0, sizeof(double) * 128));
    h->pre_on = !(getenv("FOS_CG_PRE") && atoi(getenv("FOS_CG_PRE")) == 0);
    {
        void* dp = nullptr;
        FOS_HIP(hipHostGetDevicePointer(&dp, h->mark, 0));
        const unsigned long long addr = (unsigned long long)(uintptr_t)dp;
        FOS_HIP(hipMemcpy(&h->st->hostmark, &addr, sizeof(addr), hipMemcpyHostToDevice));
    }
    FOS_HIP(hipHostMalloc((void**)&h->st_host, sizeof(DevState), hipHostMallocDefault));
    memset(h->st_host, 0, sizeof(DevState));
    FOS_TRY(dev_alloc(h, &h->partials, (size_t)6 * PART_CAP));
    FOS_TRY(dev_alloc(h, &h->reduced, 16));
    FOS_HIP(hipMemset(h->reduced, 0, sizeof(double) * 16));
    if (const char* e = getenv("FOS_CG_CHUNK")) h->cg_chunk = std::max(1, atoi(e));

    FOS_TRY(resident_setup(h, cus));                      // FOS_CG_RESIDENT: does the operator qualify, and the plan if so
    FOS_TRY(fos_set_alg(h, FOS_ALG_GAP, 0.8, 1.8, 1.8, 0.0));
    FOS_TRY(fos_set_iterate(h, nullptr));
    FOS_HIP(hipStreamSynchronize(h->stream));
    *out = h;
    guard.h = nullptr;
    return FOS_OK;
}

int fos_destroy(fos_handle h) {
    if (!h) return FOS_OK;
    (void)hipSetDevice(h->device);
    if (h->stream) (void)hipStreamSynchronize(h->stream);
    if (h->comm && g_rccl.CommDestroy) g_rccl.CommDestroy(h->comm);
    if (h->host_buf) (void)hipHostFree(h->host_buf);
    for (void* q : h->peer_opened) (void)hipIpcCloseMemHandle(q);
    for (void* q : h->vec_opened) (void)hipIpcCloseMemHandle(q);
    if (h->host_seg) {
        (void)hipHostUnregister(h->host_seg);
        (void)munmap(h->host_seg, h->host_seg_bytes);
        if (h->host_seg_fd >= 0) close(h->host_seg_fd);
        if (h->rank == 0 && !h->host_seg_name.empty()) (void)shm_unlink(h->host_seg_name.c_str());
    }
    for (auto& r : h->prof_recs) { (void)hipEventDestroy(r.a); (void)hipEventDestroy(r.b); }
    psd_sign_destroy(h->psd_big);
    for (void* p : h->pooled) uncached_release(p);
    for (void* p : h->owned) (void)hipFree(p);
    if (h->st_host) (void)hipHostFree(h->st_host);
    if (h->mark) (void)hipHostFree(h->mark);
    if (h->stream) (void)hipStreamDestroy(h->stream);
    delete h;
    return FOS_OK;
}

int fos_sizes(fos_handle h, int64_t* m, int64_t* n, int64_t* N, int64_t* nnz) {
    if (!h) { set_error("NULL handle"); return FOS_EINVAL; }
    if (m) *m = h->m;
    if (n) *n = h->n;
    if (N) *N = 2 * h->l;
    if (nnz) *nnz = h->nnz;
    return FOS_OK;
}

int fos_comm_get_unique_id(void* id128) {
    FOS_TRY(rccl_load());
    static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId is 128 bytes");
    ncclUniqueId id;
    FOS_NCCL(g_rccl.GetUniqueId(&id));
    memcpy(id128, &id, sizeof(id));
    return FOS_OK;
}

int fos_comm_init(fos_handle h, int nranks, int rank, const void* id128) {
    if (!h || nranks < 1 || rank < 0 || rank >= nranks) { set_error("bad comm arguments"); return FOS_EINVAL; }
    if (h->ls_interval > 0 || h->gapp_iproj > 0) { set_error("switch the LineSearchWrapper / GAPP off before sharding the handle (fos_set_linesearch(h, 0), fos_set_gapp(h, 0))"); return FOS_EUNSUPPORTED; }
    FOS_TRY(rccl_load());
    FOS_HIP(hipSetDevice(h->device));
    ncclUniqueId id;
    memcpy(&id, id128, sizeof(id));
    FOS_NCCL(g_rccl.CommInitRank(&h->comm, nranks, id, rank));
    h->nranks = nranks; h->rank = rank;
    return global_setup(h);            // all-reduce [n+m, ||b||^2, ||c||^2]
}

// Sharding with the CALLER's collective (MPI.jl's Allreduce!, torch.distributed on any backend, ...): every cross-rank sum is
// staged through a pinned host buffer and handed to `fn` (in place, blocking).  Correct for both shardings, slow (one stream
// synchronisation per sum): the path for a host that owns no RCCL communicator, and the one a single-GPU box can run with two
// processes (tests/test_gpu_peer_mailbox.py::test_row_sharded_two_processes_host_exchange).
int fos_comm_init_host(fos_handle h, int nranks, int rank, fos_allreduce_fn fn, void* user) {
    if (!h || !fn || nranks < 1 || rank < 0 || rank >= nranks) { set_error("bad comm arguments"); return FOS_EINVAL; }
    if (h->ls_interval > 0 || h->gapp_iproj > 0) { set_error("switch the LineSearchWrapper / GAPP off before sharding the handle (fos_set_linesearch(h, 0), fos_set_gapp(h, 0))"); return FOS_EUNSUPPORTED; }
    if (h->comm || h->peer_on) { set_error("this handle already has a communicator"); return FOS_EINVAL; }
    FOS_HIP(hipSetDevice(h->device));
    if (!h->host_buf) {
        void* q = nullptr;
        FOS_HIP(hipHostMalloc(&q, sizeof(double) * std::max<size_t>((size_t)2 * (size_t)h->n, 16), hipHostMallocDefault));
        h->host_buf = static_cast<double*>(q);
    }
    h->host_fn = fn; h->host_user = user;
    h->nranks = nranks; h->rank = rank;
    return global_setup(h);
}

// ---- peer mailboxes: the sharded scalar sums without a collective library (fos_internal.hpp, PeerBox)
int fos_peer_export(fos_handle h, void* handle64) {
    if (!h || !handle64) { set_error("NULL argument"); return FOS_EINVAL; }
    static_assert(sizeof(hipIpcMemHandle_t) == 64, "hipIpcMemHandle_t is 64 bytes");
    FOS_HIP(hipSetDevice(h->device));
    if (!h->peer_mbox) {
        void* q = nullptr;
        const size_t bytes = PEER_BOX_TOTAL_WORDS * sizeof(unsigned long long);    // region 0 + region 1 (four slots)
        FOS_TRY(uncached_acquire(h->device, bytes, false, &q, "mailbox"));
        h->pooled.push_back(q);
        FOS_HIP(hipMemset(q, 0, bytes));                         // sequence number 0 is never sent
        h->peer_mbox = reinterpret_cast<unsigned long long*>(q);
    }
    hipIpcMemHandle_t ipc;
    FOS_HIP(hipIpcGetMemHandle(&ipc, h->peer_mbox));
    memcpy(handle64, &ipc, sizeof(ipc));
    return FOS_OK;
}

int fos_peer_open(fos_handle h, int nranks, int rank, const void* handles, double timeout_s) {
    if (!h || !handles || nranks < 1 || nranks > PEER_MAX_RANKS || rank < 0 || rank >= nranks) {
        set_error("bad peer arguments (1 <= nranks <= %d)", PEER_MAX_RANKS); return FOS_EINVAL;
    }
    if (!h->peer_mbox) { set_error("fos_peer_open before fos_peer_export"); return FOS_EINVAL; }
    if (!h->peer_opened.empty() || h->peer.box) { set_error("peer mailboxes are already open"); return FOS_EINVAL; }
    if (h->comm && (h->nranks != nranks || h->rank != rank)) { set_error("peer ranks differ from the RCCL communicator's"); return FOS_EINVAL; }
    FOS_HIP(hipSetDevice(h->device));
    std::vector<unsigned long long*> tab((size_t)nranks, nullptr);
    for (int r = 0; r < nranks; ++r) {
        if (r == rank) { tab[r] = h->peer_mbox; continue; }
        hipIpcMemHandle_t ipc;
        memcpy(&ipc, (const char*)handles + (size_t)r * sizeof(ipc), sizeof(ipc));
        void* q = nullptr;
        hipError_t e = hipIpcOpenMemHandle(&q, ipc, hipIpcMemLazyEnablePeerAccess);
        if (e != hipSuccess) {
            (void)hipGetLastError();
            for (void* o : h->peer_opened) (void)hipIpcCloseMemHandle(o);
            h->peer_opened.clear();
            set_error("hipIpcOpenMemHandle(mailbox of rank %d): %s", r, hipGetErrorString(e));
            return FOS_ECOMM;
        }
        h->peer_opened.push_back(q);
        tab[r] = reinterpret_cast<unsigned long long*>(q);
        // first contact between DIFFERENT devices: the mapping can succeed where loads and stores over the link cannot -- ask the
        // runtime, and say which pair it is (the caller falls back to its collective: bench.py `peer_fallback_reason`)
        hipPointerAttribute_t attr;
        if (hipPointerGetAttributes(&attr, q) == hipSuccess && attr.device == h->device) h->peer_same_device = true;
        if (hipPointerGetAttributes(&attr, q) == hipSuccess && attr.device >= 0 && attr.device != h->device) {
            int can = 0;
            if (hipDeviceCanAccessPeer(&can, h->device, attr.device) == hipSuccess && !can) {
                for (void* o : h->peer_opened) (void)hipIpcCloseMemHandle(o);
                h->peer_opened.clear();
                set_error("device %d cannot access the memory of device %d, rank %d's (hipDeviceCanAccessPeer = 0)", h->device, attr.device, r);
                return FOS_ECOMM;
            }
        } else (void)hipGetLastError();
    }
    unsigned long long** dtab = nullptr;
    FOS_TRY(dev_upload(h, &dtab, tab));
    uint32_t* seq = nullptr;
    FOS_TRY(dev_alloc(h, &seq, 1));
    FOS_HIP(hipMemset(seq, 0, sizeof(uint32_t)));
    h->peer = PeerBox{};
    h->peer.box = dtab; h->peer.seq = seq; h->peer.nranks = nranks; h->peer.rank = rank;
    h->peer.timeout_ticks = (int64_t)((timeout_s > 0 ? timeout_s : 20.0) * 1e8);
    h->peer.loopback = (getenv("FOS_PEER_LOOPBACK") && atoi(getenv("FOS_PEER_LOOPBACK")) != 0) ? 1 : 0;
    h->nranks = nranks; h->rank = rank;
    return FOS_OK;
}

// Host-pinned mailboxes: ONE shm segment of mailbox size that every rank maps and registers; every box[r] is that segment (PeerBox::shared),
// workgroup 0 of a folded exchange republishes the peers' words in a local relay (PeerBox::relay).
int fos_peer_open_host(fos_handle h, int nranks, int rank, const char* shm_name, double timeout_s) {
    if (!h || !shm_name || shm_name[0] != '/' || nranks < 1 || nranks > PEER_MAX_RANKS || rank < 0 || rank >= nranks) {
        set_error("bad arguments (shm_name \"/...\", 1 <= nranks <= %d)", PEER_MAX_RANKS); return FOS_EINVAL;
    }
    if (!h->peer_opened.empty() || h->peer.box || h->host_seg) { set_error("peer mailboxes are already open (fos_peer_close first)"); return FOS_EINVAL; }
    if (h->row_sharded) { set_error("host-pinned mailboxes carry the scalar sums of cone-sharded handles only"); return FOS_EUNSUPPORTED; }
    if (h->comm && (h->nranks != nranks || h->rank != rank)) { set_error("peer ranks differ from the RCCL communicator's"); return FOS_EINVAL; }
    FOS_HIP(hipSetDevice(h->device));
    // (one more page behind the mailbox words: every rank's device identity, so that ranks which share a device can find out -- fos_peer_selftest)
    const size_t bytes = ((PEER_BOX_TOTAL_WORDS * sizeof(unsigned long long) + 4095) / 4096) * 4096 + 4096;
    // Rank 0 CREATES the segment -- exclusively, after unlinking whatever a crashed run left under the name: a fresh segment is zero filled (sequence number 0
    // is never sent), a reused one would carry sequence-tagged words that validate themselves.  The other ranks open it WITHOUT creating, waiting for it to
    // appear at its full size.  A rank that was quick enough to open a stale segment before rank 0 unlinked it holds an unlinked file: fos_peer_selftest
    // (behind the caller's barrier) sees st_nlink == 0 on the descriptor kept here and fails the self test.
    int fd = -1;
    if (rank == 0) {
        (void)shm_unlink(shm_name);
        fd = shm_open(shm_name, O_CREAT | O_EXCL | O_RDWR, 0600);
        if (fd < 0 && errno == EEXIST) { (void)shm_unlink(shm_name); fd = shm_open(shm_name, O_CREAT | O_EXCL | O_RDWR, 0600); }
        if (fd < 0) { set_error("shm_open(%s, O_CREAT | O_EXCL): %s", shm_name, strerror(errno)); return FOS_ECOMM; }
        if (ftruncate(fd, (off_t)bytes) != 0) { const int e = errno; close(fd); (void)shm_unlink(shm_name); set_error("ftruncate(%s, %zu): %s", shm_name, bytes, strerror(e)); return FOS_ECOMM; }
    } else {
        const auto t0 = std::chrono::steady_clock::now();
        const double wait_s = timeout_s > 0 ? std::max(timeout_s, 5.0) : 20.0;
        for (;;) {
            fd = shm_open(shm_name, O_RDWR, 0600);
            if (fd >= 0) {
                struct stat sb;
                if (fstat(fd, &sb) == 0 && (size_t)sb.st_size >= bytes) break;
                close(fd); fd = -1;
            } else if (errno != ENOENT) { set_error("shm_open(%s): %s", shm_name, strerror(errno)); return FOS_ECOMM; }
            if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > wait_s) { set_error("shm_open(%s): rank 0 did not create the segment within %.0f s", shm_name, wait_s); return FOS_ECOMM; }
            usleep(1000);
        }
    }
    void* seg = mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    if (seg == MAP_FAILED) { const int e = errno; close(fd); if (rank == 0) (void)shm_unlink(shm_name); set_error("mmap(%s): %s", shm_name, strerror(e)); return FOS_ECOMM; }
    hipError_t e = hipHostRegister(seg, bytes, hipHostRegisterMapped | hipHostRegisterPortable);
    void* dptr = nullptr;
    if (e == hipSuccess) e = hipHostGetDevicePointer(&dptr, seg, 0);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        (void)hipHostUnregister(seg);
        (void)hipGetLastError();
        munmap(seg, bytes);
        close(fd);
        if (rank == 0) (void)shm_unlink(shm_name);
        set_error("hipHostRegister / hipHostGetDevicePointer(%s): %s", shm_name, hipGetErrorString(e));
        return FOS_ECOMM;
    }
    h->host_seg_fd = fd;
    h->host_seg = seg; h->host_seg_bytes = bytes; h->host_seg_name = shm_name;
    {
        char bus[64] = {0};
        unsigned long long id = 0x9E3779B97F4A7C15ull;
        if (hipDeviceGetPCIBusId(bus, (int)sizeof(bus), h->device) == hipSuccess) { for (const char* q = bus; *q; ++q) id = (id ^ (unsigned char)*q) * 0x100000001B3ull; }
        else { (void)hipGetLastError(); id ^= (unsigned long long)(h->device + 1); }
        volatile unsigned long long* ids = reinterpret_cast<volatile unsigned long long*>(static_cast<char*>(seg) + bytes - 4096);
        ids[rank] = id | 1ull;                                  // (never zero: a zero entry = that rank has not opened the segment yet)
    }
    if (!h->peer_relay) {
        void* q = nullptr;
        const size_t rb = PEER_BOX_TOTAL_WORDS * sizeof(unsigned long long);
        FOS_TRY(uncached_acquire(h->device, rb, false, &q, "relay"));
        h->pooled.push_back(q);
        h->peer_relay = reinterpret_cast<unsigned long long*>(q);
    }
    FOS_HIP(hipMemset(h->peer_relay, 0, PEER_BOX_TOTAL_WORDS * sizeof(unsigned long long)));
    std::vector<unsigned long long*> tab((size_t)nranks, reinterpret_cast<unsigned long long*>(dptr));
    unsigned long long** dtab = nullptr;
    FOS_TRY(dev_upload(h, &dtab, tab));
    uint32_t* seq = nullptr;
    FOS_TRY(dev_alloc(h, &seq, 1));
    FOS_HIP(hipMemset(seq, 0, sizeof(uint32_t)));
    h->peer = PeerBox{};
    h->peer.box = dtab; h->peer.seq = seq; h->peer.nranks = nranks; h->peer.rank = rank;
    h->peer.timeout_ticks = (int64_t)((timeout_s > 0 ? timeout_s : 20.0) * 1e8);
    h->peer.relay = h->peer_relay; h->peer.shared = 1;
    h->peer.loopback = (getenv("FOS_PEER_LOOPBACK") && atoi(getenv("FOS_PEER_LOOPBACK")) != 0) ? 1 : 0;
    h->nranks = nranks; h->rank = rank;
    return FOS_OK;
}

// drop the open mailboxes (device or host): another transport may be opened on the handle afterwards
int fos_peer_close(fos_handle h) {
    if (!h) { set_error("NULL handle"); return FOS_EINVAL; }
    FOS_HIP(hipSetDevice(h->device));
    FOS_HIP(hipStreamSynchronize(h->stream));
    if (h->peer_on) FOS_TRY(fos_peer_enable(h, 0));
    for (void* q : h->peer_opened) (void)hipIpcCloseMemHandle(q);
    h->peer_opened.clear();
    for (void* q : h->vec_opened) (void)hipIpcCloseMemHandle(q);
    h->vec_opened.clear();
    h->vec = VecBox{};
    if (h->host_seg) {
        (void)hipHostUnregister(h->host_seg);
        (void)munmap(h->host_seg, h->host_seg_bytes);
        if (h->host_seg_fd >= 0) { close(h->host_seg_fd); h->host_seg_fd = -1; }
        if (h->rank == 0 && !h->host_seg_name.empty()) (void)shm_unlink(h->host_seg_name.c_str());
        h->host_seg = nullptr; h->host_seg_bytes = 0; h->host_seg_name.clear();
    }
    if (h->peer_mbox) FOS_HIP(hipMemset(h->peer_mbox, 0, PEER_BOX_TOTAL_WORDS * sizeof(unsigned long long)));   // (a re-opened mailbox starts its sequence numbers again)
    h->peer = PeerBox{};                       // (the small device tables stay owned by the handle until fos_destroy)
    h->peer_same_device = false;
    // a failed exchange leaves its mark in the device state: clear it, the next transport starts clean
    DevState z;
    FOS_HIP(hipMemcpy(&z, h->st, sizeof(DevState), hipMemcpyDeviceToHost));
    z.xchg_failed = 0; z.done = 0;
    FOS_HIP(hipMemcpy(h->st, &z, sizeof(DevState), hipMemcpyHostToDevice));
    h->st_host->xchg_failed = 0;
    return FOS_OK;
}

// `rounds` exchanges of known values, checked exactly; *ok = 0 on a mismatch or a time-out (the handle then keeps
// whatever reduction it had: RCCL if fos_comm_init was called).  Collective: every rank calls it with the same rounds.
int fos_peer_selftest(fos_handle h, int rounds, int32_t* ok) {
    if (!h || !ok) { set_error("NULL argument"); return FOS_EINVAL; }
    if (!h->peer.box) { set_error("fos_peer_selftest before fos_peer_open"); return FOS_EINVAL; }
    FOS_HIP(hipSetDevice(h->device));
    if (h->host_seg && h->host_seg_fd >= 0) {
        // (the caller's barrier stands between the opens and this call: rank 0 has unlinked and re-created the segment by now -- a descriptor whose
        //  file is no longer linked is a segment some crashed run left behind, opened before rank 0 got to it)
        struct stat sb;
        if (fstat(h->host_seg_fd, &sb) != 0 || sb.st_nlink == 0) {
            set_error("rank %d mapped a stale shared-memory segment under %s (a crashed run's): the self test fails, use another transport or name", h->peer.rank, h->host_seg_name.c_str());
            *ok = 0;
            return FOS_OK;
        }
    }
    if (h->host_seg) {
        // host-pinned mailboxes: do ranks share THIS device (tests: several ranks on one GPU)?  Every rank left its device's identity behind the
        // mailbox words when it opened the segment, and the caller's barrier stands between the opens and this call.  A shared device keeps the
        // PSD refinement kernel to small batches (it needs whole CUs, which a peer's spinning CG kernel may hold: DESIGN 3)
        const volatile unsigned long long* ids = reinterpret_cast<const volatile unsigned long long*>(static_cast<const char*>(h->host_seg) + h->host_seg_bytes - 4096);
        for (int r = 0; r < h->peer.nranks; ++r)
            if (r != h->peer.rank && ids[r] != 0ull && ids[r] == ids[h->peer.rank]) h->peer_same_device = true;
    }
    const bool was_on = h->peer_on;
    h->peer_on = true;
    LaunchCtx c = h->ctx();
    h->peer_on = was_on;
    *ok = 1;
    auto val = [](int r, int k, int a) { return (a == 0) ? (double)(r + 1) * (k + 1) : (a == 1 ? 0.1 * (r + 1) + 1e-3 * k : -1.0 / (r + 1 + k)); };
    for (int k = 0; k < rounds && *ok; ++k) {
        const int nacc = (k % 3 == 0) ? 3 : (k % 3 == 1 ? 1 : 6);
        double loc[6], got[6];
        for (int a = 0; a < nacc; ++a) loc[a] = val(h->peer.rank, k, a % 3) + a;
        FOS_HIP(hipMemcpyAsync(h->partials, loc, sizeof(double) * nacc, hipMemcpyHostToDevice, h->stream));
        launch_reduce1(c, 1, nacc, 0);
        FOS_HIP(hipMemcpyAsync(got, h->reduced, sizeof(double) * nacc, hipMemcpyDeviceToHost, h->stream));
        FOS_HIP(hipMemcpyAsync(h->st_host, h->st, sizeof(DevState), hipMemcpyDeviceToHost, h->stream));
        FOS_HIP(hipStreamSynchronize(h->stream));
        if (h->st_host->xchg_failed) { *ok = 0; break; }
        for (int a = 0; a < nacc; ++a) {
            double s = 0.0;
            for (int r = 0; r < h->peer.nranks; ++r) s += val(r, k, a % 3) + a;
            if (s != got[a]) *ok = 0;
        }
    }
    if (h->st_host->xchg_failed) {              // leave the handle usable with its previous reduction
        DevState z;
        FOS_HIP(hipMemcpy(&z, h->st, sizeof(DevState), hipMemcpyDeviceToHost));
        z.xchg_failed = 0; z.done = 0;
        FOS_HIP(hipMemcpy(h->st, &z, sizeof(DevState), hipMemcpyHostToDevice));
        h->st_host->xchg_failed = 0;
    }
    return FOS_OK;
}

// What ONE exchange of four doubles costs on the transport this sharded handle uses -- `rounds` of them back to back, in stream, between two events:
// mailboxes: inside one launch (no launch or host round trip between them); RCCL: `rounds` ncclAllReduce calls.  Collective (every rank, same rounds).
int fos_exchange_bench(fos_handle h, int rounds, double* us_per_exchange) {
    if (!h || !us_per_exchange || rounds < 1) { set_error("bad argument"); return FOS_EINVAL; }
    *us_per_exchange = 0.0;
    if (!h->sharded() || h->host_fn) { set_error("fos_exchange_bench: the handle has no in-stream transport (mailboxes or RCCL)"); return FOS_EUNSUPPORTED; }
    FOS_HIP(hipSetDevice(h->device));
    LaunchCtx c = h->ctx();
    hipEvent_t e0, e1;
    FOS_HIP(hipEventCreate(&e0));
    FOS_HIP(hipEventCreate(&e1));
    auto run = [&](int n) -> int {
        if (h->peer_on) { launch_peer_chain(c, n); return FOS_OK; }
        for (int r = 0; r < n; ++r) FOS_TRY(allreduce(h, 4));
        return FOS_OK;
    };
    int rc = run(4);                                     // warm
    if (rc == FOS_OK) rc = hipEventRecord(e0, h->stream) == hipSuccess ? FOS_OK : FOS_EHIP;
    if (rc == FOS_OK) rc = run(rounds);
    if (rc == FOS_OK) rc = hipEventRecord(e1, h->stream) == hipSuccess ? FOS_OK : FOS_EHIP;
    if (rc == FOS_OK) rc = poll_state(h);
    float ms = 0.f;
    if (rc == FOS_OK && hipEventElapsedTime(&ms, e0, e1) != hipSuccess) rc = FOS_EHIP;
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    if (rc != FOS_OK) return rc;
    *us_per_exchange = 1e3 * (double)ms / (double)rounds;
    return FOS_OK;
}

// Row-sharded handles: the exchange buffer of the n-vector A'y (fos_internal.hpp, VecBox).  Protocol as for the mailboxes, after
// fos_peer_open (which fixes nranks): fos_peer_vec_export -> the host all-gathers the 64-byte handles -> fos_peer_vec_open.
int fos_peer_vec_export(fos_handle h, void* handle64) {
    if (!h || !handle64) { set_error("NULL argument"); return FOS_EINVAL; }
    if (!h->row_sharded) { set_error("fos_peer_vec_export: not a row-sharded handle"); return FOS_EINVAL; }
    if (!h->peer.box) { set_error("fos_peer_vec_export before fos_peer_open"); return FOS_EINVAL; }
    FOS_HIP(hipSetDevice(h->device));
    if (!h->vec_buf) {
        const int g = h->peer.nranks;
        const size_t doubles = (size_t)2 * g * 2 * (size_t)h->n;
        const size_t bytes = doubles * sizeof(double) + (size_t)4 * g * sizeof(uint32_t) + 64;       // flags: [stage 1 | stage 2][parity][rank]
        void* q = nullptr;
        FOS_TRY(uncached_acquire(h->device, bytes, false, &q, "vector exchange buffer"));
        h->pooled.push_back(q);
        FOS_HIP(hipMemset(q, 0, bytes));                         // exchange number 0 is never sent
        h->vec_buf = reinterpret_cast<double*>(q);
        h->vec_nranks = g;
    }
    hipIpcMemHandle_t ipc;
    FOS_HIP(hipIpcGetMemHandle(&ipc, h->vec_buf));
    memcpy(handle64, &ipc, sizeof(ipc));
    return FOS_OK;
}
int fos_peer_vec_open(fos_handle h, const void* handles) {
    if (!h || !handles) { set_error("NULL argument"); return FOS_EINVAL; }
    if (!h->vec_buf || !h->peer.box) { set_error("fos_peer_vec_open before fos_peer_vec_export"); return FOS_EINVAL; }
    if (!h->vec_opened.empty() || h->vec.buf) { set_error("the vector exchange buffers are already open"); return FOS_EINVAL; }
    FOS_HIP(hipSetDevice(h->device));
    const int g = h->peer.nranks, me = h->peer.rank;
    const size_t doubles = (size_t)2 * g * 2 * (size_t)h->n;
    std::vector<double*> bt((size_t)g, nullptr);
    std::vector<uint32_t*> ft((size_t)g, nullptr);
    for (int r = 0; r < g; ++r) {
        void* q = h->vec_buf;
        if (r != me) {
            hipIpcMemHandle_t ipc;
            memcpy(&ipc, (const char*)handles + (size_t)r * sizeof(ipc), sizeof(ipc));
            hipError_t e = hipIpcOpenMemHandle(&q, ipc, hipIpcMemLazyEnablePeerAccess);
            if (e != hipSuccess) {
                (void)hipGetLastError();
                for (void* o : h->vec_opened) (void)hipIpcCloseMemHandle(o);
                h->vec_opened.clear();
                set_error("hipIpcOpenMemHandle(vector exchange buffer of rank %d): %s", r, hipGetErrorString(e));
                return FOS_ECOMM;
            }
            h->vec_opened.push_back(q);
        }
        bt[r] = reinterpret_cast<double*>(q);
        ft[r] = reinterpret_cast<uint32_t*>(reinterpret_cast<double*>(q) + doubles);
    }
    double** dbt = nullptr; uint32_t** dft = nullptr; uint32_t* cnt = nullptr;
    FOS_TRY(dev_upload(h, &dbt, bt));
    FOS_TRY(dev_upload(h, &dft, ft));
    FOS_TRY(dev_alloc(h, &cnt, 1));
    FOS_HIP(hipMemset(cnt, 0, sizeof(uint32_t)));
    h->vec.buf = dbt; h->vec.flags = dft; h->vec.counter = cnt; h->vec.nranks = g; h->vec.rank = me;
    h->vec.n2 = 2 * h->n; h->vec.timeout_ticks = h->peer.timeout_ticks;
    return FOS_OK;
}

// switch the sharded sums to the peer mailboxes (collective: all ranks make the same choice after the self test)
int fos_peer_enable(fos_handle h, int32_t on) {
    if (!h) { set_error("NULL handle"); return FOS_EINVAL; }
    if (on && !h->peer.box) { set_error("fos_peer_enable before fos_peer_open"); return FOS_EINVAL; }
    if (on && h->row_sharded && !h->vec.buf) { set_error("row-sharded handle: fos_peer_vec_export / fos_peer_vec_open before fos_peer_enable"); return FOS_EINVAL; }
    if (on && (h->ls_interval > 0 || h->gapp_iproj > 0)) { set_error("switch the LineSearchWrapper / GAPP off before sharding the handle (fos_set_linesearch(h, 0), fos_set_gapp(h, 0))"); return FOS_EUNSUPPORTED; }
    FOS_HIP(hipSetDevice(h->device));
    FOS_HIP(hipStreamSynchronize(h->stream));
    h->peer_on = on != 0;
    if (h->sharded()) return global_setup(h);
    h->l_global = h->l; h->nb = h->nb_local; h->nc = h->nc_local;
    return resident_setup(h, h->cus);          // (the whole device is this handle's again)
}

int fos_set_alg(fos_handle h, int alg, double alpha, double alpha1, double alpha2, double beta) {
    if (!h) { set_error("NULL handle"); return FOS_EINVAL; }
    if (alg < FOS_ALG_GAP || alg > FOS_ALG_DYKSTRA) { set_error("unknown algorithm %d", alg); return FOS_EINVAL; }
    FOS_HIP(hipSetDevice(h->device));
    h->alg = alg; h->alpha = alpha; h->alpha1 = alpha1; h->alpha2 = alpha2; h->beta = beta;
    h->ls_interval = 0; h->ls_now = false;                              // a fresh algorithm is unwrapped (fos_set_linesearch follows)
    h->gapp_iproj = 0; h->gapp_now = false;                             // ... and plain (fos_set_gapp follows)
    h->lp.interval = 0; h->lp.savepos = 0; h->lp.now = false;     // ... (fos_set_longstep follows)
    h->fista_t = 1.0;                                                   // fista.jl:24
    // fresh *Data: alpha12 = 2.0 (gapa.jl:29); y = xold = 0 (fista.jl:24); p = q = 0 (dykstra.jl:21)
    DevState z;
    FOS_HIP(hipStreamSynchronize(h->stream));
    FOS_HIP(hipMemcpy(&z, h->st, sizeof(DevState), hipMemcpyDeviceToHost));
    z.alpha12 = 2.0;
    z.gapa_scl = 0.0;
    FOS_HIP(hipMemcpy(h->st, &z, sizeof(DevState), hipMemcpyHostToDevice));
    FOS_HIP(hipMemsetAsync(h->Y, 0, sizeof(d2) * h->l, h->stream));
    FOS_HIP(hipMemsetAsync(h->XOLD, 0, sizeof(d2) * h->l, h->stream));
    FOS_HIP(hipStreamSynchronize(h->stream));
    return FOS_OK;
}

// direct = true: build (I + Q Q')^-1 once.  A is handed over again (the handle keeps only its device format).
// G = I + Q Q' = I - Q Q is symmetric positive definite with lambda_min >= 1; its inverse is formed by the Newton-Schulz iteration
//     X_0 = I / (1.25 lambda~),   X_{k+1} = 2 X_k - X_k (G X_k),        lambda~ = a power-iteration estimate of lambda_max(G),
// whose residual I - G X_k squares every step: ceil(log2 lambda~) + 7 steps reach rounding level (verified at the end: the
// entries of G X - I).  Only matrix products are needed: the hand-written fp64 MFMA GEMM of vecops.hip.
int fos_enable_direct(fos_handle h, const int64_t* colptr, const int64_t* rowval, const double* nzval) {
    if (!h || !colptr || (!rowval && colptr[h->n] > 1)) { set_error("NULL argument"); return FOS_EINVAL; }
    if (h->row_sharded) { set_error("direct=true is not available on row-sharded handles"); return FOS_EUNSUPPORTED; }
    if (h->Ginv || (h->blk_ginv && h->blk_ready)) { h->direct = true; h->direct_blk = h->blk_ginv != nullptr && h->blk_ready; return FOS_OK; }
    const int64_t l = h->l, nnz = colptr[h->n] - 1;
    // which exact form (FOS_DIRECT_MODE=block|dense|cg forces one; default: the first that applies)
    const char* mode_env = getenv("FOS_DIRECT_MODE");
    const std::string mode = mode_env ? mode_env : "auto";
    if (nnz != h->nnz) { set_error("fos_enable_direct: A has %lld non-zeros, the handle was created with %lld", (long long)nnz, (long long)h->nnz); return FOS_EINVAL; }
    // (1) block-separable operators: I + A'A block diagonal with small blocks -> three sweeps per projection, any size (blkdir_setup)
    if (mode == "auto" || mode == "block") {
        FOS_HIP(hipSetDevice(h->device));
        bool ok = false;
        FOS_TRY(blkdir_setup(h, colptr, rowval, nzval, &ok));
        if (ok) { h->blk_ready = true; h->direct_blk = true; h->direct = true; return FOS_OK; }
        if (mode == "block") { set_error("FOS_DIRECT_MODE=block: A'A has a diagonal block of more than %d columns", BLKDIR_MAX); return FOS_EUNSUPPORTED; }
    }
    // cone-sharded handles: only the block form (its three scalar exchanges per projection go through the handle's transport); collective -- every rank
    // has taken part in blkdir_setup's vote above
    if (h->sharded()) { set_error("direct=true on a sharded handle needs the block form on every rank (I + A'A block diagonal with blocks of at most %d columns)", BLKDIR_MAX); return FOS_EUNSUPPORTED; }
    // beyond what a dense l x l inverse can hold, S1 = IndAffine([Q -I], 0) and S1 = AffinePlusLinear(Q, 0, 0, 1) are still the SAME set (HSDE.jl:12-15 / :22): the
    // exact projection is what the warm-started CG converges to, so "direct" becomes CG run to its tolerance floor l eps from the first call on
    // (no 0.2^sqrt(i) schedule: affinepluslinear.jl:108-112 is what direct = true switches off) -- the reference's sparse factorisation is not rebuilt.
    const int64_t dense_max = getenv("FOS_DIRECT_DENSE_MAX") ? atoll(getenv("FOS_DIRECT_DENSE_MAX")) : 46000;
    if (l > dense_max || mode == "cg") { h->direct_cg = true; h->direct = false; return FOS_OK; }
    FOS_HIP(hipSetDevice(h->device));
    LaunchCtx c = h->ctx();
    const int64_t L = (l + 63) / 64 * 64;
    const size_t L2 = (size_t)L * (size_t)L;
    int64_t *dcp = nullptr, *drv = nullptr;
    double *dnz = nullptr, *B0 = nullptr, *B1 = nullptr, *B2 = nullptr;       // B0: Q, later Y = G X;  B1, B2: X ping-pong
    auto cleanup = [&]() { (void)hipFree(dcp); (void)hipFree(drv); (void)hipFree(dnz); (void)hipFree(B0); (void)hipFree(B1); (void)hipFree(B2); };
    hipError_t e = hipMalloc((void**)&dcp, sizeof(int64_t) * (h->n + 1));
    if (e == hipSuccess) e = hipMalloc((void**)&drv, sizeof(int64_t) * std::max<int64_t>(nnz, 1));
    if (e == hipSuccess) e = hipMalloc((void**)&dnz, sizeof(double) * std::max<int64_t>(nnz, 1));
    if (e == hipSuccess) e = hipMalloc((void**)&B0, sizeof(double) * L2);
    if (e == hipSuccess) e = hipMalloc((void**)&B1, sizeof(double) * L2);
    if (e == hipSuccess) e = hipMalloc((void**)&B2, sizeof(double) * L2);
    if (e != hipSuccess) { cleanup(); set_error("direct=true: hipMalloc of the dense set-up buffers (4 x %zu bytes) failed: %s", L2 * 8, hipGetErrorString(e)); return FOS_ENOMEM; }
    double* G = nullptr;
    int rc = dev_alloc(h, &G, L2);
    if (rc == FOS_OK && !h->dvec[0]) rc = dev_alloc(h, &h->dvec[0], (size_t)L);
    if (rc == FOS_OK && !h->dvec[1]) rc = dev_alloc(h, &h->dvec[1], (size_t)L);
    if (rc != FOS_OK) { cleanup(); return rc; }
    auto fail = [&](int code) { cleanup(); return code; };
#define DIRECT_HIP(expr) do { hipError_t _e = (expr); if (_e != hipSuccess) { set_error("direct=true set-up: %s -> %s", #expr, hipGetErrorString(_e)); return fail(FOS_EHIP); } } while (0)
    DIRECT_HIP(hipMemcpyAsync(dcp, colptr, sizeof(int64_t) * (h->n + 1), hipMemcpyHostToDevice, h->stream));
    if (nnz) DIRECT_HIP(hipMemcpyAsync(drv, rowval, sizeof(int64_t) * nnz, hipMemcpyHostToDevice, h->stream));
    if (nnz) DIRECT_HIP(hipMemcpyAsync(dnz, nzval, sizeof(double) * nnz, hipMemcpyHostToDevice, h->stream));
    DIRECT_HIP(hipMemsetAsync(B0, 0, sizeof(double) * L2, h->stream));
    DIRECT_HIP(hipMemsetAsync(B1, 0, sizeof(double) * L2, h->stream));
    launch_dense_q_fill(c, dcp, drv, dnz, B0, L);                           // B0 = Q (zero padded to L x L)
    launch_dense_scale_identity(c, L, B1, 1.0);                             // B1 = I
    launch_dense_gemm(c, (int)L, -1.0, B0, B0, 1.0, B1, G);                 // G = I - Q Q  (padding rows/columns: identity)
    // ---- power iteration for lambda_max(G) (Rayleigh quotients from below; host-side norms of an l-vector)
    std::vector<double> v((size_t)L, 0.0), w((size_t)L, 0.0);
    for (int64_t i = 0; i < l; ++i) v[i] = 1.0 + 0.37 * std::sin(1.7 * (double)i);
    double lam = 1.0;
    for (int it = 0; it < 20; ++it) {
        double nv = 0.0;
        for (int64_t i = 0; i < l; ++i) nv += v[i] * v[i];
        nv = std::sqrt(nv);
        for (int64_t i = 0; i < l; ++i) v[i] /= nv;
        DIRECT_HIP(hipMemcpyAsync(h->dvec[0], v.data(), sizeof(double) * l, hipMemcpyHostToDevice, h->stream));
        launch_dense_symv(c, L, G, h->dvec[0], h->dvec[1]);
        DIRECT_HIP(hipMemcpyAsync(w.data(), h->dvec[1], sizeof(double) * l, hipMemcpyDeviceToHost, h->stream));
        DIRECT_HIP(hipStreamSynchronize(h->stream));
        double nw = 0.0;
        for (int64_t i = 0; i < l; ++i) nw += w[i] * w[i];
        nw = std::sqrt(nw);
        if (!(nw == nw) || nw > 1e300) { set_error("direct=true: the operator has non-finite entries"); return fail(FOS_EINVAL); }
        lam = std::max(lam, nw);
        v.swap(w);
    }
    // ---- Newton-Schulz
    const double x0 = 1.0 / (1.25 * lam);
    launch_dense_scale_identity(c, L, B1, x0);                              // X_0 = x0 I  (B1 was I: only its diagonal is non-zero)
    double *X = B1, *Xn = B2;
    const int planned = (int)std::ceil(std::log2(std::max(1.0, lam))) + 7;
    double resid = 1.0;
    int it = 0;
    std::vector<double> part(256);
    for (; it < planned + 6; ++it) {
        launch_dense_gemm(c, (int)L, 1.0, G, X, 0.0, nullptr, B0);          // Y = G X
        if (it >= planned) {                                                // converged?  max |Y - I|
            launch_dense_resid(c, L, B0, h->partials, 256);
            DIRECT_HIP(hipMemcpyAsync(part.data(), h->partials, sizeof(double) * 256, hipMemcpyDeviceToHost, h->stream));
            DIRECT_HIP(hipStreamSynchronize(h->stream));
            resid = 0.0;
            for (double r : part) resid = (r > resid || r != r) ? r : resid;
            if (resid <= 1e-12) break;
        }
        launch_dense_gemm(c, (int)L, -1.0, X, B0, 2.0, X, Xn);              // X <- 2 X - X Y
        std::swap(X, Xn);
    }
    if (!(resid <= 1e-12)) { set_error("direct=true: the inverse of I + Q Q' did not converge (max |G X - I| = %.3e after %d steps, lambda_max ~ %.3e)", resid, it, lam); return fail(FOS_EINVAL); }
    DIRECT_HIP(hipMemcpyAsync(G, X, sizeof(double) * L2, hipMemcpyDeviceToDevice, h->stream));     // keep the inverse in the handle's buffer
    DIRECT_HIP(hipStreamSynchronize(h->stream));
#undef DIRECT_HIP
    rc = check_launch("direct=true set-up");
    cleanup();
    if (rc != FOS_OK) return rc;
    h->Ginv = G; h->Gld = L; h->direct_iters = it;
    h->direct = true;
    return FOS_OK;
}

int fos_disable_direct(fos_handle h) {
    if (!h) { set_error("NULL handle"); return FOS_EINVAL; }
    h->direct = false; h->direct_cg = false; h->direct_blk = false;
    return FOS_OK;
}

// which form S1 = IndAffine([Q -I], 0) runs in: 0 = off (AffinePlusLinear's CG schedule), 1 = dense inverse, 2 = block form, 3 = CG at its tolerance floor
int fos_get_direct_mode(fos_handle h, int32_t* mode) {
    if (!h || !mode) { set_error("NULL argument"); return FOS_EINVAL; }
    *mode = h->direct ? (h->direct_blk ? 2 : 1) : (h->direct_cg ? 3 : 0);
    return FOS_OK;
}

int fos_reset_affine(fos_handle h) {
    if (!h) { set_error("NULL handle"); return FOS_EINVAL; }
    h->prox_i = 1; h->firstrun = true; h->cgiter = 0; h->hit_max_accum = 0; h->last_cg_pred = 0;
    h->firstrun2 = true;
    return FOS_OK;
}

int fos_set_iterate(fos_handle h, const double* z) {
    if (!h) { set_error("NULL handle"); return FOS_EINVAL; }
    FOS_HIP(hipSetDevice(h->device));
    if (z) return upload_plain(h, h->X, z);
    // HSDE_getinitialvalue: zeros, tau = kappa = 1     HSDE.jl:40-47
    FOS_HIP(hipMemsetAsync(h->X, 0, sizeof(d2) * h->l, h->stream));
    const d2 one = make_double2(1.0, 1.0);
    FOS_HIP(hipMemcpyAsync(h->X + (h->l - 1), &one, sizeof(d2), hipMemcpyHostToDevice, h->stream));
    FOS_HIP(hipStreamSynchronize(h->stream));
    return FOS_OK;
}

int fos_get_iterate(fos_handle h, double* z) {
    if (!h || !z) { set_error("NULL argument"); return FOS_EINVAL; }
    FOS_HIP(hipSetDevice(h->device));
    return download_plain(h, z, h->X);
}

int fos_get_checked(fos_handle h, double* z) {
    if (!h || !z) { set_error("NULL argument"); return FOS_EINVAL; }
    if (!h->last_checked) { set_error("no convergence check has run yet"); return FOS_EINVAL; }
    FOS_HIP(hipSetDevice(h->device));
    return download_plain(h, z, h->last_checked);
}

int fos_step(fos_handle h, int64_t i_first, int64_t count, int64_t checki, double eps,
             int64_t* iters_done, int32_t* checked, fos_check_result* res) {
    if (!h || checki < 1 || count < 0) { set_error("bad fos_step arguments"); return FOS_EINVAL; }
    FOS_HIP(hipSetDevice(h->device));
    if (checked) *checked = 0;
    int64_t done = 0;
    h->shift_ready = false;
    struct InStep { fos_solver* h; ~InStep() { h->in_step = false; h->shift_ready = false; } } in_step{h};
    h->in_step = true;
    for (int64_t i = i_first; done < count; ++i) {
        const d2* check_on = nullptr;
        const bool do_check = (i % checki) == 0;                         // HSDEStatus.jl:28
        bool finish_done = false;
        h->prof_step_on = h->prof && (h->prof_steps++ % h->prof_period) == 0;
        if (h->prof_step_on) h->prof_steps_sampled += 1;
        FOS_TRY(step_once(h, i, &check_on, do_check, &finish_done));
        fos_check_result r;
        if (do_check) {
            FOS_TRY(status_check(h, check_on, eps, &r));
            h->last_checked = check_on;
        }
        if (!finish_done) { const int po = prof_begin_other(h, 0); FOS_TRY(step_finish(h, i)); prof_end(h, po); }
        h->prof_step_on = false;
        ++done;
        if (do_check) {
            if (res) *res = r;
            if (checked) *checked = 1;
            break;
        }
    }
    if (iters_done) *iters_done = done;
    FOS_TRY(check_launch("fos_step"));
    FOS_HIP(hipStreamSynchronize(h->stream));
    return FOS_OK;
}

// LineSearchWrapper(alg; lsinterval) around GAP (AP, DR) or GAPA: iterations i with i % lsinterval == 0 become a 31-point search of
// the step length along S2(S1(x)) - x; 0 switches it off.  fos_linesearch_log: what the reference prints during the last search.
int fos_set_linesearch(fos_handle h, int64_t lsinterval) {
    if (!h || lsinterval < 0) { set_error("bad argument"); return FOS_EINVAL; }
    if (lsinterval > 0 && h->alg != FOS_ALG_GAP && h->alg != FOS_ALG_GAPA) {
        set_error("this algorithm does not support line search (support_linesearch: GAP and GAPA only, solvers/defaults.jl:22)");
        return FOS_EUNSUPPORTED;
    }
    if (lsinterval > 0 && (h->sharded() || h->row_sharded)) {
        // normres / normdiff (linesearch.jl:50,62) are GLOBAL norms; the search below adds this rank's partial sums only
        set_error("LineSearchWrapper is built for single-GPU handles only (its step-length scores are global norms)");
        return FOS_EUNSUPPORTED;
    }
    if (lsinterval > 0 && h->lp.interval > 0) { set_error("LineSearchWrapper inside a LongstepWrapper is not supported"); return FOS_EUNSUPPORTED; }
    h->ls_interval = lsinterval;
    return FOS_OK;
}
// GAPP(alpha, alpha1, alpha2; iproj) (solvers/gapproj.jl, README solver table): fos_set_alg(FOS_ALG_GAP, alpha, alpha1, alpha2) and then
// fos_set_gapp(iproj > 0): iterations i with i % iproj == 0 search 21 step lengths 2^k along P_S1(P_S2(P_S1 x)) - P_S1 x.
int fos_set_gapp(fos_handle h, int64_t iproj) {
    if (!h || iproj < 0) { set_error("bad argument"); return FOS_EINVAL; }
    if (iproj > 0 && h->alg != FOS_ALG_GAP) { set_error("GAPP is GAP with a projected search: fos_set_alg(FOS_ALG_GAP, ...) first"); return FOS_EUNSUPPORTED; }
    if (iproj > 0 && (h->sharded() || h->row_sharded)) { set_error("GAPP is built for single-GPU handles only (its test norms are global norms)"); return FOS_EUNSUPPORTED; }
    if (iproj > 0 && h->ls_interval > 0) { set_error("GAPP and the LineSearchWrapper exclude each other"); return FOS_EUNSUPPORTED; }
    if (iproj > 0 && h->lp.interval > 0) { set_error("GAPP inside a LongstepWrapper is not supported (fos_set_longstep(h, 0, 0) first)"); return FOS_EUNSUPPORTED; }
    h->gapp_iproj = iproj;
    return FOS_OK;
}
int fos_gapp_log(fos_handle h, double* out23) {
    if (!h || !out23) { set_error("NULL argument"); return FOS_EINVAL; }
    memcpy(out23, h->gapp_log, sizeof(h->gapp_log));
    return FOS_OK;
}
// LongstepWrapper(alg; longinterval, nsave) around the algorithm set last (wrappers/longstep.jl:22-40): 0 switches it off
int fos_set_longstep(fos_handle h, int64_t longinterval, int64_t nsave) {
    if (!h || longinterval < 0 || nsave < 0) { set_error("bad argument"); return FOS_EINVAL; }
    if (longinterval == 0) { h->lp.interval = 0; h->lp.savepos = 0; return FOS_OK; }
    if (2 * (nsave + 1) > LONG_KMAX_ROWS) { set_error("LongstepWrapper: nsave <= %d (2 (nsave + 1) saved planes, small dual QP solved by enumeration)", LONG_KMAX_ROWS / 2 - 1); return FOS_EUNSUPPORTED; }
    if (longinterval < nsave + 1) { set_error("LongstepWrapper: longinterval must be at least nsave + 1 (every plane is written before it is read)"); return FOS_EINVAL; }
    if (h->sharded() || h->row_sharded) { set_error("LongstepWrapper is built for single-GPU handles only (the planes' products are global sums)"); return FOS_EUNSUPPORTED; }
    if (h->ls_interval > 0 || h->gapp_iproj > 0) { set_error("LongstepWrapper around a LineSearchWrapper / GAPP is not supported"); return FOS_EUNSUPPORTED; }
    FOS_HIP(hipSetDevice(h->device));
    const int64_t K = 2 * (nsave + 1);
    if (!h->lp.P || h->lp.nsave != nsave) {
        FOS_HIP(hipStreamSynchronize(h->stream));
        dev_release(h, &h->lp.P); dev_release(h, &h->lp.bpart); dev_release(h, &h->lp.dots); dev_release(h, &h->lp.nu);    // (a caller sweeping nsave must not pile buffers up)
        FOS_TRY(dev_alloc(h, &h->lp.P, (size_t)K * h->l));
        FOS_TRY(dev_alloc(h, &h->lp.bpart, (size_t)K * h->vec_blocks));
        FOS_TRY(dev_alloc(h, &h->lp.dots, (size_t)h->vec_blocks * 2 * (LONG_KMAX_ROWS + 1)));
        FOS_TRY(dev_alloc(h, &h->lp.nu, (size_t)2 * K));
        // rows a saving window never wrote (fos_step entered in the middle of one) are zero planes, not uninitialised memory
        FOS_HIP(hipMemset(h->lp.P, 0, sizeof(d2) * (size_t)K * h->l));
        FOS_HIP(hipMemset(h->lp.bpart, 0, sizeof(double) * (size_t)K * h->vec_blocks));
    }
    h->lp.interval = longinterval; h->lp.nsave = nsave; h->lp.savepos = 0; h->lp.now = false;
    h->lp.max_supports = getenv("FOS_LONG_MAX_SUPPORTS") ? atoll(getenv("FOS_LONG_MAX_SUPPORTS")) : 4096;      // (0 tries none: the failure path, tests)
    h->lp.log[7] = 0.0;
    return FOS_OK;
}
// last projection of the LongstepWrapper: out8 = iteration, active inequalities, largest KKT violation of the small dual, |x_new - x|, rows, supports tried
int fos_longstep_log(fos_handle h, double* out8) {
    if (!h || !out8) { set_error("NULL argument"); return FOS_EINVAL; }
    memcpy(out8, h->lp.log, sizeof(h->lp.log));
    return FOS_OK;
}
int fos_linesearch_log(fos_handle h, double* out34) {
    if (!h || !out34) { set_error("NULL argument"); return FOS_EINVAL; }
    memcpy(out34, h->ls_log, sizeof(h->ls_log));
    return FOS_OK;
}

int fos_getsol(fos_handle h, double* z_out, int32_t force_check, double eps, fos_check_result* res) {
    if (!h) { set_error("NULL handle"); return FOS_EINVAL; }
    FOS_HIP(hipSetDevice(h->device));
    FOS_TRY(prox_affine(h, h->X));                       // prox!(tmp1, S1, x)
    FOS_TRY(prox_cones(h, h->T2, h->SOL));               // prox!(tmp2, S2, tmp1)
    if (force_check && res) {                                              // solverwrapper.jl:32-34
        FOS_TRY(status_check(h, h->T2, eps, res));
        h->last_checked = h->T2;
    }
    if (z_out) FOS_TRY(download_plain(h, z_out, h->T2));
    FOS_HIP(hipStreamSynchronize(h->stream));
    return FOS_OK;
}

int fos_get_affine_state(fos_handle h, double* xinit, int64_t* i, int32_t* firstrun) {
    if (!h) { set_error("NULL handle"); return FOS_EINVAL; }
    FOS_HIP(hipSetDevice(h->device));
    if (xinit) FOS_TRY(download_plain(h, xinit, h->SOL));
    if (i) *i = h->prox_i;
    if (firstrun) *firstrun = h->firstrun ? 1 : 0;
    return FOS_OK;
}

int fos_set_affine_state(fos_handle h, const double* xinit, int64_t i) {
    if (!h || !xinit || i < 1) { set_error("bad argument"); return FOS_EINVAL; }
    FOS_HIP(hipSetDevice(h->device));
    FOS_TRY(upload_plain(h, h->SOL, xinit));
    h->firstrun = false;
    h->prox_i = i;
    return FOS_OK;
}

// FISTAData (y, xold, t), DykstraData (p, q), GAPAData.alpha12: the algorithm's own state     fista.jl:15-25, dykstra.jl:12-23, gapa.jl:29
int fos_get_alg_state(fos_handle h, double* a, double* b, double* scal2) {
    if (!h) { set_error("NULL handle"); return FOS_EINVAL; }
    FOS_HIP(hipSetDevice(h->device));
    if (a) FOS_TRY(download_plain(h, a, h->Y));
    if (b) FOS_TRY(download_plain(h, b, h->XOLD));
    if (scal2) {
        FOS_TRY(poll_state(h));
        scal2[0] = h->fista_t;
        scal2[1] = h->st_host->alpha12;
    }
    return FOS_OK;
}

int fos_set_alg_state(fos_handle h, const double* a, const double* b, const double* scal2) {
    if (!h) { set_error("NULL handle"); return FOS_EINVAL; }
    FOS_HIP(hipSetDevice(h->device));
    if (a) FOS_TRY(upload_plain(h, h->Y, a));
    if (b) FOS_TRY(upload_plain(h, h->XOLD, b));
    if (scal2) {
        if (!(scal2[0] >= 1.0)) { set_error("fos_set_alg_state: FISTA's t is >= 1 (fista.jl:24,45), got %g", scal2[0]); return FOS_EINVAL; }
        h->fista_t = scal2[0];
        FOS_HIP(hipStreamSynchronize(h->stream));
        FOS_HIP(hipMemcpy(&h->st->alpha12, &scal2[1], sizeof(double), hipMemcpyHostToDevice));
    }
    return FOS_OK;
}

int fos_get_cgiter(fos_handle h, int64_t* cgiter) {
    if (!h || !cgiter) { set_error("NULL argument"); return FOS_EINVAL; }
    *cgiter = h->cgiter;
    return FOS_OK;
}

int fos_get_alpha12(fos_handle h, double* a) {
    if (!h || !a) { set_error("NULL argument"); return FOS_EINVAL; }
    FOS_HIP(hipSetDevice(h->device));
    FOS_TRY(poll_state(h));
    *a = h->st_host->alpha12;
    return FOS_OK;
}

int fos_get_prox_count(fos_handle h, int64_t* i) {
    if (!h || !i) { set_error("NULL argument"); return FOS_EINVAL; }
    *i = h->prox_i;
    return FOS_OK;
}

// ---- fine-grained entries ----------------------------------------------------------------------------

int fos_q_apply(fos_handle h, double* y, const double* x, int32_t transpose) {
    if (!h || !y || !x) { set_error("NULL argument"); return FOS_EINVAL; }
    FOS_HIP(hipSetDevice(h->device));
    LaunchCtx c = h->ctx();
    const size_t l = (size_t)h->l;
    FOS_HIP(hipMemcpyAsync(h->plain, x, sizeof(double) * l, hipMemcpyHostToDevice, h->stream));
    launch_set_comp(c, h->W, h->plain, 0);
    const double sign = transpose ? -1.0 : 1.0;           // HSDEAffine.jl:61-65
    int fr = 0;
    launch_q1(c, Q_PLAIN, h->W, 0, sign, h->plain + l);
    FOS_TRY(finish_reduce(h, c, c.S.npart, 1, 0, &fr, c.S.part_off));
    launch_q1_finalize(c, Q_PLAIN, h->W, 0, sign, h->plain + l, fr);
    FOS_HIP(hipMemcpyAsync(y, h->plain + l, sizeof(double) * l, hipMemcpyDeviceToHost, h->stream));
    FOS_HIP(hipStreamSynchronize(h->stream));
    return FOS_OK;
}

int fos_kkt_apply(fos_handle h, double* y, const double* x) {
    if (!h || !y || !x) { set_error("NULL argument"); return FOS_EINVAL; }
    FOS_HIP(hipSetDevice(h->device));
    LaunchCtx c = h->ctx();
    FOS_TRY(upload_plain(h, h->W, x));
    FOS_TRY(kkt_apply_full(h, c, h->W, h->AP));
    return download_plain(h, y, h->AP);
}

int fos_cg_kkt(fos_handle h, double* x, const double* rhs, double tol, int64_t max_iters, int64_t* iters) {
    if (!h || !x || !rhs || max_iters < 1) { set_error("bad argument"); return FOS_EINVAL; }
    FOS_HIP(hipSetDevice(h->device));
    FOS_TRY(upload_plain(h, h->W, x));
    FOS_TRY(upload_plain(h, h->RHS, rhs));
    int64_t it = 0;
    int pred = h->last_cg_pred;
    h->last_cg_pred = 0;
    FOS_TRY(cg_solve(h, h->W, h->RHS, tol, (int)std::min<int64_t>(max_iters, INT32_MAX), &it));
    h->last_cg_pred = pred;
    if (iters) *iters = it;
    return download_plain(h, x, h->W);
}

int fos_prox_affine(fos_handle h, double* y, const double* x) {
    if (!h || !y || !x) { set_error("NULL argument"); return FOS_EINVAL; }
    FOS_HIP(hipSetDevice(h->device));
    FOS_TRY(upload_plain(h, h->W, x));
    FOS_TRY(prox_affine(h, h->W));
    return download_plain(h, y, h->SOL);
}

int fos_hsdematrix_prox(fos_handle h, double* y, const double* x) {
    if (!h || !y || !x) { set_error("NULL argument"); return FOS_EINVAL; }
    FOS_HIP(hipSetDevice(h->device));
    LaunchCtx c = h->ctx();
    FOS_TRY(upload_plain(h, h->RHS, x));                               // rhs = x   HSDEAffine.jl:116
    if (h->firstrun2) {                                                // :109-112
        FOS_HIP(hipMemcpyAsync(h->SOL2, h->RHS, sizeof(d2) * h->l, hipMemcpyDeviceToDevice, h->stream));
        h->firstrun2 = false;
    }
    const double eps = 2.220446049250313e-16;
    const double tol = (double)(2 * h->l_global) * eps;                // :106
    int64_t it = 0;
    int pred = h->last_cg_pred;
    h->last_cg_pred = 0;
    FOS_TRY(cg_solve(h, h->SOL2, h->RHS, tol, 1000, &it));             // :116 ; xinit .= y :119
    h->last_cg_pred = pred;
    h->cgiter = it;
    // v = Q*u                                                          :122-124
    int fr = 0;
    launch_q1(c, Q_VFROMU, h->SOL2, 0, 1.0, h->W);
    FOS_TRY(finish_reduce(h, c, c.S.npart, 1, 0, &fr, c.S.part_off));
    launch_q1_finalize(c, Q_VFROMU, h->SOL2, 0, 1.0, h->W, fr);
    return download_plain(h, y, h->W);
}

int fos_prox_cones(fos_handle h, double* y, const double* x) {
    if (!h || !y || !x) { set_error("NULL argument"); return FOS_EINVAL; }
    FOS_HIP(hipSetDevice(h->device));
    FOS_TRY(upload_plain(h, h->W, x));
    FOS_TRY(prox_cones(h, h->T2, h->W));
    return download_plain(h, y, h->T2);
}

int fos_check(fos_handle h, const double* z, double eps, fos_check_result* res) {
    if (!h || !z || !res) { set_error("NULL argument"); return FOS_EINVAL; }
    FOS_HIP(hipSetDevice(h->device));
    FOS_TRY(upload_plain(h, h->W, z));
    return status_check(h, h->W, eps, res);
}

// ---- measurement -------------------------------------------------------------------------------------

int fos_profile(fos_handle h, int32_t enable) {
    if (!h) { set_error("NULL handle"); return FOS_EINVAL; }
    h->prof = enable != 0;
    h->prof_period = enable > 1 ? enable : 1;
    for (int k = 0; k < FOS_PROF_CLASSES; ++k) h->prof_seen[k] = 0;
    h->prof_steps = 0; h->prof_steps_sampled = 0; h->prof_step_on = false;
    return FOS_OK;
}

static double kkt_bytes(const fos_solver* h) {
    // B_kkt,min = 24 nnz + 4(m+n+2) + 32(m+n)
    return 24.0 * (double)h->nnz + 4.0 * (double)(h->m + h->n + 2) + 32.0 * (double)(h->m + h->n);
}

int fos_get_cg_total(fos_handle h, int64_t* total) {
    if (!h || !total) { set_error("NULL argument"); return FOS_EINVAL; }
    *total = h->cg_total;
    return FOS_OK;
}

// launches / summed milliseconds per class of the bracketed launch groups since the last read; resets the records
int fos_profile_read_classes(fos_handle h, int64_t* launches3, double* total_ms3) {
    if (!h) { set_error("NULL handle"); return FOS_EINVAL; }
    FOS_HIP(hipSetDevice(h->device));
    FOS_HIP(hipStreamSynchronize(h->stream));
    int64_t n[FOS_PROF_CLASSES] = {0, 0, 0, 0, 0};
    double t[FOS_PROF_CLASSES] = {0.0, 0.0, 0.0, 0.0, 0.0};
    for (size_t i = 0; i < h->prof_used; ++i) {
        const auto& r = h->prof_recs[i];
        if (r.cls < 0 || r.cls >= FOS_PROF_CLASSES) continue;
        float ms = 0.f;
        FOS_HIP(hipEventElapsedTime(&ms, r.a, r.b));
        n[r.cls] += 1; t[r.cls] += ms;
    }
    n[FOS_PROF_OTHER] = h->prof_steps_sampled;           // this class is normalised per sampled OUTER ITERATION (foship.h)
    h->prof_steps_sampled = 0;
    for (int k = 0; k < FOS_PROF_CLASSES; ++k) {
        if (launches3) launches3[k] = n[k];
        if (total_ms3) total_ms3[k] = t[k];
    }
    h->prof_used = 0;
    return FOS_OK;
}

int fos_profile_read(fos_handle h, int64_t* launches, double* total_ms, double* bytes_per_launch) {
    int64_t n[FOS_PROF_CLASSES];
    double t[FOS_PROF_CLASSES];
    FOS_TRY(fos_profile_read_classes(h, n, t));
    if (launches) *launches = n[FOS_PROF_KKT];
    if (total_ms) *total_ms = t[FOS_PROF_KKT];
    if (bytes_per_launch) *bytes_per_launch = kkt_bytes(h);
    return FOS_OK;
}

int fos_operator_stats(fos_handle h, int64_t* stats) {
    if (!h || !stats) { set_error("NULL argument"); return FOS_EINVAL; }
    const HostBlkCsr& S = h->hostS;
    int64_t nell = 0, nlds = 0, nlong = 0, nrun = 0;
    for (const BlkDesc& d : S.blk) {
        if (d.kind() == BLK_ELL) ++nell; else if (d.kind() == BLK_LDS) ++nlds; else if (d.kind() == BLK_LONG) ++nlong;
        if (d.run()) ++nrun;
    }
    stats[0] = S.nblk; stats[1] = nell; stats[2] = nlds; stats[3] = nlong; stats[4] = nrun;
    stats[5] = S.nnz_padded; stats[6] = S.ncol_stored; stats[7] = h->S.nwaves;
    stats[8] = S.ntiles; stats[9] = S.nslots; stats[10] = h->S.ndef; stats[11] = S.tile_values;
    return FOS_OK;
}

int fos_window_stats(fos_handle h, int64_t* stats4) {
    if (!h || !stats4) { set_error("NULL argument"); return FOS_EINVAL; }
    for (int k = 0; k < 4; ++k) stats4[k] = h->win_stats[k];
    return FOS_OK;
}

int fos_bench_kkt(fos_handle h, int32_t reps, double* total_ms) {
    if (!h || reps < 1) { set_error("bad argument"); return FOS_EINVAL; }
    FOS_HIP(hipSetDevice(h->device));
    LaunchCtx c = h->ctx();
    hipEvent_t e0, e1;
    FOS_HIP(hipEventCreate(&e0));
    FOS_HIP(hipEventCreate(&e1));
    launch_kkt2(c, h->X, h->AP, 0);           // warm
    FOS_HIP(hipEventRecord(e0, h->stream));
    for (int i = 0; i < reps; ++i) launch_kkt2(c, h->X, h->AP, 0);
    FOS_HIP(hipEventRecord(e1, h->stream));
    FOS_HIP(hipEventSynchronize(e1));
    float ms = 0.f;
    FOS_HIP(hipEventElapsedTime(&ms, e0, e1));
    if (total_ms) *total_ms = ms;
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    return FOS_OK;
}

int fos_psd_debug(fos_handle h, int32_t collect_stats, int32_t phase_limit) {
    if (!h) { set_error("NULL handle"); return FOS_EINVAL; }
    FOS_HIP(hipSetDevice(h->device));
    if (collect_stats && !h->psd_stats && h->npsd > 0) {
        FOS_TRY(dev_alloc(h, &h->psd_stats, (size_t)2 * h->npsd));
        FOS_HIP(hipMemset(h->psd_stats, 0, sizeof(int) * 2 * h->npsd));
    }
    h->psd_phase_limit = phase_limit;
    return FOS_OK;
}

int fos_psd_stats(fos_handle h, int32_t* sweeps, int64_t cap, int64_t* count) {
    if (!h || !count) { set_error("NULL argument"); return FOS_EINVAL; }
    FOS_HIP(hipSetDevice(h->device));
    *count = h->psd_stats ? 2 * (int64_t)h->npsd : 0;
    if (sweeps && h->psd_stats) {
        FOS_HIP(hipStreamSynchronize(h->stream));
        FOS_HIP(hipMemcpy(sweeps, h->psd_stats, sizeof(int32_t) * (size_t)std::min<int64_t>(cap, *count), hipMemcpyDeviceToHost));
    }
    return FOS_OK;
}

int fos_bench_cg_chain(fos_handle h, int32_t iters, int32_t reps, int32_t use_graph, double* ms_per_iter) {
    if (!h || iters < 1 || reps < 1 || !ms_per_iter) { set_error("bad argument"); return FOS_EINVAL; }
    if (h->sharded()) { set_error("fos_bench_cg_chain: single-GPU handles only"); return FOS_EUNSUPPORTED; }
    FOS_HIP(hipSetDevice(h->device));
    LaunchCtx c = h->ctx();
    // CG on M y = X from y = 0 with a tolerance that is never met: every enqueued iteration really runs
    FOS_HIP(hipMemcpyAsync(h->RHS, h->X, sizeof(d2) * h->l, hipMemcpyDeviceToDevice, h->stream));
    FOS_HIP(hipMemsetAsync(h->W, 0, sizeof(d2) * h->l, h->stream));
    FOS_TRY(kkt_apply_full(h, c, h->W, h->AP));
    launch_cg_init(c, h->RHS, h->AP, h->R, h->PB[1]);
    launch_cg_init_finalize(c, h->R, -1.0, INT32_MAX, 0);
    const int variant = h->cg_variant >= 0 ? h->cg_variant : (h->fuse_p ? FOS_CG_FUSED_P : FOS_CG_REFERENCE);
    const bool merged = variant == FOS_CG_MERGED_SWEEP || variant == FOS_CG_MERGED_UPDATE;
    // (the producer flags of the update kernels carry the launch's sequence number: a replayed graph would present the SAME
    //  numbers again and the consumers would not wait -- the graph runs without the producers)
    if (use_graph) c.pre = nullptr;
    uint32_t chain_no = 0;
    auto chain = [&]() {
        const uint32_t seq_base = (++chain_no) * 2048u;
        if (merged) {
            CgmIter it;
            it.x = h->W; it.r = h->R; it.p = h->PB[0]; it.s = h->PB[1]; it.w = h->AP;
            it.close_in_update = variant == FOS_CG_MERGED_UPDATE; it.from_reduced = 0; it.fold = nullptr; it.seq_base = seq_base;
            it.j = 0;
            launch_cgm_apply(c, it, h->W);
            launch_cgm_start(c, it, h->RHS, h->W, -1.0, INT32_MAX);
            launch_cgm_sweep(c, it, it.close_in_update ? -1 : 0);
            for (int j = 1; j <= iters; ++j) {
                it.j = j;
                launch_cgm_update(c, it, false);
                launch_cgm_sweep(c, it, it.close_in_update ? -1 : j);
            }
            return;
        }
        for (int j = 1; j <= iters; ++j) {
            CgIter it;
            it.j = j; it.r = h->R; it.p_prev = h->PB[(j - 1) & 1]; it.p_cur = h->PB[j & 1];
            it.fuse_p = variant == FOS_CG_FUSED_P; it.rr_from_reduced = 0; it.fold = nullptr; it.seq_base = seq_base;
            launch_kkt2_cg(c, it, h->AP);
            launch_cg_update(c, it, h->W, h->R, h->AP, 0);
            if (!it.fuse_p) launch_cg_pupdate(c, it, h->W, h->PB[(j + 1) & 1]);
        }
    };
    hipEvent_t e0, e1;
    FOS_HIP(hipEventCreate(&e0));
    FOS_HIP(hipEventCreate(&e1));
    hipGraph_t graph = nullptr;
    hipGraphExec_t exec = nullptr;
    if (use_graph) {
        FOS_HIP(hipStreamSynchronize(h->stream));
        FOS_HIP(hipStreamBeginCapture(h->stream, hipStreamCaptureModeThreadLocal));
        chain();
        FOS_HIP(hipStreamEndCapture(h->stream, &graph));
        FOS_HIP(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
        FOS_HIP(hipGraphLaunch(exec, h->stream));          // warm
    } else {
        chain();                                           // warm
    }
    FOS_HIP(hipEventRecord(e0, h->stream));
    for (int rep = 0; rep < reps; ++rep) {
        if (use_graph) FOS_HIP(hipGraphLaunch(exec, h->stream));
        else chain();
    }
    FOS_HIP(hipEventRecord(e1, h->stream));
    FOS_HIP(hipEventSynchronize(e1));
    float ms = 0.f;
    FOS_HIP(hipEventElapsedTime(&ms, e0, e1));
    *ms_per_iter = (double)ms / ((double)reps * iters);
    if (exec) (void)hipGraphExecDestroy(exec);
    if (graph) (void)hipGraphDestroy(graph);
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    // leave the handle as fos_reset_affine + a fresh iterate would: the CG state used above is scratch
    FOS_TRY(check_launch("fos_bench_cg_chain"));
    FOS_TRY(poll_state(h));
    return FOS_OK;
}

int fos_host_resident_cg(int64_t m, int64_t n, const int64_t* colptr, const int64_t* rowval, const double* nzval, const double* b, const double* c,
                         int32_t gmax, double* x, const double* rhs, double tol, int64_t max_iters, int64_t* iters, int64_t* stats8) {
    if (!colptr || m < 0 || n < 0 || gmax < 1 || (x && (!rhs || !b || !c || max_iters < 1))) { set_error("bad argument"); return FOS_EINVAL; }
    HostBlkCsr S;
    const int cus = 256;                       // (the operator as fos_create builds it on a 256-CU device)
    FOS_TRY(build_stacked_csr(m, n, colptr, rowval, nzval, cus * 12, &S, cus * 28, -1, false, cus * 16));
    ResPlan P;
    const bool ok = build_resident_plan(S, m, n, gmax, &P);
    if (stats8) {
        stats8[0] = ok ? 1 : 0; stats8[1] = P.G; stats8[2] = P.nw; stats8[3] = P.stream ? -P.nt : P.rpt; stats8[4] = P.units; stats8[5] = P.tiles_wg_max; stats8[6] = P.tmax; stats8[7] = ok ? 1 : 0;
    }
    if (!ok) { set_error("FOS_CG_RESIDENT: the operator does not qualify (%s)", P.why.c_str()); return x ? FOS_EUNSUPPORTED : FOS_OK; }
    if (!x) return FOS_OK;
    const int64_t l = n + m + 1;
    std::vector<double> cb((size_t)(n + m));
    for (int64_t j = 0; j < n; ++j) cb[j] = c[j];
    for (int64_t i = 0; i < m; ++i) cb[n + i] = b[i];
    std::vector<double2> xv((size_t)l), rv((size_t)l);
    for (int64_t i = 0; i < l; ++i) { xv[i] = double2{x[i], x[l + i]}; rv[i] = double2{rhs[i], rhs[l + i]}; }      // plain [part1; part2] -> interleaved
    std::vector<double2> v0(xv);
    int it = 0;
    FOS_TRY(host_resident_cg(S, P, m, n, cb.data(), xv.data(), rv.data(), v0.data(), tol, (int)std::min<int64_t>(max_iters, 1 << 30), &it));
    for (int64_t i = 0; i < l; ++i) { x[i] = xv[i].x; x[l + i] = xv[i].y; }
    if (iters) *iters = it;
    return FOS_OK;
}

int fos_host_stacked_spmv(int64_t m, int64_t n, const int64_t* colptr, const int64_t* rowval, const double* nzval,
                          const double* v, double* out, int32_t spmv_workgroups, int32_t resident_waves, int64_t* stats) {
    if (!colptr || !v || !out || m < 0 || n < 0) { set_error("bad argument"); return FOS_EINVAL; }
    HostBlkCsr S;
    FOS_TRY(build_stacked_csr(m, n, colptr, rowval, nzval, spmv_workgroups > 0 ? spmv_workgroups : 1024, &S, resident_waves));
    std::string why;
    int rc = host_stacked_spmv(S, v, out, &why);
    if (rc != FOS_OK) { set_error("operator format check failed: %s", why.c_str()); return rc; }
    if (stats) {
        int64_t nell = 0, nlds = 0, nlong = 0, nrun = 0;
        for (const BlkDesc& d : S.blk) {
            if (d.kind() == BLK_ELL) ++nell; else if (d.kind() == BLK_LDS) ++nlds; else if (d.kind() == BLK_LONG) ++nlong;
            if (d.run()) ++nrun;
        }
        stats[0] = S.nblk; stats[1] = nell; stats[2] = nlds; stats[3] = nlong; stats[4] = nrun;
        stats[5] = S.nnz_padded; stats[6] = S.ncol_stored; stats[7] = S.nwaves;
        stats[8] = S.ntiles; stats[9] = S.nslots; stats[10] = (int64_t)S.def_rows.size(); stats[11] = S.tile_values;
    }
    return FOS_OK;
}

int fos_host_stacked_spmv_mode(int64_t m, int64_t n, const int64_t* colptr, const int64_t* rowval, const double* nzval,
                               const double* v, double* out, int32_t window_mode, int64_t* stats16) {
    if (!colptr || !v || !out || m < 0 || n < 0) { set_error("bad argument"); return FOS_EINVAL; }
    HostBlkCsr S;
    const bool rs = getenv("FOS_HOST_SPMV_ROW_SHARDED") && atoi(getenv("FOS_HOST_SPMV_ROW_SHARDED")) != 0;      // (tests: the row-sharded builder)
    FOS_TRY(build_stacked_csr(m, n, colptr, rowval, nzval, 1024, &S, 0, window_mode, rs));
    std::string why;
    int rc = host_stacked_spmv(S, v, out, &why);
    if (rc != FOS_OK) { set_error("operator format check failed: %s", why.c_str()); return rc; }
    if (stats16) {
        for (int k = 0; k < 16; ++k) stats16[k] = 0;
        stats16[0] = S.nblk; stats16[5] = S.nnz_padded; stats16[6] = S.ncol_stored; stats16[8] = S.ntiles; stats16[9] = S.nslots;
        stats16[12] = (int64_t)S.wpanel.size(); stats16[13] = (int64_t)S.wdesc.size() / S.wgeom.waves; stats16[14] = S.wnslice; stats16[15] = (int64_t)S.wval.size();
    }
    return FOS_OK;
}

int fos_sync(fos_handle h) {
    if (!h) { set_error("NULL handle"); return FOS_EINVAL; }
    FOS_HIP(hipSetDevice(h->device));
    FOS_HIP(hipStreamSynchronize(h->stream));
    return FOS_OK;
}

int fos_set_tuning(fos_handle h, int32_t spmv_workgroups, int32_t cg_chunk, int32_t fuse_p) {
    if (!h) { set_error("NULL handle"); return FOS_EINVAL; }
    FOS_HIP(hipSetDevice(h->device));
    if (fuse_p >= 0) { h->fuse_p = fuse_p != 0; h->cg_variant = -1; }
    if (cg_chunk > 0) h->cg_chunk = cg_chunk;
    if (spmv_workgroups > 0 && h->S.npanel > 0) {
        // window panels: the grid is a number of persistent workgroups walking the panels
        h->S.nwg = std::max(1, std::min<int32_t>(std::min(spmv_workgroups, 16384), h->S.npanel));
        h->S.npart = h->S.nwg; h->S.part_off = 0;
    } else if (spmv_workgroups > 0) {
        if (spmv_workgroups > 16384) spmv_workgroups = 16384;
        FOS_HIP(hipStreamSynchronize(h->stream));
        partition_workgroups(&h->hostS, spmv_workgroups);
        FOS_HIP(hipMemcpy(const_cast<int32_t*>(h->S.wave_blk0), h->hostS.wave_blk0.data(),
                          h->hostS.wave_blk0.size() * sizeof(int32_t), hipMemcpyHostToDevice));
        FOS_HIP(hipMemcpy(const_cast<BlkDesc*>(h->S.wave_first), h->hostS.wave_first.data(),
                          h->hostS.wave_first.size() * sizeof(BlkDesc), hipMemcpyHostToDevice));
        h->S.nwg = h->hostS.nwg;
        h->S.npart = h->S.nwg_def > 0 ? h->S.nwg_def : h->S.nwg;
        h->S.part_off = h->S.nwg_def > 0 ? h->S.nwg : 0;
        h->S.nwaves = h->hostS.nwaves;
        h->nwg_target = spmv_workgroups;
    }
    return FOS_OK;
}

// which CG recurrence the affine projection runs (FOS_CG_*; -1: the handle's default)
int fos_set_cg_variant(fos_handle h, int32_t variant) {
    if (!h || variant < -1 || variant > FOS_CG_RESIDENT) { set_error("unknown CG variant %d", (int)variant); return FOS_EINVAL; }
    if (variant == FOS_CG_RESIDENT && !h->res_ok) {
        set_error("FOS_CG_RESIDENT: the operator does not qualify (%s)", h->res_plan.why.empty() ? "no plan" : h->res_plan.why.c_str());
        return FOS_EUNSUPPORTED;
    }
    h->cg_variant = variant;
    if (variant >= 0) h->fuse_p = variant == FOS_CG_FUSED_P;
    h->last_cg_pred = 0; h->cg_same_run = 0;
    return FOS_OK;
}

// the variant the next affine projection will run (the default resolved: sharded handles take the merged recurrence)
int fos_get_cg_variant(fos_handle h, int32_t* variant) {
    if (!h || !variant) { set_error("NULL argument"); return FOS_EINVAL; }
    static const bool fold_env = !(getenv("FOS_PEER_FOLD") && atoi(getenv("FOS_PEER_FOLD")) == 0);
    // default: the reference's recurrence -- except where a CG iteration is bound by its launches, not its bytes: sharded handles (one
    // exchange per iteration instead of two) and cache-resident gather-type operators of 32 768 rows or more (C3: two launches per iteration
    // instead of three, 370 -> 385 iterations/s; the streamed operators C2 / C4 / C5 are faster on the reference recurrence; small problems
    // keep the reference's arithmetic).  Same Krylov iterates, same
    // iteration counting and stop test; FOS_CG_VARIANT=0 / fos_set_cg_variant restore the reference recurrence everywhere.
    int v = h->cg_variant >= 0 ? h->cg_variant
            : (h->fuse_p ? FOS_CG_FUSED_P : ((h->sharded() || (h->S.resident && !h->row_sharded && h->l >= 32768)) ? FOS_CG_MERGED_UPDATE : FOS_CG_REFERENCE));
    // the resident solve: on request wherever the operator qualifies; by default on sharded handles whose sums travel through mailboxes and
    // whose shards ALL qualify (FOS_RESIDENT_DEFAULT=0: never by default).  Sharded without mailboxes (RCCL, the caller's collective): a
    // collective call cannot sit inside a kernel -- the launch-per-iteration form of the same recurrence runs instead.
    const bool res_default = !(getenv("FOS_RESIDENT_DEFAULT") && atoi(getenv("FOS_RESIDENT_DEFAULT")) == 0);      // (read at every call: bench.py turns it off after a failed warm-up)
    const bool res_usable = h->res_ok && !h->row_sharded && (!h->sharded() || (h->peer_on && fold_env && h->res_all));
    if (h->cg_variant < 0 && !h->fuse_p && h->sharded() && res_usable && res_default) v = FOS_CG_RESIDENT;
    // one GPU: the STREAMED form where it fills at least half of the chip (C4: 66.3 us per CG iteration against 89 for three launches); operators
    // small enough for the register form keep the reference's arithmetic by default
    if (h->cg_variant < 0 && !h->fuse_p && !h->sharded() && res_usable && res_default && h->res_plan.stream && 2 * h->res_plan.G >= h->cus) v = FOS_CG_RESIDENT;
    if (v == FOS_CG_RESIDENT && !res_usable) v = FOS_CG_MERGED_UPDATE;
    if (v == FOS_CG_MERGED_SWEEP && h->sharded()) v = FOS_CG_MERGED_UPDATE;
    if (v == FOS_CG_MERGED_UPDATE && h->peer_on && !fold_env) v = FOS_CG_REFERENCE;
    *variant = v;
    return FOS_OK;
}

int fos_resident_stats(fos_handle h, int64_t* stats8) {
    if (!h || !stats8) { set_error("NULL argument"); return FOS_EINVAL; }
    const ResPlan& p = h->res_plan;
    stats8[0] = h->res_ok ? 1 : 0; stats8[1] = p.G; stats8[2] = p.nw; stats8[3] = p.stream ? -p.nt : p.rpt; stats8[4] = p.units; stats8[5] = p.tiles_wg_max;
    stats8[6] = p.tmax; stats8[7] = (h->sharded() ? h->res_all : h->res_ok) ? 1 : 0;
    if (!h->res_ok) set_error("FOS_CG_RESIDENT: %s", p.why.empty() ? "no plan" : p.why.c_str());
    return FOS_OK;
}

// test hooks
int fos_debug_set(fos_handle h, int32_t what, int64_t value) {
    if (!h) { set_error("NULL handle"); return FOS_EINVAL; }
    FOS_HIP(hipSetDevice(h->device));
    if (what == FOS_DEBUG_PUPDATE_DELAY) {
        const int32_t v = (int32_t)std::max<int64_t>(0, std::min<int64_t>(value, 100000000));     // <= 1 s
        FOS_HIP(hipStreamSynchronize(h->stream));
        FOS_HIP(hipMemcpy(&h->st->dbg_delay, &v, sizeof(v), hipMemcpyHostToDevice));
        return FOS_OK;
    }
    set_error("unknown debug switch %d", (int)what);
    return FOS_EINVAL;
}

}  // extern "C"
