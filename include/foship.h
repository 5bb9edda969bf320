/*
 * foship.h -- C ABI of libfoship.so: the MI355X (gfx950) implementation of the per-iteration hot path of
 * FirstOrderSolvers.jl (GAP / DR / AP / GAPA / FISTA over the homogeneous self-dual embedding).
 *
 * This is the drop-in boundary.  The reference has no FFI; the seam it offers is
 * Julia multiple dispatch on the algorithm type.  Each entry point below names the reference function a
 * Julia `ccall` shim (firstordersolvers.jl_amd/julia/FOSHip.jl) replaces with it.
 * All paths are relative to the reference checkout.
 *
 * Conventions
 *   - plain C, no exceptions across the boundary, no callbacks into the host language;
 *   - every function returns 0 on success, a negative FOS_E* code on failure; fos_last_error() returns a
 *     thread-local message for the last failure;
 *   - host arrays are owned by the caller and never retained; device memory is owned by the handle;
 *   - a handle is not thread safe (the reference is single threaded);
 *   - matrix input is Julia's SparseMatrixCSC{Float64,Int64}: 1-based colptr / rowval  (src/types.jl:35);
 *   - iterates use the reference's layout z = [x(n); y(m); tau; r(n); s(m); kappa], N = 2(n+m+1) doubles
 *     (src/cones.jl:126-141, src/problemforms/HSDE/HSDEStatus.jl:93-101);
 *   - all arithmetic is IEEE fp64.
 */
#ifndef FOSHIP_H
#define FOSHIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FOS_ABI_VERSION 1

/* error codes */
#define FOS_OK            0
#define FOS_EINVAL       -1   /* bad argument (the reference would fail an @assert)            */
#define FOS_EHIP         -2   /* HIP runtime error                                            */
#define FOS_ENOMEM       -3
#define FOS_EUNSUPPORTED -4   /* cone / option not implemented by the HIP path                */
#define FOS_ECOMM        -5   /* RCCL error                                                   */
#define FOS_ENODEVICE    -6   /* no MI355X visible: the product path has no CPU fallback      */

/* cone codes: the keys of `conemap`, src/cones.jl:4-14 */
#define FOS_CONE_FREE       0   /* :Free       IndFree()              */
#define FOS_CONE_ZERO       1   /* :Zero       IndZero()              */
#define FOS_CONE_NONNEG     2   /* :NonNeg     IndNonnegative()       */
#define FOS_CONE_NONPOS     3   /* :NonPos     IndNonpositive()       */
#define FOS_CONE_SOC        4   /* :SOC        IndSOC()               */
#define FOS_CONE_SOCROT     5   /* :SOCRotated IndRotatedSOC()        */
#define FOS_CONE_SDP        6   /* :SDP        IndPSD(scaling=true)   */
#define FOS_CONE_EXPPRIMAL  7   /* :ExpPrimal  IndExpPrimal()  (3 entries per cone) */
#define FOS_CONE_EXPDUAL    8   /* :ExpDual    IndExpDual()    (3 entries per cone) */

/* algorithms: src/solvers/{gap,gapa,fista,dykstra}.jl */
#define FOS_ALG_GAP     0   /* GAP(alpha, alpha1, alpha2); DR = GAP(a,2,2), AP = GAP(a,1,1)  (solvers.jl:10-11) */
#define FOS_ALG_GAPA    1   /* GAPA(alpha, beta)                                                              */
#define FOS_ALG_FISTA   2   /* FISTA(alpha)                                                                   */
#define FOS_ALG_DYKSTRA 3   /* Dykstra()                                                                      */
#define FOS_ALG_GAPP    4   /* GAPP(alpha, alpha1, alpha2; iproj): Feasibility form only (fos_feas_set_gapp)   */

/* status codes <-> Symbols of HSDEStatus.status (src/problemforms/HSDE/HSDEStatus.jl:53-63) */
#define FOS_STATUS_CONTINUE   0
#define FOS_STATUS_OPTIMAL    1
#define FOS_STATUS_UNBOUNDED  2
#define FOS_STATUS_INFEASIBLE 3

/* what a convergence check returns: the values checkstatus computes / savedata stores
 * (HSDEStatus.jl:33-38, 127-131) plus the CG count printed in the table (:44-47). */
typedef struct fos_check_result {
    double p;        /* relative primal residual   HSDEStatus.jl:34 */
    double d;        /* relative dual residual     :35 */
    double g;        /* relative duality gap       :38 */
    double ctx;      /* c'x                        :36 */
    double bty;      /* b'y                        :37 */
    double kappa;
    double tau;
    double norm_axs; /* ||A x + s||                :59 */
    double norm_aty; /* ||A'y||                    :61 */
    double norm_b;
    double norm_c;
    int64_t cgiter;  /* getcgiter(data): CG iterations of the last affine projection (defaults.jl:25) */
    int32_t status;  /* FOS_STATUS_*               :53-63 */
    int32_t cg_maxiter_hit; /* 1 if any CG since the previous check stopped on max_iters (the reference @warns, conjugategradients.jl:53) */
} fos_check_result;

typedef struct fos_solver* fos_handle;

/* ---- library ------------------------------------------------------------------------------------ */
int         fos_abi_version(void);
const char* fos_last_error(void);
int         fos_device_count(int* count);                  /* FOS_ENODEVICE if HIP sees no GPU */
int         fos_device_name(int device, char* buf, int buflen);

/* ---- problem set-up: replaces HSDE(model) + init_algorithm! ---------------------------------------
 * fos_create: loadproblem! -> init_algorithm! -> get_sets_and_status -> HSDE(model; direct=false)
 *   (src/FOSSolverInterface.jl:27-64,76-79; src/problemforms/HSDE/HSDE.jl:7-29).
 *   Problem: minimize c'x  s.t.  b - A x in K1,  x in K2;  A is m x n CSC, 1-based.
 *   Cones: K?type[i] a FOS_CONE_* code, K?start[i] 1-based first index, K?len[i] length; ranges must be
 *   contiguous, ordered and cover 1..m (K1) / 1..n (K2) exactly -- the ConeProduct assertion,
 *   src/cones.jl:66-72 -- else FOS_EINVAL.  Builds S1 = AffinePlusLinear(Q,0,0,1,decreasing_accuracy=true)
 *   and S2 = DualConeProduct(K1,K2) on the device `device`. */
int fos_create(int64_t m, int64_t n,
               const int64_t* colptr, const int64_t* rowval, const double* nzval,
               const double* b, const double* c,
               int64_t nK1, const int32_t* K1type, const int64_t* K1start, const int64_t* K1len,
               int64_t nK2, const int32_t* K2type, const int64_t* K2start, const int64_t* K2len,
               int device, fos_handle* out);

/* fos_create with flags.  FOS_CREATE_ROW_SHARDED (multi-GPU for an A that is NOT block diagonal): this rank
 * holds the ROWS of A that belong to its K1 cones (m = local rows, b local, K1 local) and ALL n columns (c, K2 whole).  x, r,
 * tau, kappa are replicated on every rank, y and s are local.  Per Q apply the n-vector A'y = sum over ranks of A_g'y_g is
 * all-reduced in stream (RCCL; HSDEAffine.jl:51), every scalar sum counts the replicated entries once (rank 0).  Follow with
 * fos_comm_init on every rank (or the peer mailboxes + fos_peer_vec_*); without a communicator the handle behaves as the only rank.  Dense
 * rectangles of the rank's rows are stored once (dual tiles); their column sums are added up locally before the n-vector crosses the ranks. */
#define FOS_CREATE_ROW_SHARDED 1
int fos_create2(int64_t m, int64_t n, const int64_t* colptr, const int64_t* rowval, const double* nzval,
                const double* b, const double* c,
                int64_t nK1, const int32_t* K1type, const int64_t* K1start, const int64_t* K1len,
                int64_t nK2, const int32_t* K2type, const int64_t* K2start, const int64_t* K2len,
                int device, int32_t flags, fos_handle* out);
int fos_destroy(fos_handle h);                             /* Julia finalizer */
int fos_sizes(fos_handle h, int64_t* m, int64_t* n, int64_t* N, int64_t* nnz);

/* Multi-GPU: the problem given to fos_create is this rank's SHARD of a
 * block-separable problem (whole cones of K1 with the matching rows/columns of a block-diagonal A; tau and
 * kappa replicated).  Only scalars cross GPUs: one RCCL all-reduce per reduction point of the CG / status.
 * fos_comm_get_unique_id: rank 0 obtains an ncclUniqueId (128 bytes), the host broadcasts it;
 * fos_comm_init: collective over all ranks.  Without it the handle is a single-GPU solver. */
int fos_comm_get_unique_id(void* id128);
int fos_comm_init(fos_handle h, int nranks, int rank, const void* id128);
/* The same with the CALLER's collective instead of RCCL (a Julia host with MPI.jl, a test harness with gloo): every cross-rank
 * sum -- scalars, and the n-vector of a row-sharded handle -- is staged through a pinned host buffer and handed to
 * `fn(user, buf, count)`, which must replace buf[0..count) by its sum over the ranks (blocking, the same call sequence on
 * every rank) and return 0.  Slow by design (one stream synchronisation per sum); excludes fos_comm_init / peer mailboxes. */
typedef int (*fos_allreduce_fn)(void* user, double* buf, int64_t count);
int fos_comm_init_host(fos_handle h, int nranks, int rank, fos_allreduce_fn fn, void* user);

/* Peer mailboxes: the same scalar sums WITHOUT a collective call in the stream.  Every rank owns a mailbox in
 * uncached device memory that its peers map through HIP IPC; the kernel that reduces a rank's partial sums stores
 * them into every peer's mailbox, waits for the peers' words (each 8-byte word carries its sequence number) and
 * adds them in rank order -- the same bits on every rank, ~one xGMI write latency instead of a library
 * all-reduce per CG inner product (conjugategradients.jl:39,46).  Ranks may be processes on different GPUs of one
 * node (or, for tests, on the same GPU).  Protocol (all collective, the host moves the 64-byte handles):
 *   fos_peer_export(h, handle64)                     -> allocate the mailbox, get its hipIpcMemHandle_t
 *   fos_peer_open(h, nranks, rank, handles, timeout) -> handles = nranks x 64 bytes in rank order (own entry ignored);
 *                                                       timeout_s <= 0: 20 s; nranks <= 16
 *   fos_peer_selftest(h, rounds, &ok)                -> exchanges of known values checked exactly; ok = 0 on mismatch/time-out
 *   fos_peer_enable(h, on)                           -> use the mailboxes (on = 0: back to RCCL if fos_comm_init was
 *                                                       called, else single GPU); recomputes the global size / norms
 * The two exchanges of every CG iteration are folded into the CG vector kernels (every workgroup polls, workgroup 0 also
 * writes), so a sharded CG iteration needs no more launches than an unsharded one.
 * An exchange that waits longer than the time-out stops the solve with FOS_ECOMM (never a hang). */
int fos_peer_export(fos_handle h, void* handle64);
int fos_peer_open(fos_handle h, int nranks, int rank, const void* handles, double timeout_s);
/* The same mailboxes in PINNED HOST MEMORY (a third transport between the device mailboxes and the RCCL all-reduce): one POSIX shared-memory
 * segment `shm_name` ("/name", the same string on every rank of the node, unique to the job) that every rank maps and registers with the HIP
 * runtime -- it needs neither peer access between the devices nor an IPC handle, so it works wherever the ranks share a host.  An exchange is
 * one posted PCIe write of the rank's words and PCIe reads by ONE polling workgroup, which republishes what it sees in local device memory for
 * the others (csrc/fos_internal.hpp, PeerBox::relay): about two PCIe latencies per CG inner product (conjugategradients.jl:39,46) instead of one
 * xGMI write latency.  Collective; replaces fos_peer_export + fos_peer_open, then fos_peer_selftest / fos_peer_enable as for the device mailboxes.
 * Cone-sharded handles (row-sharded ones keep their n-vector exchange on RCCL or the device mailboxes).  The segment is unlinked by rank 0's fos_peer_close / fos_destroy.
 * fos_peer_close: drops whichever mailboxes are open (device or host) so that another transport can be opened on the same handle; the handle falls
 * back to RCCL if fos_comm_init was called, else to a single GPU. */
int fos_peer_open_host(fos_handle h, int nranks, int rank, const char* shm_name, double timeout_s);
int fos_peer_close(fos_handle h);
int fos_peer_selftest(fos_handle h, int rounds, int32_t* ok);
/* Diagnostic of an N-rank run (no reference equivalent): the cost of ONE exchange of four doubles on the transport the sharded handle uses --
 * `rounds` exchanges back to back in stream between two events (mailboxes: inside one launch; RCCL: `rounds` ncclAllReduce calls).  Collective. */
int fos_exchange_bench(fos_handle h, int rounds, double* us_per_exchange);
/* Row-sharded handles (FOS_CREATE_ROW_SHARDED) over the mailboxes: the n-vector A'y = sum over ranks of A_g'y_g (HSDEAffine.jl:51)
 * also crosses the ranks through peer-mapped memory -- every rank pushes its 2n partial sums into every peer's exchange buffer and
 * adds what it received in rank order, in stream, no collective library.  After fos_peer_open: fos_peer_vec_export(h, handle64),
 * the host all-gathers the handles, fos_peer_vec_open(h, handles: nranks x 64 bytes in rank order), then fos_peer_enable. */
int fos_peer_vec_export(fos_handle h, void* handle64);
int fos_peer_vec_open(fos_handle h, const void* handles);
int fos_peer_enable(fos_handle h, int32_t on);

/* ---- algorithm state: replaces init_algorithm!(alg, model) data structs ---------------------------
 * fos_set_alg: GAP/GAPA/FISTA/Dykstra constructor arguments (gap.jl:13, gapa.jl:15, fista.jl:11) and a
 *   fresh *Data struct (gap.jl:23-28, gapa.jl:27-32: alpha12 = 2.0, fista.jl:20-25: t = 1, y = xold = 0,
 *   dykstra.jl:19-23: p = q = 0).  Does NOT reset S1 (call counter i, CG warm start): in the reference
 *   those live in the model and persist across solve! calls (affinepluslinear.jl:66,114).
 * fos_reset_affine: a fresh AffinePlusLinear (i = 1, firstrun = true) == a new loadproblem!. */
int fos_set_alg(fos_handle h, int alg, double alpha, double alpha1, double alpha2, double beta);
int fos_reset_affine(fos_handle h);

/* direct = true (the `direct` field of GAP/GAPA/FISTA/Dykstra; HSDE.jl:12-15): S1 becomes IndAffine([Q -I], 0), the EXACT
 * projection onto {Q u = v} through a factorisation formed once, instead of AffinePlusLinear's warm-started CG.  The
 * reference factorises the sparse [Q -I] on the CPU (ProximalOperators.IndAffine); here (I + Q Q')^-1 is formed once as a
 * dense matrix on the device (Newton-Schulz iteration on a hand-written fp64 MFMA GEMM: l <= 46000, 8 l^2 bytes of HBM kept,
 * four times that during set-up, ~(2 log2(lambda_max) + 14) l x l x l products) and a projection is two Q sweeps and one
 * dense symmetric matrix-vector product.  A (the arrays fos_create was given) is passed
 * again: the handle keeps only its device format.  No CG runs: fos_check_result.cgiter stays 0 and the host prints the table
 * without the cg column (HSDEStatus.jl:44-50,79).  fos_disable_direct returns to CG.  Sharded handles: the block form below only (see there).
 * l > 46000 (C3, C4, C5): the dense inverse does not fit; IndAffine([Q -I], 0) and AffinePlusLinear(Q, 0, 0, 1) being the same set (HSDE.jl:12-15 / :22), the
 * exact projection is then computed by the warm-started CG run to its tolerance floor l eps from the first call on (no 0.2^sqrt(i) schedule) -- the reference's
 * sparse factorisation is not rebuilt; fos_check_result.cgiter reports the CG iterations of that projection.
 * BLOCK-SEPARABLE operators (round 5; tried first, any size): when the columns of A fall into groups of at most 64 that share no row -- I + A'A block diagonal
 * with small blocks: a block-diagonal SDP with few variables per block (C4: 512 blocks of 32 columns) -- the same exact projection costs THREE KKT sweeps:
 * I + Q Q' = blkdiag(I + A'A, I + AA', delta) + a rank-3 border, (I + AA')^-1 = I - A (I + A'A)^-1 A', the small blocks inverted once on the host
 * (csrc/solver.cpp prox_affine_direct_block).  No CG, no l x l matrix; cgiter stays 0.
 * CONE-SHARDED handles (fos_comm_init / fos_set_host_allreduce / fos_peer_open*): the block form is the one that shards -- the diagonal blocks are local to the
 * rank that holds the columns, only the rank-3 border couples the ranks: THREE exchanges of <= 3 doubles per projection through the handle's transport (the
 * tau row of the first apply with the two border dots, the tau row of the result) and two more at set-up.  fos_enable_direct is then COLLECTIVE (call it on
 * every rank, after the transport is enabled); it takes the block form when every rank's columns group, and fails with FOS_EUNSUPPORTED on every rank otherwise
 * (no dense or CG-floor form there).  Row-sharded handles: FOS_EUNSUPPORTED.
 * fos_get_direct_mode: 0 = off, 1 = dense inverse, 2 = block form, 3 = CG at its tolerance floor (the host keeps the table's cg column then).
 * FOS_DIRECT_MODE=block|dense|cg (environment) forces a form. */
int fos_enable_direct(fos_handle h, const int64_t* colptr, const int64_t* rowval, const double* nzval);
int fos_disable_direct(fos_handle h);
int fos_get_direct_mode(fos_handle h, int32_t* mode);

/* getinitialvalue / option initx (solverwrapper.jl:10, HSDE.jl:40-47): z = 0, tau = kappa = 1 when z == NULL */
int fos_set_iterate(fos_handle h, const double* z);
int fos_get_iterate(fos_handle h, double* z);              /* the current x of `iterate` */
int fos_get_checked(fos_handle h, double* z);              /* the vector the last checkstatus saw (debug=2 history of
                                                              x,y,s: savedata, HSDEStatus.jl:133-136); valid until the next step */

/* ---- the hot loop: replaces Base.step + checkstatus inside iterate --------------------------------
 * fos_step runs outer iterations i = i_first, i_first+1, ... (1-based, solverwrapper.jl:23-29) entirely on
 * the device.  It returns after `count` iterations, or earlier right after an iteration on which
 * i % checki == 0 (the convergence check of S2!/step: gap.jl:56, gapa.jl:75, fista.jl:41), whichever comes
 * first.  *iters_done = iterations performed.  If the last iteration was a check iteration, *checked = 1
 * and *res holds the values (res->status != CONTINUE means `iterate` must break); else *checked = 0. */
int fos_step(fos_handle h, int64_t i_first, int64_t count, int64_t checki, double eps,
             int64_t* iters_done, int32_t* checked, fos_check_result* res);

/* LineSearchWrapper(alg; lsinterval) of src/wrappers/linesearch.jl:36-75 around GAP (AP, DR) or GAPA -- the algorithms with
 * support_linesearch == Val{:Fast} (gap.jl:89, gapa.jl:117).  After fos_set_alg: iterations i with i % lsinterval == 0 are no
 * longer a plain step but   tmp1 = x;  x = S2!(S1!(x)) (checkstatus inside S2! as usual);  res = x - tmp1;
 * for k = 0:30  alpha = 0.1 * 1.8^(k+1);  testres_k = || (tmp1 + alpha res) - S2!(S1!(tmp1 + alpha res)) ||;
 * x = tmp1 + alpha_best res  -- all on the device, every S1! a warm-started CG solve that advances the tolerance counter as in
 * the reference.  lsinterval = 0 switches the wrapper off; any other algorithm is refused (FOS_EUNSUPPORTED).
 * fos_linesearch_log: out34 = [ ||res||, testres_0..30, alpha_best, iteration ] of the last search -- what the reference
 * prints with its println calls (linesearch.jl:51,63,69). */
int fos_set_linesearch(fos_handle h, int64_t lsinterval);
int fos_linesearch_log(fos_handle h, double* out34);
/* LongstepWrapper(alg; longinterval, nsave) around the algorithm set last -- GAP (AP, DR), GAPA, FISTA or Dykstra (support_longstep in src/solvers) --
 * replaces src/wrappers/longstep.jl:5-63 (LongstepWrapper, init_algorithm!, step) and saveplanes.jl:5-35 (SavedPlanes, projectonnormals!): the last
 * nsave + 1 iterations of every longinterval save, per iteration, the half-plane through P_S1(x) with normal x - P_S1(x) (addprojeq, longstep.jl:65-79) and
 * the one of the second projection (addprojineq, :81-97), rows in the reference's order; behind the last of them the iterate is replaced by its projection onto
 * { first nsave + 1 rows as equalities, the others as inequalities C v >= d } (saveplanes.jl:17-28).  The reference solves that QP in the n variables with QPDAS
 * (BigFloat, because successive normals are nearly dependent); here its dual in the 2 (nsave + 1) multipliers is solved by enumeration of the active
 * inequalities -- the same unique point -- with the Gram products formed in double-double on the device and the small systems solved in 113-bit arithmetic.
 * nsave <= 15; longinterval >= nsave + 1; 0 switches the wrapper off.  out8 = iteration of the last projection, active inequalities, largest KKT violation of
 * the small dual, |x_new - x|, rows, candidate supports tried, failed, projections given up since fos_set_longstep.  failed = 1: no support passed the KKT
 * test within the budget -- 4096 candidate supports (FOS_LONG_MAX_SUPPORTS, read at fos_set_longstep; a count, so the same problem behaves the same on
 * every run), inconsistent or dependent planes -- : the iterate is left as the wrapped algorithm's step produced it and fos_step returns FOS_OK (the reference's QP
 * solver throws there); the host mirrors warn when out8[7] grows. */
int fos_set_longstep(fos_handle h, int64_t longinterval, int64_t nsave);
int fos_longstep_log(fos_handle h, double* out8);
/* GAPP(alpha, alpha1, alpha2; iproj) -- "projected GAP", src/solvers/gapproj.jl:5-81, the last row of the reference's solver table
 * (README.md:32-40): fos_set_alg(FOS_ALG_GAP, alpha, alpha1, alpha2, 0) and then fos_set_gapp(iproj > 0).  Iterations i with
 * i % iproj == 0 search 21 step lengths 2^k along P_S1(P_S2(P_S1 x)) - P_S1 x (gapproj.jl:34-62); out23 = the 21 test norms,
 * alpha_best, iteration of the last search (what gapproj.jl:51,57 print).  Single-GPU handles. */
int fos_set_gapp(fos_handle h, int64_t iproj);
int fos_gapp_log(fos_handle h, double* out23);

/* getsol(alg, data, x): one more S1 prox + S2 prox (gap.jl:82-87, gapa.jl:107-112, fista.jl:50-56); advances
 * the CG call counter like the reference.  z_out (N doubles) receives `guess`.  If force_check != 0 the
 * override check of solverwrapper.jl:31-34 is evaluated on the guess. */
int fos_getsol(fos_handle h, double* z_out, int32_t force_check, double eps, fos_check_result* res);

/* S1's persistent state -- CGdata.xinit (the CG warm start, affinepluslinear.jl:101-106,122) and the call counter
 * AffinePlusLinear.i (:66,114) -- for checkpoint/resume and for handing a steady-state point to the CPU baseline.
 * get: xinit (N doubles, may be NULL), *i = counter the NEXT prox! call will use, *firstrun = CGdata.firstrun.
 * set: installs xinit (firstrun becomes false) and the counter. */
int fos_get_affine_state(fos_handle h, double* xinit, int64_t* i, int32_t* firstrun);
int fos_set_affine_state(fos_handle h, const double* xinit, int64_t i);

/* The algorithm's own *Data struct, for checkpoint/resume and for handing a run over to another implementation mid-solve (the oracle in
 * tests/test_gpu_fullsize.py, bench.py's cpu_baseline): FISTAData.y, .xold, .t (fista.jl:15-25, updated at :39,:44-46);
 * DykstraData.p, .q (dykstra.jl:12-23, updated at :29,:33); GAPAData.alpha12 (gapa.jl:29,101).  GAP has none.
 * a, b: N doubles each in the reference layout -- FISTA: a = y, b = xold; Dykstra: a = p, b = q (NULL: skipped; both are zero behind fos_set_alg);
 * scal2 = [ t, alpha12 ].  set installs what is non-NULL; the CG warm start and call counter are fos_set_affine_state's. */
int fos_get_alg_state(fos_handle h, double* a, double* b, double* scal2);
int fos_set_alg_state(fos_handle h, const double* a, const double* b, const double* scal2);

int fos_get_cgiter(fos_handle h, int64_t* cgiter);         /* getcgiter(data), defaults.jl:25-30 */
int fos_get_alpha12(fos_handle h, double* alpha12);        /* GAPAData.alpha12 */
int fos_get_prox_count(fos_handle h, int64_t* i);          /* AffinePlusLinear.i (next call's counter) */

/* ---- fine-grained operator entry points (test boundary; each does H2D, compute, D2H) -------------
 * fos_q_apply:      mul!(Y, Q, B) / mul!(Y, transpose(Q), B), l = n+m+1   (HSDEAffine.jl:41-65)
 * fos_kkt_apply:    mul!(y, KKTMatrix(Q), x), 2l                          (affinepluslinear.jl:37-52)
 *                   == mul!(Y, HSDEMatrix(Q), B)                          (HSDEAffine.jl:131-147)
 * fos_cg_kkt:       conjugategradient!(x, KKTMatrix(Q), rhs, ...; tol, max_iters), x = warm start in/out;
 *                   *iters = return value                                 (conjugategradients.jl:31-55)
 * fos_prox_affine:  prox!(y, S1::AffinePlusLinear, x)  (stateful: i, xinit) (affinepluslinear.jl:83-126)
 * fos_hsdematrix_prox: prox!(y, HSDEMatrix(Q), x) with its own CGdata     (HSDEAffine.jl:105-126)
 * fos_prox_cones:   prox!(y, S2::DualConeProduct, x)                      (cones.jl:122-142)
 * fos_check:        checkstatus(status, z, override=true) values          (HSDEStatus.jl:27-63)           */
int fos_q_apply(fos_handle h, double* y, const double* x, int32_t transpose);
int fos_kkt_apply(fos_handle h, double* y, const double* x);
int fos_cg_kkt(fos_handle h, double* x, const double* rhs, double tol, int64_t max_iters, int64_t* iters);
int fos_prox_affine(fos_handle h, double* y, const double* x);
int fos_hsdematrix_prox(fos_handle h, double* y, const double* x);
int fos_prox_cones(fos_handle h, double* y, const double* x);
int fos_check(fos_handle h, const double* z, double eps, fos_check_result* res);

/* ---- measurement ---------------------------------------------------------------------------------
 * fos_profile: when enabled, every launch of the dominant kernel (the fused dual-right-hand-side KKT
 *   SpMV of the CG loop) is bracketed by HIP events on the solver's stream.
 * fos_profile_read: synchronises and returns launches, summed kernel milliseconds, and the ALGORITHMIC
 *   bytes of one launch (B_kkt,min = 24 nnz + 4(m+n+2) + 32(m+n)); resets the counters.
 * fos_bench_kkt: `reps` back-to-back KKT-apply launches on device-resident vectors; total ms by HIP events. */
/* fos_operator_stats: the 12 format statistics of fos_host_stacked_spmv (below) for the operator this handle holds on the
 * device -- what the sweep actually streams: 8 B per stored value, 4 B per stored column index, 32 B per block, 16 B per
 * partial-sum slot written and read again. */
int fos_operator_stats(fos_handle h, int64_t* stats12);
/* Window-panel storage (gather-bound random-sparse operators; csrc/fos_internal.hpp WinPanel): panels, (panel, window)
 * segments, 64-row slices, stored entries (8 B value + 2 B window offset each; 2 B row id per slice lane).  All zero when the
 * operator is held in row blocks / dual tiles (then fos_operator_stats describes it). */
int fos_window_stats(fos_handle h, int64_t* stats4);
int fos_profile(fos_handle h, int32_t enable);      /* 0: off; 1: every launch; N > 1: every N-th launch (sampling: an event
                                                        pair per launch costs ~5 % of a C4 step) */
int fos_get_cg_total(fos_handle h, int64_t* total);  /* CG iterations run since fos_create (getcgiter summed) */
int fos_profile_read(fos_handle h, int64_t* launches, double* total_ms, double* bytes_per_launch);
/* The same records split by class of launch group (arrays of FOS_PROF_CLASSES entries): the KKT sweep of a CG iteration,
 * the batched PSD projection of one cone-prox call (cones.jl:89-94 over all PSD cones), the CG vector update(s) of an
 * iteration (arrays of FOS_PROF_CLASSES = 5 entries).  Resets the records like fos_profile_read. */
#define FOS_PROF_KKT 0
#define FOS_PROF_PSD 1
#define FOS_PROF_CGVEC 2
/* everything else an outer iteration launches -- the start of a CG solve (start sweep + r_0 kernel), the relaxations, the elementwise / SOC / Exp
 * cones, the step's last pass -- bracketed group by group in every prof_period-th OUTER ITERATION; for this class launches[] counts the sampled
 * iterations, so total_ms / launches = milliseconds per outer iteration */
#define FOS_PROF_OTHER 3
/* a whole CG solve run as ONE launch (FOS_CG_RESIDENT): every prof_period-th solve is bracketed */
#define FOS_PROF_RESIDENT 4
#define FOS_PROF_CLASSES 5
int fos_profile_read_classes(fos_handle h, int64_t* launches5, double* total_ms5);
/* fos_bench_cg_chain: `iters` CG iterations (conjugategradients.jl:37-51; sweep + update [+ p update]) on the current
 * iterate as right-hand side with the stop test disabled, enqueued eagerly (use_graph = 0) or captured ONCE into a hipGraph
 * and replayed `reps` times (use_graph = 1); *ms_per_iter by HIP events.  Measurement only: it answers whether graph replay
 * shortens the dependent-launch chain of a CG solve. */
int fos_bench_cg_chain(fos_handle h, int32_t iters, int32_t reps, int32_t use_graph, double* ms_per_iter);
int fos_bench_kkt(fos_handle h, int32_t reps, double* total_ms);
/* PSD kernel diagnostics (cones.jl:89-94 -> IndPSD): collect_stats != 0 makes every projection record the number of Jacobi
 * sweeps each (cone, copy) matrix took (fos_psd_stats: 2 x #PSD cones entries, order (cone, part)); phase_limit 1..4 ends
 * the kernel after load+shift / warm-start product / sweeps / weights+basis -- results are then WRONG, timing use only. */
int fos_psd_debug(fos_handle h, int32_t collect_stats, int32_t phase_limit);
int fos_psd_stats(fos_handle h, int32_t* sweeps, int64_t cap, int64_t* count);
int fos_sync(fos_handle h);

/* Host-only self check of the device operator format (no GPU needed): builds the block format of
 * S = [[0,A'],[A,0]] exactly as fos_create does and multiplies out = S * v on the HOST by walking the blocks the way the
 * kernel does (dual tiles and the deferred-row pass included).  stats (12 x int64, may be NULL): blocks, ELL, LDS, LONG,
 * run-compressed blocks, stored values, stored column indices, wavefronts, dual tiles, partial-sum slots, deferred rows,
 * matrix entries held in tiles.  Used by the CPU test-suite. */
int fos_host_stacked_spmv(int64_t m, int64_t n, const int64_t* colptr, const int64_t* rowval, const double* nzval,
                          const double* v, double* out, int32_t spmv_workgroups, int32_t resident_waves, int64_t* stats);
/* Host-only check of FOS_CG_RESIDENT (no GPU needed): builds the operator as fos_create does on a 256-CU device, plans the resident solve for at most
 * `gmax` workgroups (stats8 as fos_resident_stats) and -- x != NULL -- runs conjugategradient!(x, KKTMatrix(Q), rhs; tol, max_iters)
 * (conjugategradients.jl:31-55) on the HOST by walking the plan the way the kernel does: tiles per workgroup, partial column sums per workgroup, the
 * unit's totals, per-workgroup partial sums of the four reductions.  x, rhs: plain N = 2 (n + m + 1) vectors.  Returns FOS_EUNSUPPORTED (x != NULL)
 * when the operator does not qualify; with x == NULL only the plan is made.  Used by the CPU test-suite. */
int fos_host_resident_cg(int64_t m, int64_t n, const int64_t* colptr, const int64_t* rowval, const double* nzval, const double* b, const double* c,
                         int32_t gmax, double* x, const double* rhs, double tol, int64_t max_iters, int64_t* iters, int64_t* stats8);
/* the same with the storage choice forced: window_mode 0 = row blocks / dual tiles, 1 = window panels (any size), -1 = as
 * fos_create decides; stats16 (may be NULL) = the 12 statistics above followed by the 4 of fos_window_stats */
int fos_host_stacked_spmv_mode(int64_t m, int64_t n, const int64_t* colptr, const int64_t* rowval, const double* nzval,
                               const double* v, double* out, int32_t window_mode, int64_t* stats16);

/* tuning knobs (0 keeps the default): workgroups of the SpMV grid, CG iterations enqueued per host poll;
 * fuse_p: -1 keeps the choice made at fos_create, 0 / 1 force the three- / two-launch CG iteration (see fos_bench_cg_chain) */
int fos_set_tuning(fos_handle h, int32_t spmv_workgroups, int32_t cg_chunk, int32_t fuse_p);

/* Which recurrence conjugategradient! (src/utilities/conjugategradients.jl:31-55) runs in.  All four produce the same Krylov
 * iterates in exact arithmetic, count iterations as the reference does and apply its stop test (norm(r) <= tol || iter >=
 * max_iters) to the same recursively updated residual; they differ in launches and reduction points per iteration:
 *   FOS_CG_REFERENCE      the reference's recurrence: sweep Ap = M p -> alpha, x, r update -> beta, p update   (3 launches, 2 reduction points)
 *   FOS_CG_FUSED_P        the p update rides on the next sweep                                                (2 launches, 2 reduction points; measured slower)
 *   FOS_CG_MERGED_SWEEP   merged-reduction (Chronopoulos-Gear) recurrence: s = M p by recurrence from w = M r, one
 *                         reduction point carrying r.r and w.r; the sweep closes the iteration            (2 launches, single GPU only)
 *   FOS_CG_MERGED_UPDATE  the same, the update kernel closes the iteration: ONE exchange of four doubles per iteration
 *                         when sharded; the sweep of the last iteration runs for nothing                  (2 launches; default for sharded handles)
 *   FOS_CG_RESIDENT       the arithmetic of FOS_CG_MERGED_UPDATE as ONE launch per SOLVE, for operators that are nothing but dual tiles of a
 *                         block-separable A; the four sums of an iteration cross the workgroups (and, sharded, the GPUs) as self-validating
 *                         words -- no grid barrier.  Two forms, chosen at fos_create (fos_resident_stats says which):
 *                           registers  the tiles AND the CG vectors stay on chip from the first to the last iteration (a workgroup's share fits
 *                                      the register file: up to 21 tiles -- a block-diagonal SDP on enough GPUs, the eighth of C4 = 133 KB per CU);
 *                           streamed   the tiles are re-read once per iteration, everything else stays on chip (r, w, x in registers, p, s in LDS):
 *                                      one pass does the sweep, its reductions and the iteration's vector updates (the whole of C4; whole
 *                                      consecutive units per workgroup, or a unit split over up to four workgroups).
 *                         Sharded: device or host-pinned mailboxes only.  fos_set_cg_variant returns FOS_EUNSUPPORTED when the operator does
 *                         not qualify (fos_last_error says why).  The default of sharded handles on the mailboxes whenever every rank's shard
 *                         qualifies, and of single handles whose streamed plan fills at least half of the device (FOS_RESIDENT_DEFAULT=0: never).
 * variant = -1 restores the handle's default (on one GPU: FOS_CG_REFERENCE, or the cases named above). */
#define FOS_CG_REFERENCE     0
#define FOS_CG_FUSED_P       1
#define FOS_CG_MERGED_SWEEP  2
#define FOS_CG_MERGED_UPDATE 3
#define FOS_CG_RESIDENT      4
int fos_set_cg_variant(fos_handle h, int32_t variant);
/* The plan of FOS_CG_RESIDENT for this handle (conjugategradients.jl:31-55 as one launch): stats8 = {qualifies (0 / 1), workgroups, wavefronts
 * per workgroup, tiles per wavefront (NEGATIVE: the streamed form -- tiles re-read from memory every iteration, whole units per workgroup, for operators
 * whose tiles exceed the register file), units (runs of columns = diagonal blocks), most tiles in one workgroup, steps per tile (32 / 64),
 * every rank qualifies (sharded handles: the vote of fos_peer_enable; else = qualifies)}. */
int fos_resident_stats(fos_handle h, int64_t* stats8);
int fos_get_cg_variant(fos_handle h, int32_t* variant);   /* the variant the next projection runs (defaults resolved) */

/* test hooks.  FOS_DEBUG_PUPDATE_DELAY: every workgroup but the first of the kernel that closes a CG iteration waits `value`
 * ticks of the 100 MHz clock at entry (tests/test_gpu_parity.py: a late workgroup must still apply the last x update). */
#define FOS_DEBUG_PUPDATE_DELAY 1
int fos_debug_set(fos_handle h, int32_t what, int64_t value);

/* ------------------------------------------------------------------------------------------------------------------
 * Feasibility form: `Feasibility(S1, S2, n)` -- find a point of S1 n S2 with the same algorithms.
 * Replaces, for device-resident sets, src/problemforms/Feasibility/Feasibility.jl:2-6,52-68 (problem, solve!, zeros(n) start,
 * populate_solution) and FeasibilityStatus.jl:32-72 (checkstatus: err = norm(prev - z) every checki-th iteration, :Optimal when
 * err <= eps; prev is refreshed at EVERY iteration and starts as NaN).  The reference accepts any two ProximalOperators objects;
 * device-resident are the two its own test uses (test/testfeasibility.jl:9-10) and its cone stack, anything else is a host callback
 * (fos_feas_set_callback):
 *   fos_feas_set_affine  IndAffine(A, b): A m x n ROW-major, full row rank, n <= 46 000; exact projection x - A'(A A')^-1 (A x - b)
 *                        through a one-time dense inverse formed on the device (Newton-Schulz on the fp64 MFMA GEMM);
 *   fos_feas_set_affine_sparse  IndAffine(A, b) with A sparse (CSC, 1-based as fos_create's), full row rank, any n: nothing dense is formed; the
 *                        projection is computed by conjugate gradients on the row-scaled normal equations A A' d = A y - b, warm-started from
 *                        the previous projection's multipliers, and ends when the RECOMPUTED residual A y - b is at the rounding level of its own
 *                        evaluation (16 eps | |A||y| + |b| |) -- y - x is in the range of A' by construction (affine_sparse.hip).  Fails
 *                        loudly (FOS_EINVAL) when that level cannot be reached (A without full row rank).  fos_feas_affine_stats: iteration counts.
 *   fos_feas_set_box     IndBox(lo, hi), scalar bounds (fos_feas_set_box_arrays: array bounds), +-INFINITY allowed;
 *   fos_feas_set_cones   the reference's own ConeProduct (src/cones.jl), any of its nine cone types.
 * `which` = 1 | 2 (S1, S2).  Steps: gap.jl:42-87 (GAP / DR / AP), gapa.jl:61-112, fista.jl:28-56, dykstra.jl:25-44; LineSearchWrapper. */
typedef struct fos_feas* fos_feas_handle;
int fos_feas_create(int64_t n, int32_t device, fos_feas_handle* out);
int fos_feas_destroy(fos_feas_handle h);
int fos_feas_set_affine(fos_feas_handle h, int32_t which, int64_t m, const double* A, const double* b);
int fos_feas_set_affine_sparse(fos_feas_handle h, int32_t which, int64_t m, const int64_t* colptr, const int64_t* rowval, const double* nzval, const double* b);
int fos_feas_affine_stats(fos_feas_handle h, int32_t which, double* out8);
int fos_feas_set_box(fos_feas_handle h, int32_t which, double lo, double hi);
int fos_feas_set_box_arrays(fos_feas_handle h, int32_t which, const double* lo, const double* hi);     /* IndBox with array bounds (n each) */
/* ConeProduct (src/cones.jl:31-94): ncones cones of type[i] (FOS_CONE_*) and len[i] entries, in order, contiguous, covering all n entries;
 * projected by the batched cone kernels of the HSDE path (PSD cones warm-started from one projection to the next) */
int fos_feas_set_cones(fos_feas_handle h, int32_t which, int64_t ncones, const int32_t* type, const int64_t* len);
/* Any OTHER ProximableFunction (Feasibility.jl:2-6 takes any two; the reference's step calls prox!(y, S, x), gap.jl:47,56): evaluated by
 * the CALLER on host vectors -- fn(ctx, n, x, y) fills y = prox_S(x), returns 0 (anything else aborts the step with FOS_EINVAL).  The
 * iterate crosses the host link twice per projection (2 x 8n bytes through pinned buffers, one stream synchronisation); everything else
 * of the iteration -- the other set, the relaxations, the status test -- stays on the device.  fn and ctx must outlive the handle. */
typedef int32_t (*fos_prox_fn)(void* ctx, int64_t n, const double* x, double* y);
int fos_feas_set_callback(fos_feas_handle h, int32_t which, fos_prox_fn fn, void* ctx);
int fos_feas_set_alg(fos_feas_handle h, int32_t alg, double alpha, double alpha1, double alpha2, double beta);
/* GAPP ("projected GAP", src/solvers/gapproj.jl:5-81; test/testfeasibility.jl:36): GAP whose every iproj-th iteration searches 21 step
 * lengths 2^k along P_S1(P_S2(P_S1 x)) - P_S1 x.  out23 = the 21 test norms, alpha_best, iteration of the last search. */
int fos_feas_set_gapp(fos_feas_handle h, double alpha, double alpha1, double alpha2, int64_t iproj);
int fos_feas_gapp_log(fos_feas_handle h, double* out23);
/* LineSearchWrapper(GAP | GAPA; lsinterval) around the algorithm set last (wrappers/linesearch.jl:36-75; 0 switches it off), and what
 * the reference prints during the last search: out34 = normres, 31 test residuals, alpha_best, iteration (as fos_linesearch_log) */
int fos_feas_set_linesearch(fos_feas_handle h, int64_t lsinterval);
/* LongstepWrapper on this form (fos_set_longstep above: same planes, same projection) */
int fos_feas_set_longstep(fos_feas_handle h, int64_t longinterval, int64_t nsave);
int fos_feas_longstep_log(fos_feas_handle h, double* out8);
int fos_feas_linesearch_log(fos_feas_handle h, double* out34);
/* x = x0 (NULL: zeros(n), Feasibility.jl:58) and the algorithm state of a fresh init_algorithm! */
int fos_feas_set_iterate(fos_feas_handle h, const double* x0);
/* iterations first_iter .. first_iter + niter - 1 of `iterate` (solverwrapper.jl:23-29); stops behind the iteration whose check
 * found err <= eps.  done = iterations run; status (FOS_STATUS_CONTINUE | FOS_STATUS_OPTIMAL), err and checked describe the last one. */
int fos_feas_step(fos_feas_handle h, int64_t first_iter, int64_t niter, int64_t checki, double eps, int64_t* done, int32_t* status, double* err,
                  int32_t* checked);
/* getsol = P_S2(P_S1(x)) (gap.jl:82-87 ...); force_check: checkstatus(status, guess, override = true) of solverwrapper.jl:31-33 */
int fos_feas_getsol(fos_feas_handle h, double* sol, int32_t force_check, double eps, int32_t* status, double* err);
int fos_feas_get_iterate(fos_feas_handle h, double* x);
/* prox!(y, S_which, x) on host vectors (the ProximableFunction protocol; test entry) */
int fos_feas_prox(fos_feas_handle h, int32_t which, const double* x, double* y);
/* diagnostics: GAPA's alpha12; per set the Newton-Schulz steps and the final max |G X - I| of an IndAffine set-up (arrays of 2) */
int fos_feas_info(fos_feas_handle h, double* alpha12, int32_t* ns_iters, double* ns_resid);

#ifdef __cplusplus
}
#endif
#endif /* FOSHIP_H */
