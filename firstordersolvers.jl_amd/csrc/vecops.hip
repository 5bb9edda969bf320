// Streaming vector kernels of the hot path: CG updates with in-pass reductions, relaxation / extrapolation
// passes, layout conversion, elementwise cones and the batched second-order-cone projection.
// All of them are HBM-bound streams over l = n+m+1 `double2` elements (16 B per lane per access).
//
// FP contraction is OFF in this file: the reference evaluates e.g. `y .= a1.*y .+ (1-a1).*x` (gap.jl:48) as
// two multiplies and an add; keeping that keeps the iterates as close to the reference as the reduction
// order allows.
#include "dev_common.hpp"

#pragma clang fp contract(off)

namespace fos {

constexpr int VEC_THREADS = 256;
constexpr int PRE_NPROD = 16;        // workgroups of cg_update_kernel that add the sweep's records for all the others
constexpr int FIN_THREADS = 1024;

template <int NACC>
__device__ __forceinline__ void block_reduce_store(double (&acc)[NACC], double* out) {
    __shared__ double smem[4 * NACC];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
#pragma unroll
    for (int a = 0; a < NACC; ++a) {
        double v = wave_sum(acc[a]);
        if (lane == 0) smem[wave * NACC + a] = v;
    }
    __syncthreads();
    if (tid < NACC) out[tid] = (smem[tid] + smem[NACC + tid]) + (smem[2 * NACC + tid] + smem[3 * NACC + tid]);
}

// ------------------------------------------------------------------------------------------------ CG

// r = rhs - Ap ; p = r ; partial r.r over the non-tau rows          conjugategradients.jl:33-35
__global__ __launch_bounds__(VEC_THREADS) void cg_init_kernel(int64_t l, const d2* __restrict__ rhs, const d2* __restrict__ Ap,
                                                              d2* __restrict__ r, d2* __restrict__ p, double* __restrict__ partials,
                                                              int64_t acc_from) {
    double acc[1] = {0.0};
    for (int64_t i = blockIdx.x * (int64_t)VEC_THREADS + threadIdx.x; i < l; i += (int64_t)gridDim.x * VEC_THREADS) {
        const d2 b = rhs[i], a = Ap[i];
        const d2 ri = make_double2(b.x - a.x, b.y - a.y);
        r[i] = ri;
        p[i] = ri;
        if (i != l - 1 && i >= acc_from) acc[0] += ri.x * ri.x + ri.y * ri.y;       // (acc_from > 0: replicated entries counted elsewhere)
    }
    block_reduce_store<1>(acc, partials + blockIdx.x);
}

__global__ __launch_bounds__(FIN_THREADS) void cg_init_finalize_kernel(const double* __restrict__ partials, int count,
                                                                       const double* __restrict__ reduced, int from_reduced,
                                                                       const d2* __restrict__ r, int64_t l, DevState* st, double tol, int maxit) {
    __shared__ double sums[1];
    if (from_reduced) { if (threadIdx.x == 0) sums[0] = reduced[0]; __syncthreads(); }
    else reduce_partials<1>(partials, count, sums);
    if (threadIdx.x == 0) {
        const d2 rt = r[l - 1];
        st->rn = sums[0] + (rt.x * rt.x + rt.y * rt.y);     // rn = dot(r,r)      :35
        st->rn2[0] = st->rn;                                 // RN[0]: iteration 1's alpha reads slot 0 (dev_common.hpp)
        st->rn2[1] = 0.0;
        st->rn_old = 0.0;
        st->iter = 1;                                        // :36
        st->done = 0;
        st->hit_max = 0;
        st->tol = tol;
        st->maxit = maxit;
    }
}

// The sweep's three sums (Ap.p without the tau rows, [c;b].p1, [c;b].p2) -> sums[0..2] (shared), the same bits in every workgroup.
// Two pieces, so that a kernel can request the records together with everything else it needs and add them later:
// SweepRecs::request (loads only) and SweepRecs::finish.
struct SweepRecs {
    PartialRegs<3> regs;
    const double* src;
    int cnt;
    bool producer, plain;
    __device__ __forceinline__ void request(const double* __restrict__ kkt_partials, int nkkt, int from_reduced, const double* pre) {
        plain = !from_reduced && pre == nullptr;
        producer = !from_reduced && pre != nullptr && blockIdx.x < PRE_NPROD;
        src = kkt_partials; cnt = nkkt;
        if (producer) {
            const int per = (nkkt + PRE_NPROD - 1) / PRE_NPROD;
            const int lo = min((int)blockIdx.x * per, nkkt);
            src = kkt_partials + 3 * (size_t)lo; cnt = min(per, nkkt - lo);
        }
        if (plain || producer) regs.load(src, cnt);
    }
    __device__ __forceinline__ void finish(double* sums, const double* __restrict__ reduced, int from_reduced, double* __restrict__ pre,
                                           uint32_t pre_seq, DevState* st) {
        if (from_reduced) { if (threadIdx.x < 3) sums[threadIdx.x] = reduced[threadIdx.x]; __syncthreads(); }
        else if (pre != nullptr) {
            // Thousands of sweep records (C4: 4 224): every workgroup adding them all again cost ~6 us of this kernel.  Instead the
            // first PRE_NPROD workgroups -- always dispatched first, so a waiting workgroup can never keep them from running --
            // add a slice each and publish their three sums; everybody waits for those (cache-bypassing loads: no atomics, nothing
            // serialises) and adds the PRE_NPROD partial results in order.  Fixed order, same bits in every workgroup.
            // The partial results travel as self-validating words, (launch number << 32) | half a double -- the idea of the peer
            // mailboxes: no separate flag, no wait for the data's acknowledgement in front of it, and the consumers' polls ARE
            // the data loads.  Chain: producer's records (one round trip) -> its stores -> a consumer's poll; it was five hops.
            unsigned long long* words = reinterpret_cast<unsigned long long*>(pre);
            __shared__ uint32_t pre_halves[6 * PRE_NPROD];
            if (producer) {
                double acc[3];
                regs.sum(src, cnt, acc);
                partials_combine<3>(acc, sums);
                if (threadIdx.x < 6) {
                    const unsigned long long bits = (unsigned long long)__double_as_longlong(sums[threadIdx.x >> 1]);
                    const uint32_t half = (threadIdx.x & 1) ? (uint32_t)(bits >> 32) : (uint32_t)bits;
                    __hip_atomic_store(words + 6 * blockIdx.x + threadIdx.x, ((unsigned long long)pre_seq << 32) | half, __ATOMIC_RELAXED,
                                       __HIP_MEMORY_SCOPE_AGENT);
                }
            }
            if (threadIdx.x < 6 * PRE_NPROD) {
                const long long t0 = wall_clock64();
                unsigned long long w;
                while ((uint32_t)((w = __hip_atomic_load(words + threadIdx.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) >> 32) != pre_seq) {
                    __builtin_amdgcn_s_sleep(1);
                    if (wall_clock64() - t0 > 500000000LL) { st->bar_failed = 1; st->done = 1; break; }      // 5 s of the 100 MHz clock: give up, flag the state
                }
                pre_halves[threadIdx.x] = (uint32_t)w;
            }
            __syncthreads();
            if (threadIdx.x < 3) {
                double sum = 0.0;
                for (int b = 0; b < PRE_NPROD; ++b) {
                    const unsigned long long lo = pre_halves[6 * b + 2 * threadIdx.x], hi = pre_halves[6 * b + 2 * threadIdx.x + 1];
                    sum += __longlong_as_double((long long)((hi << 32) | lo));
                }
                sums[threadIdx.x] = sum;
            }
            __syncthreads();
        }
        else {
            double acc[3];
            regs.sum(src, cnt, acc);
            partials_combine<3>(acc, sums);
        }
    }
};
__device__ __forceinline__ void sweep_sums3(double* sums, const double* __restrict__ kkt_partials, int nkkt, const double* __restrict__ reduced,
                                            int from_reduced, double* __restrict__ pre, uint32_t pre_seq, DevState* st) {
    SweepRecs rec;
    rec.request(kkt_partials, nkkt, from_reduced, pre);
    rec.finish(sums, reduced, from_reduced, pre, pre_seq, st);
}

// Second launch of a CG iteration.  EVERY workgroup reduces the sweep's 3 x nkkt partial sums in the same fixed order, finishes
// the tau rows of Ap = M p and alpha = rn / (Ap.p) itself (so no workgroup waits for another), then updates its slice:
// x += alpha p ; r -= alpha Ap ; partial r.r.  Workgroup 0 stores Ap[tau], pAp and alpha.     conjugategradients.jl:39-41,46
// DEF (operators with dual tiles): rows of Ap whose sum the sweep left spread over partial slots are finished HERE -- `lpr`
// lanes add a row's slot list in list order (as kkt2_deferred_kernel does for the stand-alone applies), run the row epilogue
// of EpiKkt and update x, r of that row at once; their share of Ap.p is already in the sweep's sums (EpiKkt, FOLD).
// XUPD = false (the p update is a launch of its own): x += alpha p moves into that launch, which reads p anyway -- this kernel
// then touches Ap, r (+ p, x only for the slot-spread rows): 48 instead of 96 bytes per element.
template <bool DEF, bool FOLD, bool XUPD>
__global__ __launch_bounds__(VEC_THREADS) void cg_update_kernel(int64_t l, d2* __restrict__ x, d2* __restrict__ r,
                                                                const d2* __restrict__ p, d2* __restrict__ Ap,
                                                                DevState* st, const double* __restrict__ kkt_partials, int nkkt,
                                                                const double* __restrict__ reduced, int from_reduced, int j,
                                                                double* __restrict__ partials, DevBlkCsr S, const double* __restrict__ cb, int n,
                                                                const uint32_t* __restrict__ def_mask, PeerBox pb, uint32_t seq_base, int count_repl,
                                                                double* __restrict__ pre, uint32_t pre_seq, int ndb_arg, int dlpr) {
    // ROLES (ndb > 0; operators with few slot-spread rows and long vectors, C4): the first ndb workgroups finish ONLY the rows
    // spread over slots (dlpr lanes per row), the others ONLY stream -- the slot lists are a chain of dependent loads that every
    // thread used to walk through before its first stream element; now the stream runs beside it.
    const int ndb = DEF ? ndb_arg : 0;
    const bool def_role = DEF && ndb > 0 && (int)blockIdx.x < ndb;
    const int nsblk = ndb > 0 ? (int)gridDim.x - ndb : (int)gridDim.x;
    const int sblk = ndb > 0 ? (def_role ? 0 : (int)blockIdx.x - ndb) : (int)blockIdx.x;
    // the first element of this thread's slice is requested BEFORE the scalar prologue (two dependent round trips and two
    // barriers): on small operators the prologue's latency, not bandwidth, is what this kernel costs
    const int64_t stride = (int64_t)nsblk * VEC_THREADS;
    const int64_t i0 = sblk * (int64_t)VEC_THREADS + threadIdx.x;
    // Everything the scalar prologue needs is requested FIRST and together -- the gate, the tau element of p, r.r of the previous
    // iteration (all stored by EARLIER launches) and the sweep's records: one round trip where there were three -- and behind it
    // the first elements of the thread's slice with their mask words (a slot-spread row's elements are read for nothing, but no
    // element waits for a mask word: with the mask tested first every trip of the stream was two dependent round trips).
    SweepRecs rec;
    rec.request(kkt_partials, nkkt, from_reduced, pre);
    const int done = st->done, xfail = FOLD ? st->xchg_failed : 0;
    const d2 pt = p[l - 1];
    const double rn_prev = st->rn2[(j - 1) & 1];
    constexpr int NPF = 4;                               // elements requested in front of the prologue (at most)
    const int npf = ((S.dbg_flags >> 8) & 7) ? ((S.dbg_flags >> 8) & 7) : NPF;
    bool eh[NPF];
    uint32_t em[NPF];
    d2 ea[NPF], er[NPF], ep[NPF], ex[NPF];
#pragma unroll
    for (int k = 0; k < NPF; ++k) {
        const int64_t ik = i0 + k * stride;
        eh[k] = !def_role && k < npf && ik < l;
        ea[k] = er[k] = ep[k] = ex[k] = make_double2(0.0, 0.0);
        em[k] = 0u;
        if (eh[k]) {
            if constexpr (DEF) em[k] = def_mask[ik >> 5];
            ea[k] = Ap[ik]; er[k] = r[ik];
            if constexpr (XUPD) { ep[k] = p[ik]; ex[k] = x[ik]; }
        }
    }
    if (done) return;
    if (xfail) return;
    __shared__ double sums[3];
    rec.finish(sums, reduced, from_reduced, pre, pre_seq, st);
    if constexpr (FOLD) {
        if (!peer_fold_sum<3>(pb, seq_base + 2u * (uint32_t)j, sums, st)) return;
    }
    const double S1 = sums[0], T1 = sums[1], T2 = sums[2];
    if ((S.dbg_flags & 16) && blockIdx.x != 0) return;                                                                   // (timing experiment)
    const double at1 = pt.x + T2;            // p1_tau - (Q p2)_tau ,  (Q v)_tau = -[c;b].v        HSDEAffine.jl:57
    const double at2 = -T1 - pt.y;           // (Q p1)_tau - p2_tau
    const double pAp = S1 + (at1 * pt.x + at2 * pt.y);
    const double alpha = rn_prev / pAp;
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        Ap[l - 1] = make_double2(at1, at2);
        st->pAp = pAp;
        st->alpha = alpha;
    }
    double acc[1] = {0.0};
    if constexpr (DEF) {
      if (ndb == 0 || def_role) {
        const d2* __restrict__ slots = reinterpret_cast<const d2*>(S.slots_rd);
        const int lpr = ndb > 0 ? dlpr : S.def_lpr, sh = 31 - __clz(lpr);
        const int rows_per_pass = ((ndb > 0 ? ndb : (int)gridDim.x) * VEC_THREADS) >> sh;
        const int lig = threadIdx.x & (lpr - 1);
        const int npass = (S.ndef + rows_per_pass - 1) / rows_per_pass;       // uniform trip count: the DPP sums need full waves
        int q = (blockIdx.x * VEC_THREADS + threadIdx.x) >> sh;
        if (S.dbg_flags & 4) { block_reduce_store<1>(acc, partials + blockIdx.x); return; }      // (timing experiment: rows left unfinished)
        for (int pass = 0; pass < npass; ++pass, q += rows_per_pass) {
            const bool ok = q < S.ndef;
            DefRow dr{};
            if (ok) dr = ld_defrow(S.def_rec + q);
            const int row = dr.row;
            const bool own = ok && lig == 0;
            d2 pi = make_double2(0.0, 0.0), xi = pi, ri = pi;
            double c = 0.0;
            if (own) { pi = p[row]; ri = r[row]; c = cb[row]; if constexpr (XUPD) xi = x[row]; }
            double u1 = 0.0, u2 = 0.0;
            if (ok) slot_list_sum(slots, S.def_idx, dr, lig, lpr, u1, u2);
            u1 = group_sum(u1, lpr);
            u2 = group_sum(u2, lpr);
            if (own) {
                double q1, q2;                                  // EpiKkt::row (kernels.hip)
                if (row < n) { q1 = u1 + pt.x * c; q2 = u2 + pt.y * c; }
                else { q1 = -(u1 - pt.x * c); q2 = -(u2 - pt.y * c); }
                const double a1 = pi.x - q2, a2 = q1 - pi.y;
                if constexpr (XUPD) { xi.x += alpha * pi.x; xi.y += alpha * pi.y; x[row] = xi; }
                ri.x -= alpha * a1; ri.y -= alpha * a2;
                r[row] = ri;
                if (count_repl || row >= n) acc[0] += ri.x * ri.x + ri.y * ri.y;       // (row-sharded: the replicated rows -- those of A' -- are counted by one rank)
            }
        }
      }
    }
    if (def_role) { block_reduce_store<1>(acc, partials + blockIdx.x); return; }
#pragma unroll
    for (int k = 0; k < NPF; ++k) {
        const int64_t ik = i0 + k * stride;
        if (!eh[k] || ((em[k] >> (ik & 31)) & 1u)) continue;
        if (ik == l - 1) ea[k] = make_double2(at1, at2);
        if constexpr (XUPD) { ex[k].x += alpha * ep[k].x; ex[k].y += alpha * ep[k].y; x[ik] = ex[k]; }
        er[k].x -= alpha * ea[k].x; er[k].y -= alpha * ea[k].y;
        r[ik] = er[k];
        if (ik != l - 1) acc[0] += er[k].x * er[k].x + er[k].y * er[k].y;
    }
    for (int64_t i = i0 + npf * stride; i < l; i += 2 * stride) {           // two elements per trip, every request of a trip in flight at once
        const int64_t ie[2] = {i, i + stride};
        bool on[2] = {true, ie[1] < l};
        uint32_t mw[2] = {0u, 0u};
        d2 ai[2], ri[2], pi[2], xi[2];
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            ai[q] = ri[q] = pi[q] = xi[q] = make_double2(0.0, 0.0);
            if (on[q]) {
                if constexpr (DEF) mw[q] = def_mask[ie[q] >> 5];
                ai[q] = Ap[ie[q]];
                ri[q] = r[ie[q]];
                if constexpr (XUPD) { pi[q] = p[ie[q]]; xi[q] = x[ie[q]]; }
            }
        }
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            if ((mw[q] >> (ie[q] & 31)) & 1u) on[q] = false;
            if (!on[q]) continue;
            if (ie[q] == l - 1) ai[q] = make_double2(at1, at2);
            if constexpr (XUPD) { xi[q].x += alpha * pi[q].x; xi[q].y += alpha * pi[q].y; x[ie[q]] = xi[q]; }
            ri[q].x -= alpha * ai[q].x; ri[q].y -= alpha * ai[q].y;
            r[ie[q]] = ri[q];
            if (ie[q] != l - 1) acc[0] += ri[q].x * ri[q].x + ri[q].y * ri[q].y;
        }
    }
    block_reduce_store<1>(acc, partials + blockIdx.x);
}
static int pre_min_records() {
    static const int v = getenv("FOS_PRE_MIN") ? atoi(getenv("FOS_PRE_MIN")) : 2048;
    return v;
}
void launch_cg_update(const LaunchCtx& c, const CgIter& it, double2* x, double2* r, double2* Ap, int kkt_from_reduced) {
    // r.r partials go behind the KKT partials (both live in c.partials); inside a CG iteration the sweep leaves c.S.nwg records at 0
    double* rr_out = c.partials + 3 * (size_t)PART_CAP;
    const PeerBox pb = it.fold ? *it.fold : PeerBox{};
    dim3 grid(c.cg_blocks), block(VEC_THREADS);
    // many sweep records and enough workgroups: the first PRE_NPROD of them add the records for all (cg_update_kernel)
    double* pre = (c.pre && !it.fold && !kkt_from_reduced && c.S.nwg >= pre_min_records() && c.cg_blocks >= 4 * PRE_NPROD) ? c.pre : nullptr;
    // roles (cg_update_kernel): a quarter of the grid at most for the slot-spread rows, with as many lanes per row as that allows
    int ndb = 0, dlpr = 1;
    static const bool split_env = !(getenv("FOS_UPD_SPLIT") && atoi(getenv("FOS_UPD_SPLIT")) == 0);
    if (split_env && c.S.ndef > 0 && c.l >= 8 * (int64_t)c.S.ndef && c.cg_blocks >= 64) {
        const int64_t target = c.cg_blocks / 4;
        while (dlpr < c.S.def_lpr && ((int64_t)c.S.ndef * (2 * dlpr) + VEC_THREADS - 1) / VEC_THREADS <= target) dlpr *= 2;
        ndb = (int)(((int64_t)c.S.ndef * dlpr + VEC_THREADS - 1) / VEC_THREADS);
        if (ndb < 1 || ndb > c.cg_blocks / 2) ndb = 0;
    }
#define FOS_UPD(DEF, FOLD, XUPD)                                                                                             \
    hipLaunchKernelGGL((cg_update_kernel<DEF, FOLD, XUPD>), grid, block, 0, c.stream, c.l, x, r, (const d2*)it.p_cur, Ap, c.st, c.partials, \
                       c.S.nwg, c.reduced, it.fold ? 0 : kkt_from_reduced, it.j, rr_out, c.S, c.cb, (int)c.n, c.def_mask, pb, it.seq_base, (int)c.count_repl, \
                       pre, (uint32_t)(it.seq_base + 2u * (uint32_t)it.j + 1u), ndb, dlpr)
#define FOS_UPD2(DEF, FOLD) do { if (it.fuse_p) FOS_UPD(DEF, FOLD, true); else FOS_UPD(DEF, FOLD, false); } while (0)
    if (c.S.ndef > 0) { if (it.fold) FOS_UPD2(true, true); else FOS_UPD2(true, false); }
    else { if (it.fold) FOS_UPD2(false, true); else FOS_UPD2(false, false); }
#undef FOS_UPD2
#undef FOS_UPD
}

// Third launch of a CG iteration when the p update is NOT fused into the next sweep: closes iteration j (every workgroup,
// same order: cg_close_iteration) and forms p_{j+1} = beta p_j + r.                   conjugategradients.jl:42-51
// It also carries x += alpha p_j (conjugategradients.jl:40) of the iteration it closes -- p_j is read here anyway -- and does so
// whether or not CG stops at this iteration (alpha = DevState.alpha, stored by the update kernel before this launch).
__global__ __launch_bounds__(VEC_THREADS) void cg_pupdate_kernel(int64_t l, d2* __restrict__ pnext, const d2* __restrict__ pcur,
                                                                 d2* __restrict__ x, const d2* __restrict__ r, DevState* st,
                                                                 const double* __restrict__ partials, int count,
                                                                 const double* __restrict__ reduced, int from_reduced, int j,
                                                                 PeerBox pb, uint32_t seq_base, int32_t batch_mark) {
    const int64_t stride = (int64_t)gridDim.x * VEC_THREADS;
    const int64_t i0 = blockIdx.x * (int64_t)VEC_THREADS + threadIdx.x;
    d2 p0 = make_double2(0.0, 0.0), r0 = p0, x0 = p0;
    if (i0 < l) { p0 = pcur[i0]; r0 = r[i0]; x0 = x[i0]; }     // requested before the scalar prologue (latency)
    // ... and with them everything the prologue reads (stored by earlier launches, none of it written by this one): the gate below
    // then costs no round trip of its own
    const double alpha = st->alpha;
    const int xfail = pb.nranks > 0 ? st->xchg_failed : 0;
    const CgCloseIn cin = cg_close_request(st, partials, count, reduced, from_reduced, r, l, j);
    unsigned long long w = __hip_atomic_load(reinterpret_cast<const unsigned long long*>(&st->iter), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (st->dbg_delay > 0 && blockIdx.x != 0) {                // test hook: let workgroup 0 finish first (tests/test_gpu_parity.py)
        const long long t0 = wall_clock64();
        while (wall_clock64() - t0 < st->dbg_delay) __builtin_amdgcn_s_sleep(8);
        w = __hip_atomic_load(reinterpret_cast<const unsigned long long*>(&st->iter), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    // The entry gate must not be the live `done` alone: workgroup 0 of THIS launch sets it (cg_close_iteration) when CG stops at
    // iteration j, and a workgroup that starts after that store would skip the x update of iteration j.  A stop recorded by an
    // earlier launch has iter < j; one recorded by this launch has iter == j (stored before `done`, one 8-byte word with it, so
    // a single load sees a consistent pair) -- then this workgroup goes on, takes the same stop decision and applies x += alpha p.
    if ((uint32_t)(w >> 32) != 0u && (int32_t)(uint32_t)w != j) return;
    if (xfail) return;
    const CgClose cl = cg_close_finish(st, cin, j, pb, seq_base);
    if (!cl.ok) return;
    const double beta = cl.beta;
    const bool go_on = !cl.stop;
    if (batch_mark != 0 && go_on && blockIdx.x == 0 && threadIdx.x == 0 && st->hostmark)      // the host's batch is used up, CG is not done
        __hip_atomic_store(&reinterpret_cast<HostMark*>(st->hostmark)->batch, batch_mark, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    if (i0 < l) {
        x[i0] = make_double2(x0.x + alpha * p0.x, x0.y + alpha * p0.y);
        if (go_on) pnext[i0] = make_double2(p0.x * beta + r0.x, p0.y * beta + r0.y);
    }
    for (int64_t i = i0 + stride; i < l; i += stride) {
        d2 pi = pcur[i];
        d2 xi = x[i];
        xi.x += alpha * pi.x; xi.y += alpha * pi.y;
        x[i] = xi;
        if (go_on) {
            const d2 ri = r[i];
            pi.x = pi.x * beta + ri.x;
            pi.y = pi.y * beta + ri.y;
            pnext[i] = pi;
        }
    }
}
void launch_cg_pupdate(const LaunchCtx& c, const CgIter& it, double2* x, double2* p_next) {
    hipLaunchKernelGGL(cg_pupdate_kernel, dim3(c.cg_blocks), dim3(VEC_THREADS), 0, c.stream, c.l, p_next, (const d2*)it.p_cur, x, it.r, c.st,
                       c.partials + 3 * (size_t)PART_CAP, c.cg_blocks, c.reduced, it.rr_from_reduced, it.j,
                       it.fold ? *it.fold : PeerBox{}, it.seq_base, it.batch_mark);
}

// ------------------------------------------------------------------------------------------------ merged-reduction CG
// The update launch of iteration j (i = j-1) of the merged-reduction recurrence (fos_internal.hpp, CgmIter).  EVERY workgroup
// forms the scalars itself from the same records in the same order:
//   the sweep's three sums (w.r without the tau row, [c;b].r1, [c;b].r2)  -> the tau row of w = M r_i and d_i = w.r ;
//   g_i = r_i.r_i : DevState.rn2[i & 1] (stored by the launch that closed iteration i), or -- CLOSE: this kernel closes
//   iteration i itself -- the r.r records of the previous update (+ the stop test of conjugategradients.jl:42: all return);
//   beta_i = g_i / g_{i-1},  alpha_i = g_i / (d_i - beta_i g_i / alpha_{i-1})   (i = 0: beta = 0, alpha = g_0 / d_0);
// then its slice:  p = p beta + r ; s = s beta + w ; x += alpha p ; r -= alpha s ; partial r.r  (= conjugategradients.jl:39-41,49-50
// with Ap replaced by the recurrence s = M p).  Sharded with peer mailboxes (FOLD): the four local sums cross the ranks in ONE
// exchange here.  DEF: rows of w the sweep left spread over dual-tile slots are finished here, as in cg_update_kernel.
// close_only: a one-workgroup launch at the end of a batch of enqueued iterations -- the scalar part alone.
struct CgmArgs {
    int64_t l;
    d2 *x, *r, *p, *s;
    const d2* w;
    DevState* st;
    const double* kkt_partials; int nkkt;
    const double* rr_in; int nrr;          // the r.r records of the update of iteration i
    double* rr_out;                        // ... of this one
    const double* reduced; int from_reduced;
    int j;
    int close_here, close_only;
    const double* cb; int n;
    const uint32_t* def_mask;
    PeerBox pb; uint32_t seq_base; int count_repl;
    double* pre; uint32_t pre_seq;
    int32_t batch_mark;
};
// (three wavefronts per SIMD: at two -- above 168 VGPRs -- the 512 workgroups of one launch fill the device, and two ranks that
// share ONE GPU, as the tests' ranks do, can then wait for each other's mailbox words for ever: the peer's kernel finds no slot)
template <bool DEF, bool FOLD>
__global__ __launch_bounds__(VEC_THREADS, 3) void cgm_update_kernel(CgmArgs a, DevBlkCsr S) {
    const int64_t l = a.l;
    const int64_t stride = (int64_t)gridDim.x * VEC_THREADS;
    const int64_t i0 = blockIdx.x * (int64_t)VEC_THREADS + threadIdx.x;
    const int i = a.j - 1;
    const bool first = i == 0;
    DevState* st = a.st;
    bool have0 = i0 < l && !a.close_only;
    uint32_t mw0 = 0u;                                   // (mask word and elements requested together: cg_update_kernel)
    if constexpr (DEF) { if (have0) mw0 = a.def_mask[i0 >> 5]; }
    // ---- everything this launch needs from memory is requested in STAGES, each stage's loads issued together, before the first
    // use: the kernel is a chain of memory latencies on small operators (a shard of a multi-GPU run steps through 17 x 2 of them)
    // stage 1: the thread's first element; the scalars (stored by EARLIER launches: nothing below races them) and the r.r records;
    //          the sweep's records; the slot-spread row this lane works on in the first pass (its list bounds and row number)
    d2 w0 = make_double2(0.0, 0.0), r0 = w0, x0 = w0, p0 = w0, s0 = w0;
    if (have0) { w0 = a.w[i0]; r0 = a.r[i0]; x0 = a.x[i0]; if (!first) { p0 = a.p[i0]; s0 = a.s[i0]; } }
    const bool closing = a.close_here != 0;      // (i = 0: g_0 from the start kernel's records, no stop test -- at least one iteration runs)
    const int done = st->done, xfail = st->xchg_failed, maxit = st->maxit;
    const double vtx = st->vtau[0], vty = st->vtau[1];           // tau element of r_i (stashed by the sweep that applied M to it)
    const double g_cur = st->rn2[i & 1], g_prev = st->rn2[(i + 1) & 1], a_prev = st->alpha2[(i + 1) & 1], tol = st->tol;
    const bool plain_sums = !a.from_reduced && a.pre == nullptr;
    PartialRegs<3, 4> kreg;                        // (4 x 256 records per round trip; more would cost spills at three wavefronts per SIMD)
    if (plain_sums) kreg.load(a.kkt_partials, a.nkkt);
    const int lpr = DEF ? S.def_lpr : 1, sh = 31 - __clz(lpr);
    const int rows_per_pass = (gridDim.x * VEC_THREADS) >> sh;
    const int lig = threadIdx.x & (lpr - 1);
    const int q0 = (blockIdx.x * VEC_THREADS + threadIdx.x) >> sh;
    const bool ok0 = DEF && !a.close_only && q0 < S.ndef;
    DefRow dr0{};
    if (ok0) dr0 = ld_defrow(S.def_rec + q0);
    const int drow0 = dr0.row, dn0 = dr0.count + (dr0.own >= 0 ? 1 : 0);
    // (only the first wavefront's sum is used -- thread 0 hands it on below: the others leave the 4 KB of records alone)
    const double rs = (!closing || threadIdx.x >= 64) ? 0.0 : (a.from_reduced ? a.reduced[3] : wave_sum_records(a.rr_in, a.nrr));
    // stage 2: the first NDV slot indices of that row's list; the row's vector elements
    constexpr int NDV = 5;                               // (C4: 33 slots per row on 8 lanes -- five for the first lane)
    int did[NDV];
#pragma unroll
    for (int q = 0; q < NDV; ++q) did[q] = -1;
    d2 dri = make_double2(0.0, 0.0), dpi = dri, dsi = dri, dxi = dri;
    double dc = 0.0;
    if constexpr (DEF) {
        if (ok0) {
#pragma unroll
            for (int q = 0; q < NDV; ++q) did[q] = (lig + q * lpr < dn0) ? defrow_slot(dr0, S.def_idx, lig + q * lpr) : -1;
            if (lig == 0) { dri = a.r[drow0]; dxi = a.x[drow0]; dc = a.cb[drow0]; if (!first) { dpi = a.p[drow0]; dsi = a.s[drow0]; } }
        }
    }
    if (done) return;
    if (FOLD && xfail) return;
    if ((mw0 >> (i0 & 31)) & 1u) have0 = false;         // a slot-spread row: finished below, its elements were read for nothing
    // stage 3: those slots are REQUESTED here and added behind the scalar part (the sums' reduction and the mailbox hop run while
    // they are on their way; the rest of a longer list follows there by the ordinary chunks)
    d2 dv[NDV];
    if constexpr (DEF) {
        const d2* __restrict__ slots = reinterpret_cast<const d2*>(S.slots_rd);
#pragma unroll
        for (int q = 0; q < NDV; ++q) dv[q] = did[q] >= 0 ? slots[did[q]] : make_double2(0.0, 0.0);
    }
    __shared__ double sums[4];
    if (plain_sums) {
        double acc3[3];
        kreg.sum(a.kkt_partials, a.nkkt, acc3);
        partials_combine<3>(acc3, sums);
    } else {
        sweep_sums3(sums, a.kkt_partials, a.nkkt, a.reduced, a.from_reduced, a.pre, a.pre_seq, st);
    }
    if (closing || FOLD) {
        if (threadIdx.x == 0) sums[3] = rs;
        __syncthreads();
    }
    if constexpr (FOLD) {
        if (!(S.dbg_flags & 128))                                                                        // (timing experiment, one rank: no mailbox hop)
        if (!peer_fold_sum<4>(a.pb, a.seq_base + (uint32_t)a.j, sums, st)) return;
    }
    const double S1 = sums[0], T1 = sums[1], T2 = sums[2];
    const double gam = closing ? sums[3] + (vtx * vtx + vty * vty) : g_cur;
    const bool w0blk = blockIdx.x == 0 && threadIdx.x == 0;
    if (closing && !first) {
        if (sqrt(gam) <= tol || i >= maxit) {                      // conjugategradients.jl:42 for iteration i
            if (w0blk) { st->rr = gam; cg_signal_stop(st, i, maxit, gam, a.seq_base >> 11); }
            return;
        }
    }
    const double wt1 = vtx + T2;             // r1_tau - (Q r2)_tau ,  (Q v)_tau = -[c;b].v        HSDEAffine.jl:57
    const double wt2 = -T1 - vty;            // (Q r1)_tau - r2_tau
    const double delta = S1 + (wt1 * vtx + wt2 * vty);
    double beta = 0.0, alpha;
    if (first) alpha = gam / delta;
    else {
        beta = gam / g_prev;
        alpha = gam / (delta - beta * gam / a_prev);
    }
    if (w0blk) {
        if (closing) st->rn2[i & 1] = gam;
        st->alpha2[i & 1] = alpha;
        st->alpha = alpha; st->beta = beta; st->pAp = gam / alpha; st->rr = gam;
        st->iter = a.j;
        if (a.close_only && a.batch_mark != 0 && st->hostmark)      // the host's batch is used up, CG is not done
            __hip_atomic_store(&reinterpret_cast<HostMark*>(st->hostmark)->batch, a.batch_mark, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    if (a.close_only) return;
    if ((S.dbg_flags & 64) && blockIdx.x != 0) return;                                                    // (timing experiment: scalar prologue only)
    double du1 = 0.0, du2 = 0.0;
    if constexpr (DEF) {
#pragma unroll
        for (int q = 0; q < NDV; ++q) if (did[q] >= 0) { du1 += dv[q].x; du2 += dv[q].y; }
        if (ok0) slot_list_sum(reinterpret_cast<const d2*>(S.slots_rd), S.def_idx, dr0, lig + NDV * lpr, lpr, du1, du2);
    }
    double acc[1] = {0.0};
    // one element: (w, r, p, s, x) -> (p, s, x, r); returns r.r of the element
    auto upd = [&](int64_t idx, d2 wi, d2 ri, d2 pi, d2 si, d2 xi) -> double {
        if (first) { pi = ri; si = wi; }
        else {
            pi.x = pi.x * beta + ri.x; pi.y = pi.y * beta + ri.y;          // p .*= beta ; p .+= r      :49-50
            si.x = si.x * beta + wi.x; si.y = si.y * beta + wi.y;          // s = M p by the same recurrence
        }
        xi.x += alpha * pi.x; xi.y += alpha * pi.y;                        // :40
        ri.x -= alpha * si.x; ri.y -= alpha * si.y;                        // :41
        a.p[idx] = pi; a.s[idx] = si; a.x[idx] = xi; a.r[idx] = ri;
        return ri.x * ri.x + ri.y * ri.y;
    };
    if constexpr (DEF) {
        const d2* __restrict__ slots = reinterpret_cast<const d2*>(S.slots_rd);
        const int npass = (S.dbg_flags & 4) ? 0 : (S.ndef + rows_per_pass - 1) / rows_per_pass;       // uniform trip count: the DPP sums need full waves
        int q = q0;
        for (int pass = 0; pass < npass; ++pass, q += rows_per_pass) {
            const bool ok = q < S.ndef;
            int row = drow0;
            d2 ri = dri, pi = dpi, si = dsi, xi = dxi;
            double c = dc, u1 = du1, u2 = du2;
            if (pass > 0) {                                      // (the first pass was requested in the prologue's stages)
                DefRow dr{};
                if (ok) dr = ld_defrow(S.def_rec + q);
                row = dr.row;
                ri = make_double2(0.0, 0.0); pi = ri; si = ri; xi = ri; c = 0.0; u1 = 0.0; u2 = 0.0;
                if (ok && lig == 0) { ri = a.r[row]; xi = a.x[row]; c = a.cb[row]; if (!first) { pi = a.p[row]; si = a.s[row]; } }
                if (ok) slot_list_sum(slots, S.def_idx, dr, lig, lpr, u1, u2);
            }
            const bool own = ok && lig == 0;
            u1 = group_sum(u1, lpr);
            u2 = group_sum(u2, lpr);
            if (own) {
                double q1, q2;                                  // EpiKkt::row (kernels.hip) on the applied vector r_i
                if (row < a.n) { q1 = u1 + vtx * c; q2 = u2 + vty * c; }
                else { q1 = -(u1 - vtx * c); q2 = -(u2 - vty * c); }
                const d2 wi = make_double2(ri.x - q2, q1 - ri.y);
                const double rr = upd(row, wi, ri, pi, si, xi);
                if (a.count_repl || row >= a.n) acc[0] += rr;    // (row-sharded: the replicated rows -- those of A' -- are counted by one rank)
            }
        }
    }
    if (have0) {
        if (i0 == l - 1) w0 = make_double2(wt1, wt2);
        const double rr = upd(i0, w0, r0, p0, s0, x0);
        if (i0 != l - 1) acc[0] += rr;
    }
    for (int64_t k = i0 + stride; k < l; k += stride) {
        uint32_t mw = 0u;
        if constexpr (DEF) mw = a.def_mask[k >> 5];
        d2 wi = a.w[k];
        const d2 ri = a.r[k], xi = a.x[k];
        d2 pi = make_double2(0.0, 0.0), si = pi;
        if (!first) { pi = a.p[k]; si = a.s[k]; }
        if ((mw >> (k & 31)) & 1u) continue;
        if (k == l - 1) wi = make_double2(wt1, wt2);
        const double rr = upd(k, wi, ri, pi, si, xi);
        if (k != l - 1) acc[0] += rr;
    }
    block_reduce_store<1>(acc, a.rr_out + blockIdx.x);
}
// Start of a merged-reduction solve (conjugategradients.jl:32-36 in one launch behind the sweep w = M v): finishes w -- the tau
// row from the sweep's sums, the slot-spread rows from their slot lists -- and forms r = rhs - w with the r.r records of
// "iteration 0" (added by whoever closes it: the next sweep, or the first update when sharded).  Its first workgroup opens the
// solve in DevState.  Not gated: `done` still holds the previous solve's 1.
struct CgmStartArgs {
    int64_t l;
    const d2 *rhs, *v, *w;
    d2* r;
    d2* p;                                 // non-null: also p_1 = r_0 (the reference recurrence's first direction, conjugategradients.jl:34)
    DevState* st;
    const double* kkt_partials; int nkkt;
    const double* reduced; int from_reduced;
    double* rr_out;
    const double* cb; int n;
    const uint32_t* def_mask;
    PeerBox pb; uint32_t seq_base; int count_repl;
    double tol; int maxit;
};
template <bool DEF, bool FOLD>
__global__ __launch_bounds__(VEC_THREADS) void cgm_start_kernel(CgmStartArgs a, DevBlkCsr S) {
    const int64_t l = a.l;
    const int64_t stride = (int64_t)gridDim.x * VEC_THREADS;
    const int64_t i0 = blockIdx.x * (int64_t)VEC_THREADS + threadIdx.x;
    DevState* st = a.st;
    bool have0 = i0 < l;
    uint32_t mw0 = 0u;
    if constexpr (DEF) { if (have0) mw0 = a.def_mask[i0 >> 5]; }
    d2 w0 = make_double2(0.0, 0.0), b0 = w0;
    if (have0) { w0 = a.w[i0]; b0 = a.rhs[i0]; }
    if ((mw0 >> (i0 & 31)) & 1u) have0 = false;
    const double vtx = st->vtau[0], vty = st->vtau[1];           // tau element of v (stashed by the sweep)
    if (FOLD && st->xchg_failed) return;
    __shared__ double sums[3];
    sweep_sums3(sums, a.kkt_partials, a.nkkt, a.reduced, a.from_reduced, nullptr, 0u, st);
    if constexpr (FOLD) {
        if (!peer_fold_sum<3>(a.pb, a.seq_base, sums, st)) return;
    }
    const double T1 = sums[1], T2 = sums[2];
    const double wt1 = vtx + T2, wt2 = -T1 - vty;                // HSDEAffine.jl:57
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        st->iter = 1; st->hit_max = 0; st->tol = a.tol; st->maxit = a.maxit; st->rn_old = 0.0;       // conjugategradients.jl:36
        st->done = 0;
    }
    double acc[1] = {0.0};
    if constexpr (DEF) {
        const d2* __restrict__ slots = reinterpret_cast<const d2*>(S.slots_rd);
        const int lpr = S.def_lpr, sh = 31 - __clz(lpr);
        const int rows_per_pass = (gridDim.x * VEC_THREADS) >> sh;
        const int lig = threadIdx.x & (lpr - 1);
        const int npass = (S.ndef + rows_per_pass - 1) / rows_per_pass;
        int q = (blockIdx.x * VEC_THREADS + threadIdx.x) >> sh;
        for (int pass = 0; pass < npass; ++pass, q += rows_per_pass) {
            const bool ok = q < S.ndef;
            DefRow dr{};
            if (ok) dr = ld_defrow(S.def_rec + q);
            const int row = dr.row;
            const bool own = ok && lig == 0;
            d2 vi = make_double2(0.0, 0.0), bi = vi;
            double c = 0.0;
            if (own) { vi = a.v[row]; bi = a.rhs[row]; c = a.cb[row]; }
            double u1 = 0.0, u2 = 0.0;
            if (ok) slot_list_sum(slots, S.def_idx, dr, lig, lpr, u1, u2);
            u1 = group_sum(u1, lpr);
            u2 = group_sum(u2, lpr);
            if (own) {
                double q1, q2;                                  // EpiKkt::row (kernels.hip)
                if (row < a.n) { q1 = u1 + vtx * c; q2 = u2 + vty * c; }
                else { q1 = -(u1 - vtx * c); q2 = -(u2 - vty * c); }
                const d2 ri = make_double2(bi.x - (vi.x - q2), bi.y - (q1 - vi.y));       // r = b - Ap      :33
                a.r[row] = ri;
                if (a.p) a.p[row] = ri;
                if (a.count_repl || row >= a.n) acc[0] += ri.x * ri.x + ri.y * ri.y;
            }
        }
    }
    if (have0) {
        if (i0 == l - 1) w0 = make_double2(wt1, wt2);
        const d2 ri = make_double2(b0.x - w0.x, b0.y - w0.y);
        a.r[i0] = ri;
        if (a.p) a.p[i0] = ri;
        if (i0 != l - 1) acc[0] += ri.x * ri.x + ri.y * ri.y;
    }
    for (int64_t k = i0 + stride; k < l; k += stride) {
        uint32_t mw = 0u;
        if constexpr (DEF) mw = a.def_mask[k >> 5];
        d2 wi = a.w[k];
        const d2 bi = a.rhs[k];
        if ((mw >> (k & 31)) & 1u) continue;
        if (k == l - 1) wi = make_double2(wt1, wt2);
        const d2 ri = make_double2(bi.x - wi.x, bi.y - wi.y);
        a.r[k] = ri;
        if (a.p) a.p[k] = ri;
        if (k != l - 1) acc[0] += ri.x * ri.x + ri.y * ri.y;
    }
    block_reduce_store<1>(acc, a.rr_out + blockIdx.x);
}
void launch_cgm_start(const LaunchCtx& c, const CgmIter& it, const double2* rhs, const double2* v, double tol, int maxit, double2* p_out) {
    CgmStartArgs a{};
    a.l = c.l; a.rhs = rhs; a.v = v; a.w = it.w; a.r = it.r; a.p = p_out; a.st = c.st;
    a.kkt_partials = c.partials; a.nkkt = c.S.nwg;
    a.reduced = c.reduced; a.from_reduced = it.fold ? 0 : it.from_reduced;
    a.rr_out = c.partials + 3 * (size_t)PART_CAP;                 // records of "iteration 0"
    a.cb = c.cb; a.n = (int)c.n; a.def_mask = c.def_mask;
    a.pb = it.fold ? *it.fold : PeerBox{}; a.seq_base = it.seq_base; a.count_repl = (int)c.count_repl;
    a.tol = tol; a.maxit = maxit;
    dim3 grid(c.cg_blocks), block(VEC_THREADS);
    if (c.S.ndef > 0) {
        if (it.fold) hipLaunchKernelGGL((cgm_start_kernel<true, true>), grid, block, 0, c.stream, a, c.S);
        else hipLaunchKernelGGL((cgm_start_kernel<true, false>), grid, block, 0, c.stream, a, c.S);
    } else {
        if (it.fold) hipLaunchKernelGGL((cgm_start_kernel<false, true>), grid, block, 0, c.stream, a, c.S);
        else hipLaunchKernelGGL((cgm_start_kernel<false, false>), grid, block, 0, c.stream, a, c.S);
    }
}
void launch_cgm_update(const LaunchCtx& c, const CgmIter& it, bool close_only) {
    CgmArgs a{};
    a.l = c.l; a.x = it.x; a.r = it.r; a.p = it.p; a.s = it.s; a.w = it.w; a.st = c.st;
    a.kkt_partials = c.partials; a.nkkt = c.S.nwg;
    double* rr = c.partials + 3 * (size_t)PART_CAP;
    a.rr_in = rr + (size_t)((it.j - 1) & 1) * CGM_RR_STRIDE; a.nrr = c.cg_blocks;
    a.rr_out = rr + (size_t)(it.j & 1) * CGM_RR_STRIDE;
    a.reduced = c.reduced; a.from_reduced = it.fold ? 0 : it.from_reduced;
    a.j = it.j; a.close_here = it.close_in_update ? 1 : 0; a.close_only = close_only ? 1 : 0;
    a.cb = c.cb; a.n = (int)c.n; a.def_mask = c.def_mask;
    a.pb = it.fold ? *it.fold : PeerBox{}; a.seq_base = it.seq_base; a.count_repl = (int)c.count_repl;
    a.pre = (c.pre && !it.fold && !it.from_reduced && !close_only && c.S.nwg >= pre_min_records() && c.cg_blocks >= 4 * PRE_NPROD) ? c.pre : nullptr;
    a.pre_seq = (uint32_t)(it.seq_base + 2u * (uint32_t)it.j + 1u);
    a.batch_mark = it.batch_mark;
    dim3 grid(close_only ? 1 : c.cg_blocks), block(VEC_THREADS);
    if (c.S.ndef > 0) {
        if (it.fold) hipLaunchKernelGGL((cgm_update_kernel<true, true>), grid, block, 0, c.stream, a, c.S);
        else hipLaunchKernelGGL((cgm_update_kernel<true, false>), grid, block, 0, c.stream, a, c.S);
    } else {
        if (it.fold) hipLaunchKernelGGL((cgm_update_kernel<false, true>), grid, block, 0, c.stream, a, c.S);
        else hipLaunchKernelGGL((cgm_update_kernel<false, false>), grid, block, 0, c.stream, a, c.S);
    }
}

void launch_cg_init(const LaunchCtx& c, const double2* rhs, const double2* Ap, double2* r, double2* p) {
    hipLaunchKernelGGL(cg_init_kernel, dim3(c.vec_blocks), dim3(VEC_THREADS), 0, c.stream, c.l, rhs, Ap, r, p, c.partials,
                       c.count_repl ? (int64_t)0 : c.n_repl);
}
void launch_cg_init_finalize(const LaunchCtx& c, const double2* r, double tol, int maxit, int from_reduced) {
    hipLaunchKernelGGL(cg_init_finalize_kernel, dim3(1), dim3(FIN_THREADS), 0, c.stream, c.partials, c.vec_blocks, c.reduced,
                       from_reduced, r, c.l, c.st, tol, maxit);
}

// ------------------------------------------------------------------------------------------------ outer-loop passes

#define GRID_STRIDE(i, l) for (int64_t i = blockIdx.x * (int64_t)VEC_THREADS + threadIdx.x; i < (l); i += (int64_t)gridDim.x * VEC_THREADS)

// out = a x + b y          (gap.jl:48  y .= a1.*y .+ (1-a1).*x ; fista.jl:37)
__global__ __launch_bounds__(VEC_THREADS) void axpby_kernel(int64_t l, d2* __restrict__ out, double a, const d2* __restrict__ x,
                                                            double b, const d2* __restrict__ y, const int32_t* __restrict__ gate) {
    if (gate && !*gate) return;                  // speculatively enqueued behind a CG batch that did not converge: no-op
    GRID_STRIDE(i, l) {
        const d2 xi = x[i], yi = y[i];
        out[i] = make_double2(a * xi.x + b * yi.x, a * xi.y + b * yi.y);
    }
}
// out = a12 y + (1-a12) x with a12 from the device state          gapa.jl:67
__global__ __launch_bounds__(VEC_THREADS) void relax_a12_kernel(int64_t l, d2* __restrict__ out, const d2* __restrict__ y,
                                                                const d2* __restrict__ x, const DevState* st, const int32_t* __restrict__ gate) {
    if (gate && !*gate) return;                  // speculatively enqueued behind a CG batch that did not converge: no-op
    const double a = st->alpha12, b = 1 - a;
    GRID_STRIDE(i, l) {
        const d2 xi = x[i], yi = y[i];
        out[i] = make_double2(a * yi.x + b * xi.x, a * yi.y + b * xi.y);
    }
}
// elementwise cone operations (cones.jl:97-102; used by relax_ew_kernel here and by cones_elementwise_kernel below)
__device__ __forceinline__ double ew_apply(int op, double v) {
    switch (op) {
        case EW_COPY: return v;               // IndFree / dual of IndZero           cones.jl:98
        case EW_ZERO: return 0.0;             // IndZero / dual of IndFree (IndPoint) cones.jl:100
        // comparisons, not fmax/fmin: a NaN must stay a NaN as it does in the reference (Julia's max(NaN,0) is NaN)
        case EW_MAX0: return v < 0.0 ? 0.0 : v;   // IndNonnegative (self dual)       cones.jl:101 ; tau, kappa :138,141
        default:      return v > 0.0 ? 0.0 : v;   // IndNonpositive                   cones.jl:102
    }
}
// The relaxation behind S1 and the elementwise share of the cone projection in ONE pass (one launch less per outer iteration):
//   t1 = a sol + (1 - a) x   (a: the argument, or alpha12 from the device state -- the arithmetic of axpby_kernel / relax_a12_kernel)
//   t2 = P(t1) on the indices of Free / Zero / NonNeg / NonPos cones and the (tau, kappa) element (cones_elementwise_kernel)
__global__ __launch_bounds__(VEC_THREADS) void relax_ew_kernel(int64_t l, d2* __restrict__ t1, d2* __restrict__ t2, const d2* __restrict__ sol,
                                                               const d2* __restrict__ x, double a_arg, int use_a12, const DevState* st,
                                                               const uint8_t* __restrict__ ew_op, const int32_t* __restrict__ gate) {
    if (gate && !*gate) return;                  // speculatively enqueued behind a CG batch that did not converge: no-op
    const double a = use_a12 ? st->alpha12 : a_arg, b = 1 - a;
    GRID_STRIDE(i, l) {
        const d2 si = sol[i], xi = x[i];
        const uint8_t op = ew_op[i];
        const d2 v = make_double2(a * si.x + b * xi.x, a * si.y + b * xi.y);
        t1[i] = v;
        if (op != EW_SKIP) t2[i] = make_double2(ew_apply(op & 3, v.x), ew_apply((op >> 2) & 3, v.y));
    }
}
// tmp2 = a2 tmp2 + (1-a2) tmp1 ; x = a tmp2 + (1-a) x           gap.jl:58,78
// shift_out (optional): also the vector the NEXT affine projection's CG start applies M to, sol - [0; x2_new] (shift_part2_kernel),
// so that projection needs no pass of its own for it
__global__ __launch_bounds__(VEC_THREADS) void gap_final_kernel(int64_t l, d2* __restrict__ x, const d2* __restrict__ t2,
                                                                const d2* __restrict__ t1, double alpha, double alpha2, const int32_t* __restrict__ gate,
                                                                d2* __restrict__ shift_out, const d2* __restrict__ sol) {
    if (gate && !*gate) return;                  // speculatively enqueued behind a CG batch that did not converge: no-op
    const double b2 = 1 - alpha2, b = 1 - alpha;
    GRID_STRIDE(i, l) {
        const d2 u = t2[i], v = t1[i];
        d2 xi = x[i];
        const double rx = alpha2 * u.x + b2 * v.x, ry = alpha2 * u.y + b2 * v.y;
        xi.x = alpha * rx + b * xi.x;
        xi.y = alpha * ry + b * xi.y;
        x[i] = xi;
        if (shift_out) { const d2 y = sol[i]; shift_out[i] = make_double2(y.x, y.y - xi.y); }
    }
}
// GAPA: same with a12 from the state, plus the three sums of normedScalar(tmp2,tmp1,tmp1,x)   gapa.jl:36-47,77,96,103
// tau-row contributions go to reduced[8..10] (replicated across shards, added after the all-reduce).
__global__ __launch_bounds__(VEC_THREADS) void gapa_final_kernel(int64_t l, d2* __restrict__ x, const d2* __restrict__ t2,
                                                                 const d2* __restrict__ t1, double alpha, const DevState* st,
                                                                 double* __restrict__ partials, double* __restrict__ reduced, int64_t acc_from, const int32_t* __restrict__ gate,
                                                                 d2* __restrict__ shift_out, const d2* __restrict__ sol) {
    if (gate && !*gate) return;                  // speculatively enqueued behind a CG batch that did not converge: no-op
    const double a12 = st->alpha12, b12 = 1 - a12, b = 1 - alpha;
    double acc[3] = {0.0, 0.0, 0.0};
    GRID_STRIDE(i, l) {
        const d2 u = t2[i], v = t1[i];
        d2 xi = x[i];
        const double rx = a12 * u.x + b12 * v.x, ry = a12 * u.y + b12 * v.y;
        const double d1x = rx - v.x, d1y = ry - v.y;      // tmp2 - tmp1
        const double d2x = v.x - xi.x, d2y = v.y - xi.y;  // tmp1 - x
        const double s = d1x * d2x + d1y * d2y, n1 = d1x * d1x + d1y * d1y, n2 = d2x * d2x + d2y * d2y;
        if (i == l - 1) { reduced[8] = s; reduced[9] = n1; reduced[10] = n2; }
        else if (i >= acc_from) { acc[0] += s; acc[1] += n1; acc[2] += n2; }
        xi.x = alpha * rx + b * xi.x;
        xi.y = alpha * ry + b * xi.y;
        x[i] = xi;
        if (shift_out) { const d2 y = sol[i]; shift_out[i] = make_double2(y.x, y.y - xi.y); }
    }
    block_reduce_store<3>(acc, partials + 3 * (int64_t)blockIdx.x);
}
// scl = clamp(|sum|/sqrt(n1 n2),0,1), NaN -> 0 ; s = sqrt(1-scl^2) ; a12 = (1-beta) 2/(1+s) + 2 beta     gapa.jl:96-101
__global__ __launch_bounds__(FIN_THREADS) void gapa_finalize_kernel(const double* __restrict__ partials, int count,
                                                                    double* __restrict__ reduced, int from_reduced, double beta, DevState* st, const int32_t* __restrict__ gate) {
    if (gate && !*gate) return;                  // speculatively enqueued behind a CG batch that did not converge: no-op
    __shared__ double sums[3];
    if (from_reduced) { if (threadIdx.x < 3) sums[threadIdx.x] = reduced[threadIdx.x]; __syncthreads(); }
    else reduce_partials<3>(partials, count, sums);
    if (threadIdx.x == 0) {
        const double sum = sums[0] + reduced[8], n1 = sums[1] + reduced[9], n2 = sums[2] + reduced[10];
        double scl = fabs(sum) / sqrt(n1 * n2);
        if (scl != scl) scl = 0.0;                // isnan -> 0 (clamp of NaN stays NaN in Julia, then :97)
        else scl = fmin(fmax(scl, 0.0), 1.0);
        const double s = sqrt(1 - scl * scl);
        const double aopt = 2 / (1 + s);
        st->gapa_scl = scl;
        st->alpha12 = (1 - beta) * aopt + beta * 2.0;
    }
}
// y = x + coef (x - xold)                                          fista.jl:46
__global__ __launch_bounds__(VEC_THREADS) void fista_extrap_kernel(int64_t l, d2* __restrict__ y, const d2* __restrict__ x,
                                                                   const d2* __restrict__ xold, double coef, const int32_t* __restrict__ gate) {
    if (gate && !*gate) return;
    GRID_STRIDE(i, l) {
        const d2 xi = x[i], xo = xold[i];
        y[i] = make_double2(xi.x + coef * (xi.x - xo.x), xi.y + coef * (xi.y - xo.y));
    }
}
__global__ __launch_bounds__(VEC_THREADS) void add_kernel(int64_t l, d2* __restrict__ out, const d2* __restrict__ a, const d2* __restrict__ b, const int32_t* __restrict__ gate) {
    if (gate && !*gate) return;
    GRID_STRIDE(i, l) { const d2 u = a[i], v = b[i]; out[i] = make_double2(u.x + v.x, u.y + v.y); }
}
// out = in (a copy that can be gated, unlike hipMemcpyAsync: fista.jl:39 xold .= x behind a speculatively enqueued CG batch)
__global__ __launch_bounds__(VEC_THREADS) void copy_kernel(int64_t l, d2* __restrict__ out, const d2* __restrict__ in, const int32_t* __restrict__ gate) {
    if (gate && !*gate) return;
    GRID_STRIDE(i, l) out[i] = in[i];
}
// p .= x .+ p .- y                                                 dykstra.jl:29,33
__global__ __launch_bounds__(VEC_THREADS) void dykstra_corr_kernel(int64_t l, d2* __restrict__ p, const d2* __restrict__ x, const d2* __restrict__ y, const int32_t* __restrict__ gate) {
    if (gate && !*gate) return;
    GRID_STRIDE(i, l) {
        const d2 xi = x[i], yi = y[i];
        d2 pi = p[i];
        pi.x = (xi.x + pi.x) - yi.x;
        pi.y = (xi.y + pi.y) - yi.y;
        p[i] = pi;
    }
}

// out = y - [0; x2] in the interleaved layout: (y.x, y.y - x.y)        (prox_affine: the vector the CG start residual applies M to)
__global__ __launch_bounds__(VEC_THREADS) void shift_part2_kernel(int64_t l, d2* __restrict__ out, const d2* __restrict__ y, const d2* __restrict__ x) {
    for (int64_t i = blockIdx.x * (int64_t)VEC_THREADS + threadIdx.x; i < l; i += (int64_t)gridDim.x * VEC_THREADS) {
        const d2 yi = y[i];
        out[i] = make_double2(yi.x, yi.y - x[i].y);
    }
}
void launch_shift_part2(const LaunchCtx& c, double2* out, const double2* y, const double2* x) {
    hipLaunchKernelGGL(shift_part2_kernel, dim3(c.vec_blocks), dim3(VEC_THREADS), 0, c.stream, c.l, out, y, x);
}
// partial sums of |x - y|^2 over both parts (normdiff of wrappers/linesearch.jl:77-85), one record per workgroup
__global__ __launch_bounds__(VEC_THREADS) void normdiff_kernel(int64_t l, const d2* __restrict__ x, const d2* __restrict__ y, double* __restrict__ partials) {
    double acc[1] = {0.0};
    for (int64_t i = blockIdx.x * (int64_t)VEC_THREADS + threadIdx.x; i < l; i += (int64_t)gridDim.x * VEC_THREADS) {
        const d2 a = x[i], b = y[i];
        const double dx = a.x - b.x, dy = a.y - b.y;
        acc[0] += dx * dx + dy * dy;
    }
    block_reduce_store<1>(acc, partials + blockIdx.x);
}
// ---- LongstepWrapper (wrappers/longstep.jl): the saved half-planes
// addprojeq / addprojineq (longstep.jl:65-101): row = x - y, its offset b = (x - y).y as partial sums per workgroup
__global__ __launch_bounds__(VEC_THREADS) void long_plane_kernel(int64_t l, d2* __restrict__ row, const d2* __restrict__ x, const d2* __restrict__ y,
                                                                 double* __restrict__ bpart) {
    double acc[1] = {0.0};
    for (int64_t i = blockIdx.x * (int64_t)VEC_THREADS + threadIdx.x; i < l; i += (int64_t)gridDim.x * VEC_THREADS) {
        const d2 a = x[i], b = y[i];
        const d2 d = make_double2(a.x - b.x, a.y - b.y);
        row[i] = d;
        acc[0] += d.x * b.x + d.y * b.y;
    }
    block_reduce_store<1>(acc, bpart + blockIdx.x);
}
// The Gram products of the saved planes in DOUBLE-DOUBLE.  The normals of successive iterations are nearly dependent (measured on the README
// NNLS with the default nsave = 10: cond(P) 1e8 .. 1e10), which is why the reference hands the projection to a BigFloat QP (saveplanes.jl:24);
// G = P P' formed in float64 loses cond(P)^2 eps = everything.  Products by FMA (exact), sums by TwoSum, across the workgroup as well; the host
// adds the workgroups' pairs and solves the small dual in 113-bit arithmetic (solver.cpp, long_project_planes).
struct dd { double hi, lo; };
// (plain operators under `fp contract(off)`: the error-free transformations below must not be contracted into FMAs -- the __dadd_rn / __dmul_rn
//  of this toolchain's headers are a plain + and * compiled with contraction ON, and were fused: checked in the ISA)
__device__ __forceinline__ dd dd_add(dd a, dd b) {
#pragma clang fp contract(off)
    const double s = a.hi + b.hi, z = s - a.hi;
    double e = (a.hi - (s - z)) + (b.hi - z);
    e = e + (a.lo + b.lo);
    dd r;
    r.hi = s + e;
    r.lo = e - (r.hi - s);
    return r;
}
__device__ __forceinline__ void dd_add_prod(dd& acc, double a, double b) {      // acc += a b
#pragma clang fp contract(off)
    dd p;
    p.hi = a * b;
    p.lo = __builtin_fma(a, b, -p.hi);
    acc = dd_add(acc, p);
}
template <int NV>
__device__ __forceinline__ void dd_block_reduce_store(dd (&acc)[NV], double* out) {      // out[2 v], out[2 v + 1] = (hi, lo) of the workgroup's sum of acc[v]
    __shared__ double sh_hi[VEC_THREADS], sh_lo[VEC_THREADS];
    const int tid = threadIdx.x;
#pragma unroll 1
    for (int v = 0; v < NV; ++v) {
        sh_hi[tid] = acc[v].hi; sh_lo[tid] = acc[v].lo;
        __syncthreads();
        for (int s = VEC_THREADS / 2; s > 0; s >>= 1) {
            if (tid < s) {
                const dd r = dd_add(dd{sh_hi[tid], sh_lo[tid]}, dd{sh_hi[tid + s], sh_lo[tid + s]});
                sh_hi[tid] = r.hi; sh_lo[tid] = r.lo;
            }
            __syncthreads();
        }
        if (tid == 0) { out[2 * v] = sh_hi[0]; out[2 * v + 1] = sh_lo[0]; }
        __syncthreads();
    }
}
// partial sums of row_a . row_b for b = a .. K-1 and of row_a . x: out[blockIdx][33][hi, lo] (what the small dual QP of the projection needs)
constexpr int LONG_KMAX = 32;
__global__ __launch_bounds__(VEC_THREADS) void long_dots_kernel(int64_t l, const d2* __restrict__ P, int K, int a, const d2* __restrict__ x,
                                                                double* __restrict__ out) {
    dd acc[LONG_KMAX + 1];
#pragma unroll
    for (int k = 0; k <= LONG_KMAX; ++k) acc[k] = dd{0.0, 0.0};
    const d2* __restrict__ pa = P + (int64_t)a * l;
    for (int64_t i = blockIdx.x * (int64_t)VEC_THREADS + threadIdx.x; i < l; i += (int64_t)gridDim.x * VEC_THREADS) {
        const d2 va = pa[i];
#pragma unroll
        for (int k = 0; k < LONG_KMAX; ++k) {
            if (a + k < K) { const d2 vb = P[(int64_t)(a + k) * l + i]; dd_add_prod(acc[k], va.x, vb.x); dd_add_prod(acc[k], va.y, vb.y); }
        }
        const d2 xi = x[i];
        dd_add_prod(acc[LONG_KMAX], va.x, xi.x);
        dd_add_prod(acc[LONG_KMAX], va.y, xi.y);
    }
    dd_block_reduce_store<LONG_KMAX + 1>(acc, out + (int64_t)blockIdx.x * 2 * (LONG_KMAX + 1));
}
// x += sum_k nu[k] row_k  (the projection onto the saved planes, written back: longstep.jl:57).  The multipliers of nearly dependent planes are
// large and their terms cancel: nu as (hi, lo) pairs (nu[2 k], nu[2 k + 1]), the sum in double-double, rounded once.
__global__ __launch_bounds__(VEC_THREADS) void long_apply_kernel(int64_t l, d2* __restrict__ x, const d2* __restrict__ P, int K, const double* __restrict__ nu) {
    GRID_STRIDE(i, l) {
        const d2 v = x[i];
        dd sx{v.x, 0.0}, sy{v.y, 0.0};
        for (int k = 0; k < K; ++k) {
            const double wh = nu[2 * k], wl = nu[2 * k + 1];
            const d2 r = P[(int64_t)k * l + i];
            dd_add_prod(sx, wh, r.x); dd_add_prod(sx, wl, r.x);
            dd_add_prod(sy, wh, r.y); dd_add_prod(sy, wl, r.y);
        }
        x[i] = make_double2(sx.hi + sx.lo, sy.hi + sy.lo);
    }
}
void launch_long_plane(const LaunchCtx& c, double2* row, const double2* x, const double2* y, double* bpart) {
    hipLaunchKernelGGL(long_plane_kernel, dim3(c.vec_blocks), dim3(VEC_THREADS), 0, c.stream, c.l, row, x, y, bpart);
}
void launch_long_dots(const LaunchCtx& c, const double2* P, int K, int a, const double2* x, double* out) {
    hipLaunchKernelGGL(long_dots_kernel, dim3(c.vec_blocks), dim3(VEC_THREADS), 0, c.stream, c.l, P, K, a, x, out);
}
void launch_long_apply(const LaunchCtx& c, double2* x, const double2* P, int K, const double* nu) {
    hipLaunchKernelGGL(long_apply_kernel, dim3(c.vec_blocks), dim3(VEC_THREADS), 0, c.stream, c.l, x, P, K, nu);
}
void launch_normdiff(const LaunchCtx& c, const double2* x, const double2* y) {
    hipLaunchKernelGGL(normdiff_kernel, dim3(c.vec_blocks), dim3(VEC_THREADS), 0, c.stream, c.l, x, y, c.partials);
}
void launch_axpby(const LaunchCtx& c, double2* out, double a, const double2* x, double b, const double2* y) {
    hipLaunchKernelGGL(axpby_kernel, dim3(c.vec_blocks), dim3(VEC_THREADS), 0, c.stream, c.l, out, a, x, b, y, c.gate);
}
void launch_relax_ew(const LaunchCtx& c, double2* t1, double2* t2, const double2* sol, const double2* x, double a, bool use_a12, const uint8_t* ew_op) {
    hipLaunchKernelGGL(relax_ew_kernel, dim3(c.vec_blocks), dim3(VEC_THREADS), 0, c.stream, c.l, t1, t2, sol, x, a, use_a12 ? 1 : 0, (const DevState*)c.st, ew_op, c.gate);
}
void launch_relax_a12(const LaunchCtx& c, double2* out, const double2* y, const double2* x) {
    hipLaunchKernelGGL(relax_a12_kernel, dim3(c.vec_blocks), dim3(VEC_THREADS), 0, c.stream, c.l, out, y, x, c.st, c.gate);
}
void launch_gap_final(const LaunchCtx& c, double2* x, const double2* t2, const double2* t1, double alpha, double alpha2, double2* shift_out, const double2* sol) {
    hipLaunchKernelGGL(gap_final_kernel, dim3(c.vec_blocks), dim3(VEC_THREADS), 0, c.stream, c.l, x, t2, t1, alpha, alpha2, c.gate, shift_out, sol);
}
void launch_gapa_final(const LaunchCtx& c, double2* x, const double2* t2, const double2* t1, double alpha, double2* shift_out, const double2* sol) {
    hipLaunchKernelGGL(gapa_final_kernel, dim3(c.vec_blocks), dim3(VEC_THREADS), 0, c.stream, c.l, x, t2, t1, alpha, c.st, c.partials, c.reduced,
                       c.count_repl ? (int64_t)0 : c.n_repl, c.gate, shift_out, sol);
}
void launch_gapa_finalize(const LaunchCtx& c, double beta, int from_reduced) {
    hipLaunchKernelGGL(gapa_finalize_kernel, dim3(1), dim3(FIN_THREADS), 0, c.stream, c.partials, c.vec_blocks, c.reduced, from_reduced, beta, c.st, c.gate);
}
void launch_fista_extrap(const LaunchCtx& c, double2* y, const double2* x, const double2* xold, double coef) {
    hipLaunchKernelGGL(fista_extrap_kernel, dim3(c.vec_blocks), dim3(VEC_THREADS), 0, c.stream, c.l, y, x, xold, coef, c.gate);
}
void launch_add(const LaunchCtx& c, double2* out, const double2* a, const double2* b) {
    hipLaunchKernelGGL(add_kernel, dim3(c.vec_blocks), dim3(VEC_THREADS), 0, c.stream, c.l, out, a, b, c.gate);
}
void launch_copy(const LaunchCtx& c, double2* out, const double2* in) {
    hipLaunchKernelGGL(copy_kernel, dim3(c.vec_blocks), dim3(VEC_THREADS), 0, c.stream, c.l, out, in, c.gate);
}
void launch_dykstra_corr(const LaunchCtx& c, double2* p, const double2* x, const double2* y) {
    hipLaunchKernelGGL(dykstra_corr_kernel, dim3(c.vec_blocks), dim3(VEC_THREADS), 0, c.stream, c.l, p, x, y, c.gate);
}

// ------------------------------------------------------------------------------------------------ direct = true (HSDE.jl:12-15)
// S1 = IndAffine([Q -I], 0): the exact projection of [u; v] onto {Q u = v},
//     w = (I + Q Q')^-1 (Q u - v),   u+ = u - Q'w = u + Q w,   v+ = v + w        (Q' = -Q, HSDEAffine.jl:61-65)
// with G^-1 = (I + Q Q')^-1 = (I - Q Q)^-1 formed ONCE as a dense matrix by the Newton-Schulz iteration X <- 2X - X (G X) on a
// hand-written fp64 MFMA GEMM (quadratically convergent for the symmetric positive definite G, lambda_min(G) >= 1; needs only
// matrix products -- the vendor's dense solver library alone takes minutes to load on a fresh box); per projection two Q sweeps,
// one dense symmetric matrix-vector product (HBM bound: 8 l^2 bytes) and two elementwise passes.

// dense Q (column-major, l x l) from the CSC of A: column j < n holds -A(:,j) in rows n.., -c_j in the last row; column n+i holds
// A(i,:)' in rows 0..n-1 and -b_i in the last row; the last column is [c; b; 0]
__global__ __launch_bounds__(VEC_THREADS) void dense_q_fill_kernel(int64_t m, int64_t n, const int64_t* __restrict__ colptr,
                                                                   const int64_t* __restrict__ rowval, const double* __restrict__ nzval,
                                                                   const double* __restrict__ cb, double* __restrict__ Q, int64_t ld) {
    const int64_t l = ld, lm1 = n + m;              // leading dimension (l padded to a multiple of 64); index of the tau row
    for (int64_t j = blockIdx.x; j < n; j += gridDim.x) {
        for (int64_t k = colptr[j] - 1 + threadIdx.x; k < colptr[j + 1] - 1; k += VEC_THREADS) {
            const int64_t i = rowval[k] - 1;
            const double a = nzval[k];
            Q[(n + i) + j * l] = -a;                 // -A
            Q[j + (n + i) * l] = a;                  //  A'
        }
    }
    for (int64_t q = blockIdx.x * (int64_t)VEC_THREADS + threadIdx.x; q < n + m; q += (int64_t)gridDim.x * VEC_THREADS) {
        Q[q + lm1 * l] = cb[q];                      // last column: [c; b]
        Q[lm1 + q * l] = -cb[q];                     // last row: [-c', -b']
    }
}
// w = G t for a symmetric column-major G: one wavefront per column (= row), 16-byte loads, fixed summation order
__global__ __launch_bounds__(VEC_THREADS) void dense_symv_kernel(int64_t l, int64_t ld, const double* __restrict__ G, const double* __restrict__ t,
                                                                 double* __restrict__ w) {
    const int lane = threadIdx.x & 63;
    const int64_t col = blockIdx.x * (int64_t)(VEC_THREADS / 64) + (threadIdx.x >> 6);
    if (col >= l) return;
    const double* __restrict__ g = G + col * ld;     // (ld is a multiple of 64: every column starts 16-byte aligned)
    double acc = 0.0;
    const int64_t i0 = 0;
    const int64_t npair = (l - i0) / 2;
    typedef double v2d __attribute__((ext_vector_type(2)));
    const v2d* __restrict__ g2 = reinterpret_cast<const v2d*>(g + i0);
    for (int64_t q = lane; q < npair; q += 64) {
        const v2d a = __builtin_nontemporal_load(g2 + q);
        acc += a.x * t[i0 + 2 * q] + a.y * t[i0 + 2 * q + 1];
    }
    if (lane == 0 && i0 + 2 * npair < l) acc += g[l - 1] * t[l - 1];
    acc = wave_sum(acc);
    if (lane == 0) w[col] = acc;
}
// t = (Q u) - v from W = (u, Q u) and x = (u, v)
__global__ __launch_bounds__(VEC_THREADS) void direct_rhs_kernel(int64_t l, const d2* __restrict__ W, const d2* __restrict__ x, double* __restrict__ t) {
    GRID_STRIDE(i, l) t[i] = W[i].y - x[i].y;
}
// out = (u + Q w, v + w) from x = (u, v) and W = (w, Q w)
__global__ __launch_bounds__(VEC_THREADS) void direct_finish_kernel(int64_t l, const d2* __restrict__ x, const d2* __restrict__ W, d2* __restrict__ out) {
    GRID_STRIDE(i, l) { const d2 xi = x[i], wi = W[i]; out[i] = make_double2(xi.x + wi.y, xi.y + wi.x); }
}
// C = alpha A B + gamma D  for dense column-major L x L matrices (L a multiple of 64) on v_mfma_f64_16x16x4_f64: one 64 x 64 tile
// of C per workgroup, each of the 4 wavefronts a 32 x 32 quadrant (2 x 2 MFMA tiles), K in chunks of 16 staged through LDS.
// Set-up code (the Newton-Schulz iteration below runs it ~30 times per handle), not a per-iteration kernel.
typedef double v4d_t __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(VEC_THREADS) void dense_gemm_kernel(int L, double alpha, const double* __restrict__ A, const double* __restrict__ B,
                                                                 double gamma, const double* __restrict__ D, double* __restrict__ Cm) {
    __shared__ double As[16][64 + 2];            // As[k][i]
    __shared__ double Bs[16][64 + 2];            // Bs[k][j]
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int i0 = blockIdx.x * 64, j0 = blockIdx.y * 64;
    const int wr = (wv >> 1) * 32, wc = (wv & 1) * 32;
    const int lr = lane & 15, lk = lane >> 4;
    v4d_t acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) acc[a][b] = v4d_t{0.0, 0.0, 0.0, 0.0};
    for (int k0 = 0; k0 < L; k0 += 16) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int i = tid & 63, kk = (tid >> 6) + 4 * q;
            As[kk][i] = A[(size_t)(i0 + i) + (size_t)(k0 + kk) * L];
            const int k = tid & 15, j = (tid >> 4) + 16 * q;
            Bs[k][j] = B[(size_t)(k0 + k) + (size_t)(j0 + j) * L];
        }
        __syncthreads();
#pragma unroll
        for (int ks = 0; ks < 16; ks += 4) {
            double av[2], bv[2];
#pragma unroll
            for (int a = 0; a < 2; ++a) av[a] = As[ks + lk][wr + 16 * a + lr];
#pragma unroll
            for (int b = 0; b < 2; ++b) bv[b] = Bs[ks + lk][wc + 16 * b + lr];
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < 2; ++b) acc[a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[a], bv[b], acc[a][b], 0, 0, 0);
        }
        __syncthreads();
    }
    // result map of the f64 MFMA: column lane & 15, rows (lane >> 4) + 4 r
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const size_t idx = (size_t)(i0 + wr + 16 * a + lk + 4 * r) + (size_t)(j0 + wc + 16 * b + lr) * L;
                double v = alpha * acc[a][b][r];
                if (gamma != 0.0) v += gamma * D[idx];
                Cm[idx] = v;
            }
}
// max |G X - I| over all entries (Y = G X given): one partial per workgroup
__global__ __launch_bounds__(VEC_THREADS) void dense_resid_kernel(int64_t L, const double* __restrict__ Y, double* __restrict__ partials) {
    double mx = 0.0;
    for (int64_t e = blockIdx.x * (int64_t)VEC_THREADS + threadIdx.x; e < L * L; e += (int64_t)gridDim.x * VEC_THREADS) {
        const int64_t i = e % L, j = e / L;
        const double d = fabs(Y[e] - (i == j ? 1.0 : 0.0));
        mx = (d > mx || d != d) ? d : mx;
    }
    for (int off = 32; off > 0; off >>= 1) { const double o = __shfl_xor(mx, off, 64); mx = (o > mx || o != o) ? o : mx; }
    __shared__ double sm[4];
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = mx;
    __syncthreads();
    if (threadIdx.x == 0) {
        double m = sm[0];
        for (int w = 1; w < 4; ++w) m = (sm[w] > m || sm[w] != sm[w]) ? sm[w] : m;
        partials[blockIdx.x] = m;
    }
}
__global__ __launch_bounds__(VEC_THREADS) void dense_scale_identity_kernel(int64_t L, double* __restrict__ X, double s) {
    for (int64_t i = blockIdx.x * (int64_t)VEC_THREADS + threadIdx.x; i < L; i += (int64_t)gridDim.x * VEC_THREADS) X[i + i * L] = s;
}
void launch_dense_gemm(const LaunchCtx& c, int L, double alpha, const double* A, const double* B, double gamma, const double* D, double* Cm) {
    hipLaunchKernelGGL(dense_gemm_kernel, dim3(L / 64, L / 64), dim3(VEC_THREADS), 0, c.stream, L, alpha, A, B, gamma, D, Cm);
}
void launch_dense_resid(const LaunchCtx& c, int64_t L, const double* Y, double* partials, int nblocks) {
    hipLaunchKernelGGL(dense_resid_kernel, dim3(nblocks), dim3(VEC_THREADS), 0, c.stream, L, Y, partials);
}
void launch_dense_scale_identity(const LaunchCtx& c, int64_t L, double* X, double s) {
    hipLaunchKernelGGL(dense_scale_identity_kernel, dim3(64), dim3(VEC_THREADS), 0, c.stream, L, X, s);
}

void launch_dense_q_fill(const LaunchCtx& c, const int64_t* colptr, const int64_t* rowval, const double* nzval, double* Q, int64_t ld) {
    hipLaunchKernelGGL(dense_q_fill_kernel, dim3(1024), dim3(VEC_THREADS), 0, c.stream, c.m, c.n, colptr, rowval, nzval, c.cb, Q, ld);
}
void launch_dense_symv(const LaunchCtx& c, int64_t ld, const double* G, const double* t, double* w) {
    hipLaunchKernelGGL(dense_symv_kernel, dim3((unsigned)((c.l + 3) / 4)), dim3(VEC_THREADS), 0, c.stream, c.l, ld, G, t, w);
}
void launch_direct_rhs(const LaunchCtx& c, const double2* W, const double2* x, double* t) {
    hipLaunchKernelGGL(direct_rhs_kernel, dim3(c.vec_blocks), dim3(VEC_THREADS), 0, c.stream, c.l, W, x, t);
}
void launch_direct_finish(const LaunchCtx& c, const double2* x, const double2* W, double2* out) {
    hipLaunchKernelGGL(direct_finish_kernel, dim3(c.vec_blocks), dim3(VEC_THREADS), 0, c.stream, c.l, x, W, out);
}

// ---- direct = true on a BLOCK-SEPARABLE operator (HSDE.jl:12-15: S1 = IndAffine([Q -I], 0); fos_internal.hpp BlkDirect).  The exact projection
// (u^, Q u^), u^ = (I - Q^2)^-1 (u - Q v), with I - Q^2 = blkdiag(I + A'A, I + AA', delta) + W C W' (three border columns: solver.cpp
// prox_affine_direct_block), in three KKT sweeps: the small blocks of I + A'A are inverted once on the host, (I + AA')^-1 = I - A (I + A'A)^-1 A'.
// (1) from T = M (u, v) (T.x = g = u - Q v): W2 = (0, g restricted to the rows of A), W3 zeroed outside the x part, partial sums of ph.g and pg.g
__global__ __launch_bounds__(VEC_THREADS) void blkdir_prep_kernel(int64_t l, int64_t n, const d2* __restrict__ T, const d2* __restrict__ phg,
                                                                  d2* __restrict__ W2, d2* __restrict__ W3, double* __restrict__ partials) {
    double acc[3] = {0.0, 0.0, 0.0};          // (three-wide records: what the sharded reduce kernel takes)
    GRID_STRIDE(i, l) {
        const double g = T[i].x;
        const d2 ph = phg[i];
        if (i < l - 1) { acc[0] += ph.x * g; acc[1] += ph.y * g; }      // (the tau element is replicated on sharded handles; ph, pg are zero there anyway)
        const bool yrow = i >= n && i < l - 1;
        W2[i] = make_double2(0.0, yrow ? g : 0.0);
        if (i >= n) W3[i] = make_double2(0.0, 0.0);
    }
    block_reduce_store<3>(acc, partials + 3 * (int64_t)blockIdx.x);
}
// (2) one wavefront per diagonal block of I + A'A (order s <= 64, explicit inverse, column-major): (q, x^) = Ginv (a, g1) with a = A' g2 = -R.x and
//     g1 = T.x on the block's columns; W3 = (q, x^) there -- the two vectors the third sweep applies A to; ctx_rec[b] = the block's share of c'x^
//     (the tau row of the third apply is then never needed: no tau-row kernel, no deferred-row kernel behind that sweep)
__global__ __launch_bounds__(64) void blkdir_solve_kernel(const int64_t* __restrict__ goff, const int32_t* __restrict__ ioff, const int32_t* __restrict__ idx,
                                                          const double* __restrict__ Ginv, const d2* __restrict__ R, const d2* __restrict__ T, d2* __restrict__ W3,
                                                          const double* __restrict__ cb, double* __restrict__ ctx_rec) {
    const int b = blockIdx.x, lane = threadIdx.x;
    const int i0 = ioff[b], s = ioff[b + 1] - i0;
    __shared__ double ra[64], rg[64];
    int col = -1;
    if (lane < s) { col = idx[i0 + lane]; ra[lane] = -R[col].x; rg[lane] = T[col].x; }
    __syncthreads();
    double cx = 0.0;
    if (lane < s) {
        const double* __restrict__ G = Ginv + goff[b] + lane;
        double q = 0.0, xh = 0.0;
        for (int j = 0; j < s; ++j) { const double gij = G[(size_t)j * s]; q += gij * ra[j]; xh += gij * rg[j]; }
        W3[col] = make_double2(q, xh);
        cx = cb[col] * xh;
    }
    cx = wave_sum(cx);
    if (lane == 0) ctx_rec[b] = cx;
}
// kappa = S3inv [ph.g, pg.g, g_tau / delta] (the multipliers of the three border columns) from the prep kernel's records: EVERY workgroup of the kernels
// that need it adds the same records in the same order (no launch of its own).  prm = S3inv (9), delta
__device__ __forceinline__ void blkdir_kappa(const double* __restrict__ prep_partials, int count, const d2* __restrict__ T, int64_t l, const double* __restrict__ prm,
                                             int zero, double* kap /* shared [3] */, const double* __restrict__ reduced, int from_reduced) {
    // (sharded handles: the two dots were summed over the ranks into `reduced` between the prep kernel and here)
    __shared__ double sums[3];
    if (from_reduced) { if (threadIdx.x < 2) sums[threadIdx.x] = reduced[threadIdx.x]; __syncthreads(); }
    else reduce_partials<3>(prep_partials, count, sums);
    if (threadIdx.x == 0) {
        const double r[3] = {sums[0], sums[1], T[l - 1].x / prm[9]};
        for (int a = 0; a < 3; ++a) kap[a] = zero ? 0.0 : prm[3 * a] * r[0] + prm[3 * a + 1] * r[1] + prm[3 * a + 2] * r[2];
    }
    __syncthreads();
}
// (3) d = D^-1 g = (x^, g2 - A q, g_tau / delta), Q d = (q + c t^, -A x^ + b t^, .) from the sweeps' results; out = (d - k1 ph - k2 pg, Q d - k1 Qph - k2 Qpg - k3 [c; b] / delta)
//     for all rows but the tau row; partial sums of b . y^ for that one.  V = M (q, x^): V.y = -A q and V.x = A x^ on the rows of A.
__global__ __launch_bounds__(VEC_THREADS) void blkdir_combine_kernel(int64_t l, int64_t n, const d2* __restrict__ T, const d2* __restrict__ W3, const d2* __restrict__ V,
                                                                     const d2* __restrict__ phg, const d2* __restrict__ qphg, const double* __restrict__ cb,
                                                                     const double* __restrict__ prm, int zero, const double* __restrict__ prep_partials, int nprep,
                                                                     d2* __restrict__ out, double* __restrict__ partials, const double* __restrict__ reduced, int from_reduced,
                                                                     double* __restrict__ kap_out) {
    __shared__ double kap[3];
    blkdir_kappa(prep_partials, nprep, T, l, prm, zero, kap, reduced, from_reduced);
    const double k1 = kap[0], k2 = kap[1], k3 = kap[2], delta = prm[9];
    if (blockIdx.x == 0 && threadIdx.x < 3) kap_out[threadIdx.x] = kap[threadIdx.x];        // (for the tau-row kernel of a sharded handle: `reduced` is reused by then)
    const double th = T[l - 1].x / delta;
    double acc[1] = {0.0};
    GRID_STRIDE(i, l - 1) {
        double d, qd;
        const double c = cb[i];
        if (i < n) { const d2 w = W3[i]; d = w.y; qd = w.x + c * th; }
        else { const d2 v = V[i]; d = T[i].x + v.y; qd = -v.x + c * th; acc[0] += c * d; }
        const d2 ph = phg[i], qp = qphg[i];
        out[i] = make_double2(d - k1 * ph.x - k2 * ph.y, qd - k1 * qp.x - k2 * qp.y - k3 * c / delta);
    }
    block_reduce_store<1>(acc, partials + (int64_t)blockIdx.x);
}
// (4) the tau row: u^_tau = g_tau / delta - k3 / delta, (Q u^)_tau = -c'x^ - b'y^ - k1 (Q ph)_tau - k2 (Q pg)_tau   (c'x^ from the block records)
//     sharded handles: blkdir_tausum_kernel leaves this rank's c'x^ + b'y^ as one record, the reduce kernel sums it over the ranks into reduced[0],
//     and kappa comes from where the combine kernel left it (kap_in)
__global__ __launch_bounds__(FIN_THREADS) void blkdir_tausum_kernel(const double* __restrict__ partials, int count, const double* __restrict__ ctx_rec, int nblk,
                                                                    double* __restrict__ out1) {
    __shared__ double sums[1], sumc[1];
    reduce_partials<1>(partials, count, sums);
    reduce_partials<1>(ctx_rec, nblk, sumc);
    if (threadIdx.x == 0) out1[0] = sumc[0] + sums[0];
}
//     PEER (mailbox transports): this kernel does the three steps itself -- the rank's record, its exchange through the mailboxes (the single-workgroup
//     exchange of reduce_kernel<true>), the tau row -- ONE launch instead of three on the latency chain of every projection
template <bool PEER>
__global__ __launch_bounds__(FIN_THREADS) void blkdir_tau_kernel(const double* __restrict__ partials, int count, const double* __restrict__ ctx_rec, int nblk,
                                                                 const d2* __restrict__ T, const d2* __restrict__ qphg, int64_t l, const double* __restrict__ prm,
                                                                 int zero, const double* __restrict__ prep_partials, int nprep, d2* __restrict__ out,
                                                                 const double* __restrict__ reduced, int from_reduced, const double* __restrict__ kap_in,
                                                                 DevState* st, PeerBox pb) {
    __shared__ double kap[3];
    __shared__ double sums[1], sumc[1];
    if constexpr (PEER) {
        if (st->xchg_failed) return;
        __shared__ double loc[1];
        reduce_partials<1>(partials, count, sums);
        reduce_partials<1>(ctx_rec, nblk, sumc);
        if (threadIdx.x == 0) loc[0] = sumc[0] + sums[0];          // (the same expression as blkdir_tausum_kernel)
        if (threadIdx.x < 3) kap[threadIdx.x] = kap_in[threadIdx.x];
        __syncthreads();
        if (!peer_exchange_wg(pb, loc, 1, sums, st)) return;
        if (threadIdx.x == 0) sumc[0] = 0.0;
        __syncthreads();
    } else if (from_reduced) {
        if (threadIdx.x < 3) kap[threadIdx.x] = kap_in[threadIdx.x];
        if (threadIdx.x == 0) { sums[0] = reduced[0]; sumc[0] = 0.0; }
        __syncthreads();
    } else {
        blkdir_kappa(prep_partials, nprep, T, l, prm, zero, kap, nullptr, 0);
        reduce_partials<1>(partials, count, sums);
        reduce_partials<1>(ctx_rec, nblk, sumc);
    }
    if (threadIdx.x == 0) {
        const double k1 = kap[0], k2 = kap[1], k3 = kap[2], delta = prm[9];
        const d2 qp = qphg[l - 1];
        out[l - 1] = make_double2(T[l - 1].x / delta - k3 / delta, -sumc[0] - sums[0] - k1 * qp.x - k2 * qp.y);
    }
}
void launch_blkdir_prep(const LaunchCtx& c, const double2* T, const double2* phg, double2* W2, double2* W3, double* partials) {
    hipLaunchKernelGGL(blkdir_prep_kernel, dim3(c.vec_blocks), dim3(VEC_THREADS), 0, c.stream, c.l, c.n, T, phg, W2, W3, partials);
}
void launch_blkdir_solve(const LaunchCtx& c, int nblk, const int64_t* goff, const int32_t* ioff, const int32_t* idx, const double* Ginv, const double2* R,
                         const double2* T, double2* W3, double* ctx_rec) {
    if (nblk > 0) hipLaunchKernelGGL(blkdir_solve_kernel, dim3(nblk), dim3(64), 0, c.stream, goff, ioff, idx, Ginv, R, T, W3, c.cb, ctx_rec);
}
void launch_blkdir_combine(const LaunchCtx& c, const double2* T, const double2* W3, const double2* V, const double2* phg, const double2* qphg,
                           double* prm, int zero_kappa, double2* out, const double* prep_partials, double* partials, int from_reduced) {
    hipLaunchKernelGGL(blkdir_combine_kernel, dim3(c.vec_blocks), dim3(VEC_THREADS), 0, c.stream, c.l, c.n, T, W3, V, phg, qphg, c.cb, prm, zero_kappa, prep_partials,
                       c.vec_blocks, out, partials, c.reduced, from_reduced, prm + 10);
}
void launch_blkdir_tausum(const LaunchCtx& c, const double* partials, const double* ctx_rec, int nblk, double* out1) {
    hipLaunchKernelGGL(blkdir_tausum_kernel, dim3(1), dim3(FIN_THREADS), 0, c.stream, partials, c.vec_blocks, ctx_rec, nblk, out1);
}
// from_reduced: 0 = one GPU; 1 = the rank sums were all-reduced into c.reduced[0] (tausum + reduce + all-reduce in front); 2 = mailbox transport: this launch exchanges them itself
void launch_blkdir_tau(const LaunchCtx& c, const double2* T, const double2* qphg, const double* prm, int zero_kappa, double2* out, const double* prep_partials,
                       const double* partials, const double* ctx_rec, int nblk, int from_reduced) {
    if (from_reduced == 2 && c.peer)
        hipLaunchKernelGGL(blkdir_tau_kernel<true>, dim3(1), dim3(FIN_THREADS), 0, c.stream, partials, c.vec_blocks, ctx_rec, nblk, T, qphg, c.l, prm, zero_kappa, prep_partials,
                           c.vec_blocks, out, c.reduced, 1, prm + 10, c.st, *c.peer);
    else
        hipLaunchKernelGGL(blkdir_tau_kernel<false>, dim3(1), dim3(FIN_THREADS), 0, c.stream, partials, c.vec_blocks, ctx_rec, nblk, T, qphg, c.l, prm, zero_kappa, prep_partials,
                           c.vec_blocks, out, c.reduced, from_reduced, prm + 10, c.st, PeerBox{});
}

// ------------------------------------------------------------------------------------------------ layout conversion

__global__ __launch_bounds__(VEC_THREADS) void interleave_kernel(int64_t l, d2* __restrict__ out, const double* __restrict__ plain) {
    GRID_STRIDE(i, l) out[i] = make_double2(plain[i], plain[l + i]);
}
__global__ __launch_bounds__(VEC_THREADS) void deinterleave_kernel(int64_t l, double* __restrict__ plain, const d2* __restrict__ in) {
    GRID_STRIDE(i, l) { const d2 v = in[i]; plain[i] = v.x; plain[l + i] = v.y; }
}
__global__ __launch_bounds__(VEC_THREADS) void set_comp_kernel(int64_t l, d2* __restrict__ out, const double* __restrict__ plain, int comp) {
    GRID_STRIDE(i, l) { const double v = plain[i]; out[i] = comp ? make_double2(0.0, v) : make_double2(v, 0.0); }
}
__global__ __launch_bounds__(VEC_THREADS) void get_plain_kernel(int64_t l, double* __restrict__ plain, const d2* __restrict__ in, int comp) {
    GRID_STRIDE(i, l) { const d2 v = in[i]; plain[i] = comp ? v.y : v.x; }
}
void launch_interleave(const LaunchCtx& c, double2* out, const double* plain) {
    hipLaunchKernelGGL(interleave_kernel, dim3(c.vec_blocks), dim3(VEC_THREADS), 0, c.stream, c.l, out, plain);
}
void launch_deinterleave(const LaunchCtx& c, double* plain, const double2* in) {
    hipLaunchKernelGGL(deinterleave_kernel, dim3(c.vec_blocks), dim3(VEC_THREADS), 0, c.stream, c.l, plain, in);
}
void launch_set_comp(const LaunchCtx& c, double2* out, const double* plain_l, int comp) {
    hipLaunchKernelGGL(set_comp_kernel, dim3(c.vec_blocks), dim3(VEC_THREADS), 0, c.stream, c.l, out, plain_l, comp);
}
void launch_get_plain(const LaunchCtx& c, double* plain_l, const double2* in, int comp) {
    hipLaunchKernelGGL(get_plain_kernel, dim3(c.vec_blocks), dim3(VEC_THREADS), 0, c.stream, c.l, plain_l, in, comp);
}

// ------------------------------------------------------------------------------------------------ cones

// every index whose cone is Free/Zero/NonNeg/NonPos, plus the (tau,kappa) element; op byte: part1 | part2 << 2
__global__ __launch_bounds__(VEC_THREADS) void cones_elementwise_kernel(int64_t l, d2* __restrict__ out, const d2* __restrict__ in,
                                                                        const uint8_t* __restrict__ ew_op, const int32_t* __restrict__ gate) {
    if (gate && !*gate) return;                  // speculatively enqueued behind a CG batch that did not converge: no-op
    GRID_STRIDE(i, l) {
        const uint8_t op = ew_op[i];
        if (op == EW_SKIP) continue;
        const d2 v = in[i];
        out[i] = make_double2(ew_apply(op & 3, v.x), ew_apply((op >> 2) & 3, v.y));
    }
}
void launch_cones_elementwise(const LaunchCtx& c, double2* out, const double2* in, const uint8_t* ew_op) {
    hipLaunchKernelGGL(cones_elementwise_kernel, dim3(c.vec_blocks), dim3(VEC_THREADS), 0, c.stream, c.l, out, in, ew_op, c.gate);
}

// Second-order cones, one wavefront per cone, both copies (primal on one part, Moreau dual x + P(-x) on the
// other: cones.jl:80-85) from the same 16-byte loads.  IndSOC: t = first entry; 0 if t <= -||v||, x if t >= ||v||,
// else r = (1 + t/||v||)/2, y = (r ||v||, r v).  IndRotatedSOC: rotate entries (0,1) by pi/4, project, rotate back.
struct SocOut { double y0, y1, r; int kase; };   // kase 0: zero, 1: identity, 2: scale by r

__device__ __forceinline__ SocOut soc_decide(double t, double nx) {
    SocOut o;
    if (t <= -nx) { o.kase = 0; o.r = 0.0; o.y0 = 0.0; }
    else if (t >= nx) { o.kase = 1; o.r = 1.0; o.y0 = t; }
    else { o.kase = 2; o.r = 0.5 * (1.0 + t / nx); o.y0 = o.r * nx; }
    o.y1 = 0.0;
    return o;
}

constexpr double S45 = 0.7071067811865475;

__global__ __launch_bounds__(VEC_THREADS) void cones_soc_kernel(d2* __restrict__ out, const d2* __restrict__ in,
                                                                const ConeDesc* __restrict__ cones, int ncones, const int32_t* __restrict__ gate) {
    if (gate && !*gate) return;                  // speculatively enqueued behind a CG batch that did not converge: no-op
    const int lane = threadIdx.x & 63;
    const int cone = blockIdx.x * (VEC_THREADS / 64) + (threadIdx.x >> 6);
    if (cone >= ncones) return;
    const ConeDesc cd = cones[cone];
    const d2* __restrict__ x = in + cd.start;
    d2* __restrict__ y = out + cd.start;
    const int len = cd.len;
    const bool rot = cd.type == FOS_CONE_SOCROT;
    const int head = rot ? 2 : 1;
    // sign applied to the input of each part: the dual part projects -x
    const double sg0 = (cd.dual_part == 0) ? -1.0 : 1.0, sg1 = (cd.dual_part == 1) ? -1.0 : 1.0;
    double n0 = 0.0, n1 = 0.0;
    for (int k = head + lane; k < len; k += 64) {
        const d2 v = x[k];
        n0 += v.x * v.x;
        n1 += v.y * v.y;
    }
    n0 = wave_sum(n0);
    n1 = wave_sum(n1);
    const d2 h0 = x[0];
    d2 h1 = make_double2(0.0, 0.0);
    if (rot) h1 = x[1];
    double t0, t1, x20 = 0.0, x21 = 0.0;     // t = first (rotated) entry of the (sign-flipped) input
    if (rot) {
        const double a0 = sg0 * h0.x, b0 = sg0 * h1.x, a1 = sg1 * h0.y, b1 = sg1 * h1.y;
        t0 = S45 * a0 + S45 * b0; x20 = S45 * a0 - S45 * b0;
        t1 = S45 * a1 + S45 * b1; x21 = S45 * a1 - S45 * b1;
        n0 += x20 * x20;
        n1 += x21 * x21;
    } else {
        t0 = sg0 * h0.x;
        t1 = sg1 * h0.y;
    }
    const double nx0 = sqrt(n0), nx1 = sqrt(n1);
    const SocOut o0 = soc_decide(t0, nx0), o1 = soc_decide(t1, nx1);
    // tail entries
    for (int k = head + lane; k < len; k += 64) {
        const d2 v = x[k];
        double p0 = (o0.kase == 0) ? 0.0 : ((o0.kase == 1) ? sg0 * v.x : o0.r * (sg0 * v.x));
        double p1 = (o1.kase == 0) ? 0.0 : ((o1.kase == 1) ? sg1 * v.y : o1.r * (sg1 * v.y));
        if (cd.dual_part == 0) p0 = v.x + p0;     // y = x + P(-x)    cones.jl:81-84
        if (cd.dual_part == 1) p1 = v.y + p1;
        y[k] = make_double2(p0, p1);
    }
    if (lane == 0) {
        double y00, y01 = 0.0, y10, y11 = 0.0;    // yPQ: part P, entry Q
        if (rot) {
            double a0 = o0.y0, b0 = (o0.kase == 0) ? 0.0 : ((o0.kase == 1) ? x20 : o0.r * x20);
            double a1 = o1.y0, b1 = (o1.kase == 0) ? 0.0 : ((o1.kase == 1) ? x21 : o1.r * x21);
            y00 = S45 * a0 + S45 * b0; y01 = S45 * a0 - S45 * b0;
            y10 = S45 * a1 + S45 * b1; y11 = S45 * a1 - S45 * b1;
        } else {
            y00 = o0.y0;
            y10 = o1.y0;
        }
        if (cd.dual_part == 0) { y00 = h0.x + y00; y01 = h1.x + y01; }
        if (cd.dual_part == 1) { y10 = h0.y + y10; y11 = h1.y + y11; }
        y[0] = make_double2(y00, y10);
        if (rot) y[1] = make_double2(y01, y11);
    }
}
// Exponential cone K = cl{(r,s,t): s > 0, s exp(r/s) <= t}: ProximalOperators' IndExpPrimal (a port of the SCS
// projection: closed-form cases, else bisection on the dual variable rho with a 1-D Newton solve inside, tolerance
// 1e-15, 100 iterations each); IndExpDual = Moreau  x + P_K(-x).  One thread per (cone, part).
constexpr double EXP_TOL = 1e-15;
constexpr int EXP_MAXIT = 100;

// (No FMA contraction in these three functions, and the same order of operations as the oracle's restatement: for points within ~1e-8 of the cone the dual variable
//  rho is that small, 1 / rho^2 multiplies rounding errors by 1e16 and the Newton iteration ends on noise -- an alternating projection that converges onto the cone
//  then sees the device and the CPU restatement drift apart by 1e-7 within twenty iterations (tests/fuzz_parity.py, Feasibility seed 20439) unless both round alike.)
__device__ double exp_newton_onz(double rho, double y_hat, double z_hat, double w) {
#pragma clang fp contract(off)
    double t = fmax(fmax(w - z_hat, -z_hat), EXP_TOL);
    for (int it = 0; it < EXP_MAXIT; ++it) {
        const double f = (1.0 / (rho * rho)) * t * (t + z_hat) - y_hat / rho + log(t / rho) + 1.0;
        const double fp = (1.0 / (rho * rho)) * (2.0 * t + z_hat) + 1.0 / t;
        t = t - f / fp;
        if (t <= -z_hat) { t = -z_hat; break; }
        else if (t <= 0) { t = 0.0; break; }
        else if (fabs(f) < EXP_TOL) break;
    }
    return t + z_hat;
}
__device__ double exp_calc_grad(const double* v, double rho, double warm2, double* x) {
#pragma clang fp contract(off)
    x[2] = exp_newton_onz(rho, v[1], v[2], warm2);
    x[1] = (1.0 / rho) * (x[2] - v[2]) * x[2];
    x[0] = v[0] - rho;
    return (x[1] == 0) ? x[0] : x[0] + x[1] * log(x[1] / x[2]);
}
__device__ void exp_project(const double* v, double* y) {
#pragma clang fp contract(off)
    const double r = v[0], s = v[1], t = v[2];
    if ((s > 0 && s * exp(r / s) <= t) || (r <= 0 && s == 0 && t >= 0)) { y[0] = r; y[1] = s; y[2] = t; return; }
    if ((-r < 0 && r * exp(s / r) <= -2.718281828459045 * t) || (-r == 0 && -s >= 0 && -t >= 0)) { y[0] = y[1] = y[2] = 0.0; return; }
    if (r < 0 && s < 0) { y[0] = r; y[1] = fmax(s, 0.0); y[2] = fmax(t, 0.0); return; }
    double z[3], lb = 0.0, rho = 0.125;
    double g = exp_calc_grad(v, rho, v[1], z);
    while (g > 0) {                       // getRhoUb
        lb = rho;
        rho = rho * 2;
        g = exp_calc_grad(v, rho, z[1], z);
    }
    double ub = rho;
    z[1] = v[1];                          // (the bisection's first Newton solve starts from the point itself again, as in the restatement: getRhoUb keeps its iterate to itself)
    for (int it = 0; it < EXP_MAXIT; ++it) {
        rho = (ub + lb) / 2;
        g = exp_calc_grad(v, rho, z[1], z);
        if (g > 0) lb = rho; else ub = rho;
        if (ub - lb < EXP_TOL) break;
    }
    y[0] = z[0]; y[1] = z[1]; y[2] = z[2];
}
// prox of the cone named by `type` (IndExpPrimal, or IndExpDual = x + P_K(-x))
__device__ void exp_prox(int type, const double* x, double* y) {
    if (type == FOS_CONE_EXPPRIMAL) { exp_project(x, y); return; }
    double nx[3] = {-x[0], -x[1], -x[2]}, t[3];
    exp_project(nx, t);
    y[0] = x[0] + t[0]; y[1] = x[1] + t[1]; y[2] = x[2] + t[2];
}
__global__ __launch_bounds__(64) void cones_exp_kernel(d2* __restrict__ out, const d2* __restrict__ in,
                                                       const ConeDesc* __restrict__ cones, int ncones, const int32_t* __restrict__ gate) {
    if (gate && !*gate) return;                  // speculatively enqueued behind a CG batch that did not converge: no-op
    const int job = blockIdx.x * 64 + threadIdx.x;
    if (job >= 2 * ncones) return;
    const ConeDesc cd = cones[job >> 1];
    const int part = job & 1;
    const double* xin = reinterpret_cast<const double*>(in + cd.start) + part;
    double* yout = reinterpret_cast<double*>(out + cd.start) + part;
    const double x[3] = {xin[0], xin[2], xin[4]};
    double y[3];
    if (cd.dual_part == part) {           // proxDual: y = x + prox(-x)      cones.jl:80-85
        const double nx[3] = {-x[0], -x[1], -x[2]};
        double t[3];
        exp_prox(cd.type, nx, t);
        y[0] = x[0] + t[0]; y[1] = x[1] + t[1]; y[2] = x[2] + t[2];
    } else {
        exp_prox(cd.type, x, y);
    }
    yout[0] = y[0]; yout[2] = y[1]; yout[4] = y[2];
}
void launch_cones_exp(const LaunchCtx& c, double2* out, const double2* in, const ConeDesc* cones, int ncones) {
    if (ncones <= 0) return;
    hipLaunchKernelGGL(cones_exp_kernel, dim3((2 * ncones + 63) / 64), dim3(64), 0, c.stream, out, in, cones, ncones, c.gate);
}

void launch_cones_soc(const LaunchCtx& c, double2* out, const double2* in, const ConeDesc* cones, int ncones) {
    if (ncones <= 0) return;
    const int per_block = VEC_THREADS / 64;
    hipLaunchKernelGGL(cones_soc_kernel, dim3((ncones + per_block - 1) / per_block), dim3(VEC_THREADS), 0, c.stream, out, in, cones, ncones, c.gate);
}

}  // namespace fos
