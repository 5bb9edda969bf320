// Device helpers shared by kernels.hip and vecops.hip: wavefront / workgroup reductions in a fixed order, DPP lane-group
// sums, the folded peer-mailbox exchange, and the scalar bookkeeping that closes a CG iteration.
#pragma once

#include "fos_internal.hpp"

namespace fos {

typedef double2 d2;

template <int CTRL>
__device__ __forceinline__ double dpp_f64(double v) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    // (bound_ctrl = true: a source lane that does not exist or is switched off reads as 0 -- what `old = 0` said before, but without the
    //  v_mov_b32 dst, 0 the tied `old` operand cost in front of EVERY DPP move: 160 of the ~900 vector instructions of a resident tile sweep)
    lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xF, 0xF, true);
    hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xF, 0xF, true);
    return __hiloint2double(hi, lo);
}
// v + (v of the lane 16 / 32 away), every lane, without the LDS crossbar: gfx950's v_permlane{16,32}_swap exchanges the odd
// 16-lane rows (the upper 32 lanes) of one register with the even rows (the lower 32 lanes) of another; fed the same value
// twice it leaves [r0 r0 r2 r2] / [r1 r1 r3 r3] (resp. [lo lo] / [hi hi]), whose sum is the pairwise total in every lane.
template <int W>
__device__ __forceinline__ double swap_sum(double v) {
    const unsigned lo = (unsigned)__double2loint(v), hi = (unsigned)__double2hiint(v);
    if constexpr (W == 16) {
        const auto a = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
        const auto b = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
        return __hiloint2double((int)b[0], (int)a[0]) + __hiloint2double((int)b[1], (int)a[1]);
    } else {
        const auto a = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
        const auto b = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
        return __hiloint2double((int)b[0], (int)a[0]) + __hiloint2double((int)b[1], (int)a[1]);
    }
}

// The sum over the 64 lanes in EVERY lane (the same bits in each: every step adds a lane's value and its partner's, commutatively): four DPP steps inside the
// rows of 16 lanes, then the two row exchanges of gfx950 -- no trip through the LDS crossbar (the six ds_bpermute round trips of a __shfl_xor butterfly were
// ~0.3 us per sum: three of them sat at the tail of every workgroup of every sweep and in the prologue of every CG vector kernel).
__device__ __forceinline__ double wave_sum(double v) {
    v += dpp_f64<0xB1>(v);       // quad_perm [1,0,3,2]
    v += dpp_f64<0x4E>(v);       // quad_perm [2,3,0,1]
    v += dpp_f64<0x141>(v);      // row_half_mirror
    v += dpp_f64<0x140>(v);      // row_mirror
    v = swap_sum<16>(v);
    v = swap_sum<32>(v);
    return v;
}
// sum over aligned groups of `tpr` lanes (tpr a power of two, wave-uniform); every lane of a group gets the total
__device__ __forceinline__ double group_sum(double v, int tpr) {
    if (tpr >= 2) v += dpp_f64<0xB1>(v);       // quad_perm [1,0,3,2]
    if (tpr >= 4) v += dpp_f64<0x4E>(v);       // quad_perm [2,3,0,1]
    if (tpr >= 8) v += dpp_f64<0x141>(v);      // row_half_mirror
    if (tpr >= 16) v += dpp_f64<0x140>(v);     // row_mirror
    if (tpr >= 32) v = swap_sum<16>(v);
    if (tpr >= 64) v = swap_sum<32>(v);
    return v;
}

// Column sums of a dual tile.  Every lane holds p[u] = (its row's value in column u) x (its row's vector element) for 8
// consecutive columns; returns, in EVERY lane, the sum over all 64 lanes for column (lane & 7).  Three halving stages
// inside each group of 8 lanes (a lane keeps half of its values and trades the other half with a partner that keeps
// the complementary half: i <-> 7-i, i <-> i^2, i <-> i^1), then three full additions (i^8, i^16, i^32): 7 + 3 adds
// instead of 8 x 6, and a fixed summation order.
__device__ __forceinline__ double tile_colsum8(const double (&p)[8], int lane) {
    const bool b2 = lane & 4, b1 = lane & 2, b0 = lane & 1;
    double q[4], r[2];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const double keep = b2 ? p[4 + j] : p[j], send = b2 ? p[j] : p[4 + j];
        q[j] = keep + dpp_f64<0x141>(send);            // row_half_mirror: columns 4 b2 + j
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const double keep = b1 ? q[2 + j] : q[j], send = b1 ? q[j] : q[2 + j];
        r[j] = keep + dpp_f64<0x4E>(send);             // quad_perm [2,3,0,1]: columns 4 b2 + 2 b1 + j
    }
    const double keep = b0 ? r[1] : r[0], send = b0 ? r[0] : r[1];
    double s = keep + dpp_f64<0xB1>(send);             // quad_perm [1,0,3,2]: column lane & 7
    s += dpp_f64<0x128>(s);                            // row_ror:8  (lane ^ 8)
    s = swap_sum<16>(s);                               // + the neighbouring row of 16 lanes
    s = swap_sum<32>(s);                               // + the other half of the wavefront
    return s;
}

// The same sums for products that are ALREADY in the lane's own order (the resident CG kernel chooses the register layout of its tiles: position k of
// a group holds column (lane & 7) ^ TILE_SORT[k]), so that every halving stage keeps the first half of what it holds and trades the second: no selects
// (4 v_cndmask per pair in tile_colsum8: 224 of a tile sweep's instructions).  Same pairs, same order of additions, same result in lane (lane & 7).
__device__ __forceinline__ constexpr int tile_sort(int k) { return k < 4 ? k : 11 - k; }          // {0, 1, 2, 3, 7, 6, 5, 4}
__device__ __forceinline__ double tile_colsum8_sorted(const double (&p)[8]) {
    double q[4], r[2];
#pragma unroll
    for (int j = 0; j < 4; ++j) q[j] = p[j] + dpp_f64<0x141>(p[4 + j]);      // row_half_mirror
#pragma unroll
    for (int j = 0; j < 2; ++j) r[j] = q[j] + dpp_f64<0x4E>(q[2 + j]);       // quad_perm [2,3,0,1]
    double s = r[0] + dpp_f64<0xB1>(r[1]);                                   // quad_perm [1,0,3,2]: column lane & 7
    s += dpp_f64<0x128>(s);                                                  // row_ror:8  (lane ^ 8)
    s = swap_sum<16>(s);
    s = swap_sum<32>(s);
    return s;
}

// sums partials[count][NACC] -> sums[NACC] (shared) in a fixed order; any block size that is a multiple of 64, <= 1024.
// Three pieces, so that a kernel on a latency chain can request the records early and add them late:
//   PartialRegs::load   all loads of a thread's first U records (one memory round trip for up to U x blockDim records instead of
//                       one per record), no use of the values yet;
//   PartialRegs::sum    the thread's sum in record order (records beyond U x blockDim by a plain loop) -- the same summation
//                       order as a strided loop over all records;
//   partials_combine    wavefront butterflies + one LDS pass.
template <int NACC, int U = (NACC <= 3 ? 8 : 4)>
struct PartialRegs {
    double v[U][NACC];
    __device__ __forceinline__ void load(const double* __restrict__ partials, int count) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int i = threadIdx.x + u * blockDim.x;
#pragma unroll
            for (int a = 0; a < NACC; ++a) v[u][a] = (i < count) ? partials[(int64_t)i * NACC + a] : 0.0;
        }
    }
    __device__ __forceinline__ void sum(const double* __restrict__ partials, int count, double (&acc)[NACC]) const {
#pragma unroll
        for (int a = 0; a < NACC; ++a) acc[a] = 0.0;
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if ((int)(threadIdx.x + u * blockDim.x) < count) {
#pragma unroll
                for (int a = 0; a < NACC; ++a) acc[a] += v[u][a];
            }
        }
        for (int i = threadIdx.x + U * blockDim.x; i < count; i += blockDim.x) {
#pragma unroll
            for (int a = 0; a < NACC; ++a) acc[a] += partials[(int64_t)i * NACC + a];
        }
    }
};
template <int NACC>
__device__ __forceinline__ void partials_combine(const double (&acc)[NACC], double* sums);
template <int NACC>
__device__ __forceinline__ void reduce_partials(const double* __restrict__ partials, int count, double* sums) {
    PartialRegs<NACC> regs;
    regs.load(partials, count);
    double acc[NACC];
    regs.sum(partials, count, acc);
    partials_combine<NACC>(acc, sums);
}
template <int NACC>
__device__ __forceinline__ void partials_combine(const double (&acc)[NACC], double* sums) {
    __shared__ double smem[16 * NACC];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
#pragma unroll
    for (int a = 0; a < NACC; ++a) {
        double v = wave_sum(acc[a]);
        if (lane == 0) smem[wave * NACC + a] = v;
    }
    __syncthreads();
    if (threadIdx.x < NACC) {
        double s = 0.0;
        for (int w = 0; w < nw; ++w) s += smem[w * NACC + threadIdx.x];
        sums[threadIdx.x] = s;
    }
    __syncthreads();
}

// The exchange of a SINGLE workgroup (region 0 of the mailboxes: reduce_kernel<true>, blkdir_tau_kernel<true>): sums[0..nacc) (shared) -> every rank's
// mailbox, poll the own mailbox for the peers' words of the next sequence number, add in rank order -> out[0..nacc) (the same bits on every rank).  Returns
// false (the solve is stopped, the host reports it) when a peer does not answer in time.  Call with all threads of the workgroup.
__device__ __forceinline__ bool peer_exchange_wg(const PeerBox& pb, const double* sums, int nacc, double* out, DevState* st) {
    __shared__ uint32_t xw_halves[PEER_MAX_RANKS * PEER_MAX_VALS * 2];
    __shared__ int xw_failed;
    const int t = threadIdx.x;
    if (t == 0) xw_failed = 0;
    __syncthreads();
    const uint32_t seq = *pb.seq + 1u;                 // this exchange (same number on every rank)
    const size_t par = (size_t)(seq & 1u) * PEER_MAX_RANKS;
    if (t < pb.nranks * nacc * 2) {                    // thread = (peer rank r, value v, half hh)
        const int hh = t & 1, v = (t >> 1) % nacc, r = (t >> 1) / nacc;
        const unsigned long long bits = (unsigned long long)__double_as_longlong(sums[v]);
        const unsigned long long word = ((unsigned long long)seq << 32) | (hh ? (bits >> 32) : (bits & 0xFFFFFFFFull));
        unsigned long long* dst = pb.box[r] + ((par + pb.rank) * PEER_MAX_VALS + v) * 2 + hh;
        __hip_atomic_store(dst, word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        const unsigned long long* src = pb.box[pb.rank] + ((par + r) * PEER_MAX_VALS + v) * 2 + hh;
        const long long t0 = wall_clock64();
        unsigned long long w;
        bool ok;
        do {
            w = __hip_atomic_load(src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            ok = (uint32_t)(w >> 32) == seq;
        } while (!ok && (wall_clock64() - t0) < pb.timeout_ticks);
        if (!ok) xw_failed = 1;
        xw_halves[(r * PEER_MAX_VALS + v) * 2 + hh] = (uint32_t)w;
    }
    __syncthreads();
    if (xw_failed) {                                   // a peer never arrived: stop the solve, the host reports it
        if (t == 0) { st->xchg_failed = 1; st->done = 1; }
        return false;
    }
    if (t < nacc) {
        double s = 0.0;
        for (int r = 0; r < pb.nranks; ++r) {          // rank order: every rank computes the same bits
            const unsigned long long lo = xw_halves[(r * PEER_MAX_VALS + t) * 2], hi = xw_halves[(r * PEER_MAX_VALS + t) * 2 + 1];
            s += __longlong_as_double((long long)((hi << 32) | lo));
        }
        out[t] = s;
    }
    if (t == 0) *pb.seq = seq;
    return true;
}

// Folded peer exchange (fos_internal.hpp, region 1 of the mailboxes): called by EVERY workgroup of a CG kernel with the
// same local sums; workgroup 0 also stores them into the peers' mailboxes; all poll their own mailbox for the peers' words of
// sequence number `seq` and add in rank order.  Returns false (and stops the solve) when a peer does not answer in time.
template <int NACC>
__device__ __forceinline__ bool peer_fold_sum(const PeerBox& pb, uint32_t seq, double* sums /* shared: in local, out total */, DevState* st) {
    __shared__ uint32_t halves[PEER_MAX_RANKS * NACC * 2];
    __shared__ int failed;
    const int t = threadIdx.x;
    if (t == 0) failed = 0;
    __syncthreads();
    // four slots: (CG solve number & 1, sequence number & 1).  Inside a solve consecutive exchanges alternate in the low bit; the
    // first exchange of the NEXT solve takes the other pair, so it can never overwrite words a slower peer is still polling for
    // (the merged-reduction recurrence ends a solve on either parity)
    const size_t par = (size_t)((((seq >> 11) & 1u) << 1) | (seq & 1u)) * PEER_MAX_RANKS;
    if (t < pb.nranks * NACC * 2) {
        const int hh = t & 1, v = (t >> 1) % NACC, r = (t >> 1) / NACC;
        const unsigned long long bits = (unsigned long long)__double_as_longlong(sums[v]);
        const uint32_t mine = hh ? (uint32_t)(bits >> 32) : (uint32_t)bits;
        if (r == pb.rank && !pb.loopback) {
            halves[(r * NACC + v) * 2 + hh] = mine;
        } else {
            const size_t off_in = PEER_BOX_WORDS + ((par + r) * PEER_MAX_VALS + v) * 2 + hh;          // where rank r's word arrives
            if (blockIdx.x == 0) {
                // one segment for all ranks (`shared`): the word is written once -- by the thread of the first peer (or, looping back, of the rank itself)
                const bool wr = !pb.shared || pb.loopback ? (r != pb.rank || pb.loopback) : (r == (pb.rank == 0 ? 1 : 0));
                if (wr) {
                    unsigned long long* dst = pb.box[r] + PEER_BOX_WORDS + ((par + pb.rank) * PEER_MAX_VALS + v) * 2 + hh;
                    __hip_atomic_store(dst, ((unsigned long long)seq << 32) | mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                }
            }
            // host-pinned transport: only workgroup 0 reads the segment (a PCIe round trip per poll); everybody else polls its republication
            const bool via_relay = pb.relay != nullptr && blockIdx.x != 0;
            const unsigned long long* src = via_relay ? pb.relay + off_in : pb.box[pb.rank] + off_in;
            const long long t0 = wall_clock64();
            unsigned long long w;
            bool ok;
            do {
                w = __hip_atomic_load(src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                ok = (uint32_t)(w >> 32) == seq;
            } while (!ok && (wall_clock64() - t0) < pb.timeout_ticks);
            if (!ok) failed = 1;
            else if (pb.relay != nullptr && blockIdx.x == 0) __hip_atomic_store(pb.relay + off_in, w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            halves[(r * NACC + v) * 2 + hh] = (uint32_t)w;
        }
    }
    __syncthreads();
    if (failed) {
        if (blockIdx.x == 0 && t == 0) { st->xchg_failed = 1; st->done = 1; }
        return false;
    }
    if (t < NACC) {
        double s = 0.0;
        for (int r = 0; r < pb.nranks; ++r) {
            const unsigned long long lo = halves[(r * NACC + t) * 2], hi = halves[(r * NACC + t) * 2 + 1];
            s += __longlong_as_double((long long)((hi << 32) | lo));
        }
        sums[t] = s;
    }
    __syncthreads();
    return true;
}

// A lane's share of a slot-spread row's list (DefRow, fos_internal.hpp): list elements lig, lig + lpr, ... in chunks of four --
// the four slot numbers (computed for a progression, loaded for an explicit list), then the four slot loads, then the adds in list
// order.  On small operators this loop is the latency of the kernels that add the lists.
__device__ __forceinline__ DefRow ld_defrow(const DefRow* p) {
    typedef int v4i_ __attribute__((ext_vector_type(4)));
    union { v4i_ q[2]; DefRow d; } u;
    const v4i_* src = reinterpret_cast<const v4i_*>(p);
    u.q[0] = src[0]; u.q[1] = src[1];
    return u.d;
}
__device__ __forceinline__ int defrow_slot(const DefRow& r, const int32_t* __restrict__ def_idx, int e) {
    const int k = e - (r.own >= 0 ? 1 : 0);
    if (k < 0) return r.own;
    return r.stride != DEF_EXPLICIT ? r.base + k * r.stride : def_idx[r.kidx + k];
}
__device__ __forceinline__ void slot_list_sum(const d2* __restrict__ slots, const int32_t* __restrict__ def_idx, const DefRow& r, int e0, int lpr,
                                              double& u1, double& u2) {
    const int n = r.count + (r.own >= 0 ? 1 : 0);
    for (int e = e0; e < n; e += 4 * lpr) {
        int id[4];
        d2 v[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) id[q] = (e + q * lpr < n) ? defrow_slot(r, def_idx, e + q * lpr) : -1;
#pragma unroll
        for (int q = 0; q < 4; ++q) v[q] = id[q] >= 0 ? slots[id[q]] : make_double2(0.0, 0.0);
#pragma unroll
        for (int q = 0; q < 4; ++q) if (id[q] >= 0) { u1 += v[q].x; u2 += v[q].y; }
    }
}

// Values that cross workgroups INSIDE a kernel: the L2s of the XCDs are not coherent with each other between kernel boundaries,
// so such a value is written through to memory and read past the caches (agent-scope accesses).
__device__ __forceinline__ void st_coh(double* p, double v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ double ld_coh(const double* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// What closes CG iteration `jd` (conjugategradients.jl:42-50): r.r from the partial sums the x,r update left, the stop test
// `norm(r) <= tol || iter >= max_iters`, and -- if CG goes on -- beta = rn / rnold.  EVERY workgroup of the calling kernel
// evaluates it on the same records in the same order (so all take the same decision); workgroup 0 stores the scalars.
// r.r ping-pongs: RN[k] = r.r after iteration k (RN[0]: the start) lives in st->rn2[k & 1]; iteration j's alpha reads RN[j-1].
struct CgClose { double beta; bool stop; bool ok; };

// <= 1024 per-workgroup records added by EVERY wavefront itself (lane-strided, then the butterfly): no LDS, no workgroup
// barrier, all loads in flight together; the same bits in every wavefront of the grid
__device__ __forceinline__ double wave_sum_records(const double* __restrict__ rec, int count) {
    const int lane = threadIdx.x & 63;
    double v[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) v[k] = (lane + 64 * k < count) ? rec[lane + 64 * k] : 0.0;
    // (the sum starts from v[0], not from 0.0: the compiler placed `0.0 + v[0]` -- not foldable, -0.0 -- inside the predicated block of the FIRST load, with a
    //  vmcnt(0) behind it, so the other fifteen loads went out only after that round trip: two memory latencies instead of one in every prologue that closes a CG iteration)
    double s = v[0];
#pragma unroll
    for (int k = 1; k < 16; ++k) s += v[k];
    return wave_sum(s);
}

// CG has stopped at iteration jd (one thread of one workgroup): iteration count, flag, and the record the host may be spinning on
__device__ __forceinline__ void cg_signal_stop(DevState* st, int jd, int maxit, double rr, uint32_t epoch) {
    st->iter = jd;
    st->hit_max = (jd == maxit) ? 1 : 0;                          // conjugategradients.jl:53
    __threadfence();
    st->done = 1;
    if (st->hostmark) {                                           // tell the host directly
        HostMark* m = reinterpret_cast<HostMark*>(st->hostmark);
        m->iter = jd; m->hit_max = (jd == maxit) ? 1 : 0; m->rr = rr;
        __threadfence_system();
        __hip_atomic_store(&m->seq, epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

// (two pieces: everything cg_close_iteration reads -- all of it stored by EARLIER launches -- can be requested together with a
// kernel's gate and first elements, one memory round trip for the whole prologue)
struct CgCloseIn { d2 rt; double rnold, tol, s; int maxit; };
__device__ __forceinline__ CgCloseIn cg_close_request(DevState* st, const double* __restrict__ rr_partials, int count,
                                                      const double* __restrict__ reduced, int from_reduced,
                                                      const d2* __restrict__ r, int64_t l, int jd) {
    CgCloseIn in;
    in.rt = r[l - 1];
    in.rnold = st->rn2[(jd - 1) & 1]; in.tol = st->tol;
    in.maxit = st->maxit;
    in.s = from_reduced ? reduced[0] : wave_sum_records(rr_partials, count);
    return in;
}
__device__ __forceinline__ CgClose cg_close_finish(DevState* st, const CgCloseIn& in, int jd, const PeerBox& pb, uint32_t seq_base) {
    const d2 rt = in.rt;
    const double rnold = in.rnold, tol = in.tol;
    const int maxit = in.maxit;
    double s = in.s;
    CgClose c{0.0, true, true};
    if (pb.nranks > 0) {
        __shared__ double sums[1];
        if (threadIdx.x == 0) sums[0] = s;
        __syncthreads();
        if (!peer_fold_sum<1>(pb, seq_base + 2u * (uint32_t)jd + 1u, sums, st)) { c.ok = false; return c; }
        s = sums[0];
    }
    const double rr = s + (rt.x * rt.x + rt.y * rt.y);
    c.stop = (sqrt(rr) <= tol) || (jd >= maxit);                  // :42
    c.beta = rr / rnold;                                          // :46-48
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        st->rr = rr;
        if (c.stop) {
            cg_signal_stop(st, jd, maxit, rr, seq_base >> 11);
        } else {
            st->rn_old = rnold;
            st->rn = rr;
            st->rn2[jd & 1] = rr;
            st->beta = c.beta;
            st->iter = jd + 1;                                    // :50
        }
    }
    return c;
}

__device__ __forceinline__ CgClose cg_close_iteration(DevState* st, const double* __restrict__ rr_partials, int count,
                                                      const double* __restrict__ reduced, int from_reduced,
                                                      const d2* __restrict__ r, int64_t l, int jd, const PeerBox& pb, uint32_t seq_base) {
    const CgCloseIn in = cg_close_request(st, rr_partials, count, reduced, from_reduced, r, l, jd);
    return cg_close_finish(st, in, jd, pb, seq_base);
}

// Merged-reduction CG, single GPU (CgmIter::close_in_update == false): the SWEEP behind the update of iteration jd closes it --
// r.r from that update's records, the stop test of conjugategradients.jl:42; every workgroup evaluates it on the same records in
// the same order and returns true when CG has stopped (then nothing is swept).  Workgroup 0 stores g_jd for the next update.
__device__ __forceinline__ bool cgm_close_in_sweep(DevState* st, const double* __restrict__ rr_partials, int count,
                                                   const d2* __restrict__ r, int64_t l, int jd, uint32_t epoch, int32_t batch_mark) {
    // (everything is requested before the gate on `done` is evaluated: one memory round trip for the whole prologue)
    const d2 rt = r[l - 1];
    const double tol = st->tol;
    const int maxit = st->maxit, done = st->done;
    const double rr = wave_sum_records(rr_partials, count) + (rt.x * rt.x + rt.y * rt.y);
    if (done) return true;
    if (jd == 0) {                                                // the start: g_0 = r_0.r_0, no stop test (conjugategradients.jl:35-36)
        if (blockIdx.x == 0 && threadIdx.x == 0) { st->rn2[0] = rr; st->rn = rr; }
        return false;
    }
    const bool stop = (sqrt(rr) <= tol) || (jd >= maxit);
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        st->rr = rr;
        if (stop) {
            cg_signal_stop(st, jd, maxit, rr, epoch);
        } else {
            st->rn2[jd & 1] = rr;
            st->iter = jd + 1;
            if (batch_mark != 0 && st->hostmark)                  // the host's batch is used up, CG is not done
                __hip_atomic_store(&reinterpret_cast<HostMark*>(st->hostmark)->batch, batch_mark, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
    return stop;
}

}  // namespace fos
