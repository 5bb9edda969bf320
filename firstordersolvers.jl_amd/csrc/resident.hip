// RESIDENT CG: conjugategradient!(x, KKTMatrix(Q), rhs, ...) (conjugategradients.jl:31-55, called from affinepluslinear.jl:115-117) as ONE
// launch per solve, for operators that are nothing but dual tiles of a block-separable A and small enough to be held ON CHIP.
//
// Why.  On a shard of a block-diagonal SDP (the eighth of C4: 64 blocks, 34 MB of tiles) a CG iteration of the launch-per-iteration
// forms is two dependent launches of 12.8 + 8.7 us that move 34 + 19 MB -- launch, fill, drain and a scalar prologue, not bytes.
// 34 MB over 256 CUs is 133 KB per CU: it FITS THE REGISTER FILE (512 KB per CU).  So the solve becomes one persistent launch:
//   * a UNIT = the tiles over one run of <= 64 columns of A (one diagonal block of the SDP), dealt to `wpu` consecutive workgroups,
//     one tile per (wavefront, slot); lane = row keeps its row's matrix values, its elements of x, r, p, s, w and [c;b] in REGISTERS
//     from the first to the last iteration; the unit's column elements are replicated in every workgroup of the unit (LDS);
//   * one iteration = sweep out of registers (row sums against the LDS-resident column elements, column sums by the in-register
//     butterfly of the dual tiles) -> ONE exchange -> scalars -> vector update in registers.  Nothing is read from or written to
//     HBM between the start and the end of a solve;
//   * the exchange: every workgroup publishes its four partial sums {r.r, w.r, [c;b].r1, [c;b].r2} as self-validating words
//     ((sequence number << 32) | half a double: the mailboxes' trick) and adds the G records of the grid in workgroup order -- the
//     same bits in every workgroup, no grid barrier, no L2 write-back / invalidate (what made persistent kernels lose on this
//     8-XCD part in rounds 1-2); the workgroups of a unit exchange their <= 64 column sums the same way, in the same poll;
//   * sharded: the four sums then cross the GPUs through the handle's mailboxes, as in cgm_update_kernel (same words, same slots);
//   * WAVE SPECIALISATION: a workgroup = `ncomp` COMPUTE wavefronts (tiles in registers, sweep and row update, never a memory wait inside the
//     loop) + `ncomm` COMMUNICATION wavefronts (no tiles: they reduce the workgroup's sums, publish, poll, add the records, form the scalars and
//     keep the unit's COLUMN elements and the tau element in their own registers), two workgroup barriers per iteration between the roles.
//     The first form let every wavefront do everything: the tiles' 64 registers had to survive the exchange code, the allocator spilled
//     (reloads in front of every LDS access of the update), and wavefront 0 walked its 2 + 2 (wpu - 1) poll round trips one after the other
//     -- 4.8 of an iteration's 10.5 us (tools/res_stamps.py, profiles/r06_res_stamps_v1.txt).
// Arithmetic: the merged-reduction recurrence of cgm_update_kernel (FOS_CG_MERGED_UPDATE) -- same Krylov iterates, same iteration
// counting, same stop test `norm(r) <= tol || iter >= max_iters` on the recursively updated residual (conjugategradients.jl:42).
// Summation orders are fixed by the storage: bit-reproducible run to run.
#include "dev_common.hpp"

namespace fos {

struct ResArgs {
    d2* x;                     // in: start iterate, out: solution
    const d2* rhs;
    const d2* v;               // r_0 = rhs - M v
    const double* cb;
    int n, nm;
    DevState* st;
    const BlkDesc* blk;
    const double* val;
    const ResWG* wg;
    int G;
    unsigned long long* grec;
    unsigned long long* crec;
    int tmax;                  // stride of crec
    int ncomp;                 // compute wavefronts per workgroup (the others communicate)
    int flags;                 // experiments (FOS_RES_FLAGS): bit 0 = the first communication wavefront does not poll; bits 8.. = s_sleep between polls
    double tol;
    int maxit;
    PeerBox pb;
    uint32_t seq_base;
    long long timeout_ticks;
};

__device__ __forceinline__ void res_st_word(unsigned long long* p, unsigned long long w) { __hip_atomic_store(p, w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ unsigned long long res_ld_word(const unsigned long long* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
// a double as two self-validating words at p[0], p[1]
__device__ __forceinline__ void res_publish_half(unsigned long long* p, uint32_t seq, double v, int hh) {
    const unsigned long long bits = (unsigned long long)__double_as_longlong(v);
    res_st_word(p + hh, ((unsigned long long)seq << 32) | (hh ? (bits >> 32) : (bits & 0xFFFFFFFFull)));
}
__device__ __forceinline__ bool res_poll_f64(const unsigned long long* p, uint32_t seq, long long timeout, double& out) {
    const long long t0 = wall_clock64();
    unsigned long long lo, hi;
    bool ok;
    do {
        lo = res_ld_word(p); hi = res_ld_word(p + 1);
        ok = (uint32_t)(lo >> 32) == seq && (uint32_t)(hi >> 32) == seq;
    } while (!ok && (wall_clock64() - t0) < timeout);
    out = __longlong_as_double((long long)((hi << 32) | (lo & 0xFFFFFFFFull)));
    return ok;
}

// One tile out of registers: row sums (u1, u2) of the lane's row against the unit's column elements (LDS), and the tile's column sums -- 8 columns
// at a time through the butterfly of the dual tiles -- into the workgroup's LDS array.  The tile sits in the registers in the LANE'S OWN column order
// (position k of a group = column (lane & 7) ^ tile_sort(k): tile_colsum8_sorted, dev_common.hpp), so the butterfly needs no selects; the lane reads the
// column elements in the same order (xoff[k]: its eight byte offsets inside a group of 8 x 16 bytes).
template <int TMAX>
__device__ __forceinline__ void res_tile(const double (&val)[TMAX], int T, const d2 g, const char* __restrict__ gcol, const int (&xoff)[TILE_GROUP],
                                         d2* __restrict__ colpart, int lane, double& u1, double& u2) {
    u1 = 0.0; u2 = 0.0;
#pragma unroll
    for (int grp = 0; grp < TMAX / TILE_GROUP; ++grp) {
        if (grp * TILE_GROUP < T) {                                  // wave-uniform
            // (one right-hand side after the other: eight products live at a time, not sixteen)
            double p[TILE_GROUP];
#pragma unroll
            for (int h = 0; h < 2; ++h) {                            // (four column elements = 16 registers in flight at a time)
#pragma unroll
                for (int u = 4 * h; u < 4 * h + 4; ++u) {
                    const d2 xc = *reinterpret_cast<const d2*>(gcol + grp * TILE_GROUP * (int)sizeof(d2) + xoff[u]);
                    u1 += val[grp * TILE_GROUP + u] * xc.x; u2 += val[grp * TILE_GROUP + u] * xc.y;
                    p[u] = val[grp * TILE_GROUP + u] * g.x;
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            const double s1 = tile_colsum8_sorted(p);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int u = 0; u < TILE_GROUP; ++u) p[u] = val[grp * TILE_GROUP + u] * g.y;
            const double s2 = tile_colsum8_sorted(p);
            if (lane < TILE_GROUP) colpart[grp * TILE_GROUP + lane] = make_double2(s1, s2);
            __builtin_amdgcn_sched_barrier(0);                       // (keeps the next group's LDS reads from being hoisted up here: 4 registers per column element)
        }
    }
}

// In-kernel time stamps of the resident solve (timing experiments; -DFOS_RES_STAMPS, tools/res_stamps.py): workgroups 0, 1, G / 2, G - 1; the first lane of
// every wavefront wv; iteration it (0: the start), phase ph < 16 -> g_res_stamps[((slot * 16 + wv) * 64 + it) * 16 + ph], ticks of the shader clock.
#ifdef FOS_RES_STAMPS
__device__ long long g_res_stamps[4 * 16 * 64 * 16];
// (s_memtime -- the shader clock -- not the 100 MHz s_memrealtime of wall_clock64(): the latter takes long enough to return that two stamps in a row
//  showed microseconds; slot [63][6..7] of every (workgroup, who) holds one (s_memtime, s_memrealtime) pair taken at kernel entry and slot [62][6..7] one
//  at the end, from which the tool derives the clock rate)
#define RES_STAMP(ph) do { if (stamp_slot >= 0 && stamp_it < 62) { __builtin_amdgcn_sched_barrier(0); g_res_stamps[((size_t)stamp_slot * 64 + stamp_it) * 16 + (ph)] = (long long)__builtin_readcyclecounter(); __builtin_amdgcn_sched_barrier(0); } } while (0)
#define RES_STAMP_CAL(where) do { if (stamp_slot >= 0) { g_res_stamps[((size_t)stamp_slot * 64 + (where)) * 16 + 6] = (long long)__builtin_readcyclecounter(); g_res_stamps[((size_t)stamp_slot * 64 + (where)) * 16 + 7] = wall_clock64(); } } while (0)
#define RES_STAMP_NEXT() do { stamp_it += 1; } while (0)
#else
#define RES_STAMP(ph) do { } while (0)
#define RES_STAMP_CAL(where) do { } while (0)
#define RES_STAMP_NEXT() do { } while (0)
#endif

// what the communication wavefronts hand to the compute wavefronts through LDS at the end of an exchange
enum ResCtl { RC_ALPHA = 0, RC_BETA, RC_GTX, RC_GTY, RC_STOP, RC_NEAR, RC_COUNT };     // RC_STOP: 0 go on, 1 CG has stopped, 2 an exchange failed

// The exchange of the four sums across the GPUs by ONE wavefront (the words, slots and sequence numbers of peer_fold_sum, dev_common.hpp --
// the launch-per-iteration kernels and this one speak the same protocol): lane = (rank r, value v, half hh), two rounds cover 16 ranks.
// In: tot[0..4) the local sums (the same in every lane); out: the sums over the ranks in rank order.  `first` = workgroup 0 (the only writer).
__device__ __forceinline__ bool res_peer_fold4(const PeerBox& pb, uint32_t seq, double (&tot)[4], bool first, uint32_t* halves /* LDS [PEER_MAX_RANKS * 8] */) {
    const int lane = threadIdx.x & 63;
    const size_t par = (size_t)((((seq >> 11) & 1u) << 1) | (seq & 1u)) * PEER_MAX_RANKS;
    bool bad = false;
    for (int t = lane; t < pb.nranks * 8; t += 64) {
        const int hh = t & 1, v = (t >> 1) & 3, r = t >> 3;
        const double mine_d = v == 0 ? tot[0] : (v == 1 ? tot[1] : (v == 2 ? tot[2] : tot[3]));
        const unsigned long long bits = (unsigned long long)__double_as_longlong(mine_d);
        const uint32_t mine = hh ? (uint32_t)(bits >> 32) : (uint32_t)bits;
        if (r == pb.rank && !pb.loopback) { halves[t] = mine; continue; }
        const size_t off_in = PEER_BOX_WORDS + ((par + r) * PEER_MAX_VALS + v) * 2 + hh;          // where rank r's word arrives
        if (first) {
            const bool wr = !pb.shared || pb.loopback ? (r != pb.rank || pb.loopback) : (r == (pb.rank == 0 ? 1 : 0));
            if (wr) {
                unsigned long long* dst = pb.box[r] + PEER_BOX_WORDS + ((par + pb.rank) * PEER_MAX_VALS + v) * 2 + hh;
                __hip_atomic_store(dst, ((unsigned long long)seq << 32) | mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            }
        }
        const bool via_relay = pb.relay != nullptr && !first;      // host-pinned transport: only workgroup 0 reads the segment, the others its republication
        const unsigned long long* src = via_relay ? pb.relay + off_in : pb.box[pb.rank] + off_in;
        long long t0 = 0;
        unsigned long long w;
        bool ok;
        for (uint32_t spin = 1;; ++spin) {
            w = __hip_atomic_load(src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            ok = (uint32_t)(w >> 32) == seq;
            if (ok) break;
            if ((spin & 255u) == 0u) {
                const long long now = wall_clock64();
                if (t0 == 0) t0 = now;
                else if (now - t0 >= pb.timeout_ticks) break;
            }
        }
        if (!ok) bad = true;
        else if (pb.relay != nullptr && first) __hip_atomic_store(pb.relay + off_in, w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        halves[t] = (uint32_t)w;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    if (__builtin_amdgcn_ballot_w64(bad) != 0ull) return false;
#pragma unroll
    for (int v = 0; v < 4; ++v) {
        double sacc = 0.0;
        for (int r = 0; r < pb.nranks; ++r) {                      // rank order: every rank computes the same bits
            const unsigned long long lo = halves[(r * 4 + v) * 2], hi = halves[(r * 4 + v) * 2 + 1];
            sacc += __longlong_as_double((long long)((hi << 32) | lo));
        }
        tot[v] = sacc;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    return true;
}

template <int TMAX, int RPT, int LB>
__global__ __launch_bounds__(LB) void cg_resident_kernel(ResArgs a) {
    __shared__ __attribute__((aligned(16))) d2 s_gcol[TMAX];                 // the unit's column elements of the vector being swept (v, then r)
    __shared__ double s_red[16][4];                                           // the compute wavefronts' sums
    __shared__ double s_ctl[RC_COUNT];
    __shared__ uint32_t s_halves[PEER_MAX_RANKS * 8];
    __shared__ int s_cnt, s_failed;
    // dynamic: per tile slot the rows' elements of x, p, s, w (lane-private: only the residual, which the sweep multiplies by, lives in
    // registers beside the matrix values), the slots' column sums, the other workgroups' column sums, the G records of an exchange.
    // (NSLOT is a compile-time constant so that every one of these addresses is (thread's 16-byte offset) + an immediate: held as
    //  runtime values they were spilled to scratch and reloaded one by one in front of every LDS access of the update)
    extern __shared__ __attribute__((aligned(16))) double s_dyn[];
    constexpr int NSLOT = (LB / 64) * RPT;
    d2* const s_x = reinterpret_cast<d2*>(s_dyn);                             // [NSLOT][64]
    d2* const s_p = s_x + NSLOT * 64;
    d2* const s_s = s_p + NSLOT * 64;
    d2* const s_w = s_s + NSLOT * 64;
    d2* const s_colpart = s_w + NSLOT * 64;                                   // [NSLOT][TMAX]
    double* const s_sib = reinterpret_cast<double*>(s_colpart + NSLOT * TMAX);      // [RES_WPU_MAX - 1][TMAX][2]
    double* const s_all = s_sib + (RES_WPU_MAX - 1) * TMAX * 2;                     // [4][RES_GMAX], zero beyond G (read without predicates: a
                                                                                    // predicated LDS read is a branch and a wait of its own)
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6), nw = (int)(blockDim.x >> 6);
    const int ncomp = a.ncomp, ncomm = nw - ncomp;
    const ResWG me = a.wg[blockIdx.x];
    const int T = me.T, tc = me.tc, c0 = me.c0, nm = a.nm;
    DevState* st = a.st;
#ifdef FOS_RES_STAMPS
    int stamp_slot = -1, stamp_it = 0;
    {
        const int b = (int)blockIdx.x;
        const int wslot = b == 0 ? 0 : (b == 1 ? 1 : (b == a.G / 2 ? 2 : (b == a.G - 1 ? 3 : -1)));
        if (wslot >= 0 && lane == 0 && wv < 16) stamp_slot = wslot * 16 + wv;          // (the first lane of every wavefront)
    }
#endif
    if (a.pb.nranks > 0 && st->xchg_failed) return;
    if (tid == 0) { s_cnt = 0; s_failed = 0; }
    RES_STAMP_CAL(63);

    if (wv < ncomp) {
        // =========================================================== COMPUTE wavefronts: tiles in registers, sweep, row update
        double val[RPT][TMAX];
        d2 rr[RPT], rhsr[RPT];         // rr: the rows' elements of the vector being swept -- v at the start, then the residual; rhsr: dead after the start
        double cbr[RPT];
        int row[RPT];
        bool has[RPT], valid[RPT];
#pragma unroll
        for (int q = 0; q < RPT; ++q) {
            const int ti = wv + q * ncomp;
            has[q] = ti < me.nblk;
            valid[q] = false; row[q] = 0; cbr[q] = 0.0;
            rr[q] = rhsr[q] = make_double2(0.0, 0.0);
#pragma unroll
            for (int t = 0; t < TMAX; ++t) val[q][t] = 0.0;
            if (has[q]) {
                const BlkDesc d = a.blk[me.blk0 + ti];
                valid[q] = lane < d.nrows();
                row[q] = d.row0 + lane;
                const double* __restrict__ vp = a.val + d.nnz0 + lane;
                // (all loads unconditional -- one request phase; steps beyond T and lanes beyond the tile's rows become zeros; position k of a group
                //  takes the lane's column (lane & 7) ^ tile_sort(k): res_tile)
#pragma unroll
                for (int t = 0; t < TMAX; ++t) {
                    const int tl = (t & ~7) + ((lane & 7) ^ tile_sort(t & 7));
                    const double v = vp[64 * ((t & ~7) < T ? tl : 0)];
                    val[q][t] = ((t & ~7) < T && valid[q]) ? v : 0.0;
                }
                d2 x0 = make_double2(0.0, 0.0);
                if (valid[q]) { x0 = a.x[row[q]]; rr[q] = a.v[row[q]]; rhsr[q] = a.rhs[row[q]]; cbr[q] = a.cb[row[q]]; }
                s_x[(size_t)ti * 64 + lane] = x0;
            }
        }
        d2 gt = a.v[nm];                                   // tau element of the vector being swept
        int xoff[TILE_GROUP];                              // the lane's byte offsets of its eight columns inside a group of column elements
#pragma unroll
        for (int k = 0; k < TILE_GROUP; ++k) xoff[k] = ((lane & 7) ^ tile_sort(k)) * (int)sizeof(d2);
        const char* const gcolb = reinterpret_cast<const char*>(s_gcol);
        __syncthreads();                                   // (0) the communication wavefronts have staged the v columns
        RES_STAMP(0);
        for (int i = -1;; ++i) {                           // i = -1: the start, r_0 = rhs - M v; i >= 0: iteration i + 1
            // ---- sweep: w = M g on the wavefront's rows, its tiles' column sums, its lanes' share of the sums
            double acc[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int q = 0; q < RPT; ++q) {
                if (has[q]) {
                    const size_t e = (size_t)(wv + q * ncomp) * 64 + lane;
                    const d2 gq = rr[q];
                    double u1, u2;
                    res_tile<TMAX>(val[q], T, gq, gcolb, xoff, s_colpart + (size_t)(wv + q * ncomp) * TMAX, lane, u1, u2);
                    const double c = cbr[q];
                    const double q1 = -(u1 - gt.x * c), q2 = -(u2 - gt.y * c);           // rows of A: EpiKkt::row, i >= n   (HSDEAffine.jl:52,55)
                    d2 w = make_double2(gq.x - q2, q1 - gq.y);                           // affinepluslinear.jl:45-48
                    if (!valid[q]) w = make_double2(0.0, 0.0);
                    s_w[e] = w;
                    acc[0] += gq.x * gq.x + gq.y * gq.y;                                 // r.r of what is swept (unused at the start)
                    acc[1] += w.x * gq.x + w.y * gq.y;
                    acc[2] += c * gq.x;
                    acc[3] += c * gq.y;
                }
            }
            if (has[0]) {
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const double v = wave_sum(acc[k]);
                    if (lane == 0) s_red[wv][k] = v;
                }
            } else if (lane < 4) s_red[wv][lane] = 0.0;     // (a compute wavefront without a tile: nothing to add)
            RES_STAMP(1);                                   // swept
            __syncthreads();                               // (A) column sums and the wavefronts' sums are in LDS
            RES_STAMP(2);
            __syncthreads();                               // (B) the communication wavefronts have left alpha, beta, the new columns and tau element
            RES_STAMP(3);
            const double stopf = s_ctl[RC_STOP];
            if (stopf != 0.0) break;
            const double alpha = s_ctl[RC_ALPHA], beta = s_ctl[RC_BETA];
            gt = make_double2(s_ctl[RC_GTX], s_ctl[RC_GTY]);
#pragma unroll
            for (int q = 0; q < RPT; ++q) {
                if (has[q]) {
                    const size_t e = (size_t)(wv + q * ncomp) * 64 + lane;
                    const d2 w = s_w[e];
                    if (i < 0) {
                        rr[q] = make_double2(rhsr[q].x - w.x, rhsr[q].y - w.y);          // r_0 = rhs - M v      conjugategradients.jl:33
                    } else {
                        d2 pq, sq, xq = s_x[e];
                        if (i == 0) { pq = rr[q]; sq = w; }
                        else {
                            pq = s_p[e]; sq = s_s[e];
                            pq.x = pq.x * beta + rr[q].x; pq.y = pq.y * beta + rr[q].y;  // p .*= beta ; p .+= r     :49-50
                            sq.x = sq.x * beta + w.x; sq.y = sq.y * beta + w.y;          // s = M p by the same recurrence
                        }
                        xq.x += alpha * pq.x; xq.y += alpha * pq.y;                      // :40
                        rr[q].x -= alpha * sq.x; rr[q].y -= alpha * sq.y;                // :41
                        s_p[e] = pq; s_s[e] = sq; s_x[e] = xq;
                    }
                }
            }
            RES_STAMP(4);                                   // updated
            RES_STAMP_NEXT();
            RES_STAMP(0);
        }
        // ---- the solution leaves the workgroup
#pragma unroll
        for (int q = 0; q < RPT; ++q)
            if (valid[q]) a.x[row[q]] = s_x[(size_t)(wv + q * ncomp) * 64 + lane];
        RES_STAMP_CAL(62);
        return;
    }

    // =============================================================== COMMUNICATION wavefronts
    const int cw = wv - ncomp, ct = tid - 64 * ncomp;             // wavefront / thread number among them
    const bool c0wave = cw == 0, leader = me.idx == 0;
    // the unit's columns (replicated in every workgroup of the unit) and the tau element (replicated everywhere) live in the first of them
    d2 cx = make_double2(0.0, 0.0), cr = cx, cpv = cx, csv = cx, crhs = cx;
    double cc = 0.0;
    d2 gt = a.v[nm], xt = a.x[nm], pt = make_double2(0.0, 0.0), stt = pt;
    const d2 rhst = a.rhs[nm];
    for (int q = ct; q < 4 * RES_GMAX; q += 64 * ncomm) s_all[q] = 0.0;
    for (int q = ct + me.nblk * TMAX; q < NSLOT * TMAX; q += 64 * ncomm) s_colpart[q] = make_double2(0.0, 0.0);
    for (int q = ct + ncomp * 4; q < 16 * 4; q += 64 * ncomm) (&s_red[0][0])[q] = 0.0;
    if (c0wave) {
        if (lane < tc) { cr = a.v[c0 + lane]; cx = a.x[c0 + lane]; crhs = a.rhs[c0 + lane]; cc = a.cb[c0 + lane]; }
        if (lane < TMAX) s_gcol[lane] = cr;
        if (blockIdx.x == 0 && lane == 0) { st->tol = a.tol; st->maxit = a.maxit; st->hit_max = 0; st->rn_old = 0.0; }
    }
    __syncthreads();                                           // (0)
    double colG = 0.0;                                         // the columns' share of r.r (the unit's first workgroup counts them)
    double g_prev = 0.0, a_prev = 0.0, gam = 0.0;
    int iter = 0;
    uint32_t nx = 0;                                           // exchanges so far
    RES_STAMP(0);
    for (int i = -1;; ++i) {
        const uint32_t seq = a.seq_base + (uint32_t)(i + 1);
        const size_t par = (size_t)(seq & 1u);
        unsigned long long* grec = a.grec + par * (size_t)a.G * 8;
        nx += 1;
        __syncthreads();                                       // (A)
        RES_STAMP(1);
        d2 cp = make_double2(0.0, 0.0);
        if (c0wave) {
            // ---- this workgroup's record: the compute wavefronts' sums + the columns' share (the bilinear form of EpiKkt::park / deferred_local)
            double acc[4] = {0.0, 0.0, 0.0, 0.0};
            if (lane < tc) {
                // (all slots requested together -- slots beyond the workgroup's tiles read as zero -- then added in slot order: one LDS latency, not nblk)
                d2 o[NSLOT];
#pragma unroll
                for (int s = 0; s < NSLOT; ++s) o[s] = s_colpart[(size_t)s * TMAX + lane];                 // (slots beyond nblk were zeroed once)
#pragma unroll
                for (int s = 0; s < NSLOT; ++s) { cp.x += o[s].x; cp.y += o[s].y; }
                RES_STAMP(8);                               // (finer: the slots' column sums added)
                acc[1] = cp.x * cr.y - cp.y * cr.x;                                      // this workgroup's share of (w.g) of column `lane` (i < n)
                if (leader) {                                                            // the slot-free part and the other sums: once per column
                    acc[0] = colG;
                    acc[1] += (cr.x * cr.x - cr.y * cr.y) + cc * (gt.x * cr.y - gt.y * cr.x);
                    acc[2] = cc * cr.x;
                    acc[3] = cc * cr.y;
                }
            }
            double mine[4];
            {
                // the compute wavefronts' sums: lane (w, k) reads one, lanes k + 4 w are added over w in order by a strided walk of the wavefront
                // (all reads in flight together; the same order of additions as a loop over w)
                double rv[12];
#pragma unroll
                for (int w = 0; w < 12; ++w) rv[w] = 0.0;
                const int k4 = lane & 3;
#pragma unroll
                for (int w = 0; w < 12; ++w) rv[w] = s_red[w][k4];                      // (rows beyond ncomp were zeroed once)
                double sacc = 0.0;
#pragma unroll
                for (int w = 0; w < 12; ++w) sacc += rv[w];                             // (zeros beyond ncomp)
                RES_STAMP(9);                               // (finer: the compute wavefronts' sums added)
                // lane l holds the sum of value l & 3: hand each k to every lane
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int lo = __builtin_amdgcn_readlane(__double2loint(sacc), k), hi = __builtin_amdgcn_readlane(__double2hiint(sacc), k);
                    mine[k] = __hiloint2double(hi, lo) + wave_sum(acc[k]);
                }
            }
            RES_STAMP(10);                                  // (finer: the workgroup's four sums)
            if (i < 0) mine[0] = 0.0;                                                    // (no r.r at the start)
            if (lane < 8) {
                const int k = lane >> 1;
                res_publish_half(grec + (size_t)blockIdx.x * 8 + 2 * k, seq, k == 0 ? mine[0] : (k == 1 ? mine[1] : (k == 2 ? mine[2] : mine[3])), lane & 1);
            }
            if (lane < 4) s_all[(size_t)lane * RES_GMAX + blockIdx.x] = lane == 0 ? mine[0] : (lane == 1 ? mine[1] : (lane == 2 ? mine[2] : mine[3]));
            if (me.wpu > 1 && lane < tc) {
                unsigned long long* p = a.crec + ((par * a.G + blockIdx.x) * (size_t)a.tmax + (size_t)lane) * 4;
                res_publish_half(p, seq, cp.x, 0); res_publish_half(p, seq, cp.x, 1);
                res_publish_half(p + 2, seq, cp.y, 0); res_publish_half(p + 2, seq, cp.y, 1);
            }
        }
        RES_STAMP(2);                                           // published
        // ---- everything this workgroup waits for, as ONE list dealt to the communication threads: the other workgroups' 4 record values each,
        // then the column sums of the unit's other workgroups.  A thread requests the words of PU items together and polls them together.
        {
            const bool c0polls = !(a.flags & 1) || ncomm == 1;
            const int nrec = 4 * a.G, nsib = (me.wpu - 1) * tc * 2, nthr = c0polls ? 64 * ncomm : 64 * (ncomm - 1);
            const int ctp = c0polls ? ct : ct - 64;          // this thread's number among the polling threads (< 0: does not poll)
            const int sleepn = (a.flags >> 8) & 0x7F;
            auto item = [&](int idx, double*& dst) -> const unsigned long long* {
                if (idx < nrec) {
                    const int wg = idx >> 2, k = idx & 3;
                    dst = s_all + (size_t)k * RES_GMAX + wg;
                    return wg == (int)blockIdx.x ? nullptr : grec + (size_t)wg * 8 + 2 * k;
                }
                const int j = idx - nrec, comp = j & 1, c = (j >> 1) % tc, si = (j >> 1) / tc;
                const int k = si < me.idx ? si : si + 1;                 // the unit's other workgroups, in order
                dst = s_sib + ((size_t)si * TMAX + c) * 2 + comp;
                return a.crec + ((par * a.G + (size_t)(me.wg0 + k)) * (size_t)a.tmax + (size_t)c) * 4 + 2 * comp;
            };
            constexpr int PU = 7;
            bool bad = false;
            for (int base = ctp >= 0 ? ctp : nrec + nsib; base < nrec + nsib; base += PU * nthr) {
                const unsigned long long* src[PU];
                double* dst[PU];
#pragma unroll
                for (int u = 0; u < PU; ++u) {
                    const int idx = base + u * nthr;
                    src[u] = nullptr; dst[u] = nullptr;
                    if (idx < nrec + nsib) src[u] = item(idx, dst[u]);
                }
                // (no clock read on the fast path: s_memrealtime takes about a microsecond to return -- the clock is looked at every 256th unsuccessful round)
                long long t0 = 0;
                for (uint32_t spin = 1;; ++spin) {
                    unsigned long long lo[PU], hi[PU];
#pragma unroll
                    for (int u = 0; u < PU; ++u) if (src[u]) { lo[u] = res_ld_word(src[u]); hi[u] = res_ld_word(src[u] + 1); }
                    bool pending = false;
#pragma unroll
                    for (int u = 0; u < PU; ++u) {
                        if (src[u]) {
                            if ((uint32_t)(lo[u] >> 32) == seq && (uint32_t)(hi[u] >> 32) == seq) {
                                *dst[u] = __longlong_as_double((long long)((hi[u] << 32) | (lo[u] & 0xFFFFFFFFull)));
                                src[u] = nullptr;
                            } else pending = true;
                        }
                    }
                    if (!pending) break;
                    if ((spin & 255u) == 0u) {
                        const long long now = wall_clock64();
                        if (t0 == 0) t0 = now;
                        else if (now - t0 >= a.timeout_ticks) { bad = true; break; }
                    }
                    // (back-off: 256 workgroups x 192 polling threads otherwise keep the fabric busy with reads of words that are not there yet)
                    if (sleepn >= 8) __builtin_amdgcn_s_sleep(8); else if (sleepn >= 4) __builtin_amdgcn_s_sleep(4); else if (sleepn >= 2) __builtin_amdgcn_s_sleep(2); else if (sleepn >= 1) __builtin_amdgcn_s_sleep(1);
                }
            }
            if (bad) s_failed = 1;
            // the communication wavefronts meet in LDS (the workgroup barrier belongs to both roles): a counter of arrivals
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            if (lane == 0) __hip_atomic_fetch_add(&s_cnt, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
        RES_STAMP(3);                                           // this wavefront's words have arrived
        if (c0wave) {
            {
                long long t0 = 0;
                for (uint32_t spin = 1; __hip_atomic_load(&s_cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < (int)(nx * (uint32_t)ncomm); ++spin) {
                    if ((spin & 4095u) == 0u) {
                        const long long now = wall_clock64();
                        if (t0 == 0) t0 = now;
                        else if (now - t0 >= 2 * a.timeout_ticks) { s_failed = 1; break; }
                    }
                }
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
            }
            RES_STAMP(4);                                       // ... everybody's
            bool failed = __hip_atomic_load(&s_failed, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) != 0;
            // the grid's totals: the G values of each sum lane-strided, then the butterfly -- the same bits in every workgroup
            double tot[4];
#pragma unroll
            for (int kk = 0; kk < 4; kk += 2) {                   // (two sums at a time: 16 reads in flight, zeros beyond G)
                double rv[2][RES_GMAX / 64];
#pragma unroll
                for (int k = 0; k < 2; ++k)
#pragma unroll
                    for (int j = 0; j < RES_GMAX / 64; ++j) rv[k][j] = s_all[(size_t)(kk + k) * RES_GMAX + lane + 64 * j];
#pragma unroll
                for (int k = 0; k < 2; ++k) {
                    double sacc = 0.0;
#pragma unroll
                    for (int j = 0; j < RES_GMAX / 64; ++j) sacc += rv[k][j];
                    tot[kk + k] = wave_sum(sacc);
                }
            }
            if (!failed && a.pb.nranks > 0) failed = !res_peer_fold4(a.pb, seq, tot, blockIdx.x == 0, s_halves);
            RES_STAMP(5);                                       // totals (over the ranks)
            RES_STAMP(11);                                  // (finer)
            // the unit's column sums: its workgroups in order -- the same bits in each of them
            d2 ctot = cp;
            if (me.wpu > 1 && lane < tc) {
                // (the first four of the other workgroups' sums read together, unpredicated -- clamped index, the value dropped by a select)
                const d2* sib = reinterpret_cast<const d2*>(s_sib);
                d2 sv[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) sv[q] = sib[(size_t)(q < me.wpu - 1 ? q : 0) * TMAX + lane];
                d2 t = make_double2(0.0, 0.0);
#pragma unroll
                for (int k = 0; k < 4; ++k) {                     // workgroups 0..3 of the unit, in order (slot si of s_sib: the unit's workgroups without this one)
                    const int si = k < me.idx ? k : k - 1;
                    d2 part = k == me.idx ? cp : (si == 0 ? sv[0] : (si == 1 ? sv[1] : (si == 2 ? sv[2] : sv[3])));
                    if (k >= me.wpu) part = make_double2(0.0, 0.0);
                    t.x += part.x; t.y += part.y;
                }
                for (int k = 4; k < me.wpu; ++k) {
                    d2 part = cp;
                    if (k != me.idx) { const int si = k < me.idx ? k : k - 1; part = sib[(size_t)si * TMAX + lane]; }
                    t.x += part.x; t.y += part.y;
                }
                ctot = t;
            }
            // column `lane` of w = M g (EpiKkt::row, i < n: HSDEAffine.jl:51,54) and the tau row (HSDEAffine.jl:57)
            const double q1 = ctot.x + gt.x * cc, q2 = ctot.y + gt.y * cc;
            const d2 cwv = make_double2(cr.x - q2, q1 - cr.y);
            const d2 wt = make_double2(gt.x + tot[3], -tot[2] - gt.y);
            RES_STAMP(12);                                  // (finer: column and tau rows of w)
            double stopf = failed ? 2.0 : 0.0, alpha = 0.0, beta = 0.0;
            if (!failed) {
                if (i < 0) {
                    // r_0 = rhs - M v                               conjugategradients.jl:32-36
                    cr = make_double2(crhs.x - cwv.x, crhs.y - cwv.y);
                    gt = make_double2(rhst.x - wt.x, rhst.y - wt.y);
                } else {
                    gam = tot[0] + (gt.x * gt.x + gt.y * gt.y);
                    if (i > 0 && (sqrt(gam) <= a.tol || i >= a.maxit)) { iter = i; stopf = 1.0; }       // conjugategradients.jl:42 for iteration i
                    else {
                        const double delta = tot[1] + (wt.x * gt.x + wt.y * gt.y);
                        if (i == 0) alpha = gam / delta;
                        else {
                            beta = gam / g_prev;
                            alpha = gam / (delta - beta * gam / a_prev);
                        }
                        g_prev = gam; a_prev = alpha;
                        // (w, r, p, s, x) -> (p, s, x, r)              :39-41,49-50 with Ap replaced by the recurrence s = M p
                        auto upd = [&](const d2 wi, d2& ri, d2& pi, d2& si, d2& xi) {
                            if (i == 0) { pi = ri; si = wi; }
                            else {
                                pi.x = pi.x * beta + ri.x; pi.y = pi.y * beta + ri.y;
                                si.x = si.x * beta + wi.x; si.y = si.y * beta + wi.y;
                            }
                            xi.x += alpha * pi.x; xi.y += alpha * pi.y;
                            ri.x -= alpha * si.x; ri.y -= alpha * si.y;
                        };
                        if (lane < tc) upd(cwv, cr, cpv, csv, cx);
                        upd(wt, gt, pt, stt, xt);
                    }
                }
            }
            RES_STAMP(13);                                  // (finer: scalars, column and tau update)
            if (stopf == 0.0) {
                colG = (leader && lane < tc) ? cr.x * cr.x + cr.y * cr.y : 0.0;
                if (lane < tc) s_gcol[lane] = cr;
            }
            if (lane == 0) { s_ctl[RC_ALPHA] = alpha; s_ctl[RC_BETA] = beta; s_ctl[RC_GTX] = gt.x; s_ctl[RC_GTY] = gt.y; s_ctl[RC_STOP] = stopf; }
        }
        RES_STAMP(6);                                           // scalars, columns, tau
        __syncthreads();                                       // (B)
        RES_STAMP(7);
        RES_STAMP_NEXT();
        RES_STAMP(0);
        if (s_ctl[RC_STOP] != 0.0) break;
    }
    RES_STAMP_CAL(62);
    if (c0wave) {
        const bool ok = s_ctl[RC_STOP] == 1.0;
        if (ok && leader && lane < tc) a.x[c0 + lane] = cx;
        if (blockIdx.x == 0 && lane == 0) {
            if (ok) {
                a.x[nm] = xt;
                st->rr = gam;
                cg_signal_stop(st, iter, a.maxit, gam, a.seq_base >> 11);
            } else {
                if (a.pb.nranks > 0) st->xchg_failed = 1;          // (sharded: the likeliest cause is a peer that never arrived -- every later exchange is skipped)
                st->bar_failed = 1; st->done = 1;
            }
        }
    }
}

// =====================================================================================================================================
// STREAMED form of the resident solve (round 6): for block-separable operators whose tiles do NOT fit the register file (the whole of C4: 273 MB
// of tiles against 128 MB of registers; its shards on two and four GPUs).  What stays on chip is everything BUT the matrix: a workgroup owns whole
// units (consecutive ones, <= 64 columns together), its compute wavefronts walk their tiles once per iteration -- a tile's values are read from
// HBM / L2 (plain coalesced 512-byte steps), multiplied, and dropped -- while the rows' r and w live in registers, p and s in LDS, x in global
// memory (read and written once per iteration), the columns and the tau element in the communication wavefronts' registers.  Per CG iteration the
// chip then reads the matrix once + 40 B per row, where sweep + update kernels read and write the matrix + 160 B per row and pay two launches.
// Same exchange (records of self-validating words, no grid barrier), same recurrence, same summation order rules as the register form above;
// a unit lives in ONE workgroup, so column sums never cross workgroups.
constexpr int RS_GMAX = 256;          // workgroups (= records) of the streamed form: one per CU
constexpr int RS_WPU_MAX = 4;         // workgroups a unit may be split over in the streamed form (fewer units than CUs: a shard of a four-GPU run)
constexpr int RS_NCOMP = 7;           // compute wavefronts per workgroup (+ 1 that communicates): two wavefronts per SIMD, 256 registers each

// a tile in NATURAL column order (as stored): row sums against the workgroup's column elements at gcol (LDS, uniform addresses: broadcasts),
// column sums ADDED to the wavefront's own array (lanes 0..7, program order inside the wavefront)
template <int TMAX>
__device__ __forceinline__ void res_tile_plain(const double (&val)[TMAX], int T, const d2 g, const d2* __restrict__ gcol, d2* __restrict__ colacc, int lane,
                                               double& u1, double& u2) {
    u1 = 0.0; u2 = 0.0;
#pragma unroll
    for (int grp = 0; grp < TMAX / TILE_GROUP; ++grp) {
        if (grp * TILE_GROUP < T) {                                  // wave-uniform
            double p[TILE_GROUP];
#pragma unroll
            for (int h = 0; h < 2; ++h) {                            // (four column elements = 16 registers in flight at a time)
#pragma unroll
                for (int u = 4 * h; u < 4 * h + 4; ++u) {
                    const d2 xc = gcol[grp * TILE_GROUP + u];
                    u1 += val[grp * TILE_GROUP + u] * xc.x; u2 += val[grp * TILE_GROUP + u] * xc.y;
                    p[u] = val[grp * TILE_GROUP + u] * g.x;
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            const double s1 = tile_colsum8(p, lane);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int u = 0; u < TILE_GROUP; ++u) p[u] = val[grp * TILE_GROUP + u] * g.y;
            const double s2 = tile_colsum8(p, lane);
            if (lane < TILE_GROUP) {
                d2 o = colacc[grp * TILE_GROUP + lane];
                o.x += s1; o.y += s2;
                colacc[grp * TILE_GROUP + lane] = o;
            }
            __builtin_amdgcn_sched_barrier(0);                       // (keeps the next group's LDS reads from being hoisted up here)
        }
    }
}

constexpr int RS_NTC = 3;             // tiles the communication wavefront sweeps itself (it waits at barrier (A) otherwise): 66 tiles = 7 x 9 + 3

// which tiles of a workgroup wavefront w walks (w < RS_NCOMP: compute; w == RS_NCOMP: the communication wavefront, the LAST tiles)
__host__ __device__ inline void rs_split(int nblk, int w, int& t0, int& cnt) {
    const int per = nblk / RS_NCOMP, r = nblk % RS_NCOMP, kc = r < RS_NTC ? r : RS_NTC, rem = r - kc;
    if (w < RS_NCOMP) { cnt = per + (w < rem ? 1 : 0); t0 = w * per + (w < rem ? w : rem); }
    else { cnt = kc; t0 = nblk - kc; }
}

// the rows of a wavefront's tiles that never leave the registers: residual, w = M r of the last sweep, the iterate
template <int NT> struct RsRows { d2 rr[NT], ww[NT], xx[NT]; };

template <int NT>
__device__ __forceinline__ void rs_rows_load(const ResArgs& a, int blk_first, int cnt, int lane, RsRows<NT>& R) {
#pragma unroll
    for (int q = 0; q < NT; ++q) {
        R.rr[q] = R.ww[q] = R.xx[q] = make_double2(0.0, 0.0);
        if (q < cnt) {
            const BlkDesc d = a.blk[blk_first + q];
            if (lane < d.nrows()) { R.rr[q] = a.v[d.row0 + lane]; R.xx[q] = a.x[d.row0 + lane]; }
        }
    }
}
template <int NT>
__device__ __forceinline__ void rs_rows_store(const ResArgs& a, int blk_first, int cnt, int lane, const RsRows<NT>& R) {
#pragma unroll
    for (int q = 0; q < NT; ++q) {
        if (q < cnt) {
            const BlkDesc d = a.blk[blk_first + q];
            if (lane < d.nrows()) a.x[d.row0 + lane] = R.xx[q];
        }
    }
}

// one sweep of a wavefront's tiles: w = M g on their rows (g = R.rr), their column sums into the wavefront's array, the four sums of the rows
// (the tiles' descriptors are read again in every sweep, by scalar loads: kept in registers across the solve, they and everything derived from
//  them -- addresses, masks, offsets of every tile -- were hoisted out of the loop and spilled by the hundred; a branch per tile for the same reason)
template <int TMAX, int NT>
__device__ __forceinline__ void rs_sweep(const ResArgs& a, int blk_first, int cnt, int c0, int lane, const d2* __restrict__ s_gcol, const double* s_ctl,
                                         d2* __restrict__ mycol, RsRows<NT>& R, double (&acc)[4], bool early = false) {
    // (early: an exchange round that carries r.r alone -- no tile is walked, the other three sums are zero)
    if (early) cnt = 0;
    mycol[lane] = make_double2(0.0, 0.0);
    acc[0] = acc[1] = acc[2] = acc[3] = 0.0;
#pragma unroll
    for (int q = 0; q < NT; ++q) {
        if (q < cnt) {                             // wave-uniform
            const BlkDesc d = a.blk[blk_first + q];
            const int T = d.steps(), coff = d.meta[0] - c0;
            const bool valid = lane < d.nrows();
            const double* __restrict__ vp = a.val + d.nnz0 + lane;
            // (requesting a wavefront's first tile BEFORE the exchange in front of it was measured: 70.5 against 66.3 us per iteration --
            //  the bulk loads delay the exchange's words)
            const d2 gq = R.rr[q];
            double u1 = 0.0, u2 = 0.0;
#pragma unroll
            for (int hf = 0; hf < TMAX / 32; ++hf) {       // (32 steps = 64 registers of matrix values at a time; 64-step tiles: two such passes)
                double val[32];
#pragma unroll
                for (int t = 0; t < 32; ++t) val[t] = (((32 * hf + t) & ~7) < T) ? __builtin_nontemporal_load(vp + 64 * (32 * hf + t)) : 0.0;   // (zero-padded storage beyond the tile's rows)
                double h1, h2;
                res_tile_plain<32>(val, T - 32 * hf, gq, s_gcol + coff + 32 * hf, mycol + coff + 32 * hf, lane, h1, h2);
                u1 += h1; u2 += h2;
            }
            const double c = valid ? a.cb[d.row0 + lane] : 0.0;
            const d2 gt = make_double2(s_ctl[RC_GTX], s_ctl[RC_GTY]);            // (read here, not held across the tile: registers)
            const double q1 = -(u1 - gt.x * c), q2 = -(u2 - gt.y * c);           // rows of A: EpiKkt::row, i >= n   (HSDEAffine.jl:52,55)
            d2 w = make_double2(gq.x - q2, q1 - gq.y);                           // affinepluslinear.jl:45-48
            if (!valid) w = make_double2(0.0, 0.0);
            R.ww[q] = w;
            acc[2] += c * gq.x;
            acc[3] += c * gq.y;
        }
    }
#pragma unroll
    for (int q = 0; q < NT; ++q) {                 // (r.r and w.r from the registers, in tile order: slots beyond cnt hold zeros)
        acc[0] += R.rr[q].x * R.rr[q].x + R.rr[q].y * R.rr[q].y;
        acc[1] += R.ww[q].x * R.rr[q].x + R.ww[q].y * R.rr[q].y;
    }
    if (early) acc[1] = 0.0;
}

// the rows' share of an iteration's update: i < 0: r_0 = rhs - M v; else (w, r, p, s, x) -> (p, s, x, r)   conjugategradients.jl:39-41,49-50
template <int NT>
__device__ __forceinline__ void rs_update(const ResArgs& a, int blk_first, int slot_first, int cnt, int lane, int i, double alpha, double beta,
                                          d2* __restrict__ s_ps, RsRows<NT>& R) {
#pragma unroll
    for (int q = 0; q < NT; ++q) {
        if (q < cnt) {
            const size_t e = (size_t)(slot_first + q) * 128 + lane;
            const d2 w = R.ww[q];
            if (i < 0) {
                const BlkDesc d = a.blk[blk_first + q];
                const d2 rh = lane < d.nrows() ? a.rhs[d.row0 + lane] : make_double2(0.0, 0.0);
                R.rr[q] = make_double2(rh.x - w.x, rh.y - w.y);                  // r_0 = rhs - M v      conjugategradients.jl:33
            } else {
                d2 pq, sq;
                if (i == 0) { pq = R.rr[q]; sq = w; }
                else {
                    pq = s_ps[e]; sq = s_ps[e + 64];
                    pq.x = pq.x * beta + R.rr[q].x; pq.y = pq.y * beta + R.rr[q].y;      // p .*= beta ; p .+= r     :49-50
                    sq.x = sq.x * beta + w.x; sq.y = sq.y * beta + w.y;                  // s = M p by the same recurrence
                }
                R.xx[q].x += alpha * pq.x; R.xx[q].y += alpha * pq.y;                    // :40
                R.rr[q].x -= alpha * sq.x; R.rr[q].y -= alpha * sq.y;                    // :41
                s_ps[e] = pq; s_ps[e + 64] = sq;
            }
        }
        if ((q & 1) == 1) __builtin_amdgcn_sched_barrier(0);     // (two tiles' p and s in flight at a time: all of them together were 80 registers on top of r, w, x)
    }
}

// An exchange's incoming words, dealt to ALL threads of the workgroup (the compute wavefronts wait for the exchange anyway): item idx < 4 G is value
// idx & 3 of workgroup idx >> 2's record, the items behind them are the column sums of the unit's other workgroups.  Thread `t` of `nthr` polls items
// t, t + nthr, ... -- PU of them requested and polled together.  Returns false when a word did not arrive within the time-out.
__device__ __forceinline__ bool rs_poll(const ResArgs& a, const ResWG& me, unsigned long long* grec, size_t par, uint32_t seq, int t, int nthr,
                                        double* __restrict__ s_all, double* __restrict__ s_sib) {
    const int tc = me.tc;
    const int nrec = 4 * a.G, nsib = (me.wpu - 1) * tc * 2;
    const int sleepn = (a.flags >> 8) & 0x7F;
    constexpr int PU = 3;
    bool bad = false;
    for (int base = t; base < nrec + nsib; base += PU * nthr) {
        const unsigned long long* src[PU];
        double* dst[PU];
#pragma unroll
        for (int u = 0; u < PU; ++u) {
            const int idx = base + u * nthr;
            src[u] = nullptr; dst[u] = nullptr;
            if (idx < nrec) {
                const int wg = idx >> 2, kk = idx & 3;
                dst[u] = s_all + (size_t)kk * RS_GMAX + wg;
                src[u] = wg == (int)blockIdx.x ? nullptr : grec + (size_t)wg * 8 + 2 * kk;
            } else if (idx < nrec + nsib) {
                const int j = idx - nrec, comp = j & 1, c = (j >> 1) % tc, si = (j >> 1) / tc;
                const int kk = si < me.idx ? si : si + 1;
                dst[u] = s_sib + ((size_t)si * 64 + c) * 2 + comp;
                src[u] = a.crec + ((par * a.G + (size_t)(me.wg0 + kk)) * (size_t)a.tmax + (size_t)c) * 4 + 2 * comp;
            }
        }
        long long tstart = 0;
        for (uint32_t spin = 1;; ++spin) {
            unsigned long long lo[PU], hi[PU];
#pragma unroll
            for (int u = 0; u < PU; ++u) if (src[u]) { lo[u] = res_ld_word(src[u]); hi[u] = res_ld_word(src[u] + 1); }
            bool pending = false;
#pragma unroll
            for (int u = 0; u < PU; ++u) {
                if (src[u]) {
                    if ((uint32_t)(lo[u] >> 32) == seq && (uint32_t)(hi[u] >> 32) == seq) {
                        *dst[u] = __longlong_as_double((long long)((hi[u] << 32) | (lo[u] & 0xFFFFFFFFull)));
                        src[u] = nullptr;
                    } else pending = true;
                }
            }
            if (!pending) break;
            if ((spin & 255u) == 0u) {
                const long long now = wall_clock64();
                if (tstart == 0) tstart = now;
                else if (now - tstart >= a.timeout_ticks) { bad = true; break; }
            }
            if (sleepn >= 2) __builtin_amdgcn_s_sleep(2); else if (sleepn >= 1) __builtin_amdgcn_s_sleep(1);
        }
    }
    return !bad;
}

template <int TMAX, int NT>
__global__ __launch_bounds__(64 * (RS_NCOMP + 1)) void cg_stream_kernel(ResArgs a) {
    __shared__ __attribute__((aligned(16))) d2 s_gcol[64];                   // the workgroup's column elements of the vector being swept (v, then r)
    __shared__ double s_red[16][4];
    __shared__ double s_ctl[RC_COUNT];
    __shared__ uint32_t s_halves[PEER_MAX_RANKS * 8];
    __shared__ int s_cnt, s_failed;
    extern __shared__ __attribute__((aligned(16))) double s_dyn[];
    d2* const s_colpart = reinterpret_cast<d2*>(s_dyn);                       // [RS_NCOMP + 1][64]: a wavefront's column sums of a sweep
    double* const s_all = reinterpret_cast<double*>(s_colpart + (RS_NCOMP + 1) * 64);      // [4][RS_GMAX], zero beyond G
    double* const s_sib = s_all + 4 * RS_GMAX;                                // [RS_WPU_MAX - 1][64][2]: the column sums of the unit's other workgroups
    d2* const s_ps = reinterpret_cast<d2*>(s_sib + (RS_WPU_MAX - 1) * 64 * 2);    // [tiles][2][64]: the rows' p and s
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6), nw = (int)(blockDim.x >> 6);
    constexpr int ncomp = RS_NCOMP;
    const int ncomm = nw - ncomp;
    const ResWG me = a.wg[blockIdx.x];
    const int tc = me.tc, c0 = me.c0, nm = a.nm;
    DevState* st = a.st;
#ifdef FOS_RES_STAMPS
    int stamp_slot = -1, stamp_it = 0;
    {
        const int b = (int)blockIdx.x;
        const int wslot = b == 0 ? 0 : (b == 1 ? 1 : (b == a.G / 2 ? 2 : (b == a.G - 1 ? 3 : -1)));
        if (wslot >= 0 && lane == 0 && wv < 16) stamp_slot = wslot * 16 + wv;
    }
#endif
    if (a.pb.nranks > 0 && st->xchg_failed) return;
    if (tid == 0) { s_cnt = 0; s_failed = 0; }
    RES_STAMP_CAL(63);
    int t0, cnt;
    rs_split(me.nblk, wv, t0, cnt);                        // (compute wavefronts: cnt <= NT, the communication wavefront: cnt <= RS_NTC -- the plan's promise)

    if (wv < ncomp) {
        // =========================================================== COMPUTE wavefronts: their tiles streamed once per iteration
        RsRows<NT> R;
        rs_rows_load<NT>(a, me.blk0 + t0, cnt, lane, R);
        d2* const mycol = s_colpart + wv * 64;
        __syncthreads();                                   // (0) the communication wavefront has staged the v columns and the tau element
        RES_STAMP(0);
        // one loop turn = one EXCHANGE ROUND (sequence number xr + 1).  Ordinarily a round is a sweep of iteration i and its sums; an EARLY round
        // carries r.r alone: the merged recurrence learns |r| only behind the next sweep, so when the last residual norm says the solve is about to
        // stop (RC_NEAR), the rows' r.r -- the very sum the next sweep would form -- is exchanged first, and a solve that has converged skips that sweep
        bool early = false;
        for (int i = -1, xr = 0;; ++xr) {
            double acc[4];
            rs_sweep<TMAX, NT>(a, me.blk0 + t0, cnt, c0, lane, s_gcol, s_ctl, mycol, R, acc, early);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const double v = wave_sum(acc[k]);
                if (lane == 0) s_red[wv][k] = v;
            }
            RES_STAMP(1);                                   // swept
            __syncthreads();                               // (A) column sums and the wavefronts' sums are in LDS
            RES_STAMP(2);
            {   // this wavefront's share of the exchange's incoming words (it would wait at (B) otherwise)
                const uint32_t seq = a.seq_base + (uint32_t)(xr + 1);
                const size_t par = (size_t)(seq & 1u);
                if (!rs_poll(a, me, a.grec + par * (size_t)a.G * 8, par, seq, tid, (int)blockDim.x, s_all, s_sib)) s_failed = 1;
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                if (lane == 0) __hip_atomic_fetch_add(&s_cnt, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
            __syncthreads();                               // (B) alpha, beta, the new columns and tau element
            RES_STAMP(3);
            if (s_ctl[RC_STOP] != 0.0) break;
            if (!early) {
                rs_update<NT>(a, me.blk0 + t0, t0, cnt, lane, i, s_ctl[RC_ALPHA], s_ctl[RC_BETA], s_ps, R);
                early = s_ctl[RC_NEAR] != 0.0;
                ++i;
            } else early = false;                          // (not converged yet: the sweep of the same iteration follows)
            RES_STAMP(4);                                   // updated
            RES_STAMP_NEXT();
            RES_STAMP(0);
        }
        // ---- the solution leaves the registers (an exchange that failed leaves x as it was: the caller gets an error, not a half-updated iterate)
        if (s_ctl[RC_STOP] == 1.0) rs_rows_store<NT>(a, me.blk0 + t0, cnt, lane, R);
        RES_STAMP_CAL(62);
        return;
    }

    // =============================================================== COMMUNICATION wavefronts (the register form's, a unit = this workgroup alone)
    const int cw = wv - ncomp, ct = tid - 64 * ncomp;
    const bool c0wave = cw == 0, leader = me.idx == 0;              // (a unit split over wpu workgroups: its columns are counted once, by the first)
    d2 cx = make_double2(0.0, 0.0), cr = cx, cpv = cx, csv = cx, crhs = cx;
    double cc = 0.0;
    d2 gt = a.v[nm], xt = a.x[nm], pt = make_double2(0.0, 0.0), stt = pt;
    const d2 rhst = a.rhs[nm];
    for (int q = ct; q < 4 * RS_GMAX; q += 64 * ncomm) s_all[q] = 0.0;
    for (int q = ct + (ncomp + 1) * 4; q < 16 * 4; q += 64 * ncomm) (&s_red[0][0])[q] = 0.0;
    // (this wavefront waits at barrier (A) while the others sweep: it walks the workgroup's last RS_NTC tiles itself -- 66 tiles are 7 x 9 + 3)
    RsRows<RS_NTC> R;
    rs_rows_load<RS_NTC>(a, me.blk0 + t0, c0wave ? cnt : 0, lane, R);
    d2* const mycol = s_colpart + ncomp * 64;
    if (c0wave) {
        if (lane < tc) { cr = a.v[c0 + lane]; cx = a.x[c0 + lane]; crhs = a.rhs[c0 + lane]; cc = a.cb[c0 + lane]; }
        s_gcol[lane] = cr;
        if (lane == 0) { s_ctl[RC_GTX] = gt.x; s_ctl[RC_GTY] = gt.y; s_ctl[RC_STOP] = 0.0; }      // (the compute wavefronts read the tau element here, from the start)
        if (blockIdx.x == 0 && lane == 0) { st->tol = a.tol; st->maxit = a.maxit; st->hit_max = 0; st->rn_old = 0.0; }
    }
    __syncthreads();                                           // (0)
    double colG = 0.0;
    double g_prev = 0.0, a_prev = 0.0, gam = 0.0;
    int iter = 0;
    uint32_t nx = 0;
    RES_STAMP(0);
    bool early = false;                                        // (exchange rounds: see the compute wavefronts' loop)
    for (int i = -1, xr = 0;; ++xr) {
        const uint32_t seq = a.seq_base + (uint32_t)(xr + 1);
        const size_t par = (size_t)(seq & 1u);
        unsigned long long* grec = a.grec + par * (size_t)a.G * 8;
        nx += 1;
        if (c0wave) {
            double racc[4];
            rs_sweep<TMAX, RS_NTC>(a, me.blk0 + t0, cnt, c0, lane, s_gcol, s_ctl, mycol, R, racc, early);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const double v = wave_sum(racc[k]);
                if (lane == 0) s_red[ncomp][k] = v;
            }
        }
        __syncthreads();                                       // (A)
        RES_STAMP(1);
        d2 cp = make_double2(0.0, 0.0);
        if (c0wave) {
            double acc[4] = {0.0, 0.0, 0.0, 0.0};
            {
                d2 o[ncomp + 1];
#pragma unroll
                for (int s = 0; s < ncomp + 1; ++s) o[s] = s_colpart[s * 64 + lane];       // (all in flight together, added in wavefront order)
#pragma unroll
                for (int s = 0; s < ncomp + 1; ++s) { cp.x += o[s].x; cp.y += o[s].y; }
            }
            if (lane < tc && early) { if (leader) acc[0] = colG; }                        // (an early round carries r.r alone)
            else if (lane < tc) {
                acc[1] = cp.x * cr.y - cp.y * cr.x;                                      // this workgroup's share of (w.g) of column `lane` (i < n)
                if (leader) {                                                            // the slot-free part and the other sums: once per column
                    acc[0] = colG;
                    acc[1] += (cr.x * cr.x - cr.y * cr.y) + cc * (gt.x * cr.y - gt.y * cr.x);
                    acc[2] = cc * cr.x;
                    acc[3] = cc * cr.y;
                }
            }
            double mine[4];
            {
                double rv[12];
                const int k4 = lane & 3;
#pragma unroll
                for (int w = 0; w < 12; ++w) rv[w] = s_red[w][k4];                      // (rows beyond ncomp were zeroed once)
                double sacc = 0.0;
#pragma unroll
                for (int w = 0; w < 12; ++w) sacc += rv[w];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int lo = __builtin_amdgcn_readlane(__double2loint(sacc), k), hi = __builtin_amdgcn_readlane(__double2hiint(sacc), k);
                    mine[k] = __hiloint2double(hi, lo) + wave_sum(acc[k]);
                }
            }
            if (i < 0) mine[0] = 0.0;
            if (early) { mine[1] = 0.0; mine[2] = 0.0; mine[3] = 0.0; }
            if (lane < 8) {
                const int k = lane >> 1;
                res_publish_half(grec + (size_t)blockIdx.x * 8 + 2 * k, seq, k == 0 ? mine[0] : (k == 1 ? mine[1] : (k == 2 ? mine[2] : mine[3])), lane & 1);
            }
            if (lane < 4) s_all[(size_t)lane * RS_GMAX + blockIdx.x] = lane == 0 ? mine[0] : (lane == 1 ? mine[1] : (lane == 2 ? mine[2] : mine[3]));
            if (me.wpu > 1 && lane < tc) {
                unsigned long long* p = a.crec + ((par * a.G + blockIdx.x) * (size_t)a.tmax + (size_t)lane) * 4;
                res_publish_half(p, seq, cp.x, 0); res_publish_half(p, seq, cp.x, 1);
                res_publish_half(p + 2, seq, cp.y, 0); res_publish_half(p + 2, seq, cp.y, 1);
            }
        }
        RES_STAMP(2);                                           // published
        {
            if (!rs_poll(a, me, grec, par, seq, tid, (int)blockDim.x, s_all, s_sib)) s_failed = 1;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            if (lane == 0) __hip_atomic_fetch_add(&s_cnt, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
        RES_STAMP(3);                                           // this wavefront's words have arrived
        if (c0wave) {
            {
                long long tstart = 0;
                for (uint32_t spin = 1; __hip_atomic_load(&s_cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < (int)(nx * (uint32_t)nw); ++spin) {
                    if ((spin & 4095u) == 0u) {
                        const long long now = wall_clock64();
                        if (tstart == 0) tstart = now;
                        else if (now - tstart >= 2 * a.timeout_ticks) { s_failed = 1; break; }
                    }
                }
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
            }
            RES_STAMP(4);
            bool failed = __hip_atomic_load(&s_failed, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) != 0;
            double tot[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                double rv[RS_GMAX / 64];
#pragma unroll
                for (int j = 0; j < RS_GMAX / 64; ++j) rv[j] = s_all[(size_t)k * RS_GMAX + lane + 64 * j];
                double sacc = 0.0;
#pragma unroll
                for (int j = 0; j < RS_GMAX / 64; ++j) sacc += rv[j];
                tot[k] = wave_sum(sacc);
            }
            if (!failed && a.pb.nranks > 0) failed = !res_peer_fold4(a.pb, seq, tot, blockIdx.x == 0, s_halves);
            RES_STAMP(5);                                       // totals (over the ranks)
            // the unit's column sums: its workgroups in order -- the same bits in each of them
            d2 ctot = cp;
            if (me.wpu > 1 && lane < tc) {
                const d2* sib = reinterpret_cast<const d2*>(s_sib);
                d2 t = make_double2(0.0, 0.0);
                for (int kk = 0; kk < me.wpu; ++kk) {
                    d2 part = cp;
                    if (kk != me.idx) part = sib[(size_t)(kk < me.idx ? kk : kk - 1) * 64 + lane];
                    t.x += part.x; t.y += part.y;
                }
                ctot = t;
            }
            // column `lane` of w = M g (EpiKkt::row, i < n: HSDEAffine.jl:51,54) and the tau row (HSDEAffine.jl:57)
            const double q1 = ctot.x + gt.x * cc, q2 = ctot.y + gt.y * cc;
            const d2 cwv = make_double2(cr.x - q2, q1 - cr.y);
            const d2 wt = make_double2(gt.x + tot[3], -tot[2] - gt.y);
            double stopf = failed ? 2.0 : 0.0, alpha = 0.0, beta = 0.0, near = 0.0;
            if (!failed && early) {
                gam = tot[0] + (gt.x * gt.x + gt.y * gt.y);                              // |r_i|^2 of the residual the last update left: the sum the sweep would form
                if (i > 0 && (sqrt(gam) <= a.tol || i >= a.maxit)) { iter = i; stopf = 1.0; }       // conjugategradients.jl:42 for iteration i, without its sweep
            } else if (!failed) {
                if (i < 0) {
                    cr = make_double2(crhs.x - cwv.x, crhs.y - cwv.y);                  // r_0 = rhs - M v      conjugategradients.jl:32-36
                    gt = make_double2(rhst.x - wt.x, rhst.y - wt.y);
                } else {
                    gam = tot[0] + (gt.x * gt.x + gt.y * gt.y);
                    if (i > 0 && (sqrt(gam) <= a.tol || i >= a.maxit)) { iter = i; stopf = 1.0; }       // conjugategradients.jl:42 for iteration i
                    else {
                        const double delta = tot[1] + (wt.x * gt.x + wt.y * gt.y);
                        if (i == 0) alpha = gam / delta;
                        else {
                            beta = gam / g_prev;
                            alpha = gam / (delta - beta * gam / a_prev);
                        }
                        g_prev = gam; a_prev = alpha;
                        auto upd = [&](const d2 wi, d2& ri, d2& pi, d2& si, d2& xi) {
                            if (i == 0) { pi = ri; si = wi; }
                            else {
                                pi.x = pi.x * beta + ri.x; pi.y = pi.y * beta + ri.y;
                                si.x = si.x * beta + wi.x; si.y = si.y * beta + wi.y;
                            }
                            xi.x += alpha * pi.x; xi.y += alpha * pi.y;
                            ri.x -= alpha * si.x; ri.y -= alpha * si.y;
                        };
                        if (lane < tc) upd(cwv, cr, cpv, csv, cx);
                        upd(wt, gt, pt, stt, xt);
                        // the residual norm falls by a factor of 2 .. 3 per iteration on these systems: within 3 tol (or at the cap) the next test is likely
                        // to end the solve -- an exchange of r.r alone (7 us) is tried before the sweep (50 us) that would otherwise find it out
                        near = (sqrt(gam) <= 3.0 * a.tol || i + 1 >= a.maxit) ? 1.0 : 0.0;
                    }
                }
            }
            if (stopf == 0.0 && !early) {
                colG = (leader && lane < tc) ? cr.x * cr.x + cr.y * cr.y : 0.0;
                s_gcol[lane] = lane < tc ? cr : make_double2(0.0, 0.0);
            }
            if (lane == 0) {
                if (!early) { s_ctl[RC_ALPHA] = alpha; s_ctl[RC_BETA] = beta; s_ctl[RC_GTX] = gt.x; s_ctl[RC_GTY] = gt.y; s_ctl[RC_NEAR] = near; }
                s_ctl[RC_STOP] = stopf;
            }
        }
        RES_STAMP(6);                                           // scalars, columns, tau
        __syncthreads();                                       // (B)
        RES_STAMP(7);
        RES_STAMP_NEXT();
        RES_STAMP(0);
        if (s_ctl[RC_STOP] != 0.0) break;
        if (!early) {
            if (c0wave) rs_update<RS_NTC>(a, me.blk0 + t0, t0, cnt, lane, i, s_ctl[RC_ALPHA], s_ctl[RC_BETA], s_ps, R);
            early = s_ctl[RC_NEAR] != 0.0;
            ++i;
        } else early = false;                                  // (not converged yet: the sweep of the same iteration follows)
    }
    RES_STAMP_CAL(62);
    if (c0wave) {
        const bool ok = s_ctl[RC_STOP] == 1.0;
        if (ok) rs_rows_store<RS_NTC>(a, me.blk0 + t0, cnt, lane, R);
        if (ok && leader && lane < tc) a.x[c0 + lane] = cx;
        if (blockIdx.x == 0 && lane == 0) {
            if (ok) {
                a.x[nm] = xt;
                st->rr = gam;
                cg_signal_stop(st, iter, a.maxit, gam, a.seq_base >> 11);
            } else {
                if (a.pb.nranks > 0) st->xchg_failed = 1;
                st->bar_failed = 1; st->done = 1;
            }
        }
    }
}

// dynamic LDS above the default limit needs an opt-in per kernel (a table update; a failure surfaces through the launch check)
template <class K>
static void res_lds_optin(K kernel, size_t bytes) {
    if (bytes > 48 * 1024) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
}

void launch_cg_resident(const LaunchCtx& c, const ResLaunch& rl, double2* x, const double2* rhs, const double2* v, double tol, int maxit,
                        const PeerBox* fold, uint32_t seq_base) {
    ResArgs a{};
    a.x = x; a.rhs = rhs; a.v = v; a.cb = c.cb; a.n = (int)c.n; a.nm = (int)(c.n + c.m); a.st = c.st;
    a.blk = c.S.blk; a.val = c.S.val; a.wg = rl.wg; a.G = rl.G; a.grec = rl.grec; a.crec = rl.crec; a.tmax = rl.tmax;
    a.tol = tol; a.maxit = maxit;
    a.pb = fold ? *fold : PeerBox{}; a.seq_base = seq_base;
    a.timeout_ticks = rl.timeout_ticks;
    a.ncomp = rl.nw;
    static const int res_flags = getenv("FOS_RES_FLAGS") ? atoi(getenv("FOS_RES_FLAGS")) : (2 << 8);
    a.flags = res_flags;
    if (rl.stream) {
        // (the compute wavefronts' sums land in s_red[wv]: RS_NCOMP rows, the others zeroed once)
        dim3 grid(rl.G), block(64 * (RS_NCOMP + 1));
        const size_t lds = (size_t)(RS_NCOMP + 1) * 64 * sizeof(d2) + (size_t)4 * RS_GMAX * sizeof(double) + (size_t)(RS_WPU_MAX - 1) * 64 * 2 * sizeof(double) +
                           (size_t)rl.tiles_wg_max * 128 * sizeof(d2);
        if (rl.tmax > 32 && rl.nt <= 3) { res_lds_optin(cg_stream_kernel<64, 3>, lds); hipLaunchKernelGGL((cg_stream_kernel<64, 3>), grid, block, lds, c.stream, a); }
        else if (rl.tmax > 32) { res_lds_optin(cg_stream_kernel<64, 5>, lds); hipLaunchKernelGGL((cg_stream_kernel<64, 5>), grid, block, lds, c.stream, a); }
        else if (rl.nt <= 3) { res_lds_optin(cg_stream_kernel<32, 3>, lds); hipLaunchKernelGGL((cg_stream_kernel<32, 3>), grid, block, lds, c.stream, a); }
        else if (rl.nt <= 5) { res_lds_optin(cg_stream_kernel<32, 5>, lds); hipLaunchKernelGGL((cg_stream_kernel<32, 5>), grid, block, lds, c.stream, a); }
        else if (rl.nt <= 9) { res_lds_optin(cg_stream_kernel<32, 9>, lds); hipLaunchKernelGGL((cg_stream_kernel<32, 9>), grid, block, lds, c.stream, a); }
        else { res_lds_optin(cg_stream_kernel<32, 10>, lds); hipLaunchKernelGGL((cg_stream_kernel<32, 10>), grid, block, lds, c.stream, a); }
        return;
    }
    dim3 grid(rl.G), block(64 * (rl.nw + rl.ncomm));
    auto lds_bytes = [&](size_t nslot, size_t tmax) {
        return nslot * 64 * 4 * sizeof(d2) + nslot * tmax * sizeof(d2) + (size_t)(RES_WPU_MAX - 1) * tmax * 2 * sizeof(double) + (size_t)4 * RES_GMAX * sizeof(double);
    };
    if (rl.tmax <= 32 && rl.rpt == 1) { const size_t lds = lds_bytes(12, 32); res_lds_optin(cg_resident_kernel<32, 1, 768>, lds); hipLaunchKernelGGL((cg_resident_kernel<32, 1, 768>), grid, block, lds, c.stream, a); }
    else if (rl.tmax <= 32 && rl.rpt == 2) { const size_t lds = lds_bytes(16, 32); res_lds_optin(cg_resident_kernel<32, 2, 512>, lds); hipLaunchKernelGGL((cg_resident_kernel<32, 2, 512>), grid, block, lds, c.stream, a); }
    else if (rl.tmax <= 32) { const size_t lds = lds_bytes(24, 32); res_lds_optin(cg_resident_kernel<32, 3, 512>, lds); hipLaunchKernelGGL((cg_resident_kernel<32, 3, 512>), grid, block, lds, c.stream, a); }
    else { const size_t lds = lds_bytes(8, 64); res_lds_optin(cg_resident_kernel<64, 1, 512>, lds); hipLaunchKernelGGL((cg_resident_kernel<64, 1, 512>), grid, block, lds, c.stream, a); }
}

}  // namespace fos

// (not part of the ABI: timing experiments -- tools/res_stamps.py; -1 unless compiled with -DFOS_RES_STAMPS)
extern "C" int fos_debug_res_stamps(long long* out, int n) {
#ifdef FOS_RES_STAMPS
    if (n > 4 * 16 * 64 * 16) n = 4 * 16 * 64 * 16;
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(fos::g_res_stamps), sizeof(long long) * (size_t)n);
#else
    (void)out; (void)n; return -1;
#endif
}
