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
//   * sharded: the four sums then cross the GPUs through the handle's mailboxes (peer_fold_sum), as in cgm_update_kernel.
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

// One tile out of registers: row sums (u1, u2) of the lane's row against the unit's column elements (LDS, broadcast reads), and the
// tile's column sums -- 8 columns at a time through the butterfly of the dual tiles -- into the workgroup's LDS array.
template <int TMAX>
__device__ __forceinline__ void res_tile(const double (&val)[TMAX], int T, const d2 g, const d2* __restrict__ gcol, d2* __restrict__ colpart, int lane,
                                         double& u1, double& u2) {
    u1 = 0.0; u2 = 0.0;
#pragma unroll
    for (int grp = 0; grp < TMAX / TILE_GROUP; ++grp) {
        if (grp * TILE_GROUP < T) {                                  // wave-uniform
            // (one right-hand side after the other: eight products live at a time, not sixteen -- the values, the rows' vector elements
            //  and the sums of a whole solve share the registers with this loop)
            double p[TILE_GROUP];
#pragma unroll
            for (int h = 0; h < 2; ++h) {                            // (four column elements = 16 registers in flight at a time)
#pragma unroll
                for (int u = 4 * h; u < 4 * h + 4; ++u) {
                    const int c = grp * TILE_GROUP + u;
                    const d2 xc = gcol[c];
                    u1 += val[c] * xc.x; u2 += val[c] * xc.y;
                    p[u] = val[c] * g.x;
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            const double s1 = tile_colsum8(p, lane);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int u = 0; u < TILE_GROUP; ++u) p[u] = val[grp * TILE_GROUP + u] * g.y;
            const double s2 = tile_colsum8(p, lane);
            if (lane < TILE_GROUP) colpart[grp * TILE_GROUP + lane] = make_double2(s1, s2);
            __builtin_amdgcn_sched_barrier(0);                       // (keeps the next group's LDS reads from being hoisted up here: 4 registers per column element)
        }
    }
}

template <int TMAX, int RPT, int LB>
__global__ __launch_bounds__(LB) void cg_resident_kernel(ResArgs a) {
    __shared__ __attribute__((aligned(16))) d2 s_gcol[TMAX];                 // the unit's column elements of the vector being swept (v, then r)
    __shared__ __attribute__((aligned(16))) d2 s_cx[TMAX], s_cp[TMAX], s_cs[TMAX];    // ... of x, p, s (touched by thread t = column t only)
    __shared__ __attribute__((aligned(16))) d2 s_tau[4];                      // tau elements of r, x, p, s (thread 0 updates; everybody reads r's)
    __shared__ double s_red[16][4];
    // dynamic: per tile slot the rows' elements of x, p, s, w (lane-private: only the residual, which the sweep multiplies by, lives in
    // registers beside the matrix values), the slots' column sums, the G records of an exchange
    // (NSLOT is a compile-time constant so that every one of these addresses is (thread's 16-byte offset) + an immediate: held as
    //  runtime values they were spilled to scratch and reloaded one by one in front of every LDS access of the update)
    extern __shared__ __attribute__((aligned(16))) double s_dyn[];
    constexpr int NSLOT = (LB / 64) * RPT;
    d2* const s_x = reinterpret_cast<d2*>(s_dyn);                             // [NSLOT][64]
    d2* const s_p = s_x + NSLOT * 64;
    d2* const s_s = s_p + NSLOT * 64;
    d2* const s_w = s_s + NSLOT * 64;
    d2* const s_colpart = s_w + NSLOT * 64;                                   // [NSLOT][TMAX]
    double* const s_all = reinterpret_cast<double*>(s_colpart + NSLOT * TMAX);      // [4][G]
    __shared__ double s_sums[4];
    __shared__ int s_failed;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6), nw = (int)(blockDim.x >> 6);
    const ResWG me = a.wg[blockIdx.x];
    const int T = me.T, tc = me.tc, c0 = me.c0, nm = a.nm;
    const bool leader = me.idx == 0, colthr = tid < tc;
    DevState* st = a.st;

    // ---------------- the workgroup's tiles -> registers; its rows' vector elements
    double val[RPT][TMAX];
    d2 rr[RPT], rhsr[RPT];      // (rr: the vector being swept -- v at the start, then the residual; rhsr: dead after the start)
    double cbr[RPT];
    int row[RPT];
    bool has[RPT], valid[RPT];
#pragma unroll
    for (int q = 0; q < RPT; ++q) {
        const int ti = wv + q * nw;
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
            // (all loads unconditional -- one request phase; steps beyond T and lanes beyond the tile's rows become zeros)
#pragma unroll
            for (int t = 0; t < TMAX; ++t) {
                const double v = vp[64 * (t < T ? t : T - 1)];
                val[q][t] = (t < T && valid[q]) ? v : 0.0;
            }
            d2 x0 = make_double2(0.0, 0.0);
            if (valid[q]) { x0 = a.x[row[q]]; rr[q] = a.v[row[q]]; rhsr[q] = a.rhs[row[q]]; cbr[q] = a.cb[row[q]]; }
            s_x[(size_t)ti * 64 + lane] = x0;
        }
    }
    // the unit's columns (replicated in every workgroup of the unit), the tau element (replicated everywhere)
    double cc = 0.0;
    d2 crhs = make_double2(0.0, 0.0);
    if (tid < TMAX) {
        d2 z = make_double2(0.0, 0.0);
        s_gcol[tid] = colthr ? a.v[c0 + tid] : z;
        s_cx[tid] = colthr ? a.x[c0 + tid] : z;
        s_cp[tid] = z; s_cs[tid] = z;
        if (colthr) { cc = a.cb[c0 + tid]; crhs = a.rhs[c0 + tid]; }
    }
    d2 gt = a.v[nm];                                   // tau element of the vector being swept
    const d2 rhst = a.rhs[nm];
    if (tid == 0) { s_tau[1] = a.x[nm]; s_tau[2] = make_double2(0.0, 0.0); s_tau[3] = make_double2(0.0, 0.0); }
    if (blockIdx.x == 0 && tid == 0) { st->tol = a.tol; st->maxit = a.maxit; st->hit_max = 0; st->rn_old = 0.0; }
    __syncthreads();

    // ---------------- sweep: w = M g on the workgroup's rows (wr), its partial column sums (returned for column tid), the sums
    // acc[1] += w.r (rows; columns through the bilinear form of EpiKkt::park / deferred_local), acc[2], acc[3] += [c;b].g
    auto sweep = [&](double (&acc)[4]) -> d2 {
#pragma unroll
        for (int q = 0; q < RPT; ++q) {
            if (has[q]) {
                double u1, u2;
                const d2 gq = rr[q];
                res_tile<TMAX>(val[q], T, gq, s_gcol, s_colpart + (size_t)(wv + q * nw) * TMAX, lane, u1, u2);
                const double c = cbr[q];
                const double q1 = -(u1 - gt.x * c), q2 = -(u2 - gt.y * c);           // rows of A: EpiKkt::row, i >= n   (HSDEAffine.jl:52,55)
                d2 w = make_double2(gq.x - q2, q1 - gq.y);                           // affinepluslinear.jl:45-48
                if (!valid[q]) w = make_double2(0.0, 0.0);
                s_w[(size_t)(wv + q * nw) * 64 + lane] = w;
                acc[1] += w.x * gq.x + w.y * gq.y;
                acc[2] += c * gq.x;
                acc[3] += c * gq.y;
            }
        }
        __syncthreads();
        d2 cp = make_double2(0.0, 0.0);
        if (colthr) {
            for (int s = 0; s < me.nblk; ++s) { const d2 o = s_colpart[(size_t)s * TMAX + tid]; cp.x += o.x; cp.y += o.y; }
            const d2 gc = s_gcol[tid];
            acc[1] += cp.x * gc.y - cp.y * gc.x;                                     // this workgroup's share of (w.g) of column tid (EpiKkt::park, i < n)
            if (leader) {                                                            // the slot-free part, once per column (EpiKkt::deferred_local)
                acc[1] += (gc.x * gc.x - gc.y * gc.y) + cc * (gt.x * gc.y - gt.y * gc.x);
                acc[2] += cc * gc.x;
                acc[3] += cc * gc.y;
            }
        }
        return cp;
    };

    // ---------------- exchange: acc (per lane) -> s_sums (grid totals, then totals over the ranks); cp -> the unit's column sums
    auto exchange = [&](uint32_t seq, const double (&acc)[4], const d2 cp, d2& ctot) -> bool {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const double v = wave_sum(acc[k]);
            if (lane == 0) s_red[wv][k] = v;
        }
        if (tid == 0) s_failed = 0;
        const size_t par = (size_t)(seq & 1u);
        unsigned long long* crec_me = a.crec + ((par * a.G + blockIdx.x) * (size_t)a.tmax) * 4;
        if (me.wpu > 1 && colthr) {
            unsigned long long* p = crec_me + (size_t)tid * 4;
            res_publish_half(p, seq, cp.x, 0); res_publish_half(p, seq, cp.x, 1);
            res_publish_half(p + 2, seq, cp.y, 0); res_publish_half(p + 2, seq, cp.y, 1);
        }
        __syncthreads();
        unsigned long long* grec = a.grec + par * (size_t)a.G * 8;
        if (tid < 8) {
            const int k = tid >> 1;
            double s = 0.0;
            for (int w = 0; w < nw; ++w) s += s_red[w][k];
            res_publish_half(grec + (size_t)blockIdx.x * 8 + 2 * k, seq, s, tid & 1);
        }
        bool bad = false;
        for (int idx = tid; idx < 4 * a.G; idx += (int)blockDim.x) {
            const int wg = idx >> 2, k = idx & 3;
            double s;
            if (wg == (int)blockIdx.x) { s = 0.0; for (int w = 0; w < nw; ++w) s += s_red[w][k]; }
            else if (!res_poll_f64(grec + (size_t)wg * 8 + 2 * k, seq, a.timeout_ticks, s)) bad = true;
            s_all[(size_t)k * a.G + wg] = s;
        }
        ctot = cp;
        if (me.wpu > 1 && colthr) {
            d2 t = make_double2(0.0, 0.0);
            for (int k = 0; k < me.wpu; ++k) {                     // the unit's workgroups in order: the same bits in each of them
                d2 part = cp;
                if (k != me.idx) {
                    const unsigned long long* p = a.crec + ((par * a.G + (size_t)(me.wg0 + k)) * (size_t)a.tmax + (size_t)tid) * 4;
                    if (!res_poll_f64(p, seq, a.timeout_ticks, part.x) || !res_poll_f64(p + 2, seq, a.timeout_ticks, part.y)) bad = true;
                }
                t.x += part.x; t.y += part.y;
            }
            ctot = t;
        }
        if (bad) s_failed = 1;
        __syncthreads();
        if (s_failed) {
            if (blockIdx.x == 0 && tid == 0) { st->bar_failed = 1; st->done = 1; }
            return false;
        }
        for (int k = wv; k < 4; k += nw) {
            double s = 0.0;
            for (int i = lane; i < a.G; i += 64) s += s_all[(size_t)k * a.G + i];
            s = wave_sum(s);
            if (lane == 0) s_sums[k] = s;
        }
        __syncthreads();
        if (a.pb.nranks > 0) {
            if (!peer_fold_sum<4>(a.pb, seq, s_sums, st)) return false;
        }
        return true;
    };

    // column tid of w = M g from the unit's column sums (EpiKkt::row, i < n)
    auto col_w = [&](const d2 ctot, const d2 gc) -> d2 {
        const double q1 = ctot.x + gt.x * cc, q2 = ctot.y + gt.y * cc;               // HSDEAffine.jl:51,54
        return make_double2(gc.x - q2, q1 - gc.y);
    };

    // ---------------- start: r_0 = rhs - M v                      conjugategradients.jl:32-36
    if (a.pb.nranks > 0 && st->xchg_failed) return;
    double acc[4] = {0.0, 0.0, 0.0, 0.0};
    d2 ctot;
    d2 cp = sweep(acc);
    acc[0] = 0.0;
    if (!exchange(a.seq_base, acc, cp, ctot)) return;
    double accG = 0.0;
    {
        const double T1 = s_sums[2], T2 = s_sums[3];
        const d2 wt = make_double2(gt.x + T2, -T1 - gt.y);                           // (Q v)_tau = -[c;b].v     HSDEAffine.jl:57
#pragma unroll
        for (int q = 0; q < RPT; ++q) {
            if (has[q]) {
                const d2 w = s_w[(size_t)(wv + q * nw) * 64 + lane];
                rr[q] = make_double2(rhsr[q].x - w.x, rhsr[q].y - w.y);                  // :33   (zero on lanes without a row)
                accG += rr[q].x * rr[q].x + rr[q].y * rr[q].y;
            }
        }
        d2 cr = make_double2(0.0, 0.0);
        if (colthr) {
            const d2 cw = col_w(ctot, s_gcol[tid]);
            cr = make_double2(crhs.x - cw.x, crhs.y - cw.y);
            if (leader) accG += cr.x * cr.x + cr.y * cr.y;
        }
        gt = make_double2(rhst.x - wt.x, rhst.y - wt.y);
        __syncthreads();                                   // (every wavefront has read s_sums and the v columns)
        if (colthr) s_gcol[tid] = cr;
        if (tid == 0) s_tau[0] = gt;
        __syncthreads();
    }

    // ---------------- iterations (i = j - 1): sweep w = M r_i, exchange {g_i, d_i, tau-row sums}, close iteration i, update
    double g_prev = 0.0, a_prev = 0.0, gam = 0.0;
    int iter = 0;
    for (int i = 0;; ++i) {
        acc[0] = accG; acc[1] = 0.0; acc[2] = 0.0; acc[3] = 0.0;
        cp = sweep(acc);
        if (!exchange(a.seq_base + (uint32_t)(i + 1), acc, cp, ctot)) return;
        const double S1 = s_sums[1], T1 = s_sums[2], T2 = s_sums[3];
        gam = s_sums[0] + (gt.x * gt.x + gt.y * gt.y);
        if (i > 0 && (sqrt(gam) <= a.tol || i >= a.maxit)) { iter = i; break; }      // conjugategradients.jl:42 for iteration i
        const d2 wt = make_double2(gt.x + T2, -T1 - gt.y);
        const double delta = S1 + (wt.x * gt.x + wt.y * gt.y);
        double beta = 0.0, alpha;
        if (i == 0) alpha = gam / delta;
        else {
            beta = gam / g_prev;
            alpha = gam / (delta - beta * gam / a_prev);
        }
        g_prev = gam; a_prev = alpha;
        // (w, r, p, s, x) -> (p, s, x, r)                          :39-41,49-50 with Ap replaced by the recurrence s = M p
        auto upd = [&](const d2 wi, d2& ri, d2& pi, d2& si, d2& xi) {
            if (i == 0) { pi = ri; si = wi; }
            else {
                pi.x = pi.x * beta + ri.x; pi.y = pi.y * beta + ri.y;
                si.x = si.x * beta + wi.x; si.y = si.y * beta + wi.y;
            }
            xi.x += alpha * pi.x; xi.y += alpha * pi.y;
            ri.x -= alpha * si.x; ri.y -= alpha * si.y;
        };
        accG = 0.0;
#pragma unroll
        for (int q = 0; q < RPT; ++q) {
            if (has[q]) {                                                            // (lanes without a row: zeros stay zeros)
                const size_t e = (size_t)(wv + q * nw) * 64 + lane;
                d2 pq = make_double2(0.0, 0.0), sq = pq, xq = s_x[e];
                if (i > 0) { pq = s_p[e]; sq = s_s[e]; }
                upd(s_w[e], rr[q], pq, sq, xq);
                s_p[e] = pq; s_s[e] = sq; s_x[e] = xq;
                accG += rr[q].x * rr[q].x + rr[q].y * rr[q].y;
            }
        }
        d2 cr = make_double2(0.0, 0.0);
        if (colthr) {
            cr = s_gcol[tid];
            const d2 cw = col_w(ctot, cr);
            d2 cpv = s_cp[tid], csv = s_cs[tid], cxv = s_cx[tid];
            upd(cw, cr, cpv, csv, cxv);
            s_cp[tid] = cpv; s_cs[tid] = csv; s_cx[tid] = cxv;
            if (leader) accG += cr.x * cr.x + cr.y * cr.y;
        }
        d2 rt = gt;
        {
            d2 xt = s_tau[1], pt = s_tau[2], stt = s_tau[3];
            upd(wt, rt, pt, stt, xt);
            __syncthreads();                               // (every wavefront has read s_sums, s_tau and the r columns of this iteration)
            if (tid == 0) { s_tau[0] = rt; s_tau[1] = xt; s_tau[2] = pt; s_tau[3] = stt; }
        }
        gt = rt;
        if (colthr) s_gcol[tid] = cr;
        __syncthreads();
    }

    // ---------------- the solution leaves the registers
#pragma unroll
    for (int q = 0; q < RPT; ++q)
        if (valid[q]) a.x[row[q]] = s_x[(size_t)(wv + q * nw) * 64 + lane];
    if (leader && colthr) a.x[c0 + tid] = s_cx[tid];
    if (blockIdx.x == 0 && tid == 0) {
        a.x[nm] = s_tau[1];
        st->rr = gam;
        cg_signal_stop(st, iter, a.maxit, gam, a.seq_base >> 11);
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
    dim3 grid(rl.G), block(64 * rl.nw);
    auto lds_bytes = [&](size_t nslot, size_t tmax) { return nslot * 64 * 4 * sizeof(d2) + nslot * tmax * sizeof(d2) + (size_t)4 * rl.G * sizeof(double); };
    if (rl.tmax <= 32 && rl.rpt == 1) { const size_t lds = lds_bytes(12, 32); res_lds_optin(cg_resident_kernel<32, 1, 768>, lds); hipLaunchKernelGGL((cg_resident_kernel<32, 1, 768>), grid, block, lds, c.stream, a); }
    else if (rl.tmax <= 32) { const size_t lds = lds_bytes(16, 32); res_lds_optin(cg_resident_kernel<32, 2, 512>, lds); hipLaunchKernelGGL((cg_resident_kernel<32, 2, 512>), grid, block, lds, c.stream, a); }
    else { const size_t lds = lds_bytes(8, 64); res_lds_optin(cg_resident_kernel<64, 1, 512>, lds); hipLaunchKernelGGL((cg_resident_kernel<64, 1, 512>), grid, block, lds, c.stream, a); }
}

}  // namespace fos
