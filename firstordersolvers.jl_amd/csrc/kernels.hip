// Hand-written gfx950 kernels for the GAP/DR/GAPA/FISTA-over-HSDE hot path (everything except the batched PSD
// eigen-solver, psd.hip).  See DESIGN.md for the roofline of each kernel.
//
//   kkt2_kernel      fused dual-right-hand-side KKT apply  out = [I Q'; Q -I] w          (dominant, HBM bound)
//                    = 4 reference SpMV sweeps (affinepluslinear.jl:45-48 -> HSDEAffine.jl:51-52, twice)
//                    in ONE sweep over the stacked operator S = [[0,A'],[A,0]], plus the Q epilogue
//                    (HSDEAffine.jl:54-57), the KKT identity terms and the three reductions CG needs.
//   q1_kernel        single right-hand side Q apply with fused epilogues: rhs build (affinepluslinear.jl:94-95),
//                    plain Q / Q' apply (HSDEAffine.jl:41-65), status residual sums (HSDEStatus.jl:34-38,59,61).
//   cg_*             CG vector updates with in-pass reductions (conjugategradients.jl:33-50); scalars stay on
//                    the device, every CG kernel is gated on DevState.done so the host enqueues iterations
//                    ahead and polls once per chunk.
//   cones_*          segmented elementwise cones + batched SOC (cones.jl:122-142).
//   relaxation / extrapolation passes of gap.jl:48,58,78  gapa.jl:67,77,96-103  fista.jl:31-46.
//
// Wavefront = 64 lanes; workgroups of 256 threads (4 waves) unless noted; all arithmetic fp64.
#include <algorithm>
#include <type_traits>

#include "dev_common.hpp"

namespace fos {

// ------------------------------------------------------------------------------------------------ helpers

// Blocks b and b+8 share an XCD (round-robin dispatch); give every XCD one contiguous slice of the row blocks so
// its private L2 sees a compact window of the gathered vector.  Speed only, never correctness.
__device__ __forceinline__ int xcd_remap(int g, int nwg) {
    if (nwg & 7) return g;
    return (g & 7) * (nwg >> 3) + (g >> 3);
}

template <int NACC, int THREADS>
__device__ __forceinline__ void block_reduce_store(double (&acc)[NACC], double* smem, double* out) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    constexpr int NW = THREADS / 64;
#pragma unroll
    for (int a = 0; a < NACC; ++a) {
        double v = wave_sum(acc[a]);
        if (lane == 0) smem[wave * NACC + a] = v;
    }
    __syncthreads();
    if (tid < NACC) {
        double s = 0.0;
#pragma unroll
        for (int w = 0; w < NW; ++w) s += smem[w * NACC + tid];
        out[tid] = s;
    }
    __syncthreads();
}

// ------------------------------------------------------------------------------------------------ row-block SpMV core
//
// The unit of work is ONE WAVEFRONT (no workgroup barrier anywhere in the sweep): each of the grid's persistent
// wavefronts walks its own slice of row blocks.
//   Stream block: the block's <= WNNZ entries are loaded lane-consecutively (coalesced 8 B value + 4 B column per
//   lane, non-temporal: the matrix is read once per sweep and must not evict the gathered vector from L2),
//   multiplied by the gathered vector element(s) and staged in the wavefront's private LDS slice; then `tpr` lanes
//   per row (a power of two chosen from the number of rows in the block) sum that row's LDS segment and finish with
//   an in-register DPP butterfly.  Wave-synchronous: LDS operations of one wavefront execute in order.
//   Long row (> WNNZ entries): the wavefront strides the row, 4 loads in flight per lane, and reduces in-register.
// Summation order is fixed by the storage -> bit-reproducible run to run.

template <class T>
__device__ __forceinline__ T nt_load(const T* p) { return __builtin_nontemporal_load(p); }
// matrix stream of the row-block formats: non-temporal (read once per sweep; keeps the gathered vector in L2) unless the whole
// operator is small enough to STAY in the XCDs' L2s / the Infinity Cache from one sweep to the next (G::NT = false: a shard of a
// multi-GPU run, C3) -- a non-temporal line is dropped behind its reader, so every sweep would fetch the operator from HBM again
template <bool NT, class T>
__device__ __forceinline__ T mload(const T* p) { if constexpr (NT) return __builtin_nontemporal_load(p); else return *p; }

// what a row epilogue needs from memory besides the row sum: loaded at the START of the row block so that its
// latency overlaps the sweep of the block instead of trailing it
struct RowPre { d2 v; double c; };

typedef double v2d __attribute__((ext_vector_type(2)));

// Gathered vector element(s).  load(c): plain load through the per-CU L1.  (An L1-bypassing non-temporal gather was measured
// 1.6x SLOWER on random gathers (C5, sprandn) although each L1 fill brings a 128-byte line for 16 useful bytes -- the sweep
// of a random-sparse operator runs at the L2->L1 line rate, DESIGN.md.)
//   GatherW : element c of an interleaved vector w (both right-hand sides in one 16-byte load).
//   GatherP : the CG direction formed ON THE FLY, p_new[c] = r[c] + beta p_old[c] (conjugategradients.jl:49: p .*= beta;
//             p .+= r -- a multiply and an add, no contraction, so that every wavefront that needs element c and the one
//             that stores it compute the same bits): the p update of iteration j-1 rides on the sweep of iteration j.
//   Gather1 : one component of an interleaved vector (single right-hand side applies).
// load_u(c): the same element for a WAVE-UNIFORM index, read through the constant address space so that it becomes a scalar
// load (s_load_dwordx4: no vector-memory instruction, no VGPRs, no 1 KB of L1 return data for 16 useful bytes).  Valid because
// no kernel writes a vector it gathers from (p_new goes to the other ping-pong buffer), and a kernel boundary invalidates the
// scalar cache.
typedef const __attribute__((address_space(4))) v2d* cptr_v2d;
__device__ __forceinline__ d2 ld_const(const d2* p) { const v2d t = *(cptr_v2d)(p); return make_double2(t.x, t.y); }
typedef const __attribute__((address_space(4))) double* cptr_f64;
template <bool NTV>
struct GatherWT {
    static constexpr int NRHS = 2;
    static constexpr bool FUSED = false;
    static constexpr bool NT = NTV;
    const d2* w;
    __device__ __forceinline__ d2 load(int c) const { return w[c]; }
    __device__ __forceinline__ d2 load_u(int c) const { return ld_const(w + c); }
};
typedef GatherWT<true> GatherW;
struct GatherP {
    static constexpr int NRHS = 2;
    static constexpr bool FUSED = true;
    static constexpr bool NT = true;
    const d2* r;
    const d2* pold;
    double beta;
    __device__ __forceinline__ d2 load(int c) const {
#pragma clang fp contract(off)
        const d2 a = r[c], b = pold[c];
        const double bx = b.x * beta, by = b.y * beta;
        return make_double2(bx + a.x, by + a.y);
    }
    __device__ __forceinline__ d2 load_u(int c) const {
#pragma clang fp contract(off)
        const d2 a = ld_const(r + c), b = ld_const(pold + c);
        const double bx = b.x * beta, by = b.y * beta;
        return make_double2(bx + a.x, by + a.y);
    }
    // the two addends of element c, wave-uniform (scalar loads): a tile forms its row sums as A r + beta (A p_old) so that
    // the gathered elements stay in scalar registers (32 vector registers less: 4 instead of 3 wavefronts per SIMD)
    __device__ __forceinline__ d2 r_u(int c) const { return ld_const(r + c); }
    __device__ __forceinline__ d2 pold_u(int c) const { return ld_const(pold + c); }
};
struct Gather1 {
    static constexpr int NRHS = 1;
    static constexpr bool FUSED = false;
    static constexpr bool NT = true;
    const double* w;   // points at the chosen component of an interleaved vector: element c at w[2c]
    __device__ __forceinline__ d2 load(int c) const { return make_double2(w[2 * (int64_t)c], 0.0); }
    __device__ __forceinline__ d2 load_u(int c) const { return make_double2(*(cptr_f64)(w + 2 * (int64_t)c), 0.0); }
};
// value x gathered element
template <class G>
__device__ __forceinline__ d2 gprod(const G& g, double v, int c) {
    const d2 x = g.load(c);
    if constexpr (G::NRHS == 2) return make_double2(v * x.x, v * x.y);
    else return make_double2(v * x.x, 0.0);
}

constexpr int WPL = WNNZ / 64;      // stream entries per lane

// A row the sweep has summed: run its epilogue, or -- a DEFERRED row (dual tiles hold the rest of it) -- park the sum in
// the row's own partial slot for whoever adds the slot lists (epi.park: a CG sweep adds the row's share of Ap.p there).
template <bool DEFER, class Epi>
__device__ __forceinline__ void finish_row(const DevBlkCsr& S, Epi& epi, int row, double a1, double a2, const RowPre& pr) {
    if constexpr (DEFER) {
        const int ds = S.row_defer[row];
        if (ds >= 0) {
            reinterpret_cast<d2*>(S.slots)[ds] = make_double2(a1, a2);
            epi.park(row, a1, a2, pr.v);
            return;
        }
    }
    epi.row(row, a1, a2, pr);
}

// (tile_colsum8 -- the column sums of a dual tile -- lives in dev_common.hpp: the resident CG kernel uses it too)

// NR long run-rows over the same column range [c00, c00+cnt): each gathered element feeds NR matrix values
template <int NR, bool DEFER, class G, class Epi>
__device__ __forceinline__ void long_run_rows(const DevBlkCsr& S, const G& gat, Epi& epi, const double* __restrict__ val, int64_t stride,
                                              int cnt, int c00, int row0, int lane) {
    double a1[NR], a2[NR];
#pragma unroll
    for (int i = 0; i < NR; ++i) { a1[i] = 0.0; a2[i] = 0.0; }
    RowPre pr{};
    if (lane < NR) pr = epi.pre(row0 + lane);
    constexpr int U = (NR >= 4) ? 2 : (NR == 2 ? 4 : 8);
    int k = lane;
    for (; k + (U - 1) * 64 < cnt; k += U * 64) {
        double v[NR][U];
#pragma unroll
        for (int i = 0; i < NR; ++i)
#pragma unroll
            for (int u = 0; u < U; ++u) v[i][u] = mload<G::NT>(val + i * stride + k + 64 * u);
        d2 x[U];
#pragma unroll
        for (int u = 0; u < U; ++u) x[u] = gat.load(c00 + k + 64 * u);
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
            for (int i = 0; i < NR; ++i) {
                a1[i] += v[i][u] * x[u].x;
                if constexpr (G::NRHS == 2) a2[i] += v[i][u] * x[u].y;
            }
    }
    for (; k < cnt; k += 64) {
        const d2 x = gat.load(c00 + k);
#pragma unroll
        for (int i = 0; i < NR; ++i) {
            const double v = mload<G::NT>(val + i * stride + k);
            a1[i] += v * x.x;
            if constexpr (G::NRHS == 2) a2[i] += v * x.y;
        }
    }
#pragma unroll
    for (int i = 0; i < NR; ++i) {
        a1[i] = group_sum(a1[i], 64);
        if constexpr (G::NRHS == 2) a2[i] = group_sum(a2[i], 64);
    }
    // lane i finishes row i
#pragma unroll
    for (int i = 0; i < NR; ++i)
        if (lane == i) finish_row<DEFER>(S, epi, row0 + i, a1[i], a2[i], pr);
}

// U lane-major steps of an ELL block starting at step t: all U value (and index) loads are issued before the first
// use, so a wavefront keeps U x 512 B (+ indices, + gathers) in flight.
template <int U, bool RUN, class G>
__device__ __forceinline__ void ell_steps(const G& gat, const double* __restrict__ val, const int32_t* __restrict__ col,
                                          int t, int tpr, int lig, int len, int c0, double& a1, double& a2) {
    double v[U];
    int c[U];
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] = mload<G::NT>(val + 64 * (t + u));
    if constexpr (!RUN) {
#pragma unroll
        for (int u = 0; u < U; ++u) c[u] = mload<G::NT>(col + 64 * (t + u));
    }
    bool m[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const int e = (t + u) * tpr + lig;
        m[u] = e < len;
        if constexpr (RUN) c[u] = c0 + (m[u] ? e : 0);
    }
    d2 p[U];
#pragma unroll
    for (int u = 0; u < U; ++u) p[u] = gprod(gat, v[u], c[u]);
#pragma unroll
    for (int u = 0; u < U; ++u)
        if (m[u]) {
            a1 += p[u].x;
            if constexpr (G::NRHS == 2) a2 += p[u].y;
        }
}

// U strided steps of a long row starting at entry k (lane-consecutive, 64 entries per step)
template <int U, class G>
__device__ __forceinline__ void long_steps(const G& gat, const double* __restrict__ val, const int32_t* __restrict__ col,
                                           int k, double& a1, double& a2) {
    double v[U];
    int c[U];
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] = mload<G::NT>(val + k + 64 * u);
#pragma unroll
    for (int u = 0; u < U; ++u) c[u] = mload<G::NT>(col + k + 64 * u);
    d2 p[U];
#pragma unroll
    for (int u = 0; u < U; ++u) p[u] = gprod(gat, v[u], c[u]);
#pragma unroll
    for (int u = 0; u < U; ++u) {
        a1 += p[u].x;
        if constexpr (G::NRHS == 2) a2 += p[u].y;
    }
}

// wave-uniform descriptor through the constant address space (scalar loads)
typedef int v4i __attribute__((ext_vector_type(4)));
typedef const __attribute__((address_space(4))) v4i* cptr_v4i;
__device__ __forceinline__ BlkDesc ld_desc(const BlkDesc* p) {
    static_assert(sizeof(BlkDesc) == 48, "three 16-byte scalar loads");
    union { v4i q[3]; BlkDesc d; } u;
    const cptr_v4i src = (cptr_v4i)(p);
    u.q[0] = src[0]; u.q[1] = src[1]; u.q[2] = src[2];
    return u.d;
}

// the slice of row blocks a wavefront owns and its FIRST descriptor, both requested at once (WaveWork: a kernel asks for them
// before its scalar prologue, so that the prologue's round trip covers them too)
struct WaveWork { int b_lo, b_hi; BlkDesc first; };
__device__ __forceinline__ WaveWork wave_work(const DevBlkCsr& S) {
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wave = xcd_remap(blockIdx.x, S.nwg) * SPMV_WAVES + wv;
    WaveWork w;
    w.b_lo = S.wave_blk0[wave]; w.b_hi = S.wave_blk0[wave + 1];
    w.first = ld_desc(S.wave_first + wave);
    return w;
}

template <bool DEFER, class G, class Epi>
__device__ __forceinline__ void spmv_walk(const DevBlkCsr& S, const G& gat, Epi& epi, double* prod_all, const WaveWork& ww) {
    constexpr int NRHS = G::NRHS;
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    double* prod = prod_all + (size_t)wv * (WNNZ * NRHS);          // this wavefront's LDS slice
    const int b_lo = ww.b_lo, b_hi = ww.b_hi;
    BlkDesc dnext = ww.first;
    for (int b = b_lo; b < b_hi; ++b) {
        const BlkDesc d = dnext;
        if constexpr (G::FUSED) { if (b + 1 < b_hi) dnext = S.blk[b + 1]; }     // (the two-launch variant sits at its register limit: plain load)
        else { if (b + 1 < b_hi) dnext = ld_desc(S.blk + b + 1); }             // one block ahead: its latency hides behind this block's loads
        const int kind = d.kind();
        if (kind == BLK_LONG) {
            // ---------------- long row d.row0
            double a1 = 0.0, a2 = 0.0;
            RowPre pr{};
            if (lane == 0 && !d.run()) pr = epi.pre(d.row0);
            const double* __restrict__ val = S.val + d.nnz0;
            const int32_t* __restrict__ col = S.col + d.colpos;
            const int cnt = (int)d.cnt;           // a row of S has < 2^31 entries
            int k = lane;
            if (d.run()) {
                const int c00 = S.col[d.colpos];  // consecutive columns: col(e) = c00 + e
                const int nr = d.nrows();
                const int64_t stride = (d.cnt + NNZ_ALIGN - 1) / NNZ_ALIGN * NNZ_ALIGN;
                if (nr == 4) long_run_rows<4, DEFER>(S, gat, epi, val, stride, cnt, c00, d.row0, lane);
                else if (nr == 2) long_run_rows<2, DEFER>(S, gat, epi, val, stride, cnt, c00, d.row0, lane);
                else long_run_rows<1, DEFER>(S, gat, epi, val, stride, cnt, c00, d.row0, lane);
                continue;
            } else {
                for (; k + 7 * 64 < cnt; k += 8 * 64) long_steps<8>(gat, val, col, k, a1, a2);
                for (; k + 1 * 64 < cnt; k += 2 * 64) long_steps<2>(gat, val, col, k, a1, a2);
                for (; k < cnt; k += 64) long_steps<1>(gat, val, col, k, a1, a2);
            }
            a1 = group_sum(a1, 64);
            if constexpr (NRHS == 2) a2 = group_sum(a2, 64);
            if (lane == 0) finish_row<DEFER>(S, epi, d.row0, a1, a2, pr);
        } else if (kind == BLK_TILE) {
            // ---------------- dual tile: lane = row, step = column; row sums stay in the lanes, column sums go to slots
            if constexpr (DEFER) {
                // A TALL tile: K sub-tiles of 64 rows (the last: d.nrows()) below each other over the same tc columns.  Per
                // sub-tile the row sums stay in the lanes; the column sums of 8 steps at a time come out of the butterfly in
                // lanes 0..7 and are ADDED UP over the sub-tiles in the wavefront's LDS slice (same lane, same address: no
                // hazard), so that one slot per column leaves the tall tile -- K times fewer slots for whoever adds the lists.
                const int Rl = d.nrows(), T = d.steps(), K = d.tall();
                const int c0 = d.meta[0], cslot = d.meta[1], rslot = d.meta[2], tc = d.meta[3];      // (tc: real columns; steps beyond are padding)
                d2* __restrict__ slots = reinterpret_cast<d2*>(S.slots);
                d2* colacc = reinterpret_cast<d2*>(prod);
                for (int j = 0; j < K; ++j) {
                const int R = (j + 1 < K) ? 64 : Rl;
                const bool valid = lane < R;
                const int row = d.row0 + 64 * j + lane;
                RowPre pr{};
                if (valid && rslot < 0) pr = epi.pre(row);
                const d2 wr = valid ? gat.load(row) : make_double2(0.0, 0.0);
                const double* __restrict__ val = S.val + d.nnz0 + (int64_t)j * 64 * T + lane;
                double r1 = 0.0, r2 = 0.0;
                double vn[TILE_GROUP];                 // the next group's values are in flight while this group is reduced
#pragma unroll
                for (int u = 0; u < TILE_GROUP; ++u) vn[u] = mload<G::NT>(val + 64 * u);
                for (int t = 0; t < T; t += TILE_GROUP) {
                    double v[TILE_GROUP];
                    d2 x[TILE_GROUP];
#pragma unroll
                    for (int u = 0; u < TILE_GROUP; ++u) v[u] = vn[u];
                    if (t + TILE_GROUP < T) {
#pragma unroll
                        for (int u = 0; u < TILE_GROUP; ++u) vn[u] = mload<G::NT>(val + 64 * (t + TILE_GROUP + u));
                    }
                    double p1[TILE_GROUP], p2[TILE_GROUP];
                    if constexpr (G::FUSED) {
                        // row sums of p_new = r + beta p_old as (A r) + beta (A p_old): equal to rounding, operands stay scalar
                        double ga1 = 0.0, ga2 = 0.0, gb1 = 0.0, gb2 = 0.0;
#pragma unroll
                        for (int u = 0; u < TILE_GROUP; ++u) {
                            const d2 xa = (t + u < tc) ? gat.r_u(c0 + t + u) : make_double2(0.0, 0.0);
                            const d2 xb = (t + u < tc) ? gat.pold_u(c0 + t + u) : make_double2(0.0, 0.0);
                            ga1 += v[u] * xa.x; ga2 += v[u] * xa.y;
                            gb1 += v[u] * xb.x; gb2 += v[u] * xb.y;
                            p1[u] = v[u] * wr.x; p2[u] = v[u] * wr.y;
                        }
                        r1 += ga1 + gat.beta * gb1;
                        r2 += ga2 + gat.beta * gb2;
                    } else {
#pragma unroll
                        for (int u = 0; u < TILE_GROUP; ++u) x[u] = (t + u < tc) ? gat.load_u(c0 + t + u) : make_double2(0.0, 0.0);
#pragma unroll
                        for (int u = 0; u < TILE_GROUP; ++u) {
                            r1 += v[u] * x[u].x;
                            p1[u] = v[u] * wr.x;
                            if constexpr (NRHS == 2) { r2 += v[u] * x[u].y; p2[u] = v[u] * wr.y; }
                        }
                    }
                    double s1, s2 = 0.0;
                    if (S.dbg_flags & 1) { s1 = p1[0]; if constexpr (NRHS == 2) s2 = p2[0]; }       // (timing experiment: no butterflies)
                    else {
                        s1 = tile_colsum8(p1, lane);
                        if constexpr (NRHS == 2) s2 = tile_colsum8(p2, lane);
                    }
                    if (lane < TILE_GROUP) {
                        if (j > 0) { const d2 o = colacc[t + lane]; s1 += o.x; s2 += o.y; }
                        colacc[t + lane] = make_double2(s1, s2);
                    }
                }
                if (valid) {
                    if (rslot < 0) epi.row(row, r1, r2, pr);
                    else { slots[rslot + 64 * j + lane] = make_double2(r1, r2); epi.park(row, r1, r2, wr); }
                }
                }
                // the tall tile's column sums: lane = column
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                if (lane < T) {
                    const d2 sc = colacc[lane];
                    slots[cslot + lane] = sc;
                    if constexpr (Epi::FOLDDEF) { if (lane < tc) epi.park(c0 + lane, sc.x, sc.y, gat.load(c0 + lane)); }
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
            }
        } else if (kind == BLK_ELL) {
            // ---------------- lane-major block: lane (row, lig) owns entries lig, lig + tpr, ... of its row
            const int R = d.nrows(), T = d.steps();
            const int p2 = (R <= 1) ? 1 : (1 << (32 - __clz(R - 1)));
            const int tpr = 64 / p2;
            const int sh = 31 - __clz(tpr);
            const int row = lane >> sh, lig = lane & (tpr - 1);
            const int len = (row < R) ? (int)S.row_rel[d.row0 + row] : 0;
            const bool owner = row < R && lig == 0;
            RowPre pr{};
            if (owner) pr = epi.pre(d.row0 + row);
            const double* __restrict__ val = S.val + d.nnz0 + lane;
            double a1 = 0.0, a2 = 0.0;
            int t = 0;
            if (d.run()) {
                // consecutive columns: entry e = t*tpr + lig of this lane's row sits in column c0 + e; lanes past their
                // row's length (and padding lanes) read column c0 (valid) and are masked
                const int c0 = (row < R) ? S.col[d.colpos + row] : 0;
                for (; t + 8 <= T; t += 8) ell_steps<8, true>(gat, val, nullptr, t, tpr, lig, len, c0, a1, a2);
                for (; t + 2 <= T; t += 2) ell_steps<2, true>(gat, val, nullptr, t, tpr, lig, len, c0, a1, a2);
                for (; t < T; ++t) ell_steps<1, true>(gat, val, nullptr, t, tpr, lig, len, c0, a1, a2);
            } else {
                const int32_t* __restrict__ col = S.col + d.colpos + lane;
                for (; t + 8 <= T; t += 8) ell_steps<8, false>(gat, val, col, t, tpr, lig, len, 0, a1, a2);
                for (; t + 2 <= T; t += 2) ell_steps<2, false>(gat, val, col, t, tpr, lig, len, 0, a1, a2);
                for (; t < T; ++t) ell_steps<1, false>(gat, val, col, t, tpr, lig, len, 0, a1, a2);
            }
            a1 = group_sum(a1, tpr);
            if constexpr (NRHS == 2) a2 = group_sum(a2, tpr);
            if (owner) finish_row<DEFER>(S, epi, d.row0 + row, a1, a2, pr);
        } else {
            // ---------------- LDS-staged block: rows row0 .. row0+nrows-1, cnt entries in CSR order
            const int cnt = (int)d.cnt;
            const double* __restrict__ val = S.val + d.nnz0;
            const int32_t* __restrict__ col = S.col + d.colpos;
            const int R = d.nrows();
            const int p2 = (R <= 1) ? 1 : (1 << (32 - __clz(R - 1)));
            const int tpr = 64 / p2;
            const int sh = 31 - __clz(tpr);
            const int row = lane >> sh, lig = lane & (tpr - 1);
            const bool owner = row < R && lig == 0;
            RowPre pr{};
            int s0 = 0, e = 0;
            if (row < R) {
                s0 = S.row_rel[d.row0 + row];
                e = (row + 1 < R) ? (int)S.row_rel[d.row0 + row + 1] : cnt;
            }
            if (owner) pr = epi.pre(d.row0 + row);
            {
                double v[WPL]; int c[WPL];
#pragma unroll
                for (int j = 0; j < WPL; ++j) {
                    const int k = lane + j * 64;
                    const bool ok = k < cnt;
                    v[j] = ok ? mload<G::NT>(val + k) : 0.0;
                    c[j] = ok ? mload<G::NT>(col + k) : 0;
                }
#pragma unroll
                for (int j = 0; j < WPL; ++j) {
                    const int k = lane + j * 64;
                    if (k < cnt) {
                        const d2 pv = gprod(gat, v[j], c[j]);
                        if constexpr (NRHS == 2) reinterpret_cast<d2*>(prod)[k] = pv;
                        else prod[k] = pv.x;
                    }
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            double a1 = 0.0, a2 = 0.0;
            if (row < R) {
                for (int k = s0 + lig; k < e; k += tpr) {
                    if constexpr (NRHS == 2) { const d2 p = reinterpret_cast<const d2*>(prod)[k]; a1 += p.x; a2 += p.y; }
                    else { a1 += prod[k]; }
                }
            }
            a1 = group_sum(a1, tpr);
            if constexpr (NRHS == 2) a2 = group_sum(a2, tpr);
            if (owner) finish_row<DEFER>(S, epi, d.row0 + row, a1, a2, pr);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
        }
    }
}

// ------------------------------------------------------------------------------------------------ window panels
// Gather-bound operators (fos_internal.hpp, WinPanel): ONE WORKGROUP per panel of GEO::ROWS rows.  Per column window the panel
// touches: stage the window of the vector in LDS (coalesced 16-byte loads), then every wavefront takes whole 64-row slices --
// lane = row, lane-major values and 16-bit window offsets streamed non-temporally, the vector element read from LDS -- and adds
// its lanes' sums to the panel's row sums in LDS (a row appears once per window: no conflict, fixed order).  The row epilogue
// runs once per row at the end, over consecutive rows (coalesced).
template <class GEO> constexpr size_t win_lds_bytes(int nrhs) { return (size_t)(GEO::COLS + GEO::ROWS) * 8 * nrhs + 64 * sizeof(double); }

// Start / end stamps of every workgroup of the row-block sweep (timing experiments; -DFOS_KKT_STAMPS; tools/kkt_stamps.py)
#ifdef FOS_KKT_STAMPS
__device__ long long g_kkt_stamps[2 * 16384];
#define KKT_STAMP(which) do { if (threadIdx.x == 0 && blockIdx.x < 16384) { __builtin_amdgcn_sched_barrier(0); g_kkt_stamps[2 * blockIdx.x + (which)] = wall_clock64(); __builtin_amdgcn_sched_barrier(0); } } while (0)
#else
#define KKT_STAMP(which) do { } while (0)
#endif

// In-kernel time stamps of the window walk (timing experiments; compiled in with -DFOS_WIN_STAMPS): workgroup FOS_WIN_STAMP_WG, every
// wavefront, segment k, phase ph -> g_win_stamps[(wave * 64 + k) * 8 + ph], in ticks of the 100 MHz clock.
#ifdef FOS_WIN_STAMPS
__device__ long long g_win_stamps[16 * 64 * 8];
#define WIN_STAMP(ph) do { if (stamp_on && k0 + k < 64) { __builtin_amdgcn_sched_barrier(0); g_win_stamps[((size_t)wv * 64 + k0 + k) * 8 + (ph)] = wall_clock64(); __builtin_amdgcn_sched_barrier(0); } } while (0)
#define WIN_STAMP_G(ph) do { if (blockIdx.x == 100 && (threadIdx.x & 63) == 0) { __builtin_amdgcn_sched_barrier(0); g_win_stamps[((size_t)(threadIdx.x >> 6) * 64 + 63) * 8 + (ph)] = wall_clock64(); __builtin_amdgcn_sched_barrier(0); } } while (0)
#else
#define WIN_STAMP(ph) do { } while (0)
#define WIN_STAMP_G(ph) do { } while (0)
#endif

// What a wavefront holds of one of its slices while the window is being staged: the first PRE steps of its lane's values and window
// offsets, requested BEFORE the barriers of the segment so that one memory latency covers the window and the matrix.
template <int PRE>
struct WinSliceRegs {
    double v[PRE > 0 ? PRE : 1];
    unsigned c[PRE > 0 ? PRE : 1];
    unsigned rid;          // local row of this lane, 0xFFFF: none
    int T;                 // steps of the slice, wave-uniform (0: no slice)
    int64_t off;
};
// The value / offset loads depend only on the (wave-uniform) step count and running offset, NOT on the per-lane row word (a second dependent
// memory round trip per segment costs more than the padding it would avoid reading: per-lane-count predication measured
// 210-260 us per C5 sweep).  Steps beyond a lane's count hold value 0 / offset 0.
template <bool MNT, int PRE>
__device__ __forceinline__ void win_slice_issue(const DevBlkCsr& S, int sl, int lane, WinSliceRegs<PRE>& r) {
    r.rid = 0xFFFFu;
#pragma unroll
    for (int t = 0; t < PRE; ++t) { r.v[t] = 0.0; r.c[t] = 0u; }
    if (r.T > 0) {
        r.rid = S.wrow[(size_t)sl * 64 + lane];
        const double* __restrict__ val = S.wval + r.off + lane;
        const uint16_t* __restrict__ col = S.wcol + r.off + lane;
#pragma unroll
        for (int t = 0; t < PRE; ++t)
            if (t < r.T) { r.v[t] = mload<MNT>(val + 64 * t); r.c[t] = mload<MNT>(col + 64 * t); }
    }
}
template <bool MNT, int NRHS, int PRE, class E>
__device__ __forceinline__ void win_slice_compute(const DevBlkCsr& S, const WinSliceRegs<PRE>& r, const E* __restrict__ win, E* __restrict__ acc, int lane) {
    if (r.T <= 0) return;
    double a1 = 0.0, a2 = 0.0;
    // (steps beyond T hold value 0, offset 0: the LDS reads of a group of steps go out together; a slot of four or more steps skips
    //  the groups of two its slice does not reach -- wave-uniform)
    constexpr int GRP = PRE >= 4 ? 2 : (PRE > 0 ? PRE : 1);
#pragma unroll
    for (int g = 0; g < PRE; g += GRP) {
        if (g == 0 || g < r.T) {
#pragma unroll
            for (int t = g; t < g + GRP && t < PRE; ++t) {
                const E x = win[r.c[t]];
                if constexpr (NRHS == 2) { a1 += r.v[t] * x.x; a2 += r.v[t] * x.y; } else a1 += r.v[t] * x;
            }
        }
    }
    if (r.T > PRE) {                                       // more steps than the slot holds (a row with many entries inside one window; PRE == 0: a slice without a register slot): streamed here
        const double* __restrict__ val = S.wval + r.off + lane;
        const uint16_t* __restrict__ col = S.wcol + r.off + lane;
        for (int t = PRE; t < r.T; ++t) {
            const double v = mload<MNT>(val + 64 * t);
            const E x = win[mload<MNT>(col + 64 * t)];
            if constexpr (NRHS == 2) { a1 += v * x.x; a2 += v * x.y; } else a1 += v * x;
        }
    }
    if (r.rid != 0xFFFFu) {
        if constexpr (NRHS == 2) { const d2 o = acc[r.rid]; acc[r.rid] = make_double2(o.x + a1, o.y + a2); }
        else acc[r.rid] += a1;
    }
}

// A panel holds EVERY row of its range, also the rows of a row-sharded operator that have no local entries (row_defer -2: the
// deferred-row kernel finishes them from the other ranks' shares): those are skipped here, the others go the way of finish_row.
template <bool DEFER, class Epi>
__device__ __forceinline__ void win_finish_row(const DevBlkCsr& S, Epi& epi, int row, double a1, double a2, const RowPre& pr) {
    if constexpr (DEFER) { if (S.row_defer[row] == -2) return; }
    finish_row<DEFER>(S, epi, row, a1, a2, pr);
}

// One segment = one (panel, window) tile.  EVERYTHING the segment needs from memory -- the window's vector elements and the
// first GEO::PRE0..3 steps of the slices of the wavefront -- is requested before the first barrier, and (round 4: wave streams,
// fos_internal.hpp) NOTHING of it waits for a descriptor: the panel's records sit in a register (lane k = segment k, one request
// per panel), the addresses are running offsets.  A segment costs one memory latency, two barriers and the LDS work.
// In-kernel stamps (round 4, tools/win_stamps.py, C5: 21-23 segments per panel, 4.0 us each): ISSUING the segment's ~18 loads per
// wavefront takes 1.6-2.2 us -- the wavefronts stall in the issue of their memory instructions --, the first barrier 0.1-1.0 (the
// wavefronts that issued first wait for the last), the wait for the data and the window's LDS stores 0.4-0.5, the second barrier
// 0.1-0.2, the multiply out of LDS and the row sums 0.7-1.0.  ONE workgroup per CU runs a segment in 3.0 us, two in 4.0 each.
// What was measured on this walk (per C5 sweep, MI355X): row blocks 155 us; 1 x 1024 threads, 2048 x 4096 tiles 112-116 us;
// 2 x 512 threads, 2016 x 3072 tiles 93-97 us (kept).  Slower or equal: 4096 x 4096 tiles with 1024 threads 139-160; 2048 x 2048
// tiles with a register-double-buffered pipeline across segments 136-145 (twice the barriers); per-lane-count predicated loads
// 210-260; next segment's descriptors requested one segment ahead by scalar loads 111.6, by vector loads 127; round 4: NO
// descriptor loads per segment at all (this form) 96.9 against 97.4 -- the two scalar round trips were not on the critical path;
// a body with one nest of wave-uniform step tests for the three slices, unpredicated whole-window loads / stores and a dummy
// row-sum element for padding lanes (a third fewer scalar instructions) 99-102: its LDS reads wait step by step; the same with
// per-step tests only 119; the matrix stream by ordinary loads (to keep it in the Infinity Cache) 102.6 against 101.2 and 51.8
// against 54.3 FISTA it/s; the second workgroup of every CU started 1-4 us late (out of phase) 101-103 against 99; the slice loads issued in front of the window loads 98-99 against 95-96.
template <class GEO, bool MNT, bool DEFER, class G, class Epi>
__device__ __forceinline__ void win_walk(const DevBlkCsr& S, const G& gat, Epi& epi, double* lds) {
    constexpr int NRHS = G::NRHS;
    constexpr int NWAVES = GEO::WAVES, WIN_COLS = GEO::COLS, WIN_THREADS = GEO::THREADS;
    constexpr int WPT = (WIN_COLS + WIN_THREADS - 1) / WIN_THREADS;      // window elements per thread
    using E = typename std::conditional<NRHS == 2, d2, double>::type;
    E* win = reinterpret_cast<E*>(lds);
    E* acc = win + WIN_COLS;
    const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
#ifdef FOS_WIN_STAMPS
    const bool stamp_on = blockIdx.x == 100 && lane == 0;
#endif
    for (int p = xcd_remap(blockIdx.x, gridDim.x); p < S.npanel; p += gridDim.x) {
        const WinPanel wp = S.wpanel[p];
        const WinWave ww = S.wwave[(size_t)p * NWAVES + wv];
        int64_t off = ww.off;                              // running position of this wavefront in the value / column streams
        int rs = ww.slice0;                                // ... and in the row words
        const WinDesc* __restrict__ drec = S.wdesc + wp.seg0 + (size_t)wv * wp.nseg;
        for (int i = tid; i < wp.nrows; i += WIN_THREADS) {
            if constexpr (NRHS == 2) acc[i] = make_double2(0.0, 0.0); else acc[i] = 0.0;
        }
        WIN_STAMP_G(1);
        if constexpr (GEO::PIPELINE) {
            // SOFTWARE PIPELINE over the segments, slot by slot (round 5): as soon as slot u of segment k has been multiplied, its registers take the requests of
            // slot u of segment k + 1 (and segment k + 1's window is requested behind the second barrier, as with PREFETCH) -- so the memory system works on
            // segment k + 1 WHILE segment k is multiplied out of LDS, with the registers of ONE segment.  A segment then costs max(memory, LDS work) + two barriers
            // instead of their sum.
            for (int k0 = 0; k0 < wp.nseg; k0 += 64) {
                const int nk = min(64, wp.nseg - k0);
                uint4 dv = make_uint4(0u, 0u, 0u, 0u);
                if (lane < nk) dv = *reinterpret_cast<const uint4*>(drec + k0 + lane);
                d2 wreg[WPT];
                WinSliceRegs<GEO::PRE0> r0; WinSliceRegs<GEO::PRE1> r1; WinSliceRegs<GEO::PRE2> r2; WinSliceRegs<GEO::PRE3> r3;
                int ncols;
                {   // prologue: everything of the chunk's first segment
                    const int col0 = __builtin_amdgcn_readlane((int)dv.x, 0);
                    const unsigned t01 = (unsigned)__builtin_amdgcn_readlane((int)dv.y, 0), t23 = (unsigned)__builtin_amdgcn_readlane((int)dv.z, 0);
                    ncols = __builtin_amdgcn_readlane((int)dv.w, 0);
                    r0.T = (int)(t01 & 0xFFFFu); r1.T = (int)(t01 >> 16); r2.T = (int)(t23 & 0xFFFFu); r3.T = (int)(t23 >> 16);
                    r0.off = off; off += 64 * (int64_t)r0.T;
                    r1.off = off; off += 64 * (int64_t)r1.T;
                    r2.off = off; off += 64 * (int64_t)r2.T;
                    r3.off = off; off += 64 * (int64_t)r3.T;
#pragma unroll
                    for (int q = 0; q < WPT; ++q) {
                        const int i = tid + q * WIN_THREADS;
                        wreg[q] = (i < ncols) ? gat.load(col0 + i) : make_double2(0.0, 0.0);
                    }
                    win_slice_issue<MNT>(S, rs, lane, r0); rs += r0.T > 0;
                    win_slice_issue<MNT>(S, rs, lane, r1); rs += r1.T > 0;
                    win_slice_issue<MNT>(S, rs, lane, r2); rs += r2.T > 0;
                    win_slice_issue<MNT>(S, rs, lane, r3); rs += r3.T > 0;
                }
                for (int k = 0; k < nk; ++k) {
                    WIN_STAMP(0);
                    // EVERYTHING requested so far is this segment's: one explicit wait for all of it.  (Left to the compiler, the waits sit inside the predicated
                    // window stores, so on its books the slots' registers may still be pending at their multiply -- and because the requests issued in between
                    // are predicated too, i.e. of unknown number, it can only wait for ALL of them there: vmcnt(0) in front of every slot, which drains the very
                    // requests this pipeline wants in flight.)
                    __builtin_amdgcn_s_waitcnt(0x0F70);    // vmcnt(0), expcnt / lgkmcnt untouched
                    WIN_STAMP(1);
                    __syncthreads();                       // the previous window's readers are done (first pass: acc is zeroed)
                    WIN_STAMP(2);
#pragma unroll
                    for (int q = 0; q < WPT; ++q) {
                        const int i = tid + q * WIN_THREADS;
                        if (i < ncols) { if constexpr (NRHS == 2) win[i] = wreg[q]; else win[i] = wreg[q].x; }
                    }
                    WIN_STAMP(3);
                    __syncthreads();
                    WIN_STAMP(4);
                    const bool nxt = k + 1 < nk;
                    unsigned t01n = 0u, t23n = 0u;
                    if (nxt) {
                        const int col0n = __builtin_amdgcn_readlane((int)dv.x, k + 1);
                        t01n = (unsigned)__builtin_amdgcn_readlane((int)dv.y, k + 1);
                        t23n = (unsigned)__builtin_amdgcn_readlane((int)dv.z, k + 1);
                        ncols = __builtin_amdgcn_readlane((int)dv.w, k + 1);
#pragma unroll
                        for (int q = 0; q < WPT; ++q) {
                            const int i = tid + q * WIN_THREADS;
                            wreg[q] = (i < ncols) ? gat.load(col0n + i) : make_double2(0.0, 0.0);
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);     // (the window's requests stay in front of the multiply)
                    win_slice_compute<MNT, NRHS>(S, r0, win, acc, lane);
                    __builtin_amdgcn_sched_barrier(0);
                    r0.T = (int)(t01n & 0xFFFFu); r0.off = off; off += 64 * (int64_t)r0.T;
                    win_slice_issue<MNT>(S, rs, lane, r0); rs += r0.T > 0;
                    __builtin_amdgcn_sched_barrier(0);
                    win_slice_compute<MNT, NRHS>(S, r1, win, acc, lane);
                    __builtin_amdgcn_sched_barrier(0);
                    r1.T = (int)(t01n >> 16); r1.off = off; off += 64 * (int64_t)r1.T;
                    win_slice_issue<MNT>(S, rs, lane, r1); rs += r1.T > 0;
                    __builtin_amdgcn_sched_barrier(0);
                    win_slice_compute<MNT, NRHS>(S, r2, win, acc, lane);
                    __builtin_amdgcn_sched_barrier(0);
                    r2.T = (int)(t23n & 0xFFFFu); r2.off = off; off += 64 * (int64_t)r2.T;
                    win_slice_issue<MNT>(S, rs, lane, r2); rs += r2.T > 0;
                    __builtin_amdgcn_sched_barrier(0);
                    win_slice_compute<MNT, NRHS>(S, r3, win, acc, lane);
                    __builtin_amdgcn_sched_barrier(0);
                    r3.T = (int)(t23n >> 16); r3.off = off; off += 64 * (int64_t)r3.T;
                    win_slice_issue<MNT>(S, rs, lane, r3); rs += r3.T > 0;
                    WIN_STAMP(5);
                }
            }
        } else
        for (int k0 = 0; k0 < wp.nseg; k0 += 64) {         // (one chunk unless a panel touches more than 64 windows)
            const int nk = min(64, wp.nseg - k0);
            uint4 dv = make_uint4(0u, 0u, 0u, 0u);
            if (lane < nk) dv = *reinterpret_cast<const uint4*>(drec + k0 + lane);
            d2 wreg[WPT];
            bool have_w = false;                           // PREFETCH: wreg already holds this segment's window (requested behind the previous segment's barriers)
            for (int k = 0; k < nk; ++k) {
                WIN_STAMP(0);
                const int col0 = __builtin_amdgcn_readlane((int)dv.x, k);
                const unsigned t01 = (unsigned)__builtin_amdgcn_readlane((int)dv.y, k);
                const unsigned t23 = (unsigned)__builtin_amdgcn_readlane((int)dv.z, k);
                const int ncols = __builtin_amdgcn_readlane((int)dv.w, k);
                // ---- everything this segment needs from memory is requested here, in one go
                WinSliceRegs<GEO::PRE0> r0; WinSliceRegs<GEO::PRE1> r1; WinSliceRegs<GEO::PRE2> r2; WinSliceRegs<GEO::PRE3> r3;
                r0.T = (int)(t01 & 0xFFFFu); r1.T = (int)(t01 >> 16); r2.T = (int)(t23 & 0xFFFFu); r3.T = (int)(t23 >> 16);
                r0.off = off; off += 64 * (int64_t)r0.T;
                r1.off = off; off += 64 * (int64_t)r1.T;
                r2.off = off; off += 64 * (int64_t)r2.T;
                r3.off = off; off += 64 * (int64_t)r3.T;
                if (!(GEO::PREFETCH && have_w)) {
#pragma unroll
                    for (int q = 0; q < WPT; ++q) {
                        const int i = tid + q * WIN_THREADS;
                        wreg[q] = (i < ncols) ? gat.load(col0 + i) : make_double2(0.0, 0.0);
                    }
                }
                win_slice_issue<MNT>(S, rs, lane, r0);
                win_slice_issue<MNT>(S, rs + 1, lane, r1);
                win_slice_issue<MNT>(S, rs + 2, lane, r2);
                win_slice_issue<MNT>(S, rs + 3, lane, r3);
                rs += (r0.T > 0) + (r1.T > 0) + (r2.T > 0) + (r3.T > 0);
                WIN_STAMP(1);
                __syncthreads();                           // the previous window's readers are done (first pass: acc is zeroed)
                WIN_STAMP(2);
#pragma unroll
                for (int q = 0; q < WPT; ++q) {
                    const int i = tid + q * WIN_THREADS;
                    if (i < ncols) { if constexpr (NRHS == 2) win[i] = wreg[q]; else win[i] = wreg[q].x; }
                }
                WIN_STAMP(3);
                __syncthreads();
                WIN_STAMP(4);
                if constexpr (GEO::PREFETCH) {
                    // the NEXT segment's window is requested here: its registers are free (the window is in LDS), and the request's issue and
                    // latency pass beside this segment's multiply instead of in front of the next segment's barriers
                    have_w = k + 1 < nk;
                    if (have_w) {
                        const int col0n = __builtin_amdgcn_readlane((int)dv.x, k + 1);
                        const int ncolsn = __builtin_amdgcn_readlane((int)dv.w, k + 1);
#pragma unroll
                        for (int q = 0; q < WPT; ++q) {
                            const int i = tid + q * WIN_THREADS;
                            wreg[q] = (i < ncolsn) ? gat.load(col0n + i) : make_double2(0.0, 0.0);
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);     // (the requests stay in front of the multiply)
                }
                win_slice_compute<MNT, NRHS>(S, r0, win, acc, lane);
                win_slice_compute<MNT, NRHS>(S, r1, win, acc, lane);
                win_slice_compute<MNT, NRHS>(S, r2, win, acc, lane);
                win_slice_compute<MNT, NRHS>(S, r3, win, acc, lane);
                WIN_STAMP(5);
            }
        }
        WIN_STAMP_G(2);
        // row epilogue, two rows of a thread at a time (all four at once spills): what the FIRST two need from memory besides their sums is requested here, in front of
        // the barrier that ends the walk, so that its latency passes beside the barrier instead of behind it; the second pair's while the first is finished
        RowPre pa{}, pb{};
        {
            const int i = tid, i2 = tid + WIN_THREADS;
            if (i < wp.nrows) pa = epi.pre(wp.row0 + i);
            if (i2 < wp.nrows) pb = epi.pre(wp.row0 + i2);
        }
        __syncthreads();
        WIN_STAMP_G(3);
        for (int i = tid; i < wp.nrows; i += 2 * WIN_THREADS) {
            const int i2 = i + WIN_THREADS, i3 = i + 2 * WIN_THREADS, i4 = i + 3 * WIN_THREADS;
            const bool h2 = i2 < wp.nrows;
            RowPre na{}, nb{};                                 // the next pair
            if (i3 < wp.nrows) na = epi.pre(wp.row0 + i3);
            if (i4 < wp.nrows) nb = epi.pre(wp.row0 + i4);
            if constexpr (NRHS == 2) { const d2 a = acc[i]; win_finish_row<DEFER>(S, epi, wp.row0 + i, a.x, a.y, pa); }
            else win_finish_row<DEFER>(S, epi, wp.row0 + i, acc[i], 0.0, pa);
            if (h2) {
                if constexpr (NRHS == 2) { const d2 a = acc[i2]; win_finish_row<DEFER>(S, epi, wp.row0 + i2, a.x, a.y, pb); }
                else win_finish_row<DEFER>(S, epi, wp.row0 + i2, acc[i2], 0.0, pb);
            }
            pa = na; pb = nb;
        }
        WIN_STAMP_G(4);
        if (p + (int)gridDim.x < S.npanel) __syncthreads();    // before the next panel zeroes acc (wave-uniform: p is the workgroup's)
    }
}

// ------------------------------------------------------------------------------------------------ KKT apply, 2 RHS

// Row epilogue of out = [I Q'; Q -I] w for both right-hand sides, plus the three sums CG needs.
//   FOLD (CG sweeps over an operator with dual tiles): rows whose sum is spread over partial slots are finished by the NEXT
//   kernel (cg_update_kernel adds the slot lists while it updates x and r), yet alpha = rn / (Ap.p) is needed before that.
//   Ap.p is bilinear in the partial sums: for a deferred row i with slots s_k, p = p_i, c = [c;b]_i, sg = +1 (i < n), -1 (else)
//       (Ap)_i . p_i = (p.x^2 - p.y^2) + c (wt.x p.y - wt.y p.x) + sg * sum_k (s_k.x p.y - s_k.y p.x)
//   so whoever WRITES a slot adds its term (park), and the slot-free part is added once per deferred row by kkt_deferred_local.
template <class G, bool FOLD>
struct EpiKkt {
    struct AccKeep { double v[3]; };
    __device__ __forceinline__ AccKeep acc_copy() const { AccKeep k; for (int a = 0; a < 3; ++a) k.v[a] = acc[a]; return k; }
    __device__ __forceinline__ void acc_restore(const AccKeep& k) { for (int a = 0; a < 3; ++a) acc[a] = k.v[a]; }
    static constexpr bool FOLDDEF = FOLD;
    G gat;
    d2* out;
    d2* pnew;          // GatherP: where the row's owner stores p_new[i]
    const double* cb;
    int n;
    d2 wt;             // (p1_tau, p2_tau)
    double acc[3];     // S1 = sum Ap.p (non-tau rows), T1 = [c;b].p1, T2 = [c;b].p2
    __device__ __forceinline__ RowPre pre(int i) const { return RowPre{gat.load(i), cb[i]}; }
    __device__ __forceinline__ void row(int i, double u1, double u2, const RowPre& pr) {
        const d2 p = pr.v;
        const double c = pr.c;
        double q1, q2;                                  // (Q p1)_i, (Q p2)_i   HSDEAffine.jl:51-56
        if (i < n) { q1 = u1 + wt.x * c; q2 = u2 + wt.y * c; }
        else { q1 = -(u1 - wt.x * c); q2 = -(u2 - wt.y * c); }
        const double a1 = p.x - q2;                     // y1 = Q'x2 + x1 = -(Q x2) + x1     affinepluslinear.jl:45-46
        const double a2 = q1 - p.y;                     // y2 = Q x1 - x2                    affinepluslinear.jl:47-48
        out[i] = make_double2(a1, a2);
        if constexpr (G::FUSED) pnew[i] = p;
        acc[0] += a1 * p.x + a2 * p.y;
        acc[1] += c * p.x;
        acc[2] += c * p.y;
    }
    // a partial sum (s1, s2) of row i went to a slot; p = the row's vector element
    __device__ __forceinline__ void park(int i, double s1, double s2, const d2& p) {
        if constexpr (FOLD) {
            const double t = s1 * p.y - s2 * p.x;
            acc[0] += (i < n) ? t : -t;
        }
    }
    // the slot-free part of a deferred row (and its p_new)
    __device__ __forceinline__ void deferred_local(int i) {
        const d2 p = gat.load(i);
        const double c = cb[i];
        if constexpr (G::FUSED) pnew[i] = p;
        acc[0] += (p.x * p.x - p.y * p.y) + c * (wt.x * p.y - wt.y * p.x);
        acc[1] += c * p.x;
        acc[2] += c * p.y;
    }
};

// Fused dual-RHS KKT sweep.  Template switches:
//   DEFER  the operator has dual tiles (rows spread over partial slots);
//   FUSEP  CG iteration j >= 2: the kernel first CLOSES iteration j-1 (cg_close_iteration: r.r from the update kernel's
//          partials, stop test, beta -- every workgroup, same order), then sweeps with p_j = r + beta p_{j-1} formed on the fly
//          (GatherP) and stores p_j through the rows' owners: conjugategradients.jl:42-50 without a launch of its own;
//   FOLD   the slot-spread rows' share of Ap.p is added here (EpiKkt), so no deferred-row kernel follows in a CG iteration.
struct KktArgs {
    const d2* w;               // !FUSEP: the vector to apply to (p_j);  FUSEP: p_{j-1}
    const d2* r;               // FUSEP
    d2* pnew;                  // FUSEP: p_j
    d2* out;
    const double* cb;
    int n, nm;
    double* partials;          // 3 doubles per workgroup
    DevState* st;
    int gate;                  // !FUSEP: skip when st->done
    const double* rr_partials; // FUSEP: the update kernel's r.r records
    int rr_count;
    const double* reduced;     // FUSEP + RCCL: all-reduced r.r
    int from_reduced;
    int j;                     // FUSEP: this iteration (>= 2)
    PeerBox pb;                // FUSEP: nranks > 0 -> the r.r exchange happens here (peer mailboxes)
    uint32_t seq_base;
    int count_repl;            // FOLD: 0 = the slot-free part of the replicated slot-spread rows is counted by another rank (row-sharded)
    int n_repl;                // rows below it are replicated (row-sharded: the n rows of A'; else 0)
    // merged-reduction CG (fos_internal.hpp, CgmIter): the sweep applies M to r
    double* vt_out;            // non-null: workgroup 0 stashes the applied vector's tau element here (DevState.vtau)
    int close_j;               // >= 0: close iteration close_j first (r.r records of its update, stop test) and return when CG has stopped (plain_args: -1)
    int32_t batch_mark;
};

template <bool DEFER, bool FUSEP, bool FOLD, bool NT = true>
__global__ __launch_bounds__(SPMV_THREADS, FUSEP ? 4 : 1) void kkt2_kernel(DevBlkCsr S, KktArgs a) {
    static_assert(NT || !FUSEP, "the resident-operator form exists for the plain sweeps only");
    static_assert(!FOLD || DEFER, "FOLD is about deferred rows");
    KKT_STAMP(0);
    WaveWork ww;                             // requested before the gate / the closing prologue: their round trip covers it
    if constexpr (!FUSEP) ww = wave_work(S);
    if ((FUSEP || a.gate) && a.close_j < 0 && a.st->done) return;        // (close_j > 0: cgm_close_in_sweep below tests `done` with its other loads)
    __shared__ __attribute__((aligned(16))) double prod[SPMV_WAVES * WNNZ * 2];
    __shared__ double red[16];
    using G = typename std::conditional<FUSEP, GatherP, GatherWT<NT>>::type;
    G gat;
    if constexpr (FUSEP) {
        if (a.pb.nranks > 0 && a.st->xchg_failed) return;
        const CgClose cl = cg_close_iteration(a.st, a.rr_partials, a.rr_count, a.reduced, a.from_reduced, a.r, (int64_t)a.nm + 1, a.j - 1, a.pb, a.seq_base);
        if (!cl.ok || cl.stop) return;
        gat.r = a.r; gat.pold = a.w; gat.beta = cl.beta;
        ww = wave_work(S);
    } else {
        gat.w = a.w;
        if (a.close_j >= 0 && cgm_close_in_sweep(a.st, a.rr_partials, a.rr_count, a.w, (int64_t)a.nm + 1, a.close_j, a.seq_base >> 11, a.batch_mark)) return;
    }
    EpiKkt<G, FOLD> epi;
    epi.gat = gat; epi.out = a.out; epi.pnew = a.pnew; epi.cb = a.cb; epi.n = a.n; epi.wt = gat.load_u(a.nm);
    epi.acc[0] = epi.acc[1] = epi.acc[2] = 0.0;
    if (a.vt_out && blockIdx.x == 0 && threadIdx.x == 0) { a.vt_out[0] = epi.wt.x; a.vt_out[1] = epi.wt.y; }
    if constexpr (FUSEP) {
        if (blockIdx.x == 0 && threadIdx.x == 0) a.pnew[a.nm] = epi.wt;          // the tau element has no row in S
    }
    if constexpr (FOLD) {
        if (a.count_repl || S.ndef > a.n_repl) {          // (row-sharded: the replicated rows' slot-free part is counted by one rank)
            for (int q = blockIdx.x * SPMV_THREADS + threadIdx.x; q < S.ndef; q += gridDim.x * SPMV_THREADS) {
                const int i = S.def_rows[q];
                if (a.count_repl || i >= a.n_repl) epi.deferred_local(i);
                else if constexpr (FUSEP) a.pnew[i] = gat.load(i);
            }
        } else if constexpr (FUSEP) {            // (p_new of those rows must still be stored)
            for (int q = blockIdx.x * SPMV_THREADS + threadIdx.x; q < S.ndef; q += gridDim.x * SPMV_THREADS) { const int i = S.def_rows[q]; a.pnew[i] = gat.load(i); }
        }
    }
    spmv_walk<DEFER>(S, gat, epi, prod, ww);
    block_reduce_store<3, SPMV_THREADS>(epi.acc, red, a.partials + 3 * (int64_t)blockIdx.x);
    KKT_STAMP(1);
}

// window-panel form of the sweep (stand-alone applies and CG iterations alike: no dual tiles, the p update is a kernel of its own)
template <class GEO, bool DEFER, bool FOLD>
__global__ __launch_bounds__(GEO::THREADS, 4) void kkt2_win_kernel(DevBlkCsr S, KktArgs a) {
    static_assert(!FOLD || DEFER, "FOLD is about deferred rows");
    WIN_STAMP_G(0);
    if (a.gate && a.close_j < 0 && a.st->done) return;
    extern __shared__ __attribute__((aligned(16))) double wlds[];
    GatherW gat;
    gat.w = a.w;
    if (a.close_j >= 0 && cgm_close_in_sweep(a.st, a.rr_partials, a.rr_count, a.w, (int64_t)a.nm + 1, a.close_j, a.seq_base >> 11, a.batch_mark)) return;
    EpiKkt<GatherW, FOLD> epi;
    epi.gat = gat; epi.out = a.out; epi.pnew = nullptr; epi.cb = a.cb; epi.n = a.n; epi.wt = gat.load_u(a.nm);
    epi.acc[0] = epi.acc[1] = epi.acc[2] = 0.0;
    if (a.vt_out && blockIdx.x == 0 && threadIdx.x == 0) { a.vt_out[0] = epi.wt.x; a.vt_out[1] = epi.wt.y; }
    if constexpr (FOLD) {                                 // (as kkt2_kernel: the slot-free part of the slot-spread rows' share of the sums)
        if (a.count_repl || S.ndef > a.n_repl) {
            for (int q = blockIdx.x * GEO::THREADS + threadIdx.x; q < S.ndef; q += gridDim.x * GEO::THREADS) {
                const int i = S.def_rows[q];
                if (a.count_repl || i >= a.n_repl) epi.deferred_local(i);
            }
        }
    }
    win_walk<GEO, true, DEFER>(S, gat, epi, wlds);
    block_reduce_store<3, GEO::THREADS>(epi.acc, wlds + (size_t)(GEO::COLS + GEO::ROWS) * 2, a.partials + 3 * (int64_t)blockIdx.x);
    WIN_STAMP_G(5);
}
}  // namespace fos
// (not part of the ABI: timing experiments -- tools/win_stamps.py, tools/kkt_stamps.py; -1 unless compiled with the matching -DFOS_*_STAMPS)
extern "C" int fos_debug_win_stamps(long long* out, int n) {
#ifdef FOS_WIN_STAMPS
    if (n > 16 * 64 * 8) n = 16 * 64 * 8;
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(fos::g_win_stamps), sizeof(long long) * (size_t)n);
#else
    (void)out; (void)n; return -1;
#endif
}
extern "C" int fos_debug_kkt_stamps(long long* out, int n) {
#ifdef FOS_KKT_STAMPS
    if (n > 2 * 16384) n = 2 * 16384;
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(fos::g_kkt_stamps), sizeof(long long) * (size_t)n);
#else
    (void)out; (void)n; return -1;
#endif
}
namespace fos {
// dynamic LDS above 64 KB needs an opt-in per kernel and device
template <class K>
static bool win_lds_optin(K kernel, size_t bytes) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    if (e != hipSuccess) { set_error("hipFuncSetAttribute(window kernel, %zu bytes of LDS): %s", bytes, hipGetErrorString(e)); return false; }
    return true;
}
template <class GEO, bool DEFER, bool FOLD>
static void launch_kkt2_win_as(const LaunchCtx& c, const KktArgs& a) {
    (void)win_lds_optin(kkt2_win_kernel<GEO, DEFER, FOLD>, win_lds_bytes<GEO>(2));       // cheap (a table update); a failure surfaces through check_launch
    hipLaunchKernelGGL((kkt2_win_kernel<GEO, DEFER, FOLD>), dim3(c.S.nwg), dim3(GEO::THREADS), win_lds_bytes<GEO>(2), c.stream, c.S, a);
}
// window-panel sweep; fold: the slot-spread rows' share of the sums is added by the sweep (row-sharded operators: the rows of A')
static void launch_kkt2_win(const LaunchCtx& c, const KktArgs& a, bool fold) {
    const bool defer = c.S.ndef > 0;
    if (c.S.win_tall) {
        if (!defer) launch_kkt2_win_as<WinTall, false, false>(c, a);
        else if (fold) launch_kkt2_win_as<WinTall, true, true>(c, a);
        else launch_kkt2_win_as<WinTall, true, false>(c, a);
    } else {
        if (!defer) launch_kkt2_win_as<WinStd, false, false>(c, a);
        else if (fold) launch_kkt2_win_as<WinStd, true, true>(c, a);
        else launch_kkt2_win_as<WinStd, true, false>(c, a);
    }
}

// Deferred rows (dual tiles): `lpr` lanes (a power of two <= 64, S.def_lpr) share a row: lane-strided partial sums of the
// row's slots in list order, then the fixed DPP butterfly -- long slot lists (a dense LP: one partial per 64-row tile) are
// latency bound with one thread per row.  Used by the stand-alone applies (CG start, rhs build, status, test entries); inside
// a CG iteration cg_update_kernel does the same sums itself.
// (row-sharded: rows below n_repl -- the rows of A' -- are replicated and counted in the scalar sums by ONE rank (count != 0); the
// rows of A that are spread over column chunks are local and always counted)
template <class Epi>
__device__ __forceinline__ void deferred_rows(const DevBlkCsr& S, Epi& epi, int n_repl = 0, int count = 1) {
    const d2* __restrict__ slots = reinterpret_cast<const d2*>(S.slots_rd);
    const int lpr = S.def_lpr, sh = 31 - __clz(lpr);
    const int rows_per_pass = (gridDim.x * DEF_THREADS) >> sh;
    const int lig = threadIdx.x & (lpr - 1);
    const int npass = (S.ndef + rows_per_pass - 1) / rows_per_pass;           // uniform trip count: the DPP sums need full waves
    int q = (blockIdx.x * DEF_THREADS + threadIdx.x) >> sh;
    for (int pass = 0; pass < npass; ++pass, q += rows_per_pass) {
        const bool ok = q < S.ndef;
        DefRow dr{};
        if (ok) dr = ld_defrow(S.def_rec + q);
        const int row = dr.row;
        RowPre pr{};
        if (ok && lig == 0) pr = epi.pre(row);
        double u1 = 0.0, u2 = 0.0;
        if (ok) slot_list_sum(slots, S.def_idx, dr, lig, lpr, u1, u2);
        u1 = group_sum(u1, lpr);
        u2 = group_sum(u2, lpr);
        if (ok && lig == 0) {
            if (count || row >= n_repl) epi.row(row, u1, u2, pr);
            else {                                       // finished, but its share of the sums belongs to another rank
                auto keep = epi.acc_copy();
                epi.row(row, u1, u2, pr);
                epi.acc_restore(keep);
            }
        }
    }
}
// The sweep left nwg partial-sum records; the deferred-row kernel runs after it anyway, so each of its nwg_def workgroups
// also folds a slice of them into its own record: the consumers then read nwg_def instead of nwg + nwg_def of them.
// Thread t of workgroup b takes records b, b + nwg_def, ...
template <int NACC>
__device__ __forceinline__ void fold_sweep_records(const DevBlkCsr& S, const double* __restrict__ partials, double (&acc)[NACC]) {
    for (int rec = blockIdx.x + gridDim.x * (int)threadIdx.x; rec < S.nwg; rec += gridDim.x * DEF_THREADS) {
#pragma unroll
        for (int a = 0; a < NACC; ++a) acc[a] += partials[(int64_t)rec * NACC + a];
    }
}
__global__ __launch_bounds__(DEF_THREADS) void kkt2_deferred_kernel(DevBlkCsr S, const d2* __restrict__ w, d2* __restrict__ out,
                                                                    const double* __restrict__ cb, int n, int nm,
                                                                    double* __restrict__ partials, const DevState* st, int gate, int count) {
    if (gate && st->done) return;
    __shared__ double red[16];
    EpiKkt<GatherW, false> epi;
    epi.gat.w = w; epi.out = out; epi.pnew = nullptr; epi.cb = cb; epi.n = n; epi.wt = w[nm];
    epi.acc[0] = epi.acc[1] = epi.acc[2] = 0.0;
    deferred_rows(S, epi, n, count);
    fold_sweep_records<3>(S, partials, epi.acc);
    block_reduce_store<3, DEF_THREADS>(epi.acc, red, partials + 3 * (int64_t)(S.nwg + blockIdx.x));
}

// Row-sharded operators with dual tiles: the LOCAL slot list of every row of A' (its own partial + the column sums of the tiles
// above it, list order) -> one partial sum per row, the n-vector that crosses the ranks (solver.cpp, sum_slots_over_ranks).
__global__ __launch_bounds__(DEF_THREADS) void slots_compact_kernel(int nrows, const DefRow* __restrict__ rec, const int32_t* __restrict__ idx,
                                                                    const d2* __restrict__ slots, d2* __restrict__ out, int lpr) {
    const int sh = 31 - __clz(lpr);
    const int rows_per_pass = (gridDim.x * DEF_THREADS) >> sh;
    const int lig = threadIdx.x & (lpr - 1);
    const int npass = (nrows + rows_per_pass - 1) / rows_per_pass;            // uniform trip count: the DPP sums need full waves
    int q = (blockIdx.x * DEF_THREADS + threadIdx.x) >> sh;
    for (int pass = 0; pass < npass; ++pass, q += rows_per_pass) {
        const bool ok = q < nrows;
        DefRow dr{};
        if (ok) dr = ld_defrow(rec + q);
        double u1 = 0.0, u2 = 0.0;
        if (ok) slot_list_sum(slots, idx, dr, lig, lpr, u1, u2);
        u1 = group_sum(u1, lpr);
        u2 = group_sum(u2, lpr);
        if (ok && lig == 0) out[dr.row] = make_double2(u1, u2);
    }
}
void launch_slots_compact(const LaunchCtx& c, int nrows, const DefRow* rec, const int32_t* idx, int lpr, const double* slots, double* out) {
    const int grid = (int)std::min<int64_t>(DEF_MAX_WG, ((int64_t)nrows * lpr + DEF_THREADS - 1) / DEF_THREADS);
    hipLaunchKernelGGL(slots_compact_kernel, dim3(std::max(grid, 1)), dim3(DEF_THREADS), 0, c.stream, nrows, rec, idx, reinterpret_cast<const d2*>(slots),
                       reinterpret_cast<d2*>(out), lpr);
}

constexpr int FIN_THREADS = 1024;

// generic: partials[count][nacc] -> reduced[nacc]; sharded path: the all-reduce then runs on `reduced` (RCCL), or --
// PEER -- this kernel itself exchanges the sums with the peer ranks through their mailboxes (fos_internal.hpp, PeerBox)
template <bool PEER>
__global__ __launch_bounds__(FIN_THREADS) void reduce_kernel(const double* __restrict__ partials, int count, int nacc,
                                                             double* __restrict__ reduced, DevState* st, int gate, PeerBox pb) {
    if (gate && st->done) return;
    if (PEER && st->xchg_failed) return;
    __shared__ double sums[8];
    // nacc <= 8; instantiate by value
    switch (nacc) {
        case 1: reduce_partials<1>(partials, count, sums); break;
        case 3: reduce_partials<3>(partials, count, sums); break;
        case 6: reduce_partials<6>(partials, count, sums); break;
        default: return;
    }
    if constexpr (!PEER) {
        if ((int)threadIdx.x < nacc) reduced[threadIdx.x] = sums[threadIdx.x];
    } else {
        (void)peer_exchange_wg(pb, sums, nacc, reduced, st);
    }
}

// ---- n-vector exchange of a row-sharded operator through peer-mapped memory (fos_internal.hpp, VecBox)
constexpr int VEC_X_THREADS = 256;
__global__ __launch_bounds__(VEC_X_THREADS) void vec_push_kernel(VecBox vb, uint32_t seq, const double* __restrict__ slots, DevState* st) {
    if (st->xchg_failed) return;
    const size_t half = (size_t)(seq & 1u) * (size_t)vb.nranks;
    const int64_t n2 = vb.n2;
    for (int r = 0; r < vb.nranks; ++r) {
        if (r == vb.rank) continue;
        double* dst = vb.buf[r] + (half + (size_t)vb.rank) * (size_t)n2;
        for (int64_t k = blockIdx.x * (int64_t)VEC_X_THREADS + threadIdx.x; k < n2; k += (int64_t)gridDim.x * VEC_X_THREADS)
            __builtin_nontemporal_store(slots[k], dst + k);
    }
    __threadfence_system();                                  // this thread's stores have left for the peers ...
    __syncthreads();
    __shared__ int last;
    if (threadIdx.x == 0) last = (atomicAdd(vb.counter, 1u) == gridDim.x - 1) ? 1 : 0;      // ... before the workgroup is counted
    __syncthreads();
    if (last) {
        __threadfence_system();
        if ((int)threadIdx.x < vb.nranks && (int)threadIdx.x != vb.rank)
            __hip_atomic_store(vb.flags[threadIdx.x] + half + vb.rank, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        if (threadIdx.x == 0) *vb.counter = 0u;              // (the next push is a later launch)
    }
}
__global__ __launch_bounds__(VEC_X_THREADS) void vec_sum_kernel(VecBox vb, uint32_t seq, const double* __restrict__ slots,
                                                                double* __restrict__ slots_rd, DevState* st) {
    if (st->xchg_failed) return;
    const size_t half = (size_t)(seq & 1u) * (size_t)vb.nranks;
    __shared__ int failed;
    if (threadIdx.x == 0) failed = 0;
    __syncthreads();
    if ((int)threadIdx.x < vb.nranks && (int)threadIdx.x != vb.rank) {
        const uint32_t* f = vb.flags[vb.rank] + half + threadIdx.x;
        const long long t0 = wall_clock64();
        bool ok;
        do { ok = __hip_atomic_load(f, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) == seq; } while (!ok && (wall_clock64() - t0) < vb.timeout_ticks);
        if (!ok) failed = 1;
    }
    __syncthreads();
    if (failed) { if (blockIdx.x == 0 && threadIdx.x == 0) { st->xchg_failed = 1; st->done = 1; } return; }
    const int64_t n2 = vb.n2;
    const double* own = vb.buf[vb.rank] + half * (size_t)n2;
    for (int64_t k = blockIdx.x * (int64_t)VEC_X_THREADS + threadIdx.x; k < n2; k += (int64_t)gridDim.x * VEC_X_THREADS) {
        double sacc = 0.0;
        for (int r = 0; r < vb.nranks; ++r)                  // rank order: the same bits on every rank
            sacc += (r == vb.rank) ? slots[k] : __builtin_nontemporal_load(own + (size_t)r * (size_t)n2 + k);
        slots_rd[k] = sacc;
    }
}
// ---- the same exchange as reduce-scatter + all-gather (three or more ranks: 2 (g-1) n / g doubles leave a rank instead of (g-1) n).
// Rank o owns the slice [o chunk, (o+1) chunk) of the 2n doubles.  Stage 1: every rank writes its partial sums OF THAT SLICE into
// o's buffer (region [parity][sender]); o adds them in rank order.  Stage 2: o writes the summed slice into every peer's buffer -- the
// same region [parity][o], whose entries of o's slice stage 1 never touches -- and raises the stage-2 flag; everybody copies the
// slices it does not own.  One sum per entry, made by its owner: the same bits on every rank, and the bits of the direct scheme.
__device__ __forceinline__ void vec_slice(const VecBox& vb, int o, int64_t& k0, int64_t& k1) {
    const int64_t chunk = (vb.n2 + vb.nranks - 1) / vb.nranks;
    k0 = std::min<int64_t>(vb.n2, (int64_t)o * chunk);
    k1 = std::min<int64_t>(vb.n2, k0 + chunk);
}
// raise flag `which` (0: stage 1, 1: stage 2) in every peer's flag array once the whole grid's stores have left
__device__ __forceinline__ void vec_raise_flags(const VecBox& vb, uint32_t seq, size_t half, int which) {
    __threadfence_system();
    __syncthreads();
    __shared__ int last;
    if (threadIdx.x == 0) last = (atomicAdd(vb.counter, 1u) == gridDim.x - 1) ? 1 : 0;
    __syncthreads();
    if (last) {
        __threadfence_system();
        if ((int)threadIdx.x < vb.nranks && (int)threadIdx.x != vb.rank)
            __hip_atomic_store(vb.flags[threadIdx.x] + (size_t)which * 2 * vb.nranks + half + vb.rank, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        if (threadIdx.x == 0) *vb.counter = 0u;
    }
}
// wait until every peer's flag `which` of this exchange has arrived; false (and the solve is stopped) on a time-out
__device__ __forceinline__ bool vec_wait_flags(const VecBox& vb, uint32_t seq, size_t half, int which, DevState* st) {
    __shared__ int failed;
    if (threadIdx.x == 0) failed = 0;
    __syncthreads();
    if ((int)threadIdx.x < vb.nranks && (int)threadIdx.x != vb.rank) {
        const uint32_t* f = vb.flags[vb.rank] + (size_t)which * 2 * vb.nranks + half + threadIdx.x;
        const long long t0 = wall_clock64();
        bool ok;
        do { ok = __hip_atomic_load(f, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) == seq; } while (!ok && (wall_clock64() - t0) < vb.timeout_ticks);
        if (!ok) failed = 1;
    }
    __syncthreads();
    if (failed) { if (blockIdx.x == 0 && threadIdx.x == 0) { st->xchg_failed = 1; st->done = 1; } return false; }
    return true;
}
__global__ __launch_bounds__(VEC_X_THREADS) void vec_rs_push_kernel(VecBox vb, uint32_t seq, const double* __restrict__ slots, DevState* st) {
    if (st->xchg_failed) return;
    const size_t half = (size_t)(seq & 1u) * (size_t)vb.nranks;
    for (int o = 0; o < vb.nranks; ++o) {
        if (o == vb.rank) continue;
        int64_t k0, k1;
        vec_slice(vb, o, k0, k1);
        double* dst = vb.buf[o] + (half + (size_t)vb.rank) * (size_t)vb.n2;
        for (int64_t k = k0 + blockIdx.x * (int64_t)VEC_X_THREADS + threadIdx.x; k < k1; k += (int64_t)gridDim.x * VEC_X_THREADS)
            __builtin_nontemporal_store(slots[k], dst + k);
    }
    vec_raise_flags(vb, seq, half, 0);
}
__global__ __launch_bounds__(VEC_X_THREADS) void vec_rs_sum_kernel(VecBox vb, uint32_t seq, const double* __restrict__ slots,
                                                                   double* __restrict__ slots_rd, DevState* st) {
    if (st->xchg_failed) return;
    const size_t half = (size_t)(seq & 1u) * (size_t)vb.nranks;
    if (!vec_wait_flags(vb, seq, half, 0, st)) return;
    int64_t k0, k1;
    vec_slice(vb, vb.rank, k0, k1);
    const double* own = vb.buf[vb.rank] + half * (size_t)vb.n2;
    for (int64_t k = k0 + blockIdx.x * (int64_t)VEC_X_THREADS + threadIdx.x; k < k1; k += (int64_t)gridDim.x * VEC_X_THREADS) {
        double sacc = 0.0;
        for (int r = 0; r < vb.nranks; ++r)                  // rank order: the bits of the direct scheme
            sacc += (r == vb.rank) ? slots[k] : __builtin_nontemporal_load(own + (size_t)r * (size_t)vb.n2 + k);
        slots_rd[k] = sacc;
        for (int p = 0; p < vb.nranks; ++p)
            if (p != vb.rank) __builtin_nontemporal_store(sacc, vb.buf[p] + (half + (size_t)vb.rank) * (size_t)vb.n2 + k);
    }
    vec_raise_flags(vb, seq, half, 1);
}
__global__ __launch_bounds__(VEC_X_THREADS) void vec_ag_copy_kernel(VecBox vb, uint32_t seq, double* __restrict__ slots_rd, DevState* st) {
    if (st->xchg_failed) return;
    const size_t half = (size_t)(seq & 1u) * (size_t)vb.nranks;
    if (!vec_wait_flags(vb, seq, half, 1, st)) return;
    const double* own = vb.buf[vb.rank] + half * (size_t)vb.n2;
    for (int o = 0; o < vb.nranks; ++o) {
        if (o == vb.rank) continue;
        int64_t k0, k1;
        vec_slice(vb, o, k0, k1);
        for (int64_t k = k0 + blockIdx.x * (int64_t)VEC_X_THREADS + threadIdx.x; k < k1; k += (int64_t)gridDim.x * VEC_X_THREADS)
            slots_rd[k] = __builtin_nontemporal_load(own + (size_t)o * (size_t)vb.n2 + k);
    }
}
void launch_vec_exchange(const LaunchCtx& c, const VecBox& vb, uint32_t seq, const double* slots, double* slots_rd) {
    const int grid = (int)std::max<int64_t>(1, std::min<int64_t>(1024, (vb.n2 + VEC_X_THREADS - 1) / VEC_X_THREADS));
    // two ranks move the same bytes either way (and the direct scheme has one hop less); FOS_VEC_RSAG=0/1 forces
    static const int rsag_env = getenv("FOS_VEC_RSAG") ? atoi(getenv("FOS_VEC_RSAG")) : -1;
    const bool rsag = rsag_env >= 0 ? rsag_env != 0 : vb.nranks >= 3;
    if (rsag) {
        const int g1 = (int)std::max<int64_t>(1, std::min<int64_t>(1024, (vb.n2 / vb.nranks + VEC_X_THREADS) / VEC_X_THREADS));
        hipLaunchKernelGGL(vec_rs_push_kernel, dim3(g1), dim3(VEC_X_THREADS), 0, c.stream, vb, seq, slots, c.st);
        hipLaunchKernelGGL(vec_rs_sum_kernel, dim3(g1), dim3(VEC_X_THREADS), 0, c.stream, vb, seq, slots, slots_rd, c.st);
        hipLaunchKernelGGL(vec_ag_copy_kernel, dim3(g1), dim3(VEC_X_THREADS), 0, c.stream, vb, seq, slots_rd, c.st);
        return;
    }
    hipLaunchKernelGGL(vec_push_kernel, dim3(grid), dim3(VEC_X_THREADS), 0, c.stream, vb, seq, slots, c.st);
    hipLaunchKernelGGL(vec_sum_kernel, dim3(grid), dim3(VEC_X_THREADS), 0, c.stream, vb, seq, slots, slots_rd, c.st);
}

// tau rows of out = M w from the sweep's partials (the CG iteration does this inside cg_update_kernel)
__global__ __launch_bounds__(FIN_THREADS) void kkt_finalize_kernel(const double* __restrict__ partials, int count,
                                                                   const double* __restrict__ reduced, int from_reduced,
                                                                   const d2* __restrict__ w, d2* __restrict__ out, int nm,
                                                                   const DevState* st, int gate) {
    if (gate && st->done) return;
    __shared__ double sums[3];
    if (from_reduced) {
        if (threadIdx.x < 3) sums[threadIdx.x] = reduced[threadIdx.x];
        __syncthreads();
    } else {
        reduce_partials<3>(partials, count, sums);
    }
    if (threadIdx.x == 0) {
        const d2 pt = w[nm];
        const double T1 = sums[1], T2 = sums[2];
        // (Q v)_tau = -c'v_x - b'v_y = -T(v)                          HSDEAffine.jl:57
        out[nm] = make_double2(pt.x + T2, -T1 - pt.y);      // p1_tau - (Q p2)_tau ; (Q p1)_tau - p2_tau
    }
}

// closes CG iteration j-1 when no sweep of iteration j follows in the same batch of launches (the host then learns whether CG
// has stopped from DevState.done); if CG goes on, the next batch's sweep simply repeats the evaluation
__global__ __launch_bounds__(SPMV_THREADS) void cg_stop_check_kernel(KktArgs a) {
    if (a.st->done) return;
    if (a.pb.nranks > 0 && a.st->xchg_failed) return;
    (void)cg_close_iteration(a.st, a.rr_partials, a.rr_count, a.reduced, a.from_reduced, a.r, (int64_t)a.nm + 1, a.j - 1, a.pb, a.seq_base);
}

static KktArgs plain_args(const LaunchCtx& c, const double2* w, double2* out, int gate) {
    KktArgs a{};
    a.w = w; a.out = out; a.cb = c.cb; a.n = (int)c.n; a.nm = (int)(c.n + c.m); a.partials = c.partials; a.st = c.st; a.gate = gate;
    a.count_repl = c.count_repl; a.n_repl = (int)c.n_repl;
    a.close_j = -1;
    return a;
}
// the plain (not fused-p) sweep in the form the operator wants: FOLD = slot-spread rows' share of the sums added by the sweep;
// resident operators read their matrix stream with ordinary (cache-retaining) loads
static void launch_plain_sweep(const LaunchCtx& c, const KktArgs& a, bool fold) {
    if (c.S.npanel > 0) { launch_kkt2_win(c, a, fold); return; }
    dim3 grid(c.S.nwg), block(SPMV_THREADS);
    if (c.S.ndef > 0) {
        if (fold) {
            if (c.S.resident) hipLaunchKernelGGL((kkt2_kernel<true, false, true, false>), grid, block, 0, c.stream, c.S, a);
            else hipLaunchKernelGGL((kkt2_kernel<true, false, true>), grid, block, 0, c.stream, c.S, a);
        } else {
            if (c.S.resident) hipLaunchKernelGGL((kkt2_kernel<true, false, false, false>), grid, block, 0, c.stream, c.S, a);
            else hipLaunchKernelGGL((kkt2_kernel<true, false, false>), grid, block, 0, c.stream, c.S, a);
        }
    } else {
        if (c.S.resident) hipLaunchKernelGGL((kkt2_kernel<false, false, false, false>), grid, block, 0, c.stream, c.S, a);
        else hipLaunchKernelGGL((kkt2_kernel<false, false, false>), grid, block, 0, c.stream, c.S, a);
    }
}
// stand-alone apply: sweep (+ deferred-row kernel when the operator has dual tiles); leaves c.S.npart records at c.S.part_off
void launch_kkt2(const LaunchCtx& c, const double2* w, double2* out, int gate, bool finish_deferred) {
    const KktArgs a = plain_args(c, w, out, gate);
    if (c.S.ndef > 0 && finish_deferred) {
        launch_plain_sweep(c, a, false);
        if (c.between) (void)c.between(c.between_arg);        // row-sharded: the slots are summed over the ranks here
        hipLaunchKernelGGL(kkt2_deferred_kernel, dim3(c.S.nwg_def), dim3(DEF_THREADS), 0, c.stream, c.S, w, out, c.cb, (int)c.n,
                           (int)(c.n + c.m), c.partials, c.st, gate, (int)c.count_repl);
    } else {
        // (finish_deferred = false: the caller needs only rows the sweep finishes itself -- the slot-spread rows of `out` stay unwritten)
        launch_plain_sweep(c, a, false);
    }
}
// the sweep of CG iteration j (gated on DevState.done); leaves c.S.nwg records at record 0.
//   fuse_p && j >= 2: closes iteration j-1 and forms p_j = r + beta p_{j-1} on the fly (p_prev -> p_cur);
//   otherwise applies to p_cur as it stands (iteration 1, or the p update ran as its own kernel).
void launch_kkt2_cg(const LaunchCtx& c, const CgIter& it, double2* Ap) {
    KktArgs a = plain_args(c, it.p_cur, Ap, 1);
    const bool fused = it.fuse_p && it.j >= 2;
    if (fused) {
        a.w = it.p_prev; a.r = it.r; a.pnew = it.p_cur;
        a.rr_partials = c.partials + 3 * (size_t)PART_CAP; a.rr_count = c.cg_blocks;
        a.reduced = c.reduced; a.from_reduced = it.rr_from_reduced; a.j = it.j;
        if (it.fold) { a.pb = *it.fold; a.seq_base = it.seq_base; }
    }
    if (it.j == 1 && it.start_fused) {       // the start kernel left the r.r records of "iteration 0": this sweep adds them (g_0)
        a.close_j = 0; a.seq_base = it.seq_base;
        a.rr_partials = c.partials + 3 * (size_t)PART_CAP; a.rr_count = c.cg_blocks;
    }
    if (!fused || c.S.npanel > 0) { launch_plain_sweep(c, a, true); return; }       // (fuse_p is never set for window-panel operators)
    dim3 grid(c.S.nwg), block(SPMV_THREADS);
    if (c.S.ndef > 0) hipLaunchKernelGGL((kkt2_kernel<true, true, true>), grid, block, 0, c.stream, c.S, a);
    else hipLaunchKernelGGL((kkt2_kernel<false, true, false>), grid, block, 0, c.stream, c.S, a);
}
// merged-reduction CG: w = M r with the sums of w.r and of the tau row (c.S.nwg records at record 0), gated on DevState.done.
// closes > 0 (single GPU): the sweep first closes that iteration from the r.r records its update left (cgm_close_in_sweep).
void launch_cgm_sweep(const LaunchCtx& c, const CgmIter& it, int closes) {
    KktArgs a = plain_args(c, it.r, it.w, 1);
    a.vt_out = c.st->vtau;
    a.seq_base = it.seq_base;
    if (closes >= 0) {
        a.close_j = closes; a.batch_mark = it.batch_mark;
        a.rr_partials = c.partials + 3 * (size_t)PART_CAP + (size_t)(closes & 1) * CGM_RR_STRIDE; a.rr_count = c.cg_blocks;
    }
    launch_plain_sweep(c, a, true);
}
// merged-reduction CG, start of a solve: it.w = M v, every row but tau, slot-spread rows left in their slots (the start kernel
// finishes them); the sums of the tau row at record 0 (FOLD form: complete without the deferred-row kernel); NOT gated
void launch_cgm_apply(const LaunchCtx& c, const CgmIter& it, const double2* v) {
    KktArgs a = plain_args(c, v, it.w, 0);
    a.vt_out = c.st->vtau;
    launch_plain_sweep(c, a, true);
}
void launch_cg_stop_check(const LaunchCtx& c, const CgIter& it) {
    KktArgs a = plain_args(c, nullptr, nullptr, 1);
    a.r = it.r;
    a.rr_partials = c.partials + 3 * (size_t)PART_CAP; a.rr_count = c.cg_blocks;
    a.reduced = c.reduced; a.from_reduced = it.rr_from_reduced; a.j = it.j;
    if (it.fold) { a.pb = *it.fold; a.seq_base = it.seq_base; }
    hipLaunchKernelGGL(cg_stop_check_kernel, dim3(1), dim3(SPMV_THREADS), 0, c.stream, a);
}
void launch_reduce1(const LaunchCtx& c, int count, int nacc, int gate, int off) {
    const double* part = c.partials + (size_t)nacc * off;
    if (c.peer)
        hipLaunchKernelGGL(reduce_kernel<true>, dim3(1), dim3(FIN_THREADS), 0, c.stream, part, count, nacc, c.reduced, c.st, gate, *c.peer);
    else
        hipLaunchKernelGGL(reduce_kernel<false>, dim3(1), dim3(FIN_THREADS), 0, c.stream, part, count, nacc, c.reduced, c.st, gate, PeerBox{});
}
// `rounds` exchanges of four doubles back to back inside ONE launch (region 0 of the mailboxes, the protocol of reduce_kernel<true>): what an exchange
// costs on the transport the handle uses, without a launch or a host round trip between them (fos_exchange_bench: a diagnostic of N-rank runs)
__global__ __launch_bounds__(FIN_THREADS) void peer_chain_kernel(PeerBox pb, int rounds, double* __restrict__ reduced, DevState* st) {
    if (st->xchg_failed) return;
    __shared__ double sums[4];
    for (int r = 0; r < rounds; ++r) {
        if (threadIdx.x < 4) sums[threadIdx.x] = (double)(pb.rank + 1) + 0.25 * threadIdx.x + (double)r;
        __syncthreads();
        if (!peer_exchange_wg(pb, sums, 4, reduced, st)) return;
        __syncthreads();
    }
}
void launch_peer_chain(const LaunchCtx& c, int rounds) {
    hipLaunchKernelGGL(peer_chain_kernel, dim3(1), dim3(FIN_THREADS), 0, c.stream, *c.peer, rounds, c.reduced, c.st);
}
void launch_kkt_finalize(const LaunchCtx& c, const double2* w, double2* out, int gate, int from_reduced) {
    hipLaunchKernelGGL(kkt_finalize_kernel, dim3(1), dim3(FIN_THREADS), 0, c.stream, c.partials + 3 * (size_t)c.S.part_off, c.S.npart, c.reduced,
                       from_reduced, w, out, (int)(c.n + c.m), c.st, gate);
}

// ------------------------------------------------------------------------------------------------ single RHS Q apply

struct EpiQPlain {
    struct AccKeep { double v[1]; };
    __device__ __forceinline__ AccKeep acc_copy() const { AccKeep k; for (int a = 0; a < 1; ++a) k.v[a] = acc[a]; return k; }
    __device__ __forceinline__ void acc_restore(const AccKeep& k) { for (int a = 0; a < 1; ++a) acc[a] = k.v[a]; }
    static constexpr bool FOLDDEF = false;
    __device__ __forceinline__ void park(int, double, double, const d2&) {}     // out_plain[i] = sign * (Q v)_i ; acc[0] = [c;b].v
    const double* vcomp; double* out; const double* cb; int n; double vt, sign; double acc[1];
    __device__ __forceinline__ void init(double vtau) { vt = vtau; }
    __device__ __forceinline__ RowPre pre(int i) const { return RowPre{make_double2(vcomp[2 * (int64_t)i], 0.0), cb[i]}; }
    __device__ __forceinline__ void row(int i, double u, double, const RowPre& pr) {
        const double c = pr.c;
        const double q = (i < n) ? (u + vt * c) : -(u - vt * c);
        out[i] = sign * q;
        acc[0] += c * pr.v.x;
    }
};
struct EpiQRhs {
    struct AccKeep { double v[1]; };
    __device__ __forceinline__ AccKeep acc_copy() const { AccKeep k; for (int a = 0; a < 1; ++a) k.v[a] = acc[a]; return k; }
    __device__ __forceinline__ void acc_restore(const AccKeep& k) { for (int a = 0; a < 1; ++a) acc[a] = k.v[a]; }
    static constexpr bool FOLDDEF = false;
    __device__ __forceinline__ void park(int, double, double, const d2&) {}       // out[i] = (x1_i - (Q x2)_i, 0)      affinepluslinear.jl:94-95 (beta = 1, q = 0, rhs2 = b = 0)
    const d2* x; d2* out; const double* cb; int n; double vt; double acc[1];
    __device__ __forceinline__ void init(double vtau) { vt = vtau; }
    __device__ __forceinline__ RowPre pre(int i) const { return RowPre{x[i], cb[i]}; }
    __device__ __forceinline__ void row(int i, double u, double, const RowPre& pr) {
        const double c = pr.c;
        const d2 xi = pr.v;
        const double q = (i < n) ? (u + vt * c) : -(u - vt * c);
        out[i] = make_double2(-q + xi.x, 0.0);          // rhs1 .= beta.*rhs1 .+ x1 .- q with rhs1 = Q'x2 = -(Q x2)
        acc[0] += c * xi.y;
    }
};
struct EpiQVfromU {
    struct AccKeep { double v[1]; };
    __device__ __forceinline__ AccKeep acc_copy() const { AccKeep k; for (int a = 0; a < 1; ++a) k.v[a] = acc[a]; return k; }
    __device__ __forceinline__ void acc_restore(const AccKeep& k) { for (int a = 0; a < 1; ++a) acc[a] = k.v[a]; }
    static constexpr bool FOLDDEF = false;
    __device__ __forceinline__ void park(int, double, double, const d2&) {}    // out[i] = (y_i.x, (Q y.x)_i)        HSDEAffine.jl:122-124  v = Q u
    const d2* y; d2* out; const double* cb; int n; double vt; double acc[1];
    __device__ __forceinline__ void init(double vtau) { vt = vtau; }
    __device__ __forceinline__ RowPre pre(int i) const { return RowPre{y[i], cb[i]}; }
    __device__ __forceinline__ void row(int i, double u, double, const RowPre& pr) {
        const double c = pr.c;
        const d2 yi = pr.v;
        const double q = (i < n) ? (u + vt * c) : -(u - vt * c);
        out[i] = make_double2(yi.x, q);
        acc[0] += c * yi.x;
    }
};
struct EpiQStatus {
    struct AccKeep { double v[6]; };
    __device__ __forceinline__ AccKeep acc_copy() const { AccKeep k; for (int a = 0; a < 6; ++a) k.v[a] = acc[a]; return k; }
    __device__ __forceinline__ void acc_restore(const AccKeep& k) { for (int a = 0; a < 6; ++a) acc[a] = k.v[a]; }
    static constexpr bool FOLDDEF = false;
    __device__ __forceinline__ void park(int, double, double, const d2&) {}    // residual sums of checkstatus  HSDEStatus.jl:34-38,59,61  (z = [x;y;tau | r;s;kappa] interleaved)
    const d2* z; const double* cb; int n; double tau; double acc[6];
    __device__ __forceinline__ void init(double vtau) { tau = vtau; }
    __device__ __forceinline__ RowPre pre(int i) const { return RowPre{z[i], cb[i]}; }
    __device__ __forceinline__ void row(int i, double u, double, const RowPre& pr) {
        const d2 zi = pr.v;
        const double c = pr.c;
        if (i < n) {                       // u = (A'y)_i, zi = (x_i, r_i), c = c_i
            const double rd = (u / tau + c) - zi.y / tau;
            acc[ST_RD2] += rd * rd;
            acc[ST_ATY2] += u * u;
            acc[ST_CTX] += c * zi.x;
        } else {                           // u = (A x)_j, zi = (y_j, s_j), c = b_j
            const double rp = (u / tau + zi.y / tau) - c;
            acc[ST_RP2] += rp * rp;
            const double t = u + zi.y;
            acc[ST_AXS2] += t * t;
            acc[ST_BTY] += c * zi.x;
        }
    }
};

template <class Epi, int NACC, bool DEFER>
__global__ __launch_bounds__(SPMV_THREADS) void q1_kernel(DevBlkCsr S, const double* __restrict__ vcomp, Epi epi, int nm,
                                                          double* __restrict__ partials) {
    __shared__ __attribute__((aligned(16))) double prod[SPMV_WAVES * WNNZ];
    __shared__ double red[8 * NACC > 16 ? 8 * NACC : 16];
#pragma unroll
    for (int a = 0; a < NACC; ++a) epi.acc[a] = 0.0;
    epi.init(vcomp[2 * (int64_t)nm]);      // the tau entry of the gathered component
    Gather1 gat{vcomp};
    spmv_walk<DEFER>(S, gat, epi, prod, wave_work(S));
    block_reduce_store<NACC, SPMV_THREADS>(epi.acc, red, partials + NACC * (int64_t)blockIdx.x);
}
template <class Epi, int NACC>
__global__ __launch_bounds__(DEF_THREADS) void q1_deferred_kernel(DevBlkCsr S, const double* __restrict__ vcomp, Epi epi, int nm,
                                                                  double* __restrict__ partials, int count, int n_repl) {
    __shared__ double red[8 * NACC > 16 ? 8 * NACC : 16];
#pragma unroll
    for (int a = 0; a < NACC; ++a) epi.acc[a] = 0.0;
    epi.init(vcomp[2 * (int64_t)nm]);
    deferred_rows(S, epi, n_repl, count);
    fold_sweep_records<NACC>(S, partials, epi.acc);
    block_reduce_store<NACC, DEF_THREADS>(epi.acc, red, partials + NACC * (int64_t)(S.nwg + blockIdx.x));
}
template <class GEO, class Epi, int NACC, bool DEFER>
__global__ __launch_bounds__(GEO::THREADS, 4) void q1_win_kernel(DevBlkCsr S, const double* __restrict__ vcomp, Epi epi, int nm,
                                                             double* __restrict__ partials) {
    extern __shared__ __attribute__((aligned(16))) double wlds[];
#pragma unroll
    for (int a = 0; a < NACC; ++a) epi.acc[a] = 0.0;
    epi.init(vcomp[2 * (int64_t)nm]);
    Gather1 gat{vcomp};
    win_walk<GEO, true, DEFER>(S, gat, epi, wlds);
    block_reduce_store<NACC, GEO::THREADS>(epi.acc, wlds + (size_t)(GEO::COLS + GEO::ROWS), partials + NACC * (int64_t)blockIdx.x);
}
template <class GEO, class Epi, int NACC, bool DEFER>
static void launch_q1_win_as(const LaunchCtx& c, const double* vcomp, const Epi& e, int nm) {
    const size_t lds = win_lds_bytes<GEO>(1) + 16 * NACC * sizeof(double);
    (void)win_lds_optin(q1_win_kernel<GEO, Epi, NACC, DEFER>, lds);
    hipLaunchKernelGGL((q1_win_kernel<GEO, Epi, NACC, DEFER>), dim3(c.S.nwg), dim3(GEO::THREADS), lds, c.stream, c.S, vcomp, e, nm, c.partials);
}
template <class Epi, int NACC>
static void launch_q1_kernels(const LaunchCtx& c, const double* vcomp, const Epi& e, int nm) {
    dim3 grid(c.S.nwg), block(SPMV_THREADS);
    if (c.S.npanel > 0) {
        if (c.S.ndef > 0) {
            if (c.S.win_tall) launch_q1_win_as<WinTall, Epi, NACC, true>(c, vcomp, e, nm); else launch_q1_win_as<WinStd, Epi, NACC, true>(c, vcomp, e, nm);
            if (c.between) (void)c.between(c.between_arg);
            hipLaunchKernelGGL((q1_deferred_kernel<Epi, NACC>), dim3(c.S.nwg_def), dim3(DEF_THREADS), 0, c.stream, c.S, vcomp, e, nm, c.partials, (int)c.count_repl, (int)c.n_repl);
        } else {
            if (c.S.win_tall) launch_q1_win_as<WinTall, Epi, NACC, false>(c, vcomp, e, nm); else launch_q1_win_as<WinStd, Epi, NACC, false>(c, vcomp, e, nm);
        }
        return;
    }
    if (c.S.ndef > 0) {
        hipLaunchKernelGGL((q1_kernel<Epi, NACC, true>), grid, block, 0, c.stream, c.S, vcomp, e, nm, c.partials);
        if (c.between) (void)c.between(c.between_arg);
        hipLaunchKernelGGL((q1_deferred_kernel<Epi, NACC>), dim3(c.S.nwg_def), dim3(DEF_THREADS), 0, c.stream, c.S, vcomp, e, nm, c.partials, (int)c.count_repl, (int)c.n_repl);
    } else {
        hipLaunchKernelGGL((q1_kernel<Epi, NACC, false>), grid, block, 0, c.stream, c.S, vcomp, e, nm, c.partials);
    }
}

__global__ __launch_bounds__(FIN_THREADS) void q1_finalize_kernel(const double* __restrict__ partials, int count,
                                                                  const double* __restrict__ reduced, int from_reduced,
                                                                  int mode, const d2* __restrict__ v, double sign, void* out, int nm) {
    __shared__ double sums[1];
    if (from_reduced) {
        if (threadIdx.x == 0) sums[0] = reduced[0];
        __syncthreads();
    } else {
        reduce_partials<1>(partials, count, sums);
    }
    if (threadIdx.x == 0) {
        const double T = sums[0];            // [c;b].v ;  (Q v)_tau = -T
        if (mode == Q_PLAIN) reinterpret_cast<double*>(out)[nm] = sign * (-T);
        else if (mode == Q_RHS) reinterpret_cast<d2*>(out)[nm] = make_double2(T + v[nm].x, 0.0);   // x1_tau - (Q x2)_tau
        else if (mode == Q_VFROMU) reinterpret_cast<d2*>(out)[nm] = make_double2(v[nm].x, -T);
    }
}

// status: sums -> DevState.stat (tau and kappa appended)
__global__ __launch_bounds__(FIN_THREADS) void status_finalize_kernel(const double* __restrict__ partials, int count,
                                                                      const double* __restrict__ reduced, int from_reduced,
                                                                      const d2* __restrict__ z, int nm, DevState* st) {
    __shared__ double sums[6];
    if (from_reduced) {
        if (threadIdx.x < 6) sums[threadIdx.x] = reduced[threadIdx.x];
        __syncthreads();
    } else {
        reduce_partials<6>(partials, count, sums);
    }
    if (threadIdx.x < 6) st->stat[threadIdx.x] = sums[threadIdx.x];
    if (threadIdx.x == 0) { st->stat[ST_TAU] = z[nm].x; st->stat[ST_KAPPA] = z[nm].y; }
}

void launch_q1(const LaunchCtx& c, QMode mode, const double2* v, int comp, double sign, void* out) {
    const double* vcomp = reinterpret_cast<const double*>(v) + comp;
    const int nm = (int)(c.n + c.m);
    if (mode == Q_PLAIN) {
        EpiQPlain e; e.vcomp = vcomp; e.out = (double*)out; e.cb = c.cb; e.n = (int)c.n; e.vt = 0; e.sign = sign;
        launch_q1_kernels<EpiQPlain, 1>(c, vcomp, e, nm);
    } else if (mode == Q_RHS) {
        EpiQRhs e; e.x = v; e.out = (d2*)out; e.cb = c.cb; e.n = (int)c.n; e.vt = 0;
        launch_q1_kernels<EpiQRhs, 1>(c, vcomp, e, nm);
    } else if (mode == Q_VFROMU) {
        EpiQVfromU e; e.y = v; e.out = (d2*)out; e.cb = c.cb; e.n = (int)c.n; e.vt = 0;
        launch_q1_kernels<EpiQVfromU, 1>(c, vcomp, e, nm);
    } else {
        EpiQStatus e; e.z = v; e.cb = c.cb; e.n = (int)c.n; e.tau = 0;
        launch_q1_kernels<EpiQStatus, 6>(c, vcomp, e, nm);
    }
}
void launch_q1_finalize(const LaunchCtx& c, QMode mode, const double2* v, int comp, double sign, void* out, int from_reduced) {
    (void)comp;
    hipLaunchKernelGGL(q1_finalize_kernel, dim3(1), dim3(FIN_THREADS), 0, c.stream, c.partials + (size_t)c.S.part_off, c.S.npart, c.reduced,
                       from_reduced, (int)mode, v, sign, out, (int)(c.n + c.m));
}
void launch_status_finalize(const LaunchCtx& c, const double2* z, int from_reduced) {
    hipLaunchKernelGGL(status_finalize_kernel, dim3(1), dim3(FIN_THREADS), 0, c.stream, c.partials + 6 * (size_t)c.S.part_off, c.S.npart, c.reduced,
                       from_reduced, z, (int)(c.n + c.m), c.st);
}

}  // namespace fos
