// Host-side construction of the device operator format.
//
// Input : A as Julia SparseMatrixCSC{Float64,Int64} (1-based colptr/rowval), m x n   (src/types.jl:35).
// Output: the stacked matrix  S = [[0, A'],[A, 0]]  of size (n+m) x (n+m) in block-padded CSR:
//   rows 0..n-1   = rows of A' (= columns of A): entries (n + j, a_ji)  -> gather from the y part
//   rows n..n+m-1 = rows of A                  : entries (i, a_ji)      -> gather from the x part
// S * [vx; vy] = [A'vy; A vx] -- the two SpMVs of HSDEMatrixQ.mul! (HSDEAffine.jl:51-52) in one sweep.
//
// Rows are grouped into ROW BLOCKS (CSR-adaptive): consecutive rows with at most NNZ_BLK non-zeros and at
// most ROWS_BLK rows form a "stream" block (staged through LDS, then reduced per row); a row with more than
// NNZ_BLK non-zeros is a block of its own ("long" row, reduced by the whole workgroup).  Each block's entries
// start at a multiple of NNZ_ALIGN in the padded arrays (pad: value 0, a valid column; never summed).
// Row blocks are then split among the persistent workgroups of the SpMV grid, balanced by non-zeros.
#include <algorithm>
#include <cstdarg>

#include "fos_internal.hpp"

namespace fos {

static thread_local char g_err[1024] = "";

void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

const char* last_error_cstr() { return g_err; }

void partition_workgroups(HostBlkCsr* S, int nwg_target) {
    int nblk = S->nblk;
    int nwg = std::max(1, std::min(nwg_target, nblk));
    if (nwg >= 8) nwg -= nwg % 8;     // XCD remap in the kernel wants a multiple of 8
    // cost of a block: its padded non-zeros plus a per-row term for the epilogue
    std::vector<double> cost(nblk + 1, 0.0);
    for (int b = 0; b < nblk; ++b) {
        double nz = double(S->blk_nnz1[b] - S->blk_nnz0[b]);
        double rows = double(S->blk_row0[b + 1] - S->blk_row0[b]);
        cost[b + 1] = cost[b] + nz + 4.0 * rows + 32.0;
    }
    double total = cost[nblk];
    S->wg_blk0.assign(nwg + 1, 0);
    int b = 0;
    for (int g = 1; g < nwg; ++g) {
        double target = total * double(g) / double(nwg);
        while (b < nblk && cost[b + 1] <= target) ++b;
        // choose the closer boundary
        if (b < nblk && (target - cost[b]) > (cost[b + 1] - target)) ++b;
        if (b < S->wg_blk0[g - 1]) b = S->wg_blk0[g - 1];
        S->wg_blk0[g] = b;
    }
    S->wg_blk0[nwg] = nblk;
    S->nwg = nwg;
}

int build_stacked_csr(int64_t m, int64_t n, const int64_t* colptr, const int64_t* rowval, const double* nzval,
                      int nwg_target, HostBlkCsr* out) {
    if (m < 0 || n < 0) { set_error("negative dimension"); return FOS_EINVAL; }
    if (n + m + 1 > (int64_t)INT32_MAX / 2) { set_error("n+m too large for int32 column indices"); return FOS_EUNSUPPORTED; }
    if (colptr[0] != 1) { set_error("colptr must be 1-based (colptr[1] == 1)"); return FOS_EINVAL; }
    const int64_t nnz = colptr[n] - 1;
    for (int64_t j = 0; j < n; ++j)
        if (colptr[j + 1] < colptr[j]) { set_error("colptr not monotone at column %lld", (long long)j + 1); return FOS_EINVAL; }
    for (int64_t k = 0; k < nnz; ++k)
        if (rowval[k] < 1 || rowval[k] > m) { set_error("rowval[%lld] = %lld out of 1..m", (long long)k + 1, (long long)rowval[k]); return FOS_EINVAL; }

    const int64_t nrows = n + m;
    // ---- plain CSR of S: row pointers
    std::vector<int64_t> rp(nrows + 1, 0);
    for (int64_t j = 0; j < n; ++j) rp[j + 1] = colptr[j + 1] - colptr[j];          // A' rows = A columns
    for (int64_t k = 0; k < nnz; ++k) rp[n + rowval[k]] += 1;                       // A rows (rowval 1-based -> n + (r-1) + 1)
    for (int64_t r = 0; r < nrows; ++r) rp[r + 1] += rp[r];
    if (rp[nrows] != 2 * nnz) { set_error("internal: stacked nnz mismatch"); return FOS_EINVAL; }

    // ---- row blocks
    HostBlkCsr& S = *out;
    S = HostBlkCsr();
    S.nrows = nrows;
    S.nnz = 2 * nnz;
    S.row_rel.assign(nrows, 0);
    std::vector<int64_t> row_pos(nrows, 0);   // position of every row's first entry in the padded arrays
    int64_t pos = 0;
    int64_t r = 0;
    while (r < nrows) {
        int64_t r0 = r;
        int64_t len0 = rp[r + 1] - rp[r];
        int64_t blk_start = (pos + NNZ_ALIGN - 1) / NNZ_ALIGN * NNZ_ALIGN;
        int64_t cnt = 0;
        if (len0 > NNZ_BLK) {                   // long row: a block of its own
            row_pos[r] = blk_start;
            cnt = len0;
            r += 1;
        } else {
            while (r < nrows && (r - r0) < ROWS_BLK) {
                int64_t len = rp[r + 1] - rp[r];
                if (cnt + len > NNZ_BLK) break;
                row_pos[r] = blk_start + cnt;
                S.row_rel[r] = (uint16_t)cnt;
                cnt += len;
                r += 1;
            }
        }
        S.blk_row0.push_back((int32_t)r0);
        S.blk_nnz0.push_back(blk_start);
        S.blk_nnz1.push_back(blk_start + cnt);
        pos = blk_start + cnt;
    }
    S.blk_row0.push_back((int32_t)nrows);
    S.nblk = (int32_t)S.blk_nnz0.size();
    S.nnz_padded = (pos + NNZ_ALIGN - 1) / NNZ_ALIGN * NNZ_ALIGN;
    if (S.nnz_padded == 0) S.nnz_padded = NNZ_ALIGN;

    S.val.assign(S.nnz_padded, 0.0);
    S.col.assign(S.nnz_padded, 0);
    // ---- fill A' rows (row j of S = column j of A, entries already sorted by row index)
    for (int64_t j = 0; j < n; ++j) {
        int64_t dst = row_pos[j];
        for (int64_t k = colptr[j] - 1; k < colptr[j + 1] - 1; ++k, ++dst) {
            S.val[dst] = nzval[k];
            S.col[dst] = (int32_t)(n + rowval[k] - 1);
        }
    }
    // ---- fill A rows by a counting transpose (column order inside each row = ascending column index)
    {
        std::vector<int64_t> fill(m, 0);
        for (int64_t j = 0; j < n; ++j) {
            for (int64_t k = colptr[j] - 1; k < colptr[j + 1] - 1; ++k) {
                int64_t i = rowval[k] - 1;
                int64_t dst = row_pos[n + i] + fill[i]++;
                S.val[dst] = nzval[k];
                S.col[dst] = (int32_t)j;
            }
        }
    }
    partition_workgroups(&S, nwg_target);
    return FOS_OK;
}

}  // namespace fos
