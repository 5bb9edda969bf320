// Host-side construction of the device operator format.
//
// Input : A as Julia SparseMatrixCSC{Float64,Int64} (1-based colptr/rowval), m x n   (src/types.jl:35).
// Output: the stacked matrix  S = [[0, A'],[A, 0]]  of size (n+m) x (n+m) in block-padded CSR:
//   rows 0..n-1   = rows of A' (= columns of A): entries (n + j, a_ji)  -> gather from the y part
//   rows n..n+m-1 = rows of A                  : entries (i, a_ji)      -> gather from the x part
// S * [vx; vy] = [A'vy; A vx] -- the two SpMVs of HSDEMatrixQ.mul! (HSDEAffine.jl:51-52) in one sweep.
//
// Rows are grouped into ROW BLOCKS, the unit of work of ONE wavefront.  Three kinds:
//   ELL   consecutive rows of similar length (<= WROWS rows, padding <= ~1/6): `tpr` = 64 / nextpow2(rows) lanes per
//         row, entries stored LANE-MAJOR (entry e of row i at step e / tpr, lane i*tpr + e % tpr), so the
//         wavefront's coalesced 64-entry loads land every product in the lane that sums it -- no LDS, no shuffles
//         beyond log2(tpr) DPP steps;
//   LDS   irregular rows (<= WNNZ entries): CSR order, products staged through the wavefront's LDS slice and
//         reduced per row;
//   LONG  a row with more than ELL_MAX entries: strided by the wavefront, reduced in-register.
// Index compression: a block whose rows all have CONSECUTIVE column indices (dense blocks of an SDP, dense LP rows,
// banded matrices) is a RUN block: it stores one first-column per row instead of 4 bytes per entry (8 instead of
// 12 bytes per non-zero on the wire).  Blocks are homogeneous in run-ness.
// Each block's entries start at a multiple of NNZ_ALIGN (pad: value 0, column 0; masked in the kernel).
// Row blocks are then split among the persistent wavefronts of the SpMV grid, balanced by stored entries.
#include <algorithm>
#include <cstdarg>
#include <cmath>
#include <cstdlib>

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
    const int nblk = S->nblk;
    int nwg = std::max(1, std::min(nwg_target, (nblk + SPMV_WAVES - 1) / SPMV_WAVES));
    // one block per wavefront when the grid allows it (blocks of unequal cost -- tall tiles of 4 and 5 sub-tiles -- must not be
    // paired up by the cost balance below while other wavefronts stay empty)
    const bool one_each = (nblk + SPMV_WAVES - 1) / SPMV_WAVES <= nwg_target;
    if (nwg >= 8) nwg = one_each ? std::min(16384, (nwg + 7) / 8 * 8) : nwg - nwg % 8;     // XCD remap in the kernel wants a multiple of 8
    const int nwaves = nwg * SPMV_WAVES;
    if (one_each && nblk <= nwaves) {
        S->wave_blk0.assign(nwaves + 1, 0);
        for (int g = 0; g <= nwaves; ++g) S->wave_blk0[g] = std::min(g, nblk);
        S->wave_first.assign(nwaves, BlkDesc{});
        for (int g = 0; g < nblk; ++g) S->wave_first[g] = S->blk[g];
        S->nwg = nwg;
        S->nwaves = nwaves;
        return;
    }
    // cost of a block: its non-zeros plus a per-row term for the epilogue and a fixed per-block term
    std::vector<double> cost(nblk + 1, 0.0);
    for (int b = 0; b < nblk; ++b)
        cost[b + 1] = cost[b] + double(S->blk[b].cnt) * (S->blk[b].kind() == BLK_LONG ? S->blk[b].nrows() : (S->blk[b].kind() == BLK_TILE ? 1.5 : 1.0)) + 4.0 * double(S->blk[b].nrows()) + 32.0;
    const double total = cost[nblk];
    S->wave_blk0.assign(nwaves + 1, 0);
    int b = 0;
    for (int g = 1; g < nwaves; ++g) {
        const double target = total * double(g) / double(nwaves);
        while (b < nblk && cost[b + 1] <= target) ++b;
        if (b < nblk && (target - cost[b]) > (cost[b + 1] - target)) ++b;     // the closer boundary
        if (b < S->wave_blk0[g - 1]) b = S->wave_blk0[g - 1];
        S->wave_blk0[g] = b;
    }
    S->wave_blk0[nwaves] = nblk;
    S->wave_first.assign(nwaves, BlkDesc{});
    for (int g = 0; g < nwaves; ++g)
        if (S->wave_blk0[g] < S->wave_blk0[g + 1] && !S->blk.empty()) S->wave_first[g] = S->blk[S->wave_blk0[g]];
    S->nwg = nwg;
    S->nwaves = nwaves;
}

// lane group of a 16-byte LDS read (ds_read_b128: {0-3,12-15,20-27}, {4-11,16-19,28-31}, and the same for the upper 32 lanes;
// MI355X_MICROARCH.md, LDS); a 16-byte store is served in groups of 8 contiguous lanes
static inline int win_read_group(int lane) {
    const int l = lane & 31, g = (l < 4 || (l >= 12 && l < 16) || (l >= 20 && l < 28)) ? 0 : 1;
    return g + ((lane >> 5) << 1);
}

// The slot lists in the form the kernels read: own partial first, then a progression of tile partials where it is one.
static void build_def_records(HostBlkCsr& S) {
    S.def_rec.resize(S.def_rows.size());
    for (size_t q = 0; q < S.def_rows.size(); ++q) {
        DefRow& d = S.def_rec[q];
        d = DefRow{};
        d.row = S.def_rows[q];
        int32_t k0 = S.def_ptr[q];
        const int32_t k1 = S.def_ptr[q + 1];
        d.own = -1;
        if (!S.row_defer.empty() && S.row_defer[d.row] >= 0 && k0 < k1 && S.def_idx[k0] == S.row_defer[d.row]) { d.own = S.def_idx[k0]; ++k0; }
        d.count = k1 - k0;
        d.kidx = k0;
        d.base = d.count > 0 ? S.def_idx[k0] : 0;
        d.stride = d.count > 1 ? S.def_idx[k0 + 1] - S.def_idx[k0] : 0;
        for (int32_t k = k0 + 1; k < k1 && d.stride != DEF_EXPLICIT; ++k)
            if (S.def_idx[k] - S.def_idx[k - 1] != d.stride) d.stride = DEF_EXPLICIT;
        if (getenv("FOS_DEF_EXPLICIT")) d.stride = DEF_EXPLICIT;
    }
}

// Window-panel storage of S (fos_internal.hpp, WinPanel).  Returns true when built; false = "not worth it" (the caller falls
// back to row blocks): fewer entries inside the panel-window tiles than half the vector elements the tiles stage (the window
// loads would then cost more than the gathers they replace), or too few panels to fill the GPU.
static bool build_window_panels(int64_t m, int64_t n, const int64_t* colptr, const int64_t* rowval, const double* nzval,
                                bool force, const WinGeomRt& G, bool row_sharded, int cus, HostBlkCsr* out) {
    const int64_t WIN_ROWS = G.rows, WIN_COLS = G.cols;
    const int WIN_WAVES = G.waves, WIN_USL = (int)((WIN_ROWS + 64 * WIN_WAVES - 1) / (64 * WIN_WAVES));
    const int64_t nrows = n + m, nnz = colptr[n] - 1;
    if (nrows >= (int64_t)1 << 31 || 2 * nnz >= ((int64_t)1 << 40)) return false;
    // ---- CSR of S: rows 0..n-1 = columns of A (entries at stacked column n + i), rows n.. = rows of A (stacked column j)
    std::vector<int64_t> rp(nrows + 1, 0);
    for (int64_t j = 0; j < n; ++j) rp[j + 1] = colptr[j + 1] - colptr[j];
    for (int64_t k = 0; k < nnz; ++k) rp[n + rowval[k]] += 1;          // rowval is 1-based: row i -> index n + i + 1 - 1 + 1
    for (int64_t r = 0; r < nrows; ++r) rp[r + 1] += rp[r];
    std::vector<int32_t> ci((size_t)2 * nnz);
    std::vector<double> vv((size_t)2 * nnz);
    {
        std::vector<int64_t> fill(rp.begin() + n, rp.end() - 1);
        for (int64_t j = 0; j < n; ++j) {
            int64_t q = rp[j];
            for (int64_t k = colptr[j] - 1; k < colptr[j + 1] - 1; ++k, ++q) {
                const int64_t i = rowval[k] - 1;
                ci[q] = (int32_t)(n + i); vv[q] = nzval[k];
                const int64_t f = fill[i]++;
                ci[f] = (int32_t)j; vv[f] = nzval[k];
            }
        }
    }
    for (int64_t r = 0; r < nrows; ++r)                    // the window walk needs ascending columns inside every row
        for (int64_t q = rp[r] + 1; q < rp[r + 1]; ++q)
            if (ci[q] <= ci[q - 1]) return false;
    HostBlkCsr& S = *out;
    S.wgeom = G;
    // ---- the panels' row ranges.  A panel whose rows reach into two column ranges that have nothing to do with each other -- the y part
    // and the x part of the stacked vector (rows of A' / rows of A), or two diagonal blocks of a block-diagonal A -- walks the windows
    // of both: twice the segments of its neighbours, and the sweep lasts as long as its slowest panel (C5, 8 blocks: every 16th panel).
    // So the rows are first cut into GROUPS at such boundaries (inside each half of S: row r starts a group when every later row of
    // the half lies to the right of all rows of the current group), consecutive groups are packed into panels while they fit, and a
    // group larger than a panel is cut into equal panels of its own (C5: 16 blocks x 16 tall panels of 3907 rows = one per CU).
    std::vector<std::pair<int64_t, int64_t>> pranges;      // (first row, rows)
    for (int half = 0; half < 2; ++half) {
        const int64_t h0 = half == 0 ? 0 : n, h1 = half == 0 ? n : nrows;
        if (h1 <= h0) continue;
        std::vector<int64_t> smin((size_t)(h1 - h0 + 1), INT64_MAX);      // smallest column of rows >= r of this half
        for (int64_t r = h1 - 1; r >= h0; --r) {
            smin[r - h0] = smin[r - h0 + 1];
            if (rp[r] < rp[r + 1]) smin[r - h0] = std::min<int64_t>(smin[r - h0], ci[rp[r]]);
        }
        std::vector<int64_t> gstart;                      // group starts
        gstart.push_back(h0);
        int64_t pmax = -1;
        for (int64_t r = h0; r < h1; ++r) {
            if (r > gstart.back() && pmax >= 0 && smin[r - h0] != INT64_MAX && pmax < smin[r - h0]) { gstart.push_back(r); pmax = -1; }
            if (rp[r] < rp[r + 1]) pmax = std::max<int64_t>(pmax, ci[rp[r + 1] - 1]);
        }
        gstart.push_back(h1);
        int64_t cur0 = h0, cur = 0;                       // the open panel: rows [cur0, cur0 + cur)
        auto close = [&]() { if (cur > 0) pranges.emplace_back(cur0, cur); cur0 += cur; cur = 0; };
        for (size_t g = 0; g + 1 < gstart.size(); ++g) {
            const int64_t len = gstart[g + 1] - gstart[g];
            if (cur + len <= WIN_ROWS) { cur += len; continue; }
            close();
            if (len <= WIN_ROWS) { cur = len; continue; }
            const int64_t np = (len + WIN_ROWS - 1) / WIN_ROWS, per = (len + np - 1) / np;
            for (int64_t q = 0; q < len; q += per) { cur = std::min<int64_t>(per, len - q); close(); }
        }
        close();
    }
    const int64_t npanel = (int64_t)pranges.size();
    int64_t staged = 0;                                   // vector elements the tiles stage
    std::vector<int32_t> cnt(WIN_ROWS), order(WIN_ROWS), cursor(WIN_ROWS);
    struct SegRow { int32_t i, cnt; int64_t q0; };        // an active row of a segment: local row, entries inside the window, first of them
    struct SegTmp { int64_t col0; int32_t ncols; std::vector<SegRow> rows; };      // rows in descending entry count: slice j = rows 64 j ..
    std::vector<SegTmp> segs;
    std::vector<int32_t> slice_eorder;
    int64_t pos = 0;
    for (int64_t p = 0; p < npanel; ++p) {
        const int64_t r0 = pranges[p].first, R = pranges[p].second;
        // ---- pass 1: the panel's windows -- the span of its columns cut into equal parts of at most WIN_COLS columns (multiples of 8) --,
        //      ascending (every row's entries are sorted by column), and per window that holds entries the active rows sorted by
        //      their entry count (SELL-sigma)
        segs.clear();
        int64_t cmin = -1, cmax = -1;
        for (int64_t i = 0; i < R; ++i) {
            cursor[i] = 0;
            if (rp[r0 + i] < rp[r0 + i + 1]) {
                const int64_t a = ci[rp[r0 + i]], b = ci[rp[r0 + i + 1] - 1];
                if (cmin < 0 || a < cmin) cmin = a;
                if (b > cmax) cmax = b;
            }
        }
        int64_t wwidth = WIN_COLS;
        if (cmin >= 0) {
            cmin -= cmin % 8;                              // (16-byte elements: 128-byte aligned window starts)
            const int64_t span = cmax - cmin + 1, nwin = (span + WIN_COLS - 1) / WIN_COLS;
            wwidth = std::min<int64_t>(WIN_COLS, ((span + nwin - 1) / nwin + 7) / 8 * 8);
        }
        while (cmin >= 0) {
            int64_t wmin = -1;                            // the smallest window with an unconsumed entry
            for (int64_t i = 0; i < R; ++i) {
                const int64_t q = rp[r0 + i] + cursor[i];
                if (q < rp[r0 + i + 1]) { const int64_t w = (ci[q] - cmin) / wwidth; if (wmin < 0 || w < wmin) wmin = w; }
            }
            if (wmin < 0) break;
            const int64_t c0 = cmin + wmin * wwidth, c1 = std::min<int64_t>(c0 + wwidth, nrows);
            int64_t nact = 0;
            for (int64_t i = 0; i < R; ++i) {
                int32_t k = 0;
                for (int64_t q = rp[r0 + i] + cursor[i]; q < rp[r0 + i + 1] && ci[q] < c1; ++q) ++k;
                cnt[i] = k;
                if (k > 0) order[nact++] = (int32_t)i;
            }
            std::stable_sort(order.begin(), order.begin() + nact, [&](int32_t a, int32_t b) { return cnt[a] > cnt[b]; });
            segs.emplace_back();
            SegTmp& sg = segs.back();
            sg.col0 = c0; sg.ncols = (int32_t)(c1 - c0);
            sg.rows.resize((size_t)nact);
            for (int64_t l = 0; l < nact; ++l) { const int32_t i = order[l]; sg.rows[l] = SegRow{i, cnt[i], rp[r0 + i] + cursor[i]}; }
            for (int64_t i = 0; i < R; ++i) cursor[i] += cnt[i];
            staged += c1 - c0;
        }
        // ---- pass 2: one stream per wavefront: segment after segment, its slices j = w, w + WIN_WAVES, ...
        WinPanel wp;
        wp.row0 = (int32_t)r0; wp.nrows = (int32_t)R; wp.seg0 = (int32_t)S.wdesc.size(); wp.nseg = (int32_t)segs.size();
        if (S.wdesc.size() + (size_t)WIN_WAVES * segs.size() >= ((size_t)1 << 31)) return false;
        for (int w = 0; w < WIN_WAVES; ++w) {
            WinWave ww;
            ww.off = pos; ww.slice0 = (int32_t)(S.wrow.size() / 64); ww.pad = 0;
            S.wwave.push_back(ww);
            for (const SegTmp& sg : segs) {
                const int64_t nact = (int64_t)sg.rows.size();
                uint32_t T4[4] = {0, 0, 0, 0};
                for (int u = 0; u < WIN_USL; ++u) {
                    const int64_t s0 = 64 * (int64_t)(w + WIN_WAVES * u);
                    if (s0 >= nact) break;
                    const int64_t ns = std::min<int64_t>(64, nact - s0);
                    const int32_t T = sg.rows[s0].cnt;
                    T4[u] = (uint32_t)T;
                    S.wval.resize((size_t)(pos + 64 * (int64_t)T), 0.0);
                    S.wcol.resize((size_t)(pos + 64 * (int64_t)T), 0);
                    const size_t rbase = S.wrow.size();
                    S.wrow.resize(rbase + 64, (uint16_t)0xFFFF);
                    // Which lane a row of the slice takes, and in which order its entries take the steps, is free -- and decides the LDS bank
                    // conflicts of the walk: a 16-byte LDS access is served in fixed lane groups (reads: four groups of 16 lanes, stores: eight of
                    // 8 contiguous lanes -- win_lane_groups) and is conflict-free when the group's element numbers differ mod 16.  Greedy: rows in
                    // descending entry count take the free lane whose groups hold the fewest rows with the same window-offset residue (step 0)
                    // and the same row-sum residue; then, step by step, every lane takes of its remaining entries the one whose residue its read
                    // group has seen least at that step.  C5: conflict cycles 5.16 M -> 3.76 M per sweep (0.45 -> 0.38 of the LDS cycles), sweep 65.5 -> 64 us;
                    // dealing the rows of equal entry count over their slices by residue as well (so that no slice holds more than four rows per
                    // residue) changed neither (3.72 M): what is left are the row-sum updates and the padding lanes.
                    int lane_of[64];
                    {
                        int cres[4][16] = {}, rres[4][16] = {}, wres[8][16] = {}, nfree[16];      // (write group, read group) cells of 4 lanes: cell = wg * 2 + (rg & 1)
                        for (int c = 0; c < 16; ++c) nfree[c] = 4;
                        bool used[64] = {};
                        for (int64_t l = 0; l < ns; ++l) {
                            const SegRow& sr = sg.rows[s0 + l];
                            const int cr = (int)((ci[sr.q0] - sg.col0) & 15), rr = sr.i & 15;
                            int best = -1, bestcost = INT32_MAX;
                            for (int c = 0; c < 16; ++c) {
                                if (!nfree[c]) continue;
                                const int wg = c >> 1, rg = win_read_group(8 * wg + 4 * (c & 1));
                                const int cost = 4 * cres[rg][cr] + 2 * rres[rg][rr] + 3 * wres[wg][rr];
                                if (cost < bestcost) { bestcost = cost; best = c; }
                            }
                            const int wg = best >> 1, lane0 = 8 * wg + 4 * (best & 1), rg = win_read_group(lane0);
                            int lane = lane0;
                            while (used[lane]) ++lane;
                            used[lane] = true; nfree[best] -= 1;
                            cres[rg][cr] += 1; rres[rg][rr] += 1; wres[wg][rr] += 1;
                            lane_of[l] = lane;
                        }
                    }
                    int sres[4][16];
                    std::vector<int32_t>& eorder = slice_eorder;      // per row of the slice: its entries (offsets from q0) in step order
                    eorder.assign((size_t)ns * (size_t)std::max<int32_t>(T, 1), 0);
                    for (int64_t l = 0; l < ns; ++l)
                        for (int32_t t = 0; t < sg.rows[s0 + l].cnt; ++t) eorder[(size_t)l * T + t] = t;
                    for (int32_t t = 1; t < T; ++t) {
                        for (int g = 0; g < 4; ++g) for (int c = 0; c < 16; ++c) sres[g][c] = 0;
                        for (int64_t l = 0; l < ns; ++l) {
                            const SegRow& sr = sg.rows[s0 + l];
                            if (t >= sr.cnt) continue;
                            const int rg = win_read_group(lane_of[l]);
                            int bestk = t, bestc = INT32_MAX;
                            for (int32_t k = t; k < sr.cnt; ++k) {
                                const int c = sres[rg][(ci[sr.q0 + eorder[(size_t)l * T + k]] - sg.col0) & 15];
                                if (c < bestc) { bestc = c; bestk = k; }
                            }
                            std::swap(eorder[(size_t)l * T + t], eorder[(size_t)l * T + bestk]);
                            sres[rg][(ci[sr.q0 + eorder[(size_t)l * T + t]] - sg.col0) & 15] += 1;
                        }
                    }
                    for (int64_t l = 0; l < ns; ++l) {
                        const SegRow& sr = sg.rows[s0 + l];
                        const int lane = lane_of[l];
                        S.wrow[rbase + lane] = (uint16_t)sr.i;
                        for (int32_t t = 0; t < sr.cnt; ++t) {
                            const int64_t q = sr.q0 + eorder[(size_t)l * T + t];
                            S.wval[(size_t)(pos + 64 * (int64_t)t + lane)] = vv[q];
                            S.wcol[(size_t)(pos + 64 * (int64_t)t + lane)] = (uint16_t)(ci[q] - sg.col0);
                        }
                    }
                    pos += 64 * (int64_t)T;
                    S.wnslice += 1;
                }
                S.wdesc.push_back(WinDesc{(uint32_t)sg.col0, T4[0] | (T4[1] << 16), T4[2] | (T4[3] << 16), (uint32_t)sg.ncols});
            }
        }
        S.wpanel.push_back(wp);
    }
    // worth it?  (i) bytes: the windows the panels stage must not outweigh the gathers they replace; (ii) time, from what the walks were
    // measured to take on MI355X (C5, in-kernel stamps): a sweep lasts as long as its longest panel -- segments x (3.0 us alone on a CU,
    // 4.0 beside a second workgroup, 6.0 for a tall panel's four-times-larger segments) + ~8 us around the walk --, per round of
    // workgroups; the row-block form it competes with ran C5's 2 x 10^7 stored entries in 155 us (gather-bound: time ~ entries).
    int64_t maxseg = 0;
    for (const WinPanel& wp : S.wpanel) maxseg = std::max<int64_t>(maxseg, wp.nseg);
    const int64_t slots = (int64_t)std::max(1, cus) * G.wg_per_cu, rounds = (npanel + slots - 1) / slots;
    const double tseg = G.wg_per_cu == 1 ? 6.0 : (npanel > cus ? 4.0 : 3.0);
    const double t_win = (double)rounds * ((double)maxseg * tseg + 8.0), t_blocks = 2.0 * (double)nnz * (155.0 / 2.0e7);
    const bool worth = 2 * nnz >= staged / 2 && npanel >= 64 && t_win < t_blocks;
    if (!worth && !force) {
        S.wpanel.clear(); S.wwave.clear(); S.wdesc.clear(); S.wnslice = 0; S.wval.clear(); S.wcol.clear(); S.wrow.clear();
        std::vector<double>().swap(S.wval);
        return false;
    }
    S.nnz_padded = pos;
    S.ncol_stored = pos;
    if (row_sharded) {
        // rows of A' are summed over the ranks: the sweep parks its share of row j in slot j (the all-reduce buffer is indexed by column),
        // the deferred-row kernel finishes all n of them from the summed slots (a row without local entries keeps its zero)
        S.row_sharded = true;
        S.row_defer.assign((size_t)nrows, -1);
        S.def_ptr.push_back(0);
        for (int64_t j = 0; j < n; ++j) {
            S.row_defer[j] = rp[j + 1] == rp[j] ? -2 : (int32_t)j;
            S.def_rows.push_back((int32_t)j);
            S.def_idx.push_back((int32_t)j);
            S.def_ptr.push_back((int32_t)S.def_idx.size());
        }
        S.nslots = n;
        build_def_records(S);
    }
    // the kernel's prefetch reads a fixed number of steps per slice, present or not (masked afterwards): one slice of slack
    S.wval.resize((size_t)pos + 64 * 8, 0.0);
    S.wcol.resize((size_t)pos + 64 * 8, 0);
    if (S.wrow.empty()) S.wrow.assign(64, (uint16_t)0xFFFF);
    return true;
}

int build_stacked_csr(int64_t m, int64_t n, const int64_t* colptr, const int64_t* rowval, const double* nzval,
                      int nwg_target, HostBlkCsr* out, int resident_waves, int window_mode, bool row_sharded, int tall_target) {
    if (m < 0 || n < 0) { set_error("negative dimension"); return FOS_EINVAL; }
    if (n + m + 1 > (int64_t)INT32_MAX / 2) { set_error("n+m too large for int32 column indices"); return FOS_EUNSUPPORTED; }
    if (colptr[0] != 1) { set_error("colptr must be 1-based (colptr[1] == 1)"); return FOS_EINVAL; }
    const int64_t nnz = colptr[n] - 1;
    for (int64_t j = 0; j < n; ++j)
        if (colptr[j + 1] < colptr[j]) { set_error("colptr not monotone at column %lld", (long long)j + 1); return FOS_EINVAL; }
    for (int64_t k = 0; k < nnz; ++k)
        if (rowval[k] < 1 || rowval[k] > m) { set_error("rowval[%lld] = %lld out of 1..m", (long long)k + 1, (long long)rowval[k]); return FOS_EINVAL; }

    const int64_t nrows = n + m;
    const bool compress = getenv("FOS_NO_INDEX_COMPRESSION") == nullptr;
    // row-sharded operators (fos_internal.hpp): every row of A' is finished from its partial sum over the ranks, so those rows use
    // the deferred-row machinery -- ONE slot per row without dual tiles; with them (dense rectangles of A stored once) the row's local
    // slot list is added up first (slots_compact_kernel) and the n sums cross the ranks.  No window panels.
    const bool tiles_on = compress && getenv("FOS_NO_TILES") == nullptr &&
                          (!row_sharded || !(getenv("FOS_ROW_SHARDED_TILES") && atoi(getenv("FOS_ROW_SHARDED_TILES")) == 0));

    // ---- rows of A: length, first column, and whether the columns are consecutive ("run": dense blocks, banded rows)
    std::vector<uint8_t> is_run(nrows, 1);
    std::vector<int32_t> first_col(nrows, 0);
    std::vector<int64_t> alen(m, 0);
    {
        std::vector<int64_t> last(m, -1);
        for (int64_t j = 0; j < n; ++j) {                   // columns arrive in ascending order
            for (int64_t k = colptr[j] - 1; k < colptr[j + 1] - 1; ++k) {
                const int64_t i = rowval[k] - 1;
                if (last[i] < 0) first_col[n + i] = (int32_t)j;
                else if (last[i] != j - 1) is_run[n + i] = 0;
                last[i] = j;
                alen[i] += 1;
            }
        }
    }

    // ---- dual tiles: groups of <= 64 consecutive rows of A that are the same run of columns (fos_internal.hpp, BLK_TILE)
    // columns per tile.  Measured on the dense LP (C2, rows of 10 000 entries): 32: 112 us, 64: 100 us, 128: 110 us per KKT
    // apply -- narrower tiles shorten a wavefront's serial chain, but every chunk adds a row-partial slot per row
    int64_t tcmax = TILE_TC_MAX;
    if (getenv("FOS_TILE_TC")) tcmax = std::max(TILE_GROUP, atoi(getenv("FOS_TILE_TC")) / TILE_GROUP * TILE_GROUP);
    struct Group { int64_t i0; int R; int32_t c0; int64_t C; int nchunk; int64_t first_tile; int tall; int sub; };
    // TALL tiles: K vertically adjacent groups over the same run of columns (all but the last 64 rows high) form ONE block, walked
    // by one wavefront, whose column sums are added up over the K sub-tiles before they go to the slot array: K times fewer
    // partial-sum slots and slot-list entries for the columns (C4: 33 -> 9 per column at K = 4).  FOS_TILE_TALL = largest K.
    struct Tall { int first_group; int K; };
    std::vector<Tall> talls;
    std::vector<Group> groups;
    std::vector<int32_t> tile_of(m, -1);
    if (tiles_on) {
        int64_t i = 0;
        while (i < m) {
            if (!is_run[n + i] || alen[i] < TILE_MIN_COLS) { ++i; continue; }
            int64_t j = i + 1;
            while (j < m && j - i < WROWS && is_run[n + j] && alen[j] == alen[i] && first_col[n + j] == first_col[n + i]) ++j;
            // worth it only if the lane-major tile (64 lanes x padded steps) stores no more than the two copies it replaces
            const int64_t nch = (alen[i] + tcmax - 1) / tcmax;
            const int64_t steps = (alen[i] - (nch - 1) * tcmax + TILE_GROUP - 1) / TILE_GROUP * TILE_GROUP + (nch - 1) * tcmax;
            // (round 6: the ragged LAST group of a stack of full tiles over the same columns is a tile whatever its padding -- a block of 136 or
            //  528 rows then consists of tiles only, which is what the resident CG solve needs; one tile more per block)
            const bool continues_stack = !groups.empty() && groups.back().R == 64 && groups.back().i0 + 64 == i &&
                                         groups.back().c0 == first_col[n + i] && groups.back().C == alen[i] && nch == 1;
            if ((j - i >= TILE_MIN_ROWS && 64 * steps <= 2 * (j - i) * alen[i]) || continues_stack) {
                Group g;
                g.i0 = i; g.R = (int)(j - i); g.c0 = first_col[n + i]; g.C = alen[i];
                g.nchunk = (int)((g.C + tcmax - 1) / tcmax);
                g.first_tile = 0; g.tall = 0; g.sub = 0;
                for (int64_t q = i; q < j; ++q) tile_of[q] = (int32_t)groups.size();
                groups.push_back(g);
            }
            i = j;
        }
    }
    const bool have_tiles = !groups.empty();
    {
        // stack height.  Measured (MI355X, outer iterations per second, FOS_TILE_TALL = 1 / 2 / 3 / 4 / 5 / 6 / 8):
        //   C4 (33 groups per block column, ONE column chunk):   517 / 511 / 521 / 483 / 491 / 515 / 496 -- the update kernel gains what
        //      the sweep loses (fewer, longer wavefronts; grids that do not divide evenly over the CUs lose 5 %): K = 1 stays;
        //   C2 (dense LP, 79 groups x 157 column chunks):        196 / 203 / - / 193 / - / - / 212 (16: 159) -- a quarter of the sweep's
        //      slot traffic and of the update kernel's list entries goes away: K = 8.
        // So: stack only operators whose rows are cut into several column chunks (their slot arrays are large), eight high.
        bool multi_chunk = false;
        for (const Group& g : groups) multi_chunk = multi_chunk || g.nchunk > 1;
        int kmax = multi_chunk ? 8 : 1;
        if (const char* e = getenv("FOS_TILE_TALL")) kmax = std::max(1, std::min(TILE_TALL_MAX, atoi(e)));
        size_t g0 = 0;
        while (g0 < groups.size()) {
            size_t g1 = g0 + 1;        // [g0, g1): a stack of groups directly below each other over the same columns
            while (g1 < groups.size() && groups[g1 - 1].R == 64 && groups[g1].i0 == groups[g1 - 1].i0 + 64 &&
                   groups[g1].c0 == groups[g0].c0 && groups[g1].C == groups[g0].C) ++g1;
            const size_t G = g1 - g0, ntall = (G + kmax - 1) / kmax;
            size_t g = g0;
            for (size_t q = 0; q < ntall; ++q) {                 // sizes as equal as possible: G = 33, kmax = 4 -> 4,4,4,4,4,4,3,3,3
                const size_t K = G / ntall + (q < G % ntall ? 1 : 0);
                for (size_t j = 0; j < K; ++j) { groups[g + j].tall = (int)talls.size(); groups[g + j].sub = (int)j; }
                talls.push_back(Tall{(int)g, (int)K});
                g += K;
            }
            g0 = g1;
        }
    }

    // ---- gather-bound operators: window panels instead of row blocks (decided from the operator; FOS_WINDOWS=0/1 forces)
    {
        int mode = window_mode;
        if (const char* e = getenv("FOS_WINDOWS")) mode = atoi(e);
        int64_t runA = 0;
        for (int64_t i = 0; i < m; ++i) runA += (is_run[n + i] && alen[i] > 1) ? 1 : 0;
        const bool candidate = !have_tiles && nnz >= 4 * nrows / 2 && runA < m / 10 && nrows >= 64 * (int64_t)WinStd::ROWS;
        if (mode == 1 || mode == 2 || (mode != 0 && candidate)) {
            // geometry, from what a panel's walk was measured to take on C5 (in-kernel stamps, tools/win_stamps.py): a standard panel 84 us
            // beside a second one on its CU and 63 us alone, a tall one (twice the rows, one per CU) 66 us -- times the rounds of workgroups
            const int cus = tall_target > 0 ? tall_target / 16 : 256;
            const int64_t np_std = (nrows + WIN_GEOM_STD.rows - 1) / WIN_GEOM_STD.rows, np_tall = (nrows + WIN_GEOM_TALL.rows - 1) / WIN_GEOM_TALL.rows;
            const double cost_std = (double)((np_std + 2 * cus - 1) / (2 * cus)) * (np_std > cus ? 84.0 : 63.0);
            const double cost_tall = (double)((np_tall + cus - 1) / cus) * 66.0;
            bool tall = mode == 2 || (mode != 1 && cost_tall < cost_std);
            if (const char* e = getenv("FOS_WIN_GEOM")) tall = atoi(e) == 2;
            *out = HostBlkCsr();
            out->nrows = nrows;
            out->nnz = 2 * nnz;
            if (build_window_panels(m, n, colptr, rowval, nzval, mode == 1 || mode == 2, tall ? WIN_GEOM_TALL : WIN_GEOM_STD, row_sharded, cus, out)) {
                out->nblk = 0; out->nwg = (int32_t)std::min<size_t>(out->wpanel.size(), 16384); out->nwaves = 0;
                out->wave_blk0.assign(1, 0);
                return FOS_OK;
            }
        }
    }

    // ---- plain CSR of what the sweep still stores: rows of A' lose the entries that tiles cover, tile rows of A vanish
    std::vector<int64_t> rp(nrows + 1, 0);
    for (int64_t j = 0; j < n; ++j) {                       // A' rows = A columns: first column / run-ness of what remains
        int64_t cnt = 0, prev = -1;
        for (int64_t k = colptr[j] - 1; k < colptr[j + 1] - 1; ++k) {
            const int64_t i = rowval[k] - 1;
            if (tile_of[i] >= 0) continue;
            if (cnt == 0) first_col[j] = (int32_t)(n + i);
            else if (i != prev + 1) is_run[j] = 0;
            prev = i;
            ++cnt;
        }
        rp[j + 1] = cnt;
    }
    for (int64_t i = 0; i < m; ++i) rp[n + i + 1] = tile_of[i] >= 0 ? 0 : alen[i];
    for (int64_t r = 0; r < nrows; ++r) rp[r + 1] += rp[r];

    // ---- deferred rows: columns of A that tiles cover, rows of A whose tiles are split into column chunks
    HostBlkCsr& S = *out;
    S = HostBlkCsr();
    S.nrows = nrows;
    S.nnz = 2 * nnz;
    std::vector<int32_t> ndef_slots;                        // per row: slots it will sum (0: not deferred)
    std::vector<uint8_t> skip(nrows, 0);                    // rows the ordinary row blocks do not contain
    S.row_sharded = row_sharded;
    if (row_sharded && !have_tiles) {
        S.row_defer.assign(nrows, -1);
        ndef_slots.assign(nrows, 0);
        for (int64_t j = 0; j < n; ++j) {
            ndef_slots[j] = 1;                                   // its own partial, slot j (the all-reduce buffer is indexed by column)
            if (rp[j + 1] == rp[j]) { S.row_defer[j] = -2; skip[j] = 1; }       // no local entries: the slot keeps its zero
            else S.row_defer[j] = (int32_t)j;
        }
        S.nslots = n;
    }
    if (have_tiles) {
        S.row_defer.assign(nrows, -1);
        ndef_slots.assign(nrows, 0);
        for (const Tall& t : talls) {
            const Group& g = groups[t.first_group];
            for (int64_t c = g.c0; c < g.c0 + g.C; ++c) ndef_slots[c] += 1;
        }
        for (const Group& g : groups) {
            for (int q = 0; q < g.R; ++q) {
                skip[n + g.i0 + q] = 1;
                if (g.nchunk > 1) { ndef_slots[n + g.i0 + q] = g.nchunk; S.row_defer[n + g.i0 + q] = -2; }
            }
        }
        for (int64_t j = 0; j < n; ++j) {
            if (ndef_slots[j] == 0 && !row_sharded) continue;    // (row-sharded: EVERY row of A' waits for the other ranks' shares)
            if (rp[j + 1] == rp[j]) { S.row_defer[j] = -2; skip[j] = 1; }
            else { S.row_defer[j] = 0; ndef_slots[j] += 1; }      // slot number assigned below
        }
    }

    // ---- row blocks
    S.row_rel.assign(nrows, 0);
    // where the e-th entry of row r goes: ELL rows (tpr > 0): base + (e / tpr) * 64 + lane0 + e % tpr ; else base + e
    std::vector<int64_t> row_base(nrows, 0), row_cbase(nrows, -1);      // value position / column position (-1: run block)
    std::vector<uint8_t> row_tpr(nrows, 0), row_lane0(nrows, 0);
    auto nextpow2 = [](int64_t v) { int64_t p = 1; while (p < v) p <<= 1; return p; };
    auto align = [](int64_t v) { return (v + NNZ_ALIGN - 1) / NNZ_ALIGN * NNZ_ALIGN; };
    // lane-major (ELL) blocks are accepted while padded <= cnt * (1 + slack) + 32.  Irregular sparse rows are bound by the
    // latency of their random 16-byte gathers, not by HBM: measured on C5 (sprandn, ~20/row) a 60 %-padded ELL block
    // beats the LDS-staged path by 27 %, so up to 2x padding is accepted
    const double ell_slack = getenv("FOS_ELL_SLACK") ? atof(getenv("FOS_ELL_SLACK")) : 1.0;
    // Block size adapts to the operator: a wavefront walks its blocks serially (2 dependent memory round trips per 8
    // lane-major steps), so a SMALL operator wants many small blocks (latency: every resident wavefront gets ~2 of them),
    // a large one the full ELL_MAX (fewer descriptors, more rows per lane sweep).
    int64_t ell_cap = ELL_MAX;
    if (resident_waves > 0) {
        ell_cap = (2 * nnz / (2 * (int64_t)resident_waves) + 63) / 64 * 64;
        ell_cap = std::max<int64_t>(256, std::min<int64_t>(ELL_MAX, ell_cap));
    }
    if (getenv("FOS_ELL_CAP")) ell_cap = std::max(64, atoi(getenv("FOS_ELL_CAP")) / 64 * 64);
    struct TileRec { int32_t blk; int32_t group; int32_t chunk; int32_t tpad; int32_t cslot, rslot; int32_t K; };
    std::vector<TileRec> tiles;
    int64_t pos = 0, cpos = 0;
    int64_t r = 0;
    while (r < nrows) {
        if (r >= n && tile_of[r - n] >= 0) {
            // ---- the tile blocks of this TALL tile (K groups below each other), one per chunk of tcmax columns; sub-tile j of a
            //      block holds rows i0 + 64 j .. and sits at nnz0 + j * 64 * tpad
            Group& g = groups[tile_of[r - n]];
            const Tall& tl = talls[g.tall];
            int64_t rows_total = 0;
            for (int j = 0; j < tl.K; ++j) rows_total += groups[tl.first_group + j].R;
            const int r_last = groups[tl.first_group + tl.K - 1].R;
            for (int j = 0; j < tl.K; ++j) groups[tl.first_group + j].first_tile = (int64_t)tiles.size();
            for (int k = 0; k < g.nchunk; ++k) {
                const int64_t tc = std::min<int64_t>(tcmax, g.C - (int64_t)k * tcmax);
                const int64_t tpad = (tc + TILE_GROUP - 1) / TILE_GROUP * TILE_GROUP;
                BlkDesc d{};
                d.nnz0 = align(pos);
                d.colpos = align(cpos);
                d.cnt = (int64_t)tl.K * 64 * tpad;
                S.tile_tmax = std::max<int32_t>(S.tile_tmax, (int32_t)tpad);
                d.row0 = (int32_t)r;
                d.info = (int32_t)r_last | (BLK_TILE << 8) | (1 << 10) | (tl.K << 11) | ((int32_t)tpad << 16);
                TileRec t;
                t.blk = (int32_t)S.blk.size(); t.group = tl.first_group; t.chunk = k; t.tpad = (int32_t)tpad; t.K = tl.K;
                t.cslot = (int32_t)S.nslots; S.nslots += tpad;
                t.rslot = -1;
                if (g.nchunk > 1) { t.rslot = (int32_t)S.nslots; S.nslots += 64 * (tl.K - 1) + r_last; }
                tiles.push_back(t);
                S.blk.push_back(d);
                pos = d.nnz0 + d.cnt;
                cpos = d.colpos + 4;                // first column, column-slot base, row-slot base, real columns
                S.tile_values += rows_total * tc;
            }
            r += rows_total;
            continue;
        }
        if (skip[r]) { ++r; continue; }
        const int64_t r0 = r;
        const int64_t len0 = rp[r + 1] - rp[r];
        const int64_t blk_start = align(pos), col_start = align(cpos);
        const bool run0 = compress && is_run[r];
        // a block is homogeneous in run-ness; its rows end where the flag flips (or at a row the sweep does not contain)
        int64_t r_lim = r;
        while (r_lim < nrows && (r_lim - r0) < WROWS && !skip[r_lim] && (compress && is_run[r_lim]) == run0) ++r_lim;
        BlkDesc d{};
        d.nnz0 = blk_start;
        d.colpos = col_start;
        d.row0 = (int32_t)r0;
        int64_t ncol_used = 0;
        int64_t stored = -1;                    // values the block occupies (default: d.cnt)
        if (len0 > ELL_MAX) {                   // long row(s): strided by one wavefront
            // consecutive long RUN rows over the SAME column range (a dense block) are processed together, up to
            // LONG_ROWS at a time: the gathered vector element is loaded once per LONG_ROWS matrix values
            int64_t nr = 1;
            if (run0) {
                // (at most 16384 entries per block: a wavefront walks its block serially, and a few very long rows
                // left over next to dual tiles must not become the tail of the whole sweep)
                while (nr < LONG_ROWS && (nr + 1) * len0 <= 16384 && r + nr < r_lim && (rp[r + nr + 1] - rp[r + nr]) == len0 &&
                       first_col[r + nr] == first_col[r]) ++nr;
                if (nr == 3) nr = 2;
            }
            const int64_t stride = align(len0);
            for (int64_t i = 0; i < nr; ++i) row_base[r + i] = blk_start + i * stride;
            d.cnt = len0;
            d.info = (int32_t)nr | (BLK_LONG << 8) | ((run0 ? 1 : 0) << 10);
            stored = nr * stride;
            if (run0) { ncol_used = 1; } else { row_cbase[r] = col_start; ncol_used = len0; }
            r += nr;
        } else {
            // candidate lane-major block: up to WROWS rows / ELL_MAX entries
            int64_t cnt = 0, maxlen = 0, rr = r;
            while (rr < r_lim) {
                const int64_t len = rp[rr + 1] - rp[rr];
                if (rr > r0 && cnt + len > ell_cap) break;
                const int64_t R1 = rr - r0 + 1;
                const int64_t tpr1 = 64 / nextpow2(R1);
                const int64_t ml = std::max(maxlen, len);
                const int64_t T1 = (ml + tpr1 - 1) / tpr1;
                if (rr > r0 && 64 * T1 > std::max<int64_t>(ell_cap, 64 * ((len0 + 63) / 64))) break;
                cnt += len;
                maxlen = ml;
                rr += 1;
            }
            // pick the largest row count (the greedy one, else the powers of two below it) whose lane-major layout
            // wastes at most ~1/6 of the stored entries
            int64_t R = rr - r0, tpr = 1, T = 0, padded = 0;
            bool ell_ok = false;
            for (int64_t Rc = R; Rc >= 1; Rc = (Rc == nextpow2(Rc) ? Rc / 2 : nextpow2(Rc) / 2)) {
                int64_t c2 = 0, ml = 0;
                for (int64_t i = 0; i < Rc; ++i) {
                    const int64_t len = rp[r0 + i + 1] - rp[r0 + i];
                    c2 += len;
                    ml = std::max(ml, len);
                }
                const int64_t tp = 64 / nextpow2(Rc);
                const int64_t Tc = (ml + tp - 1) / tp;
                if ((double)(64 * Tc) <= (double)c2 * (1.0 + ell_slack) + 32.0 && 64 * Tc <= ELL_MAX) {
                    R = Rc; tpr = tp; T = Tc; padded = 64 * Tc; cnt = c2; rr = r0 + Rc;
                    ell_ok = true;
                    break;
                }
                if (Rc == 1) break;
            }
            if (ell_ok) {
                // ---- ELL block: lane = row * tpr + (e % tpr), step = e / tpr
                for (int64_t i = 0; i < R; ++i) {
                    row_base[r0 + i] = blk_start;
                    row_tpr[r0 + i] = (uint8_t)tpr;
                    row_lane0[r0 + i] = (uint8_t)(i * tpr);
                    S.row_rel[r0 + i] = (uint16_t)(rp[r0 + i + 1] - rp[r0 + i]);      // row LENGTH
                    if (!run0) row_cbase[r0 + i] = col_start;
                }
                d.cnt = padded;
                d.info = (int32_t)R | (BLK_ELL << 8) | ((run0 ? 1 : 0) << 10) | ((int32_t)T << 16);
                ncol_used = run0 ? R : padded;
                r = rr;
            } else {
                // ---- irregular rows: LDS-staged block of at most WNNZ entries, always with per-entry indices
                cnt = 0;
                rr = r;
                while (rr < r_lim) {
                    const int64_t len = rp[rr + 1] - rp[rr];
                    if (cnt + len > WNNZ) break;
                    row_base[rr] = blk_start + cnt;
                    row_cbase[rr] = col_start + cnt;
                    S.row_rel[rr] = (uint16_t)cnt;                                      // row START
                    cnt += len;
                    rr += 1;
                }
                if (rr == r) {       // first row alone exceeds WNNZ (but <= ELL_MAX): one row, 64 lanes, lane-major
                    const int64_t T1 = (len0 + 63) / 64;
                    row_base[r] = blk_start;
                    row_tpr[r] = 64;
                    row_lane0[r] = 0;
                    S.row_rel[r] = (uint16_t)len0;
                    if (!run0) row_cbase[r] = col_start;
                    d.cnt = 64 * T1;
                    d.info = 1 | (BLK_ELL << 8) | ((run0 ? 1 : 0) << 10) | ((int32_t)T1 << 16);
                    ncol_used = run0 ? 1 : d.cnt;
                    r += 1;
                } else {
                    d.cnt = cnt;
                    d.info = (int32_t)(rr - r0) | (BLK_LDS << 8);
                    ncol_used = cnt;
                    r = rr;
                }
            }
        }
        S.blk.push_back(d);
        pos = blk_start + (stored >= 0 ? stored : d.cnt);
        cpos = col_start + ncol_used;
    }
    S.nblk = (int32_t)S.blk.size();
    S.ntiles = (int64_t)tiles.size();
    S.nnz_padded = std::max<int64_t>(align(pos), NNZ_ALIGN);
    S.ncol_stored = std::max<int64_t>(align(cpos), NNZ_ALIGN);

    S.val.assign(S.nnz_padded, 0.0);
    S.col.assign(S.ncol_stored, 0);
    // run blocks: one first-column per row; tile blocks: first column of the chunk and the slot bases
    for (const BlkDesc& d : S.blk) {
        if (!d.run() || d.kind() == BLK_TILE) continue;
        for (int i = 0; i < d.nrows(); ++i) S.col[d.colpos + i] = first_col[d.row0 + i];
    }
    for (const TileRec& t : tiles) {
        const BlkDesc& d = S.blk[t.blk];
        S.col[d.colpos] = groups[t.group].c0 + t.chunk * tcmax;
        S.col[d.colpos + 1] = t.cslot;
        S.col[d.colpos + 2] = t.rslot;
        S.col[d.colpos + 3] = (int32_t)std::min<int64_t>(tcmax, groups[t.group].C - (int64_t)t.chunk * tcmax);
        for (int q = 0; q < 4; ++q) S.blk[t.blk].meta[q] = S.col[d.colpos + q];      // the kernels read the descriptor's copy
    }
    auto place = [&](int64_t row, int64_t e) -> int64_t {        // offset of entry e of `row` relative to its block base
        const int64_t tpr = row_tpr[row];
        if (tpr == 0) return e;
        return (e / tpr) * 64 + row_lane0[row] + (e % tpr);
    };
    // ---- fill A' rows (row j of S = column j of A, entries already sorted by row index), minus what tiles cover
    for (int64_t j = 0; j < n; ++j) {
        int64_t e = 0;
        for (int64_t k = colptr[j] - 1; k < colptr[j + 1] - 1; ++k) {
            if (tile_of[rowval[k] - 1] >= 0) continue;
            const int64_t off = place(j, e);
            S.val[row_base[j] + off] = nzval[k];
            if (row_cbase[j] >= 0) S.col[row_cbase[j] + off] = (int32_t)(n + rowval[k] - 1);
            ++e;
        }
    }
    // ---- fill A rows by a counting transpose (column order inside each row = ascending column index); tile rows go
    //      lane-major into their tile: value (row i0 + lane, column c0 + chunk * tcmax + t) at nnz0 + 64 t + lane
    {
        std::vector<int64_t> fill(m, 0);
        for (int64_t j = 0; j < n; ++j) {
            for (int64_t k = colptr[j] - 1; k < colptr[j + 1] - 1; ++k) {
                const int64_t i = rowval[k] - 1;
                if (tile_of[i] >= 0) {
                    const Group& g = groups[tile_of[i]];
                    const int64_t cc = j - g.c0;
                    const TileRec& t = tiles[g.first_tile + cc / tcmax];
                    S.val[S.blk[t.blk].nnz0 + (int64_t)g.sub * 64 * t.tpad + (cc % tcmax) * 64 + (i - g.i0)] = nzval[k];
                    continue;
                }
                const int64_t off = place(n + i, fill[i]++);
                S.val[row_base[n + i] + off] = nzval[k];
                if (row_cbase[n + i] >= 0) S.col[row_cbase[n + i] + off] = (int32_t)j;
            }
        }
    }
    if (row_sharded && !have_tiles) {
        S.def_ptr.push_back(0);
        for (int64_t j = 0; j < n; ++j) {
            S.def_rows.push_back((int32_t)j);
            S.def_idx.push_back((int32_t)j);
            S.def_ptr.push_back((int32_t)S.def_idx.size());
        }
    }
    // ---- slot lists of the deferred rows: [own partial from the sweep] + tile partials in block order
    if (have_tiles) {
        for (int64_t j = 0; j < n; ++j)
            if (S.row_defer[j] == 0) S.row_defer[j] = (int32_t)S.nslots++;
        if (S.nslots > (int64_t)INT32_MAX / 4) { set_error("too many partial slots"); return FOS_EUNSUPPORTED; }
        std::vector<int64_t> lp(nrows + 1, 0);
        for (int64_t q = 0; q < nrows; ++q) lp[q + 1] = lp[q] + ndef_slots[q];
        std::vector<int32_t> idx(lp[nrows], 0);
        std::vector<int64_t> cur(lp.begin(), lp.end() - 1);
        for (int64_t j = 0; j < n; ++j)
            if (S.row_defer[j] >= 0) idx[cur[j]++] = S.row_defer[j];
        for (const TileRec& t : tiles) {
            const Group& g = groups[t.group];
            const int64_t cc0 = g.c0 + (int64_t)t.chunk * tcmax;
            const int64_t tc = std::min<int64_t>(tcmax, g.C - (int64_t)t.chunk * tcmax);
            for (int64_t c = 0; c < tc; ++c) idx[cur[cc0 + c]++] = t.cslot + (int32_t)c;
            if (t.rslot >= 0)
                for (int j = 0; j < t.K; ++j) {
                    const Group& gj = groups[t.group + j];
                    for (int q = 0; q < gj.R; ++q) idx[cur[n + gj.i0 + q]++] = t.rslot + 64 * j + q;
                }
        }
        S.def_ptr.push_back(0);
        for (int64_t q = 0; q < nrows; ++q) {
            if (ndef_slots[q] == 0 && !(row_sharded && q < n)) continue;      // (row-sharded: a row of A' without local entries has an empty list)
            if (cur[q] != lp[q + 1]) { set_error("internal: slot list of row %lld incomplete", (long long)q); return FOS_EINVAL; }
            S.def_rows.push_back((int32_t)q);
            S.def_idx.insert(S.def_idx.end(), idx.begin() + lp[q], idx.begin() + lp[q + 1]);
            S.def_ptr.push_back((int32_t)S.def_idx.size());
        }
    }
    build_def_records(S);
    partition_workgroups(&S, nwg_target);
    return FOS_OK;
}

// Host emulation of the device traversal (same block kinds, same lane/step mapping, same masking, same slot lists):
// out = S * v for the stacked vector v = [vx(n); vy(m)].  Used by the CPU tests to validate the format construction
// without a GPU.
int host_stacked_spmv(const HostBlkCsr& S, const double* v, double* out, std::string* why) {
    if (!S.wpanel.empty()) {
        // window panels: per panel the row sums accumulate window by window; inside a window every wavefront takes its slices from
        // its own stream (running offsets, as the kernel walks them), lane by lane
        auto failw = [&](const char* msg, long long a) { if (why) *why = std::string(msg) + " " + std::to_string(a); return FOS_EINVAL; };
        std::vector<int> covered(S.nrows, 0);
        int64_t next_row = 0;
        const int64_t WIN_ROWS = S.wgeom.rows, WIN_COLS = S.wgeom.cols;
        const int WIN_WAVES = S.wgeom.waves, WIN_USL = (int)((WIN_ROWS + 64 * WIN_WAVES - 1) / (64 * WIN_WAVES));
        if (WIN_WAVES < 1 || WIN_WAVES > 16 || WIN_USL > 4) return failw("bad geometry, wavefronts:", WIN_WAVES);
        if (S.wwave.size() != S.wpanel.size() * WIN_WAVES) return failw("wave table size", (long long)S.wwave.size());
        int64_t slices_seen = 0, values_seen = 0;
        std::vector<double> wslots((size_t)std::max<int64_t>(S.nslots, 1), 0.0);
        std::vector<int> wslot_written((size_t)std::max<int64_t>(S.nslots, 1), 0);
        for (size_t p = 0; p < S.wpanel.size(); ++p) {
            const WinPanel& wp = S.wpanel[p];
            if (wp.row0 != next_row || wp.nrows < 1 || wp.nrows > WIN_ROWS) return failw("bad panel", (long long)p);
            if (wp.seg0 < 0 || wp.nseg < 0 || (size_t)wp.seg0 + (size_t)WIN_WAVES * wp.nseg > S.wdesc.size()) return failw("bad record range in panel", (long long)p);
            next_row += wp.nrows;
            std::vector<double> acc(wp.nrows, 0.0);
            int64_t off[16]; int64_t rs[16];
            for (int w = 0; w < WIN_WAVES; ++w) { off[w] = S.wwave[p * WIN_WAVES + w].off; rs[w] = S.wwave[p * WIN_WAVES + w].slice0; }
            int64_t prev_end = 0;
            for (int32_t k = 0; k < wp.nseg; ++k) {
                const int64_t col0 = S.wdesc[(size_t)wp.seg0 + k].col0, ncols = S.wdesc[(size_t)wp.seg0 + k].ncols;
                if (col0 < prev_end || col0 % 8 || ncols < 1 || ncols > WIN_COLS || col0 + ncols > S.nrows) return failw("bad window in panel", (long long)p);
                prev_end = col0 + ncols;
                std::vector<int> used(wp.nrows, 0);
                for (int w = 0; w < WIN_WAVES; ++w) {
                    const WinDesc& d = S.wdesc[(size_t)wp.seg0 + (size_t)w * wp.nseg + k];
                    if ((int64_t)d.col0 != col0 || (int64_t)d.ncols != ncols) return failw("wavefronts disagree on the window in panel", (long long)p);
                    const uint32_t T4[4] = {d.t01 & 0xFFFFu, d.t01 >> 16, d.t23 & 0xFFFFu, d.t23 >> 16};
                    for (int u = 0; u < 4; ++u) {
                        const int64_t T = T4[u];
                        if (T == 0) { for (int u2 = u + 1; u2 < 4; ++u2) if (T4[u2]) return failw("slice after an empty one in panel", (long long)p); break; }
                        if (u >= WIN_USL || off[w] % 64 || off[w] + 64 * T > (int64_t)S.wval.size() || (size_t)(rs[w] + 1) * 64 > S.wrow.size()) return failw("bad slice in panel", (long long)p);
                        for (int lane = 0; lane < 64; ++lane) {
                            const uint16_t rid = S.wrow[(size_t)rs[w] * 64 + lane];
                            double a = 0.0;
                            for (int64_t t = 0; t < T; ++t) {
                                const size_t e = (size_t)(off[w] + 64 * t + lane);
                                const int c = S.wcol[e];
                                if (c >= ncols) return failw("column outside the window in panel", (long long)p);
                                if (rid == 0xFFFF && S.wval[e] != 0.0) return failw("padding lane with a value in panel", (long long)p);
                                a += S.wval[e] * v[col0 + c];
                            }
                            if (rid == 0xFFFF) continue;
                            if (rid >= wp.nrows || used[rid]++) return failw("row listed twice in a window, panel", (long long)p);
                            acc[rid] += a;
                        }
                        off[w] += 64 * T; rs[w] += 1; slices_seen += 1; values_seen += 64 * T;
                    }
                }
            }
            for (int w = 0; w < WIN_WAVES; ++w) {           // every stream ends where the next one starts
                const size_t nx = p * WIN_WAVES + w + 1;
                const int64_t end_off = nx < S.wwave.size() ? S.wwave[nx].off : S.nnz_padded;
                const int64_t end_rs = nx < S.wwave.size() ? S.wwave[nx].slice0 : S.wnslice;
                if (off[w] != end_off || rs[w] != end_rs) return failw("stream of a wavefront does not end at the next one, panel", (long long)p);
            }
            for (int32_t i = 0; i < wp.nrows; ++i) {
                const int64_t row = wp.row0 + i;
                covered[row]++;
                const int32_t ds = S.row_defer.empty() ? -1 : S.row_defer[row];
                if (ds >= 0) { if (ds >= S.nslots || wslot_written[ds]++) return failw("slot written twice, row", row); wslots[ds] = acc[i]; }
                else if (ds == -2) { if (acc[i] != 0.0) return failw("row marked as having no local entries has some:", row); }
                else out[row] = acc[i];
            }
        }
        for (size_t q = 0; q < S.def_rows.size(); ++q) {    // deferred rows: their slot lists, in list order (one rank: the sum over the ranks is the slot itself)
            double a = 0.0;
            for (int32_t k = S.def_ptr[q]; k < S.def_ptr[q + 1]; ++k) a += wslots[S.def_idx[k]];
            out[S.def_rows[q]] = a;
        }
        if (slices_seen != S.wnslice || values_seen != S.nnz_padded) return failw("slices walked:", slices_seen);
        if (next_row != S.nrows) return failw("panels do not cover all rows:", next_row);
        for (int64_t r = 0; r < S.nrows; ++r) if (covered[r] != 1) return failw("row not covered exactly once:", r);
        return FOS_OK;
    }
    std::vector<int> seen(S.nrows, 0), swept(S.nrows, 0);
    std::vector<double> slots((size_t)std::max<int64_t>(S.nslots, 1), 0.0);
    std::vector<int> slot_written((size_t)std::max<int64_t>(S.nslots, 1), 0);
    const bool defer = !S.row_defer.empty();
    auto fail = [&](const char* msg, long long a) { if (why) *why = std::string(msg) + " " + std::to_string(a); return FOS_EINVAL; };
    if ((int)S.wave_blk0.size() != S.nwaves + 1 || S.wave_blk0.front() != 0 || S.wave_blk0.back() != S.nblk) return fail("bad wave partition", S.nwaves);
    for (int w = 0; w < S.nwaves; ++w) if (S.wave_blk0[w] > S.wave_blk0[w + 1]) return fail("wave partition not monotone at", w);
    // a row finished by the sweep: straight to `out`, or -- deferred rows -- its own partial slot
    auto finish = [&](int64_t row, double acc) {
        swept[row]++;
        if (defer && S.row_defer[row] >= 0) { slots[S.row_defer[row]] = acc; slot_written[S.row_defer[row]]++; }
        else if (defer && S.row_defer[row] == -2) { seen[row] += 100; }          // must never be in an ordinary block
        else { out[row] = acc; seen[row]++; }
    };
    for (int b = 0; b < S.nblk; ++b) {
        const BlkDesc& d = S.blk[b];
        if (d.nnz0 % NNZ_ALIGN) return fail("block not aligned", b);
        const int R = d.nrows();
        if (d.kind() == BLK_TILE) {
            const int T = d.steps(), K = d.tall();
            if (K < 1 || K > TILE_TALL_MAX || d.cnt != (int64_t)K * 64 * T || T % TILE_GROUP || R > 64) return fail("bad tile block", b);
            const int64_t c0 = d.meta[0], cslot = d.meta[1], rslot = d.meta[2], tc = d.meta[3];
            for (int q = 0; q < 4; ++q) if (S.col[d.colpos + q] != d.meta[q]) return fail("tile descriptor and column copy disagree in block", b);
            const int64_t rows_total = 64 * (int64_t)(K - 1) + R;
            if (c0 < 0 || tc < 1 || tc > T || T - tc >= TILE_GROUP || c0 + tc > S.nrows) return fail("tile columns out of range in block", b);
            if (cslot < 0 || cslot + T > S.nslots || (rslot >= 0 && rslot + rows_total > S.nslots)) return fail("tile slots out of range in block", b);
            for (int t = 0; t < T; ++t) {                                        // column sums: sub-tile by sub-tile, over the lanes
                double tot = 0.0;
                for (int j = 0; j < K; ++j) {
                    const int Rj = (j + 1 < K) ? 64 : R;
                    double acc = 0.0;
                    for (int lane = 0; lane < 64; ++lane) {
                        const double a = S.val[d.nnz0 + ((int64_t)j * T + t) * 64 + lane];
                        if (lane >= Rj || t >= tc) { if (a != 0.0) return fail("tile padding not zero in block", b); continue; }
                        acc += a * v[d.row0 + 64 * j + lane];
                    }
                    tot = (j == 0) ? acc : tot + acc;
                }
                slots[cslot + t] = tot;
                slot_written[cslot + t]++;
            }
            for (int j = 0; j < K; ++j) {                                        // row sums: over the steps
                const int Rj = (j + 1 < K) ? 64 : R;
                for (int lane = 0; lane < Rj; ++lane) {
                    double acc = 0.0;
                    for (int t = 0; t < tc; ++t) acc += S.val[d.nnz0 + ((int64_t)j * T + t) * 64 + lane] * v[c0 + t];
                    const int64_t row = d.row0 + 64 * j + lane;
                    if (rslot >= 0) { slots[rslot + 64 * j + lane] = acc; slot_written[rslot + 64 * j + lane]++; }
                    else { out[row] = acc; seen[row]++; }
                }
            }
        } else if (d.kind() == BLK_LONG) {
            const int64_t stride = (d.cnt + NNZ_ALIGN - 1) / NNZ_ALIGN * NNZ_ALIGN;
            for (int i = 0; i < R; ++i) {
                double acc = 0.0;
                for (int64_t k = 0; k < d.cnt; ++k) {
                    const int64_t c = d.run() ? (int64_t)S.col[d.colpos] + k : S.col[d.colpos + k];
                    if (c < 0 || c >= S.nrows) return fail("column out of range in block", b);
                    acc += S.val[d.nnz0 + i * stride + k] * v[c];
                }
                finish(d.row0 + i, acc);
            }
        } else if (d.kind() == BLK_ELL) {
            int p2 = 1; while (p2 < R) p2 <<= 1;
            const int tpr = 64 / p2, T = d.steps();
            if (d.cnt != 64 * (int64_t)T) return fail("ELL count mismatch in block", b);
            for (int i = 0; i < R; ++i) {
                const int len = S.row_rel[d.row0 + i];
                double acc = 0.0;
                for (int lig = 0; lig < tpr; ++lig) {            // the lane sums, then the butterfly (any order on the host)
                    for (int t = 0; t < T; ++t) {
                        const int e = t * tpr + lig;
                        if (e >= len) continue;
                        const int64_t pos = (int64_t)t * 64 + i * tpr + lig;
                        const int64_t c = d.run() ? (int64_t)S.col[d.colpos + i] + e : S.col[d.colpos + pos];
                        if (c < 0 || c >= S.nrows) return fail("column out of range in block", b);
                        acc += S.val[d.nnz0 + pos] * v[c];
                    }
                }
                finish(d.row0 + i, acc);
            }
        } else {
            if (d.cnt > WNNZ) return fail("LDS block too large", b);
            for (int i = 0; i < R; ++i) {
                const int s0 = S.row_rel[d.row0 + i];
                const int e0 = (i + 1 < R) ? (int)S.row_rel[d.row0 + i + 1] : (int)d.cnt;
                double acc = 0.0;
                for (int k = s0; k < e0; ++k) {
                    const int64_t c = S.col[d.colpos + k];
                    if (c < 0 || c >= S.nrows) return fail("column out of range in block", b);
                    acc += S.val[d.nnz0 + k] * v[c];
                }
                finish(d.row0 + i, acc);
            }
        }
    }
    // deferred rows: the second kernel adds their slots in list order
    if (S.def_ptr.size() != S.def_rows.size() + 1 && !(S.def_rows.empty() && S.def_ptr.empty())) return fail("bad deferred-row lists", (long long)S.def_rows.size());
    for (size_t q = 0; q < S.def_rows.size(); ++q) {
        const int64_t row = S.def_rows[q];
        if (row < 0 || row >= S.nrows || !defer || S.row_defer[row] == -1) return fail("deferred row without a flag:", row);
        if (q > 0 && S.def_rows[q - 1] >= row) return fail("deferred rows not ascending at", (long long)q);
        double acc = 0.0;
        if (S.def_rec.size() != S.def_rows.size()) return fail("slot-list records missing", (long long)S.def_rec.size());
        {
            const DefRow& dr = S.def_rec[q];                    // the form the kernels read must spell the same list
            if (dr.row != row || dr.count + (dr.own >= 0 ? 1 : 0) != S.def_ptr[q + 1] - S.def_ptr[q]) return fail("slot-list record of row", row);
            for (int32_t e = 0; e < S.def_ptr[q + 1] - S.def_ptr[q]; ++e) {
                const int32_t k = e - (dr.own >= 0 ? 1 : 0);
                const int32_t sl2 = (dr.own >= 0 && e == 0) ? dr.own : (dr.stride != DEF_EXPLICIT ? dr.base + k * dr.stride : S.def_idx[dr.kidx + k]);
                if (sl2 != S.def_idx[S.def_ptr[q] + e]) return fail("slot-list record disagrees with the list of row", row);
            }
        }
        for (int32_t k = S.def_ptr[q]; k < S.def_ptr[q + 1]; ++k) {
            const int32_t sl = S.def_idx[k];
            if (sl < 0 || sl >= S.nslots) return fail("slot out of range for row", row);
            if (slot_written[sl] != 1 && !(S.row_sharded && S.row_defer[row] == -2 && slot_written[sl] == 0)) return fail("slot not written exactly once, row", row);
            acc += slots[sl];
        }
        out[row] = acc;
        seen[row]++;
    }
    for (int64_t r = 0; r < S.nrows; ++r) if (seen[r] != 1) return fail("row not covered exactly once:", r);
    if (defer)
        for (int64_t r = 0; r < S.nrows; ++r)
            if (S.row_defer[r] >= 0 && swept[r] != 1) return fail("deferred row with entries not swept exactly once:", r);
    return FOS_OK;
}

// The plan of a RESIDENT CG solve (fos_internal.hpp, resident.hip): which workgroup holds which tiles.  Qualifies: an operator that is
// nothing but single-height dual tiles with complete rows (every row of A inside ONE tile: at most 64 columns per unit), whose units --
// the tiles over one run of columns -- have disjoint column ranges that cover all n columns (a block-diagonal A with dense blocks: C4 and
// its shards), and whose largest workgroup share fits the registers: at most 12 tiles of <= 32 steps (one per wavefront), 16 (two per
// wavefront of eight) or 8 tiles of <= 64 steps.
bool build_resident_plan(const HostBlkCsr& S, int64_t m, int64_t n, int gmax, ResPlan* out) {
    *out = ResPlan();
    auto no = [&](const char* why) { out->why = why; out->G = 0; out->wg.clear(); return false; };
    if (!S.wpanel.empty()) return no("window panels");
    if (S.row_sharded) return no("row-sharded operator");
    if (S.nblk <= 0 || (int64_t)S.blk.size() != S.nblk) return no("no row blocks");
    if (gmax < 1) return no("no workgroups");
    struct Unit { int32_t blk0, nblk, c0, tc, T; };
    std::vector<Unit> units;
    int64_t rows = 0;
    int tmax = 0;
    for (int32_t b = 0; b < S.nblk; ++b) {
        const BlkDesc& d = S.blk[b];
        if (d.kind() != BLK_TILE) return no("a row block that is not a dual tile");
        if (d.tall() != 1) return no("tall tiles");
        if (d.meta[2] >= 0) return no("rows spread over column chunks");
        if (d.steps() > 64 || d.steps() < 1) return no("tile wider than 64 steps");
        if (d.row0 < n) return no("tile outside the rows of A");
        rows += d.nrows();
        tmax = std::max(tmax, d.steps());
        if (!units.empty() && units.back().c0 == d.meta[0] && units.back().tc == d.meta[3] && units.back().T == d.steps()) units.back().nblk += 1;
        else units.push_back(Unit{b, 1, d.meta[0], d.meta[3], d.steps()});
    }
    if (rows != m) return no("rows of A outside the tiles");
    {   // disjoint column ranges that cover [0, n)
        std::vector<std::pair<int32_t, int32_t>> rng;
        for (const Unit& u : units) rng.emplace_back(u.c0, u.tc);
        std::sort(rng.begin(), rng.end());
        int64_t at = 0;
        for (const auto& r : rng) {
            if (r.first != at) return no("the units' column ranges overlap or leave columns out");
            at += r.second;
        }
        if (at != n) return no("columns of A outside the units");
    }
    // ---- the STREAMED form (resident.hip, cg_stream_kernel): whole units per workgroup, consecutive ones whose column ranges adjoin, at most 64
    // columns and 70 tiles per workgroup, tiles of at most 32 steps; one workgroup per CU
    auto stream_plan = [&](const char* why_not_registers) -> bool {
        const bool allow = !(getenv("FOS_RESIDENT_STREAM") && atoi(getenv("FOS_RESIDENT_STREAM")) == 0);
        constexpr int RS_GMAX = 256, RS_NCOMP = 7, RS_NT_MAX = 10;
        if (!allow) return no(why_not_registers);
        const int nt_cap = tmax > 32 ? 5 : RS_NT_MAX;          // (64-step tiles: 128 registers of matrix values, five tiles' r, w, x beside them)
        const int gm = std::min(gmax, RS_GMAX);
        const int nu = (int)units.size();
        const int upw = (nu + gm - 1) / gm;
        std::vector<ResWG> wgs;
        int tiles_max = 0;
        for (int u0 = 0; u0 < nu; u0 += upw) {
            const int u1 = std::min(nu, u0 + upw);
            int tcw = 0, tiles = 0;
            for (int u = u0; u < u1; ++u) {
                if (u > u0 && (units[u].c0 != units[u - 1].c0 + units[u - 1].tc || units[u].blk0 != units[u - 1].blk0 + units[u - 1].nblk)) return no(why_not_registers);
                tcw += units[u].tc; tiles += units[u].nblk;
            }
            if (tcw > 64 || tiles > nt_cap * RS_NCOMP + 3) return no(why_not_registers);
            // fewer units than half of the workgroups (a shard of a four-GPU run: 128 blocks on 256 CUs): a unit's tiles are dealt to `split`
            // workgroups, which exchange their column sums as the register form's do
            const int split = upw == 1 ? std::max(1, std::min({4, gm / nu, tiles})) : 1;
            const int wg0 = (int)wgs.size();
            for (int k = 0; k < split; ++k) {
                const int ta = (int)((int64_t)k * tiles / split), tb = (int)((int64_t)(k + 1) * tiles / split);
                wgs.push_back(ResWG{units[u0].blk0 + ta, tb - ta, units[u0].c0, tcw, 32, wg0, split, k});
                tiles_max = std::max(tiles_max, tb - ta);
            }
        }
        // tiles per compute wavefront (resident.hip, rs_split: the communication wavefront walks the last min(nblk % 7, 3) tiles itself)
        int nt = 0;
        for (const ResWG& w : wgs) {
            const int per = w.nblk / RS_NCOMP, r = w.nblk % RS_NCOMP, kc = std::min(r, 3);
            nt = std::max(nt, per + (r - kc > 0 ? 1 : 0));
        }
        if (nt > nt_cap) return no(why_not_registers);
        out->stream = 1; out->nt = nt <= 3 ? 3 : (nt <= 5 ? 5 : (nt <= 9 ? 9 : 10));
        out->nw = RS_NCOMP; out->ncomm = 1; out->rpt = 0; out->tmax = tmax <= 32 ? 32 : 64; out->tiles_wg_max = tiles_max; out->units = nu;
        out->wg = wgs; out->G = (int)wgs.size();
        out->why.clear();
        return true;
    };
    const bool stream_first = getenv("FOS_RESIDENT_STREAM") && atoi(getenv("FOS_RESIDENT_STREAM")) == 2;       // (tests: the streamed form where the register form would do)
    if (stream_first) return stream_plan("the streamed form was asked for and does not fit");
    if ((int64_t)units.size() > gmax) return stream_plan("more units than workgroups");
    if (gmax > RES_GMAX) gmax = RES_GMAX;
    // tiles per workgroup: the smallest tp whose workgroup count fits
    int64_t total = S.nblk;
    int tp = (int)std::max<int64_t>(1, (total + gmax - 1) / gmax);
    for (;; ++tp) {
        int64_t G = 0;
        bool ok = true;
        for (const Unit& u : units) {
            const int w = (u.nblk + tp - 1) / tp;
            if (w > RES_WPU_MAX) ok = false;
            G += w;
        }
        if (ok && G <= gmax) break;
        if (tp > RES_SLOTS) return stream_plan("a workgroup would hold more tiles than its registers");
    }
    int tiles_max = 0;
    for (const Unit& u : units) {
        const int w = (u.nblk + tp - 1) / tp;
        for (int k = 0; k < w; ++k) {
            const int t0 = (int)((int64_t)k * u.nblk / w), t1 = (int)((int64_t)(k + 1) * u.nblk / w);
            tiles_max = std::max(tiles_max, t1 - t0);
        }
    }
    // what a workgroup can hold: kernel <32 steps, 1 tile per wavefront, 12 wavefronts> up to 11 compute wavefronts (+ 1..3 that communicate);
    // <32, 2, 8> / <32, 3, 8> seven compute wavefronts of two / three tiles + one that communicates; <64, 1, 8> seven of one tile + one
    int nw, rpt, ncomm;
    if (tmax <= 32 && tiles_max <= 11) { nw = tiles_max; rpt = 1; ncomm = std::min(3, 12 - nw); }
    else if (tmax <= 32 && tiles_max <= 14) { nw = 7; rpt = 2; ncomm = 1; }
    else if (tmax <= 32 && tiles_max <= 21) { nw = 7; rpt = 3; ncomm = 1; }          // (a shard of a four-GPU run of C4: 17 tiles per workgroup)
    else if (tiles_max <= 7) { nw = tiles_max; rpt = 1; ncomm = 1; }
    else return stream_plan("a workgroup's tiles do not fit the registers");
    out->ncomm = ncomm;
    out->tmax = tmax <= 32 ? 32 : 64;
    out->nw = nw; out->rpt = rpt; out->tiles_wg_max = tiles_max; out->units = (int)units.size();
    for (const Unit& u : units) {
        const int w = (u.nblk + tp - 1) / tp;
        const int wg0 = (int)out->wg.size();
        for (int k = 0; k < w; ++k) {
            const int t0 = (int)((int64_t)k * u.nblk / w), t1 = (int)((int64_t)(k + 1) * u.nblk / w);
            out->wg.push_back(ResWG{u.blk0 + t0, t1 - t0, u.c0, u.tc, u.T, wg0, w, k});
        }
    }
    out->G = (int)out->wg.size();
    return true;
}

// Host emulation of the resident CG solve (resident.hip): the same plan walked workgroup by workgroup -- tiles out of the tile storage,
// a workgroup's partial column sums, the unit's totals over its workgroups, the four sums of an iteration as per-workgroup partials (rows by
// their lanes, a unit's columns once, by its first workgroup, the bilinear share of w.r by whoever holds the partial column sum) -- and the
// merged-reduction recurrence around it.  Checks the plan and the arithmetic without a GPU (CPU tests); not a product path.
int host_resident_cg(const HostBlkCsr& S, const ResPlan& P, int64_t m, int64_t n, const double* cb, double2* x, const double2* rhs, const double2* v0,
                     double tol, int maxit, int* iters) {
    const int64_t nm = n + m, l = nm + 1;
    std::vector<double2> r((size_t)l), p((size_t)l, double2{0.0, 0.0}), sv((size_t)l, double2{0.0, 0.0}), w((size_t)l), g(v0, v0 + l);
    std::vector<std::vector<double2>> colpart(P.wg.size());
    auto sweep = [&](const double2 gt, double (&tot)[4]) {
        tot[1] = tot[2] = tot[3] = 0.0;
        for (size_t q = 0; q < P.wg.size(); ++q) {
            const ResWG& me = P.wg[q];
            double acc[4] = {0.0, 0.0, 0.0, 0.0};
            std::vector<double2>& cp = colpart[q];
            cp.assign((size_t)me.tc, double2{0.0, 0.0});
            for (int ti = 0; ti < me.nblk; ++ti) {
                const BlkDesc& d = S.blk[me.blk0 + ti];
                for (int lane = 0; lane < d.nrows(); ++lane) {
                    const int64_t row = d.row0 + lane;
                    double u1 = 0.0, u2 = 0.0;
                    const int tc0 = d.meta[0], tcn = d.meta[3];          // the tile's own columns (streamed form: a part of the workgroup's)
                    for (int t = 0; t < tcn; ++t) {
                        const double a = S.val[d.nnz0 + 64 * (int64_t)t + lane];
                        u1 += a * g[tc0 + t].x; u2 += a * g[tc0 + t].y;
                        cp[tc0 - me.c0 + t].x += a * g[row].x; cp[tc0 - me.c0 + t].y += a * g[row].y;
                    }
                    const double c = cb[row];
                    const double q1 = -(u1 - gt.x * c), q2 = -(u2 - gt.y * c);
                    w[row] = double2{g[row].x - q2, q1 - g[row].y};
                    acc[1] += w[row].x * g[row].x + w[row].y * g[row].y;
                    acc[2] += c * g[row].x; acc[3] += c * g[row].y;
                }
            }
            for (int t = 0; t < me.tc; ++t) {
                const double2 gc = g[me.c0 + t];
                const double cc = cb[me.c0 + t];
                acc[1] += cp[t].x * gc.y - cp[t].y * gc.x;
                if (me.idx == 0) {
                    acc[1] += (gc.x * gc.x - gc.y * gc.y) + cc * (gt.x * gc.y - gt.y * gc.x);
                    acc[2] += cc * gc.x; acc[3] += cc * gc.y;
                }
            }
            for (int k = 1; k < 4; ++k) tot[k] += acc[k];
        }
        for (size_t q = 0; q < P.wg.size(); ++q) {              // a unit's column sums: its workgroups in order
            const ResWG& me = P.wg[q];
            if (me.idx != 0) continue;
            for (int t = 0; t < me.tc; ++t) {
                double2 ct{0.0, 0.0};
                for (int k = 0; k < me.wpu; ++k) { ct.x += colpart[me.wg0 + k][t].x; ct.y += colpart[me.wg0 + k][t].y; }
                const double2 gc = g[me.c0 + t];
                const double cc = cb[me.c0 + t];
                const double q1 = ct.x + gt.x * cc, q2 = ct.y + gt.y * cc;
                w[me.c0 + t] = double2{gc.x - q2, q1 - gc.y};
            }
        }
    };
    auto rr_sum = [&]() { double sacc = 0.0; for (int64_t i = 0; i < nm; ++i) sacc += r[i].x * r[i].x + r[i].y * r[i].y; return sacc; };
    double tot[4] = {0.0, 0.0, 0.0, 0.0};
    double2 gt = g[nm];
    sweep(gt, tot);
    {
        const double2 wt{gt.x + tot[3], -tot[2] - gt.y};
        for (int64_t i = 0; i < nm; ++i) r[i] = double2{rhs[i].x - w[i].x, rhs[i].y - w[i].y};
        r[nm] = double2{rhs[nm].x - wt.x, rhs[nm].y - wt.y};
    }
    double g_prev = 0.0, a_prev = 0.0;
    int it = 0;
    for (int i = 0;; ++i) {
        g = r;
        gt = r[nm];
        tot[0] = rr_sum();
        sweep(gt, tot);
        const double gam = tot[0] + (gt.x * gt.x + gt.y * gt.y);
        if (i > 0 && (std::sqrt(gam) <= tol || i >= maxit)) { it = i; break; }
        const double2 wt{gt.x + tot[3], -tot[2] - gt.y};
        w[nm] = wt;
        const double delta = tot[1] + (wt.x * gt.x + wt.y * gt.y);
        double beta = 0.0, alpha;
        if (i == 0) alpha = gam / delta;
        else { beta = gam / g_prev; alpha = gam / (delta - beta * gam / a_prev); }
        g_prev = gam; a_prev = alpha;
        for (int64_t k = 0; k < l; ++k) {
            if (i == 0) { p[k] = r[k]; sv[k] = w[k]; }
            else {
                p[k].x = p[k].x * beta + r[k].x; p[k].y = p[k].y * beta + r[k].y;
                sv[k].x = sv[k].x * beta + w[k].x; sv[k].y = sv[k].y * beta + w[k].y;
            }
            x[k].x += alpha * p[k].x; x[k].y += alpha * p[k].y;
            r[k].x -= alpha * sv[k].x; r[k].y -= alpha * sv[k].y;
        }
    }
    *iters = it;
    return FOS_OK;
}

}  // namespace fos
