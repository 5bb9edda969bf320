// Internal declarations of libfoship (not part of the ABI).
//
// Device data layout (DESIGN.md "Data layout in HBM"):
//   * every N-vector of the reference (N = 2l, l = n+m+1, layout [x;y;tau;r;s;kappa]) lives on the device
//     as l `double2` elements, INTERLEAVED:  v[i] = (part1[i], part2[i])  with part1 = [x;y;tau],
//     part2 = [r;s;kappa].  All CG / relaxation kernels are elementwise or reductions, so the layout is
//     free; interleaving makes the dual-right-hand-side KKT SpMV gather ONE 16-byte element per non-zero and
//     lets one cone workgroup project the primal and the dual copy of a cone from the same loads.
//   * the operator is stored once as the stacked matrix S = [[0, A'],[A, 0]] ((n+m) x (n+m)) in a
//     block-padded CSR (fp64 values, int32 columns into the stacked vector [x(n); y(m)]).
#pragma once

#include <hip/hip_runtime.h>

#include <cstddef>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "foship.h"

namespace fos {

// ---------------------------------------------------------------------------------- errors
void set_error(const char* fmt, ...);

#define FOS_HIP(expr)                                                                         \
    do {                                                                                      \
        hipError_t _e = (expr);                                                               \
        if (_e != hipSuccess) {                                                               \
            ::fos::set_error("%s:%d: %s -> %s", __FILE__, __LINE__, #expr, hipGetErrorString(_e)); \
            return FOS_EHIP;                                                                  \
        }                                                                                     \
    } while (0)

#define FOS_TRY(expr)                    \
    do {                                 \
        int _r = (expr);                 \
        if (_r != FOS_OK) return _r;     \
    } while (0)

// ---------------------------------------------------------------------------------- SpMV storage
constexpr int SPMV_THREADS = 256;   // 4 independent wavefronts per workgroup
constexpr int SPMV_WAVES = SPMV_THREADS / 64;
constexpr int WNNZ = 256;           // LDS-staged row block: max non-zeros (16 B each for 2 RHS in the wavefront's LDS slice)
constexpr int ELL_MAX = 2048;       // lane-major ("ELL") row block: max padded entries (a multiple of 64)
constexpr int WROWS = 64;           // max rows per row block (>= 1 epilogue lane per row)
constexpr int LONG_ROWS = 4;        // long run-rows over one column range handled together by a wavefront
constexpr int NNZ_ALIGN = 4;        // every row block starts at a multiple of 4 entries (16/32-byte aligned)

enum BlkKind : int32_t { BLK_LDS = 0, BLK_ELL = 1, BLK_LONG = 2, BLK_TILE = 3 };

// Dual tiles (BLK_TILE).  A dense rectangle of A -- R <= 64 consecutive rows whose non-zeros are the SAME run of
// consecutive columns -- is stored ONCE, lane-major (lane = row, step = column), and serves both products of the
// stacked operator in one pass: the row sums (A x) stay in the lanes, the column sums (A'y) are formed by an
// in-register butterfly over 8 steps at a time and written as PARTIALS to a slot array; the A' entries the tile
// covers are not stored at all.  Rows of S whose sum is spread over several tiles (columns of A covered by tiles;
// rows of A wider than TILE_TC_MAX columns) are DEFERRED: a second small kernel adds their partial slots in a fixed
// order and runs the row epilogue.  Halves the matrix bytes of dense operators (C2, C4).
constexpr int TILE_MIN_ROWS = 16;   // fewer rows: not worth a 64-lane wavefront
constexpr int TILE_MIN_COLS = 8;
constexpr int TILE_TC_MAX = 64;     // columns (steps) per tile: 32 KB of values per wavefront work unit (measured best on C2: 32..128 tried)
constexpr int TILE_GROUP = 8;       // steps per butterfly; stored steps are padded to a multiple
constexpr int TILE_TALL_MAX = 16;   // sub-tiles a tall tile may stack (5 bits in the descriptor)

// Window panels (gather-bound random-sparse operators, C5 class).  A PANEL = WIN_ROWS consecutive rows of S, the unit of
// work of ONE WORKGROUP (two of them share a CU), which walks the column WINDOWS (WIN_COLS consecutive entries of the stacked vector) its rows touch:
// the window is staged in LDS by coalesced loads, the panel's non-zeros inside it are multiplied against LDS (no 128-byte L1
// line fill per 16-byte gather: the bound of the row-block formats on such operators), and the row sums are kept in LDS across
// the windows.  Inside a (panel, window) SEGMENT the rows that have entries there are sorted by their entry count (SELL-sigma)
// and cut into SLICES of 64 rows stored lane-major (lane = row, step = entry; padded to the slice's longest row: ~10 %, not the
// 40-50 % of panel-wide ELL), with 16-bit column offsets inside the window and a 16-bit local row id per lane.
// Two geometries (round 4), chosen per operator at fos_create (csr_build.cpp, window_geometry):
//   WinStd   2016-row panels, windows of <= 3072 columns, 512 threads, TWO workgroups per CU (2 x 80 KB of LDS);
//   WinTall  4032-row panels, windows of <= 6144 columns, 1024 threads, ONE workgroup per CU (160 KB): four times the entries per
//            segment against the same fixed cost of a segment (its memory latency, two barriers) and half the window bytes per entry
//            -- C5 sweep 96.6 -> 89.2 us -- for operators with enough rows to give (nearly) every CU a panel.
struct WinStd {
    static constexpr int COLS = 3072, ROWS = 2016, THREADS = 512, WG_PER_CU = 2, WAVES = THREADS / 64;
    // steps of a wavefront's slice u (descending step counts) requested with the window, so that one memory latency covers the window
    // and the matrix; the rest -- and, here, a fourth slice -- is streamed behind the barriers
    static constexpr int PRE0 = 6, PRE1 = 6, PRE2 = 6, PRE3 = 0;
    static constexpr bool PREFETCH = false;
    static constexpr bool PIPELINE = false;
};
struct WinTall {
    static constexpr int COLS = 6144, ROWS = 4032, THREADS = 1024, WG_PER_CU = 1, WAVES = THREADS / 64;
    static constexpr int PRE0 = 8, PRE1 = 4, PRE2 = 2, PRE3 = 2;
#ifdef FOS_WIN_PREFETCH
    static constexpr bool PREFETCH = FOS_WIN_PREFETCH != 0;   // the next segment's window requested behind this segment's second barrier (one workgroup per CU: nobody else fills the bubble)
#else
    static constexpr bool PREFETCH = true;
#endif
#ifdef FOS_WIN_PIPELINE
    static constexpr bool PIPELINE = FOS_WIN_PIPELINE != 0;   // slot-level software pipeline over the segments (kernels.hip, win_walk)
#else
    static constexpr bool PIPELINE = true;
#endif
};
static_assert((WinStd::ROWS + 64 * WinStd::WAVES - 1) / (64 * WinStd::WAVES) <= 4 && (WinTall::ROWS + 64 * WinTall::WAVES - 1) / (64 * WinTall::WAVES) <= 4,
              "a WinDesc holds the step counts of four slices per wavefront");
struct WinGeomRt { int32_t rows, cols, waves, wg_per_cu; };       // the same numbers for the host-side builder / emulation
constexpr WinGeomRt WIN_GEOM_STD = {WinStd::ROWS, WinStd::COLS, WinStd::WAVES, WinStd::WG_PER_CU};
constexpr WinGeomRt WIN_GEOM_TALL = {WinTall::ROWS, WinTall::COLS, WinTall::WAVES, WinTall::WG_PER_CU};
// Storage = WAVE STREAMS (round 4).  Slice j of a segment (slices in descending step count) belongs to wavefront j % WIN_WAVES; everything
// one wavefront reads of a panel -- its slices' values / column offsets / row words, segment after segment -- is CONTIGUOUS, so the
// wavefront walks it with a running offset, and what it has to be told per segment (the window and the step counts of its <= 4
// slices) is one 16-byte record, all records of a panel loaded in ONE request at the panel's start (lane k holds segment k's record,
// read back by v_readlane): no dependent descriptor load sits in front of a segment's memory requests any more (the round-3 form paid
// two scalar round trips -- segment, then slice descriptor -- before the slice loads of EVERY segment could be issued).
struct WinPanel { int32_t row0, nrows, seg0, nseg; };      // seg0: first record of wavefront 0 in wdesc; wavefront w: seg0 + w * nseg
struct WinWave { int64_t off; int32_t slice0, pad; };       // per (panel, wavefront): first value / column offset (multiple of 64), first row-word slice
struct WinDesc { uint32_t col0, t01, t23, ncols; };         // per (panel, wavefront, segment): the window [col0, col0 + ncols) of the stacked vector; steps of the
                                                            // wavefront's slices u = 0..3 (16 bits each, 0: no such slice; descending).  Values at off + 64 t + lane,
                                                            // then the next slice.  Windows are cut PER PANEL: the span of the panel's columns in equal parts of at most
                                                            // COLS columns (C5: every panel of a block gets 21 -- tall: 11 -- equally filled segments; a global grid gave
                                                            // some panels one more, nearly empty one, and the sweep lasts as long as its slowest panel).


// One slot-spread row and its list of partial-sum slots, in summation order: [its own partial `own` (>= 0) from the sweep] followed
// by `count` tile partials, which for regular tilings form an arithmetic progression base + k * stride (dense blocks: one slot
// per tile below each other) -- then the kernels compute the slot numbers instead of loading them (one memory round trip less
// in the kernels that add the lists: they sit on the latency chain of every CG iteration); stride == DEF_EXPLICIT: the slot
// numbers are def_idx[kidx + k].
constexpr int32_t DEF_EXPLICIT = INT32_MIN;
struct DefRow { int32_t row, own, base, stride, count, kidx, pad0, pad1; };     // 32 bytes

struct BlkDesc {                    // one row block = the unit of work of ONE wavefront (48 bytes, wave-uniform: read by scalar loads)
    int64_t nnz0;                   // first value in `val` (multiple of NNZ_ALIGN)
    int64_t colpos;                 // first entry in `col`: per-entry column indices, or -- for a RUN block, whose rows
                                    // all have consecutive columns -- one first-column per row (index compression)
    int64_t cnt;                    // stored values (ELL: 64 * steps, padding included; LONG: entries per row, rows at stride align4(cnt))
    int32_t row0;                   // first row
    int32_t info;                   // nrows (bits 0..7; dual tile: rows of its LAST sub-tile) | kind << 8 | run << 10 | sub-tiles of a
                                    // tall dual tile << 11 (5 bits) | ELL steps T << 16
    int32_t meta[4];                // dual tile: first column, column-slot base, row-slot base (-1: none), real columns -- in the
                                    // descriptor so that a tile costs ONE dependent load before its values, not two (0 otherwise)
    __host__ __device__ int nrows() const { return info & 0xFF; }
    __host__ __device__ int kind() const { return (info >> 8) & 0x3; }
    __host__ __device__ int run() const { return (info >> 10) & 0x1; }
    __host__ __device__ int steps() const { return (info >> 16) & 0xFFFF; }
    __host__ __device__ int tall() const { return (info >> 11) & 0x1F; }
};

// Host-side result of building the stacked operator; uploaded verbatim.
struct HostBlkCsr {
    int64_t nrows = 0, nnz = 0, nnz_padded = 0, ncol_stored = 0;
    std::vector<double> val;
    std::vector<int32_t> col;
    std::vector<BlkDesc> blk;          // [nblk]
    std::vector<uint16_t> row_rel;     // [nrows]  row start relative to its block's nnz0 (stream blocks)
    std::vector<int32_t> wave_blk0;    // [nwaves+1]  row blocks owned by each wavefront of the grid
    std::vector<BlkDesc> wave_first;   // [nwaves]    blk[wave_blk0[w]] (zero for a wavefront without blocks)
    int32_t tile_tmax = 0;             // longest dual tile, in steps (0: no tile)
    int32_t nblk = 0, nwg = 0, nwaves = 0;
    // dual tiles / deferred rows (empty when the operator has no tile)
    int64_t nslots = 0;                // partial-sum slots (2 doubles each)
    int64_t ntiles = 0, tile_values = 0;
    std::vector<int32_t> row_defer;    // [nrows] -1: epilogue in the sweep; >= 0: the sweep stores the row's own partial in that
                                       // slot; -2: deferred, no entries outside tiles (the sweep never sees the row)
    std::vector<int32_t> def_rows;     // [ndef]   deferred rows, ascending
    std::vector<int32_t> def_ptr;      // [ndef+1] their slot lists
    std::vector<int32_t> def_idx;      // slot indices, in summation order
    std::vector<DefRow> def_rec;       // [ndef]   the same lists in the form the kernels read (progressions where they are ones)
    bool row_sharded = false;          // every row of A' is deferred: the single slot `row`, or -- with dual tiles -- its local slot list (summed over the ranks before use)
    // window panels (empty unless the operator is stored that way; then blk is empty and val/col are unused)
    std::vector<WinPanel> wpanel;
    WinGeomRt wgeom = WIN_GEOM_STD;    // geometry the panels were built for
    std::vector<WinWave> wwave;        // [npanel * waves]
    std::vector<WinDesc> wdesc;        // [sum over panels of waves * nseg]
    int64_t wnslice = 0;               // slices stored
    std::vector<double> wval;
    std::vector<uint16_t> wcol;        // column offset inside the window, per stored entry
    std::vector<uint16_t> wrow;        // [nslice * 64] local row of each lane (0xFFFF: no row)
};

struct DevBlkCsr {
    int64_t nrows, nnz, nnz_padded;
    const double* val;
    const int32_t* col;
    const BlkDesc* blk;
    const uint16_t* row_rel;
    const int32_t* wave_blk0;
    const BlkDesc* wave_first;         // [nwaves] copy of each wavefront's first descriptor (requested together with wave_blk0)
    int32_t resident;                  // 1: the whole operator stays in the caches from sweep to sweep -- ordinary loads for the matrix stream
    int32_t dbg_flags;                 // timing experiments only (FOS_DBG_FLAGS; WRONG results): 1 = dual tiles skip the column-sum butterflies;
                                       // cg_update_kernel: 4 = slot-spread rows left unfinished, 16 = scalar prologue only, 32 = no prologue
                                       // (sums taken from the first record); bits 8-10 = elements requested in front of its prologue (1..4)
    int32_t nblk, nwg, nwaves;
    // dual tiles / deferred rows (ndef == 0: none)
    double* slots;                     // [nslots][2]   written by the sweeps
    const double* slots_rd;            // what the slot-list sums read: == slots, or -- row-sharded operators -- the all-reduced copy
                                       // (with dual tiles: [n summed rows of A' | the sweep's slot array], solver.cpp cmp_rec)
    const int32_t* row_defer;          // nullptr when ndef == 0
    const int32_t* def_rows;
    const int32_t* def_ptr;
    const int32_t* def_idx;
    const DefRow* def_rec;
    int32_t ndef, nwg_def;             // deferred rows, workgroups of the deferred-row kernel
    int32_t def_lpr;                   // lanes per deferred row (power of two <= 64)
    int32_t npart, part_off;           // the partial-sum records a sweep leaves for its consumers: `npart` records starting at
                                       // record `part_off` (no tiles: the sweep's nwg; tiles: the deferred-row kernel folds the
                                       // sweep's records into its own nwg_def, stored behind them)
    // window panels (npanel == 0: row-block storage)
    int32_t npanel;
    int32_t win_tall;                  // 1: WinTall geometry (else WinStd)
    const WinPanel* wpanel;
    const WinWave* wwave;
    const WinDesc* wdesc;
    const double* wval;
    const uint16_t* wcol;
    const uint16_t* wrow;
};

// Builds S = [[0, A'],[A, 0]] from Julia CSC (1-based).  Returns FOS_EINVAL on malformed input.
int build_stacked_csr(int64_t m, int64_t n, const int64_t* colptr, const int64_t* rowval, const double* nzval,
                      int nwg_target, HostBlkCsr* out, int resident_waves = 0, int window_mode = -1,   // -1: decide from the operator, 0: no window panels, 1 / 2: force WinStd / WinTall
                      bool row_sharded = false, int tall_target = 0);   // tall_target: tile blocks that fit one resident round (0: 4096)
void partition_workgroups(HostBlkCsr* S, int nwg_target);
int host_stacked_spmv(const HostBlkCsr& S, const double* v, double* out, std::string* why);
constexpr int DEF_THREADS = 256;    // deferred-row kernel: one thread per deferred row
constexpr int DEF_MAX_WG = 1024;    // its grid (grid-stride beyond)
constexpr int PART_CAP = 16384 + DEF_MAX_WG + 8;   // per-workgroup partial-sum records a kernel may produce (sweep + deferred rows)

// ---------------------------------------------------------------------------------- device scalar state
// One struct in device memory; kernels read/write it, the host polls a pinned copy.
struct DevState {
    // CG  (conjugategradients.jl:31-55)
    double rn, rn_old, alpha, beta, pAp, rr, tol;
    double rn2[2];         // r.r ping-pong: CG iteration j reads rn2[j&1] and writes rn2[(j+1)&1] (lets every workgroup of the
                           // fused finalize+p-update kernel derive beta itself without racing the one that stores it)
    int32_t iter, done;    // (adjacent and 8-byte aligned: cg_pupdate_kernel reads the pair with ONE load)
    int32_t maxit, hit_max;
    int32_t xchg_failed;   // a peer-mailbox exchange timed out (PeerBox below); every later exchange is skipped
    int32_t bar_failed;    // a wait between workgroups inside a kernel timed out (cg_update_kernel's producer flags)
    // algorithm scalars
    double alpha12;        // GAPAData.alpha12 (gapa.jl:29,101)
    double gapa_scl;       // last normedScalar value (diagnostic)
    // status sums, filled by status_finalize: see StatusIdx
    double stat[12];
    // host-visible (pinned, mapped) record the kernel that ends a CG solve fills: lets the host see the end of the solve
    // without a stream synchronisation (HostMark below; 0: none)
    unsigned long long hostmark;
    // merged-reduction CG (cgm_update_kernel): the step lengths ping-pong like r.r (iteration j reads alpha2[j & 1], written by
    // the launch before, and writes alpha2[(j + 1) & 1]); vtau = the tau element of the vector the last sweep applied M to
    // (stashed by the sweep: the update kernel overwrites that vector while other workgroups still need the element)
    double alpha2[2];
    double vtau[2];
    int32_t dbg_delay;     // test hook (fos_debug_set): workgroups != 0 of cg_pupdate_kernel wait this many 100 MHz ticks at entry
    int32_t pad_;
};
static_assert(offsetof(DevState, iter) % 8 == 0 && offsetof(DevState, done) == offsetof(DevState, iter) + 4, "iter/done are read as one 64-bit word");
struct HostMark { uint32_t seq; int32_t iter; int32_t hit_max; int32_t batch; double rr; };   // seq = the solve's number, written last;
                                                                                            // batch: id of a batch of iterations that ended WITHOUT convergence

enum StatusIdx { ST_RP2 = 0, ST_RD2, ST_CTX, ST_BTY, ST_AXS2, ST_ATY2, ST_TAU, ST_KAPPA, ST_COUNT };

// ---------------------------------------------------------------------------------- cone tables
// Elementwise op per (index, part):  2 bits each.
enum EwOp : uint8_t { EW_COPY = 0, EW_ZERO = 1, EW_MAX0 = 2, EW_MIN0 = 3 };
constexpr uint8_t EW_SKIP = 0xFF;   // index belongs to a SOC / PSD cone (handled by the batched kernels)

struct ConeDesc {          // a SOC / rotated SOC / PSD cone in stacked index space [0, n+m)
    int64_t start;         // first stacked index
    int32_t len;           // number of entries
    int32_t k;             // PSD order (0 for SOC)
    int32_t dual_part;     // which part (0: part1, 1: part2) takes the DUAL projection (Moreau); the other takes the primal
    int32_t type;          // FOS_CONE_*
};

// ---------------------------------------------------------------------------------- peer mailboxes (sharded scalar sums)
// All-reduce of <= PEER_MAX_VALS doubles over <= PEER_MAX_RANKS GPUs without a collective library: every rank owns a
// mailbox in UNCACHED device memory that its peers map through HIP IPC.  An exchange (inside reduce_kernel) stores the
// rank's local sums into every peer's mailbox and polls its own until all ranks' words of this exchange have arrived,
// then adds them in rank order -- bitwise the same result on every rank.  Every 64-bit word carries its own validity:
//   word = (sequence number << 32) | 32 payload bits         (two words per double; 8-byte stores are single-copy atomic)
// so no fence or separate flag is needed.  Layout: [parity 2][source rank][value][half]; the parity of the sequence
// number selects the half a given exchange uses (a rank can be at most one exchange ahead of a peer).
constexpr int PEER_MAX_RANKS = 16;
constexpr int PEER_MAX_VALS = 8;
constexpr size_t PEER_BOX_WORDS = (size_t)2 * PEER_MAX_RANKS * PEER_MAX_VALS * 2;     // region 0; region 1 (folded exchanges, four slots) is twice that
constexpr size_t PEER_BOX_TOTAL_WORDS = 3 * PEER_BOX_WORDS;
// A mailbox has TWO such regions: region 0 for the exchanges of the single-workgroup reduce kernel (sequence = a device
// counter of executed exchanges), region 1 for the two exchanges of a CG iteration that are FOLDED into the CG vector kernels
// (every workgroup reduces the local records itself, workgroup 0 also writes them to the peers, every workgroup polls; the
// sequence number is (CG solve number, iteration, phase), known at launch, its parity = the phase).
// Second transport, HOST-PINNED mailboxes (round 5; fos_peer_open_host): the same words in ONE segment of pinned host memory (POSIX shm, registered
// by every rank with hipHostRegister) -- no peer access between the devices and no IPC handle is needed, every box[r] is that one segment
// (`shared`: a rank writes its words once, not once per peer).  A read of host memory is a PCIe round trip, so of the workgroups of a folded
// exchange only workgroup 0 polls the segment; it republishes every word it has seen in `relay`, a mailbox-shaped array in local uncached
// device memory that the other workgroups poll (the words validate themselves there as well).
struct PeerBox {
    unsigned long long* const* box;   // device table [nranks]: box[r] = mailbox of rank r (own entry: the local allocation)
    uint32_t* seq;                    // device counter of EXECUTED exchanges (gated no-op launches do not count)
    int32_t nranks, rank;
    int64_t timeout_ticks;            // wall_clock64() ticks (100 MHz) an exchange may wait for its peers
    unsigned long long* relay;        // host-pinned transport: local republication of the peers' words (nullptr: device mailboxes)
    int32_t shared;                   // 1: all box[r] are one segment
    int32_t loopback;                 // measurement (FOS_PEER_LOOPBACK=1): a rank's OWN words travel through the mailbox too, so that a single
                                      // rank pays the store -> poll latency of the transport in every folded exchange
};

// Row-sharded operators without a collective library: the n-vector exchange of HSDEAffine.jl:51 (A'y = sum over the ranks of
// A_g' y_g, both right-hand sides: 2n doubles) through peer-mapped memory.  Every rank owns a buffer [2 parities][nranks][2n]
// doubles + flags [2][nranks] in UNCACHED device memory that its peers map through HIP IPC.  An exchange = two launches:
//   push   every workgroup copies a chunk of this rank's slots into every peer's buffer (slot [parity][this rank]); the LAST
//          workgroup to finish (a local counter) raises this rank's flag in every peer's flag array with the exchange's number;
//   sum    every workgroup waits for all peers' flags of this exchange, then adds the contributions in rank order (its own from
//          the local slots) into slots_rd -- the same bits on every rank.
// The parity of the exchange number selects the half: a rank can be at most one exchange ahead of a peer (its next push needs
// the peer's flag of the one in between).  Bandwidth: 16 n (g-1) bytes written over xGMI per rank and apply.
struct LaunchCtx;
struct VecBox {
    double* const* buf;        // device table [nranks]: rank r's buffer (own entry: the local allocation)
    uint32_t* const* flags;    // device table [nranks]: rank r's flag array
    uint32_t* counter;         // local: workgroups of the push kernel that have finished
    int32_t nranks, rank;
    int64_t n2;                // doubles per contribution (2 n)
    int64_t timeout_ticks;
};
void launch_vec_exchange(const LaunchCtx& c, const VecBox& vb, uint32_t seq, const double* slots, double* slots_rd);
// row-sharded + dual tiles: the local slot lists of the rows of A' -> one partial sum per row (kernels.hip)
void launch_slots_compact(const LaunchCtx& c, int nrows, const DefRow* rec, const int32_t* idx, int lpr, const double* slots, double* out);

// ---------------------------------------------------------------------------------- kernel launchers (kernels.hip / psd.hip)
struct LaunchCtx {
    hipStream_t stream;
    DevBlkCsr S;
    const double* cb;      // [n+m] = [c; b]
    int64_t n, m, l;       // l = n+m+1
    DevState* st;
    double* partials;      // scratch for per-workgroup partial sums
    double* reduced;       // small buffer of locally reduced sums (all-reduced in place when sharded)
    int32_t vec_blocks;    // grid of the vector kernels
    int32_t cg_blocks;     // grid of the two fused CG kernels (every workgroup re-reduces the partials: fewer, fatter groups)
    const PeerBox* peer;   // non-null: launch_reduce1 also exchanges the sums with the peer ranks (no RCCL call follows)
    const uint32_t* def_mask;   // bit i set: row i of S is finished from partial slots (nullptr: no dual tiles)
    // row-sharded operators (rank g owns rows of A, the column space is replicated): the slots the sweep filled are
    // summed over the ranks between the sweep and whatever adds the slot lists (`between`, called by the launchers); sums over
    // replicated entries (indices < n_repl) are counted by ONE rank only (count_repl).
    int (*between)(void*);
    void* between_arg;
    double* pre = nullptr;           // 3 x 16 doubles + 16 flags: the producer workgroups' sums of the sweep records (cg_update_kernel)
    const int32_t* gate = nullptr;   // non-null: the relaxation / cone kernels run only if *gate != 0 (DevState.done: enqueued behind a CG batch)
    // batched PSD projection: the handle's device and switches (read once at fos_create)
    int32_t cus = 0;                 // compute units of the handle's device
    int32_t psd_wave = -1;           // FOS_PSD_WAVE: -1 decide from the batch size, 0 / 1 force the workgroup / wavefront kernel at order 64
    bool psd_narrow = false, psd_wide = false;
    int32_t psd_wide_threads = 512;
    bool* psd_attr_set = nullptr;    // the handle's "LDS opt-in done" flag
    bool* psd_attr_set_r = nullptr;  // the same for the refinement kernel
    int32_t psd_refine = -1;         // FOS_PSD_REFINE: -1 by batch size, 0 / 1 never / always (order 64, warm)
    bool psd_extrapolate = true;     // FOS_PSD_EXTRAPOLATE=0: the refinement starts from the previous basis only
    double psd_theta = 0.0;          // FOS_PSD_THETA: rotation threshold of the refinement (0: the built-in value)
    int32_t psd_refine_max_mats = 0; // > 0: peer ranks share THIS device (tests): the refinement kernel only for batches this small (psd.hip)
    int32_t count_repl;         // 1: this rank counts the replicated entries in scalar sums (always 1 when not row-sharded)
    int64_t n_repl;             // replicated leading entries of every vector (0 when not row-sharded)
};

// KKT apply, 2 RHS interleaved:  out = [I Q'; Q -I] * w   (rows 0..n+m-1; the tau row is written by kkt_finalize).
// Stand-alone form: sweep (+ deferred-row kernel for operators with dual tiles); partial sums: c.S.npart records at c.S.part_off
void launch_kkt2(const LaunchCtx& c, const double2* w, double2* out, int gate, bool finish_deferred = true);
// finishes the tau rows of out = M w from the sweep's partial sums (CG init / test entry)
void launch_kkt_finalize(const LaunchCtx& c, const double2* w, double2* out, int gate, int from_reduced);

// One CG iteration = TWO dependent launches (conjugategradients.jl:37-51):
//   launch_kkt2_cg   iteration j >= 2 with fuse_p: closes iteration j-1 (r.r, stop test, beta), then sweeps with
//                    p_j = r + beta p_{j-1} formed on the fly and stored through the rows' owners; partial sums of Ap.p and
//                    of the tau rows, including the share of rows spread over dual-tile slots (c.S.nwg records at record 0);
//   launch_cg_update alpha, tau rows, the slot-spread rows of Ap, x += alpha p, r -= alpha Ap, r.r partials.
// Without fuse_p (gather-bound sparse operators: forming p on the fly would double the gathers) the p update stays a third
// launch (launch_cg_pupdate).  p_j lives in buffer j & 1.
struct CgIter {
    int j;                     // iteration number, from 1
    const double2* r;
    const double2* p_prev;     // p_{j-1}
    double2* p_cur;            // p_j
    bool fuse_p;
    int rr_from_reduced;       // r.r of iteration j-1 was all-reduced into c.reduced (sharded, RCCL)
    const PeerBox* fold;       // non-null: the two exchanges of the iteration happen inside the two kernels (peer mailboxes)
    uint32_t seq_base;
    int32_t batch_mark = 0;    // != 0: this is the last iteration the host enqueued; if CG goes on after it, the p update tells the host (HostMark.batch)
    bool start_fused = false;  // the solve was started by launch_cgm_apply / launch_cgm_start: the sweep of iteration 1 adds g_0 = r_0.r_0
};
void launch_kkt2_cg(const LaunchCtx& c, const CgIter& it, double2* Ap);
void launch_cg_stop_check(const LaunchCtx& c, const CgIter& it);      // closes iteration it.j - 1 when no sweep follows in this batch
void launch_cg_update(const LaunchCtx& c, const CgIter& it, double2* x, double2* r, double2* Ap, int kkt_from_reduced);
void launch_cg_pupdate(const LaunchCtx& c, const CgIter& it, double2* x, double2* p_next);   // !fuse_p: x += alpha p_j, closes iteration j, p_{j+1} = r + beta p_j

// Merged-reduction CG (Chronopoulos & Gear 1989; SURVEY 7 "hard parts"): the same Krylov iterates as conjugategradients.jl:31-55
// in exact arithmetic, rearranged so that ONE reduction point per iteration carries both inner products -- one exchange per
// iteration when sharded, and two launches instead of three:
//     w_0 = M r_0 ;  iteration j (i = j-1):   beta_i = g_i / g_{i-1} (0 for i = 0),  alpha_i = g_i / (d_i - beta_i g_i / alpha_{i-1})
//     UPDATE  p = r + beta p ; s = w + beta s (= M p) ; x += alpha p ; r -= alpha s ; partial sums of g_{i+1} = r.r
//     SWEEP   w = M r ; partial sums of d_{i+1} = w.r and of the tau row
// with g = r.r, d = w.r.  The stop test of iteration j (norm(r) <= tol || j >= max_iters, conjugategradients.jl:42) needs g_j:
//   close_in_update = false  the sweep behind the update adds the r.r records in its prologue and returns when CG has stopped
//                            (single GPU: no sweep is wasted; costs every sweep workgroup a short prologue);
//   close_in_update = true   the NEXT update adds them together with the sweep's sums -- one exchange of four doubles per iteration
//                            (sharded handles); the sweep of the last iteration has then run for nothing, and a batch of
//                            enqueued iterations ends with a one-workgroup launch of the update kernel that only closes.
// p lives in PB[0], s in PB[1], w in AP.
struct CgmIter {
    int j;                     // iteration number, from 1
    double2 *x, *r, *p, *s, *w;
    bool close_in_update;
    int from_reduced;          // the four sums were all-reduced into c.reduced (RCCL / host collective)
    const PeerBox* fold;       // non-null: the exchange happens inside the update kernel (peer mailboxes)
    uint32_t seq_base;
    int32_t batch_mark = 0;    // != 0: the launch that ends the host's batch tells the host when CG goes on after it (HostMark.batch)
};
constexpr int CGM_RR_STRIDE = 1024;   // r.r records ping-pong: the update of iteration j writes [j & 1][...]
void launch_cgm_sweep(const LaunchCtx& c, const CgmIter& it, int closes);        // w = M r (gated); closes >= 0: first closes that iteration (0: the start, g_0 only)
void launch_cgm_update(const LaunchCtx& c, const CgmIter& it, bool close_only);
// the start of a solve: the sweep it.w = M v (all rows but the tau row, slot-spread rows unfinished; launch_cgm_apply) and
// r = rhs - M v with the r.r records of "iteration 0"; its first workgroup also opens the solve in DevState (done = 0, tol, ...)
void launch_cgm_apply(const LaunchCtx& c, const CgmIter& it, const double2* v);
void launch_cgm_start(const LaunchCtx& c, const CgmIter& it, const double2* rhs, const double2* v, double tol, int maxit,
                      double2* p_out = nullptr);     // p_out: also p_1 = r_0 (the reference recurrence started this way)

// RESIDENT CG (round 6; resident.hip): the merged-reduction recurrence above as ONE launch per solve, for operators that are
// nothing but dual tiles of a block-separable A (a block-diagonal SDP: C4 and its shards) and small enough per workgroup to be
// HELD IN REGISTERS.  A UNIT = the tiles over one run of <= 64 columns of A (one diagonal block); its tiles are dealt to `wpu`
// consecutive workgroups, one tile per (wavefront, slot): lane = row keeps its row's values, its vector elements x, r, p, s, w
// and [c;b] in registers for the whole solve, the unit's <= 64 column elements are replicated in every workgroup of the unit.
// Per iteration nothing crosses workgroups but (i) the four sums {r.r, w.r, [c;b].r1, [c;b].r2} -- every workgroup publishes its
// record as self-validating words (the mailboxes' trick) and adds all G records in workgroup order: the same bits everywhere, no
// grid barrier, no L2 write-back or invalidate -- and (ii) the unit's column sums between its `wpu` workgroups, the same way.
// Across GPUs the four sums go on through the handle's mailboxes (peer_fold_sum), exactly as in cgm_update_kernel.
struct ResWG { int32_t blk0, nblk, c0, tc, T, wg0, wpu, idx; };     // 32 bytes: tiles blk0 .. blk0 + nblk - 1 of S.blk, the unit's columns [c0, c0 + tc), T steps;
                                                                    // the unit's workgroups wg0 .. wg0 + wpu - 1, this one is number idx
constexpr int RES_GMAX = 512;        // workgroups of a resident solve (records every workgroup adds)
constexpr int RES_WPU_MAX = 16;      // workgroups per unit
constexpr int RES_SLOTS = 21;        // tiles per workgroup
struct ResPlan {                     // host
    int G = 0, nw = 0, ncomm = 0, rpt = 0, tmax = 0, tiles_wg_max = 0, units = 0;      // nw compute wavefronts + ncomm communication wavefronts per workgroup
    int stream = 0, nt = 0;          // stream: the STREAMED form (tiles re-read every iteration, whole units per workgroup); nt: tiles per compute wavefront
    std::vector<ResWG> wg;
    std::string why;                 // why the operator does not qualify (G == 0)
};
// the plan for at most `gmax` workgroups; false (and why) when the operator does not qualify
bool build_resident_plan(const HostBlkCsr& S, int64_t m, int64_t n, int gmax, ResPlan* out);
// host emulation of the resident solve over the same plan (CPU tests of the plan and of the arithmetic; csr_build.cpp)
int host_resident_cg(const HostBlkCsr& S, const ResPlan& P, int64_t m, int64_t n, const double* cb, double2* x, const double2* rhs, const double2* v0,
                     double tol, int maxit, int* iters);
struct ResLaunch {
    const ResWG* wg; int G, nw, ncomm, rpt, tmax, stream, nt, tiles_wg_max;
    unsigned long long* grec;        // [2][G][4][2] words: the workgroups' records of the four sums
    unsigned long long* crec;        // [2][G][tmax][2][2] words: the workgroups' column sums (units of more than one workgroup)
    int64_t timeout_ticks;
};
// x (in: the start iterate, out: the solution), r_0 = rhs - M v; fold != nullptr: the sums cross the ranks through the mailboxes
void launch_cg_resident(const LaunchCtx& c, const ResLaunch& rl, double2* x, const double2* rhs, const double2* v, double tol, int maxit,
                        const PeerBox* fold, uint32_t seq_base);

// single right-hand side Q apply on component `comp` of an interleaved vector
//   Q_PLAIN : out_plain[i] = sign * (Q v)_i            (rows 0..n+m-1; tau row by q1_finalize)
//   Q_RHS   : out[i] = (x[i].x - (Q x.y)_i, 0)         (affinepluslinear.jl:94-95 with beta=1, q=0, b=0)
//   Q_STATUS: residual sums of checkstatus             (HSDEStatus.jl:34-38,59,61)
enum QMode { Q_PLAIN = 0, Q_RHS = 1, Q_STATUS = 2, Q_VFROMU = 3 };
void launch_q1(const LaunchCtx& c, QMode mode, const double2* v, int comp, double sign, void* out);
void launch_q1_finalize(const LaunchCtx& c, QMode mode, const double2* v, int comp, double sign, void* out, int from_reduced);

// CG vector kernels
void launch_cg_init(const LaunchCtx& c, const double2* rhs, const double2* Ap, double2* r, double2* p);
void launch_cg_init_finalize(const LaunchCtx& c, const double2* r, double tol, int maxit, int from_reduced);
void launch_reduce1(const LaunchCtx& c, int count, int nacc, int gate, int off = 0);
void launch_peer_chain(const LaunchCtx& c, int rounds);     // `rounds` mailbox exchanges of four doubles inside one launch (fos_exchange_bench)   // partials[off.. +count][nacc] -> reduced[nacc] (+ peer exchange)

// outer-loop vector kernels (gap.jl:48,58,78; gapa.jl:67,77,96-103; fista.jl:31-46; dykstra.jl)
constexpr int LONG_KMAX_ROWS = 32;   // saved planes of a LongstepWrapper: 2 (nsave + 1) <= 32
// LongstepWrapper state of a handle (wrappers/longstep.jl:12-22 LongstepWrapperData + saveplanes.jl:5-11 SavedPlanes), device resident
struct LongPlanes {
    int64_t interval = 0, nsave = 0;
    int64_t savepos = 0;               // LongstepWrapperData.savepos (0 at the start, -1 after a projection)
    bool now = false;                  // the iteration in flight saves planes
    double2* P = nullptr;              // [2 (nsave + 1)][l] saved rows x - y, in the reference's row order (equality, inequality, equality, ...)
    double* bpart = nullptr;           // [rows][vec_blocks] partial sums of the offsets (x - y).y
    double* dots = nullptr;            // [vec_blocks][33][hi, lo] scratch of the Gram products (double-double)
    double* nu = nullptr;              // [rows][hi, lo] multipliers, device copy
    double log[8] = {0};               // last projection: iteration, active inequalities, KKT violation, |x_new - x|, rows, supports tried, [6] given up (0 / 1),
                                       // [7] projections given up since fos_set_longstep
    int64_t max_supports = 4096;       // candidate supports a projection may try (FOS_LONG_MAX_SUPPORTS, read at fos_set_longstep)
};
struct LaunchCtx;
void long_save_plane(const LaunchCtx& c, LongPlanes& lp, int which, const double2* y, const double2* x);      // addprojeq (0) / addprojineq (1) at lp.savepos
int long_project_planes(const LaunchCtx& c, LongPlanes& lp, double2* X, int64_t i);                          // projectonnormals! + x .= tmp   (solver.cpp)
void launch_long_plane(const LaunchCtx& c, double2* row, const double2* x, const double2* y, double* bpart);          // row = x - y; bpart[blocks]: partial sums of (x - y).y
void launch_long_dots(const LaunchCtx& c, const double2* P, int K, int a, const double2* x, double* out);              // out[blocks][33][hi, lo]: row_a . row_{a+k}, k < 32; [32]: row_a . x (double-double)
void launch_long_apply(const LaunchCtx& c, double2* x, const double2* P, int K, const double* nu);                     // x += sum_k nu[k] row_k
void launch_normdiff(const LaunchCtx& c, const double2* x, const double2* y);   // c.partials[0 .. vec_blocks) = partial sums of |x - y|^2
void launch_shift_part2(const LaunchCtx& c, double2* out, const double2* y, const double2* x);   // out = (y.x, y.y - x.y)
void launch_axpby(const LaunchCtx& c, double2* out, double a, const double2* x, double b, const double2* y);          // out = a x + b y
void launch_relax_a12(const LaunchCtx& c, double2* out, const double2* y, const double2* x);                          // out = a12 y + (1-a12) x, a12 from state
// shift_out != nullptr: also shift_out = sol - [0; x2_new] (launch_shift_part2 for the next affine projection, saved a launch)
void launch_gap_final(const LaunchCtx& c, double2* x, const double2* t2, const double2* t1, double alpha, double alpha2,
                      double2* shift_out = nullptr, const double2* sol = nullptr);   // x = alpha(alpha2 t2+(1-alpha2)t1) + (1-alpha) x
void launch_gapa_final(const LaunchCtx& c, double2* x, const double2* t2, const double2* t1, double alpha,
                       double2* shift_out = nullptr, const double2* sol = nullptr);  // + normedScalar partials
void launch_gapa_finalize(const LaunchCtx& c, double beta, int from_reduced);
void launch_fista_extrap(const LaunchCtx& c, double2* y, const double2* x, const double2* xold, double coef);         // y = x + coef (x - xold)
void launch_add(const LaunchCtx& c, double2* out, const double2* a, const double2* b);                                // out = a + b
void launch_copy(const LaunchCtx& c, double2* out, const double2* in);                                                 // out = in (gateable)
void launch_dykstra_corr(const LaunchCtx& c, double2* p, const double2* x, const double2* y);                         // p = x + p - y

// direct = true (HSDE.jl:12-15): dense set-up helpers and the per-projection kernels
void launch_dense_q_fill(const LaunchCtx& c, const int64_t* colptr, const int64_t* rowval, const double* nzval, double* Q, int64_t ld);
void launch_dense_gemm(const LaunchCtx& c, int L, double alpha, const double* A, const double* B, double gamma, const double* D, double* Cm);
void launch_dense_resid(const LaunchCtx& c, int64_t L, const double* Y, double* partials, int nblocks);
void launch_dense_scale_identity(const LaunchCtx& c, int64_t L, double* X, double s);
void launch_dense_symv(const LaunchCtx& c, int64_t ld, const double* G, const double* t, double* w);
void launch_direct_rhs(const LaunchCtx& c, const double2* W, const double2* x, double* t);
void launch_direct_finish(const LaunchCtx& c, const double2* x, const double2* W, double2* out);

// direct = true on a block-separable operator (I + A'A block diagonal with blocks of order <= BLKDIR_MAX): vecops.hip, solver.cpp prox_affine_direct_block
constexpr int BLKDIR_MAX = 64;
void launch_blkdir_prep(const LaunchCtx& c, const double2* T, const double2* phg, double2* W2, double2* W3, double* partials);
void launch_blkdir_solve(const LaunchCtx& c, int nblk, const int64_t* goff, const int32_t* ioff, const int32_t* idx, const double* Ginv, const double2* R,
                         const double2* T, double2* W3, double* ctx_rec);
void launch_blkdir_combine(const LaunchCtx& c, const double2* T, const double2* W3, const double2* V, const double2* phg, const double2* qphg,
                           double* prm, int zero_kappa, double2* out, const double* prep_partials, double* partials, int from_reduced);
void launch_blkdir_tausum(const LaunchCtx& c, const double* partials, const double* ctx_rec, int nblk, double* out1);
void launch_blkdir_tau(const LaunchCtx& c, const double2* T, const double2* qphg, const double* prm, int zero_kappa, double2* out, const double* prep_partials,
                       const double* partials, const double* ctx_rec, int nblk, int from_reduced);

// layout conversion at the ABI boundary
void launch_interleave(const LaunchCtx& c, double2* out, const double* plain);    // plain [part1(l); part2(l)] -> interleaved
void launch_deinterleave(const LaunchCtx& c, double* plain, const double2* in);
void launch_set_comp(const LaunchCtx& c, double2* out, const double* plain_l, int comp);   // out[i].comp = plain[i], other comp = 0
void launch_get_plain(const LaunchCtx& c, double* plain_l, const double2* in, int comp);

// cones (cones.jl:122-142)
void launch_cones_elementwise(const LaunchCtx& c, double2* out, const double2* in, const uint8_t* ew_op);
// t1 = a sol + (1 - a) x (a, or alpha12 of the device state) and t2 = P(t1) on the elementwise cones' indices, one pass (vecops.hip)
void launch_relax_ew(const LaunchCtx& c, double2* t1, double2* t2, const double2* sol, const double2* x, double a, bool use_a12, const uint8_t* ew_op);
void launch_cones_soc(const LaunchCtx& c, double2* out, const double2* in, const ConeDesc* cones, int ncones);
void launch_cones_exp(const LaunchCtx& c, double2* out, const double2* in, const ConeDesc* cones, int ncones);
// Fusion of the GAP / DR step around the batched PSD(64) projection (psd64_refine_kernel): the kernel's input is not read from a vector t1
// but FORMED, t1 = a1 sol + (1 - a1) x (the relaxation behind S1, gap.jl:48), and its output is not written to t2 but taken on to the step's
// last pass, x = alpha (alpha2 t2 + (1 - alpha2) t1) + (1 - alpha) x (gap.jl:58,78), with the vector the next CG start applies M to
// (sol - [0; x2]) -- for the entries of the PSD cones; the other entries (elementwise cones) are done by extra workgroups of the same launch.
struct PsdFuse {
    int on;
    const double2* sol; double2* xv; double2* shift;     // shift: nullptr when the next projection is not a CG solve inside fos_step
    double a1, alpha, alpha2;
    const uint8_t* ew_op; int64_t l; int nmat;           // elementwise entries: workgroups nmat.. of the grid
};
bool psd_fuse_possible(const LaunchCtx& c, int ncones, int kmin, int kmax, const double* vin, const double* vout, int have_prev, const int32_t* redo, int phase_limit);
int  launch_cones_psd(const LaunchCtx& c, double2* out, const double2* in, const ConeDesc* cones, int ncones,
                      int kmin, int kmax, double* gscratch, const double* vin, double* vout, int have_prev, int* stats, int phase_limit,
                      int32_t* redo = nullptr, const PsdFuse* fuse = nullptr);
// PSD cones of order > 64: the projection by matrix products only, P = (M + M sign(M)) / 2 with sign(M) from an inverse-free polynomial iteration on the fp64
// matrix cores (psd_sign.hip).  psd_sign_setup takes those cones OUT of `psd` (what stays goes to the kernels of psd.hip) and allocates their workspace.
struct PsdSign;
int psd_sign_setup(std::vector<ConeDesc>& psd, PsdSign** out);
void psd_sign_destroy(PsdSign* p);
int psd_sign_count(const PsdSign* p);
int launch_cones_psd_sign(const LaunchCtx& c, PsdSign* p, double2* out, const double2* in);
// IndAffine(A, b) with a sparse A (affine_sparse.hip): exact projection by warm-started CG on the row-scaled normal equations, the true residual re-checked
struct SparseAffine;
int sparse_affine_setup(int64_t m, int64_t n, const int64_t* colptr, const int64_t* rowval, const double* nzval, const double* b, int cus, SparseAffine** out);
int sparse_affine_project(SparseAffine* a, hipStream_t stream, double* y, const double* x);      // y, x: device vectors of length n, y must not alias x
void sparse_affine_reset(SparseAffine* a, hipStream_t stream);                                   // a new solve: lambda = 0
void sparse_affine_stats(const SparseAffine* a, double* out8);
void sparse_affine_destroy(SparseAffine* a);
size_t psd_scratch_bytes(int kmax, int ncones);
size_t psd_basis_doubles(int kmax, int ncones);

void launch_status_finalize(const LaunchCtx& c, const double2* z, int from_reduced);

}  // namespace fos
