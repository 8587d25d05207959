// mnmf engine: context, resident matrix, the three MU solver loops and the C-ABI (include/mnmf.h).
// gfx950 only.  Host orchestration is deliberately thin: every numeric expression of the reference
// loops runs in a HIP kernel from spmm.hpp / dense.hpp / gram_mfma.hpp.
#include <dlfcn.h>
#include <chrono>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <string>
#include <unordered_map>
#include <vector>

#include <rocprim/rocprim.hpp>

#include "../../include/mnmf.h"
#include "common.hpp"
#include "dense.hpp"
#include "eig.hpp"
#include "gram_mfma.hpp"
#include "histogram.hpp"
#include "rk_mfma_f64.hpp"
#include "gram_mfma_f64.hpp"
#include "sparse_prep.hpp"
#include "spmm.hpp"
#include "topk.hpp"

using namespace mnmf;

#include "impl_base.hpp"
#include "ctx.hpp"

namespace {

struct DevBuf {
  void* p = nullptr;
  size_t cap = 0;
  ~DevBuf() { release(); }
  void release() {
    if (p) (void)hipFree(p);
    p = nullptr;
    cap = 0;
  }
  hipError_t ensure(size_t bytes) {
    if (bytes <= cap) return hipSuccess;
    release();
    if (bytes == 0) return hipSuccess;
    hipError_t e = mnmf_counted_malloc(&p, bytes);
    if (e == hipSuccess) cap = bytes; else p = nullptr;
    return e;
  }
  template <typename U>
  U* as() const { return reinterpret_cast<U*>(p); }
};

// Tunables (environment overrides are for experiments; defaults are what the benchmarks use).
// sets Engine::data_cols for the duration of a solver call
struct ColsGuard {
  int& ref;
  bool* line = nullptr;
  ColsGuard(int& r, int k) : ref(r) { ref = k; }
  ColsGuard(int& r, int k, bool& line_layout, bool on) : ref(r), line(&line_layout) { ref = k; line_layout = on; }
  ~ColsGuard() {
    ref = 0;
    if (line) *line = false;
  }
  ColsGuard(const ColsGuard&) = delete;
  ColsGuard& operator=(const ColsGuard&) = delete;
};

struct Tune {
  int64_t x_q = 512;                  // target nnz per segment, X stream with the LDS-hot kernel (256: configs[1] X.H^T +4 %)
  int64_t x_q_plain = 256;            // ... with the plain kernel (512: X.H^T +7..10 % at 1e7-2.5e7 nnz, equal elsewhere)
  int64_t xt_q = 0;                   // target nnz per segment, X^T blocked stream; 0 = 64 below 7e7 nnz, 128 above (X^T.W -2..7 % / +3 % across that line)
  int64_t xt_block_bytes = 0;         // bytes of W rows per document block of X^T (upper limit; see spmm()); 0 = 4 MiB, 8 MiB at 32 lanes per row
  int xcd_affinity = 1;               // order X^T segments so that one XCD works on one document block
  int ld_align = 16;                  // leading-dimension granule (elements) for k > 32
  int permute_vocab = 1;              // renumber vocabulary rows by descending document frequency
  int64_t hot_lds_bytes = 152 << 10;  // LDS given to the most frequent rows of H^T in the X.H^T SpMM (0: off; 96 / 128 / 152 KiB: X.H^T 1.175 / 1.157 / 1.145 ms at configs[1])
  int xt_hot_words = -1;              // X^T: this many of the most frequent words go through LDS tiles of W (0: none; -1: 512; 256 when a tile holds fewer than 256 documents, 1024 from 1024)
  int64_t tile_lds_bytes = 156 << 10; // LDS budget of the W tiles of one CU (divided by tile_wgs_per_cu)
  int tile_min_rows = 128;            // tile stream only if a tile holds at least this many documents (174-document tiles: X^T.W -10 % at k = 100 fp64; 93: +6 %)
  int tile_min_tiles_pct = 60;        // ... and there are at least this many tiles, in percent of the CU count (64 %: X^T.W -15..-20 %; 38 %: +14 %)
  int ld_align_small = 0;             // padding of factor rows with k <= 32 (elements); 0 = whole 128-byte lines (32 floats /
                                      // 16 doubles): C4, k = 20: 80-byte rows -> 128-byte rows, +10 %; C1, k = 10: +17 %
  int pack_exchange = 1;              // sharded runs: all-reduce only the k live columns of ld-padded rows
  int wide_gram = 1;                  // symmetric Grams with 128 < ld <= 256: tiles split over the waves of a workgroup
  int xt_min_piece = 4;               // X^T blocked stream: block a word only if df >= this many nnz per document block
  int kxk_chain = 1;                  // NMTF: the k x k products of an iteration in two launches instead of nine (ld <= 32)
  int seg_target = 4096;              // shorten the segments of a stream until there are this many per lane group of a wave
  int side_prio = -1;                 // priority of the side stream: -1 low, 0 as the caller's stream, 1 high
  int xt_overlap = 1;                 // X^T.W: tile stream and blocked stream on two HIP streams at once
  int tile_threads = 0;               // threads of the tile kernel's workgroup (one per CU); 0: 512 beside the blocked
                                      // stream (half of a CU's registers stay free for it), 1024 when it runs alone
  int tile_wgs_per_cu = 1;            // workgroups per CU in the tile kernel (1, 2 or 4)
  int64_t xth_q = 96;                 // X^T tile stream: > 0 target nnz per segment; <= 0: equal split, -xth_q segments per lane group
  int mfma_gram = 1;                  // fp32 Gram products on the matrix cores
  int fuse_grams = 1;                 // Grams of the new factor / of |new - old| inside the update kernel (k <= 64)
  int mfma_rk = 1;                    // fp32 rows x (k x k) products + update on the matrix cores
  int64_t hot_min_nnz = 1 << 20;      // X^T.W tile stream (and, when set through the environment, X.H^T too): smallest matrix
  int64_t hot_x_min_nnz = -1;         // X.H^T: smallest matrix that takes the LDS-hot kernel; -1 = the measured rule (hot_x_threshold)
  int64_t xt_long_floor = 128;        // X^T blocked stream: words with at most this many nnz are never cut at document blocks
  int lds_pad = -1;                   // padding of a factor row in the LDS images (elements); < 0: lds_pad_for()
  int sweep = 1;                      // X^T.W: rare words through the synchronised sweep (0 off; 1 when the shape suits it; 2 always: tests)
  int64_t sweep_lds_bytes = 150 << 10; // LDS for the accumulator rows of one partition
  int64_t sweep_win_bytes = 8 << 20;  // bytes of W rows per document window (configs[1], sweep kernel alone for df <= 244: 1 / 2 / 4 / 8 / 16 / 64 MiB 544 / 333 / 304 / 246 / 260 / 305 us: fewer barriers against reuse in L2)
  int sweep_min_touch_pct = 100;      // use it only if a document is gathered at least this often (percent) per XCD: below, no reuse
  int64_t sweep_max_df = 0;           // words with up to this many nnz (at least the uncut ones) go to the sweep; 0 = 32 per window
  int64_t sweep_min_table_bytes = 64 << 20;  // ... and only if the gathered table is at least this large
  int sweep_order = 1;                // 0: sweep first, beside the blocked stream; 1: tile || blocked, then the sweep alone; 2: blocked || tile after, sweep last
  int rk_one_pass = 0;                // (measured 4 % slower at the configs[4] shard: the kernel is matrix-core bound) fp32 update with 128 < k <= 256: all column tiles of a row tile in one wave
  int mfma_f64 = 1;                   // fp64 update + fused Grams on the matrix cores (k <= 64)
  int rs_ag = 1;                      // sharded NMF / NMTF: reduce-scatter -> slice update -> all-gather (needs mnmf_set_comm)
  int force_wide = 0;                 // tests: take the 64-bit-address SpMM variant whatever the table size
  int exact_seg = -1;                 // segments of exactly q nnz (pieces cut at the boundary): -1 by size (ensure_segments), 0 never, 1 X^T blocked stream, 2 every stream
  int queue = 0;                      // X^T blocked stream as a persistent launch with per-lane-group segment queues (2: the plain X stream too)
  int queue_wgs = 0;                  // ... workgroups per CU; 0 = what the occupancy calculator says
  int early_exchange = 0;             // sharded NMF: exchange the rows of the less frequent words beside the tile stream of X^T.W.
                                      // Off: it gives up the tile || blocked overlap, which costs what it can hide (one rank's compute at the
                                      // configs[1] shard sizes of 1 / 4 / 8 ranks: 363 -> 324, 1101 -> 996, 1765 -> 1608 it/s)
  int early_check = 0;                // tests: fail instead of falling back when the early exchange cannot be taken
  int async_gram = 1;                 // sharded NMF: the k x k all-reduce of an iteration's end on a stream of its own, beside the next X^T.W
  int hot_wgs_per_cu = 1;             // X.H^T LDS-hot kernel: workgroups per CU (each with its share of hot_lds_bytes) ...
  int hot_threads = 1024;             // ... of this many threads
  int fix_merge = 1;                  // small streams: one fix-up launch for long rows, short rows and rows without nnz
  int64_t fix_merge_bytes = 16 << 20; // ... up to this many bytes of carry slots (at 128 bytes a slot)
  int lds_compact = 1;                // LDS images of factor rows hold only the columns with data (spmm())
  int two_vec = 1;                    // rows of 17..32 vectors (fp64 k = 33..64, fp32 k = 65..128): 16 lanes x 2 vectors per row instead of 32 x 1
  Tune() {
    auto geti = [](const char* n, int64_t d) { const char* v = std::getenv(n); return v ? (int64_t)std::atoll(v) : d; };
    x_q = geti("MNMF_X_Q", x_q);
    x_q_plain = geti("MNMF_X_Q_PLAIN", std::getenv("MNMF_X_Q") ? x_q : x_q_plain);
    xt_q = geti("MNMF_XT_Q", xt_q);
    xt_block_bytes = geti("MNMF_XT_BLOCK_BYTES", xt_block_bytes);
    xcd_affinity = (int)geti("MNMF_XCD_AFFINITY", xcd_affinity);
    ld_align = (int)geti("MNMF_LD_ALIGN", ld_align);
    permute_vocab = (int)geti("MNMF_PERMUTE_VOCAB", permute_vocab);
    hot_lds_bytes = geti("MNMF_HOT_LDS_BYTES", hot_lds_bytes);
    hot_x_min_nnz = geti("MNMF_HOT_X_MIN_NNZ", std::getenv("MNMF_HOT_MIN_NNZ") ? geti("MNMF_HOT_MIN_NNZ", hot_min_nnz) : hot_x_min_nnz);
    hot_min_nnz = geti("MNMF_HOT_MIN_NNZ", hot_min_nnz);
    xt_hot_words = (int)geti("MNMF_XT_HOT_WORDS", xt_hot_words);
    tile_lds_bytes = geti("MNMF_TILE_LDS_BYTES", tile_lds_bytes);
    xth_q = geti("MNMF_XTH_Q", xth_q);
    tile_min_rows = (int)geti("MNMF_TILE_MIN_ROWS", tile_min_rows);
    tile_min_tiles_pct = (int)geti("MNMF_TILE_MIN_TILES_PCT", tile_min_tiles_pct);
    ld_align_small = (int)geti("MNMF_LD_ALIGN_SMALL", ld_align_small);
    xt_min_piece = (int)geti("MNMF_XT_MIN_PIECE", xt_min_piece);
    wide_gram = (int)geti("MNMF_WIDE_GRAM", wide_gram);
    pack_exchange = (int)geti("MNMF_PACK_EXCHANGE", pack_exchange);
    xt_overlap = (int)geti("MNMF_XT_OVERLAP", xt_overlap);
    side_prio = (int)geti("MNMF_SIDE_PRIO", side_prio);
    seg_target = (int)geti("MNMF_SEG_TARGET", seg_target);
    kxk_chain = (int)geti("MNMF_KXK_CHAIN", kxk_chain);
    tile_threads = (int)geti("MNMF_TILE_THREADS", tile_threads);
    if (tile_threads <= 0) tile_threads = xt_overlap ? 512 : 1024;
    tile_threads = std::max(256, std::min(1024, tile_threads / 64 * 64));
    tile_wgs_per_cu = (int)geti("MNMF_TILE_WGS", tile_wgs_per_cu);
    if (tile_wgs_per_cu != 2 && tile_wgs_per_cu != 4) tile_wgs_per_cu = 1;
    mfma_gram = (int)geti("MNMF_MFMA_GRAM", mfma_gram);
    mfma_rk = (int)geti("MNMF_MFMA_RK", mfma_rk);
    fuse_grams = (int)geti("MNMF_FUSE_GRAMS", fuse_grams);
    lds_pad = (int)geti("MNMF_LDS_PAD", lds_pad);
    xt_long_floor = geti("MNMF_XT_LONG_FLOOR", xt_long_floor);
    force_wide = (int)geti("MNMF_FORCE_WIDE", force_wide);
    two_vec = (int)geti("MNMF_TWO_VEC", two_vec);
    lds_compact = (int)geti("MNMF_LDS_COMPACT", lds_compact);
    fix_merge = (int)geti("MNMF_FIX_MERGE", fix_merge);
    fix_merge_bytes = geti("MNMF_FIX_MERGE_BYTES", fix_merge_bytes);
    hot_wgs_per_cu = (int)geti("MNMF_HOT_WGS", hot_wgs_per_cu);
    hot_threads = (int)geti("MNMF_HOT_THREADS", hot_threads);
    async_gram = (int)geti("MNMF_ASYNC_GRAM", async_gram);
    early_exchange = (int)geti("MNMF_EARLY_EXCHANGE", early_exchange);
    early_check = (int)geti("MNMF_EARLY_CHECK", early_check);
    queue = (int)geti("MNMF_QUEUE", queue);
    exact_seg = (int)geti("MNMF_EXACT_SEG", exact_seg);
    queue_wgs = (int)geti("MNMF_QUEUE_WGS", queue_wgs);
    rs_ag = (int)geti("MNMF_RS_AG", rs_ag);
    mfma_f64 = (int)geti("MNMF_MFMA_F64", mfma_f64);
    rk_one_pass = (int)geti("MNMF_RK_ONE_PASS", rk_one_pass);
    sweep = (int)geti("MNMF_SWEEP", sweep);
    sweep_order = (int)geti("MNMF_SWEEP_ORDER", sweep_order);
    sweep_max_df = geti("MNMF_SWEEP_MAX_DF", sweep_max_df);
    sweep_lds_bytes = geti("MNMF_SWEEP_LDS_BYTES", sweep_lds_bytes);
    sweep_win_bytes = geti("MNMF_SWEEP_WIN_BYTES", sweep_win_bytes);
    sweep_min_table_bytes = geti("MNMF_SWEEP_MIN_TABLE_BYTES", sweep_min_table_bytes);
    sweep_min_touch_pct = (int)geti("MNMF_SWEEP_MIN_TOUCH_PCT", sweep_min_touch_pct);
  }
};

constexpr int kLongFix = 32;  // carry lists longer than this are summed by a whole block

// roctx ranges (tracing) without a link-time dependency: MNMF_ROCTX=1 loads the marker library of
// rocprofiler-sdk (what `rocprofv3 --marker-trace` listens to; libroctx64 as fallback) on first use and brackets every
// phase of the solver loops -- the host-side window in which the phase's kernels are enqueued -- with a named range.
struct Roctx {
  using push_fn = int (*)(const char*);
  using pop_fn = int (*)();
  push_fn push = nullptr;
  pop_fn pop = nullptr;
  static Roctx& get() {
    static Roctx r = [] {
      Roctx x;
      const char* e = std::getenv("MNMF_ROCTX");
      if (e && *e && *e != '0') {
        for (const char* lib : {"librocprofiler-sdk-roctx.so", "librocprofiler-sdk-roctx.so.1", "libroctx64.so", "libroctx64.so.4"}) {
          void* h = dlopen(lib, RTLD_NOW | RTLD_GLOBAL);
          if (!h) continue;
          x.push = reinterpret_cast<push_fn>(dlsym(h, "roctxRangePushA"));
          x.pop = reinterpret_cast<pop_fn>(dlsym(h, "roctxRangePop"));
          if (x.push && x.pop) break;
          x.push = nullptr, x.pop = nullptr;
        }
      }
      return x;
    }();
    return r;
  }
  bool on() const { return push != nullptr; }
};

struct PhaseTimer {
  // HIP events on the solver's stream at phase boundaries; summed after the loop has drained.
  static constexpr int kMaxPhases = 8;
  static constexpr const char* kNames[kMaxPhases] = {"mnmf:gram_w", "mnmf:spmm_xt", "mnmf:h_side", "mnmf:gram_h_kxk",
                                                     "mnmf:spmm_x", "mnmf:w_update", "mnmf:check", "mnmf:other"};
  hipStream_t s;
  bool on;
  bool range_open = false;
  std::vector<hipEvent_t> ev;
  std::vector<int> tag;  // phase that ENDS at this event (-1 = start marker)
  PhaseTimer(hipStream_t s_, bool on_) : s(s_), on(on_) {}
  ~PhaseTimer() {
    close_range();
    for (auto e : ev) (void)hipEventDestroy(e);
  }
  void close_range() {
    if (range_open) (void)Roctx::get().pop();
    range_open = false;
  }
  // `phase` ends here; `next` (if >= 0) is the phase whose kernels are enqueued from here on
  void mark(int phase, int next = -1) {
    if (Roctx::get().on()) {
      close_range();
      if (next >= 0 && next < kMaxPhases) range_open = Roctx::get().push(kNames[next]) >= 0;
    }
    if (!on) return;
    hipEvent_t e;
    if (hipEventCreate(&e) != hipSuccess) return;
    (void)hipEventRecord(e, s);
    ev.push_back(e);
    tag.push_back(phase);
  }
  void collect(double* out /*kMaxPhases*/) {
    close_range();
    for (int i = 0; i < kMaxPhases; ++i) out[i] = 0.0;
    for (size_t i = 1; i < ev.size(); ++i) {
      if (tag[i] < 0 || tag[i] >= kMaxPhases) continue;
      float ms = 0.f;
      if (hipEventElapsedTime(&ms, ev[i - 1], ev[i]) == hipSuccess) out[tag[i]] += ms;
    }
  }
};

template <typename T>
struct StreamMat {
  int64_t rows = 0, tab_rows = 0, nnz = 0;  // output rows, gather-table rows
  DevBuf cols, vals;                         // uint32 gather index (end-flagged), T
  HostPieces pieces;                         // processing-order pieces (host)
  bool built = false, flags_set = false;
  int block_shift = -1;                      // X^T only: log2(documents per block)
  int tile_rows = 0;                         // X^T tile stream only: documents per LDS tile
  int64_t n_tiles = 0;
  int n_hot = 0;                             // words in the tile stream (renumbered ids < n_hot)
  DevBuf tile_first;                         // [n_tiles + 1] first segment of each tile
  // segment description for the current seg_q
  int64_t seg_q = -1, n_seg = 0, n_dest = 0, n_empty = 0, n_fix = 0, n_fix_long = 0, n_carry = 0;
  bool fix_merged = false;   // one fix-up launch for everything, rows without nnz included (ensure_segments)
  int seg_group = -1;
  DevBuf seg_desc, dest, empty_rows, fix_row, fix_first, fix_count, carry;
};

// the sweep stream of X^T (rare words; spmm.hpp: spmm_sweep_kernel)
struct SweepMat {
  int64_t nnz = 0;
  int n_windows = 0, rpp = 0, P = 0, win_shift = 0;
  int ngw = 0;                  // segments per cell the segment table was built for
  size_t req_row_bytes = 0;     // what ensure_xt was asked for (0: no sweep wanted)
  uint32_t base = 0;            // first sweep key
  DevBuf docs, lrows, vals, words, keys, cell_start, seg;
};

inline int pow2_ceil(int x) {
  int p = 1;
  while (p < x) p <<= 1;
  return p;
}

template <typename T>
class Engine : public ImplBase {
 public:
  explicit Engine(mnmf_ctx* c) : ctx(c) {
    int v = 0;
    if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, c->device) == hipSuccess && v > 0) n_cus = v;
  }
  ~Engine() override {
    if (h_pinned) (void)hipHostFree(h_pinned);
    if (ev_fetch) (void)hipEventDestroy(ev_fetch);
    if (ev_fork) (void)hipEventDestroy(ev_fork);
    if (ev_join) (void)hipEventDestroy(ev_join);
    if (side_stream) (void)hipStreamDestroy(side_stream);
    if (comm_stream) (void)hipStreamDestroy(comm_stream);
    if (ev_comm_in) (void)hipEventDestroy(ev_comm_in);
    if (ev_comm_out) (void)hipEventDestroy(ev_comm_out);
    if (ev_cold) (void)hipEventDestroy(ev_cold);
    if (ev_hot) (void)hipEventDestroy(ev_hot);
    if (ev_h_done) (void)hipEventDestroy(ev_h_done);
  }

  mnmf_ctx* ctx;
  Tune tune;
  int64_t m = 0, V = 0, nnz = 0;
  bool have_matrix = false;
  StreamMat<T> X, Xt, Xth;   // X; X^T blocked stream; X^T tile stream of the most frequent words
  SweepMat Xs;               // X^T sweep stream of the rare words
  std::vector<int64_t> h_rowptr;
  DevBuf d_rowptr;
  // vocabulary renumbering by descending document frequency (identical on every rank)
  bool prepared = false;     // ensure_prepared() has run to its end for the current matrix and hook
  bool renumbered = false;   // ... and the vocabulary is renumbered (a decision every rank takes alike)
  DevBuf d_newid, d_order;   // int32 newid[old], uint32 order[new]
  const int32_t* vmap() const { return (prepared && renumbered) ? d_newid.template as<int32_t>() : nullptr; }
  int n_cus = 256;
  std::unordered_map<const void*, int> occ_cache;   // resident 256-thread workgroups per CU of the persistent SpMM kernels
  // solver workspaces
  DevBuf Wb[2], Hb[2], Nw, Ex, Aux1, Aux2, slab, small, tmpA, tmpB, tmpC, tmpD, ar_pack;
  double* h_pinned = nullptr;
  size_t h_pinned_cap = 0;
  double phase_ms[PhaseTimer::kMaxPhases] = {0};

  // the caller's stream, or the engine's side stream while the X^T tile stream is being enqueued (see spmm)
  hipStream_t side_stream = nullptr, st_override = nullptr;
  hipStream_t comm_stream = nullptr;                 // small collectives that the next SpMM does not wait for (ensure_comm_stream)
  hipEvent_t ev_comm_in = nullptr, ev_comm_out = nullptr;
  hipEvent_t ev_cold = nullptr, ev_hot = nullptr, ev_h_done = nullptr;   // early exchange of a sharded NMF iteration
  int data_cols = 0;                    // columns of the factors that hold data in the running solver call (0: all ld), ColsGuard
  bool line_layout = false;             // run_nmf / run_nmf_sweep with one 128-byte line per factor row: the layout of X^T must
                                        // not depend on k (spmm())
  bool xt_layout_only = false;          // spmm(Xt, ...) stops after the streams are laid out
  hipEvent_t xt_cold_event = nullptr;   // set around spmm(Xt, ...): cold streams first, this event, then the tile stream
  hipEvent_t ev_fork = nullptr, ev_join = nullptr;
  hipStream_t st() const { return st_override ? st_override : ctx->stream; }
  // leading dimension of every factor matrix: rows of up to 32 elements fill whole 128-byte lines
  // (tune.ld_align_small), longer rows are padded to whole 64-byte multiples (tune.ld_align elements),
  // so a gathered row starts line-aligned and never straddles one line more than it has to.
  int ld_of(int k) const {
    const int small = tune.ld_align_small > 0 ? std::max(4, tune.ld_align_small) : (int)(128 / sizeof(T));
    const int a = (k > 32 && tune.ld_align > 4) ? tune.ld_align : small;
    return (int)round_up(k, a);
  }

  int row_stride(int k) const override { return k > 0 ? ld_of(k) : 0; }

  int fail(int code, const std::string& msg) {
    ctx->fail(code, msg);
    return code;
  }
#define E_HIP(expr)                                                                                   \
  do {                                                                                                \
    hipError_t e__ = (expr);                                                                          \
    if (e__ != hipSuccess)                                                                            \
      return fail(MNMF_EHIP, std::string(#expr) + ": " + hipGetErrorString(e__) + " (engine.hip:" +   \
                                 std::to_string(__LINE__) + ")");                                     \
  } while (0)
#define E_LAUNCH() E_HIP(hipGetLastError())

  int pinned(size_t doubles) {
    if (doubles <= h_pinned_cap) return 0;
    if (h_pinned) (void)hipHostFree(h_pinned);
    h_pinned = nullptr;
    h_pinned_cap = 0;
    g_mnmf_allocs.fetch_add(1, std::memory_order_relaxed);
    E_HIP(hipHostMalloc((void**)&h_pinned, doubles * sizeof(double), hipHostMallocDefault));
    h_pinned_cap = doubles;
    return 0;
  }

  // ------------------------------------------------------------------ matrix
  int set_matrix(int64_t m_, int64_t V_, int64_t nnz_, const int64_t* rowptr, const int32_t* colidx,
                 const void* vals, int on_device) override {
    have_matrix = false;
    if (m_ < 0 || V_ <= 0 || nnz_ < 0 || !rowptr) return fail(MNMF_EINVAL, "set_matrix: bad shape");
    if (nnz_ > 0 && (!colidx || !vals)) return fail(MNMF_EINVAL, "set_matrix: null arrays");
    if (nnz_ >= (int64_t)1 << 31 || m_ >= (int64_t)1 << 31 || V_ >= (int64_t)1 << 31)
      return fail(MNMF_EINVAL, "set_matrix: per-rank m, V and nnz must be below 2^31 (shard the rows)");
    E_HIP(hipSetDevice(ctx->device));
    m = m_;
    V = V_;
    nnz = nnz_;
    const hipMemcpyKind kind = on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice;
    h_rowptr.assign(m + 1, 0);
    if (on_device) {
      E_HIP(hipMemcpyAsync(h_rowptr.data(), rowptr, (m + 1) * sizeof(int64_t), hipMemcpyDeviceToHost, st()));
      E_HIP(hipStreamSynchronize(st()));
    } else {
      std::memcpy(h_rowptr.data(), rowptr, (m + 1) * sizeof(int64_t));
    }
    if (h_rowptr[0] != 0 || h_rowptr[m] != nnz) return fail(MNMF_EINVAL, "set_matrix: rowptr[0] must be 0 and rowptr[m] == nnz");
    for (int64_t r = 0; r < m; ++r)
      if (h_rowptr[r + 1] < h_rowptr[r]) return fail(MNMF_EINVAL, "set_matrix: rowptr not monotone");
    X = StreamMat<T>();
    Xt = StreamMat<T>();
    Xth = StreamMat<T>();
    Xs = SweepMat();
    prepared = false;
    renumbered = false;
    X.rows = m;
    X.tab_rows = V;
    X.nnz = nnz;
    Xt.rows = V;
    Xt.tab_rows = m;
    Xt.nnz = nnz;
    Xth.rows = V;
    Xth.tab_rows = m;
    Xth.nnz = 0;
    const size_t n = (size_t)std::max<int64_t>(nnz, 1);
    E_HIP(X.cols.ensure((n + kStreamPad) * 4));   // the SpMM kernels fetch whole chunks: slack behind the last nnz
    E_HIP(X.vals.ensure((n + kStreamPad) * sizeof(T)));
    E_HIP(d_rowptr.ensure((m + 1) * 8));
    pieces_from_rowptr(h_rowptr.data(), m, X.pieces);
    X.built = true;
    if (nnz == 0) {
      Xt.pieces.start.assign(1, 0);
      Xt.built = true;
      have_matrix = true;
      return 0;
    }
    E_HIP(hipMemcpyAsync(X.cols.p, colidx, nnz * 4, kind, st()));
    E_HIP(hipMemcpyAsync(X.vals.p, vals, nnz * sizeof(T), kind, st()));
    E_HIP(hipMemcpyAsync(d_rowptr.p, h_rowptr.data(), (m + 1) * 8, hipMemcpyHostToDevice, st()));
    // validate column ids before anything gathers through them
    {
      DevBuf tmp, d_max;
      size_t bytes = 0;
      E_HIP(d_max.ensure(4));
      E_HIP(rocprim::reduce(nullptr, bytes, X.cols.template as<uint32_t>(), d_max.template as<uint32_t>(), 0u, (size_t)nnz,
                            rocprim::maximum<uint32_t>(), st()));
      E_HIP(tmp.ensure(bytes));
      E_HIP(rocprim::reduce(tmp.p, bytes, X.cols.template as<uint32_t>(), d_max.template as<uint32_t>(), 0u, (size_t)nnz,
                            rocprim::maximum<uint32_t>(), st()));
      uint32_t mx = 0;
      E_HIP(hipMemcpyAsync(&mx, d_max.p, 4, hipMemcpyDeviceToHost, st()));
      E_HIP(hipStreamSynchronize(st()));
      if ((int64_t)mx >= V) return fail(MNMF_EINVAL, "set_matrix: column index out of range");
    }
    have_matrix = true;
    return 0;
  }

  // df[c] = number of nnz in column c of X (end flags must be clear)
  int column_histogram(int32_t* df) {
    const int copies = hist_copies(V);
    DevBuf parts;
    E_HIP(parts.ensure((size_t)copies * V * 4));
    E_HIP(launch_histogram(X.cols.template as<uint32_t>(), nnz, V, ~kEndFlag, parts.template as<int32_t>(), copies, df, st()));
    E_HIP(hipStreamSynchronize(st()));  // parts is freed at scope exit
    return 0;
  }

  int clear_x_flags() {
    if (X.flags_set && nnz > 0) {
      clear_flags_kernel<<<(unsigned)cdiv(nnz, 256), 256, 0, st()>>>(X.cols.template as<uint32_t>(), nnz);
      E_LAUNCH();
    }
    X.flags_set = false;
    return 0;
  }

  // Renumber the vocabulary by descending document frequency (ties: ascending id), so that the rows of
  // H^T that are gathered most are the first rows.  df is summed over ranks through the all-reduce hook
  // before sorting, so every rank derives the same numbering (H^T is replicated and all-reduced).
  // A rank whose shard has no nnz still takes part in the df exchange (with zeros) and still builds the maps: the
  // other ranks' first collective is this one, and H^T must be in the same row order everywhere.
  int ensure_prepared() {
    if (prepared) return 0;
    renumbered = false;
    if (!tune.permute_vocab || (nnz == 0 && !ctx->sharded())) {
      prepared = true;
      return 0;
    }
    const int TB = 256;
    DevBuf df, key, key_out, iota, tmp;
    E_HIP(df.ensure((size_t)V * 4));
    E_HIP(key.ensure((size_t)V * sizeof(T)));
    E_HIP(key_out.ensure((size_t)V * sizeof(T)));
    E_HIP(iota.ensure((size_t)V * 4));
    E_HIP(d_newid.ensure((size_t)V * 4));
    E_HIP(d_order.ensure((size_t)V * 4));
    MNMF_TRY(clear_x_flags());
    if (nnz > 0) MNMF_TRY(column_histogram(df.template as<int32_t>()));
    else E_HIP(hipMemsetAsync(df.p, 0, (size_t)V * 4, st()));
    df_to_key_kernel<T><<<(unsigned)cdiv(V, TB), TB, 0, st()>>>(df.template as<int32_t>(), V, key.template as<T>(), iota.template as<uint32_t>());
    E_LAUNCH();
    MNMF_TRY(allreduce(key.template as<T>(), V));
    size_t bytes = 0;
    E_HIP(rocprim::radix_sort_pairs_desc(nullptr, bytes, key.template as<T>(), key_out.template as<T>(), iota.template as<uint32_t>(),
                                         d_order.template as<uint32_t>(), (size_t)V, 0u, (unsigned)(8 * sizeof(T)), st()));
    E_HIP(tmp.ensure(bytes));
    E_HIP(rocprim::radix_sort_pairs_desc(tmp.p, bytes, key.template as<T>(), key_out.template as<T>(), iota.template as<uint32_t>(),
                                         d_order.template as<uint32_t>(), (size_t)V, 0u, (unsigned)(8 * sizeof(T)), st()));
    invert_perm_kernel<<<(unsigned)cdiv(V, TB), TB, 0, st()>>>(d_order.template as<uint32_t>(), V, d_newid.template as<int32_t>());
    E_LAUNCH();
    if (nnz > 0) {
      remap_cols_kernel<<<(unsigned)cdiv(nnz, TB), TB, 0, st()>>>(X.cols.template as<uint32_t>(), nnz, d_newid.template as<int32_t>());
      E_LAUNCH();
    }
    E_HIP(hipStreamSynchronize(st()));
    Xt.built = false;
    Xt.seg_q = -1;
    Xth.built = false;
    Xth.seg_q = -1;
    renumbered = true;
    prepared = true;   // only now: a failure above leaves the matrix as it was and the next call tries again
    return 0;
  }

  // The all-reduce hook (hence the rank count) changed: the renumbering must be derived again.
  int hook_changed() override {
    if (!have_matrix || !prepared) return 0;
    E_HIP(hipSetDevice(ctx->device));
    if (renumbered && nnz > 0) {
      MNMF_TRY(clear_x_flags());
      // back to the caller's ids: order[new] = old, as int32 map
      remap_cols_kernel<<<(unsigned)cdiv(nnz, 256), 256, 0, st()>>>(X.cols.template as<uint32_t>(), nnz, d_order.template as<int32_t>());
      E_LAUNCH();
      E_HIP(hipStreamSynchronize(st()));
    }
    prepared = false;
    renumbered = false;
    Xt.built = false;
    Xt.seg_q = -1;
    Xth.built = false;
    Xth.seg_q = -1;
    return 0;
  }

  // Build X^T in processing order (see sparse_prep.hpp): the n_hot most frequent words as a tile stream (tiles of
  // tile_rows documents, Xth), the rest blocked by document blocks of 2^bs rows (Xt).  n_hot == 0: everything in Xt.
  // sweep_row_bytes > 0: the rare words go to the sweep stream if their accumulators (that many LDS bytes per word) fit
  // a partition's LDS budget and the documents are gathered often enough per XCD for the shared window to pay
  int ensure_xt(int bs, int n_hot, int tile_rows, size_t sweep_row_bytes = 0) {
    if (nnz == 0) return 0;
    if (n_hot <= 0 || tile_rows <= 0) n_hot = 0, tile_rows = 1;
    const uint32_t n_tiles = n_hot > 0 ? (uint32_t)cdiv(m, tile_rows) : 0;
    if ((uint64_t)n_tiles * (uint64_t)n_hot >= ((uint64_t)1 << 31)) n_hot = 0;
    const uint32_t hot_span = n_hot > 0 ? n_tiles * (uint32_t)n_hot : 0u;
    while ((uint64_t)hot_span + (uint64_t)(cdiv(m, (int64_t)1 << bs) + 1) * (uint64_t)V >= ((uint64_t)1 << 32)) ++bs;
    if (Xt.built && Xt.block_shift == bs && Xth.n_hot == n_hot && (n_hot == 0 || Xth.tile_rows == tile_rows) &&
        Xs.req_row_bytes == sweep_row_bytes)
      return 0;
    const uint32_t n_blocks = (uint32_t)cdiv(m, (int64_t)1 << bs);
    const size_t n = (size_t)nnz;
    const int TB = 256;
    DevBuf rowid, iota, key, keys_out, perm, tmp, df, flag, pos, cnt, pkey;
    E_HIP(rowid.ensure(n * 4));
    E_HIP(iota.ensure(n * 4));
    E_HIP(key.ensure(n * 4));
    E_HIP(keys_out.ensure(n * 4));
    E_HIP(perm.ensure(n * 4));
    E_HIP(df.ensure((size_t)V * 4));
    E_HIP(cnt.ensure(8));
    E_HIP(hipMemsetAsync(df.p, 0, (size_t)V * 4, st()));
    MNMF_TRY(clear_x_flags());  // the histogram / keys need plain column ids
    expand_rowids_kernel<<<(unsigned)cdiv(nnz, TB), TB, 0, st()>>>(d_rowptr.template as<int64_t>(), m, nnz,
                                                                   rowid.template as<uint32_t>(), iota.template as<uint32_t>());
    E_LAUNCH();
    MNMF_TRY(column_histogram(df.template as<int32_t>()));
    int64_t n_hot_nnz = 0;
    if (n_hot > 0) {  // nnz of the tile stream = sum of the first n_hot document frequencies
      size_t bytes = 0;
      E_HIP(rocprim::reduce(nullptr, bytes, df.template as<int32_t>(), cnt.template as<int32_t>(), 0, (size_t)n_hot,
                            rocprim::plus<int32_t>(), st()));
      E_HIP(tmp.ensure(bytes));
      E_HIP(rocprim::reduce(tmp.p, bytes, df.template as<int32_t>(), cnt.template as<int32_t>(), 0, (size_t)n_hot,
                            rocprim::plus<int32_t>(), st()));
      int32_t h = 0;
      E_HIP(hipMemcpyAsync(&h, cnt.p, 4, hipMemcpyDeviceToHost, st()));
      E_HIP(hipStreamSynchronize(st()));
      n_hot_nnz = h;
    }
    E_HIP(Xt.cols.ensure((n - (size_t)n_hot_nnz + kStreamPad) * 4));
    E_HIP(Xt.vals.ensure((n - (size_t)n_hot_nnz + kStreamPad) * sizeof(T)));
    E_HIP(Xth.cols.ensure(((size_t)n_hot_nnz + kStreamPad) * 4));
    E_HIP(Xth.vals.ensure(((size_t)n_hot_nnz + kStreamPad) * sizeof(T)));
    // a word is cut at document-block boundaries only if its pieces stay long enough to pay for their carry slots
    const int32_t long_df = (int32_t)std::min<int64_t>(std::max<int64_t>(tune.xt_long_floor, (int64_t)tune.xt_min_piece * n_blocks), 1 << 30);
    // ---- sweep stream of the rare words (not cut at document blocks: df <= long_df)
    Xs = SweepMat();
    Xs.req_row_bytes = sweep_row_bytes;
    SweepMap sw{nullptr, 0, 1, 0, 0};
    DevBuf rare, rare_df, rare_idx;
    int64_t n_sweep = 0;
    if (sweep_row_bytes > 0 && tune.sweep > 0) {
      E_HIP(rare.ensure((size_t)V * 4));
      E_HIP(rare_df.ensure((size_t)V * 4));
      E_HIP(rare_idx.ensure((size_t)V * 4));
      int ws = 8;   // window: about sweep_win_bytes of table rows (ld * sizeof(T) bytes each = sweep_row_bytes less the padding)
      while (ws < 24 && ((int64_t)2 << ws) * (int64_t)sweep_row_bytes <= tune.sweep_win_bytes) ++ws;
      const int64_t n_win = cdiv(m, (int64_t)1 << ws);
      // which words: the uncut ones, and up to about 32 nnz per window (one lane group accumulates a word's nnz of a
      // window one after the other; beyond that the heaviest word holds its window back: configs[1], 32 windows,
      // X^T.W with df <= 700 / 1000 / 1500 / 2000 / 4000: 1.197 / 1.196 / 1.227 / 1.259 / 1.397 ms)
      const int64_t auto_df = tune.sweep_max_df > 0 ? tune.sweep_max_df : 32 * n_win;
      const int32_t sweep_df = (int32_t)std::min<int64_t>(std::max<int64_t>(long_df, auto_df), 1 << 30);
      // The accumulator rows of all partitions hold `cap` words.  When more words qualify (wide rows: few rows per
      // partition; large vocabularies), the rarest ones stay in the blocked stream: words with lo < df <= sweep_df go to
      // the sweep, lo doubled until they fit -- a word with a handful of nnz brings the least work per accumulator row.
      const int P = n_cus;
      const int64_t cap = (int64_t)P * (int64_t)(tune.sweep_lds_bytes / (int64_t)sweep_row_bytes);
      int32_t h2[2] = {0, 0};
      int32_t sweep_lo = 0;
      for (;;) {
        rare_mark_kernel<<<(unsigned)cdiv(V, TB), TB, 0, st()>>>(df.template as<int32_t>(), V, (uint32_t)n_hot, sweep_lo, sweep_df,
                                                                 rare.template as<int32_t>(), rare_df.template as<int32_t>());
        E_LAUNCH();
        for (int which = 0; which < 2; ++which) {   // number of rare words, and their nnz
          const int32_t* src = which == 0 ? rare.template as<int32_t>() : rare_df.template as<int32_t>();
          size_t bytes = 0;
          E_HIP(rocprim::reduce(nullptr, bytes, src, cnt.template as<int32_t>(), 0, (size_t)V, rocprim::plus<int32_t>(), st()));
          E_HIP(tmp.ensure(bytes));
          E_HIP(rocprim::reduce(tmp.p, bytes, src, cnt.template as<int32_t>(), 0, (size_t)V, rocprim::plus<int32_t>(), st()));
          E_HIP(hipMemcpyAsync(&h2[which], cnt.p, 4, hipMemcpyDeviceToHost, st()));
          E_HIP(hipStreamSynchronize(st()));
        }
        if (h2[0] <= cap || cap <= 0 || sweep_lo >= sweep_df) break;
        sweep_lo = sweep_lo > 0 ? sweep_lo * 2 : 1;
      }
      const int64_t n_rare = h2[0], rare_nnz = h2[1];
      const int64_t rpp = std::max<int64_t>(1, cdiv(n_rare, P));
      const uint64_t sbase = (uint64_t)hot_span + (uint64_t)(n_blocks + 1) * (uint64_t)V;
      const bool fits = n_rare > 0 && (uint64_t)rpp * sweep_row_bytes <= (uint64_t)tune.sweep_lds_bytes &&
                        sbase + (uint64_t)n_win * P * (uint64_t)rpp < ((uint64_t)1 << 32) && rare_nnz < ((int64_t)1 << 31) &&
                        m < ((int64_t)1 << 24) && (uint64_t)m * sweep_row_bytes < ((uint64_t)1 << 32) &&
                        n_win * P * 64 < ((int64_t)1 << 30);
      // a document's row is fetched from the fabric once per XCD and sweep; it pays when the XCD then uses it again
      // ... and when the table is beyond what the L2s hold together (configs[0], 2.5 MB of W: X^T.W 0.055 -> 0.093 ms with
      // the sweep; there every gather of the blocked stream is an L2 hit anyway)
      const bool pays = rare_nnz * 100 >= (int64_t)tune.sweep_min_touch_pct * 8 * m && nnz >= tune.hot_min_nnz &&
                        (int64_t)m * (int64_t)sweep_row_bytes >= tune.sweep_min_table_bytes;
      if (fits && (pays || tune.sweep >= 2)) {
        size_t bytes = 0;
        E_HIP(rocprim::exclusive_scan(nullptr, bytes, rare.template as<int32_t>(), rare_idx.template as<int32_t>(), 0, (size_t)V,
                                      rocprim::plus<int32_t>(), st()));
        E_HIP(tmp.ensure(bytes));
        E_HIP(rocprim::exclusive_scan(tmp.p, bytes, rare.template as<int32_t>(), rare_idx.template as<int32_t>(), 0, (size_t)V,
                                      rocprim::plus<int32_t>(), st()));
        E_HIP(Xs.words.ensure((size_t)rpp * P * 4));
        E_HIP(hipMemsetAsync(Xs.words.p, 0xFF, (size_t)rpp * P * 4, st()));
        rare_place_kernel<<<(unsigned)cdiv(V, TB), TB, 0, st()>>>(rare.template as<int32_t>(), rare_idx.template as<int32_t>(), V,
                                                                  (uint32_t)P, Xs.words.template as<int32_t>());
        E_LAUNCH();
        n_sweep = rare_nnz;
        Xs.nnz = rare_nnz;
        Xs.P = P;
        Xs.rpp = (int)rpp;
        Xs.win_shift = ws;
        Xs.n_windows = (int)n_win;
        Xs.base = (uint32_t)sbase;
        sw = SweepMap{rare_idx.template as<int32_t>(), (uint32_t)P, (uint32_t)rpp, (uint32_t)ws, (uint32_t)sbase};
        E_HIP(Xs.docs.ensure((size_t)n_sweep * 4));
        E_HIP(Xs.lrows.ensure((size_t)n_sweep * 4));
        E_HIP(Xs.vals.ensure((size_t)n_sweep * sizeof(T)));
        E_HIP(Xs.keys.ensure((size_t)n_sweep * 4));
      }
    }
    transpose_key_kernel<<<(unsigned)cdiv(nnz, TB), TB, 0, st()>>>(X.cols.template as<uint32_t>(), rowid.template as<uint32_t>(),
                                                                   df.template as<int32_t>(), nnz, V, bs, n_blocks,
                                                                   long_df, (uint32_t)n_hot, (uint32_t)tile_rows,
                                                                   hot_span, sw, key.template as<uint32_t>());
    E_LAUNCH();
    {  // stable sort by key: documents stay ascending inside a piece
      int bits = 1;
      const int64_t key_end = (int64_t)hot_span + ((int64_t)n_blocks + 1) * V + (n_sweep > 0 ? (int64_t)Xs.n_windows * Xs.P * Xs.rpp : 0);
      while (((int64_t)1 << bits) < key_end) ++bits;
      size_t bytes = 0;
      E_HIP(rocprim::radix_sort_pairs(nullptr, bytes, key.template as<uint32_t>(), keys_out.template as<uint32_t>(),
                                      iota.template as<uint32_t>(), perm.template as<uint32_t>(), n, 0u, (unsigned)bits, st()));
      E_HIP(tmp.ensure(bytes));
      E_HIP(rocprim::radix_sort_pairs(tmp.p, bytes, key.template as<uint32_t>(), keys_out.template as<uint32_t>(),
                                      iota.template as<uint32_t>(), perm.template as<uint32_t>(), n, 0u, (unsigned)bits, st()));
    }
    gather_transposed_kernel<T><<<(unsigned)cdiv(nnz, TB), TB, 0, st()>>>(
        perm.template as<uint32_t>(), rowid.template as<uint32_t>(), X.vals.template as<T>(), nnz, n_hot_nnz, (uint32_t)tile_rows,
        Xth.cols.template as<uint32_t>(), Xth.vals.template as<T>(), Xt.cols.template as<uint32_t>(), Xt.vals.template as<T>(),
        n_sweep, keys_out.template as<uint32_t>(), Xs.base, (uint32_t)std::max(Xs.rpp, 1), Xs.docs.template as<uint32_t>(),
        Xs.lrows.template as<uint32_t>(), Xs.vals.template as<T>());
    E_LAUNCH();
    if (n_sweep > 0) {  // the sorted keys of the sweep stream stay: the segment table is cut from them (ensure_sweep_segments)
      E_HIP(hipMemcpyAsync(Xs.keys.p, keys_out.template as<uint32_t>() + (nnz - n_sweep), (size_t)n_sweep * 4, hipMemcpyDeviceToDevice, st()));
      const int64_t n_cells = (int64_t)Xs.n_windows * Xs.P;
      E_HIP(Xs.cell_start.ensure((size_t)(n_cells + 1) * 4));
      sweep_cells_kernel<<<(unsigned)cdiv(n_cells + 1, TB), TB, 0, st()>>>(Xs.keys.template as<uint32_t>(), n_sweep, Xs.base,
                                                                          (uint32_t)Xs.rpp, n_cells, Xs.cell_start.template as<int32_t>());
      E_LAUNCH();
    }
    // From here on only the tile and the blocked stream: pieces are cut on the host from their part of the sorted keys.
    // (Names below keep `nnz` for the length of that part.)
    return finish_xt(bs, n_hot, tile_rows, n_tiles, hot_span, n_blocks, n_hot_nnz, nnz - n_sweep, keys_out, iota, flag, cnt, tmp, pkey);
  }

  // second half of ensure_xt: `nnz` here is the length of the tile + blocked part of the sorted keys
  int finish_xt(int bs, int n_hot, int tile_rows, uint32_t n_tiles, uint32_t hot_span, uint32_t n_blocks, int64_t n_hot_nnz,
                int64_t nnz, DevBuf& keys_out, DevBuf& iota, DevBuf& flag, DevBuf& cnt, DevBuf& tmp, DevBuf& pkey) {
    const int TB = 256;
    const size_t n = (size_t)nnz;
    // piece starts = positions where the key changes
    E_HIP(flag.ensure(std::max<size_t>(n, 1)));
    if (nnz > 0) {
      piece_flag_kernel<<<(unsigned)cdiv(nnz, TB), TB, 0, st()>>>(keys_out.template as<uint32_t>(), nnz, flag.template as<uint8_t>());
      E_LAUNCH();
    }
    uint32_t* d_pos = iota.template as<uint32_t>();  // iota is dead after the sort: reuse as the position list
    {
      size_t bytes = 0;
      E_HIP(rocprim::select(nullptr, bytes, rocprim::counting_iterator<uint32_t>(0), flag.template as<uint8_t>(), d_pos,
                            cnt.template as<uint32_t>(), n, st()));
      E_HIP(tmp.ensure(bytes));
      E_HIP(rocprim::select(tmp.p, bytes, rocprim::counting_iterator<uint32_t>(0), flag.template as<uint8_t>(), d_pos,
                            cnt.template as<uint32_t>(), n, st()));
    }
    uint32_t n_pieces = 0;
    E_HIP(hipMemcpyAsync(&n_pieces, cnt.p, 4, hipMemcpyDeviceToHost, st()));
    E_HIP(hipStreamSynchronize(st()));
    E_HIP(pkey.ensure((size_t)std::max<uint32_t>(n_pieces, 1) * 4));
    gather_u32_kernel<<<(unsigned)cdiv(n_pieces, TB), TB, 0, st()>>>(keys_out.template as<uint32_t>(), d_pos, n_pieces,
                                                                     pkey.template as<uint32_t>());
    E_LAUNCH();
    std::vector<uint32_t> h_pos(n_pieces), h_key(n_pieces);
    E_HIP(hipMemcpyAsync(h_pos.data(), d_pos, (size_t)n_pieces * 4, hipMemcpyDeviceToHost, st()));
    E_HIP(hipMemcpyAsync(h_key.data(), pkey.p, (size_t)n_pieces * 4, hipMemcpyDeviceToHost, st()));
    E_HIP(hipStreamSynchronize(st()));
    HostPieces& hp = Xth.pieces;
    HostPieces& cp = Xt.pieces;
    hp = HostPieces();
    cp = HostPieces();
    for (uint32_t i = 0; i < n_pieces; ++i) {
      if (h_key[i] < hot_span) {
        hp.start.push_back(h_pos[i]);
        hp.row.push_back((int32_t)(h_key[i] % (uint32_t)n_hot));
        hp.block.push_back((int32_t)(h_key[i] / (uint32_t)n_hot));
      } else {
        const uint64_t kk = (uint64_t)h_key[i] - hot_span;
        cp.start.push_back((int64_t)h_pos[i] - n_hot_nnz);
        cp.row.push_back((int32_t)(kk % (uint64_t)V));
        const uint32_t b = (uint32_t)(kk / (uint64_t)V);
        cp.block.push_back(b >= n_blocks ? -1 : (int32_t)b);
      }
    }
    hp.start.push_back(n_hot_nnz);
    cp.start.push_back(nnz - n_hot_nnz);
    Xt.nnz = nnz - n_hot_nnz;
    Xt.block_shift = bs;
    Xt.built = true;
    Xt.flags_set = false;
    Xt.seg_q = -1;
    Xth.nnz = n_hot_nnz;
    Xth.n_hot = n_hot;
    Xth.tile_rows = n_hot > 0 ? tile_rows : 0;
    Xth.n_tiles = n_tiles;
    Xth.built = true;
    Xth.flags_set = false;
    Xth.seg_q = -1;
    return 0;
  }

  int upload(DevBuf& b, const void* src, size_t bytes) {
    E_HIP(b.ensure(std::max<size_t>(bytes, 16)));
    if (bytes) E_HIP(hipMemcpyAsync(b.p, src, bytes, hipMemcpyHostToDevice, st()));
    return 0;
  }

  int ensure_segments(StreamMat<T>& M, int64_t q, int xcd_group, int ng) {
    if (M.seg_q == q && M.seg_group == xcd_group && M.flags_set) return 0;
    HostSegments s;
    std::vector<int32_t> tile_first;
    const bool tiled = M.tile_rows > 0;
    if (tiled && q <= 0) build_tile_segments_host(M.pieces, M.rows, M.n_tiles, (int)(-q), kSubPiece, s, tile_first);  // -q = groups
    else {
      // Exact-length segments (pieces cut at the segment boundary) for small streams: the whole stream is then about one
      // round of the resident waves, the kernel lasts as long as its slowest wave, and a wave as long as the longest of
      // its segments (configs[0], 1.8e6 nnz: X^T.W 0.054 -> 0.036 ms, X.H^T 0.033 -> 0.028 ms, 8740 -> 10870 it/s; 30k
      // documents: X^T.W 0.055 -> 0.044; 50k: X^T.W 0.060 -> 0.057 but X.H^T 0.048 -> 0.053).  With more rounds the
      // dispatcher evens the waves out and the extra carry slots of the cut pieces cost more than they save (100k x
      // 50k, k = 20: X^T.W 0.096 -> 0.113 ms; configs[1]: 1.355 -> 1.404 ms).
      (void)ng;
      const bool few_waves = nnz <= (&M == &Xt ? (int64_t)5000000 : (int64_t)2200000);   // the whole matrix, not this stream's share
      const bool exact = tune.exact_seg < 0 ? (few_waves && !tiled) : (tune.exact_seg > 0 && (&M == &Xt || tune.exact_seg >= 2));
      build_segments_host(M.pieces, M.rows, q, kSubPiece, s, xcd_group, tiled ? &tile_first : nullptr, M.n_tiles, exact);
    }
    if (tiled) MNMF_TRY(upload(M.tile_first, tile_first.data(), tile_first.size() * 4));
    M.seg_group = xcd_group;
    M.n_seg = s.n_seg();
    M.n_dest = (int64_t)s.dest.size();
    M.n_empty = (int64_t)s.empty_rows.size();
    M.n_fix = (int64_t)s.fix_row.size();
    M.n_carry = s.n_carry;
    {  // rows with long carry lists first: they get a block each (spmm_fixup_long_kernel), the rest a wave
      std::vector<int32_t> order(s.fix_row.size());
      for (size_t i = 0; i < order.size(); ++i) order[i] = (int32_t)i;
      auto mid = std::stable_partition(order.begin(), order.end(), [&](int32_t i) { return s.fix_count[i] > kLongFix; });
      M.n_fix_long = (int64_t)(mid - order.begin());
      std::vector<int32_t> r(order.size()), f(order.size()), c(order.size());
      for (size_t i = 0; i < order.size(); ++i) {
        r[i] = s.fix_row[order[i]];
        f[i] = s.fix_first[order[i]];
        c[i] = s.fix_count[order[i]];
      }
      s.fix_row.swap(r);
      s.fix_first.swap(f);
      s.fix_count.swap(c);
    }
    // X^T beside a tile or sweep stream: the rows those streams write need no zeroing -- at configs[1] that is all but
    // a handful of the 95k "rows without nnz" of the blocked stream (24 MB of zeros per X^T.W that were overwritten
    // right away).  The tile stream owns the rows of the words with a piece in some tile, the sweep those of its word table.
    if (&M == &Xt && !s.empty_rows.empty() && (Xth.nnz > 0 || Xs.nnz > 0)) {
      std::vector<char> owned((size_t)M.rows, 0);
      if (Xth.nnz > 0)
        for (int32_t r : Xth.pieces.row)   // the words that have a piece in some tile
          if (r >= 0 && r < M.rows) owned[(size_t)r] = 1;
      if (Xs.nnz > 0) {
        std::vector<int32_t> words((size_t)Xs.rpp * Xs.P);
        E_HIP(hipMemcpyAsync(words.data(), Xs.words.p, words.size() * 4, hipMemcpyDeviceToHost, st()));
        E_HIP(hipStreamSynchronize(st()));
        for (int32_t w : words)
          if (w >= 0 && w < M.rows) owned[(size_t)w] = 1;
      }
      std::vector<int32_t> keep;
      for (int32_t r : s.empty_rows)
        if (!owned[(size_t)r]) keep.push_back(r);
      s.empty_rows.swap(keep);
      M.n_empty = (int64_t)s.empty_rows.size();
    }
    // Small streams that are alone on their output (X; X^T without a tile or sweep stream beside it): one fix-up launch
    // for everything, and the rows without nnz join it as rows with no carry slots instead of a zeroing launch of their
    // own in front of the stream kernel.
    {
      const bool alone = &M == &X || (&M == &Xt && Xth.nnz == 0 && Xs.nnz == 0);
      const int64_t fix_bytes = (M.n_carry + M.n_empty) * (int64_t)128;
      M.fix_merged = tune.fix_merge && alone && !tiled && fix_bytes <= tune.fix_merge_bytes && M.n_fix_long <= 4096;
      if (M.fix_merged) {
        for (int32_t r : s.empty_rows) {
          s.fix_row.push_back(r);
          s.fix_first.push_back(0);
          s.fix_count.push_back(0);
        }
        s.empty_rows.clear();
        M.n_empty = 0;
        M.n_fix = (int64_t)s.fix_row.size();
      }
    }
    {  // one 16-byte descriptor per segment: a lane group fetches it with a single load
      std::vector<SegDesc> sd((size_t)s.n_seg());
      for (size_t i = 0; i < sd.size(); ++i) sd[i] = SegDesc{s.seg_start[i], s.seg_len[i], s.seg_info[i]};
      MNMF_TRY(upload(M.seg_desc, sd.data(), sd.size() * sizeof(SegDesc)));
    }
    s.dest.resize(s.dest.size() + kStreamPad, 0);   // the kernels fetch whole chunks of the list
    MNMF_TRY(upload(M.dest, s.dest.data(), s.dest.size() * 4));
    MNMF_TRY(upload(M.empty_rows, s.empty_rows.data(), s.empty_rows.size() * 4));
    MNMF_TRY(upload(M.fix_row, s.fix_row.data(), s.fix_row.size() * 4));
    MNMF_TRY(upload(M.fix_first, s.fix_first.data(), s.fix_first.size() * 4));
    MNMF_TRY(upload(M.fix_count, s.fix_count.data(), s.fix_count.size() * 4));
    if (!M.flags_set && !s.piece_end.empty()) {  // piece ends do not depend on q
      DevBuf pe;
      MNMF_TRY(upload(pe, s.piece_end.data(), s.piece_end.size() * 8));
      set_flags_at_kernel<<<(unsigned)cdiv((int64_t)s.piece_end.size(), 256), 256, 0, st()>>>(
          pe.template as<int64_t>(), (int64_t)s.piece_end.size(), M.cols.template as<uint32_t>());
      E_LAUNCH();
      E_HIP(hipStreamSynchronize(st()));
    }
    M.flags_set = true;
    E_HIP(hipStreamSynchronize(st()));  // host vectors die at scope exit
    M.seg_q = q;
    return 0;
  }

  // ------------------------------------------------------------------ kernels
  // B, out and the carry buffer are already offset to the first of the `nc` columns this launch handles
  template <int LPR, int VPL>
  int spmm_launch(StreamMat<T>& M, const T* B, int ld, int nc, int c0, T* out, int hot, int lds_ld) {
    constexpr int NG = kWave / LPR;
    constexpr size_t kLdsSlack = (size_t)LPR * VPL * 16;   // idle lanes read past the last row (spmm.hpp, lbytes)
    const int64_t waves = cdiv(M.n_seg, NG);
    const size_t tab_bytes = ((size_t)M.tab_rows * ld - (size_t)c0) * sizeof(T);
    const bool wide = tune.force_wide || M.tab_rows >= ((int64_t)1 << 24) || tab_bytes >= ((size_t)1 << 32) ||
                      (size_t)ld * sizeof(T) >= ((size_t)1 << 24);
    T* carry = M.carry.template as<T>() + c0;
    SpmmArgs<T> a{M.seg_desc.template as<SegDesc>(), M.n_seg,
                  M.cols.template as<uint32_t>(), M.vals.template as<T>(), M.dest.template as<int32_t>(), M.n_dest,
                  B, ld, nc, out, carry, hot, lds_ld, wide ? 0u : (uint32_t)tab_bytes,
                  M.tile_first.template as<int32_t>(), (int)M.n_tiles, M.tile_rows, M.tab_rows};
    a.kreal = (c0 == 0 && lds_ld < ld) ? std::min(nc, lds_ld) : nc;   // compact LDS images (spmm(): lds_compact)
    if (&M == &Xth) {
      if (waves > 0) {  // tile stream: every gather is an LDS read (indices are local to the tile)
        auto kern = spmm_stream_kernel<T, LPR, VPL, 1, SPMM_TILE, false>;
        const size_t lds_bytes = (size_t)M.tile_rows * lds_ld * sizeof(T) + kLdsSlack;
        E_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
        a.hot_rows = 0;
        const int wgs = std::max(1, tune.tile_wgs_per_cu);  // workgroups per CU: one stages while another computes
        kern<<<(unsigned)std::min<int64_t>((int64_t)n_cus * wgs, M.n_tiles), std::min(tune.tile_threads, 1024 / wgs), lds_bytes, st()>>>(a);
        E_LAUNCH();
      }
    } else if (waves > 0 && hot > 0 && !wide) {
      auto kern = spmm_stream_kernel<T, LPR, VPL, 0, SPMM_HOT, false>;
      const size_t lds_bytes = (size_t)hot * lds_ld * sizeof(T) + kLdsSlack;
      E_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
      const int wgs = std::max(1, tune.hot_wgs_per_cu);
      kern<<<(unsigned)(n_cus * wgs), std::min(tune.hot_threads, 1024), lds_bytes, st()>>>(a);
      E_LAUNCH();
    } else if (waves > 0 && !wide && tune.queue && (&M == &Xt || tune.queue >= 2)) {
      // persistent PLAIN: lane groups walk their segment queues back to back (spmm.hpp, SPMM_QUEUE).  The grid is what
      // is resident at once, a multiple of 8 workgroups so that workgroup w still runs the segments meant for XCD w % 8.
      a.hot_rows = 0;
      auto launch_queue = [&](auto kern) -> int {
        int& occ = occ_cache[reinterpret_cast<const void*>(kern)];
        if (occ == 0) {
          int o = 0;
          E_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&o, kern, 256, 0));
          occ = std::max(1, o);
        }
        const int per_cu = tune.queue_wgs > 0 ? tune.queue_wgs : occ;
        const unsigned grid = (unsigned)std::min<int64_t>(round_up(cdiv(waves, 4), 8), (int64_t)n_cus * per_cu);
        kern<<<grid, 256, 0, st()>>>(a);
        E_LAUNCH();
        return 0;
      };
      if (&M == &Xt) MNMF_TRY(launch_queue(spmm_stream_kernel<T, LPR, VPL, 1, SPMM_QUEUE, false>));
      else MNMF_TRY(launch_queue(spmm_stream_kernel<T, LPR, VPL, 0, SPMM_QUEUE, false>));
    } else if (waves > 0) {
      a.hot_rows = 0;
      const unsigned grid = (unsigned)cdiv(waves, 4);
      if (&M == &Xt) {
        if (wide) spmm_stream_kernel<T, LPR, VPL, 1, SPMM_PLAIN, true><<<grid, 256, 0, st()>>>(a);
        else spmm_stream_kernel<T, LPR, VPL, 1, SPMM_PLAIN, false><<<grid, 256, 0, st()>>>(a);
      } else {
        if (wide) spmm_stream_kernel<T, LPR, VPL, 0, SPMM_PLAIN, true><<<grid, 256, 0, st()>>>(a);
        else spmm_stream_kernel<T, LPR, VPL, 0, SPMM_PLAIN, false><<<grid, 256, 0, st()>>>(a);
      }
      E_LAUNCH();
    }
    const int32_t* fr = M.fix_row.template as<int32_t>();
    const int32_t* ff = M.fix_first.template as<int32_t>();
    const int32_t* fc = M.fix_count.template as<int32_t>();
    // small streams: long rows, short rows and the rows without nnz (ensure_segments) in ONE launch -- at configs[0] a
    // launch costs as much as the work it starts
    if (M.fix_merged) {
      if (M.n_fix > 0) {
        const int64_t n_short = M.n_fix - M.n_fix_long;
        spmm_fixup_both_kernel<T, LPR, VPL><<<(unsigned)(M.n_fix_long + cdiv(n_short, 4)), 256, 0, st()>>>(
            fr, ff, fc, (int)M.n_fix_long, M.n_fix, carry, ld, nc, out);
        E_LAUNCH();
      }
      return 0;
    }
    if (M.n_fix_long > 0) {
      // few rows with very long lists (tile stream): 16 waves per row; many rows with moderate lists: 4
      if (M.n_fix_long <= (int64_t)8 * n_cus)
        spmm_fixup_long_kernel<T, LPR, VPL, 16><<<(unsigned)M.n_fix_long, 1024, 0, st()>>>(fr, ff, fc, carry, ld, nc, out);
      else
        spmm_fixup_long_kernel<T, LPR, VPL, 4><<<(unsigned)M.n_fix_long, 256, 0, st()>>>(fr, ff, fc, carry, ld, nc, out);
      E_LAUNCH();
    }
    if (M.n_fix > M.n_fix_long) {
      const int64_t n = M.n_fix - M.n_fix_long, o = M.n_fix_long;
      spmm_fixup_kernel<T, LPR, VPL><<<(unsigned)cdiv(n, 4), 256, 0, st()>>>(fr + o, ff + o, fc + o, n, carry, ld, nc, out);
      E_LAUNCH();
    }
    return 0;
  }

  // out (M.rows x ld) = M . B   (B: M.tab_rows x ld)
  int spmm(StreamMat<T>& M, const T* B, int ld, T* out) {
    constexpr int VEC = VecOf<T>::N;
    if (ld % VEC != 0) return fail(MNMF_EINVAL, "spmm: the leading dimension must be a whole number of 16-byte vectors");
    if (ld / VEC > 128) {
      // A wave covers at most 128 vectors of a row (512 floats / 256 doubles): wider factors go through the plain
      // stream in column chunks of that width -- the nnz stream is read once per chunk, nothing else changes.
      MNMF_TRY(ensure_prepared());
      if (&M == &Xt) {
        int bs = 8;
        while (bs < 20 && ((int64_t)2 << bs) * ld * (int64_t)sizeof(T) <= xt_block_limit(64)) ++bs;
        MNMF_TRY(ensure_xt(bs, 0, 0));
      }
      MNMF_TRY(spmm_stream(M, B, ld, out, 64, 2, 1, ld, 1));   // segments, carry buffer, empty rows
      for (int c0 = 0; c0 < ld; c0 += 128 * VEC) {
        const int nc = std::min(128 * VEC, ld - c0);
        const int nv = nc / VEC;
        if (nv > 64) MNMF_TRY((spmm_launch<64, 2>(M, B + c0, ld, nc, c0, out + c0, 0, ld)));
        else MNMF_TRY(spmm_dispatch(M, B + c0, ld, nc, c0, out + c0, 0, ld, 1, std::max(4, pow2_ceil(nv))));
      }
      return 0;
    }
    const int nvec = ld / VEC;
    int vpl = nvec > 64 ? 2 : 1;
    int lpr = vpl == 2 ? 64 : std::max(4, pow2_ceil(nvec));
    // rows of 17..32 vectors (272..512 bytes: fp64 k = 33..64, fp32 k = 65..128): 16 lanes x 2 vectors per row instead
    // of 32 x 1 -- four rows per wave-level gather as the narrower rows have, half the steps per nnz, and the 16-lane
    // LDS paths (hot rows of H^T, W tiles) that never paid at 32 lanes (hot_x_threshold).  configs[1] in fp64 155 -> 161
    // it/s, configs[2] (k = 100, fp32) 153 -> 161
    // (the same one step up -- 33..64 vectors on 32 lanes x 2 instead of 64 x 1 -- lost: configs[4] shard 27.63 -> 27.03
    // it/s, configs[2] in fp64 84.4 -> 79.8)
    // ... and one step down -- 9..16 vectors on 8 lanes x 2 instead of 16 x 1 -- too: configs[1] X.H^T 1.14 -> 1.38 ms,
    // X^T.W 1.18 -> 1.26.  Sixteen lanes per row it is.
    if (tune.two_vec && lpr == 32) lpr = 16, vpl = 2;
    const int ng = kWave / lpr;
    // LDS images of factor rows (hot rows of H^T, W tiles): the row in memory plus padding, or -- when the solver says
    // how many columns hold data (k = 50 lives in rows of 64) -- only those: a quarter more rows in the same LDS
    const int lds_ld_sw = ld + lds_pad_for(ld, lpr);   // the sweep's accumulator rows keep the memory layout
    const int dcols = (int)round_up(data_cols, VEC);
    // Not for the tile stream of standard NMF with one 128-byte line per row (k <= 32 floats / 16 doubles): the tiles,
    // and with them the order in which a word's partial sums are added, would depend on k, and a problem in a batched
    // topic-count sweep (run_nmf_sweep, all of them side by side in the line) must see exactly the sums of its own
    // run_nmf.  Which rows of H^T come from LDS in X.H^T changes no sum.  (MNMF_LDS_COMPACT=2: compact there too.)
    const bool compact = tune.lds_compact && data_cols > 0 && dcols < ld && !(&M == &Xt && line_layout && tune.lds_compact < 2);
    const int lds_ld = compact ? dcols : lds_ld_sw;
    MNMF_TRY(ensure_prepared());
    if (&M == &Xt) {
      // document block = rows of W worth about one XCD's 4 MiB L2 (measured optimum at C2 / C3 / C4: 4 MiB against
      // 2 MiB +0.6 / +2.7 / +1.9 %, 8 MiB -4 %: fewer pieces and carries against L2 residency), but never so large that
      // a small matrix is left with fewer than about four blocks (C1, 2.5 MB of W: one block -13 %, four or more +9 %)
      const int64_t w_bytes = std::max<int64_t>(m, 1) * ld * (int64_t)sizeof(T);
      const int64_t block_bytes = std::min<int64_t>(xt_block_limit(lpr), std::max<int64_t>((int64_t)256 << 10, w_bytes / 4));
      int bs = 8;
      while (bs < 20 && ((int64_t)2 << bs) * ld * (int64_t)sizeof(T) <= block_bytes) ++bs;
      // tile stream for the most frequent words: documents per tile = what one workgroup's LDS holds
      int n_hot = 0, tile_rows = 0;
      if (vmap() && tune.xt_hot_words != 0 && nnz >= tune.hot_min_nnz) {
        tile_rows = (int)std::min<int64_t>(tune.tile_lds_bytes / tune.tile_wgs_per_cu / ((int64_t)lds_ld * (int64_t)sizeof(T)), 1 << 15);
        if (lds_ld < lds_ld_sw) {
          // compact rows make room for more documents per tile, but not at the price of the tiles that keep the CUs busy
          // (100k x 50k, k = 20, fp64: 950-document tiles are 105 tiles, under the minimum below -- the tile stream was
          // dropped and X^T.W went 0.208 -> 0.287 ms; NMTF 200k x 50k with 155 tiles of 1298 documents instead of 165 of
          // 1218: X^T.W 0.118 -> 0.123): as many documents as fit while there are two tiles per CU, never fewer than the
          // padded layout holds -- the large matrices gain (configs[3] +5 %), the others keep their tiles
          const int64_t padded = std::min<int64_t>(tune.tile_lds_bytes / tune.tile_wgs_per_cu / ((int64_t)lds_ld_sw * (int64_t)sizeof(T)), 1 << 15);
          tile_rows = (int)std::max<int64_t>(padded, std::min<int64_t>(tile_rows, m / (2 * (int64_t)n_cus)));
        }
        // every (tile, hot word) pair is a carry row: with wide rows (few documents per tile, many tiles) half as many
        // hot words cost less in carries than they lose in LDS gathers (832-byte rows: X^T.W 17.4 -> 15.8 ms at the
        // configs[4] shard; 448-512-byte rows: equal)
        // ... and with 128-byte rows (1218 documents per tile) twice as many pay: X^T.W -2..4 % at configs[3] and its parts
        const int hot_words = tune.xt_hot_words > 0 ? tune.xt_hot_words : (tile_rows < 256 ? 256 : tile_rows >= 1024 ? 1024 : 512);
        n_hot = (int)std::min<int64_t>(hot_words, V / 2);
        // worth it only with long pieces (enough documents per tile: fewer carries than gathers saved; measured
        // beside the blocked stream C2 +10 %, C4 +8 %, C3 with 344-document tiles +6 %) and enough tiles to
        // occupy the CUs (C2 shards of 500k / 250k / 125k documents +13 / +11 / +5 %, 60k documents -4 %)
        const int64_t min_rows = tune.tile_min_rows, min_tiles = (int64_t)tune.tile_min_tiles_pct * n_cus / 100;
        if (tile_rows < min_rows || n_hot < 8 || cdiv(m, tile_rows) < std::max<int64_t>(min_tiles, 4)) n_hot = 0;
        // sharded: the hot rows are an exchange range of their own (early exchange, run_nmf): a whole number of rows per rank
        if (ctx->n_ranks > 1) n_hot -= n_hot % ctx->n_ranks;
      }
      // rare words: synchronised sweep with LDS accumulators (one vector per lane only)
      const size_t sweep_rb = (vmap() && tune.sweep > 0 && (vpl == 1 || lpr == 16) && !tune.force_wide) ? (size_t)lds_ld_sw * sizeof(T) : 0;
      MNMF_TRY(ensure_xt(bs, n_hot, tile_rows, sweep_rb));
      if (xt_layout_only) return 0;   // agree_early_exchange: only the streams' layout is wanted
    }
    if (&M == &Xt && xt_cold_event != nullptr) {
      // Early exchange (sharded NMF): the streams of the less frequent words first -- their rows, all of `out` from row
      // Xth.n_hot on, are final when xt_cold_event fires -- then the tile stream of the most frequent words, beside
      // which the caller exchanges the rows that are done.  One after the other on the caller's stream.
      MNMF_TRY(spmm_stream(Xt, B, ld, out, lpr, vpl, ng, lds_ld, 1));
      if (Xth.nnz > 0) MNMF_TRY(spmm_stream(Xth, B, ld, out, lpr, vpl, ng, lds_ld, 1));
      if (Xt.nnz > 0) MNMF_TRY(spmm_stream(Xt, B, ld, out, lpr, vpl, ng, lds_ld, 2));
      if (Xs.nnz > 0) MNMF_TRY(sweep_dispatch(B, ld, out, lds_ld_sw, lpr, vpl));
      E_HIP(hipEventRecord(xt_cold_event, ctx->stream));
      if (Xth.nnz > 0) MNMF_TRY(spmm_stream(Xth, B, ld, out, lpr, vpl, ng, lds_ld, 2));
      return 0;
    }
    if (&M == &Xt && Xs.nnz > 0) {
      // Three streams of X^T: sweep (rare words; LDS accumulators, needs its workgroups resident together and in step,
      // so it is launched first, on the idle GPU), blocked (the words in between; no LDS, small workgroups that fill the
      // other half of every CU: on the side stream), tile (the most frequent words; most of a CU's LDS: behind the sweep).
      // They write disjoint rows of `out`; rows without nnz are zeroed before the fork.
      MNMF_TRY(spmm_stream(Xt, B, ld, out, lpr, vpl, ng, lds_ld, 1));
      if (Xth.nnz > 0) MNMF_TRY(spmm_stream(Xth, B, ld, out, lpr, vpl, ng, lds_ld, 1));
      const bool fork = tune.xt_overlap && Xt.nnz > 0;
      if (fork) {
        MNMF_TRY(ensure_side_stream());
        E_HIP(hipEventRecord(ev_fork, ctx->stream));
        E_HIP(hipStreamWaitEvent(side_stream, ev_fork, 0));
      }
      if (tune.sweep_order == 0) MNMF_TRY(sweep_dispatch(B, ld, out, lds_ld_sw, lpr, vpl));
      if ((tune.sweep_order == 1 || tune.sweep_order == 3) && Xth.nnz > 0) MNMF_TRY(spmm_stream(Xth, B, ld, out, lpr, vpl, ng, lds_ld, 2));
      if (Xt.nnz > 0) {
        if (fork) st_override = side_stream;
        const int r = spmm_stream(Xt, B, ld, out, lpr, vpl, ng, lds_ld, 2);
        st_override = nullptr;
        MNMF_TRY(r);
      }
      if (tune.sweep_order != 1 && tune.sweep_order != 3 && Xth.nnz > 0) MNMF_TRY(spmm_stream(Xth, B, ld, out, lpr, vpl, ng, lds_ld, 2));
      // order 3: the sweep follows the tile stream on the caller's stream, before the join -- its workgroups take the CUs
      // as the tile kernel's drain, beside the tail of the blocked stream and the fix-up kernels of both streams
      if (tune.sweep_order == 3) MNMF_TRY(sweep_dispatch(B, ld, out, lds_ld_sw, lpr, vpl));
      if (fork) {
        E_HIP(hipEventRecord(ev_join, side_stream));
        E_HIP(hipStreamWaitEvent(ctx->stream, ev_join, 0));
      }
      if (tune.sweep_order != 0 && tune.sweep_order != 3) MNMF_TRY(sweep_dispatch(B, ld, out, lds_ld_sw, lpr, vpl));
      return 0;
    }
    if (&M == &Xt && Xth.nnz > 0 && tune.xt_overlap) {
      // The blocked stream is bound by the vector-memory path and the fabric, the tile stream by LDS and VALU
      // issue: enqueue them on two streams so the CUs run both at once.  They write disjoint rows of `out`,
      // except that the blocked stream first zeroes the rows it has no nnz for (the tile stream's rows among
      // them): that happens before the fork.
      MNMF_TRY(ensure_side_stream());
      MNMF_TRY(spmm_stream(Xt, B, ld, out, lpr, vpl, ng, lds_ld, 1));   // segments + zero rows only
      MNMF_TRY(spmm_stream(Xth, B, ld, out, lpr, vpl, ng, lds_ld, 1));
      // The tile kernel's workgroups need most of a CU (LDS, registers): they are launched first, on the caller's
      // stream, so they are resident before the blocked stream's small workgroups fill what is left.
      E_HIP(hipEventRecord(ev_fork, ctx->stream));
      E_HIP(hipStreamWaitEvent(side_stream, ev_fork, 0));
      MNMF_TRY(spmm_stream(Xth, B, ld, out, lpr, vpl, ng, lds_ld, 2));
      st_override = side_stream;
      const int r = spmm_stream(Xt, B, ld, out, lpr, vpl, ng, lds_ld, 2);
      st_override = nullptr;
      MNMF_TRY(r);
      E_HIP(hipEventRecord(ev_join, side_stream));
      E_HIP(hipStreamWaitEvent(ctx->stream, ev_join, 0));
      return 0;
    }
    MNMF_TRY(spmm_stream(M, B, ld, out, lpr, vpl, ng, lds_ld));
    if (&M == &Xt && Xth.nnz > 0) MNMF_TRY(spmm_stream(Xth, B, ld, out, lpr, vpl, ng, lds_ld));
    return 0;
  }

  // The second stream of X^T.W.  HIP hands streams to a few hardware queues in turn (4 per process on the target
  // boxes) and two streams on one queue run one after the other: with the streams a process group creates before the
  // engine, the side stream landed on the caller's queue and the overlap was gone (X^T.W 1.34 -> 1.86 ms at configs[1],
  // kernel trace of the RCCL rehearsal).  Streams of another priority come from another set of queues, whatever else
  // the process has created.
  // A stream of its own (high priority) for the collectives whose result the next kernels on the solver's stream do not
  // need: the all-reduce of [w.T@w | dW Gram] at the end of a sharded NMF iteration is only read by the H update, one
  // X^T.W later.  The hook is handed this stream, so an RCCL collective runs beside the SpMM instead of before it.
  int ensure_comm_stream() {
    if (comm_stream) return 0;
    int lo = 0, hi = 0;   // numerically lowest = highest priority
    E_HIP(hipDeviceGetStreamPriorityRange(&lo, &hi));
    if (lo != hi) E_HIP(hipStreamCreateWithPriority(&comm_stream, hipStreamNonBlocking, lo));
    else E_HIP(hipStreamCreateWithFlags(&comm_stream, hipStreamNonBlocking));
    E_HIP(hipEventCreateWithFlags(&ev_comm_in, hipEventDisableTiming));
    E_HIP(hipEventCreateWithFlags(&ev_comm_out, hipEventDisableTiming));
    E_HIP(hipEventCreateWithFlags(&ev_cold, hipEventDisableTiming));
    E_HIP(hipEventCreateWithFlags(&ev_hot, hipEventDisableTiming));
    E_HIP(hipEventCreateWithFlags(&ev_h_done, hipEventDisableTiming));
    return 0;
  }

  int ensure_side_stream() {
    if (side_stream) return 0;
    int lo = 0, hi = 0;   // numerically lowest = highest priority
    E_HIP(hipDeviceGetStreamPriorityRange(&lo, &hi));
    const int prio = tune.side_prio > 0 ? lo : tune.side_prio < 0 ? hi : 0;
    if (tune.side_prio != 0 && lo != hi) E_HIP(hipStreamCreateWithPriority(&side_stream, hipStreamNonBlocking, prio));
    else E_HIP(hipStreamCreateWithFlags(&side_stream, hipStreamNonBlocking));
    E_HIP(hipEventCreateWithFlags(&ev_fork, hipEventDisableTiming));
    E_HIP(hipEventCreateWithFlags(&ev_join, hipEventDisableTiming));
    return 0;
  }

  // Document block of the X^T blocked stream: 4 MiB of W rows (one XCD's L2) measured best at 128- and 256-byte rows
  // (8 MiB: -3 %); rows of 448 bytes and more (32 and 64 lanes per row) gain 2-7 % on X^T.W with 8 MiB
  // (tools/sweep_block.sh: configs[2] 2.80 -> 2.66 ms, fp64 configs[1] 3.23 -> 3.11 ms, k = 200 5.37 -> 4.97 ms,
  // configs[4] shard 15.8 -> 15.1 ms); 16 MiB loses everywhere.
  int64_t xt_block_limit(int lpr) const {
    if (tune.xt_block_bytes > 0) return tune.xt_block_bytes;
    return lpr >= 32 ? (int64_t)8 << 20 : (int64_t)4 << 20;
  }

  // Smallest X (nnz) for which X.H^T takes the LDS-hot kernel rather than the plain stream kernel, by lanes per factor
  // row.  The hot kernel pays a fixed price (one 1024-thread workgroup per CU stages the table; 16 waves per CU; a select
  // between the LDS and the memory operand on every step) for serving the frequent half of the nnz from LDS.  Measured
  // on MI355X (tools/sweep_hotmin.sh, Zipf matrices, X.H^T ms hot / plain): 16 lanes (256-byte rows) 7.9e6 nnz
  // 0.167 / 0.143, 1.2e7 0.204 / 0.215, 9.8e7 1.17 / 1.39; 8 lanes (128-byte rows) 2.5e7 0.297 / 0.273, 4.9e7
  // 0.444 / 0.474; 64 lanes 1.2e7 0.595 / 0.580, 4.9e7 2.14 / 2.22; 32 lanes (two nnz per step: the select costs what
  // the LDS saves) 9.8e7 fp32 k = 70 2.43 / 1.98, k = 100 2.50 / 2.47, fp64 k = 50 2.91 / 2.62, 4.9e8 18.1 / 17.4.
  int64_t hot_x_threshold(int lpr) const {
    if (tune.hot_x_min_nnz >= 0) return tune.hot_x_min_nnz;
    switch (lpr) {
      case 16: return 10000000;
      case 32: return INT64_MAX;
      case 64: return 30000000;
      default: return 35000000;
    }
  }

  // part: 0 = everything, 1 = preparation (segments, carry buffer, zeroing of empty rows), 2 = kernels only
  int spmm_stream(StreamMat<T>& M, const T* B, int ld, T* out, int lpr, int vpl, int ng, int lds_ld, int part = 0) {
    const bool x_hot = &M == &X && vmap() && tune.hot_lds_bytes > 0 && M.nnz >= hot_x_threshold(lpr);
    const int64_t xt_q = tune.xt_q > 0 ? tune.xt_q : (nnz < 70000000 ? 64 : 128);
    int64_t q = (&M == &Xt) ? xt_q : (&M == &Xth) ? tune.xth_q : x_hot ? tune.x_q : tune.x_q_plain;
    if (&M == &Xth) {
      if (q <= 0) q = -(int64_t)(16 / tune.tile_wgs_per_cu * ng) * std::max<int64_t>(1, -q);  // equal split: (-xth_q) segments per lane group and tile
    } else {
      while (q > 32 && M.nnz / q < (int64_t)tune.seg_target * ng) q >>= 1;
      // the persistent LDS-hot kernel hands segments to n_cus x 16 waves: keep about six rounds of them, or the last
      // round's idle tail shows (measured on 1/8 and 1/4 shards of C2: X.H^T -13 % / -5 %)
      if (&M == &X && vmap() && tune.hot_lds_bytes > 0 && M.nnz >= hot_x_threshold(lpr))
        while (q > 32 && M.nnz / q < (int64_t)11 * n_cus * 16 * ng / 2) q >>= 1;
    }
    const int xcd_group = (&M == &Xt && tune.xcd_affinity && Xt.pieces.block.size() == Xt.pieces.row.size() &&
                           !Xt.pieces.row.empty()) ? 4 * ng : 0;
    if (part != 2) {
      MNMF_TRY(ensure_segments(M, q, xcd_group, ng));
      if (M.n_carry > 0) E_HIP(M.carry.ensure((size_t)M.n_carry * ld * sizeof(T)));
      if (M.n_empty > 0) {
        zero_rows_kernel<T><<<(unsigned)cdiv(M.n_empty * ld, 256), 256, 0, st()>>>(M.empty_rows.template as<int32_t>(), M.n_empty,
                                                                                  ld, out);
        E_LAUNCH();
      }
      if (part == 1) return 0;
    }
    // X direction: stage the most frequent rows of the (renumbered) factor in LDS
    int hot = 0;
    if (&M == &X && vmap() && tune.hot_lds_bytes > 0 && M.nnz >= hot_x_threshold(lpr) &&
        (int64_t)M.tab_rows * ld * (int64_t)sizeof(T) < ((int64_t)1 << 31)) {
      hot = (int)std::min<int64_t>(M.tab_rows, tune.hot_lds_bytes / std::max(1, tune.hot_wgs_per_cu) / ((int64_t)lds_ld * (int64_t)sizeof(T)));
      if (hot < 16) hot = 0;
    }
    return spmm_dispatch(M, B, ld, ld, 0, out, hot, lds_ld, vpl, lpr);
  }

  template <int LPR, int VPL = 1>
  int sweep_launch(const T* B, int ld, T* out, int lds_ld) {
    constexpr int NGW = 16 * (kWave / LPR);
    const int64_t n_cells = (int64_t)Xs.n_windows * Xs.P;
    if (Xs.ngw != NGW) {
      E_HIP(Xs.seg.ensure((size_t)(n_cells * NGW + 1) * 4));
      sweep_segs_kernel<<<(unsigned)cdiv(n_cells * NGW + 1, 256), 256, 0, st()>>>(Xs.keys.template as<uint32_t>(), Xs.cell_start.template as<int32_t>(),
                                                                                 n_cells, NGW, Xs.seg.template as<int32_t>());
      E_LAUNCH();
      Xs.ngw = NGW;
    }
    SweepArgs<T> a{Xs.seg.template as<int32_t>(), Xs.n_windows, Xs.docs.template as<uint32_t>(), Xs.lrows.template as<uint32_t>(),
                   Xs.vals.template as<T>(), B, ld, lds_ld, Xs.rpp, (uint32_t)((size_t)m * ld * sizeof(T)),
                   Xs.words.template as<int32_t>(), out};
    auto kern = spmm_sweep_kernel<T, LPR, VPL>;
    const size_t lds_bytes = (size_t)Xs.rpp * lds_ld * sizeof(T) + 64;
    E_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
    kern<<<(unsigned)Xs.P, 1024, lds_bytes, st()>>>(a);
    E_LAUNCH();
    return 0;
  }
  int sweep_dispatch(const T* B, int ld, T* out, int lds_ld, int lpr, int vpl = 1) {
    if (vpl == 2) {
      if (lpr == 16) return sweep_launch<16, 2>(B, ld, out, lds_ld);
      return fail(MNMF_ESTATE, "sweep: two vectors per lane only for rows on 16 lanes");
    }
    switch (lpr) {
      case 4: return sweep_launch<4>(B, ld, out, lds_ld);
      case 8: return sweep_launch<8>(B, ld, out, lds_ld);
      case 16: return sweep_launch<16>(B, ld, out, lds_ld);
      case 32: return sweep_launch<32>(B, ld, out, lds_ld);
      default: return sweep_launch<64>(B, ld, out, lds_ld);
    }
  }

  int spmm_dispatch(StreamMat<T>& M, const T* B, int ld, int nc, int c0, T* out, int hot, int lds_ld, int vpl, int lpr) {
    if (vpl == 2 && lpr == 16) return spmm_launch<16, 2>(M, B, ld, nc, c0, out, hot, lds_ld);
    if (vpl == 2) return spmm_launch<64, 2>(M, B, ld, nc, c0, out, hot, lds_ld);
    switch (lpr) {
      case 4: return spmm_launch<4, 1>(M, B, ld, nc, c0, out, hot, lds_ld);
      case 8: return spmm_launch<8, 1>(M, B, ld, nc, c0, out, hot, lds_ld);
      case 16: return spmm_launch<16, 1>(M, B, ld, nc, c0, out, hot, lds_ld);
      case 32: return spmm_launch<32, 1>(M, B, ld, nc, c0, out, hot, lds_ld);
      default: return spmm_launch<64, 1>(M, B, ld, nc, c0, out, hot, lds_ld);
    }
  }

  // Padding (elements) of a factor row in the LDS images of the HOT and TILE modes.  A wave-level ds_read_b128 is
  // served in four cycles of 16 lanes, and the 16 lanes of one cycle come from two different lane groups of the
  // gather (MI355X LDS: {0-3,12-15,20-27}, ...), i.e. from two different rows when a row is covered by 16 lanes:
  // the two rows' banks interleave without conflict only if both start on the same bank, so the row stride has to
  // be a whole 256 bytes.  Round 1 padded every row by 16 bytes (the usual recipe against column-access conflicts),
  // which is what made 37-39 % of the LDS cycles conflict cycles (profiles/r01_spmm_counters_tile_kernel.txt): 7.7
  // cycles per instruction instead of 4 at ld = 64 fp32.  32 lanes per row and more: any stride is conflict-free once
  // idle lanes no longer all read offset 0 (spmm.hpp, lbytes).  8 or 4 lanes per row: no stride avoids conflicts
  // (7 of 4 cycles at best), the unpadded one is the least bad.
  int lds_pad_for(int ld, int lpr) const {
    if (tune.lds_pad >= 0) return (int)round_up(tune.lds_pad, VecOf<T>::N);
    if (lpr == 16) return (int)(round_up((int64_t)ld * sizeof(T), 256) / sizeof(T)) - ld;
    return 0;
  }

  template <int NT, int TR>
  int gram_mfma_launch(const float* A, const float* A2, const float* B, const float* B2, int64_t rows, int ld, bool diff,
                       float* out) {
    const int64_t elems = (int64_t)ld * ld;
    const int64_t blocks = std::max<int64_t>(1, std::min<int64_t>(2 * n_cus, cdiv(rows, 256)));
    const int64_t rpb = round_up(cdiv(rows, blocks), 16);
    const int64_t nb = cdiv(rows, rpb);
    E_HIP(slab.ensure((size_t)nb * elems * sizeof(float)));
    dim3 grid((unsigned)nb, (unsigned)(NT / TR));
    float* sl = slab.template as<float>();
    const bool sym = (A == B && A2 == B2);
    if (sym) {
      if (diff) gram_mfma_f32_kernel<NT, TR, true, true><<<grid, 512, 0, st()>>>(A, A2, B, B2, rows, ld, rpb, sl);
      else gram_mfma_f32_kernel<NT, TR, false, true><<<grid, 512, 0, st()>>>(A, A2, B, B2, rows, ld, rpb, sl);
    } else if constexpr (NT * TR <= 9) {  // a full TR x NT block of tiles has to fit the register file
      if (diff) gram_mfma_f32_kernel<NT, TR, true, false><<<grid, 512, 0, st()>>>(A, A2, B, B2, rows, ld, rpb, sl);
      else gram_mfma_f32_kernel<NT, TR, false, false><<<grid, 512, 0, st()>>>(A, A2, B, B2, rows, ld, rpb, sl);
    } else {
      return fail(MNMF_EINVAL, "gram: tile shape reserved for symmetric products");
    }
    E_LAUNCH();
    slab_reduce_kernel<float><<<(unsigned)cdiv(elems, 16), 256, 0, st()>>>(slab.template as<float>(), nb, elems, out);
    E_LAUNCH();
    return 0;
  }

  template <int NT>
  int gram64_launch(const double* A, const double* A2, int64_t rows, bool diff, double* out) {
    const int64_t elems = (int64_t)(16 * NT) * (16 * NT);
    const int64_t blocks = std::max<int64_t>(1, std::min<int64_t>(2 * n_cus, cdiv(rows, 256)));
    const int64_t rpb = round_up(cdiv(rows, blocks), 16);
    const int64_t nb = cdiv(rows, rpb);
    E_HIP(slab.ensure((size_t)nb * elems * sizeof(double)));
    double* sl = slab.template as<double>();
    if (diff) gram_mfma_f64_kernel<NT, true><<<(unsigned)nb, 256, 0, st()>>>(A, A2, rows, rpb, sl);
    else gram_mfma_f64_kernel<NT, false><<<(unsigned)nb, 256, 0, st()>>>(A, A2, rows, rpb, sl);
    E_LAUNCH();
    slab_reduce_kernel<double><<<(unsigned)cdiv(elems, 16), 256, 0, st()>>>(sl, nb, elems, out);
    E_LAUNCH();
    return 0;
  }

  // symmetric Gram of a wide factor (5..8 column tiles): the waves of a workgroup split the tiles (gram_mfma.hpp)
  template <int NT>
  int gram_mfma_wide_launch(const float* A, const float* A2, int64_t rows, int ld, bool diff, float* out) {
    const int64_t elems = (int64_t)ld * ld;
    const int64_t blocks = std::max<int64_t>(1, std::min<int64_t>(2 * n_cus, cdiv(rows, 256)));
    const int64_t rpb = round_up(cdiv(rows, blocks), 16);
    const int64_t nb = cdiv(rows, rpb);
    E_HIP(slab.ensure((size_t)nb * elems * sizeof(float)));
    float* sl = slab.template as<float>();
    if (diff) gram_mfma_wide_f32_kernel<NT, true><<<(unsigned)nb, 512, 0, st()>>>(A, A2, rows, ld, rpb, sl);
    else gram_mfma_wide_f32_kernel<NT, false><<<(unsigned)nb, 512, 0, st()>>>(A, A2, rows, ld, rpb, sl);
    E_LAUNCH();
    slab_reduce_kernel<float><<<(unsigned)cdiv(elems, 16), 256, 0, st()>>>(sl, nb, elems, out);
    E_LAUNCH();
    return 0;
  }

  // out (ld x ld) = A^T B over rows;  diff: operands |A - A2|, |B - B2|
  int gram(const T* A, const T* A2, const T* B, const T* B2, int64_t rows, int ld, bool diff, T* out) {
    const int64_t elems = (int64_t)ld * ld;
    if (rows <= 0) {
      E_HIP(hipMemsetAsync(out, 0, elems * sizeof(T), st()));
      return 0;
    }
    if constexpr (sizeof(T) == 4) {
      if (tune.mfma_gram && ld <= 256) {  // matrix-core path (fp32); fp64 and ld > 256 use the VALU tiles below
        const float* a = (const float*)A; const float* a2 = (const float*)A2;
        const float* b = (const float*)B; const float* b2 = (const float*)B2;
        float* o = (float*)out;
        if (a == b && a2 == b2 && ld > 128 && tune.wide_gram) {
          switch ((ld + 31) / 32) {
            case 5: return gram_mfma_wide_launch<5>(a, a2, rows, ld, diff, o);
            case 6: return gram_mfma_wide_launch<6>(a, a2, rows, ld, diff, o);
            case 7: return gram_mfma_wide_launch<7>(a, a2, rows, ld, diff, o);
            default: return gram_mfma_wide_launch<8>(a, a2, rows, ld, diff, o);
          }
        }
        switch ((ld + 31) / 32) {
          case 1: return gram_mfma_launch<1, 1>(a, a2, b, b2, rows, ld, diff, o);
          case 2: return gram_mfma_launch<2, 2>(a, a2, b, b2, rows, ld, diff, o);
          case 3: return gram_mfma_launch<3, 3>(a, a2, b, b2, rows, ld, diff, o);
          case 4:
            if (a == b && a2 == b2) return gram_mfma_launch<4, 4>(a, a2, b, b2, rows, ld, diff, o);
            return gram_mfma_launch<4, 2>(a, a2, b, b2, rows, ld, diff, o);
          case 5: return gram_mfma_launch<5, 1>(a, a2, b, b2, rows, ld, diff, o);
          case 6: return gram_mfma_launch<6, 1>(a, a2, b, b2, rows, ld, diff, o);
          case 7: return gram_mfma_launch<7, 1>(a, a2, b, b2, rows, ld, diff, o);
          default: return gram_mfma_launch<8, 1>(a, a2, b, b2, rows, ld, diff, o);
        }
      }
    }
    if constexpr (sizeof(T) == 8) {
      if (tune.mfma_gram && tune.mfma_f64 && A == B && A2 == B2 && ld % 16 == 0 && ld <= 112) {  // gram_mfma_f64.hpp (8 tiles spill)
        const double* a = (const double*)A; const double* a2 = (const double*)A2;
        switch (ld / 16) {
          case 1: return gram64_launch<1>(a, a2, rows, diff, (double*)out);
          case 2: return gram64_launch<2>(a, a2, rows, diff, (double*)out);
          case 3: return gram64_launch<3>(a, a2, rows, diff, (double*)out);
          case 4: return gram64_launch<4>(a, a2, rows, diff, (double*)out);
          case 5: return gram64_launch<5>(a, a2, rows, diff, (double*)out);
          case 6: return gram64_launch<6>(a, a2, rows, diff, (double*)out);
          default: return gram64_launch<7>(a, a2, rows, diff, (double*)out);
        }
      }
    }
    const int64_t rpc = std::max<int64_t>(256, round_up(cdiv(rows, 1024), 16));
    const int64_t chunks = cdiv(rows, rpc);
    E_HIP(slab.ensure((size_t)chunks * elems * sizeof(T)));
    const int tiles = (ld + 63) / 64;
    dim3 grid((unsigned)chunks, (unsigned)(tiles * tiles));
    if (diff)
      gram_partial_kernel<T, true><<<grid, 256, 0, st()>>>(A, A2, B, B2, rows, ld, rpc, slab.template as<T>());
    else
      gram_partial_kernel<T, false><<<grid, 256, 0, st()>>>(A, A2, B, B2, rows, ld, rpc, slab.template as<T>());
    E_LAUNCH();
    slab_reduce_kernel<T><<<(unsigned)cdiv(elems, 16), 256, 0, st()>>>(slab.template as<T>(), chunks, elems, out);
    E_LAUNCH();
    return 0;
  }

  int rk(int mode, const T* A1, const T* G1, const T* A2, const T* G2, const T* upd, const T* num, T* out,
         int64_t rows, int ld, double eps) {
    if (rows <= 0) return 0;
    RkArgs<T> a{A1, G1, A2, G2, upd, num, out, rows, ld, (T)eps};
    if constexpr (sizeof(T) == 4) {
      if (tune.mfma_rk && ld <= 1024 && (ld <= 32 || ld % 16 == 0)) return rk_mfma(mode, a);   // beyond 128 columns: groups of 128 (grid.y)
    } else {
      if (can_fuse(ld)) return rk64(mode, a, nullptr, nullptr);
      if (tune.mfma_rk && tune.mfma_f64 && ld > 64 && ld <= 128 && ld % 16 == 0) return rk64_wide(mode, a);
    }
    dim3 grid((unsigned)cdiv(rows, 64), (unsigned)cdiv(ld, 64));
    if (mode == RK_PROD) dense_rk_kernel<T, RK_PROD><<<grid, 256, 0, st()>>>(a);
    else if (mode == RK_MU) dense_rk_kernel<T, RK_MU><<<grid, 256, 0, st()>>>(a);
    else dense_rk_kernel<T, RK_MU_SQRT><<<grid, 256, 0, st()>>>(a);
    E_LAUNCH();
    return 0;
  }

  template <int NT>
  int rk_mfma_nt(int mode, const RkArgs<float>& a) {
    const int64_t n_tiles = cdiv(a.rows, 32);
    const dim3 grid((unsigned)cdiv(n_tiles, 4), (unsigned)cdiv(a.ld, 32 * NT));
    if (mode == RK_PROD) dense_rk_mfma_f32_kernel<NT, RK_PROD, 0><<<grid, 256, 0, st()>>>(a, n_tiles, nullptr, nullptr);
    else if (mode == RK_MU) dense_rk_mfma_f32_kernel<NT, RK_MU, 0><<<grid, 256, 0, st()>>>(a, n_tiles, nullptr, nullptr);
    else dense_rk_mfma_f32_kernel<NT, RK_MU_SQRT, 0><<<grid, 256, 0, st()>>>(a, n_tiles, nullptr, nullptr);
    E_LAUNCH();
    return 0;
  }

  // update + the Grams of the new factor and of |new - old| in one pass (fp32, ld <= 64)
  template <int NT, int MODE, int FUSE>
  int rk_fused_launch(const RkArgs<float>& a, float* gram_new, float* gram_diff) {
    const int64_t elems = (int64_t)a.ld * a.ld;
    const int64_t n_tiles = cdiv(a.rows, 32);
    const int64_t nb = std::max<int64_t>(1, std::min<int64_t>(cdiv(n_tiles, 4), 2 * (int64_t)n_cus));
    E_HIP(slab.ensure((size_t)(2 * nb * elems) * sizeof(float)));
    float* s_new = slab.template as<float>();
    float* s_dif = s_new + nb * elems;
    dense_rk_mfma_f32_kernel<NT, MODE, FUSE><<<(unsigned)nb, 256, 0, st()>>>(a, n_tiles, s_new, s_dif);
    E_LAUNCH();
    if (FUSE == 3) {  // both Grams in one launch
      slab_reduce_kernel<float><<<dim3((unsigned)cdiv(elems, 16), 2), 256, 0, st()>>>(s_new, nb, elems, gram_new, s_dif, gram_diff);
      E_LAUNCH();
    } else if (FUSE & 1) {
      slab_reduce_kernel<float><<<(unsigned)cdiv(elems, 16), 256, 0, st()>>>(s_new, nb, elems, gram_new);
      E_LAUNCH();
    } else if (FUSE & 2) {
      slab_reduce_kernel<float><<<(unsigned)cdiv(elems, 16), 256, 0, st()>>>(s_dif, nb, elems, gram_diff);
      E_LAUNCH();
    }
    return 0;
  }
  template <int NT>
  int rk_fused_nt(int mode, const RkArgs<float>& a, float* gram_new, float* gram_diff) {
    const int fuse = (gram_new ? 1 : 0) | (gram_diff ? 2 : 0);
    if (mode == RK_MU) {
      if (fuse == 3) return rk_fused_launch<NT, RK_MU, 3>(a, gram_new, gram_diff);
      if (fuse == 1) return rk_fused_launch<NT, RK_MU, 1>(a, gram_new, gram_diff);
      return rk_fused_launch<NT, RK_MU, 2>(a, gram_new, gram_diff);
    }
    if (fuse == 3) return rk_fused_launch<NT, RK_MU_SQRT, 3>(a, gram_new, gram_diff);
    if (fuse == 1) return rk_fused_launch<NT, RK_MU_SQRT, 1>(a, gram_new, gram_diff);
    return rk_fused_launch<NT, RK_MU_SQRT, 2>(a, gram_new, gram_diff);
  }
  bool can_fuse(int ld) const {
    if (sizeof(T) == 8) return tune.mfma_rk && tune.mfma_f64 && (ld == 16 || ld == 32 || ld == 48 || ld == 64);
    return tune.mfma_rk && tune.mfma_gram && tune.fuse_grams && (ld <= 32 || ld == 48 || ld == 64);
  }

  // fp64 on the matrix cores (rk_mfma_f64.hpp): product / update, with the Grams of the new factor and of |new - old|
  // in the same pass when asked for
  template <int NT, int MODE>
  int rk64_launch(const RkArgs<double>& a, double* gram_new, double* gram_diff) {
    const int fuse = MODE == RK_PROD ? 0 : (gram_new ? 1 : 0) | (gram_diff ? 2 : 0);
    const int64_t elems = (int64_t)a.ld * a.ld;
    const int64_t n_tiles = cdiv(a.rows, 16);
    const int64_t nb = std::max<int64_t>(1, std::min<int64_t>(cdiv(n_tiles, 4), 2 * (int64_t)n_cus));
    double* s_new = nullptr;
    double* s_dif = nullptr;
    if (fuse) {
      E_HIP(slab.ensure((size_t)(2 * nb * elems) * sizeof(double)));
      s_new = slab.template as<double>();
      s_dif = s_new + nb * elems;
    }
    if constexpr (MODE == RK_PROD) {
      dense_rk_mfma_f64_kernel<NT, MODE, 0><<<(unsigned)nb, 256, 0, st()>>>(a, n_tiles, nullptr, nullptr);
    } else {
      if (fuse == 3) dense_rk_mfma_f64_kernel<NT, MODE, 3><<<(unsigned)nb, 256, 0, st()>>>(a, n_tiles, s_new, s_dif);
      else if (fuse == 1) dense_rk_mfma_f64_kernel<NT, MODE, 1><<<(unsigned)nb, 256, 0, st()>>>(a, n_tiles, s_new, s_dif);
      else if (fuse == 2) dense_rk_mfma_f64_kernel<NT, MODE, 2><<<(unsigned)nb, 256, 0, st()>>>(a, n_tiles, s_new, s_dif);
      else dense_rk_mfma_f64_kernel<NT, MODE, 0><<<(unsigned)nb, 256, 0, st()>>>(a, n_tiles, nullptr, nullptr);
    }
    E_LAUNCH();
    if (fuse == 3) {
      slab_reduce_kernel<double><<<dim3((unsigned)cdiv(elems, 16), 2), 256, 0, st()>>>(s_new, nb, elems, gram_new, s_dif, gram_diff);
      E_LAUNCH();
    } else if (fuse == 1) {
      slab_reduce_kernel<double><<<(unsigned)cdiv(elems, 16), 256, 0, st()>>>(s_new, nb, elems, gram_new);
      E_LAUNCH();
    } else if (fuse == 2) {
      slab_reduce_kernel<double><<<(unsigned)cdiv(elems, 16), 256, 0, st()>>>(s_dif, nb, elems, gram_diff);
      E_LAUNCH();
    }
    return 0;
  }
  template <int NT>
  int rk64_nt(int mode, const RkArgs<double>& a, double* gram_new, double* gram_diff) {
    if (mode == RK_PROD) return rk64_launch<NT, RK_PROD>(a, nullptr, nullptr);
    if (mode == RK_MU) return rk64_launch<NT, RK_MU>(a, gram_new, gram_diff);
    return rk64_launch<NT, RK_MU_SQRT>(a, gram_new, gram_diff);
  }
  // 64 < ld <= 128 (k = 65 ... 128 in fp64): product / update on the matrix cores without the fused Grams (their
  // accumulators would not fit the register file next to 5 ... 8 column tiles); the Grams stay separate passes
  template <int NT>
  int rk64_wide_nt(int mode, const RkArgs<double>& a) {
    const int64_t n_tiles = cdiv(a.rows, 16);
    const unsigned nb = (unsigned)std::max<int64_t>(1, std::min<int64_t>(cdiv(n_tiles, 4), 2 * (int64_t)n_cus));
    if (mode == RK_PROD) dense_rk_mfma_f64_kernel<NT, RK_PROD, 0><<<nb, 256, 0, st()>>>(a, n_tiles, nullptr, nullptr);
    else if (mode == RK_MU) dense_rk_mfma_f64_kernel<NT, RK_MU, 0><<<nb, 256, 0, st()>>>(a, n_tiles, nullptr, nullptr);
    else dense_rk_mfma_f64_kernel<NT, RK_MU_SQRT, 0><<<nb, 256, 0, st()>>>(a, n_tiles, nullptr, nullptr);
    E_LAUNCH();
    return 0;
  }
  int rk64_wide(int mode, const RkArgs<double>& a) {
    switch (a.ld / 16) {
      case 5: return rk64_wide_nt<5>(mode, a);
      case 6: return rk64_wide_nt<6>(mode, a);
      case 7: return rk64_wide_nt<7>(mode, a);
      default: return rk64_wide_nt<8>(mode, a);
    }
  }
  int rk64(int mode, const RkArgs<double>& a, double* gram_new, double* gram_diff) {
    switch (a.ld / 16) {
      case 1: return rk64_nt<1>(mode, a, gram_new, gram_diff);
      case 2: return rk64_nt<2>(mode, a, gram_new, gram_diff);
      case 3: return rk64_nt<3>(mode, a, gram_new, gram_diff);
      default: return rk64_nt<4>(mode, a, gram_new, gram_diff);
    }
  }

  // rk() followed by gram(out) -> gram_new and gram(|out - upd|) -> gram_diff (either may be null); one pass
  // over the rows where the fused kernel applies, separate passes otherwise.  The results are the same either way
  // up to summation order.
  int rk_grams(int mode, const T* A1, const T* G1, const T* A2, const T* G2, const T* upd, const T* num, T* out,
               int64_t rows, int ld, double eps, T* gram_new, T* gram_diff) {
    if (!gram_new && !gram_diff) return rk(mode, A1, G1, A2, G2, upd, num, out, rows, ld, eps);
    if constexpr (sizeof(T) == 4) {
      if (can_fuse(ld) && rows > 0 && mode != RK_PROD) {
        RkArgs<float> a{A1, G1, A2, G2, upd, num, out, rows, ld, (float)eps};
        return ld <= 32 ? rk_fused_nt<1>(mode, a, gram_new, gram_diff) : rk_fused_nt<2>(mode, a, gram_new, gram_diff);
      }
    } else {
      if (can_fuse(ld) && rows > 0 && mode != RK_PROD) {
        RkArgs<double> a{A1, G1, A2, G2, upd, num, out, rows, ld, (double)eps};
        return rk64(mode, a, gram_new, gram_diff);
      }
    }
    MNMF_TRY(rk(mode, A1, G1, A2, G2, upd, num, out, rows, ld, eps));
    if (gram_new) MNMF_TRY(gram(out, nullptr, out, nullptr, rows, ld, false, gram_new));
    if (gram_diff) MNMF_TRY(gram(out, upd, out, upd, rows, ld, true, gram_diff));
    return 0;
  }

  int rk_mfma(int mode, const RkArgs<float>& a) {
    switch ((a.ld + 31) / 32) {
      case 1: return rk_mfma_nt<1>(mode, a);
      case 2: return rk_mfma_nt<2>(mode, a);
      case 3: return rk_mfma_nt<3>(mode, a);
      case 4: return rk_mfma_nt<4>(mode, a);
    }
    // ld > 128.  One pass: a wave keeps all 5..8 column tiles of its 32 rows (80..128 accumulator registers), so the
    // row tile is read once and every operand pair feeds 5..8 MFMAs; or (MNMF_RK_ONE_PASS=0) column groups of 128
    // (grid.y), which read the row tile once per group.
    if (tune.rk_one_pass) {
      switch ((a.ld + 31) / 32) {
        case 5: return rk_mfma_nt<5>(mode, a);
        case 6: return rk_mfma_nt<6>(mode, a);
        case 7: return rk_mfma_nt<7>(mode, a);
        default: return rk_mfma_nt<8>(mode, a);
      }
    }
    return rk_mfma_nt<4>(mode, a);
  }

  int kxk(const T* A, const T* B, T* C, int ld, int ta, int tb) {
    kxk_gemm_kernel<T><<<(unsigned)cdiv((int64_t)ld * ld, 256), 256, 0, st()>>>(A, B, C, ld, ta, tb);
    E_LAUNCH();
    return 0;
  }

  int allreduce(T* p, int64_t count) {
    if (ctx->comm) return collective(MNMF_COMM_ALLREDUCE, p, count);
    if (!ctx->ar) return 0;
    const int r = ctx->ar(ctx->ar_user, p, count, ctx->dtype, (void*)st());
    if (r != 0) return fail(MNMF_ECOMM, "all-reduce hook failed with code " + std::to_string(r));
    return 0;
  }
  int collective(int op, T* p, int64_t count) {
    const int r = ctx->comm(ctx->comm_user, op, p, count, ctx->dtype, (void*)st());
    if (r != 0) return fail(MNMF_ECOMM, "collective hook (op " + std::to_string(op) + ") failed with code " + std::to_string(r));
    return 0;
  }

  // ---- reduce-scatter -> slice update -> all-gather
  // The vocabulary rows of the exchanged V x ld matrices are cut into n_ranks equal slices of slice_rows() rows (the
  // buffers are allocated with the padded row count and the padding rows stay zero: the SpMM never writes them and a
  // multiplicative update of a zero row is zero).  Rank r owns rows [r * slice_rows, (r + 1) * slice_rows).
  bool use_slices() const { return ctx->comm != nullptr && tune.rs_ag && (ctx->n_ranks > 1 || tune.rs_ag >= 2); }
  int64_t slice_rows() const { return use_slices() ? cdiv(V, ctx->n_ranks) : V; }
  int64_t padded_V() const { return use_slices() ? slice_rows() * ctx->n_ranks : V; }
  // live columns on the wire only when at least a quarter of a padded row is padding (as allreduce_rows)
  bool pack_rows(int k, int ld) const { return tune.pack_exchange && (ld - k) * 4 >= ld; }
  // buf (padded_V x ld): every rank holds a partial; afterwards this rank's slice holds the sum over ranks
  int reduce_scatter_rows(T* buf, int k, int ld) { return reduce_scatter_range(buf, slice_rows(), k, ld); }
  // the same for a row range of n_ranks x Vs rows that starts at buf (rank r receives rows [r * Vs, (r + 1) * Vs) of it)
  int reduce_scatter_range(T* buf, int64_t Vs, int k, int ld) {
    const int64_t Vp = Vs * ctx->n_ranks;
    if (Vs <= 0) return 0;
    if (!pack_rows(k, ld)) return collective(MNMF_COMM_REDUCE_SCATTER, buf, Vs * ld);
    E_HIP(ar_pack.ensure((size_t)Vp * k * sizeof(T)));
    T* q = ar_pack.template as<T>();
    unpad_rows_kernel<T><<<(unsigned)cdiv(Vp * k, 256), 256, 0, st()>>>(buf, Vp, k, ld, T(0), q);
    E_LAUNCH();
    MNMF_TRY(collective(MNMF_COMM_REDUCE_SCATTER, q, Vs * k));
    const int64_t lo = Vs * ctx->rank;
    pad_rows_kernel<T><<<(unsigned)cdiv(Vs * ld, 256), 256, 0, st()>>>(q + lo * k, Vs, k, ld, buf + lo * ld);
    E_LAUNCH();
    return 0;
  }
  // buf (padded_V x ld): this rank's slice is current; afterwards every slice holds its owner's rows
  int all_gather_rows(T* buf, int k, int ld) { return all_gather_range(buf, slice_rows(), k, ld); }
  int all_gather_range(T* buf, int64_t Vs, int k, int ld) {
    const int64_t Vp = Vs * ctx->n_ranks;
    if (Vs <= 0) return 0;
    if (!pack_rows(k, ld)) return collective(MNMF_COMM_ALLGATHER, buf, Vs * ld);
    E_HIP(ar_pack.ensure((size_t)Vp * k * sizeof(T)));
    T* q = ar_pack.template as<T>();
    const int64_t lo = Vs * ctx->rank;
    unpad_rows_kernel<T><<<(unsigned)cdiv(Vs * k, 256), 256, 0, st()>>>(buf + lo * ld, Vs, k, ld, T(0), q + lo * k);
    E_LAUNCH();
    MNMF_TRY(collective(MNMF_COMM_ALLGATHER, q, Vs * k));
    pad_rows_kernel<T><<<(unsigned)cdiv(Vp * ld, 256), 256, 0, st()>>>(q, Vp, k, ld, buf);
    E_LAUNCH();
    return 0;
  }

  // Exchange of a buffer made of rows of ld elements whose columns k..ld-1 are zero on every rank (factor-shaped
  // partial products and the Gram blocks behind them): only the k live columns go on the wire.  With rows padded to
  // whole cache lines the padding is 69 % of the message at k = 10 and 37 % at k = 20; the exchange is the one part of
  // a sharded iteration that nothing can hide.  Below 25 % padding (k = 50: 22 %) the two copy kernels cost about what
  // the shorter message saves, and the buffer goes out as it is.
  int allreduce_rows(T* p, int64_t count, int k, int ld) {
    if (!ctx->sharded()) return 0;
    if (!tune.pack_exchange || count % ld != 0 || (ld - k) * 4 < ld) return allreduce(p, count);  // < 25 % padding: as is
    const int64_t rows = count / ld;
    E_HIP(ar_pack.ensure((size_t)rows * k * sizeof(T)));
    T* q = ar_pack.template as<T>();
    unpad_rows_kernel<T><<<(unsigned)cdiv(rows * k, 256), 256, 0, st()>>>(p, rows, k, ld, T(0), q);
    E_LAUNCH();
    MNMF_TRY(allreduce(q, rows * k));
    pad_rows_kernel<T><<<(unsigned)cdiv(rows * ld, 256), 256, 0, st()>>>(q, rows, k, ld, p);
    E_LAUNCH();
    return 0;
  }

  // lambda_max / trace of device Gram(s) -> host.  grams: n matrices of ld x ld, contiguous.
  int fetch_grams(const T* d, int n, int ld, std::vector<double>& out) {
    const size_t cnt = (size_t)n * ld * ld;
    MNMF_TRY(pinned(cnt * (sizeof(T) == 8 ? 1 : 1)));
    E_HIP(hipMemcpyAsync(h_pinned, d, cnt * sizeof(T), hipMemcpyDeviceToHost, st()));
    E_HIP(hipStreamSynchronize(st()));
    out.resize(cnt);
    const T* hp = reinterpret_cast<const T*>(h_pinned);
    for (size_t i = 0; i < cnt; ++i) out[i] = (double)hp[i];
    return 0;
  }
  // Split-phase variant: the copies are enqueued behind the kernels that produce the data, the host reads them
  // later (after it has enqueued more work), so the GPU never idles while the host looks at the norms.
  size_t fetch_count = 0;
  hipEvent_t ev_fetch = nullptr;
  int fetch_begin(size_t max_elems) {
    MNMF_TRY(pinned(max_elems));
    fetch_count = 0;
    if (!ev_fetch) E_HIP(hipEventCreateWithFlags(&ev_fetch, hipEventDisableTiming));
    return 0;
  }
  int fetch_add(const T* d, size_t cnt) {
    E_HIP(hipMemcpyAsync(reinterpret_cast<T*>(h_pinned) + fetch_count, d, cnt * sizeof(T), hipMemcpyDeviceToHost, st()));
    fetch_count += cnt;
    return 0;
  }
  int fetch_commit() {
    E_HIP(hipEventRecord(ev_fetch, st()));
    return 0;
  }
  int fetch_end(std::vector<double>& out) {
    E_HIP(hipEventSynchronize(ev_fetch));
    out.resize(fetch_count);
    const T* hp = reinterpret_cast<const T*>(h_pinned);
    for (size_t i = 0; i < fetch_count; ++i) out[i] = (double)hp[i];
    return 0;
  }
  // A Gram with a non-finite entry (a factor went NaN / inf) must not look like a small norm: std::max(0.0, NaN) is
  // 0.0 and the eigen-solver skips NaN diagonals, so test the entries before anything clamps.
  static bool all_finite(const double* g, size_t n) {
    for (size_t i = 0; i < n; ++i)
      if (!std::isfinite(g[i])) return false;
    return true;
  }
  static double spectral_from_gram(const double* g, int ld) {
    if (!all_finite(g, (size_t)ld * ld)) return NAN;
    return std::sqrt(std::max(0.0, nonneg_lambda_max(g, ld)));
  }
  static double fro_from_gram(const double* g, int ld) {
    double t = 0.0;
    for (int i = 0; i < ld; ++i) t += g[(size_t)i * ld + i];
    if (!std::isfinite(t)) return NAN;
    return std::sqrt(std::max(0.0, t));
  }

  int load_W(const void* W, int k, int ld, T* dst) {
    pad_rows_kernel<T><<<(unsigned)cdiv(std::max<int64_t>(m * ld, 1), 256), 256, 0, st()>>>((const T*)W, m, k, ld, dst);
    E_LAUNCH();
    return 0;
  }
  int load_H(const void* H, int k, int ld, T* dst) {
    dim3 grid((unsigned)cdiv(V, 32), (unsigned)cdiv(ld, 32));
    MNMF_TRY(ensure_prepared());
    transpose_in_kernel<T><<<grid, 256, 0, st()>>>((const T*)H, k, V, ld, dst, vmap());
    E_LAUNCH();
    return 0;
  }
  int store_W(const T* src, int k, int ld, double thr, void* W) {
    if (m * k == 0) return 0;
    unpad_rows_kernel<T><<<(unsigned)cdiv(m * k, 256), 256, 0, st()>>>(src, m, k, ld, (T)thr, (T*)W);
    E_LAUNCH();
    return 0;
  }
  int store_H(const T* src, int k, int ld, double thr, void* H) {
    dim3 grid((unsigned)cdiv(V, 32), (unsigned)cdiv(ld, 32));
    transpose_out_kernel<T><<<grid, 256, 0, st()>>>(src, k, V, ld, (T)thr, (T*)H, vmap());
    E_LAUNCH();
    return 0;
  }

  int check_ready(int k) {
    if (!have_matrix) return fail(MNMF_ESTATE, "no matrix: call mnmf_set_matrix first");
    if (k <= 0) return fail(MNMF_EINVAL, "k must be positive");
    E_HIP(hipSetDevice(ctx->device));
    return 0;
  }

  struct LoopClock {
    hipEvent_t e0 = nullptr, e1 = nullptr;
    std::chrono::steady_clock::time_point t0;
    ~LoopClock() {  // error paths leave the loop without clock_stop
      if (e0) (void)hipEventDestroy(e0);
      if (e1) (void)hipEventDestroy(e1);
    }
  };
  int clock_start(LoopClock& c) {
    E_HIP(hipEventCreate(&c.e0));
    E_HIP(hipEventCreate(&c.e1));
    E_HIP(hipStreamSynchronize(st()));
    c.t0 = std::chrono::steady_clock::now();
    E_HIP(hipEventRecord(c.e0, st()));
    return 0;
  }
  int clock_stop(LoopClock& c, mnmf_stats* s) {
    E_HIP(hipEventRecord(c.e1, st()));
    E_HIP(hipStreamSynchronize(st()));
    const auto t1 = std::chrono::steady_clock::now();
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, c.e0, c.e1);
    if (s) {
      s->ms_total = std::chrono::duration<double, std::milli>(t1 - c.t0).count();
      s->ms_kernels = ms;
    }
    (void)hipEventDestroy(c.e0);
    (void)hipEventDestroy(c.e1);
    c.e0 = c.e1 = nullptr;
    return 0;
  }

  void model(mnmf_stats* s, int k, double xw, double ww, double hw, double f_nnz, double f_m, double f_v) {
    if (!s) return;
    const double sv = sizeof(T), p = std::max(1, ctx->n_ranks);
    const double Xb = (double)nnz * (sv + 4) + (double)(m + 1) * 8, Wbt = (double)m * k * sv, Hbt = (double)k * V * sv;
    (void)p;
    s->bytes_model = xw * Xb + ww * Wbt + hw * Hbt;
    s->flops_model = f_nnz * (double)nnz * k + f_m * (double)m * k * k + f_v * (double)V * k * k;
  }

  // ------------------------------------------------------------------ loop A
  // Early exchange: every rank prepares its X^T streams, then the ranks sum (have a tile stream ? 1 : 0, its word
  // count): the schedule is taken only if all of them have one with the same number of words (the word count follows
  // from the shapes, the decision to use the tile stream from the local document count).
  int agree_early_exchange(int ld, int64_t& eh) {
    eh = 0;
    constexpr int VEC = VecOf<T>::N;
    if (ld % VEC != 0 || ld / VEC > 128) return 0;
    xt_layout_only = true;
    const int r = spmm(Xt, nullptr, ld, nullptr);
    xt_layout_only = false;
    MNMF_TRY(r);
    const int64_t mine = (Xth.nnz > 0 && Xth.n_hot > 0 && Xth.n_hot % ctx->n_ranks == 0 && Xth.n_hot < V) ? Xth.n_hot : 0;
    E_HIP(small.ensure(4 * sizeof(T)));
    MNMF_TRY(pinned(4));
    T h[2] = {(T)(mine > 0 ? 1 : 0), (T)mine};
    E_HIP(hipMemcpyAsync(small.p, h, sizeof(h), hipMemcpyHostToDevice, st()));
    E_HIP(hipStreamSynchronize(st()));   // h is on the stack
    MNMF_TRY(allreduce(small.template as<T>(), 2));
    E_HIP(hipMemcpyAsync(h, small.p, sizeof(h), hipMemcpyDeviceToHost, st()));
    E_HIP(hipStreamSynchronize(st()));
    if (mine > 0 && (int64_t)std::llround((double)h[0]) == ctx->n_ranks && (int64_t)std::llround((double)h[1]) == mine * ctx->n_ranks) {
      MNMF_TRY(ensure_comm_stream());
      eh = mine;
    }
    return 0;
  }

  int run_nmf(int k, void* W, void* H, const mnmf_opts* o, mnmf_stats* stats) override {
    ColsGuard cols_guard(data_cols, k, line_layout, k > 0 && (size_t)ld_of(k) * sizeof(T) == 128);
    MNMF_TRY(check_ready(k));
    const int ld = ld_of(k);
    const int64_t max_iter = (o && o->max_iter >= 0) ? o->max_iter : 10000;
    const int fixed = o ? o->fixed_iters : 0;
    const int every = (o && o->check_every > 0) ? o->check_every : 1;
    const bool force_check = o && o->force_check;
    const double thr = (o && o->norm_thresh >= 0) ? o->norm_thresh : 0.005;
    const double eps = (o && o->eps >= 0) ? o->eps : 1e-10;
    const double zthr = (o && o->zero_threshold >= 0) ? o->zero_threshold : 1e-5;
    const bool finalize = !o || o->finalize;
    const bool sliced = use_slices();  // sharded: reduce-scatter -> update of this rank's slice of H^T -> all-gather
    const int64_t Vp = padded_V(), Vs = slice_rows();
    const int64_t s_lo = sliced ? Vs * ctx->rank : 0;
    const int64_t s_rows = sliced ? std::max<int64_t>(0, std::min<int64_t>(Vs, V - s_lo)) : V;
    const size_t wsz = (size_t)std::max<int64_t>(m, 1) * ld * sizeof(T), hsz = (size_t)Vp * ld * sizeof(T);
    const size_t gsz = (size_t)ld * ld * sizeof(T);
    for (int i = 0; i < 2; ++i) {
      E_HIP(Wb[i].ensure(wsz));
      E_HIP(Hb[i].ensure(hsz));
    }
    E_HIP(Nw.ensure(wsz));
    E_HIP(Ex.ensure(hsz + gsz));       // [Nh (V x ld) | WtW (ld x ld)]: one all-reduce
    E_HIP(small.ensure(4 * gsz));      // HHt, dW gram, dH gram
    T* Nh = Ex.template as<T>();
    // not sliced: w.T@w sits behind the k x V partial and is summed with it in one all-reduce.  Sliced: it sits in front
    // of the dW Gram, which the same w update produces -- [w.T@w | dW gram] is one small all-reduce per iteration.
    T* Gw = sliced ? small.template as<T>() : Nh + (size_t)Vp * ld;
    T* Gh = small.template as<T>() + (sliced ? 3 : 0) * (size_t)ld * ld;
    T* Gd = small.template as<T>() + (size_t)ld * ld;      // [dW gram | dH gram] contiguous
    if (Vp > V) {  // the padding rows of the sliced buffers are zero and stay zero
      const size_t off = (size_t)V * ld * sizeof(T), len = (size_t)(Vp - V) * ld * sizeof(T);
      E_HIP(hipMemsetAsync(Hb[0].template as<char>() + off, 0, len, st()));
      E_HIP(hipMemsetAsync(Hb[1].template as<char>() + off, 0, len, st()));
      E_HIP(hipMemsetAsync(Ex.template as<char>() + off, 0, len, st()));
    }
    MNMF_TRY(load_W(W, k, ld, Wb[0].template as<T>()));
    MNMF_TRY(load_H(H, k, ld, Hb[0].template as<T>()));
    // Early exchange: the rows of X^T.W that belong to the less frequent words -- all but the
    // Xth.n_hot rows of the tile stream, i.e. nearly all of the k x V bytes -- are final before the tile stream starts;
    // their reduce-scatter, this rank's share of the h update and the all-gather run on the communication stream beside
    // the tile kernel.  Every rank must take the same schedule with the same row ranges: agreed here.
    int64_t eh = 0;   // rows of the hot range (a multiple of the rank count); 0: the plain schedule
    if (sliced && tune.early_exchange) {
      MNMF_TRY(agree_early_exchange(ld, eh));
      if (eh == 0 && tune.early_check) return fail(MNMF_ESTATE, "nmf: the early exchange was expected (MNMF_EARLY_CHECK) but the ranks do not all have a tile stream");
    }
    const int64_t Vs_h = eh / ctx->n_ranks, Vs_c = eh > 0 ? (Vp - eh) / ctx->n_ranks : 0;
    int cw = 0, ch = 0;
    double w_norm = INFINITY, h_norm = INFINITY;
    int64_t i = 0;
    int converged = 0;
    bool gw_valid = false;  // Gw holds this rank's w.T@w of the current w (left there by the previous w update)
    bool gw_in_flight = false;  // ... and its all-reduce is still running on the communication stream
    LoopClock clk;
    MNMF_TRY(clock_start(clk));
    PhaseTimer pt(st(), o && o->profile_phases);
    pt.mark(-1, 0);
    std::vector<double> g;
    // The norms of iteration i are read while iteration i+1's SpMMs are already queued (its H update only writes
    // the buffer of h_{i-1}); if iteration i turns out to have converged, that queued work is simply dropped.
    bool pending = false;
    int64_t done = 0;  // completed iterations
    auto resolve = [&]() -> int {
      MNMF_TRY(fetch_end(g));
      pending = false;
      w_norm = spectral_from_gram(g.data(), ld);
      h_norm = spectral_from_gram(g.data() + (size_t)ld * ld, ld);
      if (!std::isfinite(w_norm) || !std::isfinite(h_norm))
        return fail(MNMF_ENAN, "nmf: non-finite update norm at iteration " + std::to_string(done));
      return 0;
    };
    MNMF_TRY(pinned(2 * (size_t)ld * ld));
    while (true) {
      T* Wc = Wb[cw].template as<T>();
      T* Wn = Wb[1 - cw].template as<T>();
      T* Hc = Hb[ch].template as<T>();
      T* Hn = Hb[1 - ch].template as<T>();
      const bool do_check = (fixed > 0) ? (force_check && ((i + 1) % every == 0)) : ((i + 1) % every == 0);
      if (!gw_valid) {
        MNMF_TRY(gram(Wc, nullptr, Wc, nullptr, m, ld, false, Gw));  // wT_w       nmf_basic.py:39
        if (sliced) MNMF_TRY(allreduce(Gw, (int64_t)ld * ld));
      }
      pt.mark(0, 1);
      if (eh > 0) {
        xt_cold_event = ev_cold;
        const int r = spmm(Xt, Wc, ld, Nh);                             // (w.T @ in_mat)^T    :40, cold rows first
        xt_cold_event = nullptr;
        MNMF_TRY(r);
        E_HIP(hipEventRecord(ev_hot, ctx->stream));
      } else {
        MNMF_TRY(spmm(Xt, Wc, ld, Nh));                                 // (w.T @ in_mat)^T    :40
      }
      pt.mark(1, 2);
      if (gw_in_flight) {  // w.T@w of the previous w update: summed over the ranks beside the SpMM above
        E_HIP(hipStreamWaitEvent(ctx->stream, ev_comm_out, 0));
        gw_in_flight = false;
      }
      if (eh > 0) {
        // two row ranges, each cut into one slice per rank: [eh, Vp) as soon as the cold streams are done (the tile kernel
        // is still running on the solver's stream), then [0, eh).  Everything here is enqueued on the communication stream.
        E_HIP(hipStreamWaitEvent(comm_stream, ev_cold, 0));
        st_override = comm_stream;
        int r = 0;
        for (int range = 0; range < 2 && r == 0; ++range) {
          const int64_t base = range == 0 ? eh : 0, Vr = range == 0 ? Vs_c : Vs_h;
          if (range == 1) {
            const hipError_t e = hipStreamWaitEvent(comm_stream, ev_hot, 0);
            if (e != hipSuccess) { r = fail(MNMF_EHIP, "early exchange: hipStreamWaitEvent failed"); break; }
          }
          const int64_t lo = base + Vr * ctx->rank;                                   // this rank's rows of the range
          const int64_t rows = std::max<int64_t>(0, std::min<int64_t>(Vr, V - lo));
          r = reduce_scatter_range(Nh + (size_t)base * ld, Vr, k, ld);
          if (r == 0 && rows > 0)
            r = rk(RK_MU, Hc + (size_t)lo * ld, Gw, nullptr, nullptr, Hc + (size_t)lo * ld, Nh + (size_t)lo * ld, Hn + (size_t)lo * ld, rows, ld, eps);   // :41-42
          if (r == 0) r = all_gather_range(Hn + (size_t)base * ld, Vr, k, ld);
        }
        if (r == 0 && hipEventRecord(ev_h_done, comm_stream) != hipSuccess) r = fail(MNMF_EHIP, "early exchange: hipEventRecord failed");
        st_override = nullptr;
        MNMF_TRY(r);
        E_HIP(hipStreamWaitEvent(ctx->stream, ev_h_done, 0));
        MNMF_TRY(gram(Hn, nullptr, Hn, nullptr, V, ld, false, Gh));                                     // :44
        if (do_check) MNMF_TRY(gram(Hn, Hc, Hn, Hc, V, ld, true, Gd + (size_t)ld * ld));                // :53
      } else if (sliced) {
        // every rank sums and updates one slice of the vocabulary, then the slices are gathered: the same bytes on
        // the wire as an all-reduce, the h update done once instead of n_ranks times, h identical on every rank
        MNMF_TRY(reduce_scatter_rows(Nh, k, ld));
        const size_t o = (size_t)s_lo * ld;
        MNMF_TRY(rk(RK_MU, Hc + o, Gw, nullptr, nullptr, Hc + o, Nh + o, Hn + o, s_rows, ld, eps));    // :41-42
        MNMF_TRY(all_gather_rows(Hn, k, ld));
        MNMF_TRY(gram(Hn, nullptr, Hn, nullptr, V, ld, false, Gh));                                     // :44
        if (do_check) MNMF_TRY(gram(Hn, Hc, Hn, Hc, V, ld, true, Gd + (size_t)ld * ld));                // :53
      } else {
        MNMF_TRY(allreduce_rows(Nh, (int64_t)V * ld + (int64_t)ld * ld, k, ld));
        // h update :41-42, with h_hT of the new h (:44) and, for the norms, D^T D of D = |h - h_old| (:53) in the same pass
        MNMF_TRY(rk_grams(RK_MU, Hc, Gw, nullptr, nullptr, Hc, Nh, Hn, V, ld, eps, Gh, do_check ? Gd + (size_t)ld * ld : nullptr));
      }
      pt.mark(2, 4);
      MNMF_TRY(spmm(X, Hn, ld, Nw.template as<T>()));                            // in_mat @ h.T         :45
      pt.mark(4, pending ? 6 : 5);
      if (pending) {  // norms of the previous iteration                 :52-53, :62
        const int r = resolve();
        pt.mark(6, 5);
        if (r != 0) {
          (void)clock_stop(clk, stats);
          return r;
        }
        if (fixed <= 0 && w_norm < thr && h_norm < thr) {
          converged = 1;
          break;
        }
      }
      // w update :46-47, with the next iteration's wT_w (:39) and D^T D of D = |w - w_old| (:52)
      MNMF_TRY(rk_grams(RK_MU, Wc, Gh, nullptr, nullptr, Wc, Nw.template as<T>(), Wn, m, ld, eps, Gw, do_check ? Gd : nullptr));
      gw_valid = true;
      pt.mark(5, do_check ? 6 : 0);
      // sliced: the next iteration's w.T@w and this iteration's dW Gram in one collective (the dH Gram is computed by
      // every rank from the gathered h); otherwise w.T@w travels with the next k x V partial
      const bool async_gram = sliced && tune.async_gram;
      if (async_gram) {  // the collective and the copy of the norms' Grams leave the solver's stream: the next X^T.W does not wait
        MNMF_TRY(ensure_comm_stream());
        E_HIP(hipEventRecord(ev_comm_in, ctx->stream));
        E_HIP(hipStreamWaitEvent(comm_stream, ev_comm_in, 0));
        st_override = comm_stream;
      }
      int rc_comm = 0;
      if (sliced) rc_comm = allreduce(Gw, (int64_t)ld * ld * (do_check ? 2 : 1));
      if (rc_comm == 0 && do_check) {
        if (!sliced) rc_comm = allreduce(Gd, (int64_t)ld * ld);
        if (rc_comm == 0) rc_comm = fetch_begin(2 * (size_t)ld * ld);
        if (rc_comm == 0) rc_comm = fetch_add(Gd, 2 * (size_t)ld * ld);
        if (rc_comm == 0) rc_comm = fetch_commit();
        pending = rc_comm == 0;
      }
      if (async_gram) {
        st_override = nullptr;
        if (rc_comm == 0) {
          E_HIP(hipEventRecord(ev_comm_out, comm_stream));
          gw_in_flight = true;
        }
      }
      MNMF_TRY(rc_comm);
      if (do_check) pt.mark(6, 0);
      cw = 1 - cw;
      ch = 1 - ch;
      ++done;
      if (fixed > 0) {
        if (i + 1 >= fixed) break;
      } else {
        if (i >= max_iter) break;                                      // :57
      }
      ++i;
    }
    if (pending) {  // the last iteration's norms (the loop ended on its iteration cap)
      const int r = resolve();
      if (r != 0) {
        (void)clock_stop(clk, stats);
        return r;
      }
      if (fixed <= 0 && done <= max_iter && w_norm < thr && h_norm < thr) converged = 1;   // :57 comes before :62
    }
    if (comm_stream) E_HIP(hipStreamWaitEvent(ctx->stream, ev_comm_out, 0));   // nothing of this run is left on the communication stream
    MNMF_TRY(clock_stop(clk, stats));
    pt.collect(phase_ms);
    if (stats) std::memcpy(stats->phase_ms, phase_ms, sizeof(phase_ms));
    MNMF_TRY(store_W(Wb[cw].template as<T>(), k, ld, finalize ? zthr : -1.0, W));  // :86
    MNMF_TRY(store_H(Hb[ch].template as<T>(), k, ld, finalize ? zthr : -1.0, H));  // :87
    E_HIP(hipStreamSynchronize(st()));
    if (stats) {
      stats->iters = (int32_t)done;
      stats->converged = converged;
      stats->w_norm = w_norm;
      stats->h_norm = h_norm;
      stats->s_norm = 0;
      model(stats, k, 2, 3, 6, 4, 4, 4);
    }
    return 0;
  }

  // ------------------------------------------------------------------ loop A, several topic counts in one pass over X
  // The topic-count sweep of the reference (optimization_pipeline.py:105-136, 193-198) calls run_nmf once per candidate k
  // on the same matrix.  For small k the engine pads every factor row to one 128-byte line (ld = 32 floats), and most
  // of a gathered line is padding (69 % at k = 10); here n problems with sum(k) <= 32 share that line: their factors sit
  // side by side in the padded rows, so BOTH SpMMs of an iteration serve all problems in one pass over X (the SpMM works
  // column by column, a column's sum does not depend on its neighbours).  The dense side runs per problem, on its own
  // columns in place (RkArgs::coff / kcols) with its own k x k Grams, through the same fused update kernel as a single
  // run: every number a problem sees is produced by the same operations in the same order as in its own run_nmf, so
  // each result is bit-identical to the single-k run (tests/test_gpu_sweep.py).  Each problem stops by the reference's
  // rule on its own norms (nmf_basic.py:52-62): its factors are copied out at that iteration while the others go on.
  // fp64 alike with 16 doubles per line (sum(k) <= 16; dense_rk_mfma_f64_kernel takes the same column window).  Not for
  // sharded contexts (the exchange would have to be per problem); the Python side falls back to one run_nmf per k.
  int run_nmf_sweep(int n, const int32_t* ks, void* const* Ws, void* const* Hs, const mnmf_opts* o, mnmf_stats* stats) override {
    {
      if (n < 1 || n > 32 || !ks || !Ws || !Hs) return fail(MNMF_EINVAL, "run_nmf_sweep: bad arguments");
      int sum = 0;
      std::vector<int> off(n);
      for (int p = 0; p < n; ++p) {
        if (ks[p] <= 0 || !Ws[p] || !Hs[p]) return fail(MNMF_EINVAL, "run_nmf_sweep: bad problem " + std::to_string(p));
        off[p] = sum;
        sum += ks[p];
      }
      MNMF_TRY(check_ready(sum));
      ColsGuard cols_guard(data_cols, sum, line_layout, true);
      const int ld = ld_of(sum);
      constexpr int kLine = (int)(128 / sizeof(T));   // factor columns in one 128-byte row: 32 floats / 16 doubles
      if (sum > kLine || ld != kLine || !can_fuse(ld))
        return fail(MNMF_EINVAL, "run_nmf_sweep: the topic counts of one batch must add up to at most " + std::to_string(kLine) +
                                     " (one 128-byte factor row)");
      // the fused update of one problem: the same kernel a single run_nmf of that topic count takes
      auto update = [&](const RkArgs<T>& a, T* gram_new, T* gram_diff) -> int {
        if constexpr (sizeof(T) == 4) return rk_fused_nt<1>(RK_MU, a, gram_new, gram_diff);
        else return rk64_nt<1>(RK_MU, a, gram_new, gram_diff);
      };
      if (ctx->sharded()) return fail(MNMF_EINVAL, "run_nmf_sweep: not available on a sharded context");
      const int64_t max_iter = (o && o->max_iter >= 0) ? o->max_iter : 10000;
      const int fixed = o ? o->fixed_iters : 0;
      const int every = (o && o->check_every > 0) ? o->check_every : 1;
      const bool force_check = o && o->force_check;
      const double thr = (o && o->norm_thresh >= 0) ? o->norm_thresh : 0.005;
      const double eps = (o && o->eps >= 0) ? o->eps : 1e-10;
      const double zthr = (o && o->zero_threshold >= 0) ? o->zero_threshold : 1e-5;
      const bool finalize = !o || o->finalize;
      const size_t wsz = (size_t)std::max<int64_t>(m, 1) * ld * sizeof(T), hsz = (size_t)V * ld * sizeof(T);
      const size_t gel = (size_t)ld * ld, gsz = gel * sizeof(T);
      for (int i = 0; i < 2; ++i) {
        E_HIP(Wb[i].ensure(wsz));
        E_HIP(Hb[i].ensure(hsz));
      }
      E_HIP(Nw.ensure(wsz));
      E_HIP(Ex.ensure(hsz + gsz));
      E_HIP(small.ensure((size_t)(4 * n + 1) * gsz));   // per problem: W^T W, H H^T, dW gram, dH gram; + the concatenated Gram
      T* Nh = Ex.template as<T>();
      T* sm = small.template as<T>();
      auto Gw = [&](int p) { return sm + (size_t)(4 * p) * gel; };
      auto Gh = [&](int p) { return sm + (size_t)(4 * p + 1) * gel; };
      auto Gd = [&](int p) { return sm + (size_t)(4 * p + 2) * gel; };   // [dW gram | dH gram]
      T* Gcat = sm + (size_t)(4 * n) * gel;
      MNMF_TRY(ensure_prepared());
      E_HIP(hipMemsetAsync(Wb[0].p, 0, wsz, st()));   // columns beyond sum(k) stay zero
      E_HIP(hipMemsetAsync(Hb[0].p, 0, hsz, st()));
      for (int p = 0; p < n; ++p) {
        if (m > 0) {
          put_cols_kernel<T><<<(unsigned)cdiv(m * ks[p], 256), 256, 0, st()>>>((const T*)Ws[p], m, ks[p], ld, off[p], 0, Wb[0].template as<T>(), nullptr);
          E_LAUNCH();
        }
        put_cols_kernel<T><<<(unsigned)cdiv(V * ks[p], 256), 256, 0, st()>>>((const T*)Hs[p], V, ks[p], ld, off[p], 1, Hb[0].template as<T>(), vmap());
        E_LAUNCH();
      }
      E_HIP(hipMemcpyAsync(Wb[1].p, Wb[0].p, wsz, hipMemcpyDeviceToDevice, st()));   // the other halves: same zero columns
      E_HIP(hipMemcpyAsync(Hb[1].p, Hb[0].p, hsz, hipMemcpyDeviceToDevice, st()));
      std::vector<int> state(n, 0);           // 0 running, 1 stopped by the rule (factors copied out)
      std::vector<double> wn(n, INFINITY), hn(n, INFINITY);
      std::vector<int64_t> it_done(n, 0);
      int cw = 0, ch = 0;
      int64_t i = 0, done = 0;
      bool gw_valid = false, pending = false;
      LoopClock clk;
      MNMF_TRY(clock_start(clk));
      std::vector<double> g;
      MNMF_TRY(pinned((size_t)n * 2 * gel));
      auto copy_out = [&](int p, const T* Wsrc, const T* Hsrc) -> int {
        const T t = (T)(finalize ? zthr : -1.0);
        if (m > 0) {
          unpad_rows_kernel<T><<<(unsigned)cdiv(m * ks[p], 256), 256, 0, st()>>>(Wsrc, m, ks[p], ld, t, (T*)Ws[p], nullptr, off[p]);   // :86
          E_LAUNCH();
        }
        get_cols_t_kernel<T><<<(unsigned)cdiv(V * ks[p], 256), 256, 0, st()>>>(Hsrc, V, ks[p], ld, off[p], t, (T*)Hs[p], vmap());     // :87
        E_LAUNCH();
        return 0;
      };
      auto fill = [&](int p, int conv) {
        if (!stats) return;
        stats[p].iters = (int32_t)it_done[p];
        stats[p].converged = conv;
        stats[p].w_norm = wn[p];
        stats[p].h_norm = hn[p];
        stats[p].s_norm = 0;
      };
      // norms of the last checked iteration; problems that meet the rule are copied out of the CURRENT halves, which
      // still hold that iteration's factors (see run_nmf: the next iteration has only written the other halves)
      auto resolve = [&]() -> int {
        MNMF_TRY(fetch_end(g));
        pending = false;
        for (int p = 0; p < n; ++p) {
          if (state[p]) continue;
          wn[p] = spectral_from_gram(g.data() + (size_t)(2 * p) * gel, ld);
          hn[p] = spectral_from_gram(g.data() + (size_t)(2 * p + 1) * gel, ld);
          if (!std::isfinite(wn[p]) || !std::isfinite(hn[p]))
            return fail(MNMF_ENAN, "nmf sweep: non-finite update norm of problem " + std::to_string(p) + " at iteration " + std::to_string(done));
          if (fixed <= 0 && wn[p] < thr && hn[p] < thr) {
            state[p] = 1;
            it_done[p] = done;
            MNMF_TRY(copy_out(p, Wb[cw].template as<T>(), Hb[ch].template as<T>()));
            fill(p, 1);
          }
        }
        return 0;
      };
      auto all_stopped = [&]() {
        for (int p = 0; p < n; ++p)
          if (!state[p]) return false;
        return true;
      };
      while (true) {
        T* Wc = Wb[cw].template as<T>();
        T* Wn = Wb[1 - cw].template as<T>();
        T* Hc = Hb[ch].template as<T>();
        T* Hn = Hb[1 - ch].template as<T>();
        const bool do_check = (fixed > 0) ? (force_check && ((i + 1) % every == 0)) : ((i + 1) % every == 0);
        if (!gw_valid) {  // w.T @ w of every problem: the diagonal blocks of the concatenated Gram      nmf_basic.py:39
          MNMF_TRY(gram(Wc, nullptr, Wc, nullptr, m, ld, false, Gcat));
          for (int p = 0; p < n; ++p) {
            diag_block_kernel<T><<<(unsigned)cdiv((int64_t)gel, 256), 256, 0, st()>>>(Gcat, ld, off[p], ks[p], Gw(p));
            E_LAUNCH();
          }
        }
        MNMF_TRY(spmm(Xt, Wc, ld, Nh));                                    // every problem's (w.T @ in_mat)^T     :40
        for (int p = 0; p < n; ++p) {                                      // h update + h @ h.T + |dh| gram        :41-44, :53
          if (state[p]) continue;
          RkArgs<T> a{Hc, Gw(p), nullptr, nullptr, Hc, Nh, Hn, V, ld, (T)eps, off[p], ks[p]};
          MNMF_TRY(update(a, Gh(p), do_check ? Gd(p) + gel : nullptr));
        }
        MNMF_TRY(spmm(X, Hn, ld, Nw.template as<T>()));                    // every problem's in_mat @ h.T         :45
        if (pending) {
          const int r = resolve();
          if (r != 0) return r;
          if (all_stopped()) break;
        }
        for (int p = 0; p < n; ++p) {                                      // w update + next w.T @ w + |dw| gram   :46-47, :39, :52
          if (state[p]) continue;
          RkArgs<T> a{Wc, Gh(p), nullptr, nullptr, Wc, Nw.template as<T>(), Wn, m, ld, (T)eps, off[p], ks[p]};
          MNMF_TRY(update(a, Gw(p), do_check ? Gd(p) : nullptr));
        }
        gw_valid = true;
        if (do_check) {
          MNMF_TRY(fetch_begin((size_t)n * 2 * gel));
          for (int p = 0; p < n; ++p) MNMF_TRY(fetch_add(Gd(p), 2 * gel));
          MNMF_TRY(fetch_commit());
          pending = true;
        }
        cw = 1 - cw;
        ch = 1 - ch;
        ++done;
        if (fixed > 0) {
          if (i + 1 >= fixed) break;
        } else {
          if (i >= max_iter) break;                                        // :57
        }
        ++i;
      }
      if (pending) {
        const int r = resolve();
        if (r != 0) return r;
      }
      MNMF_TRY(clock_stop(clk, stats));
      const double ms_total = stats ? stats[0].ms_total : 0.0, ms_kernels = stats ? stats[0].ms_kernels : 0.0;
      for (int p = 0; p < n; ++p) {
        if (!state[p]) {   // ran to the iteration cap (or a fixed count): the current factors
          it_done[p] = done;
          MNMF_TRY(copy_out(p, Wb[cw].template as<T>(), Hb[ch].template as<T>()));
          fill(p, 0);
        }
        if (stats) {
          stats[p].ms_total = ms_total;      // of the whole batch
          stats[p].ms_kernels = ms_kernels;
          stats[p].bytes_model = 0;
          stats[p].flops_model = 0;
          std::memset(stats[p].phase_ms, 0, sizeof(stats[p].phase_ms));
        }
      }
      E_HIP(hipStreamSynchronize(st()));
      return 0;
    }
  }

  // ------------------------------------------------------------------ loop B
  int run_pnmf(int k, void* W, void* H, const mnmf_opts* o, mnmf_stats* stats) override {
    ColsGuard cols_guard(data_cols, k);
    MNMF_TRY(check_ready(k));
    const int ld = ld_of(k);
    const int64_t max_iter = (o && o->max_iter >= 0) ? o->max_iter : 1000;
    const int fixed = o ? o->fixed_iters : 0;
    const int every = (o && o->check_every > 0) ? o->check_every : 1;
    const bool force_check = o && o->force_check;
    const double delta = (o && o->norm_thresh >= 0) ? o->norm_thresh : 0.005;
    const double eps = (o && o->eps >= 0) ? o->eps : 1e-10;
    const double zthr = (o && o->zero_threshold >= 0) ? o->zero_threshold : 1e-10;
    const bool finalize = !o || o->finalize;
    const size_t wsz = (size_t)std::max<int64_t>(m, 1) * ld * sizeof(T), hsz = (size_t)V * ld * sizeof(T);
    const size_t gsz = (size_t)ld * ld * sizeof(T);
    for (int i = 0; i < 2; ++i) E_HIP(Wb[i].ensure(wsz));
    E_HIP(Nw.ensure(wsz));         // pay
    E_HIP(Ex.ensure(hsz + gsz));   // [vtw (V x ld) | wtw]
    E_HIP(small.ensure(4 * gsz));  // wt_pay, dW gram, colnorm gram, norms
    T* vtw = Ex.template as<T>();
    T* wtw = vtw + (size_t)V * ld;
    T* wt_pay = small.template as<T>();
    T* Gd = wt_pay + (size_t)ld * ld;
    T* Gn = Gd + (size_t)ld * ld;
    T* nrm = Gn + (size_t)ld * ld;
    T* pay = Nw.template as<T>();
    MNMF_TRY(load_W(W, k, ld, Wb[0].template as<T>()));
    int cw = 0;
    double w_norm = INFINITY;
    int64_t i = 0;
    int converged = 0;
    const int64_t limit = fixed > 0 ? fixed : max_iter;
    bool wtw_valid = false;  // wtw holds this rank's W.T@W of the current W (left there by the previous update)
    LoopClock clk;
    MNMF_TRY(clock_start(clk));
    PhaseTimer pt(st(), o && o->profile_phases);
    pt.mark(-1, 0);
    std::vector<double> g;
    // The norm of iteration i is read while iteration i+1's SpMMs are already queued (they only read W_{i+1}); the
    // reference breaks BEFORE assigning w_new (:101-107), so when iteration i turns out to have converged the flip
    // is undone -- W_i is still intact, iteration i+1 has not reached its update yet.
    bool pending = false;
    auto resolve = [&]() -> int {
      MNMF_TRY(fetch_end(g));
      pending = false;
      w_norm = spectral_from_gram(g.data(), ld);
      if (!std::isfinite(w_norm)) return fail(MNMF_ENAN, "pnmf: non-finite update norm at iteration " + std::to_string(i));
      return 0;
    };
    MNMF_TRY(pinned((size_t)ld * ld));
    while (i < limit) {                                                  // nmf_projective_basic.py:61
      T* Wc = Wb[cw].template as<T>();
      T* Wn = Wb[1 - cw].template as<T>();
      const bool do_check = (fixed > 0) ? (force_check && ((i + 1) % every == 0)) : ((i + 1) % every == 0);
      if (!wtw_valid) MNMF_TRY(gram(Wc, nullptr, Wc, nullptr, m, ld, false, wtw));  // wtw = W.T @ W   :68
      pt.mark(0, 1);
      MNMF_TRY(spmm(Xt, Wc, ld, vtw));                                   // vtw = X.T @ W       :69
      pt.mark(1, 4);
      MNMF_TRY(allreduce_rows(vtw, (int64_t)V * ld + (int64_t)ld * ld, k, ld));
      MNMF_TRY(spmm(X, vtw, ld, pay));                                   // pay = X @ vtw       :72
      pt.mark(4, 3);
      // wt_pay = W.T @ pay (:75) = W.T @ X @ vtw = vtw.T @ vtw: the Gram of the (summed) V x k matrix instead of a
      // product over the m document rows -- symmetric, a tenth of the rows at C3, and in a sharded run every rank
      // computes it from the exchanged vtw, so the second all-reduce of the iteration is gone.
      MNMF_TRY(gram(vtw, nullptr, vtw, nullptr, V, ld, false, wt_pay));
      pt.mark(3, pending ? 6 : 5);
      if (pending) {  // norm of the previous iteration                  :83, :101
        const int r = resolve();
        pt.mark(6, 5);
        if (r != 0) {
          (void)clock_stop(clk, stats);
          return r;
        }
        if (fixed <= 0 && w_norm < delta) {                              // break BEFORE W = w_new   :101-107
          converged = 1;
          cw = 1 - cw;
          --i;
          break;
        }
      }
      // payda = W @ wt_pay + pay @ wtw + eps ; w_new = W * sqrt(pay / payda)                  :76-79
      // ... with the next iteration's wtw (:68) and D^T D of D = |w_new - W| (:83) in the same pass
      MNMF_TRY(rk_grams(RK_MU_SQRT, Wc, wt_pay, pay, wtw, Wc, pay, Wn, m, ld, eps, wtw, do_check ? Gd : nullptr));
      wtw_valid = true;
      pt.mark(5, do_check ? 6 : 0);
      if (do_check) {
        MNMF_TRY(allreduce(Gd, (int64_t)ld * ld));
        MNMF_TRY(fetch_begin((size_t)ld * ld));
        MNMF_TRY(fetch_add(Gd, (size_t)ld * ld));
        MNMF_TRY(fetch_commit());
        pending = true;
        pt.mark(6, 0);
      }
      cw = 1 - cw;
      ++i;
    }
    if (pending) {  // the loop ended on its iteration cap: the last iteration's norm
      const int r = resolve();
      if (r != 0) {
        (void)clock_stop(clk, stats);
        return r;
      }
      if (fixed <= 0 && w_norm < delta) {
        converged = 1;
        cw = 1 - cw;
        --i;
      }
    }
    MNMF_TRY(clock_stop(clk, stats));
    pt.collect(phase_ms);
    if (stats) std::memcpy(stats->phase_ms, phase_ms, sizeof(phase_ms));
    T* Wc = Wb[cw].template as<T>();
    if (finalize) {
      threshold_kernel<T><<<(unsigned)cdiv(std::max<int64_t>(m * ld, 1), 256), 256, 0, st()>>>(Wc, m * ld, (T)zthr);  // :111
      E_LAUNCH();
      MNMF_TRY(gram(Wc, nullptr, Wc, nullptr, m, ld, false, Gn));        // column sums of squares = diag
      MNMF_TRY(allreduce(Gn, (int64_t)ld * ld));
      const T tiny = (T)(10.0 * (sizeof(T) == 8 ? 2.220446049250313e-16 : 1.1920929e-07));
      inv_colnorm_kernel<T><<<(unsigned)cdiv(ld, 256), 256, 0, st()>>>(Gn, ld, tiny, nrm);
      E_LAUNCH();
      if (m > 0) {
        div_cols_kernel<T><<<(unsigned)cdiv(m * ld, 256), 256, 0, st()>>>(Wc, m, ld, nrm);  // normalize(W, 'l2', axis=0) :114
        E_LAUNCH();
      }
    }
    MNMF_TRY(spmm(Xt, Wc, ld, vtw));                                      // h = w.T @ X         :115
    MNMF_TRY(allreduce_rows(vtw, (int64_t)V * ld, k, ld));
    MNMF_TRY(store_W(Wc, k, ld, -1.0, W));
    MNMF_TRY(store_H(vtw, k, ld, -1.0, H));
    E_HIP(hipStreamSynchronize(st()));
    if (stats) {
      stats->iters = (int32_t)i;
      stats->converged = converged;
      stats->w_norm = w_norm;
      stats->h_norm = 0;
      stats->s_norm = 0;
      model(stats, k, 2, 6, 3, 4, 8, 0);
    }
    return 0;
  }

  // ------------------------------------------------------------------ loop C
  int run_nmtf(int k, void* W, void* S, void* H, const mnmf_opts* o, mnmf_stats* stats) override {
    ColsGuard cols_guard(data_cols, k);
    MNMF_TRY(check_ready(k));
    const int ld = ld_of(k);
    const int64_t max_iter = (o && o->max_iter >= 0) ? o->max_iter : 1000;
    const int fixed = o ? o->fixed_iters : 0;
    const int every = (o && o->check_every > 0) ? o->check_every : 1;
    const bool force_check = o && o->force_check;
    const double thr = (o && o->norm_thresh >= 0) ? o->norm_thresh : 0.005;
    const double eps = (o && o->eps >= 0) ? o->eps : 1e-9;
    const double zthr = (o && o->zero_threshold >= 0) ? o->zero_threshold : 1e-5;
    const bool finalize = !o || o->finalize;
    const bool sliced = use_slices();  // sharded: reduce-scatter -> S and this rank's slice of H^T -> all-gather
    const int64_t Vp = padded_V(), Vs = slice_rows();
    const int64_t s_lo = sliced ? Vs * ctx->rank : 0;
    const int64_t s_rows = sliced ? std::max<int64_t>(0, std::min<int64_t>(Vs, V - s_lo)) : V;
    const size_t wsz = (size_t)std::max<int64_t>(m, 1) * ld * sizeof(T), hsz = (size_t)Vp * ld * sizeof(T);
    const size_t gel = (size_t)ld * ld, gsz = gel * sizeof(T);
    for (int i = 0; i < 2; ++i) {
      E_HIP(Wb[i].ensure(wsz));
      E_HIP(Hb[i].ensure(hsz));
    }
    E_HIP(Nw.ensure(wsz));
    E_HIP(Ex.ensure(hsz + gsz));     // [Pt (V x ld) | w1T w1]
    E_HIP(Aux1.ensure(hsz));         // M1 = hT sT, later M2 = Pt s1
    E_HIP(small.ensure(16 * gsz));
    T* Pt = Ex.template as<T>();
    // sliced: w1.T@w1 in front of the dW Gram (same w update, one all-reduce for both); otherwise behind the partial
    T* Gw1 = sliced ? small.template as<T>() + 10 * gel : Pt + (size_t)Vp * ld;
    if (Vp > V) {  // the padding rows of the sliced buffers are zero and stay zero
      const size_t off = (size_t)V * ld * sizeof(T), len = (size_t)(Vp - V) * ld * sizeof(T);
      E_HIP(hipMemsetAsync(Hb[0].template as<char>() + off, 0, len, st()));
      E_HIP(hipMemsetAsync(Hb[1].template as<char>() + off, 0, len, st()));
      E_HIP(hipMemsetAsync(Ex.template as<char>() + off, 0, len, st()));
    }
    T* M1 = Aux1.template as<T>();
    T* sm = small.template as<T>();
    T* Sb[2] = {sm, sm + gel};
    T* ST = sm + 2 * gel;
    T* hhT = sm + 3 * gel;
    T* T1 = sm + 4 * gel;
    T* Dm = sm + 5 * gel;
    T* num_s = sm + 6 * gel;
    T* T2 = sm + 7 * gel;
    T* den_s = sm + 8 * gel;
    T* T3 = sm + 9 * gel;
    T* Qm = sm + 13 * gel;
    T* Gd = sm + 11 * gel;  // [dW gram | dH gram]; sm + 10 * gel: w1.T@w1 of a sliced run
    // S in: k x k -> ld x ld padded
    pad_sq(S, k, ld, Sb[0]);
    E_LAUNCH();
    MNMF_TRY(load_W(W, k, ld, Wb[0].template as<T>()));
    MNMF_TRY(load_H(H, k, ld, Hb[0].template as<T>()));
    int cw = 0, ch = 0, cs = 0;
    double w_norm = INFINITY, h_norm = INFINITY, s_norm = INFINITY;
    int64_t i = 0;
    int converged = 0;
    LoopClock clk;
    MNMF_TRY(clock_start(clk));
    PhaseTimer pt(st(), o && o->profile_phases);
    pt.mark(-1, 2);
    std::vector<double> g;
    const unsigned kb = (unsigned)cdiv((int64_t)gel, 256);
    // k x k chains in one launch each (dense.hpp) while one workgroup is quick enough: measured +14 % / +4 % / +1 %
    // on the iteration at k = 10 / 20 (1.8e6, 1e7, 2e8 nnz), -6 % at ld = 64 (4096 elements of 64 terms on one CU)
    const bool chain = tune.kxk_chain && ld <= 32;
    const size_t chain_lds = 2 * gsz;
    bool hht_valid = false;  // hhT holds h @ h.T of the current h (left there by the previous h update)
    // The norms of iteration i are read while iteration i+1 is already queued up to its W update: that work only
    // writes scratch and the other halves of the ping-pong buffers, so when iteration i turns out to have converged
    // the factors it produced are still current and the queued work is simply dropped.
    bool pending = false;
    auto resolve = [&]() -> int {
      MNMF_TRY(fetch_end(g));
      pending = false;
      w_norm = fro_from_gram(g.data(), ld);
      h_norm = fro_from_gram(g.data() + gel, ld);
      double acc = 0;
      for (size_t e = 0; e < gel; ++e) {
        const double d = g[2 * gel + e] - g[3 * gel + e];
        acc += d * d;
      }
      s_norm = std::sqrt(acc);
      if (!std::isfinite(w_norm) || !std::isfinite(h_norm) || !std::isfinite(s_norm))
        return fail(MNMF_ENAN, "nmtf: non-finite update norm at iteration " + std::to_string(i));
      return 0;
    };
    MNMF_TRY(pinned(4 * gel));
    while (true) {
      T* Wc = Wb[cw].template as<T>();
      T* Wn = Wb[1 - cw].template as<T>();
      T* Hc = Hb[ch].template as<T>();
      T* Hn = Hb[1 - ch].template as<T>();
      T* Sc = Sb[cs];
      T* Sn = Sb[1 - cs];
      // ---- W update                                                   nmtf.py:151-153
      if (chain) {  // one launch: s^T, s (h hT), s (h hT) sT
        if (!hht_valid) MNMF_TRY(gram(Hc, nullptr, Hc, nullptr, V, ld, false, hhT));  // h @ h.T
        nmtf_chain_w_kernel<T><<<1, 1024, chain_lds, st()>>>(Sc, hhT, ST, T1, Dm, ld);
        E_LAUNCH();
      } else {
        kxk_transpose_kernel<T><<<kb, 256, 0, st()>>>(Sc, ST, ld);
        E_LAUNCH();
      }
      MNMF_TRY(rk(RK_PROD, Hc, ST, nullptr, nullptr, nullptr, nullptr, M1, V, ld, 0));  // h.T @ s.T
      pt.mark(2, 4);
      MNMF_TRY(spmm(X, M1, ld, Nw.template as<T>()));                             // in_mat @ (h.T @ s.T)
      pt.mark(4, 3);
      if (!chain) {
        if (!hht_valid) MNMF_TRY(gram(Hc, nullptr, Hc, nullptr, V, ld, false, hhT));  // h @ h.T
        pt.mark(3, pending ? 6 : 5);
        MNMF_TRY(kxk(Sc, hhT, T1, ld, 0, 0));                              // s (h hT)
        MNMF_TRY(kxk(T1, Sc, Dm, ld, 0, 1));                               // s (h hT) sT
      }
      if (pending) {  // norms of the previous iteration                 :179-181, :200
        const int r = resolve();
        pt.mark(6, 5);
        if (r != 0) {
          (void)clock_stop(clk, stats);
          return r;
        }
        if (fixed <= 0 && w_norm < thr && h_norm < thr && s_norm < thr) {
          converged = 1;
          break;
        }
      }
      const bool do_check = (fixed > 0) ? (force_check && ((i + 1) % every == 0)) : ((i + 1) % every == 0);
      // w update with w1.T @ w1 (:157) and, for the norms, D^T D of D = |w1 - w| (:179) in the same pass
      MNMF_TRY(rk_grams(RK_MU, Wc, Dm, nullptr, nullptr, Wc, Nw.template as<T>(), Wn, m, ld, eps, Gw1, do_check ? Gd : nullptr));
      if (sliced) MNMF_TRY(allreduce(Gw1, (int64_t)gel * (do_check ? 2 : 1)));   // [w1.T@w1 | dW gram]
      pt.mark(5, 0);
      // ---- S update                                                   nmtf.py:156-158
      pt.mark(0, 1);
      MNMF_TRY(spmm(Xt, Wn, ld, Pt));                                    // (w1.T @ in_mat)^T
      pt.mark(1, 3);
      const size_t so = (size_t)s_lo * ld;   // this rank's slice of the vocabulary rows (everything when not sliced)
      if (sliced) {
        MNMF_TRY(reduce_scatter_rows(Pt, k, ld));
        MNMF_TRY(gram(Pt + so, nullptr, Hc + so, nullptr, s_rows, ld, false, num_s));   // this slice's share of (w1.T X) @ h.T
        MNMF_TRY(allreduce(num_s, (int64_t)gel));
      } else {
        MNMF_TRY(allreduce_rows(Pt, (int64_t)V * ld + (int64_t)gel, k, ld));
        MNMF_TRY(gram(Pt, nullptr, Hc, nullptr, V, ld, false, num_s));  // (w1.T X) @ h.T
      }
      if (chain) {  // one launch: (w1T w1) s (h hT), the s update, s1T (w1T w1) s1
        nmtf_chain_s_kernel<T><<<1, 1024, chain_lds, st()>>>(Gw1, Sc, hhT, num_s, (T)eps, T2, den_s, Sn, T3, Qm, ld);
        E_LAUNCH();
      } else {
        MNMF_TRY(kxk(Gw1, Sc, T2, ld, 0, 0));
        MNMF_TRY(kxk(T2, hhT, den_s, ld, 0, 0));                           // (w1T w1) s (h hT)
        kxk_mu_kernel<T><<<kb, 256, 0, st()>>>(Sc, num_s, den_s, (T)eps, Sn, ld);
        E_LAUNCH();
      }
      pt.mark(3, 2);
      // ---- H update                                                   nmtf.py:161-163
      MNMF_TRY(rk(RK_PROD, Pt + so, Sn, nullptr, nullptr, nullptr, nullptr, M1 + so, s_rows, ld, 0));  // (s1.T @ P)^T = Pt s1
      if (!chain) {
        MNMF_TRY(kxk(Sn, Gw1, T3, ld, 1, 0));                              // s1T (w1T w1)
        MNMF_TRY(kxk(T3, Sn, Qm, ld, 0, 0));                               // s1T (w1T w1) s1   (symmetric)
      }
      if (sliced) {
        MNMF_TRY(rk(RK_MU, Hc + so, Qm, nullptr, nullptr, Hc + so, M1 + so, Hn + so, s_rows, ld, eps));
        MNMF_TRY(all_gather_rows(Hn, k, ld));
        MNMF_TRY(gram(Hn, nullptr, Hn, nullptr, V, ld, false, hhT));                                   // :152
        if (do_check) MNMF_TRY(gram(Hn, Hc, Hn, Hc, V, ld, true, Gd + gel));                           // :181
      } else {
        // h update with the next iteration's h @ h.T (:152) and D^T D of D = |h1 - h| (:181) in the same pass
        MNMF_TRY(rk_grams(RK_MU, Hc, Qm, nullptr, nullptr, Hc, M1, Hn, V, ld, eps, hhT, do_check ? Gd + gel : nullptr));
      }
      hht_valid = true;
      pt.mark(2, do_check ? 6 : 2);
      if (do_check) {                                                     // Frobenius norms   nmtf.py:179-181
        if (!sliced) MNMF_TRY(allreduce(Gd, (int64_t)gel));
        MNMF_TRY(fetch_begin(4 * gel));
        MNMF_TRY(fetch_add(Gd, 2 * gel));          // [dW gram | dH gram]
        MNMF_TRY(fetch_add(Sb[0], 2 * gel));       // both S buffers: ||s1 - s||_F on the host
        MNMF_TRY(fetch_commit());
        pending = true;
        pt.mark(6, 2);
      }
      cw = 1 - cw;
      ch = 1 - ch;
      cs = 1 - cs;
      ++i;                                                                // :197
      if (fixed > 0) {
        if (i >= fixed) break;
      } else {
        if (i > max_iter) break;                                          // :204 (the norms of this iteration: below)
      }
    }
    if (pending) {  // the loop ended on its iteration cap or count: the last iteration's norms       :200
      const int r = resolve();
      if (r != 0) {
        (void)clock_stop(clk, stats);
        return r;
      }
      if (fixed <= 0 && w_norm < thr && h_norm < thr && s_norm < thr) converged = 1;
    }
    MNMF_TRY(clock_stop(clk, stats));
    pt.collect(phase_ms);
    if (stats) std::memcpy(stats->phase_ms, phase_ms, sizeof(phase_ms));
    MNMF_TRY(store_W(Wb[cw].template as<T>(), k, ld, finalize ? zthr : -1.0, W));   // :210-212
    MNMF_TRY(store_H(Hb[ch].template as<T>(), k, ld, finalize ? zthr : -1.0, H));
    unpad_sq(Sb[cs], k, ld, finalize ? zthr : -1.0, S);
    E_LAUNCH();
    E_HIP(hipStreamSynchronize(st()));
    if (stats) {
      stats->iters = (int32_t)i;
      stats->converged = converged;
      stats->w_norm = w_norm;
      stats->h_norm = h_norm;
      stats->s_norm = s_norm;
      model(stats, k, 2, 5, 9, 4, 6, 10);
    }
    return 0;
  }

  void pad_sq(const void* S, int k, int ld, T* dst) {
    pad_sq_kernel<T><<<(unsigned)cdiv((int64_t)ld * ld, 256), 256, 0, st()>>>((const T*)S, k, ld, dst);
  }
  void unpad_sq(const T* src, int k, int ld, double thr, void* S) {
    unpad_rows_kernel<T><<<(unsigned)cdiv((int64_t)k * k, 256), 256, 0, st()>>>(src, k, k, ld, (T)thr, (T*)S);
  }

  // ------------------------------------------------------------------ kernel-level entry points
  int k_spmm(int transpose, int k, const void* B, void* out) override {
    MNMF_TRY(check_ready(k));
    ColsGuard cols_guard(data_cols, k);
    const int ld = ld_of(k);
    StreamMat<T>& M = transpose ? Xt : X;
    E_HIP(tmpA.ensure((size_t)std::max<int64_t>(M.tab_rows, 1) * ld * sizeof(T)));
    E_HIP(tmpB.ensure((size_t)std::max<int64_t>(M.rows, 1) * ld * sizeof(T)));
    MNMF_TRY(ensure_prepared());
    // vocabulary-indexed operands (B for X.B, the result for X^T.B) go through the renumbering
    pad_rows_kernel<T><<<(unsigned)cdiv(std::max<int64_t>(M.tab_rows * ld, 1), 256), 256, 0, st()>>>(
        (const T*)B, M.tab_rows, k, ld, tmpA.template as<T>(), transpose ? nullptr : vmap());
    E_LAUNCH();
    MNMF_TRY(spmm(M, tmpA.template as<T>(), ld, tmpB.template as<T>()));
    if (M.rows * k > 0) {
      unpad_rows_kernel<T><<<(unsigned)cdiv(M.rows * k, 256), 256, 0, st()>>>(tmpB.template as<T>(), M.rows, k, ld, (T)-1, (T*)out,
                                                                             transpose ? vmap() : nullptr);
      E_LAUNCH();
    }
    E_HIP(hipStreamSynchronize(st()));
    return 0;
  }

  int pad_into(DevBuf& b, const void* src, int64_t rows, int k, int ld) {
    E_HIP(b.ensure((size_t)std::max<int64_t>(rows, 1) * ld * sizeof(T)));
    pad_rows_kernel<T><<<(unsigned)cdiv(std::max<int64_t>(rows * ld, 1), 256), 256, 0, st()>>>((const T*)src, rows, k, ld, b.template as<T>());
    E_LAUNCH();
    return 0;
  }

  int k_gram(int64_t rows, int k, const void* A, const void* B, void* out) override {
    if (k <= 0 || rows < 0) return fail(MNMF_EINVAL, "gram: bad shape");
    E_HIP(hipSetDevice(ctx->device));
    const int ld = ld_of(k);
    MNMF_TRY(pad_into(tmpA, A, rows, k, ld));
    if (B) MNMF_TRY(pad_into(tmpB, B, rows, k, ld));
    E_HIP(tmpC.ensure((size_t)ld * ld * sizeof(T)));
    MNMF_TRY(gram(tmpA.template as<T>(), nullptr, B ? tmpB.template as<T>() : tmpA.template as<T>(), nullptr, rows, ld, false, tmpC.template as<T>()));
    unpad_sq(tmpC.template as<T>(), k, ld, -1.0, out);
    E_LAUNCH();
    E_HIP(hipStreamSynchronize(st()));
    return 0;
  }

  int k_update(int64_t rows, int k, const void* A, const void* num, const void* G, const void* G2, double eps,
               int sqrt_mode, void* out) override {
    if (k <= 0 || rows < 0) return fail(MNMF_EINVAL, "update: bad shape");
    E_HIP(hipSetDevice(ctx->device));
    const int ld = ld_of(k);
    const size_t gel = (size_t)ld * ld;
    MNMF_TRY(pad_into(tmpA, A, rows, k, ld));
    MNMF_TRY(pad_into(tmpB, num, rows, k, ld));
    E_HIP(tmpC.ensure(2 * gel * sizeof(T)));
    E_HIP(tmpD.ensure((size_t)std::max<int64_t>(rows, 1) * ld * sizeof(T)));
    pad_sq(G, k, ld, tmpC.template as<T>());
    E_LAUNCH();
    if (sqrt_mode) {
      if (!G2) return fail(MNMF_EINVAL, "update: sqrt_mode needs G2");
      pad_sq(G2, k, ld, tmpC.template as<T>() + gel);
      E_LAUNCH();
      MNMF_TRY(rk(RK_MU_SQRT, tmpA.template as<T>(), tmpC.template as<T>(), tmpB.template as<T>(), tmpC.template as<T>() + gel, tmpA.template as<T>(), tmpB.template as<T>(),
                  tmpD.template as<T>(), rows, ld, eps));
    } else {
      MNMF_TRY(rk(RK_MU, tmpA.template as<T>(), tmpC.template as<T>(), nullptr, nullptr, tmpA.template as<T>(), tmpB.template as<T>(), tmpD.template as<T>(), rows,
                  ld, eps));
    }
    if (rows * k > 0) {
      unpad_rows_kernel<T><<<(unsigned)cdiv(rows * k, 256), 256, 0, st()>>>(tmpD.template as<T>(), rows, k, ld, (T)-1, (T*)out);
      E_LAUNCH();
    }
    E_HIP(hipStreamSynchronize(st()));
    return 0;
  }

  int k_topk(int64_t rows, int64_t cols, const void* A, int by_column, int n, int32_t* idx, void* val) override {
    if (rows <= 0 || cols <= 0 || n <= 0 || n > 1024 || !A || !idx || !val) return fail(MNMF_EINVAL, "topk: bad arguments");
    if ((by_column ? rows : cols) >= ((int64_t)1 << 31)) return fail(MNMF_EINVAL, "topk: more than 2^31 candidates per topic");
    E_HIP(hipSetDevice(ctx->device));
    topk_kernel<T><<<(unsigned)(by_column ? cols : rows), 256, 0, st()>>>((const T*)A, rows, cols, by_column, n, idx, (T*)val);
    E_LAUNCH();
    E_HIP(hipStreamSynchronize(st()));
    return 0;
  }

  int k_s_confusion(int k, const void* S, int32_t* col, int32_t* row, void* val) override {
    if (k <= 0 || k > 1024 || !S || !col || !row || !val) return fail(MNMF_EINVAL, "s_confusion: bad arguments (1 <= k <= 1024)");
    E_HIP(hipSetDevice(ctx->device));
    const int threads = (int)round_up(k, 64);
    s_confusion_kernel<T><<<1, threads, (size_t)k * sizeof(T), st()>>>((const T*)S, k, k, col, row, (T*)val);
    E_LAUNCH();
    E_HIP(hipStreamSynchronize(st()));
    return 0;
  }

  int k_diff_norm(int64_t rows, int k, const void* A, const void* B, int kind, double* out) override {
    if (k <= 0 || rows < 0 || !out) return fail(MNMF_EINVAL, "diff_norm: bad arguments");
    E_HIP(hipSetDevice(ctx->device));
    const int ld = ld_of(k);
    MNMF_TRY(pad_into(tmpA, A, rows, k, ld));
    MNMF_TRY(pad_into(tmpB, B, rows, k, ld));
    E_HIP(tmpC.ensure((size_t)ld * ld * sizeof(T)));
    MNMF_TRY(gram(tmpA.template as<T>(), tmpB.template as<T>(), tmpA.template as<T>(), tmpB.template as<T>(), rows, ld, true, tmpC.template as<T>()));
    std::vector<double> g;
    MNMF_TRY(fetch_grams(tmpC.template as<T>(), 1, ld, g));
    *out = kind == 0 ? spectral_from_gram(g.data(), ld) : fro_from_gram(g.data(), ld);
    return 0;
  }
};

}  // namespace

#ifndef MNMF_ENGINE_DTYPE
#error "compile engine.hip with -DMNMF_ENGINE_DTYPE=32 or 64"
#endif
#if MNMF_ENGINE_DTYPE == 32
ImplBase* mnmf_make_engine_f32(mnmf_ctx* c) { return new Engine<float>(c); }
#else
ImplBase* mnmf_make_engine_f64(mnmf_ctx* c) { return new Engine<double>(c); }
#endif
