// C-ABI of libmnmf.so (include/mnmf.h): context lifetime, dispatch to the typed engine, host-only helpers.
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/mnmf.h"
#include "common.hpp"
#include "ctx.hpp"
#include "dense.hpp"
#include "eig.hpp"
#include "impl_base.hpp"
#include "sparse_prep.hpp"

using namespace mnmf;

// ---------------------------------------------------------------------------------------------------------------
// Scratch of mnmf_gram_static, the one entry point that allocates nothing (include/mnmf.h).  A file-scope device
// array: every GPU the library runs on holds its own copy, at its own address, until the process ends.  In host code
// the array's NAME is only a stand-in for that symbol, not a device address; the address is asked for once per device
// (mnmf_create, with that device current) through hipGetSymbolAddress and kept in g_scratch[device].  Host code
// below uses that pointer only -- kernel arguments included.
constexpr size_t kScratchBytes = (size_t)8 << 20;   // partial Grams: n_chunks x k x k elements
constexpr int kStaticGramMaxK = 256;                // 256 x 256 doubles = 512 KiB per partial -> at least 16 chunks
constexpr int kMaxDevices = 64;
static __device__ __align__(256) unsigned char scratch[kScratchBytes];
static void* g_scratch[kMaxDevices] = {nullptr};

// once per device, with `device` current; returns hipSuccess when the pointer is known
static hipError_t scratch_setup(int device) {
  if (device < 0 || device >= kMaxDevices) return hipErrorInvalidDevice;
  if (__atomic_load_n(&g_scratch[device], __ATOMIC_ACQUIRE)) return hipSuccess;
  void* p = nullptr;
  const hipError_t e = hipGetSymbolAddress(&p, HIP_SYMBOL(scratch));
  if (e != hipSuccess) return e;
  if (!p) return hipErrorInvalidSymbol;
  __atomic_store_n(&g_scratch[device], p, __ATOMIC_RELEASE);   // every thread would store the same address
  return hipSuccess;
}

template <typename T>
static int gram_static_launch(mnmf_ctx* ctx, void* slab_ptr, int64_t rows, int k, const void* A, const void* B, void* out) {
  const int64_t elems = (int64_t)k * k;
  const int64_t max_chunks = (int64_t)(kScratchBytes / ((size_t)elems * sizeof(T)));
  // same chunking rule as the engine's VALU Gram, capped by what the scratch holds
  int64_t chunks = std::min<int64_t>(std::min<int64_t>(1024, max_chunks), cdiv(rows, 256));
  const int64_t rpc = round_up(cdiv(rows, chunks), 16);
  chunks = cdiv(rows, rpc);
  T* slab = reinterpret_cast<T*>(slab_ptr);
  const T* a = reinterpret_cast<const T*>(A);
  const T* b = B ? reinterpret_cast<const T*>(B) : a;
  const int tiles = (k + 63) / 64;
  // leading dimension = k: the kernels read the caller's unpadded rows directly, no staging copy
  gram_partial_kernel<T, false><<<dim3((unsigned)chunks, (unsigned)(tiles * tiles)), 256, 0, ctx->stream>>>(
      a, nullptr, b, nullptr, rows, k, rpc, slab);
  if (hipGetLastError() != hipSuccess) return MNMF_EHIP;
  // every chunk wrote all k x k entries of its partial: nothing in the scratch has to start at zero
  slab_reduce_kernel<T><<<(unsigned)cdiv(elems, 16), 256, 0, ctx->stream>>>(slab, chunks, elems, reinterpret_cast<T*>(out));
  if (hipGetLastError() != hipSuccess) return MNMF_EHIP;
  return MNMF_OK;
}

// =========================================================================== C-ABI
extern "C" {

const char* mnmf_version(void) { return "mnmf 0.1 (gfx950)"; }

int64_t mnmf_alloc_count(void) { return (int64_t)g_mnmf_allocs.load(std::memory_order_relaxed); }

int mnmf_create(mnmf_ctx** out, int device, int dtype, void* hip_stream) {
  if (!out || (dtype != MNMF_F32 && dtype != MNMF_F64)) return MNMF_EINVAL;
  *out = nullptr;
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || device < 0 || device >= n) return MNMF_EHIP;
  if (hipSetDevice(device) != hipSuccess) return MNMF_EHIP;
  if (scratch_setup(device) != hipSuccess) return MNMF_EHIP;   // address of this GPU's copy of `scratch`
  mnmf_ctx* c = new mnmf_ctx();
  c->device = device;
  c->dtype = dtype;
  c->stream = (hipStream_t)hip_stream;
  c->impl = dtype == MNMF_F32 ? mnmf_make_engine_f32(c) : mnmf_make_engine_f64(c);
  *out = c;
  return MNMF_OK;
}

void mnmf_destroy(mnmf_ctx* ctx) {
  if (!ctx) return;
  (void)hipSetDevice(ctx->device);
  delete ctx->impl;
  delete ctx;
}

const char* mnmf_last_error(const mnmf_ctx* ctx) { return ctx ? ctx->err.c_str() : "null context"; }

int mnmf_row_stride(const mnmf_ctx* ctx, int k) { return (ctx && k > 0) ? ctx->impl->row_stride(k) : 0; }

int mnmf_set_allreduce(mnmf_ctx* ctx, mnmf_allreduce_fn fn, void* user, int n_ranks) {
  if (!ctx) return MNMF_EINVAL;
  ctx->ar = fn;
  ctx->ar_user = user;
  ctx->n_ranks = n_ranks > 0 ? n_ranks : 1;
  return ctx->impl->hook_changed();
}

int mnmf_set_comm(mnmf_ctx* ctx, mnmf_comm_fn fn, void* user, int rank, int n_ranks) {
  if (!ctx) return MNMF_EINVAL;
  if (fn && (n_ranks < 1 || rank < 0 || rank >= n_ranks)) {
    ctx->fail(MNMF_EINVAL, "set_comm: rank must be in [0, n_ranks)");
    return MNMF_EINVAL;
  }
  ctx->comm = fn;
  ctx->comm_user = user;
  ctx->rank = fn ? rank : 0;
  ctx->n_ranks = fn ? n_ranks : 1;
  return ctx->impl->hook_changed();
}

int mnmf_set_matrix(mnmf_ctx* ctx, int64_t m, int64_t V, int64_t nnz, const int64_t* rowptr, const int32_t* colidx,
                    const void* vals, int on_device) {
  if (!ctx) return MNMF_EINVAL;
  return ctx->impl->set_matrix(m, V, nnz, rowptr, colidx, vals, on_device);
}

int mnmf_run_nmf(mnmf_ctx* ctx, int k, void* W, void* H, const mnmf_opts* opts, mnmf_stats* stats) {
  if (!ctx || !W || !H) return MNMF_EINVAL;
  return ctx->impl->run_nmf(k, W, H, opts, stats);
}
int mnmf_run_nmf_sweep(mnmf_ctx* ctx, int n, const int32_t* ks, void* const* W, void* const* H, const mnmf_opts* opts,
                       mnmf_stats* stats) {
  if (!ctx) return MNMF_EINVAL;
  return ctx->impl->run_nmf_sweep(n, ks, W, H, opts, stats);
}
int mnmf_run_pnmf(mnmf_ctx* ctx, int k, void* W, void* H, const mnmf_opts* opts, mnmf_stats* stats) {
  if (!ctx || !W || !H) return MNMF_EINVAL;
  return ctx->impl->run_pnmf(k, W, H, opts, stats);
}
int mnmf_run_nmtf(mnmf_ctx* ctx, int k, void* W, void* S, void* H, const mnmf_opts* opts, mnmf_stats* stats) {
  if (!ctx || !W || !S || !H) return MNMF_EINVAL;
  return ctx->impl->run_nmtf(k, W, S, H, opts, stats);
}
int mnmf_spmm(mnmf_ctx* ctx, int transpose, int k, const void* B, void* out) {
  if (!ctx || !B || !out) return MNMF_EINVAL;
  return ctx->impl->k_spmm(transpose, k, B, out);
}
int mnmf_gram(mnmf_ctx* ctx, int64_t rows, int k, const void* A, const void* B, void* out) {
  if (!ctx || !A || !out) return MNMF_EINVAL;
  return ctx->impl->k_gram(rows, k, A, B, out);
}
int mnmf_update(mnmf_ctx* ctx, int64_t rows, int k, const void* A, const void* num, const void* G, const void* G2,
                double eps, int sqrt_mode, void* out) {
  if (!ctx || !A || !num || !G || !out) return MNMF_EINVAL;
  return ctx->impl->k_update(rows, k, A, num, G, G2, eps, sqrt_mode, out);
}
int mnmf_diff_norm(mnmf_ctx* ctx, int64_t rows, int k, const void* A, const void* B, int norm_kind, double* out) {
  if (!ctx || !A || !B) return MNMF_EINVAL;
  return ctx->impl->k_diff_norm(rows, k, A, B, norm_kind, out);
}

int mnmf_topk(mnmf_ctx* ctx, int64_t rows, int64_t cols, const void* A, int by_column, int n, int32_t* idx_out,
              void* val_out) {
  if (!ctx) return MNMF_EINVAL;
  return ctx->impl->k_topk(rows, cols, A, by_column, n, idx_out, val_out);
}

int mnmf_gram_static(mnmf_ctx* ctx, int64_t rows, int k, const void* A, const void* B, void* out) {
  if (!ctx) return MNMF_EINVAL;
  if (!A || !out || rows < 0 || k <= 0) {
    ctx->fail(MNMF_EINVAL, "gram_static: bad arguments");
    return MNMF_EINVAL;
  }
  if (k > kStaticGramMaxK) {
    ctx->fail(MNMF_EINVAL, "gram_static: k = " + std::to_string(k) + " exceeds the static scratch (k <= " +
                               std::to_string(kStaticGramMaxK) + "); use mnmf_gram");
    return MNMF_EINVAL;
  }
  void* slab = (ctx->device >= 0 && ctx->device < kMaxDevices) ? __atomic_load_n(&g_scratch[ctx->device], __ATOMIC_ACQUIRE) : nullptr;
  if (!slab) {
    ctx->fail(MNMF_ESTATE, "gram_static: no scratch address for this device (mnmf_create sets it up)");
    return MNMF_ESTATE;
  }
  if (hipSetDevice(ctx->device) != hipSuccess) {
    ctx->fail(MNMF_EHIP, "gram_static: hipSetDevice failed");
    return MNMF_EHIP;
  }
  const size_t esz = ctx->dtype == MNMF_F32 ? 4 : 8;
  if (rows == 0) {
    if (hipMemsetAsync(out, 0, (size_t)k * k * esz, ctx->stream) != hipSuccess) {
      ctx->fail(MNMF_EHIP, "gram_static: memset failed");
      return MNMF_EHIP;
    }
    return MNMF_OK;
  }
  const int rc = ctx->dtype == MNMF_F32 ? gram_static_launch<float>(ctx, slab, rows, k, A, B, out)
                                        : gram_static_launch<double>(ctx, slab, rows, k, A, B, out);
  if (rc != MNMF_OK) ctx->fail(rc, "gram_static: kernel launch failed");
  return rc;
}

int mnmf_s_confusion(mnmf_ctx* ctx, int k, const void* S, int32_t* col_out, int32_t* row_out, void* val_out) {
  if (!ctx) return MNMF_EINVAL;
  return ctx->impl->k_s_confusion(k, S, col_out, row_out, val_out);
}

int mnmf_memcpy_d2d(mnmf_ctx* ctx, void* dst, const void* src, int64_t bytes) {
  if (!ctx || !dst || !src || bytes < 0) return MNMF_EINVAL;
  if (hipMemcpyAsync(dst, src, (size_t)bytes, hipMemcpyDeviceToDevice, ctx->stream) != hipSuccess) {
    ctx->fail(MNMF_EHIP, "mnmf_memcpy_d2d failed");
    return MNMF_EHIP;
  }
  return MNMF_OK;
}

static void export_segments(const HostSegments& s, mnmf_segments* out) {
  auto dup64 = [](const std::vector<int64_t>& v) {
    int64_t* p = (int64_t*)std::malloc(std::max<size_t>(v.size(), 1) * 8);
    std::memcpy(p, v.data(), v.size() * 8);
    return p;
  };
  auto dup32 = [](const std::vector<int32_t>& v) {
    int32_t* p = (int32_t*)std::malloc(std::max<size_t>(v.size(), 1) * 4);
    std::memcpy(p, v.data(), v.size() * 4);
    return p;
  };
  out->n_seg = s.n_seg();
  out->seg_start = dup64(s.seg_start);
  out->seg_len = dup32(s.seg_len);
  out->seg_info = dup32(s.seg_info);
  out->n_dest = (int64_t)s.dest.size();
  out->dest = dup32(s.dest);
  out->piece_end = dup64(s.piece_end);
  out->n_empty = (int64_t)s.empty_rows.size();
  out->empty_rows = dup32(s.empty_rows);
  out->n_fix = (int64_t)s.fix_row.size();
  out->fix_row = dup32(s.fix_row);
  out->fix_first = dup32(s.fix_first);
  out->fix_count = dup32(s.fix_count);
  out->n_carry = s.n_carry;
}

int mnmf_build_segments(const int64_t* rowptr, int64_t rows, int64_t seg_nnz, int64_t sub_piece, mnmf_segments* out) {
  if (!rowptr || !out || rows < 0) return MNMF_EINVAL;
  HostPieces pc;
  pieces_from_rowptr(rowptr, rows, pc);
  HostSegments s;
  build_segments_host(pc, rows, seg_nnz, sub_piece, s);
  export_segments(s, out);
  return MNMF_OK;
}

int mnmf_build_segments_pieces(const int64_t* piece_start, const int32_t* piece_row, const int32_t* piece_block,
                               int64_t n_pieces, int64_t rows, int64_t seg_nnz, int64_t sub_piece, int xcd_group,
                               mnmf_segments* out) {
  return mnmf_build_segments_exact(piece_start, piece_row, piece_block, n_pieces, rows, seg_nnz, sub_piece, xcd_group, 0, out);
}

int mnmf_build_segments_exact(const int64_t* piece_start, const int32_t* piece_row, const int32_t* piece_block,
                              int64_t n_pieces, int64_t rows, int64_t seg_nnz, int64_t sub_piece, int xcd_group,
                              int exact, mnmf_segments* out) {
  if (!piece_start || !out || rows < 0 || n_pieces < 0 || (n_pieces > 0 && !piece_row)) return MNMF_EINVAL;
  HostPieces pc;
  pc.start.assign(piece_start, piece_start + n_pieces + 1);
  pc.row.assign(piece_row, piece_row + n_pieces);
  if (piece_block) pc.block.assign(piece_block, piece_block + n_pieces);
  for (int64_t i = 0; i < n_pieces; ++i)
    if (pc.row[i] < 0 || pc.row[i] >= rows || pc.start[i + 1] <= pc.start[i]) return MNMF_EINVAL;
  HostSegments s;
  build_segments_host(pc, rows, seg_nnz, sub_piece, s, piece_block ? xcd_group : 0, nullptr, 0, exact != 0);
  export_segments(s, out);
  return MNMF_OK;
}

int mnmf_build_segments_tiles(const int64_t* piece_start, const int32_t* piece_row, const int32_t* piece_tile,
                              int64_t n_pieces, int64_t rows, int64_t n_tiles, int64_t seg_nnz, int64_t sub_piece,
                              mnmf_segments* out, int32_t* tile_first) {
  if (!piece_start || !out || !tile_first || rows < 0 || n_pieces < 0 || n_tiles < 0 || (n_pieces > 0 && (!piece_row || !piece_tile)))
    return MNMF_EINVAL;
  HostPieces pc;
  pc.start.assign(piece_start, piece_start + n_pieces + 1);
  pc.row.assign(piece_row, piece_row + n_pieces);
  pc.block.assign(piece_tile, piece_tile + n_pieces);
  for (int64_t i = 0; i < n_pieces; ++i)
    if (pc.row[i] < 0 || pc.row[i] >= rows || pc.start[i + 1] <= pc.start[i] || pc.block[i] < 0 || pc.block[i] >= n_tiles ||
        (i > 0 && pc.block[i] < pc.block[i - 1]))
      return MNMF_EINVAL;
  HostSegments s;
  std::vector<int32_t> tf;
  if (seg_nnz <= 0) build_tile_segments_host(pc, rows, n_tiles, (int)std::max<int64_t>(1, -seg_nnz), sub_piece, s, tf);
  else build_segments_host(pc, rows, seg_nnz, sub_piece, s, 0, &tf, n_tiles);
  tf.resize((size_t)n_tiles + 1, (int32_t)s.n_seg());
  std::memcpy(tile_first, tf.data(), ((size_t)n_tiles + 1) * sizeof(int32_t));
  export_segments(s, out);
  return MNMF_OK;
}

void mnmf_free_segments(mnmf_segments* s) {
  if (!s) return;
  std::free(s->seg_start);
  std::free(s->seg_len);
  std::free(s->seg_info);
  std::free(s->dest);
  std::free(s->piece_end);
  std::free(s->empty_rows);
  std::free(s->fix_row);
  std::free(s->fix_first);
  std::free(s->fix_count);
  std::memset(s, 0, sizeof(*s));
}

double mnmf_sym_lambda_max(const double* a, int n) { return a ? sym_lambda_max_jacobi(a, n) : 0.0; }

}  // extern "C"
