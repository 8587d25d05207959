// The typed engine behind a context, as the C-ABI layer (capi.hip) sees it.  Engine<float> and Engine<double>
// (engine.hip) are compiled as two translation units, so the library builds in parallel.
#pragma once
#include <stdint.h>

#include "../../include/mnmf.h"

struct ImplBase {
  virtual ~ImplBase() {}
  virtual int set_matrix(int64_t m, int64_t V, int64_t nnz, const int64_t* rowptr, const int32_t* colidx,
                         const void* vals, int on_device) = 0;
  virtual int run_nmf(int k, void* W, void* H, const mnmf_opts* o, mnmf_stats* st) = 0;
  virtual int run_nmf_sweep(int n, const int32_t* ks, void* const* W, void* const* H, const mnmf_opts* o, mnmf_stats* st) = 0;
  virtual int run_pnmf(int k, void* W, void* H, const mnmf_opts* o, mnmf_stats* st) = 0;
  virtual int run_nmtf(int k, void* W, void* S, void* H, const mnmf_opts* o, mnmf_stats* st) = 0;
  virtual int k_spmm(int transpose, int k, const void* B, void* out) = 0;
  virtual int k_gram(int64_t rows, int k, const void* A, const void* B, void* out) = 0;
  virtual int k_update(int64_t rows, int k, const void* A, const void* num, const void* G, const void* G2,
                       double eps, int sqrt_mode, void* out) = 0;
  virtual int k_diff_norm(int64_t rows, int k, const void* A, const void* B, int kind, double* out) = 0;
  virtual int hook_changed() = 0;
  virtual int row_stride(int k) const = 0;
  virtual int k_s_confusion(int k, const void* S, int32_t* col, int32_t* row, void* val) = 0;
  virtual int k_topk(int64_t rows, int64_t cols, const void* A, int by_column, int n, int32_t* idx, void* val) = 0;
};

struct mnmf_ctx;
ImplBase* mnmf_make_engine_f32(mnmf_ctx* c);
ImplBase* mnmf_make_engine_f64(mnmf_ctx* c);
