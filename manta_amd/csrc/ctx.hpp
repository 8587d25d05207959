// The context object behind the C-ABI handle (include/mnmf.h); shared by the translation units of libmnmf.so.
#pragma once
#include <hip/hip_runtime.h>
#include <atomic>
#include <string>

#include "../../include/mnmf.h"

struct ImplBase;  // the typed engine (engine.hip)

// Every device / pinned allocation the library makes is counted here (mnmf_alloc_count): how a test can see that an
// entry point which promises to allocate nothing keeps the promise.
inline std::atomic<long long> g_mnmf_allocs{0};
inline hipError_t mnmf_counted_malloc(void** p, size_t bytes) {
  g_mnmf_allocs.fetch_add(1, std::memory_order_relaxed);
  return hipMalloc(p, bytes);
}

struct mnmf_ctx {
  int device = 0;
  int dtype = MNMF_F32;
  hipStream_t stream = nullptr;
  std::string err;
  mnmf_allreduce_fn ar = nullptr;   // all-reduce only (mnmf_set_allreduce)
  void* ar_user = nullptr;
  mnmf_comm_fn comm = nullptr;      // all-reduce / reduce-scatter / all-gather (mnmf_set_comm); wins over `ar`
  void* comm_user = nullptr;
  int rank = 0;
  int n_ranks = 1;
  bool sharded() const { return ar != nullptr || comm != nullptr; }
  ImplBase* impl = nullptr;
  void fail(int, const std::string& msg) { err = msg; }
};
