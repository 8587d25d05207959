// The context object behind the C-ABI handle (include/mnmf.h); shared by the translation units of libmnmf.so.
#pragma once
#include <hip/hip_runtime.h>
#include <string>

#include "../../include/mnmf.h"

struct ImplBase;  // the typed engine (engine.hip)

struct mnmf_ctx {
  int device = 0;
  int dtype = MNMF_F32;
  hipStream_t stream = nullptr;
  std::string err;
  mnmf_allreduce_fn ar = nullptr;
  void* ar_user = nullptr;
  int n_ranks = 1;
  ImplBase* impl = nullptr;
  void fail(int, const std::string& msg) { err = msg; }
};
