"""ctypes binding of include/mnmf.h.  Loading fails loudly: there is no CPU fallback."""
from __future__ import annotations

import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MNMF_LIB") or os.path.join(HERE, "libmnmf.so")

MNMF_F32, MNMF_F64 = 0, 1
STATUS = {0: "MNMF_OK", -1: "MNMF_EINVAL", -2: "MNMF_EHIP", -3: "MNMF_ECOMM", -4: "MNMF_ENAN", -5: "MNMF_ESTATE"}


class MnmfOpts(C.Structure):
    _fields_ = [("max_iter", C.c_int32), ("fixed_iters", C.c_int32), ("check_every", C.c_int32),
                ("force_check", C.c_int32), ("finalize", C.c_int32), ("profile_phases", C.c_int32),
                ("norm_thresh", C.c_double), ("eps", C.c_double), ("zero_threshold", C.c_double)]


class MnmfStats(C.Structure):
    _fields_ = [("iters", C.c_int32), ("converged", C.c_int32), ("w_norm", C.c_double), ("h_norm", C.c_double),
                ("s_norm", C.c_double), ("ms_total", C.c_double), ("ms_kernels", C.c_double),
                ("bytes_model", C.c_double), ("flops_model", C.c_double), ("phase_ms", C.c_double * 8)]


class MnmfSegments(C.Structure):
    _fields_ = [("n_seg", C.c_int64), ("seg_start", C.POINTER(C.c_int64)), ("seg_len", C.POINTER(C.c_int32)),
                ("seg_info", C.POINTER(C.c_int32)),
                ("n_dest", C.c_int64), ("dest", C.POINTER(C.c_int32)), ("piece_end", C.POINTER(C.c_int64)),
                ("n_empty", C.c_int64), ("empty_rows", C.POINTER(C.c_int32)),
                ("n_fix", C.c_int64), ("fix_row", C.POINTER(C.c_int32)), ("fix_first", C.POINTER(C.c_int32)),
                ("fix_count", C.POINTER(C.c_int32)), ("n_carry", C.c_int64)]


ALLREDUCE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_void_p)
COMM_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int64, C.c_int, C.c_void_p)
COMM_ALLREDUCE, COMM_REDUCE_SCATTER, COMM_ALLGATHER = 0, 1, 2

# every symbol include/mnmf.h declares: (restype, argtypes)
SIGNATURES = {
    "mnmf_create": (C.c_int, [C.POINTER(C.c_void_p), C.c_int, C.c_int, C.c_void_p]),
    "mnmf_destroy": (None, [C.c_void_p]),
    "mnmf_last_error": (C.c_char_p, [C.c_void_p]),
    "mnmf_version": (C.c_char_p, []),
    "mnmf_alloc_count": (C.c_int64, []),
    "mnmf_row_stride": (C.c_int, [C.c_void_p, C.c_int]),
    "mnmf_set_allreduce": (C.c_int, [C.c_void_p, ALLREDUCE_FN, C.c_void_p, C.c_int]),
    "mnmf_set_comm": (C.c_int, [C.c_void_p, COMM_FN, C.c_void_p, C.c_int, C.c_int]),
    "mnmf_set_matrix": (C.c_int, [C.c_void_p, C.c_int64, C.c_int64, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]),
    "mnmf_run_nmf": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.POINTER(MnmfOpts), C.POINTER(MnmfStats)]),
    "mnmf_run_nmf_sweep": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(MnmfOpts), C.c_void_p]),
    "mnmf_run_pnmf": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.POINTER(MnmfOpts), C.POINTER(MnmfStats)]),
    "mnmf_run_nmtf": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(MnmfOpts), C.POINTER(MnmfStats)]),
    "mnmf_spmm": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "mnmf_gram": (C.c_int, [C.c_void_p, C.c_int64, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
    "mnmf_gram_static": (C.c_int, [C.c_void_p, C.c_int64, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
    "mnmf_update": (C.c_int, [C.c_void_p, C.c_int64, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_double, C.c_int, C.c_void_p]),
    "mnmf_diff_norm": (C.c_int, [C.c_void_p, C.c_int64, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.POINTER(C.c_double)]),
    "mnmf_topk": (C.c_int, [C.c_void_p, C.c_int64, C.c_int64, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "mnmf_s_confusion": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "mnmf_tfidf_count_terms": (C.c_int, [C.c_void_p, C.c_int64, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                         C.c_void_p]),
    "mnmf_tfidf_fill_terms": (C.c_int, [C.c_void_p, C.c_int64, C.c_int64, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p,
                                        C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "mnmf_tfidf_scale": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                   C.c_void_p]),
    "mnmf_memcpy_d2d": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64]),
    "mnmf_build_segments": (C.c_int, [C.c_void_p, C.c_int64, C.c_int64, C.c_int64, C.POINTER(MnmfSegments)]),
    "mnmf_build_segments_pieces": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_int64,
                                             C.c_int64, C.c_int, C.POINTER(MnmfSegments)]),
    "mnmf_build_segments_exact": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_int64,
                                            C.c_int64, C.c_int, C.c_int, C.POINTER(MnmfSegments)]),
    "mnmf_build_segments_tiles": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_int64,
                                            C.c_int64, C.c_int64, C.POINTER(MnmfSegments), C.c_void_p]),
    "mnmf_free_segments": (None, [C.POINTER(MnmfSegments)]),
    "mnmf_sym_lambda_max": (C.c_double, [C.c_void_p, C.c_int]),
}

_lib = None


def load():
    """Load libmnmf.so (built by ``python -m manta_amd.build`` / ``__graft_entry__.build()``)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"manta_amd: HIP library missing at {LIB_PATH}. Build it with `python -m manta_amd.build` "
            "(hipcc --offload-arch=gfx950). There is no CPU fallback.")
    import torch  # noqa: F401  -- load torch's HIP runtime first so both share one libamdhip64
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is not exported
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib
