"""Python handle on one mnmf context (one GPU).  torch supplies device memory, the stream and
torch.distributed; every numeric step runs in libmnmf.so (HIP)."""
from __future__ import annotations

import ctypes as C
import functools
from dataclasses import dataclass, field

import numpy as np
import torch

from . import _lib

_TORCH_DTYPE = {"fp32": torch.float32, "fp64": torch.float64}
_CODE = {"fp32": _lib.MNMF_F32, "fp64": _lib.MNMF_F64}
PHASES = ("gram_w", "spmm_xt", "h_side", "gram_h_kxk", "spmm_x", "w_update", "check", "unused")


class MnmfError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"{_lib.STATUS.get(code, code)}: {msg}")
        self.code = code


@dataclass
class Stats:
    iters: int = 0
    converged: bool = False
    w_norm: float = float("inf")
    h_norm: float = float("inf")
    s_norm: float = float("inf")
    ms_total: float = 0.0
    ms_kernels: float = 0.0
    bytes_model: float = 0.0
    flops_model: float = 0.0
    phase_ms: dict = field(default_factory=dict)


def _opts(max_iter=-1, fixed_iters=0, check_every=1, force_check=False, finalize=True, profile=False,
          norm_thresh=-1.0, eps=-1.0, zero_threshold=-1.0):
    o = _lib.MnmfOpts()
    o.max_iter = int(max_iter)
    o.fixed_iters = int(fixed_iters or 0)
    o.check_every = int(check_every)
    o.force_check = int(bool(force_check))
    o.finalize = int(bool(finalize))
    o.profile_phases = int(bool(profile))
    o.norm_thresh = float(norm_thresh)
    o.eps = float(eps)
    o.zero_threshold = float(zero_threshold)
    return o


def _engine_stream(fn):
    """Run a method with the engine's stream as torch's current one.  The library enqueues its kernels on the stream
    captured when the Engine was created; the torch side of a call (uploads, clones, output buffers, the collectives of
    the hook) must be ordered on the same stream, whatever stream the caller happens to be under.  The caller's stream
    waits for the engine's before the method returns."""
    @functools.wraps(fn)
    def wrapped(self, *a, **kw):
        cur = torch.cuda.current_stream(self.device)
        if cur.cuda_stream == self.stream.cuda_stream:
            return fn(self, *a, **kw)
        self.stream.wait_stream(cur)
        try:
            with torch.cuda.stream(self.stream):
                return fn(self, *a, **kw)
        finally:
            cur.wait_stream(self.stream)
    return wrapped


class Engine:
    """One resident matrix + solver workspaces on one MI355X."""

    def __init__(self, device=None, dtype="fp32"):
        if dtype not in _CODE:
            raise ValueError("dtype must be 'fp32' or 'fp64'")
        self._lib = _lib.load()
        if not torch.cuda.is_available():
            raise RuntimeError("manta_amd: no GPU visible (torch.cuda.is_available() is False); "
                               "the engine has no CPU fallback")
        if device is None:
            device = torch.cuda.current_device()
        self.device = torch.device("cuda", int(device) if not isinstance(device, torch.device) else device.index or 0)
        self.dtype = dtype
        self.tdtype = _TORCH_DTYPE[dtype]
        torch.cuda.set_device(self.device)
        self.stream = torch.cuda.current_stream(self.device)
        h = C.c_void_p()
        rc = self._lib.mnmf_create(C.byref(h), self.device.index, _CODE[dtype], C.c_void_p(self.stream.cuda_stream))
        if rc != 0:
            raise MnmfError(rc, "mnmf_create failed")
        self._h = h
        self.m = self.V = self.nnz = 0
        self._hook = None      # keep the ctypes callback alive
        self._stage = None
        self._resident_key = None   # identity of the scipy matrix run_nmf last uploaded (manta_amd.nmf)
        self._svd_cache = None      # leading right singular vectors of the resident matrix (manta_amd.init_device)

    # ------------------------------------------------------------------ plumbing
    def close(self):
        if getattr(self, "_h", None):
            self._lib.mnmf_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc):
        if rc != 0:
            raise MnmfError(rc, self._lib.mnmf_last_error(self._h).decode())

    def _dev(self, a):
        """numpy / torch -> contiguous device tensor of the engine dtype."""
        if isinstance(a, torch.Tensor):
            return a.to(device=self.device, dtype=self.tdtype).contiguous()
        return torch.from_numpy(np.ascontiguousarray(a)).to(device=self.device, dtype=self.tdtype).contiguous()

    @staticmethod
    def _host(t):
        return t.detach().to("cpu", torch.float64).numpy()

    def _stats(self, s):
        return Stats(iters=s.iters, converged=bool(s.converged), w_norm=s.w_norm, h_norm=s.h_norm, s_norm=s.s_norm,
                     ms_total=s.ms_total, ms_kernels=s.ms_kernels, bytes_model=s.bytes_model,
                     flops_model=s.flops_model, phase_ms={n: s.phase_ms[i] for i, n in enumerate(PHASES)})

    # ------------------------------------------------------------------ matrix
    @_engine_stream
    def set_matrix(self, X=None, *, rowptr=None, colidx=None, vals=None, shape=None):
        """``X``: a scipy sparse matrix (host), or give device/host tensors ``rowptr`` (int64), ``colidx``
        (int32), ``vals`` and ``shape``.  The engine copies and keeps X and X^T resident."""
        self._resident_key = None   # whoever replaces the matrix invalidates run_nmf's residency marker
        self._svd_cache = None      # ... and what was computed from the old matrix
        if X is not None:
            X = X.tocsr()
            X.sort_indices()
            shape = X.shape
            rowptr = torch.from_numpy(X.indptr.astype(np.int64))
            colidx = torch.from_numpy(X.indices.astype(np.int32))
            vals = torch.from_numpy(np.asarray(X.data))
        m, V = int(shape[0]), int(shape[1])
        rowptr = rowptr.to(device=self.device, dtype=torch.int64).contiguous()
        colidx = colidx.to(device=self.device, dtype=torch.int32).contiguous()
        vals = vals.to(device=self.device, dtype=self.tdtype).contiguous()
        nnz = int(colidx.numel())
        if rowptr.numel() != m + 1 or vals.numel() != nnz:
            raise ValueError("CSR arrays do not match the shape")
        self._check(self._lib.mnmf_set_matrix(self._h, m, V, nnz, rowptr.data_ptr(), colidx.data_ptr(),
                                              vals.data_ptr(), 1))
        self.m, self.V, self.nnz = m, V, nnz
        self._vals_minmax = (float(vals.min()), float(vals.max())) if nnz else (0.0, 0.0)

    def row_stride(self, k):
        """Elements per factor row in the engine's padded layout (mnmf_row_stride)."""
        return int(self._lib.mnmf_row_stride(self._h, int(k)))

    def value_range(self):
        """(X.min(), X.max()) as scipy defines them for a sparse matrix: implicit zeros count unless X is full."""
        lo, hi = self._vals_minmax
        if self.nnz < self.m * self.V:
            lo, hi = min(lo, 0.0), max(hi, 0.0)
        return lo, hi

    # ------------------------------------------------------------------ multi-GPU hook
    def _alias(self, ptr, count):
        """A torch tensor over `count` elements of the engine's own device memory at `ptr` (no copy): what the
        collectives of torch.distributed operate on in place."""
        class _DevMem:
            pass
        holder = _DevMem()
        holder.__cuda_array_interface__ = {"shape": (int(count),), "typestr": "<f4" if self.dtype == "fp32" else "<f8",
                                           "data": (int(ptr), False), "version": 2}
        return torch.as_tensor(holder, device=self.device)

    def set_process_group(self, group=None, collectives=True, single_rank_hook=False):
        """Install the collective hook over ``torch.distributed`` (nccl = RCCL over xGMI): all-reduce,
        reduce-scatter and all-gather run IN PLACE on tensors that alias the engine's exchange buffers, no staging
        copies.  With the gloo backend (tests: several ranks on one GPU) the same operations are staged through the
        host.  ``collectives=False`` installs the older all-reduce-only hook (all-reduce + replicated H update).
        A group of one rank needs no hook and gets none, unless ``single_rank_hook`` asks for it (together with
        ``MNMF_RS_AG=2`` the engine then walks the whole sharded schedule on one GPU: how the RCCL calls are rehearsed
        where no second GPU is at hand)."""
        import torch.distributed as dist

        if not dist.is_initialized() or (dist.get_world_size(group) == 1 and not single_rank_hook):
            self._lib.mnmf_set_comm(self._h, C.cast(None, _lib.COMM_FN), None, 0, 1)
            self._lib.mnmf_set_allreduce(self._h, C.cast(None, _lib.ALLREDUCE_FN), None, 1)
            self._hook = None
            return
        backend = dist.get_backend(group)
        world, rank = dist.get_world_size(group), dist.get_rank(group)
        eng = self
        SUM = dist.ReduceOp.SUM

        def run(op, ptr, count):
            n = count if op == _lib.COMM_ALLREDUCE else count * world
            t = eng._alias(ptr, n)
            mine = t[rank * count:(rank + 1) * count] if op != _lib.COMM_ALLREDUCE else None
            if backend == "nccl":
                if op == _lib.COMM_ALLREDUCE:
                    dist.all_reduce(t, op=SUM, group=group)
                elif op == _lib.COMM_REDUCE_SCATTER:
                    dist.reduce_scatter_tensor(mine, t, op=SUM, group=group)     # in place: recv = send + rank * count
                else:
                    dist.all_gather_into_tensor(t, mine, group=group)            # in place: send = recv + rank * count
                return
            # gloo: through the host (correctness rehearsal only)
            if op == _lib.COMM_ALLGATHER:
                parts = [torch.empty(count, dtype=eng.tdtype) for _ in range(world)]
                dist.all_gather(parts, mine.cpu(), group=group)
                t.copy_(torch.cat(parts))
                return
            host = t.cpu()
            dist.all_reduce(host, op=SUM, group=group)
            if op == _lib.COMM_ALLREDUCE:
                t.copy_(host)
            else:
                mine.copy_(host[rank * count:(rank + 1) * count])

        def on_engine_stream(stream_ptr):
            # include/mnmf.h: the collective is ordered against `stream`, the one the engine enqueues its kernels on --
            # not against whatever stream happens to be torch's current one when the solver is called
            sp = int(stream_ptr or 0)
            st = torch.cuda.ExternalStream(sp, device=eng.device) if sp else torch.cuda.default_stream(eng.device)
            return torch.cuda.stream(st)

        def comm_hook(_user, op, ptr, count, _dtype, _stream):
            try:
                with on_engine_stream(_stream):
                    run(int(op), ptr, int(count))
                return 0
            except Exception as e:  # never let an exception cross the C boundary
                print(f"[manta_amd] collective hook failed: {e!r}", flush=True)
                return 3

        def ar_hook(_user, ptr, count, _dtype, _stream):
            try:
                with on_engine_stream(_stream):
                    run(_lib.COMM_ALLREDUCE, ptr, int(count))
                return 0
            except Exception as e:
                print(f"[manta_amd] all-reduce hook failed: {e!r}", flush=True)
                return 3

        if collectives:
            self._hook = _lib.COMM_FN(comm_hook)
            self._lib.mnmf_set_allreduce(self._h, C.cast(None, _lib.ALLREDUCE_FN), None, 1)
            self._check(self._lib.mnmf_set_comm(self._h, self._hook, None, rank, world))
        else:
            self._hook = _lib.ALLREDUCE_FN(ar_hook)
            self._lib.mnmf_set_comm(self._h, C.cast(None, _lib.COMM_FN), None, 0, 1)
            self._check(self._lib.mnmf_set_allreduce(self._h, self._hook, None, world))

    # ------------------------------------------------------------------ solvers
    @_engine_stream
    def run_nmf(self, W0, H0, **kw):
        W, H = self._dev(W0).clone(), self._dev(H0).clone()
        k = W.shape[1]
        if W.shape != (self.m, k) or H.shape != (k, self.V):
            raise ValueError(f"W0 {tuple(W.shape)} / H0 {tuple(H.shape)} do not match X {self.m}x{self.V}")
        o, s = _opts(**kw), _lib.MnmfStats()
        self._check(self._lib.mnmf_run_nmf(self._h, k, W.data_ptr(), H.data_ptr(), C.byref(o), C.byref(s)))
        return W, H, self._stats(s)

    @_engine_stream
    def run_nmf_sweep(self, W0s, H0s, **kw):
        """Standard NMF for several topic counts in one pass over X per SpMM (mnmf_run_nmf_sweep): ``W0s[p]`` is
        m x k_p, ``H0s[p]`` is k_p x V, the k_p of one call add up to at most 32 (fp32) / 16 (fp64).  Returns a list of
        ``(W, H, stats)``, each bit-identical to ``run_nmf(W0s[p], H0s[p])`` on its own."""
        n = len(W0s)
        if n == 0 or n != len(H0s):
            raise ValueError("W0s and H0s must be equally long, non-empty lists")
        Ws = [self._dev(w).clone() for w in W0s]
        Hs = [self._dev(h).clone() for h in H0s]
        ks = [int(w.shape[1]) for w in Ws]
        for w, h, k in zip(Ws, Hs, ks):
            if w.shape != (self.m, k) or h.shape != (k, self.V):
                raise ValueError(f"W0 {tuple(w.shape)} / H0 {tuple(h.shape)} do not match X {self.m}x{self.V}")
        ks_c = (C.c_int32 * n)(*ks)
        wp = (C.c_void_p * n)(*[w.data_ptr() for w in Ws])
        hp = (C.c_void_p * n)(*[h.data_ptr() for h in Hs])
        st = (_lib.MnmfStats * n)()
        o = _opts(**kw)
        self._check(self._lib.mnmf_run_nmf_sweep(self._h, n, ks_c, wp, hp, C.byref(o), st))
        return [(Ws[p], Hs[p], self._stats(st[p])) for p in range(n)]

    @_engine_stream
    def run_pnmf(self, W0, **kw):
        W = self._dev(W0).clone()
        k = W.shape[1]
        if W.shape[0] != self.m:
            raise ValueError("W0 does not match X")
        H = torch.empty((k, self.V), dtype=self.tdtype, device=self.device)
        o, s = _opts(**kw), _lib.MnmfStats()
        self._check(self._lib.mnmf_run_pnmf(self._h, k, W.data_ptr(), H.data_ptr(), C.byref(o), C.byref(s)))
        return W, H, self._stats(s)

    @_engine_stream
    def run_nmtf(self, W0, S0, H0, **kw):
        W, S, H = self._dev(W0).clone(), self._dev(S0).clone(), self._dev(H0).clone()
        k = W.shape[1]
        if W.shape != (self.m, k) or H.shape != (k, self.V) or S.shape != (k, k):
            raise ValueError("W0/S0/H0 do not match X")
        o, s = _opts(**kw), _lib.MnmfStats()
        self._check(self._lib.mnmf_run_nmtf(self._h, k, W.data_ptr(), S.data_ptr(), H.data_ptr(), C.byref(o), C.byref(s)))
        return W, S, H, self._stats(s)

    # ------------------------------------------------------------------ kernel-level
    @_engine_stream
    def spmm(self, B, transpose=False):
        B = self._dev(B)
        k = B.shape[1]
        rows_in, rows_out = (self.m, self.V) if transpose else (self.V, self.m)
        if B.shape[0] != rows_in:
            raise ValueError("B has the wrong number of rows")
        out = torch.empty((rows_out, k), dtype=self.tdtype, device=self.device)
        self._check(self._lib.mnmf_spmm(self._h, int(bool(transpose)), k, B.data_ptr(), out.data_ptr()))
        return out

    @_engine_stream
    def gram(self, A, B=None):
        A = self._dev(A)
        rows, k = A.shape
        Bp = None
        if B is not None:
            B = self._dev(B)
            if B.shape != A.shape:
                raise ValueError("A and B must have the same shape")
            Bp = B.data_ptr()
        out = torch.empty((k, k), dtype=self.tdtype, device=self.device)
        self._check(self._lib.mnmf_gram(self._h, rows, k, A.data_ptr(), Bp, out.data_ptr()))
        return out

    @_engine_stream
    def update(self, A, num, G, G2=None, eps=1e-10, sqrt_mode=False):
        A, num, G = self._dev(A), self._dev(num), self._dev(G)
        rows, k = A.shape
        if num.shape != A.shape or G.shape != (k, k):
            raise ValueError("shape mismatch")
        G2p = None
        if G2 is not None:
            G2 = self._dev(G2)
            G2p = G2.data_ptr()
        out = torch.empty_like(A)
        self._check(self._lib.mnmf_update(self._h, rows, k, A.data_ptr(), num.data_ptr(), G.data_ptr(), G2p,
                                          float(eps), int(bool(sqrt_mode)), out.data_ptr()))
        return out

    @_engine_stream
    def topk(self, A, n, by_column=False):
        """The ``n`` largest entries of every row (or column) of ``A``, descending, ties: larger index first.
        Returns (indices int32 [groups, n], values [groups, n]) as device tensors."""
        A = self._dev(A)
        groups = A.shape[1] if by_column else A.shape[0]
        idx = torch.empty((groups, int(n)), dtype=torch.int32, device=self.device)
        val = torch.empty((groups, int(n)), dtype=self.tdtype, device=self.device)
        self._check(self._lib.mnmf_topk(self._h, A.shape[0], A.shape[1], A.data_ptr(), 1 if by_column else 0, int(n),
                                        idx.data_ptr(), val.data_ptr()))
        return idx, val

    def topic_rankings(self, W, H, S=None, top_n=15):
        """Ordered word ids and document ids per topic from factors on the device (topic_extractor.py:265-274 for
        NMTF -- topic t reads the H row ``argmax(S[t, :])``, :7-23 -- and :294-300 otherwise).  Returns two int32
        device tensors (topics x top_n)."""
        W, H = self._dev(W), self._dev(H)
        words, _ = self.topk(H, top_n, by_column=False)
        docs, _ = self.topk(W, top_n, by_column=True)
        if S is not None:
            rows = torch.argmax(self._dev(S), dim=1)
            words = words[rows]
        return words, docs

    @_engine_stream
    def s_matrix_confusion(self, S, console=None):
        """The NMTF topic-relationship summary of ``s_matrix_confusion`` (nmtf/nmtf.py:62-80) computed where S lives:
        per column of S its strongest row (``np.argmax``) and that strength, strongest coupling first.  Returns
        ``(columns, rows, strengths)`` as numpy arrays in rank order; with ``console`` (anything with the reference's
        ``print_debug(msg, tag=...)``) it also emits the reference's log lines -- which number the couplings by RANK,
        not by column ("Topic {rank} has connection with Topic {row}")."""
        S = self._dev(S)
        k = S.shape[0]
        if S.shape != (k, k):
            raise ValueError("S must be square")
        col = torch.full((k,), -1, dtype=torch.int32, device=self.device)
        row = torch.full((k,), -1, dtype=torch.int32, device=self.device)
        val = torch.full((k,), float("nan"), dtype=self.tdtype, device=self.device)
        self._check(self._lib.mnmf_s_confusion(self._h, k, S.data_ptr(), col.data_ptr(), row.data_ptr(), val.data_ptr()))
        cols, rows, vals = col.cpu().numpy(), row.cpu().numpy(), val.cpu().double().numpy()
        if console is not None:
            for line in confusion_lines(rows, vals):
                console.print_debug(line, tag="S-MATRIX")
        return cols, rows, vals

    def relevance_rankings(self, W, H, S=None, lambda_val=0.6, top_n=15):
        """LDAvis-style relevance ranking of the coherence step evaluated where the factors live
        (gensim_coherence.py:40-140): relevance = lambda * log p(word|topic) + (1 - lambda) * log lift, words with topic
        weight <= 1e-10 excluded.  Returns (word ids int32 [topics, top_n], -1 past the valid ones; relevance float64).
        The arithmetic is float64 on the device whatever the engine's dtype; ties rank the larger id first."""
        def f64(a):                                    # not through the engine's dtype: float64 inputs stay float64
            return (a if isinstance(a, torch.Tensor) else torch.from_numpy(np.asarray(a))).to(self.device, torch.float64)
        W, H = f64(W), f64(H)
        topic_weights = W.sum(dim=0)
        term_frequency = (H * topic_weights[:, None]).sum(dim=0)
        overall = term_frequency / (term_frequency.sum() + 1e-10)
        if S is not None:
            H = H[torch.argmax(f64(S), dim=1)]
        p = H / (H.sum(dim=1, keepdim=True) + 1e-10)
        lift = torch.clamp(p / (overall + 1e-10), min=1e-10)
        rel = lambda_val * torch.log(p + 1e-10) + (1 - lambda_val) * torch.log(lift)
        masked = torch.where(H > 1e-10, rel, torch.full_like(rel, float("-inf")))
        eng = self
        if self.dtype != "fp64":                       # rank in float64: a float32 copy would merge close scores
            from .nmf import _engine
            eng = _engine("fp64", self.device)
        idx, val = eng.topk(masked.contiguous(), top_n)
        return torch.where(torch.isinf(val), torch.full_like(idx, -1), idx), val

    @_engine_stream
    def diff_norm(self, A, B, kind="spectral"):
        A, B = self._dev(A), self._dev(B)
        if A.shape != B.shape:
            raise ValueError("shape mismatch")
        out = C.c_double()
        self._check(self._lib.mnmf_diff_norm(self._h, A.shape[0], A.shape[1], A.data_ptr(), B.data_ptr(),
                                             0 if kind == "spectral" else 1, C.byref(out)))
        return out.value


def confusion_lines(rows, strengths):
    """The log lines of the reference's s_matrix_confusion (nmtf.py:77-78) for couplings in rank order."""
    return [f"Topic {r} has connection with Topic {rows[r]}, strength: {strengths[r]}" for r in range(len(rows))]


def build_segments(rowptr=None, seg_nnz=64, sub_piece=512, *, piece_start=None, piece_row=None, piece_block=None,
                   rows=None, xcd_group=0, n_tiles=None, exact=False):
    """Host-only: the segment description the engine builds (no GPU needed; used by CPU tests).
    Give a CSR ``rowptr`` (one piece per non-empty row) or explicit pieces in processing order.
    With ``n_tiles`` the pieces describe the tile stream of X^T (``piece_block`` = tile of each piece) and the
    result also holds ``tile_first``; ``seg_nnz <= 0`` then means -seg_nnz equal segments per tile.  ``exact`` (explicit
    pieces only): segments of exactly ``seg_nnz`` nnz, pieces cut at the boundary."""
    lib = _lib.load()
    s = _lib.MnmfSegments()
    tile_first = None
    if n_tiles is not None:
        ps = np.ascontiguousarray(piece_start, dtype=np.int64)
        pr = np.ascontiguousarray(piece_row, dtype=np.int32)
        pb = np.ascontiguousarray(piece_block, dtype=np.int32)
        tile_first = np.zeros(int(n_tiles) + 1, np.int32)
        rc = lib.mnmf_build_segments_tiles(ps.ctypes.data, pr.ctypes.data, pb.ctypes.data, len(pr), int(rows), int(n_tiles),
                                           int(seg_nnz), int(sub_piece), C.byref(s), tile_first.ctypes.data)
    elif rowptr is not None:
        rp = np.ascontiguousarray(rowptr, dtype=np.int64)
        rc = lib.mnmf_build_segments(rp.ctypes.data, len(rp) - 1, int(seg_nnz), int(sub_piece), C.byref(s))
    else:
        ps = np.ascontiguousarray(piece_start, dtype=np.int64)
        pr = np.ascontiguousarray(piece_row, dtype=np.int32)
        pb = None if piece_block is None else np.ascontiguousarray(piece_block, dtype=np.int32)
        rc = lib.mnmf_build_segments_exact(ps.ctypes.data, pr.ctypes.data, None if pb is None else pb.ctypes.data,
                                           len(pr), int(rows), int(seg_nnz), int(sub_piece), int(xcd_group),
                                           int(bool(exact)), C.byref(s))
    if rc != 0:
        raise MnmfError(rc, "mnmf_build_segments")
    try:
        def arr(p, n, dt):
            return np.ctypeslib.as_array(p, shape=(max(int(n), 1),))[: int(n)].astype(dt).copy()
        extra = {} if tile_first is None else {"tile_first": tile_first}
        return dict(**extra, seg_start=arr(s.seg_start, s.n_seg, np.int64), seg_len=arr(s.seg_len, s.n_seg, np.int32),
                    seg_info=arr(s.seg_info, s.n_seg, np.int32),
                    dest=arr(s.dest, s.n_dest, np.int32), piece_end=arr(s.piece_end, s.n_dest, np.int64),
                    empty_rows=arr(s.empty_rows, s.n_empty, np.int32),
                    fix_row=arr(s.fix_row, s.n_fix, np.int32), fix_first=arr(s.fix_first, s.n_fix, np.int32),
                    fix_count=arr(s.fix_count, s.n_fix, np.int32), n_carry=int(s.n_carry))
    finally:
        lib.mnmf_free_segments(C.byref(s))
