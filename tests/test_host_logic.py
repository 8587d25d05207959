"""CPU-only checks: the C-ABI library loads and exports every declared symbol, and the host-side
logic (segment builder, eigen-solver, drop-in argument handling) is right.  No GPU compute."""
import re
import os

import numpy as np
import pytest
import scipy.sparse as sp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    from manta_amd import _lib

    lib = _lib.load()
    header = open(os.path.join(ROOT, "include", "mnmf.h")).read()
    declared = set(re.findall(r"\b(mnmf_[a-z0-9_]+)\s*\(", header))
    declared -= {"mnmf_allreduce_fn", "mnmf_comm_fn"}
    assert declared, "no declarations found"
    for name in sorted(declared):
        assert hasattr(lib, name), f"{name} declared in mnmf.h but not exported"
        assert name in _lib.SIGNATURES, f"{name} has no ctypes signature"
    assert lib.mnmf_version().startswith(b"mnmf")


def _walk(seg, nnz):
    """Python emulation of the stream kernel: which destination every nnz is accumulated into.
    Returns dest_of_nnz (>=0 row, <0 ~carry slot)."""
    endflag = np.zeros(nnz, bool)
    endflag[seg["piece_end"]] = True
    out = np.full(nnz, np.iinfo(np.int64).min, np.int64)
    for s in range(len(seg["seg_info"])):
        a, b = seg["seg_start"][s], seg["seg_start"][s] + seg["seg_len"][s]
        j = seg["seg_info"][s]
        start = a
        for p in range(a, b):
            if endflag[p]:
                out[start:p + 1] = seg["dest"][j]
                j += 1
                start = p + 1
        assert start == b, "a segment must end on a piece boundary"
    return out


def _check_cover(seg, truth_rows, n_rows, sub):
    """Every nnz lands in its row, directly or through that row's carry slots (in slot order)."""
    nnz = len(truth_rows)
    d = _walk(seg, nnz)
    direct = d >= 0
    assert np.array_equal(d[direct], truth_rows[direct])
    slot_row = np.full(seg["n_carry"], -1, np.int64)
    for r, first, cnt in zip(seg["fix_row"], seg["fix_first"], seg["fix_count"]):
        assert cnt >= 2
        slot_row[first:first + cnt] = r
    assert (slot_row >= 0).all()
    slots = ~d[~direct]
    assert np.array_equal(slot_row[slots], truth_rows[~direct])
    assert sorted(set(slots.tolist())) == list(range(seg["n_carry"])), "every carry slot is written"
    # a row is either direct (one piece) or carried, never both
    assert not set(truth_rows[direct].tolist()) & set(seg["fix_row"].tolist())
    # sub-pieces are never longer than `sub`
    ends = np.sort(seg["piece_end"])
    assert np.all(np.diff(np.concatenate([[-1], ends])) <= sub)
    present = np.zeros(n_rows, bool)
    present[truth_rows] = True
    assert sorted(seg["empty_rows"].tolist()) == np.nonzero(~present)[0].tolist()


@pytest.mark.parametrize("seg_nnz", [1, 32, 64, 512])
def test_segments_cover_every_nnz_once(seg_nnz):
    from manta_amd.engine import build_segments

    rng = np.random.RandomState(seg_nnz)
    lens = rng.poisson(20, 400)
    lens[[0, 5, 6, 399]] = 0                 # empty rows incl. first and last
    lens[[10, 200]] = [2000, 513]            # long rows
    lens[11] = 512                           # exactly at the threshold: one piece
    rowptr = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    seg = build_segments(rowptr, seg_nnz, 512)
    assert seg["seg_start"][0] == 0 and seg["seg_len"].sum() == rowptr[-1]
    assert np.all(np.diff(seg["seg_start"]) > 0)
    assert sorted(seg["fix_row"].tolist()) == [10, 200]
    _check_cover(seg, np.repeat(np.arange(400), lens), 400, 512)


def test_segments_blocked_pieces():
    """Document-blocked X^T: a vocabulary row appears as several pieces (one per block), interleaved
    with pieces of other rows; short rows come last as single pieces."""
    from manta_amd.engine import build_segments

    rng = np.random.RandomState(7)
    n_rows, n_blocks = 50, 6
    n_blocks = 19
    rows, lens, blocks = [], [], []
    for b in range(n_blocks):
        for r in range(0, 30):               # "long" rows: present in most blocks
            if rng.rand() < 0.8:
                rows.append(r)
                lens.append(int(rng.randint(1, 1400)))
                blocks.append(b)
    for r in range(30, 48):                  # short rows: one piece each; rows 48, 49 empty
        rows.append(r)
        lens.append(int(rng.randint(1, 40)))
        blocks.append(-1)
    start = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    for q, group in ((32, 0), (512, 0), (64, 16), (512, 4)):
        seg = build_segments(piece_start=start, piece_row=np.array(rows), piece_block=np.array(blocks), rows=n_rows,
                             seg_nnz=q, sub_piece=512, xcd_group=group)
        _check_cover(seg, np.repeat(np.array(rows), lens), n_rows, 512)
        if group:
            # XCD-affine order: workgroup g holds `group` segments, all of blocks congruent to g mod 8
            nb_of_nnz = np.repeat(np.array(blocks), lens)
            assert len(seg["seg_len"]) % (8 * group) == 0
            for s_i in range(len(seg["seg_len"])):
                if seg["seg_len"][s_i] == 0:
                    continue
                blk = nb_of_nnz[seg["seg_start"][s_i]]
                if blk >= 0:
                    assert blk % 8 == (s_i // group) % 8
            # inside one XCD's list the blocks come in ascending order
            for x in range(8):
                ids = [i for i in range(len(seg["seg_len"])) if (i // group) % 8 == x and seg["seg_len"][i] > 0]
                bl = [nb_of_nnz[seg["seg_start"][i]] for i in ids]
                bl = [b for b in bl if b >= 0]
                assert bl == sorted(bl)
        # carry slots of a row are numbered in processing order
        d = _walk(seg, int(start[-1]))
        for r, first, cnt in zip(seg["fix_row"], seg["fix_first"], seg["fix_count"]):
            sl = ~d[(np.repeat(np.array(rows), lens) == r)]
            assert np.all(np.diff(sl) >= 0) and sl.min() == first and sl.max() == first + cnt - 1


@pytest.mark.parametrize("seg_nnz", [96, 17, -4, -64])
def test_segments_tile_stream(seg_nnz):
    """Tile stream of X^T: every hot word has one piece per tile; segments never cross a tile, tile_first indexes
    them, equal-split mode cuts every tile into the requested number of segments."""
    from manta_amd.engine import build_segments

    rng = np.random.RandomState(11)
    n_hot, n_tiles = 12, 9
    rows, lens, tiles = [], [], []
    for t in range(n_tiles):
        if t == 4:
            continue                          # a tile without any hot nnz
        for r in range(n_hot):
            if r == 7 and t != 2:
                continue                      # a word present in one tile only: a direct row, no carries
            if rng.rand() < 0.9:
                rows.append(r)
                lens.append(int(rng.randint(1, 900)))
                tiles.append(t)
    start = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    truth = np.repeat(np.array(rows), lens)
    tile_of_nnz = np.repeat(np.array(tiles), lens)
    seg = build_segments(piece_start=start, piece_row=np.array(rows), piece_block=np.array(tiles), rows=40,
                         seg_nnz=seg_nnz, sub_piece=512, n_tiles=n_tiles)
    seg_plain = dict(seg, empty_rows=np.nonzero(~np.isin(np.arange(40), truth))[0].astype(np.int32))  # not reported
    assert len(seg["empty_rows"]) == 0
    _check_cover(seg_plain, truth, 40, 512)
    tf = seg["tile_first"]
    assert tf[0] == 0 and tf[-1] == len(seg["seg_len"]) and np.all(np.diff(tf) >= 0)
    assert tf[5] == tf[4], "an empty tile owns no segment"
    for t in range(n_tiles):
        for s_i in range(tf[t], tf[t + 1]):
            a, b = seg["seg_start"][s_i], seg["seg_start"][s_i] + seg["seg_len"][s_i]
            assert np.all(tile_of_nnz[a:b] == t)
        if seg_nnz <= 0 and tf[t + 1] > tf[t]:
            ln = seg["seg_len"][tf[t]:tf[t + 1]]
            assert len(ln) <= -seg_nnz and ln.max() - ln.min() <= ln.max() - ln[-1] and np.all(ln[:-1] == ln[0])
    assert seg["seg_len"].sum() == start[-1]


def test_segments_degenerate():
    from manta_amd.engine import build_segments

    seg = build_segments(np.zeros(6, np.int64), 64)      # all rows empty
    assert len(seg["seg_info"]) == 0 and len(seg["empty_rows"]) == 5
    seg = build_segments(np.array([0], np.int64), 64)    # no rows
    assert len(seg["seg_info"]) == 0


def test_lambda_max():
    from manta_amd import _lib

    lib = _lib.load()
    rng = np.random.RandomState(0)
    for n in (1, 3, 8, 52):
        d = np.abs(rng.randn(200, n))
        g = np.ascontiguousarray(d.T @ d)
        got = lib.mnmf_sym_lambda_max(g.ctypes.data, n)
        assert abs(np.sqrt(got) - np.linalg.norm(d, 2)) <= 1e-10 * np.linalg.norm(d, 2)


def test_run_nmf_argument_semantics_without_gpu():
    """Everything run_nmf decides before touching the GPU behaves like the reference
    (nmf_orchestrator.py:28-29, 105-106, 129-130)."""
    import manta_amd

    X = sp.random(30, 20, density=0.3, format="csr", random_state=1)
    with pytest.raises(ValueError, match="CSC"):
        manta_amd.find_max_rank(X)
    assert manta_amd.find_max_rank(X.tocsc()) == int(X.count_nonzero() / 50)
    calls = []

    def init(mat, rank):
        calls.append((sp.isspmatrix_csc(mat), rank))
        return np.ones((30, rank)), np.ones((rank, 20))

    with pytest.raises(ValueError, match="Unknown NMF method: bogus"):
        manta_amd.run_nmf(-1, X, init=init, nmf_method="bogus")
    assert calls == [(True, int(X.count_nonzero() / 50))]   # init ran first, on CSC, with the auto rank


def test_engine_refuses_to_run_without_gpu():
    import torch

    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from manta_amd.engine import Engine

    with pytest.raises(RuntimeError, match="no CPU fallback"):
        Engine()


def test_tfidf_flatten_documents_and_reduceat_lengths():
    """Host side of the TF-IDF builder (manta_amd/tfidf.py): ragged token lists -> flat arrays; the np.add.reduceat
    document lengths of tfidf_english_calculator.py:107 with their quirk for empty documents."""
    import torch
    from manta_amd import tfidf

    docs = [[3, 1, 3], [], [0], [], [], [2, 2, 2, 5]]
    tokens, ptr = tfidf.flatten_documents(docs)
    assert tokens.dtype == torch.int32 and ptr.dtype == torch.int64
    assert tokens.tolist() == [3, 1, 3, 0, 2, 2, 2, 5] and ptr.tolist() == [0, 3, 3, 4, 4, 4, 8]
    # count matrix rows: [1:1, 3:2], [], [0:1], [], [], [2:3, 5:1]
    rowptr = torch.tensor([0, 2, 2, 3, 3, 3, 5])
    counts = torch.tensor([1, 2, 1, 3, 1], dtype=torch.int32)
    lens = tfidf._reduceat_lengths(ptr, rowptr, counts)
    data = np.array([1, 2, 1, 3, 1])
    assert lens.tolist() == np.add.reduceat(data, rowptr[:-1].numpy()).tolist() == [3, 1, 1, 3, 3, 4]
    with pytest.raises(IndexError):                                   # empty last document
        tfidf._reduceat_lengths(torch.tensor([0, 2, 2]), torch.tensor([0, 1, 1]), torch.tensor([2], dtype=torch.int32))
    with pytest.raises(IndexError):
        tfidf.flatten_documents([[2 ** 40]])


def test_init_device_dense_helpers_on_cpu():
    """The tall-skinny helpers of manta_amd/init_device.py are plain torch: slab-split A^T.B equals the direct product,
    CholeskyQR2 gives an orthonormal basis of a well-conditioned block and the Householder fallback drops dependent
    columns of a rank-deficient one."""
    import torch
    from manta_amd.init_device import _orthonormal_columns, _tsgemm

    g = torch.Generator().manual_seed(0)
    A = torch.randn((10007, 13), generator=g, dtype=torch.float64)
    B = torch.randn((10007, 5), generator=g, dtype=torch.float64)
    assert torch.allclose(_tsgemm(A, B), A.t() @ B, rtol=1e-12, atol=1e-9)
    assert torch.allclose(_tsgemm(A[:100], B[:100]), A[:100].t() @ B[:100], rtol=1e-12, atol=1e-12)   # below one slab
    Q = _orthonormal_columns(A)
    assert Q.shape == A.shape and torch.allclose(Q.t() @ Q, torch.eye(13, dtype=torch.float64), atol=1e-12)
    assert torch.allclose(Q @ (Q.t() @ A), A, atol=1e-9)                      # same column space
    D = torch.cat([A[:, :4], A[:, :2] @ torch.randn((2, 3), generator=g, dtype=torch.float64)], dim=1)   # rank 4 of 7
    Qd = _orthonormal_columns(D)
    assert Qd.shape[1] == 4 and torch.allclose(Qd.t() @ Qd, torch.eye(4, dtype=torch.float64), atol=1e-10)
    assert _orthonormal_columns(D, drop=False) is None


def test_product_nndsvd_matches_reference_fixture(golden):
    """manta_amd.init (the default `init` of the product's run_nmf) against W0/H0 and the NMTF legacy starts that the
    unmodified reference produced under the fixture's numpy seed (nmf_initialization.py:38-83, nmtf_init.py:19-41):
    bit for bit, signed zeros aside (`==` does not tell them apart and the reference's own clean-up removes them)."""
    from manta_amd import init as pi

    g = golden("converged_D_600x500_k6_nmtf")
    np.random.seed(int(g["seed"]))
    w, h = pi.nmf_initialization_nndsvd(g["X"].tocsc(), int(g["k"]))
    assert w.flags.c_contiguous and h.flags.c_contiguous and w.dtype == np.float64
    assert np.array_equal(w, g["W0"]) and np.array_equal(h, g["H0"])
    wt, st, ht = pi.nmtf_initialization_nndsvd_legacy(g["X"].tocsr(), int(g["k"]))
    assert np.array_equal(wt, g["Wt0"]) and np.array_equal(st, g["St0"]) and np.array_equal(ht, g["Ht0"])
    assert not np.signbit(w[w == 0]).any() and not np.signbit(h[h == 0]).any()
    for name in ("nmf", "pnmf"):   # the same start is what the other two converged fixtures recorded
        g2 = golden(f"converged_D_600x500_k6_{name}")
        assert np.array_equal(w, g2["W0"]) and np.array_equal(h, g2["H0"])


def test_install_rebinds_run_nmf_where_manta_binds_it():
    """install() must replace `run_nmf` on every module that binds the name at import time
    (modeling_pipeline.py:10, optimization_pipeline.py:16, the nmf package and its orchestrator) and, with tfidf=True,
    the TF-IDF builders where they are bound (english_entry.py:10, turkish_entry.py:4).  MANTA itself is not installed
    here: stand-in modules with the same names and the same attributes are registered for the duration of the test."""
    import sys
    import types

    import manta_amd
    from manta_amd import tfidf as t

    sentinel = object()
    with_run = ["manta._functions.nmf.nmf_orchestrator", "manta._functions.nmf", "manta._functions",
                "manta.pipeline.modeling_pipeline", "manta.pipeline.optimization_pipeline"]
    with_tfidf = ["manta._functions.tfidf", "manta._functions", "manta._functions.english.english_entry",
                  "manta._functions.turkish.turkish_entry"]
    others = ["manta", "manta.pipeline", "manta._functions.english", "manta._functions.turkish", "manta.pipeline.text_pipeline"]
    saved = {n: sys.modules.get(n) for n in set(with_run + with_tfidf + others)}
    try:
        for n in saved:
            mod = types.ModuleType(n)
            mod.__path__ = []
            sys.modules[n] = mod
        for n in with_run:
            sys.modules[n].run_nmf = sentinel
        for n in with_tfidf:
            sys.modules[n].tf_idf_english = sentinel
        sys.modules["manta._functions.turkish.turkish_entry"].tf_idf_turkish = sentinel
        sys.modules["manta.pipeline.text_pipeline"].unrelated = sentinel
        patched = manta_amd.install()
        assert sorted(patched) == sorted(with_run)
        for n in with_run:
            assert sys.modules[n].run_nmf is manta_amd.run_nmf
        assert sys.modules["manta._functions.tfidf"].tf_idf_english is sentinel      # not asked for
        assert not hasattr(sys.modules["manta.pipeline"], "run_nmf")                   # nothing is added where nothing was bound
        # install(init="device"): the rebound function fills in the GPU start only when the caller passes no init
        manta_amd.install(init="device")
        bound = sys.modules["manta.pipeline.modeling_pipeline"].run_nmf
        assert bound is not manta_amd.run_nmf and bound.__wrapped__ is manta_amd.run_nmf
        seen = {}
        real = bound.__globals__["run_nmf"]
        try:
            bound.__globals__["run_nmf"] = lambda *a, **kw: seen.update(kw) or "ok"
            assert bound(5, None, norm_thresh=0.005, nmf_method="nmf") == "ok" and seen["init"] == "device"
            seen.clear()
            assert bound(5, None, init=len) == "ok" and seen["init"] is len
        finally:
            bound.__globals__["run_nmf"] = real
        with pytest.raises(ValueError):
            manta_amd.install(init="host")
        manta_amd.install()
        patched = manta_amd.install(tfidf=True)
        for n in with_tfidf:
            assert sys.modules[n].tf_idf_english is t.tf_idf_english and f"{n}.tf_idf_english" in patched
        assert sys.modules["manta._functions.turkish.turkish_entry"].tf_idf_turkish is t.tf_idf_turkish
        assert not hasattr(sys.modules["manta._functions.tfidf"], "tf_idf_turkish")
        assert sys.modules["manta.pipeline.text_pipeline"].unrelated is sentinel
    finally:
        for n, mod in saved.items():
            if mod is None:
                sys.modules.pop(n, None)
            else:
                sys.modules[n] = mod


def test_nmtf_needs_all_three_explicit_starts():
    """W0/H0 without S0 used to be dropped silently for nmtf; now it is an error before anything touches the GPU."""
    import manta_amd

    X = sp.random(30, 20, density=0.3, format="csr", random_state=0)
    W0, H0 = np.ones((30, 3)), np.ones((3, 20))
    with pytest.raises(ValueError, match="W0, S0 and H0"):
        manta_amd.run_nmf(3, X, nmf_method="nmtf", W0=W0, H0=H0)


def test_cached_engine_key_is_the_gpu_not_the_spelling():
    from manta_amd import nmf

    import torch
    assert nmf._device_index(0) == nmf._device_index("cuda:0") == nmf._device_index(torch.device("cuda", 0)) == 0
    assert nmf._device_index(torch.device("cuda", 3)) == 3
    assert isinstance(nmf._device_index(None), int)



@pytest.mark.parametrize("q,group", [(32, 0), (128, 0), (64, 16)])
def test_segments_exact_length(q, group):
    """Exact-length segments (small matrices): every segment holds exactly q nnz unless it ends a block or the stream;
    cut pieces are summed from carry slots, everything still lands in its row."""
    from manta_amd.engine import build_segments

    rng = np.random.RandomState(q + group)
    n_rows, n_blocks = 60, 11
    rows, lens, blocks = [], [], []
    for b in range(n_blocks):
        for r in range(40):
            if rng.rand() < 0.7:
                rows.append(r)
                lens.append(int(rng.randint(1, 700)))
                blocks.append(b)
    for r in range(40, 58):
        rows.append(r)
        lens.append(int(rng.randint(1, 90)))
        blocks.append(-1)
    start = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    nnz_block = np.repeat(np.array(blocks), lens)
    seg = build_segments(piece_start=start, piece_row=np.array(rows), piece_block=np.array(blocks), rows=n_rows,
                         seg_nnz=q, sub_piece=512, xcd_group=group, exact=True)
    _check_cover(seg, np.repeat(np.array(rows), lens), n_rows, 512)
    for a, n in zip(seg["seg_start"], seg["seg_len"]):
        if n == 0:
            continue
        assert n <= q
        end = a + n
        if n < q:   # short only where the block changes or the stream ends (blocks matter only in the XCD-affine mode)
            assert end == start[-1] or (group and nnz_block[end] != nnz_block[end - 1])
    assert int(seg["seg_len"].sum()) == int(start[-1])
