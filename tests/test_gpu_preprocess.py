"""GPU: the format-generation pipeline (spmm_trace_formats) against the oracle in its stable
tie order and against the fixtures made from the compiled reference.  Every stage's output is
compared, bit for bit."""
import numpy as np
import pytest

from oracle import oracle as orc
from spmm_amd import capi, synth
from tests.helpers import case_input, golden_json, golden_npz, hx

pytestmark = pytest.mark.gpu

STAGES = ["seq_bitmap", "r_indptr", "r_indices", "r_data", "bsize_list", "panel_ptr", "panel_list", "spv8_list",
          "out_seq_all", "b_indptr", "b_colidx_orig", "b_data", "b_local", "keys", "seq_offset", "seq_input", "rseq"]


# The greedy block chain as the two-pointer sweep over all row starts plus the walk (default: chunks by the input's
# size), the same with one chunk (one wave sweeps every row) and with more chunks than rows (every chunk counts its own
# window and moves over at most one row), and as one launch per step of the chain.
CHAINS = {"sweep": {}, "sweep_one_chunk": {"SPMM_SWEEP_CHUNKS": "1"}, "sweep_chunk_per_row": {"SPMM_SWEEP_CHUNKS": "1048576"},
          "stepwise": {"SPMM_CHAIN_SWEEP": "0"},
          # the first-touch renumbering over the whole stream at once instead of block by block in LDS
          "renumber_whole_stream": {"SPMM_RENUMBER_LDS": "0"},
          # block by block in LDS, but from two stable sorts (by column, then by block) instead of one sort of a 32-bit
          # (block, column) key: the form for block and column numbers of more than 32 bits together
          "renumber_two_sorts": {"SPMM_RENUMBER_LDS": "2"}}


@pytest.fixture(scope="module", params=list(CHAINS))
def ctx(request):
    import os
    env = CHAINS[request.param]
    old = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    try:
        c = capi.Context(0)
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    yield c
    c.close()


def _compare(g, o):
    assert g["bnum"] == o["bnum"]
    for k in STAGES:
        assert g[k].shape == o[k].shape, (k, g[k].shape, o[k].shape)
        if not np.array_equal(g[k], o[k]):
            bad = np.flatnonzero(g[k] != o[k])
            raise AssertionError(f"stage output {k} differs at {bad.size} of {g[k].size} positions, first {bad[:5]}: "
                                 f"gpu {g[k][bad[:5]]} oracle {o[k][bad[:5]]}")


@pytest.mark.parametrize("name", ["tiny", "rmat12d", "rmat17d", "rmat18", "rmat20d"])
def test_pipeline_matches_golden(ctx, name):
    J = golden_json("pre_" + name)
    n, row, col = case_input(J)
    ip, ix, d = orc.csr_from_coo(n, row, col)
    g = ctx.trace_formats(capi.Csr(n, n, ip, ix, d))
    assert g["bnum"] == J["bnum"]
    assert g["bsize_list"].tolist() == J["bsize_list"] and g["seq_offset"].tolist() == J["seq_offset"]
    assert hx(g["seq_bitmap"]) == J["seq_bitmap"]
    assert hx(g["r_indptr"]) == J["r_indptr"] and hx(g["r_indices"]) == J["r_indices"]
    assert hx(g["panel_list"]) == J["panel_list"] and hx(g["spv8_list"]) == J["spv8_list"]
    assert hx(g["out_seq_all"]) == J["stable_out_seq"]
    assert hx(g["b_colidx_orig"]) == J["stable_b_colidx"]
    assert hx(g["b_local"]) == J["stable_b_local"]
    assert hx(g["keys"]) == J["stable_keys"]
    assert hx(g["seq_input"]) == J["stable_seq_input"]
    assert hx(g["rseq"]) == J["stable_rseq"]
    assert np.all(g["b_data"] == 1.0)


@pytest.mark.parametrize("threshold,sect", [(64, 2048), (500, 16), (65536, 2048)])
def test_pipeline_matches_oracle_many_blocks(ctx, threshold, sect):
    n = 1 << 13
    row, col = synth.rmat(13, 8, 77, True)
    ip, ix, d = orc.csr_from_coo(n, row, col)
    rng = np.random.default_rng(1)
    d = rng.integers(1, 9, d.size).astype(np.float64)     # general values: the gathers must carry them
    o = orc.trace_formats(n, ip, ix, d, sect=sect, threshold=threshold, sort_mode=1)
    g = ctx.trace_formats(capi.Csr(n, n, ip, ix, d), sect=sect, threshold=threshold)
    _compare(g, o)


def test_pipeline_edge_cases(ctx):
    # empty rows, rows longer than 32 and groups of short rows, a row of length > 4096, no entries at all
    rng = np.random.default_rng(9)
    n = 3000
    rows = [np.full(5000, 7), rng.integers(0, n, 20000), np.repeat(np.arange(100, 400), 3)]
    cols = [rng.integers(0, n, 5000), rng.integers(0, n, 20000), rng.integers(0, n, 900)]
    row, col = np.concatenate(rows), np.concatenate(cols)
    keep = row % 5 != 0          # every fifth row empty
    row, col = row[keep], col[keep]
    order = np.lexsort((row, col))
    ip, ix, d = orc.csr_from_coo(n, row[order], col[order])
    for thr in (65536, 300):
        o = orc.trace_formats(n, ip, ix, d, threshold=thr, sort_mode=1)
        g = ctx.trace_formats(capi.Csr(n, n, ip, ix, d), threshold=thr)
        _compare(g, o)
    ip0 = np.zeros(9, np.int32)
    o = orc.trace_formats(8, ip0, np.zeros(0, np.int32), np.zeros(0), sort_mode=1)
    g = ctx.trace_formats(capi.Csr(8, 8, ip0, np.zeros(0, np.int32), np.zeros(0)))
    _compare(g, o)


def test_panel_with_more_long_rows_than_the_lds_sort_holds(ctx):
    """10 000 rows of 33 entries in one section (one bucket, so they stay together) beside rows that carry ten times
    their entries: one panel then holds more than the 8192 long rows the per-panel sort keeps in LDS, and the call
    falls back to the two global sorts.  The outputs are the oracle's either way."""
    rng = np.random.default_rng(12)
    n, short = 20480, 10000
    row = [np.repeat(np.arange(short), 33)]
    col = [np.tile(np.arange(33), short)]
    for r in range(short, n):
        c = rng.choice(n, 315, replace=False)
        row.append(np.full(c.size, r))
        col.append(c)
    ip, ix, d = orc.csr_from_coo(n, np.concatenate(row), np.concatenate(col))
    o = orc.trace_formats(n, ip, ix, d, sort_mode=1)
    lens = np.diff(o["r_indptr"])
    most = 0
    for b in range(o["bnum"]):
        lst = o["panel_list"][o["panel_ptr"][b]:o["panel_ptr"][b + 1]] + o["bsize_list"][b]
        most = max([most] + [int((lens[lo:hi] >= 33).sum()) for lo, hi in zip(lst[:-1], lst[1:])])
    assert most > 8192
    _compare(ctx.trace_formats(capi.Csr(n, n, ip, ix, d)), o)


def test_block_with_more_elements_than_the_lds_renumbering_holds(ctx):
    """1000 rows of 600 entries over 61 000 columns never reach 65 536 distinct columns: one block of 600 000 elements,
    more than the 524 288 the block-local renumbering keeps flags for, so the call renumbers over the whole stream."""
    rng = np.random.default_rng(13)
    n = 61000
    row = np.repeat(rng.choice(n, 1000, replace=False), 600)
    col = np.concatenate([rng.choice(n, 600, replace=False) for _ in range(1000)])
    ip, ix, d = orc.csr_from_coo(n, row, col)
    o = orc.trace_formats(n, ip, ix, d, sort_mode=1)
    assert o["bnum"] == 1 and ix.size > 524288
    _compare(ctx.trace_formats(capi.Csr(n, n, ip, ix, d)), o)


def test_blocks_of_random_matrices_and_thresholds(ctx):
    """gen_blocks against the oracle on 24 random matrices (empty rows, repeated columns inside a row, a few long rows)
    with thresholds from 1 (every row that has an entry ends a block) to more columns than the matrix has."""
    rng = np.random.default_rng(3)
    for trial in range(24):
        n = int(rng.integers(5, 600))
        m = int(rng.integers(1, 20000))
        row = rng.integers(0, n, m)
        col = rng.integers(0, max(2, n // int(rng.integers(1, 8))), m)
        if trial % 3 == 0:                      # a few rows hold most entries
            row[: m // 2] = rng.integers(0, 3, m // 2)
        keep = row % int(rng.integers(2, 7)) != 0
        row, col = row[keep], col[keep]
        if row.size == 0:
            continue
        ip, ix, d = orc.csr_from_coo(n, row, col)
        R = capi.Csr(n, n, ip, ix, d)
        for thr in (1, 3, 17, 64, 300, 100000):
            want = orc.gen_blocks(n, ip, ix, thr)
            got = ctx.gen_blocks(R, thr, cap=n + 8)
            assert np.array_equal(got, want), (trial, n, thr, got[:8], want[:8])


def test_pipeline_of_random_matrices(ctx):
    """Every array of the pipeline against the oracle on random matrices with random section widths and thresholds
    (panels of one row, blocks of one row, rows longer than 32, empty rows)."""
    rng = np.random.default_rng(8)
    for trial in range(12):
        n = int(rng.integers(9, 3000))
        m = int(rng.integers(1, 40000))
        row = rng.integers(0, n, m)
        col = rng.integers(0, n, m)
        if trial % 2 == 0:
            row[: m // 3] = rng.integers(0, 5, m // 3)
        ip, ix, d = orc.csr_from_coo(n, row, col)
        d = rng.integers(1, 9, d.size).astype(np.float64)
        sect, thr = int(rng.choice([1, 7, 64, 2048])), int(rng.choice([1, 5, 40, 700, 65536]))
        o = orc.trace_formats(n, ip, ix, d, sect=sect, threshold=thr, sort_mode=1)
        g = ctx.trace_formats(capi.Csr(n, n, ip, ix, d), sect=sect, threshold=thr)
        _compare(g, o)


def test_entry_count_without_room_for_index_arithmetic_is_refused(ctx):
    """The kernels index the stream with 32-bit integers and look up to a tile ahead of a position; an entry count
    within 65 536 of the largest int32 is refused (SPMM_E_ARG) before any array is touched."""
    import ctypes as C
    ip = np.zeros(9, np.int32)
    one = np.zeros(1, np.int32)
    val = np.zeros(1, np.float64)
    for nnz, code in ((2 ** 31 - 1, -1), (2 ** 31 - 65535, -1)):
        v = capi.CsrView(8, 8, nnz, ip.ctypes.data, one.ctypes.data, val.ctypes.data)
        keys = np.zeros(8, np.int32)
        rc = capi.lib().spmm_row_keys(ctx._h, C.byref(v), C.c_int32(2048), keys.ctypes.data_as(C.c_void_p))
        assert rc == code, (nnz, rc)


def test_pipeline_block_capacity_is_an_error(ctx):
    n = 1 << 11
    row, col = synth.rmat(11, 8, 21, True)
    ip, ix, d = orc.csr_from_coo(n, row, col)
    with pytest.raises(capi.SpmmError) as e:
        ctx.trace_formats(capi.Csr(n, n, ip, ix, d), threshold=64, cap_blocks=5)
    assert e.value.code == -3


@pytest.mark.parametrize("threshold,sect", [(500, 16), (65536, 2048)])
def test_stage_entry_points_chain_to_the_pipeline(ctx, threshold, sect):
    """The five stage calls, chained through host arrays, give what spmm_trace_formats gives, and
    each stage equals the oracle's function for that stage."""
    n = 1 << 13
    row, col = synth.rmat(13, 8, 21, True)
    ip, ix, d = orc.csr_from_coo(n, row, col)
    d = np.random.default_rng(2).integers(1, 9, d.size).astype(np.float64)
    A = capi.Csr(n, n, ip, ix, d)
    full = ctx.trace_formats(A, sect=sect, threshold=threshold)

    seq, R = ctx.bitmap_reorder(A, sect)
    o_ip, o_ix, o_d, o_seq = orc.bitmap_reorder(n, ip, ix, d, sect)
    assert np.array_equal(seq, o_seq) and np.array_equal(R.indptr, o_ip) and np.array_equal(R.indices, o_ix)
    assert np.array_equal(R.data, o_d) and ctx.last_kernel_ms() > 0

    bsize = ctx.gen_blocks(R, threshold)
    assert np.array_equal(bsize, orc.gen_blocks(n, R.indptr, R.indices, threshold))
    with pytest.raises(capi.SpmmError) as e:
        ctx.gen_blocks(R, threshold=8, cap=3)               # more blocks than the list holds
    assert e.value.code == -3

    P = ctx.panels_sort_pack(R, bsize)
    for k in ("panel_ptr", "panel_list", "spv8_list", "out_seq_all", "b_indptr", "b_colidx_orig", "b_data"):
        assert np.array_equal(P[k], full[k]), k

    local, keys, off = ctx.first_touch(R, bsize, P["b_colidx_orig"])
    assert np.array_equal(local, full["b_local"]) and np.array_equal(keys, full["keys"])
    assert np.array_equal(off, full["seq_offset"])

    seq_input, rseq = ctx.wbseq(n, seq, P["out_seq_all"], bsize, keys)
    assert np.array_equal(seq_input, full["seq_input"]) and np.array_equal(rseq, full["rseq"])
    o_in, o_rseq = orc.wbseq(seq, P["out_seq_all"], bsize, keys)
    assert np.array_equal(seq_input, o_in) and np.array_equal(rseq, o_rseq)
    with pytest.raises(capi.SpmmError):
        ctx.wbseq(n, seq, P["out_seq_all"], bsize, np.array([n + 5], np.int32))    # a key that is no row id


@pytest.mark.parametrize("case", ["tiny", "rmat11_small_blocks", "rmat16"])
def test_spmv_over_the_packed_formats(ctx, case):
    """The consumer of the 8-row interleave and the per-block column numbers: y = A x computed from the arrays
    of trace_formats equals the CSR product, exactly for small integer x, within 1e-12 relative for random x.
    The oracle's own walk over the packed formats is checked against the CSR product on the small cases."""
    if case == "tiny":
        n, thr = 5000, 65536
        row, col = synth.uniform(n, 40000, 1)
    elif case == "rmat11_small_blocks":
        n, thr = 1 << 11, 300
        row, col = synth.rmat(11, 8, 21, True)
    else:
        n, thr = 1 << 16, 65536
        row, col = synth.rmat(16, 12, 5, True)
    ip, ix, d = orc.csr_from_coo(n, row, col)
    A = capi.Csr(n, n, ip, ix, d)
    T = ctx.trace_formats(A, threshold=thr)
    assert sum(T["spv8_list"]) > 0 and (case != "rmat11_small_blocks" or T["bnum"] > 10)
    import scipy.sparse as sp
    M = sp.csr_matrix((d, ix, ip), shape=(n, n))
    rng = np.random.default_rng(2)
    xi = rng.integers(1, 9, n).astype(np.float64)
    y = ctx.spv8_spmv(T, n, xi)
    assert np.array_equal(y, M @ xi)
    if n <= 5000:
        assert np.array_equal(orc.spv8_spmv(orc.trace_formats(n, ip, ix, d, threshold=thr), xi), M @ xi)
    xr = rng.standard_normal(n)
    np.testing.assert_allclose(ctx.spv8_spmv(T, n, xr), M @ xr, rtol=1e-12, atol=1e-12)


@pytest.mark.parametrize("nvec", [1, 2, 3, 8, 13])
def test_spmm_over_the_packed_formats(ctx, nvec):
    """Several right-hand vectors at once (passes of 8, 4, 2 and 1): every column of Y = A X equals the matrix-vector
    product of that column bit for bit (the same sums in the same order), and the CSR product exactly for small
    integer X."""
    n, thr = 1 << 11, 300
    row, col = synth.rmat(11, 8, 21, True)
    ip, ix, d = orc.csr_from_coo(n, row, col)
    d = np.random.default_rng(4).integers(1, 5, d.size).astype(np.float64)
    T = ctx.trace_formats(capi.Csr(n, n, ip, ix, d), threshold=thr)
    import scipy.sparse as sp
    M = sp.csr_matrix((d, ix, ip), shape=(n, n))
    rng = np.random.default_rng(5)
    Xi = rng.integers(1, 9, (n, nvec)).astype(np.float64)
    Y = ctx.spv8_spmm(T, n, Xi)
    assert Y.shape == (n, nvec) and np.array_equal(Y, M @ Xi)
    Xr = rng.standard_normal((n, nvec))
    Y = ctx.spv8_spmm(T, n, Xr)
    for v in range(nvec):
        assert np.array_equal(Y[:, v], ctx.spv8_spmv(T, n, Xr[:, v])), v
    np.testing.assert_allclose(Y, M @ Xr, rtol=1e-12, atol=1e-12)
    with pytest.raises(capi.SpmmError):
        ctx.spv8_spmm(T, n, np.zeros((n, 0)))


class _OneRank:
    """torch.distributed's calls for a world of one rank."""

    @staticmethod
    def all_gather_into_tensor(out, inp):
        out.copy_(inp.reshape(out.shape))


@pytest.mark.parametrize("threshold", [65536, 200])
def test_sharded_pipeline_stages_on_the_gpu(threshold):
    """spmm_amd/shard_pre.py with the GPU stages (GpuStages over the C ABI, spmm_row_keys among them) in a world of one
    rank against spmm_trace_formats and the oracle; the exchanges between ranks are covered on the CPU over gloo
    (tests/test_shard_pre_gloo.py)."""
    from spmm_amd import shard_pre
    n = 1 << 12
    row, col = synth.rmat(12, 8, 21, diagonal=True)
    ip, ix, d = orc.csr_from_coo(n, row, col)
    c = capi.Context(0)
    try:
        assert np.array_equal(c.row_keys(capi.Csr(n, n, ip, ix, d), 256), orc.row_keys(ip, ix, 256))
        S = shard_pre.trace_formats_sharded(shard_pre.GpuStages(c), _OneRank, 0, 1, n, n, ip, ix, d, 256, threshold)
        T = c.trace_formats(capi.Csr(n, n, ip, ix, d), sect=256, threshold=threshold)
    finally:
        c.close()
    W = orc.trace_formats(n, ip, ix, d, sect=256, threshold=threshold, sort_mode=1)
    assert S["blocks"] == (0, W["bnum"])
    for k in ("seq_bitmap", "r_indptr", "r_indices", "bsize_list", "panel_ptr", "panel_list", "spv8_list", "out_seq_all", "b_indptr",
              "b_colidx_orig", "b_local", "keys", "seq_offset", "seq_input", "rseq"):
        assert np.array_equal(S[k], W[k]), k
        assert np.array_equal(T[k], W[k]), k
