"""GPU: CSR build and the product C = A*B through the C ABI against the oracle and the
golden fixtures.  Structure (row pointers, column ids) must be bit-exact; values are exact
for the all-ones matrices the reference loader produces and within 1e-12 relative for
general fp64."""
import numpy as np
import pytest

from oracle import oracle as orc
from spmm_amd import capi, synth
from tests.helpers import case_input, golden_json, golden_npz, hx

pytestmark = pytest.mark.gpu


# Kernel selections the library reads from the environment when a context is created: the default
# plan, every window row through the dense-accumulator kernel, the dense kernel off (cursor kernel
# with 2^15-column table windows), and the cursor kernel for the symbolic phase with fixed-width
# numeric windows; rows of more than 8 entries shared by several workgroups of the dense kernel; dense rows cut into
# runs of 4 table windows, one workgroup each (runs with few entries of C as one bitmap window: by default up to 8192
# entries, never, and up to 200 so that both forms meet in one row); the dense kernel reading its window bounds from
# the window table instead of per-row bound tables, and with the bound tables built for a few rows at a time (a
# budget of some thousand bytes: rows beyond it read the window table); every window
# row of at most 65535 products through the 256-thread cursor workgroups (16-bit rank prefixes), with two-level
# windows (a bitmap bit per 32 columns) and with plain ones; the 1024-thread cursor kernel with two cursor sweeps
# instead of dense accumulators for a table window that overfills a register-resident window; the sorting kernels for rows of up to 8192 products (the
# default is 4096) and of up to 512 only; no window table of
# B at all (every window row through the launches over B's slab order), with wide windows and dense accumulators,
# and with narrow windows, the bitmap-rank accumulators and 256-thread workgroups; rows of very many entries NOT shared
# between workgroups; such rows shared by two workgroups that take runs of three table windows in turn; a budget of
# 1 MiB for the window table, so that the products of the larger test matrices fall back to the slab order on their own
# while the small ones keep the table.
PLANS = {
    "default": {},
    "dense_all": {"SPMM_DENSE_MIN": "1"},
    "dense_off": {"SPMM_DENSE_MIN": "0"},
    "cursor_sym_fixed_windows": {"SPMM_SYM_CFG": "1", "SPMM_ADAPTIVE": "0"},
    "split_rows": {"SPMM_DENSE_MIN": "1", "SPMM_GIANT_LEN": "8", "SPMM_GIANT_MAJOR": "0"},
    "dense_window_major": {"SPMM_DENSE_MIN": "1", "SPMM_MAJOR_CHUNK": "4"},
    "dense_window_major_no_sparse_runs": {"SPMM_DENSE_MIN": "1", "SPMM_MAJOR_CHUNK": "4", "SPMM_DENSE_SPARSE": "0"},
    "dense_window_major_mixed_runs": {"SPMM_DENSE_MIN": "1", "SPMM_MAJOR_CHUNK": "4", "SPMM_DENSE_SPARSE": "200"},
    "dense_window_major_runs_of_16": {"SPMM_DENSE_MIN": "1", "SPMM_MAJOR_CHUNK": "16"},
    "sliced_launches": {"SPMM_MAX_GRID_THREADS": "8192", "SPMM_DENSE_MIN": "1"},
    "dense_bounds_from_window_table": {"SPMM_DENSE_MIN": "1", "SPMM_MAJOR_CHUNK": "4", "SPMM_DENSE_BOUNDS": "0"},
    "dense_bounds_in_small_groups": {"SPMM_DENSE_MIN": "1", "SPMM_MAJOR_CHUNK": "4", "SPMM_DENSE_BOUNDS_MAX": "6000"},
    "dense_bounds_in_small_groups_sliced": {"SPMM_DENSE_MIN": "1", "SPMM_MAX_GRID_THREADS": "8192", "SPMM_DENSE_BOUNDS_MAX": "20000"},
    "slab": {"SPMM_SLAB": "1"},
    "slab_small_windows": {"SPMM_SLAB": "1", "SPMM_SLAB_WMAX": "4096", "SPMM_SLAB_SMAX": "2048", "SPMM_SLAB_DENSE_COLS": "1024",
                           "SPMM_SLAB_THREADS": "256"},
    "cursor_overfull_windows_in_two_sweeps": {"SPMM_WIN_DENSE_OVER": "0"},
    "sorted_rows_up_to_8192": {"SPMM_ESC_MAX": "8192"},
    "sorted_rows_up_to_4096": {"SPMM_ESC_MAX": "4096"},
    "sorted_rows_up_to_512": {"SPMM_ESC_MAX": "512"},
    "small_cursor_workgroups": {"SPMM_LIGHT_MAX": "65535"},
    # the rule of 256 products per table window without its floor of 8192: on the narrow test matrices the 1024-thread
    # cursor workgroups then take the rows from 257 products, as they do from 65 537 on a matrix of 2^22 columns
    "cursor_rows_from_257_products": {"SPMM_LIGHT_MAX": "256"},
    "small_cursor_workgroups_one_level": {"SPMM_LIGHT_MAX": "65535", "SPMM_LIGHT_TWO": "0", "SPMM_LIGHT_SYM": "0"},
    "rows_not_shared": {"SPMM_SPLIT_GIANT": "0"},
    "split_rows_two_workgroups": {"SPMM_DENSE_MIN": "1", "SPMM_GIANT_LEN": "8", "SPMM_GIANT_MAJOR": "0", "SPMM_GIANT_TASKS": "2",
                                  "SPMM_GIANT_CHUNK": "3"},
    "table_budget_falls_back_to_slab": {"SPMM_TABLE_MAX_MIB": "1"},
    "dense_tables_not_kept": {"SPMM_DENSE_MIN": "1", "SPMM_MAJOR_CHUNK": "4", "SPMM_DENSE_KEEP_MIB": "0"},
    "dense_tables_kept_in_part": {"SPMM_DENSE_MIN": "1", "SPMM_DENSE_KEEP_MIB": "1"},
}


@pytest.fixture(scope="module", params=list(PLANS))
def ctx(request):
    import os
    env = PLANS[request.param]
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


@pytest.mark.parametrize("name,value", [("SPMM_GIANT_TASKS", "1"), ("SPMM_GIANT_TASKS", "x"), ("SPMM_MAJOR_CHUNK", "6"), ("SPMM_ESC_MAX", "100000"),
                                        ("SPMM_SLAB_THREADS", "384"), ("SPMM_DENSE_SPARSE", "-1"), ("SPMM_LIGHT_MAX", "12abc"),
                                        ("SPMM_MAX_GRID_THREADS", "5000000000"), ("SPMM_SLAB", "2")])
def test_switch_outside_its_range_fails_the_context(name, value):
    """A kernel-selection switch that is not a number or lies outside the range its path was written for is an error of
    spmm_ctx_create (SPMM_E_ARG), not a silently clamped or wrapped setting."""
    import os
    old = os.environ.get(name)
    os.environ[name] = value
    try:
        with pytest.raises(capi.SpmmError) as e:
            capi.Context(0)
        assert e.value.code == -1 and name in str(e.value)
    finally:
        if old is None:
            os.environ.pop(name, None)
        else:
            os.environ[name] = old


def _ones_csr(n, row, col):
    ip, ix, d = orc.csr_from_coo(n, row, col)
    return capi.Csr(n, n, ip, ix, d)


@pytest.mark.parametrize("name", ["tiny", "rmat12d", "rmat17d", "rmat18"])
def test_csr_build_matches_golden(ctx, name):
    J = golden_json("pre_" + name)
    n, row, col = case_input(J)
    M = ctx.csr_from_coo(n, n, row, col)
    assert hx(M.indptr) == J["csr_indptr"] and hx(M.indices) == J["csr_indices"]
    assert np.all(M.data == 1.0)


def test_csr_build_edge_cases(ctx):
    # empty matrix, ragged rows, duplicates, unsorted input order
    M = ctx.csr_from_coo(5, 7, np.zeros(0, np.int32), np.zeros(0, np.int32))
    assert M.indptr.tolist() == [0] * 6 and M.nnz == 0
    rng = np.random.default_rng(3)
    row = rng.integers(0, 50, 5000).astype(np.int32)
    col = rng.integers(0, 9000, 5000).astype(np.int32)
    row[:100] = 7      # one long row
    M = ctx.csr_from_coo(50, 9000, row, col)
    ip, ix, d = orc.csr_from_coo(50, row, col)
    assert np.array_equal(M.indptr, ip) and np.array_equal(M.indices, ix)
    with pytest.raises(capi.SpmmError):
        ctx.csr_from_coo(50, 10, row, col)     # column ids out of range


@pytest.mark.parametrize("symmetric", [False, True])
@pytest.mark.parametrize("sum_duplicates", [False, True])
@pytest.mark.parametrize("with_values", [False, True])
def test_csr_build_loader_options(ctx, symmetric, sum_duplicates, with_values):
    """Caller values, symmetric expansion, duplicate summation: bit for bit the oracle's result (the order of
    every sum is fixed), and the same matrix scipy builds from the expanded entries."""
    import scipy.sparse as sp
    rng = np.random.default_rng(17)
    n, m = 3000, 60000
    row = rng.integers(0, n, m).astype(np.int32)
    col = rng.integers(0, n, m).astype(np.int32)
    row[:2000], col[:2000] = row[2000:4000], col[2000:4000]        # duplicates
    row[4000:4200] = col[4000:4200]                                # diagonal entries
    row[5000:5600], col[5000:5600] = 7, rng.integers(0, 40, 600)   # one row with many repeats
    val = rng.standard_normal(m) if with_values else None
    M = ctx.csr_from_coo_opts(n, n, row, col, val, symmetric, sum_duplicates)
    ip, ix, d = orc.csr_from_coo_opts(n, row, col, val, symmetric, sum_duplicates)
    assert np.array_equal(M.indptr, ip) and np.array_equal(M.indices, ix) and np.array_equal(M.data, d)
    v = np.ones(m) if val is None else val
    r2, c2, v2 = row, col, v
    if symmetric:
        off = row != col
        r2, c2, v2 = np.concatenate([row, col[off]]), np.concatenate([col, row[off]]), np.concatenate([v, v[off]])
    S = sp.coo_matrix((v2, (r2, c2)), shape=(n, n)).tocsr()
    G = sp.csr_matrix((M.data, M.indices, M.indptr), shape=(n, n))
    assert abs(S - G).max() <= 1e-12 * max(1.0, abs(S).max())
    if sum_duplicates:
        G.sort_indices()
        assert np.array_equal(G.indices, M.indices) and G.nnz == S.nnz
    if not (symmetric or sum_duplicates or with_values):
        P = ctx.csr_from_coo(n, n, row, col)                       # the reference's loader rule
        assert np.array_equal(P.indptr, M.indptr) and np.array_equal(P.indices, M.indices) and np.array_equal(P.data, M.data)
    with pytest.raises(capi.SpmmError):
        ctx.csr_from_coo_opts(n, n + 1, row, col, val, True, sum_duplicates)    # symmetric needs a square matrix


@pytest.mark.parametrize("name", ["tiny", "rmat10", "rmat12d", "rmat14"])
def test_product_matches_golden(ctx, name):
    J = golden_json("spgemm_" + name)
    n, row, col = case_input(J)
    A = _ones_csr(n, row, col)
    c_ip, nnz_c, prods = ctx.spgemm_symbolic(A)
    assert prods == J["products"] and nnz_c == J["nnzC"]
    assert hx(c_ip) == J["c_indptr"]
    ci, cd = ctx.spgemm_numeric(A, None, c_ip)
    assert hx(ci) == J["c_indices"]
    assert hx(cd) == J["c_data"]
    assert cd.sum() == J["products"]


_SNAP_LIKE = {}


def _snap_like(name):
    """(A, c_indptr, c_indices, c_data) of the oracle for a graph of the size and rough shape of a SNAP input the
    benchmark names (the files are not available offline); computed once for all plans."""
    if name not in _SNAP_LIKE:
        n, gen = {"web-Google": (916428, synth.web_like), "sx-askubuntu": (159316, synth.askubuntu_like),
                  "web-Stanford": (281903, synth.stanford_like)}[name]
        row, col = gen()
        ip, ix, d = orc.csr_from_coo(n, row, col)
        o_ip = orc.spgemm_symbolic(n, n, ip, ix, ip, ix)
        oi, od, _ = orc.spgemm_numeric(n, (ip, ix, d), (ip, ix, d), o_ip)
        _SNAP_LIKE[name] = (capi.Csr(n, n, ip, ix, d), o_ip, oi, od)
    return _SNAP_LIKE[name]


@pytest.mark.parametrize("name", ["web-Stanford", "web-Google", "sx-askubuntu"])
def test_product_snap_sized_graphs(ctx, name):
    """Whole C of a web-Stanford-sized graph (2.3 M entries, a popular head that puts rows into the window bins), of a
    web-Google-sized graph (5.1 M entries, every row in the smallest sorted bin) and of an
    sx-askubuntu-sized one (0.9 M entries, 5.4e8 products, skewed rows in every bin) against the oracle."""
    A, o_ip, oi, od = _snap_like(name)
    c_ip, nnz_c, _ = ctx.spgemm_symbolic(A)
    assert nnz_c == int(o_ip[-1]) and np.array_equal(c_ip, o_ip)
    ci, cd = ctx.spgemm_numeric(A, None, c_ip)
    assert np.array_equal(ci, oi)
    assert np.array_equal(cd, od)


def test_product_full_arrays_and_row_ranges(ctx):
    Z = golden_npz("spgemm_rmat10")
    A = _ones_csr(1024, Z["row"], Z["col"])
    c_ip, ci, cd = ctx.spgemm(A)
    assert np.array_equal(c_ip, Z["c_indptr"]) and np.array_equal(ci, Z["c_indices"]) and np.array_equal(cd, Z["c_data"])
    for r0, r1 in [(0, 1), (100, 400), (1000, 1024), (512, 512)]:
        ci2, cd2 = ctx.spgemm_numeric(A, None, c_ip, r0, r1)
        assert np.array_equal(ci2, ci[c_ip[r0]:c_ip[r1]]) and np.array_equal(cd2, cd[c_ip[r0]:c_ip[r1]])


def test_product_rectangular_fp64(ctx):
    Z = golden_npz("spgemm_rect")
    A = capi.Csr(300, 200, Z["a_indptr"], Z["a_indices"], Z["a_data"])
    B = capi.Csr(200, 500, Z["b_indptr"], Z["b_indices"], Z["b_data"])
    c_ip, ci, cd = ctx.spgemm(A, B)
    assert np.array_equal(c_ip, Z["c_indptr"]) and np.array_equal(ci, Z["c_indices"])
    np.testing.assert_allclose(cd, Z["c_data"], rtol=1e-12, atol=1e-13)


@pytest.mark.parametrize("log_ncol", [25, 27])
def test_product_very_wide_b(ctx, log_ncol):
    """B with 2^25 / 2^27 columns: the window table switches to 2^15 / 2^17-column windows (at most 1024
    windows), so the dense kernel sweeps every table window in 2 / 8 sub-ranges, and the symbolic bitmap
    needs 32 / 128 passes.  A's rows span every bin, up to 360 000 products."""
    rng = np.random.default_rng(23)
    ka, ncol = 1500, 1 << log_ncol
    b_len = rng.integers(1, 4000, ka)
    b_ip = np.zeros(ka + 1, np.int32)
    b_ip[1:] = np.cumsum(b_len)
    b_ix = np.concatenate([np.sort(rng.choice(ncol, l, replace=False)) for l in b_len]).astype(np.int32)
    b_ix[:b_len[0]] = np.sort(rng.choice(40000, b_len[0], replace=False))         # one row crowded into a few windows
    b_d = rng.integers(1, 4, b_ix.size).astype(np.float64)
    rows = 260
    a_len = np.concatenate([rng.integers(0, 4, 100), rng.integers(4, 40, 120), rng.integers(60, 180, 40)])
    a_ip = np.zeros(rows + 1, np.int32)
    a_ip[1:] = np.cumsum(a_len)
    a_ix = np.concatenate([np.sort(rng.choice(ka, l, replace=False)) for l in a_len]).astype(np.int32)
    a_d = rng.integers(1, 4, a_ix.size).astype(np.float64)
    A, B = capi.Csr(rows, ka, a_ip, a_ix, a_d), capi.Csr(ka, ncol, b_ip, b_ix, b_d)
    prod, _ = orc.row_products(a_ip, a_ix, b_ip)
    assert prod.max() > 262144 and ((prod > 8192) & (prod < 262144)).any() and (prod <= 512).any()
    c_ip, nnz_c, prods = ctx.spgemm_symbolic(A, B)
    o_ip = orc.spgemm_symbolic(rows, ncol, a_ip, a_ix, b_ip, b_ix)
    assert np.array_equal(c_ip, o_ip) and prods == int(prod.sum())
    ci, cd = ctx.spgemm_numeric(A, B, c_ip)
    oi, od, _ = orc.spgemm_numeric(ncol, (a_ip, a_ix, a_d), (b_ip, b_ix, b_d), o_ip)
    assert np.array_equal(ci, oi) and np.array_equal(cd, od)


@pytest.mark.parametrize("ncol", [1 << 22, (1 << 22) + 1, 1 << 23, (1 << 23) + 1, 1 << 26, (1 << 26) + 1])
def test_sorted_rows_at_the_limits_of_the_register_sort(ctx, ncol):
    """The sorting kernels keep (column << bits | product number) words in registers while the column ids leave room
    for the product number: 26 bits for rows of up to 64 products, 23 up to 512, 22 up to 1024.  Rows with exactly 63,
    64, 65, 128, 129, 256, 257, 384, 512, 513, 1024 and 1025 products, few and many entries, over matrices one column
    below and above each limit, with B's columns at the top of the range (the largest words there are) and repeated
    across its rows (runs of equal columns to sum)."""
    rng = np.random.default_rng(31)
    lens = [300] * 4 + [128] * 28 + [16] * 16 + [1] * 16            # rows of B
    pool = np.arange(ncol - 700, ncol, dtype=np.int64)
    b_ip = np.zeros(len(lens) + 1, np.int32)
    b_ip[1:] = np.cumsum(lens)
    b_ix = np.concatenate([np.sort(rng.choice(pool, n, replace=False)) for n in lens]).astype(np.int32)
    assert (b_ix[b_ip[1:] - 1] == ncol - 1).any()                     # the last column is some rows' last product
    b_d = rng.integers(1, 4, b_ix.size).astype(np.float64)
    big, mid, small, one = list(range(0, 4)), list(range(4, 32)), list(range(32, 48)), list(range(48, 64))
    rows = [mid[:8], mid[1:8] + small[:8], mid[:8] + one[:1], mid[:4], mid[:4] + one[:1], mid[:3] + small[:8] + one[:1],
            mid[:3], mid[:2], mid[:2] + one[:1], mid[:1], mid[:1] + one[:1], small[:4], small[:4] + one[:1],
            small[:3] + one[:15], big[:2], big[:3] + mid[4:5], big[:3] + small[:7] + one[:12], [mid[0]] * 1 + small[:16] + one[:16],
            mid[20:28], one[:5], []]
    a_ip = np.zeros(len(rows) + 1, np.int32)
    a_ip[1:] = np.cumsum([len(r) for r in rows])
    a_ix = np.concatenate([np.sort(np.array(r, np.int32)) for r in rows if r]).astype(np.int32)
    a_d = rng.integers(1, 4, a_ix.size).astype(np.float64)
    prod, _ = orc.row_products(a_ip, a_ix, b_ip)
    assert {63, 64, 65, 128, 129, 256, 257, 384, 512, 513, 1024, 1025} <= set(int(x) for x in prod)
    A, B = capi.Csr(len(rows), len(lens), a_ip, a_ix, a_d), capi.Csr(len(lens), ncol, b_ip, b_ix, b_d)
    c_ip, _, _ = ctx.spgemm_symbolic(A, B)
    o_ip = orc.spgemm_symbolic(len(rows), ncol, a_ip, a_ix, b_ip, b_ix)
    assert np.array_equal(c_ip, o_ip)
    ci, cd = ctx.spgemm_numeric(A, B, c_ip)
    oi, od, _ = orc.spgemm_numeric(ncol, (a_ip, a_ix, a_d), (b_ip, b_ix, b_d), o_ip)
    assert np.array_equal(ci, oi) and np.array_equal(cd, od)


def test_product_heavy_rows_all_bins(ctx):
    """A matrix whose rows fall in every bin, including rows with more distinct output columns
    than one value sweep holds and duplicate entries in A (kept by the reference loader)."""
    rng = np.random.default_rng(11)
    n = 300000                      # two numeric windows of 262144 columns
    rows, cols = [], []
    # row 0: 40000 entries -> dense output over both windows; row 1: duplicates
    rows.append(np.zeros(40000, np.int64)); cols.append(rng.integers(0, n, 40000))
    rows.append(np.ones(600, np.int64)); cols.append(np.repeat(rng.integers(0, n, 300), 2))
    m = 600000
    rows.append(rng.integers(2, n, m)); cols.append(rng.integers(0, n, m))
    # a hub column region so that some rows get many products
    rows.append(rng.integers(2, 300, 9000)); cols.append(rng.integers(0, 50, 9000))
    rows.append(rng.integers(0, 50, 100000)); cols.append(rng.integers(0, n, 100000))
    row = np.concatenate(rows); col = np.concatenate(cols)
    order = np.lexsort((row, col))           # (col,row) order, duplicates kept
    row, col = row[order].astype(np.int32), col[order].astype(np.int32)
    ip, ix, d = orc.csr_from_coo(n, row, col)
    d = rng.integers(1, 4, d.size).astype(np.float64)
    A = capi.Csr(n, n, ip, ix, d)
    prod, total = orc.row_products(ip, ix, ip)
    assert prod.max() > 100000 and (prod == 0).any() and ((prod > 0) & (prod <= 512)).any() and ((prod > 512) & (prod <= 4096)).any()
    c_ip, nnz_c, prods = ctx.spgemm_symbolic(A)
    o_ip = orc.spgemm_symbolic(n, n, ip, ix, ip, ix)
    assert prods == total
    assert np.array_equal(c_ip, o_ip)
    assert np.diff(o_ip).max() > 8192 * 2      # more than two value sweeps in one row
    ci, cd = ctx.spgemm_numeric(A, None, c_ip)
    oi, od, _ = orc.spgemm_numeric(n, (ip, ix, d), (ip, ix, d), o_ip)
    assert np.array_equal(ci, oi)
    assert np.array_equal(cd, od)              # small integers: exact


def test_product_rows_shared_by_workgroups(ctx):
    """Rows of A with more than 8192 entries and at least 2^18 products: the numeric phase splits such a row
    over several workgroups by table windows, placed by the symbolic phase's per-window counts.  Checked for the
    whole matrix, for row ranges, and for a numeric call whose symbolic counts were replaced by another
    product's in between (the rows then go through the unsplit kernel)."""
    rng = np.random.default_rng(41)
    ka, ncol, rows = 30000, 1 << 20, 40
    b_len = rng.integers(20, 120, ka)
    b_len[:50] = rng.integers(3000, 9000, 50)                  # long rows of B: whole-wave sweeps
    b_ip = np.zeros(ka + 1, np.int32)
    b_ip[1:] = np.cumsum(b_len)
    skew = (rng.random(int(b_len.sum())) ** 3 * ncol).astype(np.int64)   # crowded towards low columns
    b_rows = [np.unique(skew[b_ip[i]:b_ip[i + 1]]) for i in range(ka)]
    b_ix = np.concatenate(b_rows).astype(np.int32)
    b_ip[1:] = np.cumsum([r.size for r in b_rows])
    b_d = rng.integers(1, 4, b_ix.size).astype(np.float64)
    a_len = np.concatenate([[20000, 9000, 8193, 8192, 5000], rng.integers(0, 3000, rows - 5)])
    a_ip = np.zeros(rows + 1, np.int32)
    a_ip[1:] = np.cumsum(a_len)
    a_ix = np.concatenate([np.sort(rng.choice(ka, l, replace=False)) for l in a_len]).astype(np.int32)
    a_d = rng.integers(1, 4, a_ix.size).astype(np.float64)
    A, B = capi.Csr(rows, ka, a_ip, a_ix, a_d), capi.Csr(ka, ncol, b_ip, b_ix, b_d)
    prod, _ = orc.row_products(a_ip, a_ix, b_ip)
    assert prod[:4].min() >= 1 << 18
    o_ip = orc.spgemm_symbolic(rows, ncol, a_ip, a_ix, b_ip, b_ix)
    oi, od, _ = orc.spgemm_numeric(ncol, (a_ip, a_ix, a_d), (b_ip, b_ix, b_d), o_ip)
    c_ip, nnz_c, prods = ctx.spgemm_symbolic(A, B)
    assert np.array_equal(c_ip, o_ip) and prods == int(prod.sum())
    ci, cd = ctx.spgemm_numeric(A, B, c_ip)
    assert np.array_equal(ci, oi) and np.array_equal(cd, od)
    for r0, r1 in [(0, 1), (1, 7), (3, rows)]:
        ci2, cd2 = ctx.spgemm_numeric(A, B, c_ip, r0, r1)
        assert np.array_equal(ci2, oi[o_ip[r0]:o_ip[r1]]) and np.array_equal(cd2, od[o_ip[r0]:o_ip[r1]])
    Z = golden_npz("spgemm_rmat10")
    ctx.spgemm(_ones_csr(1024, Z["row"], Z["col"]))          # another product's plan in between
    ci, cd = ctx.spgemm_numeric(A, B, c_ip)
    assert np.array_equal(ci, oi) and np.array_equal(cd, od)


_PAIR = {}


def _same_shape_pair():
    """Two matrices with the same nrow, ncol and nnz and different structure (the second holds the rows of the first in
    reverse order), rows in every bin, and the oracle's C = A*A of both."""
    if not _PAIR:
        n = 1 << 13
        row, col = synth.rmat(13, 16, 3)
        ip, ix, d = orc.csr_from_coo(n, row, col)
        lens = np.diff(ip)[::-1]
        ip2 = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
        ix2 = np.concatenate([ix[ip[r]:ip[r + 1]] for r in range(n - 1, -1, -1)]).astype(np.int32)
        mats = [(ip, ix, d), (ip2, ix2, d.copy())]
        outs = []
        for m in mats:
            o_ip = orc.spgemm_symbolic(n, n, m[0], m[1], m[0], m[1])
            oi, od, _ = orc.spgemm_numeric(n, m, m, o_ip)
            outs.append((o_ip, oi, od))
        prod, _ = orc.row_products(ip, ix, ip)
        assert prod.max() > 70000 and ((prod > 8192) & (prod < 65536)).any() and (prod <= 64).any()
        assert not np.array_equal(outs[0][0], outs[1][0])
        _PAIR.update(n=n, mats=mats, outs=outs)
    return _PAIR["n"], _PAIR["mats"], _PAIR["outs"]


def test_plan_of_another_matrix_is_not_reused_host(ctx):
    """symbolic(A1), symbolic(A2), numeric(A1), numeric(A2) with equal nrow/ncol/nnz through the host-pointer API (every
    call uploads into the same device slots): the numeric calls must not bin A1's rows by A2's row products."""
    n, mats, outs = _same_shape_pair()
    A = [capi.Csr(n, n, *m) for m in mats]
    c = [ctx.spgemm_symbolic(a)[0] for a in A]
    for k in (0, 1):
        assert np.array_equal(c[k], outs[k][0])
    for k in (0, 1, 1, 0):
        ci, cd = ctx.spgemm_numeric(A[k], None, c[k])
        assert np.array_equal(ci, outs[k][1]) and np.array_equal(cd, outs[k][2])
    r0, r1 = 5, n // 3
    ci, cd = ctx.spgemm_numeric(A[1], None, c[1], r0, r1)
    assert np.array_equal(ci, outs[1][1][c[1][r0]:c[1][r1]]) and np.array_equal(cd, outs[1][2][c[1][r0]:c[1][r1]])


def test_repeated_steps_keep_the_plan_and_stay_exact(ctx):
    """symbolic, numeric (in two row blocks), symbolic, numeric on ONE pair of matrices, as the steps of the benchmark
    do: the second symbolic call keeps the row products and the window table of the first (content digests equal) and
    counts again; then the other matrix of the same shape in between (a miss), and the first once more."""
    n, mats, outs = _same_shape_pair()
    A = [capi.Csr(n, n, *m) for m in mats]
    for step, k in enumerate((0, 0, 1, 0, 0, 0)):
        if step == 5:
            ctx.forget_plan()          # the last step rebuilds everything although the matrices are the same
        c_ip, nnz_c, _ = ctx.spgemm_symbolic(A[k])
        assert np.array_equal(c_ip, outs[k][0]) and nnz_c == int(outs[k][0][-1])
        mid = n // 2
        for r0, r1 in ((0, mid), (mid, n)):
            ci, cd = ctx.spgemm_numeric(A[k], None, c_ip, r0, r1)
            assert np.array_equal(ci, outs[k][1][c_ip[r0]:c_ip[r1]]) and np.array_equal(cd, outs[k][2][c_ip[r0]:c_ip[r1]])
            if step == 5 and r0 == 0:
                ctx.forget_plan()      # and between the numeric calls of one product


def test_plan_of_another_matrix_is_not_reused_dev_refill(ctx):
    """The same through the device-pointer API with the matrices refilled IN PLACE between the calls: the pointers,
    shapes and sizes of every call are identical, only the contents differ."""
    import torch
    n, mats, outs = _same_shape_pair()
    dev = torch.device("cuda", 0)
    nnz = int(mats[0][1].size)
    t_ip = torch.empty(n + 1, dtype=torch.int32, device=dev)
    t_ix = torch.empty(nnz, dtype=torch.int32, device=dev)
    t_d = torch.empty(nnz, dtype=torch.float64, device=dev)
    v = capi.dev_view(n, n, nnz, t_ip, t_ix, t_d)

    def fill(k):
        ctx.sync()
        t_ip.copy_(torch.from_numpy(mats[k][0]))
        t_ix.copy_(torch.from_numpy(mats[k][1]))
        t_d.copy_(torch.from_numpy(mats[k][2]))
        torch.cuda.synchronize()

    c_ip = [torch.empty(n + 1, dtype=torch.int64, device=dev) for _ in (0, 1)]
    for k in (0, 1):
        fill(k)
        nnz_c, _ = ctx.spgemm_symbolic_dev(v, v, c_ip[k])
        ctx.sync()
        assert nnz_c == int(outs[k][0][-1]) and np.array_equal(c_ip[k].cpu().numpy(), outs[k][0])
    cap = int(max(o[0][-1] for o in outs))
    c_ix = torch.empty(cap, dtype=torch.int32, device=dev)
    c_val = torch.empty(cap, dtype=torch.float64, device=dev)
    for k in (0, 1, 0):
        fill(k)
        ctx.spgemm_numeric_dev(v, v, c_ip[k], 0, n, c_ix, c_val)
        ctx.sync()
        m = int(outs[k][0][-1])
        assert np.array_equal(c_ix[:m].cpu().numpy(), outs[k][1]) and np.array_equal(c_val[:m].cpu().numpy(), outs[k][2])


def test_numeric_sees_values_changed_in_place(ctx):
    """numeric, new values written into the same device array, numeric again (twice, and on two row ranges): the
    second and third results carry the new values, i.e. B's packed records are not reused across a change of values,
    and are reusable when nothing changed."""
    import torch
    n, mats, outs = _same_shape_pair()
    dev = torch.device("cuda", 0)
    ip, ix, d = mats[0]
    t_ip, t_ix = torch.from_numpy(ip).to(dev), torch.from_numpy(ix).to(dev)
    t_d = torch.from_numpy(d).to(dev)
    v = capi.dev_view(n, n, int(ix.size), t_ip, t_ix, t_d)
    c_ip = torch.empty(n + 1, dtype=torch.int64, device=dev)
    ctx.spgemm_symbolic_dev(v, v, c_ip)
    ctx.sync()
    m = int(outs[0][0][-1])
    c_ix = torch.empty(m, dtype=torch.int32, device=dev)
    c_val = torch.empty(m, dtype=torch.float64, device=dev)
    rng = np.random.default_rng(5)
    d2 = rng.integers(1, 5, d.size).astype(np.float64)
    o_ip = outs[0][0]
    _, od2, _ = orc.spgemm_numeric(n, (ip, ix, d2), (ip, ix, d2), o_ip)
    half = n // 2
    for vals, want in ((d, outs[0][2]), (d2, od2), (d2, od2), (d, outs[0][2])):
        ctx.sync()
        t_d.copy_(torch.from_numpy(vals))
        torch.cuda.synchronize()
        c_val.zero_()
        torch.cuda.synchronize()
        # two row ranges of the same product: the second call finds the first call's records
        ctx.spgemm_numeric_dev(v, v, c_ip, 0, half, c_ix, c_val)
        ctx.sync()
        k = int(o_ip[half])
        assert np.array_equal(c_val[:k].cpu().numpy(), want[:k])
        ctx.spgemm_numeric_dev(v, v, c_ip, half, n, c_ix, c_val)
        ctx.sync()
        assert np.array_equal(c_val[:m - k].cpu().numpy(), want[k:]) and np.array_equal(c_ix[:m - k].cpu().numpy(), outs[0][1][k:])


def test_product_sorted_rows_with_many_empty_b_rows(ctx):
    """Rows of A with thousands of entries but at most 8192 products (most of the rows of B they name are
    empty): the expand step of the sorting kernels walks many chunks of entries that add nothing."""
    rng = np.random.default_rng(5)
    kb, ncol = 30000, 70000
    b_len = np.zeros(kb, np.int64)
    full = rng.choice(kb, 600, replace=False)
    b_len[full] = rng.integers(1, 90, full.size)
    b_ip = np.zeros(kb + 1, np.int32)
    b_ip[1:] = np.cumsum(b_len)
    b_ix = np.concatenate([np.sort(rng.choice(ncol, l, replace=False)) for l in b_len if l]).astype(np.int32)
    b_d = rng.integers(1, 4, b_ix.size).astype(np.float64)
    a_rows = []
    for nent, nfull in [(9000, 120), (5000, 100), (3000, 60), (700, 100), (12000, 5), (200, 0)]:
        ent = np.concatenate([rng.choice(full, nfull, replace=False), rng.choice(np.setdiff1d(np.arange(kb), full), nent - nfull, replace=False)])
        a_rows.append(np.sort(ent))
    a_ip = np.zeros(len(a_rows) + 1, np.int32)
    a_ip[1:] = np.cumsum([r.size for r in a_rows])
    a_ix = np.concatenate(a_rows).astype(np.int32)
    a_d = rng.integers(1, 4, a_ix.size).astype(np.float64)
    prod, _ = orc.row_products(a_ip, a_ix, b_ip)
    assert prod.max() <= 8192 and prod[0] > 512 and prod[-1] == 0
    A, B = capi.Csr(len(a_rows), kb, a_ip, a_ix, a_d), capi.Csr(kb, ncol, b_ip, b_ix, b_d)
    c_ip, nnz_c, prods = ctx.spgemm_symbolic(A, B)
    o_ip = orc.spgemm_symbolic(len(a_rows), ncol, a_ip, a_ix, b_ip, b_ix)
    assert np.array_equal(c_ip, o_ip) and prods == int(prod.sum())
    ci, cd = ctx.spgemm_numeric(A, B, c_ip)
    oi, od, _ = orc.spgemm_numeric(ncol, (a_ip, a_ix, a_d), (b_ip, b_ix, b_d), o_ip)
    assert np.array_equal(ci, oi) and np.array_equal(cd, od)


def test_light_rows_with_many_entries():
    """Light window rows (the 256-thread workgroups of both phases) whose row of A has more than 256 entries: the
    directory is then built chunk by chunk and the windows stay plain bitmaps; rows of fewer entries beside them take
    the two-level windows; one row's first table window overfills a register-resident window.  Both light forms."""
    import os
    rng = np.random.default_rng(77)
    nb_rows, ncol = 3000, 600000               # B: 3000 x 600000 (37 table windows)
    b_cols = []
    for k in range(nb_rows):
        if k < 2900:                           # short rows spread over all columns
            c = np.unique(rng.integers(0, ncol, rng.integers(5, 40)))
        else:                                  # rows crowded into the first table window
            c = np.unique(rng.integers(0, 16000, 60))
        b_cols.append(c)
    b_ip = np.zeros(nb_rows + 1, np.int32)
    b_ip[1:] = np.cumsum([c.size for c in b_cols])
    b_ix = np.concatenate(b_cols).astype(np.int32)
    b_d = rng.integers(1, 5, b_ix.size).astype(np.float64)
    a_rows = [np.unique(rng.integers(0, 2900, m)) for m in (700, 300, 257, 40, 1000)]
    a_rows.append(np.arange(2900, 3000))       # 100 entries x 60: about 6000 products in table window 0
    a_rows.append(np.zeros(0, np.int64))
    na = len(a_rows)
    a_ip = np.zeros(na + 1, np.int32)
    a_ip[1:] = np.cumsum([c.size for c in a_rows])
    a_ix = np.concatenate(a_rows).astype(np.int32)
    a_d = rng.integers(1, 5, a_ix.size).astype(np.float64)
    assert (np.diff(a_ip) > 256).sum() >= 3
    A = capi.Csr(na, nb_rows, a_ip, a_ix, a_d)
    B = capi.Csr(nb_rows, ncol, b_ip, b_ix, b_d)
    o_ip = orc.spgemm_symbolic(na, ncol, a_ip, a_ix, b_ip, b_ix)
    oi, od, _ = orc.spgemm_numeric(ncol, (a_ip, a_ix, a_d), (b_ip, b_ix, b_d), o_ip)
    for env in ({"SPMM_LIGHT_MAX": "65535"}, {"SPMM_LIGHT_MAX": "65535", "SPMM_LIGHT_TWO": "0", "SPMM_LIGHT_SYM": "0"}):
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
        c_ip, nnz_c, prods = c.spgemm_symbolic(A, B)
        assert np.array_equal(c_ip, o_ip)
        ci, cd = c.spgemm_numeric(A, B, c_ip)
        assert np.array_equal(ci, oi) and np.array_equal(cd, od)
        c.close()


def test_plan_threshold_follows_window_count():
    """Without SPMM_DENSE_MIN the dense kernel takes rows from 3072 products per table window of B (2^14
    columns per window up to 1024 windows), and not below 2^16 products."""
    c = capi.Context(0)
    try:
        assert c.plan_info(1 << 22) == (14, 256, 786432)
        assert c.plan_info(1 << 20) == (14, 64, 196608)
        assert c.plan_info(1000) == (14, 1, 65536)
        shift, nbase, dmin = c.plan_info(1 << 27)
        assert (shift, nbase, dmin) == (17, 1024, 3072 * 1024)
        # and the sorting kernels rows of up to four products per table window, between 1024 and 4096
        assert [c.sort_limit(n) for n in (1000, 1 << 20, 1 << 22, 1 << 23, 1 << 24, 1 << 27)] == [1024, 1024, 1024, 2048, 4096, 4096]
    finally:
        c.close()


def test_product_rejects_unsorted_b(ctx):
    ip = np.array([0, 2, 3], np.int32)
    ix = np.array([1, 0, 1], np.int32)
    A = capi.Csr(2, 2, ip, ix, np.ones(3))
    with pytest.raises(capi.SpmmError) as e:
        ctx.spgemm_symbolic(A)
    assert e.value.code == -4


@pytest.mark.parametrize("bad_ip", [[0, 3, 2, 4], [0, 2, 5, 4], [-1, 2, 3, 4], [0, 2, 3, 9]])
def test_product_rejects_broken_row_pointers(ctx, bad_ip):
    """Row pointers that descend or point outside the entry arrays are an argument error, found by the structure
    check before any kernel indexes the entries through them."""
    rng = np.random.default_rng(5)
    n = 3
    ix = np.sort(rng.integers(0, n, 4)).astype(np.int32)
    good = capi.Csr(n, n, np.array([0, 1, 3, 4], np.int32), np.array([0, 1, 2, 0], np.int32), np.ones(4))
    bad = capi.Csr(n, n, np.array(bad_ip, np.int32), np.array([0, 1, 2, 0], np.int32), np.ones(4))
    del ix
    for A, B in ((bad, good), (good, bad)):
        with pytest.raises(capi.SpmmError) as e:
            ctx.spgemm_symbolic(A, B)
        assert e.value.code == -1
    c_ip, _, _ = ctx.spgemm_symbolic(good, good)        # the context still works afterwards
    assert c_ip.tolist() == orc.spgemm_symbolic(n, n, good.indptr, good.indices, good.indptr, good.indices).tolist()


def test_product_empty(ctx):
    A = capi.Csr(4, 4, np.zeros(5, np.int32), np.zeros(0, np.int32), np.zeros(0))
    c_ip, nnz_c, prods = ctx.spgemm_symbolic(A)
    assert c_ip.tolist() == [0] * 5 and nnz_c == 0 and prods == 0
    ci, cd = ctx.spgemm_numeric(A, None, c_ip)
    assert ci.size == 0


def test_product_of_a_row_shard_view(ctx):
    """The view of A that the multi-GPU benchmark hands every rank: row pointers starting at the shard's first row
    (indptr[r_lo:], first value not 0), the indices and data of the WHOLE matrix, nnz of the whole matrix.  Symbolic
    and numeric over such views (one of them with no rows at all) must give the matching rows of the whole product."""
    import torch
    n, mats, outs = _same_shape_pair()
    ip, ix, d = mats[0]
    o_ip, oi, od = outs[0]
    dev = torch.device("cuda", 0)
    t_ip, t_ix, t_d = (torch.from_numpy(x).to(dev) for x in (ip, ix, d))
    nnz = int(ix.size)
    vB = capi.dev_view(n, n, nnz, t_ip, t_ix, t_d)
    torch.cuda.synchronize()
    for r_lo, r_hi in [(0, 3), (3, 3), (3, 2000), (2000, n)]:
        rows = r_hi - r_lo
        vA = capi.dev_view(rows, n, nnz, t_ip[r_lo:], t_ix, t_d)
        c_ip = torch.empty(rows + 1, dtype=torch.int64, device=dev)
        nnz_c, prods = ctx.spgemm_symbolic_dev(vA, vB, c_ip)
        ctx.sync()
        assert np.array_equal(c_ip.cpu().numpy(), o_ip[r_lo:r_hi + 1] - o_ip[r_lo]) and nnz_c == int(o_ip[r_hi] - o_ip[r_lo])
        c_ix = torch.empty(max(nnz_c, 1), dtype=torch.int32, device=dev)
        c_val = torch.empty(max(nnz_c, 1), dtype=torch.float64, device=dev)
        ctx.spgemm_numeric_dev(vA, vB, c_ip, 0, rows, c_ix, c_val)
        ctx.sync()
        assert np.array_equal(c_ix[:nnz_c].cpu().numpy(), oi[o_ip[r_lo]:o_ip[r_hi]])
        assert np.array_equal(c_val[:nnz_c].cpu().numpy(), od[o_ip[r_lo]:o_ip[r_hi]])
