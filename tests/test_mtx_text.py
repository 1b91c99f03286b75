"""MatrixMarket text -> (row, col): the reference's reader is serial_newblock_clock.cpp:47-104.  The fixture
tests/golden/mtx_text_cases.json (tools/gen_mtx_fixtures.py) holds texts and the entries the compiled
reference's csr_matrix() agrees with; the oracle's token-level restatement and the GPU converter must both
return them, and the GPU converter must refuse the inputs on which the reference reads garbage."""
import ctypes as C
import json
import os

import numpy as np
import pytest

from oracle import oracle as orc
from spmm_amd import capi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FIX = json.load(open(os.path.join(ROOT, "tests", "golden", "mtx_text_cases.json")))


def _header(text):
    nrow, ncol, nnz, off, per = C.c_int32(), C.c_int32(), C.c_int64(), C.c_size_t(), C.c_int32()
    rc = capi.lib().spmm_mtx_header(text, C.c_size_t(len(text)), C.byref(nrow), C.byref(ncol), C.byref(nnz), C.byref(off), C.byref(per))
    return rc, nrow.value, ncol.value, nnz.value, off.value, per.value


@pytest.mark.parametrize("case", FIX["cases"], ids=[c["name"] for c in FIX["cases"]])
def test_oracle_reader_matches_fixture(case):
    nrow, ncol, row, col = orc.mtx_tokens(case["text"].encode())
    assert (nrow, ncol) == (case["nrow"], case["ncol"])
    assert row.tolist() == case["row"] and col.tolist() == case["col"]


@pytest.mark.parametrize("case", FIX["cases"], ids=[c["name"] for c in FIX["cases"]])
def test_header_on_the_host(case):
    text = case["text"].encode()
    rc, nrow, ncol, nnz, off, per = _header(text)
    assert rc == 0 and (nrow, ncol, nnz) == (case["nrow"], case["ncol"], len(case["row"]))
    body = text[off:]
    assert text[:off].endswith(b"\n") and per == (2 if body.split(b"\n", 1)[0].count(b" ") == 1 else 3)


def test_header_rejects_a_bad_size_line():
    assert _header(b"%%MatrixMarket\n5 x 3\n1 2\n")[0] == -1
    assert _header(b"%only comments\n")[0] == -1


@pytest.mark.gpu
@pytest.mark.parametrize("case", FIX["cases"], ids=[c["name"] for c in FIX["cases"]])
def test_gpu_reader_matches_fixture(case):
    ctx = capi.Context(0)
    try:
        nrow, ncol, row, col = ctx.mtx_parse(case["text"].encode())
    finally:
        ctx.close()
    assert (nrow, ncol) == (case["nrow"], case["ncol"])
    assert row.tolist() == case["row"] and col.tolist() == case["col"]


@pytest.mark.gpu
@pytest.mark.parametrize("name", list(FIX["errors"]))
def test_gpu_reader_refuses_what_the_reference_misreads(name):
    ctx = capi.Context(0)
    try:
        with pytest.raises(capi.SpmmError) as e:
            ctx.mtx_parse(FIX["errors"][name].encode())
        assert e.value.code == -1
    finally:
        ctx.close()


@pytest.mark.gpu
@pytest.mark.parametrize("values", [False, True])
def test_gpu_reader_large_text_round_trip(values):
    """2 million entries (tens of megabytes, thousands of 4096-byte tiles): the entries written are the
    entries read, and the CSR built from them equals the oracle's."""
    rng = np.random.default_rng(9)
    n, m = 1 << 20, 2_000_000
    r, c = rng.integers(1, n + 1, m), rng.integers(1, n + 1, m)
    import pandas as pd
    cols = {"r": r, "c": c}
    if values:
        cols["v"] = rng.integers(1, 1000, m)
    body = pd.DataFrame(cols).to_csv(sep=" ", header=False, index=False)
    text = (f"%%MatrixMarket matrix coordinate {'integer' if values else 'pattern'} general\n% generated\n{n} {n} {m}\n" + body).encode()
    ctx = capi.Context(0)
    try:
        nrow, ncol, row, col = ctx.mtx_parse(text)
        assert (nrow, ncol) == (n, n)
        assert np.array_equal(row, r - 1) and np.array_equal(col, c - 1)
        M = ctx.csr_from_coo(n, n, row, col)
    finally:
        ctx.close()
    ip, ix, _ = orc.csr_from_coo(n, (r - 1).astype(np.int32), (c - 1).astype(np.int32))
    assert np.array_equal(M.indptr, ip) and np.array_equal(M.indices, ix)


TRICKY = ["0", "-0.0", "1", "-1", "3.25", "+3.25E+2", ".5", "5.", "1e22", "1e23", "1e-22", "1e-23", "9007199254740991", "9007199254740992",
          "9007199254740993", "0.1", "0.30000000000000004", "2.2250738585072014e-308", "4.9e-324", "1e-400", "1.7976931348623157e308",
          "1e400", "123456789012345678901234567890", "0.000000000000000000000000000001", "1.5e+300", "6.02214076e23", "inf", "-inf",
          "12345.678901234567", "1E5", "000123.4500", "1e0000000005", "0e999", "7.", "-.25e-2"]


def _values_text(tokens, n=97):
    body = "".join(f"{1 + i % n} {1 + (7 * i) % n} {t}\n" for i, t in enumerate(tokens))
    return f"%%MatrixMarket matrix coordinate real general\n{n} {n} {len(tokens)}\n{body}".encode()


def test_oracle_reader_values():
    nrow, ncol, row, col, val = orc.mtx_tokens(_values_text(TRICKY), values=True)
    assert val.tolist() == [float(t) for t in TRICKY] and row[3] == 3 and col[3] == 21


@pytest.mark.gpu
def test_gpu_reader_values_equal_strtod():
    """Value tokens: short decimals are converted on the GPU, the rest by strtod on the host; both must give
    the correctly rounded fp64, bit for bit (Python's float() is the yardstick), for hand-picked edge cases and
    for a few hundred thousand random ones in several notations."""
    rng = np.random.default_rng(4)
    x = rng.standard_normal(200000) * 10.0 ** rng.integers(-30, 30, 200000)
    tokens = TRICKY + [repr(float(v)) for v in x[:50000]] + [f"{v:.6e}" for v in x[50000:100000]] + \
        [f"{v:.3f}" for v in x[100000:150000] % 1000] + [str(int(v)) for v in (x[150000:] % 1e6)]
    text = _values_text(tokens, n=100003)
    ctx = capi.Context(0)
    try:
        nrow, ncol, row, col, val = ctx.mtx_parse_values(text)
        want = np.array([float(t) for t in tokens])
        assert np.array_equal(val.view(np.uint64), want.view(np.uint64))          # bit for bit, signed zeros included
        o = orc.mtx_tokens(text, values=True)
        assert np.array_equal(row, o[2]) and np.array_equal(col, o[3])
        with pytest.raises(capi.SpmmError):
            ctx.mtx_parse_values(_values_text(["1.5", "abc", "2"]))
        with pytest.raises(capi.SpmmError):
            ctx.mtx_parse_values(_values_text(["1.5", "1.5x", "2"]))
    finally:
        ctx.close()
