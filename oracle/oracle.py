"""ctypes front end of the CPU oracle (oracle/liboracle.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, by __graft_entry__.smoke() and by
bench.py's cpu_baseline leg.  The shipped package (spmm_amd) never imports it.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

I32 = np.int32
I64 = np.int64
F64 = np.float64


def build():
    """Compile liboracle.so (and oracle/_ref when the reference tree is present)."""
    subprocess.check_call(["make", "-s", "-C", _HERE])


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "liboracle.so")
        if not os.path.exists(path):
            subprocess.check_call(["make", "-s", "-C", _HERE, "liboracle.so"])
        _LIB = C.CDLL(path)
        _LIB.orc_spgemm_symbolic.restype = C.c_int64
        _LIB.orc_spgemm_numeric.restype = C.c_int64
        _LIB.orc_row_products.restype = C.c_int64
        _LIB.orc_fnv1a64_u32.restype = C.c_uint64
    return _LIB


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def _i32(a):
    return np.ascontiguousarray(a, dtype=I32)


def fnv(a):
    a = np.ascontiguousarray(a)
    assert a.dtype.itemsize == 4
    return int(lib().orc_fnv1a64_u32(_p(a), C.c_int64(a.size)))


def read_mtx(path):
    """serial_newblock_clock.cpp:47-104: returns (nrow, ncol, row, col) zero-based."""
    nrow, ncol, nnz = C.c_int32(), C.c_int32(), C.c_int64()
    rc = lib().orc_read_mtx(path.encode(), C.byref(nrow), C.byref(ncol), C.byref(nnz), None, None, 1)
    if rc:
        raise IOError(f"orc_read_mtx({path}) = {rc}")
    row = np.empty(nnz.value, I32)
    col = np.empty(nnz.value, I32)
    rc = lib().orc_read_mtx(path.encode(), C.byref(nrow), C.byref(ncol), C.byref(nnz), _p(row), _p(col), 0)
    if rc:
        raise IOError(f"orc_read_mtx({path}) = {rc}")
    return nrow.value, ncol.value, row, col


def mtx_tokens(text, values=False):
    """Token-level restatement of the reader, serial_newblock_clock.cpp:47-104, on the bytes of a file:
    leading '%' lines skipped (:50, :64); `fin >> num_row >> num_col >> num_lines` (:65); two or three
    tokens per entry decided by the ' ' bytes of the line after the size line (:52-58, :80); then
    `fin >> row >> col` (:86) or `fin >> row >> col >> s` (:98) num_lines times: whitespace-separated
    tokens, line ends carry no meaning, what follows the last entry is not read.
    Returns (nrow, ncol, row, col), zero-based int32."""
    p = 0
    while p < len(text) and text[p:p + 1] == b"%":
        q = text.find(b"\n", p)
        p = len(text) if q < 0 else q + 1
    q = text.find(b"\n", p)
    q = len(text) if q < 0 else q
    nrow, ncol, nnz = (int(x) for x in text[p:q].split()[:3])
    body = text[q + 1:]
    per = 2 if body.split(b"\n", 1)[0].count(b" ") + 1 == 2 else 3     # :52-58, :80
    tok = body.split()
    if len(tok) < per * nnz:
        raise ValueError("fewer tokens than entries")
    row = np.array([int(t) for t in tok[0:per * nnz:per]], dtype=np.int64) - 1
    col = np.array([int(t) for t in tok[1:per * nnz:per]], dtype=np.int64) - 1
    if values:
        # the token the reference reads into a scratch string and drops (:96-98), as the fp64 strtod() gives
        # (Python's float() is the same correctly rounded conversion)
        if per != 3:
            raise ValueError("no value tokens")
        return nrow, ncol, row.astype(I32), col.astype(I32), np.array([float(t) for t in tok[2:3 * nnz:3]], F64)
    return nrow, ncol, row.astype(I32), col.astype(I32)


def csr_from_coo(nrow, row, col):
    row, col = _i32(row), _i32(col)
    nnz = row.size
    indptr = np.empty(nrow + 1, I32)
    indices = np.empty(nnz, I32)
    data = np.empty(nnz, F64)
    lib().orc_csr_from_coo(C.c_int32(nrow), C.c_int64(nnz), _p(row), _p(col), _p(indptr), _p(indices), _p(data))
    return indptr, indices, data


def csr_from_coo_opts(nrow, row, col, val=None, symmetric=False, sum_duplicates=False):
    """CSR build with loader options the reference does not have (it forces values to 1.0, keeps duplicates,
    ignores symmetry: serial_newblock_clock.cpp:84, 96-98).  Mirrored entries follow all the original ones in
    input order; without summing a row keeps stream order (like :108-112), with summing its columns ascend and
    equal (row, column) entries are added one after the other in stream order."""
    row, col = np.asarray(row, np.int64), np.asarray(col, np.int64)
    val = np.ones(row.size, F64) if val is None else np.asarray(val, F64)
    if symmetric:
        off = row != col
        row, col, val = np.concatenate([row, col[off]]), np.concatenate([col, row[off]]), np.concatenate([val, val[off]])
    if not sum_duplicates:
        order = np.argsort(row, kind="stable")
        r, c, v = row[order], col[order], val[order]
    else:
        order = np.lexsort((col, row))          # stable: equal (row, col) stay in stream order
        r, c, v = row[order], col[order], val[order]
        head = np.ones(r.size, bool)
        head[1:] = (r[1:] != r[:-1]) | (c[1:] != c[:-1])
        starts = np.flatnonzero(head)
        out = np.empty(starts.size, F64)
        ends = np.append(starts[1:], r.size)
        for k, (a, b) in enumerate(zip(starts, ends)):
            acc = v[a]
            for j in range(a + 1, b):
                acc += v[j]
            out[k] = acc
        r, c, v = r[starts], c[starts], out
    indptr = np.zeros(nrow + 1, I32)
    np.add.at(indptr, r + 1, 1)
    return np.cumsum(indptr).astype(I32), c.astype(I32), v


def row_keys(indptr, indices, sect=2048):
    indptr, indices = _i32(indptr), _i32(indices)
    keys = np.empty(indptr.size - 1, I32)
    lib().orc_row_keys(C.c_int32(indptr.size - 1), _p(indptr), _p(indices), C.c_int32(sect), _p(keys))
    return keys


def bitmap_reorder(ncol, indptr, indices, data, sect=2048):
    indptr, indices = _i32(indptr), _i32(indices)
    data = np.ascontiguousarray(data, F64)
    nrow = indptr.size - 1
    o_ip = np.empty(nrow + 1, I32)
    o_ix = np.empty(indices.size, I32)
    o_d = np.empty(indices.size, F64)
    seq = np.empty(nrow, I32)
    lib().orc_bitmap_reorder(C.c_int32(nrow), C.c_int32(ncol), _p(indptr), _p(indices), _p(data),
                             C.c_int32(sect), _p(o_ip), _p(o_ix), _p(o_d), _p(seq))
    return o_ip, o_ix, o_d, seq


def gen_blocks(ncol, indptr, indices, threshold=65536, cap=None):
    indptr, indices = _i32(indptr), _i32(indices)
    nrow = indptr.size - 1
    cap = cap if cap is not None else nrow + 2
    out = np.empty(cap, I32)
    bnum = lib().orc_gen_blocks(C.c_int32(nrow), C.c_int32(ncol), _p(indptr), _p(indices),
                                C.c_int32(threshold), _p(out), C.c_int32(cap))
    if bnum < 0:
        raise OverflowError("block list capacity exceeded")
    return out[:bnum + 1].copy()


def panel_list(indptr, row_beg, row_end):
    indptr = _i32(indptr)
    out = np.empty((row_end - row_beg) // 8 + 3, I32)
    n = lib().orc_panel_list(_p(indptr), C.c_int32(row_beg), C.c_int32(row_end), _p(out))
    return out[:n].copy()


def panel_sort(indptr, row_beg, row_end, plist, mode=1):
    indptr, plist = _i32(indptr), _i32(plist)
    out_seq = np.empty(row_end - row_beg, I32)
    spv8 = np.empty(plist.size - 1, I32)
    lib().orc_panel_sort(_p(indptr), C.c_int32(row_beg), C.c_int32(row_end), _p(plist),
                         C.c_int32(plist.size), C.c_int(mode), _p(out_seq), _p(spv8))
    return out_seq, spv8


def gather_spv8(indptr, indices, data, row_beg, row_end, out_seq, plist, spv8):
    indptr, indices, out_seq, plist, spv8 = map(_i32, (indptr, indices, out_seq, plist, spv8))
    data = np.ascontiguousarray(data, F64)
    rows = row_end - row_beg
    nz = int(indptr[row_end] - indptr[row_beg])
    b_ip = np.empty(rows + 1, I32)
    b_ci = np.empty(nz, I32)
    b_d = np.empty(nz, F64)
    lib().orc_gather_spv8(_p(indptr), _p(indices), _p(data), C.c_int32(row_beg), C.c_int32(row_end),
                          _p(out_seq), _p(plist), C.c_int32(plist.size), _p(spv8), _p(b_ip), _p(b_ci), _p(b_d))
    return b_ip, b_ci, b_d


def first_touch(ncol, colidx):
    colidx = _i32(colidx)
    local = np.empty(colidx.size, I32)
    keys = np.empty(colidx.size, I32)
    n = lib().orc_first_touch(C.c_int32(ncol), C.c_int64(colidx.size), _p(colidx), _p(local), _p(keys))
    return local, keys[:n].copy()


def wbseq(seq_bitmap, out_seq_all, bsize_list, keys):
    seq_bitmap, out_seq_all, bsize_list, keys = map(_i32, (seq_bitmap, out_seq_all, bsize_list, keys))
    nrow = seq_bitmap.size
    seq_input = np.empty(keys.size, I32)
    rseq = np.empty(nrow, I32)
    lib().orc_wbseq(C.c_int32(nrow), _p(seq_bitmap), _p(out_seq_all), _p(bsize_list),
                    C.c_int32(bsize_list.size - 1), _p(keys), C.c_int64(keys.size), _p(seq_input), _p(rseq))
    return seq_input, rseq


class _TraceOut(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("seq_bitmap", "r_indptr", "r_indices", "r_data", "bsize_list", "seq_offset")] + \
               [("cap_blocks", C.c_int32)] + \
               [(n, C.c_void_p) for n in ("panel_ptr", "panel_list", "spv8_list", "out_seq_all", "b_indptr",
                                          "b_colidx_orig", "b_local", "b_data", "keys", "seq_input", "rseq")] + \
               [(n, C.c_int32) for n in ("bnum", "n_keys", "n_panel_entries", "n_spv8")]


def trace_formats(ncol, indptr, indices, data, sect=2048, threshold=65536, sort_mode=1):
    """serial_newblock_clock.cpp:221-499.  Returns a dict of every array the pipeline produces."""
    indptr, indices = _i32(indptr), _i32(indices)
    data = np.ascontiguousarray(data, F64)
    nrow, nnz = indptr.size - 1, indices.size
    cap = nrow + 1
    a = dict(
        seq_bitmap=np.empty(nrow, I32), r_indptr=np.empty(nrow + 1, I32), r_indices=np.empty(nnz, I32),
        r_data=np.empty(nnz, F64), bsize_list=np.empty(cap + 1, I32), seq_offset=np.empty(cap + 1, I32),
        panel_ptr=np.empty(cap + 1, I32), panel_list=np.empty(nrow // 8 + 3 * cap + 8, I32),
        spv8_list=np.empty(nrow // 8 + 3 * cap + 8, I32), out_seq_all=np.empty(nrow, I32),
        b_indptr=np.empty(nrow + cap + 1, I32), b_colidx_orig=np.empty(nnz, I32), b_local=np.empty(nnz, I32),
        b_data=np.empty(nnz, F64), keys=np.empty(nnz, I32), seq_input=np.empty(nnz, I32), rseq=np.empty(nrow, I32))
    o = _TraceOut()
    for k, v in a.items():
        setattr(o, k, v.ctypes.data)
    o.cap_blocks = cap
    bnum = lib().orc_trace_formats(C.c_int32(nrow), C.c_int32(ncol), _p(indptr), _p(indices), _p(data),
                                   C.c_int32(sect), C.c_int32(threshold), C.c_int(sort_mode), C.byref(o))
    if bnum < 0:
        raise OverflowError("orc_trace_formats failed")
    nk = o.n_keys
    out = dict(a)
    out["bnum"] = bnum
    out["bsize_list"] = a["bsize_list"][:bnum + 1].copy()
    out["seq_offset"] = a["seq_offset"][:bnum + 1].copy()
    out["panel_ptr"] = a["panel_ptr"][:bnum + 1].copy()
    out["panel_list"] = a["panel_list"][:o.n_panel_entries].copy()
    out["spv8_list"] = a["spv8_list"][:o.n_spv8].copy()
    out["b_indptr"] = a["b_indptr"][:nrow + bnum].copy()
    out["keys"] = a["keys"][:nk].copy()
    out["seq_input"] = a["seq_input"][:nk].copy()
    return out


def spv8_spmv(T, x):
    """y = A x read from the packed per-block formats of trace_formats() (the layout written at
    serial_newblock_clock.cpp:336-399 and :187-204): block b holds rows bsize_list[b]..bsize_list[b+1] of the
    reordered matrix in out_seq order; its element stream starts at r_indptr[bsize_list[b]]; b_indptr gives
    block-local row offsets; inside panel p the first spv8_list[p] rows are groups of 8 equal-length rows stored
    element-interleaved (element r of row c at 8*r + c); b_local are per-block column numbers whose original ids
    are keys[seq_offset[b] + local].  Packed row t of block b is original row seq_bitmap[bs + out_seq_all[bs + t]].
    Plain Python loops: small inputs only."""
    x = np.asarray(x, F64)
    nrow = T["seq_bitmap"].size
    y = np.zeros(nrow, F64)
    bs_list, ro = T["bsize_list"], T["r_indptr"]
    for b in range(T["bnum"]):
        bs, be = int(bs_list[b]), int(bs_list[b + 1])
        base = int(ro[bs])
        bip = T["b_indptr"][bs + b: be + b + 1]
        pl = T["panel_list"][T["panel_ptr"][b]: T["panel_ptr"][b + 1]]
        sp = T["spv8_list"][T["panel_ptr"][b] - b: T["panel_ptr"][b + 1] - b - 1]
        xk = x[T["keys"][T["seq_offset"][b]: T["seq_offset"][b + 1]]]
        grouped = np.zeros(be - bs, bool)
        for p in range(len(pl) - 1):
            grouped[pl[p]: pl[p] + sp[p]] = True
        for t in range(be - bs):
            if grouped[t]:
                g0 = t - (t - int(pl[np.searchsorted(pl, t, side="right") - 1])) % 8
                L = int(bip[g0 + 1] - bip[g0])
                pos = base + int(bip[g0]) + (t - g0) + 8 * np.arange(L)
            else:
                pos = base + np.arange(int(bip[t]), int(bip[t + 1]))
            orig = T["seq_bitmap"][bs + T["out_seq_all"][bs + t]]
            acc = 0.0
            for q in pos:
                acc += T["b_data"][q] * xk[T["b_local"][q]]
            y[orig] = acc
    return y


def spgemm_symbolic(nrow, ncolB, a_indptr, a_indices, b_indptr, b_indices):
    a_indptr, a_indices, b_indptr, b_indices = map(_i32, (a_indptr, a_indices, b_indptr, b_indices))
    c_indptr = np.empty(nrow + 1, I64)
    lib().orc_spgemm_symbolic(C.c_int32(nrow), C.c_int32(ncolB), _p(a_indptr), _p(a_indices),
                              _p(b_indptr), _p(b_indices), _p(c_indptr))
    return c_indptr


def spgemm_numeric(ncolB, a, b, c_indptr, row_beg=0, row_end=None):
    """a, b = (indptr, indices, data).  Returns (c_indices, c_data, products) for the row range."""
    a_ip, a_ix = _i32(a[0]), _i32(a[1])
    b_ip, b_ix = _i32(b[0]), _i32(b[1])
    a_d = np.ascontiguousarray(a[2], F64)
    b_d = np.ascontiguousarray(b[2], F64)
    c_indptr = np.ascontiguousarray(c_indptr, I64)
    row_end = a_ip.size - 1 if row_end is None else row_end
    nz = int(c_indptr[row_end] - c_indptr[row_beg])
    ci = np.empty(nz, I32)
    cd = np.empty(nz, F64)
    prods = lib().orc_spgemm_numeric(C.c_int32(row_beg), C.c_int32(row_end), C.c_int32(ncolB), _p(a_ip), _p(a_ix),
                                     _p(a_d), _p(b_ip), _p(b_ix), _p(b_d), _p(c_indptr), _p(ci), _p(cd))
    return ci, cd, int(prods)


def spgemm_rows(ncolB, a, b, rows, threads=1):
    """Gustavson over a list of rows with `threads` OpenMP threads (the CPU baseline of the benchmark).  Returns
    (entries of C per listed row, sum of all values, intermediate products)."""
    a_ip, a_ix = _i32(a[0]), _i32(a[1])
    b_ip, b_ix = _i32(b[0]), _i32(b[1])
    a_d = np.ascontiguousarray(a[2], F64)
    b_d = np.ascontiguousarray(b[2], F64)
    rows = _i32(rows)
    counts = np.zeros(rows.size, I64)
    vsum = C.c_double()
    f = lib().orc_spgemm_rows_omp
    f.restype = C.c_int64
    prods = f(C.c_int32(rows.size), _p(rows), C.c_int32(ncolB), _p(a_ip), _p(a_ix), _p(a_d), _p(b_ip), _p(b_ix), _p(b_d),
              C.c_int32(int(threads)), _p(counts), C.byref(vsum))
    return counts, float(vsum.value), int(prods)


def spgemm(ncolB, a, b):
    c_indptr = spgemm_symbolic(len(a[0]) - 1, ncolB, a[0], a[1], b[0], b[1])
    ci, cd, _ = spgemm_numeric(ncolB, a, b, c_indptr)
    return c_indptr, ci, cd


def row_products(a_indptr, a_indices, b_indptr):
    a_indptr, a_indices, b_indptr = map(_i32, (a_indptr, a_indices, b_indptr))
    prod = np.empty(a_indptr.size - 1, I64)
    total = lib().orc_row_products(C.c_int32(a_indptr.size - 1), _p(a_indptr), _p(a_indices), _p(b_indptr), _p(prod))
    return prod, int(total)
