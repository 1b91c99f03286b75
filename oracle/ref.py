"""ctypes front end of the compiled reference (oracle/_ref/libref.so).

TEST INFRASTRUCTURE ONLY, and only where the reference tree was present to build
it (this container).  Used by tools/gen_fixtures.py to produce tests/golden/ and
by the CPU tests that cross-check the oracle when the library exists.

The reference is only safe on matrices WITHOUT empty rows (its panel sort
indexes a scratch array sized by block nnz with block rows, v8sort.h:157-160)
and with at most 999 blocks (serial_newblock_clock.cpp:552-553).
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_PATH = os.path.join(_HERE, "_ref", "libref.so")
_LIB = None
I32, F64 = np.int32, np.float64


def available():
    return os.path.exists(_PATH)


def lib():
    global _LIB
    if _LIB is None:
        _LIB = C.CDLL(_PATH)
    return _LIB


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def load(path):
    shape = np.zeros(3, I32)
    lib().ref_load(path.encode(), _p(shape))
    nrow, ncol, nnz = map(int, shape)
    indptr, indices, data = np.empty(nrow + 1, I32), np.empty(nnz, I32), np.empty(nnz, F64)
    lib().ref_get_csr(_p(indptr), _p(indices), _p(data))
    return nrow, ncol, indptr, indices, data


def set_csr(ncol, indptr, indices, data):
    indptr = np.ascontiguousarray(indptr, I32)
    indices = np.ascontiguousarray(indices, I32)
    data = np.ascontiguousarray(data, F64)
    lib().ref_set_csr(C.c_int(indptr.size - 1), C.c_int(ncol), C.c_int(indices.size), _p(indptr), _p(indices), _p(data))


def trace_formats(nrow, nnz):
    seq = np.empty(max(nnz, 1), I32)
    rseq = np.empty(nrow, I32)
    bsize = np.zeros(1000, I32)
    soff = np.zeros(1000, I32)
    end_seq, bnum = C.c_int(), C.c_int()
    lib().ref_trace_formats(_p(seq), C.byref(end_seq), _p(rseq), C.byref(bnum), _p(bsize), _p(soff))
    b = bnum.value
    return dict(seq_input=seq[:end_seq.value].copy(), rseq=rseq, bnum=b, bsize_list=bsize[:b + 1].copy(),
                seq_offset=soff[:b + 1].copy())


def bitmap_reorder(nrow, nnz, sect=2048):
    seq = np.empty(nrow, I32)
    indptr, indices, data = np.empty(nrow + 1, I32), np.empty(nnz, I32), np.empty(nnz, F64)
    lib().ref_bitmap_reorder(C.c_int(sect), _p(seq), _p(indptr), _p(indices), _p(data))
    return indptr, indices, data, seq


def gen_new_panels():
    bsize = np.zeros(1000, I32)
    bnum = C.c_int()
    lib().ref_gen_new_panels(_p(bsize), C.byref(bnum))
    return bsize[:bnum.value + 1].copy()


def stage_block(r_indptr, row_beg, row_end):
    rows = row_end - row_beg
    nz = int(r_indptr[row_end] - r_indptr[row_beg])
    plist = np.empty(max(nz, rows) + 8, I32)
    npl = C.c_int()
    out_seq = np.empty(rows, I32)
    spv8 = np.empty(rows // 8 + 8, I32)
    b_ip = np.empty(rows + 1, I32)
    b_ci, b_d = np.empty(max(nz, 1), I32), np.empty(max(nz, 1), F64)
    b_lo, keys = np.empty(max(nz, 1), I32), np.empty(max(nz, 1), I32)
    nk = lib().ref_stage_block(C.c_int(row_beg), C.c_int(row_end), _p(plist), C.byref(npl), _p(out_seq), _p(spv8),
                               _p(b_ip), _p(b_ci), _p(b_d), _p(b_lo), _p(keys))
    n = npl.value
    return dict(panel_list=plist[:n].copy(), out_seq=out_seq, spv8_list=spv8[:n - 1].copy(), b_indptr=b_ip,
                b_colidx=b_ci[:nz].copy(), b_data=b_d[:nz].copy(), b_local=b_lo[:nz].copy(), keys=keys[:nk].copy())


def wbseq(seq_bitmap, out_seq_all, bsize_list, keys):
    seq_bitmap = np.ascontiguousarray(seq_bitmap, I32)
    out_seq_all = np.ascontiguousarray(out_seq_all, I32)
    bsize_list = np.ascontiguousarray(bsize_list, I32)
    keys = np.ascontiguousarray(keys, I32)
    nrow = seq_bitmap.size
    seq_input, rseq = np.empty(max(keys.size, 1), I32), np.empty(nrow, I32)
    n = lib().ref_wbseq(C.c_int(nrow), _p(seq_bitmap), _p(out_seq_all), _p(bsize_list), C.c_int(bsize_list.size - 1),
                        _p(keys), C.c_int(keys.size), _p(seq_input), _p(rseq))
    return seq_input[:n].copy(), rseq
