"""ctypes binding of libspmm_hip.so (include/spmm_hip.h).

This is the host-side mirror used by the Python tests and by bench.py; the C++ harness
(spmm_amd/harness/serial.cpp) links the same library directly.  There is no CPU fallback:
if the library is missing, or no gfx950 GPU is visible, the calls raise.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# SPMM_LIB_PATH names another build of the same library (the diagnostic build of `make -C spmm_amd/csrc diag`)
LIB_PATH = os.environ.get("SPMM_LIB_PATH") or os.path.join(_HERE, "csrc", "libspmm_hip.so")
_LIB = None

I32, I64, F64 = np.int32, np.int64, np.float64


PH_BINS, PH_WINDOW_SYM, PH_ESC_SYM, PH_WINDOW_NUM, PH_ESC_NUM, PH_SCAN, PH_CSR_BUILD = range(7)
PH_DENSE_NUM = 13
PH_LIGHT_NUM = 14
PH_DENSE_SYM = 15
PH_SLAB_PLAN = 16


def slab_windows(blk, ncol, wmax, smax):
    """Column-window boundaries from the entries per block of 64 columns (host rule of the slab order)."""
    blk = np.ascontiguousarray(blk, dtype=np.uint32)
    wb = np.empty(blk.size + 2, I32)
    n = C.c_int32()
    _chk(lib().spmm_slab_windows(_p(blk), C.c_int32(blk.size), C.c_int32(ncol), C.c_int32(wmax), C.c_int64(smax), _p(wb),
                                 C.c_int32(wb.size), C.byref(n)))
    return wb[:n.value + 1].copy()


def bin_limits():
    a, b = C.c_int32(), C.c_int32()
    lib().spmm_spgemm_bin_limits(C.byref(a), C.byref(b))
    return int(a.value), int(b.value)


class SpmmError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"libspmm_hip error {code}: {msg}")
        self.code = code


class CsrView(C.Structure):
    """spmm_csr: the reference's SpM layout (csr.h:8-20)."""
    _fields_ = [("nrow", C.c_int32), ("ncol", C.c_int32), ("nnz", C.c_int32),
                ("indptr", C.c_void_p), ("indices", C.c_void_p), ("data", C.c_void_p)]


class TraceOut(C.Structure):
    """spmm_trace (include/spmm_hip.h): host arrays of every stage of the pipeline."""
    _fields_ = [(n, C.c_void_p) for n in ("seq_bitmap", "r_indptr", "r_indices", "r_data", "bsize_list", "seq_offset")] + \
               [("cap_blocks", C.c_int32)] + \
               [(n, C.c_void_p) for n in ("panel_ptr", "panel_list", "spv8_list", "out_seq_all", "b_indptr",
                                          "b_colidx_orig", "b_local", "b_data", "keys", "seq_input", "rseq")] + \
               [(n, C.c_int32) for n in ("bnum", "n_keys", "n_panel_entries", "n_spv8")]


def lib():
    global _LIB
    if _LIB is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(f"{LIB_PATH} is missing: build it with `make -C spmm_amd/csrc` "
                              "(or python -c 'import __graft_entry__ as g; g.build()')")
        _LIB = C.CDLL(LIB_PATH)
        _LIB.spmm_last_error.restype = C.c_char_p
        _LIB.spmm_last_kernel_ms.restype = C.c_double
        _LIB.spmm_last_kernel_ms.argtypes = [C.c_void_p]
    return _LIB


def _chk(rc):
    if rc != 0:
        raise SpmmError(rc, lib().spmm_last_error().decode())


def _p(a):
    if a is None:
        return None
    if isinstance(a, np.ndarray):
        return C.c_void_p(a.ctypes.data)
    if isinstance(a, int):
        return C.c_void_p(a)
    return C.c_void_p(a.data_ptr())   # torch tensor


def _np(a, dt):
    return np.ascontiguousarray(a, dtype=dt)


class Csr:
    """Host CSR triple with the reference's dtypes."""

    def __init__(self, nrow, ncol, indptr, indices, data):
        self.nrow, self.ncol = int(nrow), int(ncol)
        self.indptr, self.indices, self.data = _np(indptr, I32), _np(indices, I32), _np(data, F64)
        self.nnz = int(self.indices.size)

    def view(self):
        return CsrView(self.nrow, self.ncol, self.nnz, self.indptr.ctypes.data, self.indices.ctypes.data,
                       self.data.ctypes.data)


def dev_view(nrow, ncol, nnz, indptr, indices, data):
    """spmm_csr over device arrays (torch tensors or raw pointers)."""
    return CsrView(int(nrow), int(ncol), int(nnz), _p(indptr).value, _p(indices).value if nnz else None,
                   _p(data).value if nnz else None)


class Context:
    def __init__(self, device=0):
        self._h = C.c_void_p()
        _chk(lib().spmm_ctx_create(C.c_int(device), C.byref(self._h)))

    def close(self):
        if self._h:
            lib().spmm_ctx_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def sync(self):
        _chk(lib().spmm_ctx_sync(self._h))

    def set_stream(self, stream_ptr):
        _chk(lib().spmm_ctx_set_stream(self._h, C.c_void_p(stream_ptr)))

    def profile(self, enable=True):
        _chk(lib().spmm_ctx_profile(self._h, C.c_int(1 if enable else 0)))

    def phase_ms(self, phase):
        ms, n = C.c_double(), C.c_int()
        _chk(lib().spmm_ctx_phase_ms(self._h, C.c_int(phase), C.byref(ms), C.byref(n)))
        return float(ms.value), int(n.value)

    def debug_stamps(self, reset=True):
        out = np.zeros(32, np.uint64)
        _chk(lib().spmm_debug_stamps(self._h, _p(out), C.c_int(1 if reset else 0)))
        return out

    def forget_plan(self):
        """Drop everything kept between the calls of a product (the next call rebuilds it from its own arrays)."""
        _chk(lib().spmm_ctx_forget_plan(self._h))

    def sort_limit(self, b_ncol):
        """Rows with at most this many products take the sorting kernels for a right-hand matrix of b_ncol columns."""
        lib().spmm_spgemm_sort_limit.restype = C.c_int32
        return int(lib().spmm_spgemm_sort_limit(self._h, C.c_int32(int(b_ncol))))

    def plan_info(self, b_ncol):
        """(shift, nbase, dense_min_products) the context uses for a right-hand matrix of b_ncol columns."""
        sh, nb, dm = C.c_int32(), C.c_int32(), C.c_int64()
        _chk(lib().spmm_spgemm_plan_info(self._h, C.c_int32(int(b_ncol)), C.byref(sh), C.byref(nb), C.byref(dm)))
        return int(sh.value), int(nb.value), int(dm.value)

    def slab_info(self):
        """(windows, windows with entries of B = launches per call, pieces, widest window) of the slab order the last
        symbolic call built."""
        nw, nl, np_, wd = C.c_int32(), C.c_int32(), C.c_int64(), C.c_int32()
        _chk(lib().spmm_spgemm_slab_info(self._h, C.byref(nw), C.byref(nl), C.byref(np_), C.byref(wd)))
        return int(nw.value), int(nl.value), int(np_.value), int(wd.value)

    def light_max(self, b_ncol):
        """Window rows with at most this many products take the 256-thread cursor workgroups."""
        v = C.c_int64()
        _chk(lib().spmm_spgemm_light_max(self._h, C.c_int32(int(b_ncol)), C.byref(v)))
        return int(v.value)

    def device_allocs(self):
        lib().spmm_ctx_device_allocs.restype = C.c_longlong
        return int(lib().spmm_ctx_device_allocs(self._h))

    def last_kernel_ms(self):
        return float(lib().spmm_last_kernel_ms(self._h))

    # ---- MatrixMarket text (serial_newblock_clock.cpp:47-104) ---------------
    def mtx_parse(self, text):
        """bytes of a MatrixMarket coordinate file -> (nrow, ncol, row, col), zero-based int32 arrays;
        the header is read on the host, the entries are converted on the GPU."""
        buf = bytes(text)
        nrow, ncol, nnz, off, per = C.c_int32(), C.c_int32(), C.c_int64(), C.c_size_t(), C.c_int32()
        _chk(lib().spmm_mtx_header(buf, C.c_size_t(len(buf)), C.byref(nrow), C.byref(ncol), C.byref(nnz), C.byref(off), C.byref(per)))
        row, col = np.empty(nnz.value, I32), np.empty(nnz.value, I32)
        _chk(lib().spmm_mtx_entries(self._h, buf, C.c_size_t(len(buf)), off, per, nnz, _p(row), _p(col)))
        return nrow.value, ncol.value, row, col

    def mtx_parse_values(self, text):
        """bytes of a MatrixMarket coordinate file with values -> (nrow, ncol, row, col, val)."""
        buf = bytes(text)
        nrow, ncol, nnz, off, per = C.c_int32(), C.c_int32(), C.c_int64(), C.c_size_t(), C.c_int32()
        _chk(lib().spmm_mtx_header(buf, C.c_size_t(len(buf)), C.byref(nrow), C.byref(ncol), C.byref(nnz), C.byref(off), C.byref(per)))
        if per.value != 3:
            raise SpmmError(-1, "mtx: the entries have no value token")
        row, col, val = np.empty(nnz.value, I32), np.empty(nnz.value, I32), np.empty(nnz.value, F64)
        _chk(lib().spmm_mtx_entries_values(self._h, buf, C.c_size_t(len(buf)), off, nnz, _p(row), _p(col), _p(val)))
        return nrow.value, ncol.value, row, col, val

    def mtx_entries_dev(self, text_ptr, nbytes, data_offset, per, nnz, row_ptr, col_ptr):
        _chk(lib().spmm_mtx_entries_dev(self._h, C.c_void_p(text_ptr), C.c_size_t(nbytes), C.c_size_t(data_offset), C.c_int32(per),
                                        C.c_int64(nnz), C.c_void_p(row_ptr), C.c_void_p(col_ptr)))

    # ---- CSR build (serial_newblock_clock.cpp:78-112) ---------------------
    def csr_from_coo(self, nrow, ncol, row, col):
        row, col = _np(row, I32), _np(col, I32)
        nnz = row.size
        indptr, indices, data = np.empty(nrow + 1, I32), np.empty(nnz, I32), np.empty(nnz, F64)
        _chk(lib().spmm_csr_from_coo(self._h, C.c_int32(nrow), C.c_int32(ncol), C.c_int32(nnz), _p(row), _p(col),
                                     _p(indptr), _p(indices), _p(data)))
        return Csr(nrow, ncol, indptr, indices, data)

    def csr_from_coo_dev(self, nrow, ncol, nnz, row, col, indptr, indices, data):
        _chk(lib().spmm_csr_from_coo_dev(self._h, C.c_int32(nrow), C.c_int32(ncol), C.c_int32(nnz), _p(row), _p(col),
                                         _p(indptr), _p(indices), _p(data)))

    def csr_from_coo_dev_noalloc(self, nrow, ncol, nnz, row, col, indptr, indices, data):
        """The allocation-free variant (static device scratch, at most 2^20 entries and rows)."""
        _chk(lib().spmm_csr_from_coo_dev_noalloc(self._h, C.c_int32(nrow), C.c_int32(ncol), C.c_int32(nnz), _p(row), _p(col),
                                                 _p(indptr), _p(indices), _p(data)))

    # ---- format-generation pipeline (serial_newblock_clock.cpp:221-499) ---------
    def csr_from_coo_opts(self, nrow, ncol, row, col, val=None, symmetric=False, sum_duplicates=False):
        """CSR build with caller values, symmetric expansion and duplicate summation (options the reference's
        loader does not have, serial_newblock_clock.cpp:84, 96-98)."""
        row, col = _np(row, I32), _np(col, I32)
        nnz = row.size
        v = None if val is None else _np(val, F64)
        cap = nnz * (2 if symmetric else 1)
        ip, ix, d = np.empty(nrow + 1, I32), np.empty(max(cap, 1), I32), np.empty(max(cap, 1), F64)
        n_out = C.c_int32()
        flags = (1 if symmetric else 0) | (2 if sum_duplicates else 0)
        _chk(lib().spmm_csr_from_coo_opts(self._h, C.c_int32(nrow), C.c_int32(ncol), C.c_int32(nnz), _p(row), _p(col),
                                          None if v is None else _p(v), C.c_int32(flags), C.c_int32(cap), _p(ip), _p(ix), _p(d),
                                          C.byref(n_out)))
        return Csr(nrow, ncol, ip, ix[:n_out.value].copy(), d[:n_out.value].copy())

    def trace_formats(self, A, sect=2048, threshold=65536, cap_blocks=None, guard=0):
        """Returns a dict with every array the pipeline produces (names as in spmm_trace).  guard > 0 (tests): every
        output array gets that many extra elements behind its worst-case size, filled with a sentinel, and the dict
        also holds them as "_raw" for the caller to check that nothing wrote there."""
        nrow, nnz = A.nrow, A.nnz
        cap = cap_blocks if cap_blocks is not None else max(1, min(nrow + 1, 1 << 20))
        sizes = dict(seq_bitmap=(nrow, I32), r_indptr=(nrow + 1, I32), r_indices=(nnz, I32), r_data=(nnz, F64),
                     bsize_list=(cap + 1, I32), seq_offset=(cap + 1, I32), panel_ptr=(cap + 1, I32),
                     panel_list=(nrow // 8 + 3 * cap + 8, I32), spv8_list=(nrow // 8 + 3 * cap + 8, I32), out_seq_all=(nrow, I32),
                     b_indptr=(nrow + cap, I32), b_colidx_orig=(nnz, I32), b_local=(nnz, I32), b_data=(nnz, F64), keys=(nnz, I32),
                     seq_input=(nnz, I32), rseq=(nrow, I32))
        raw = {k: (np.empty(n, t) if guard == 0 else np.full(n + guard, -77, t)) for k, (n, t) in sizes.items()}
        a = {k: raw[k][:sizes[k][0]] for k in sizes}
        o = TraceOut()
        for k, v in a.items():
            setattr(o, k, v.ctypes.data)
        o.cap_blocks = cap
        va = A.view()
        _chk(lib().spmm_trace_formats(self._h, C.byref(va), C.c_int32(sect), C.c_int32(threshold), C.byref(o)))
        b, nk = o.bnum, o.n_keys
        out = dict(a)
        # (views of the worst-case buffers, not copies: keys and seq_input alone are hundreds of megabytes at scale 22)
        out.update(bnum=b, bsize_list=a["bsize_list"][:b + 1], seq_offset=a["seq_offset"][:b + 1],
                   panel_ptr=a["panel_ptr"][:b + 1], panel_list=a["panel_list"][:o.n_panel_entries],
                   spv8_list=a["spv8_list"][:o.n_spv8], b_indptr=a["b_indptr"][:nrow + b],
                   keys=a["keys"][:nk], seq_input=a["seq_input"][:nk])
        if guard:
            out["_raw"] = raw
        return out

    @staticmethod
    def _trace_view(T):
        keep = {k: _np(T[k], F64 if k == "b_data" else I32) for k in
                ("seq_bitmap", "r_indptr", "bsize_list", "seq_offset", "panel_ptr", "panel_list", "spv8_list", "out_seq_all",
                 "b_indptr", "b_local", "b_data", "keys")}
        o = TraceOut()
        for k, v in keep.items():
            setattr(o, k, v.ctypes.data)
        o.bnum, o.n_keys = int(T["bnum"]), int(keep["keys"].size)
        o.n_panel_entries, o.n_spv8 = int(keep["panel_list"].size), int(keep["spv8_list"].size)
        return o, keep

    def spv8_spmv(self, T, ncol, x):
        """y = A x read from the packed formats in T (the dict trace_formats returns); x and y are in the
        original numbering of the matrix."""
        o, keep = self._trace_view(T)
        nrow, nnz = keep["seq_bitmap"].size, keep["b_local"].size
        x = _np(x, F64)
        assert x.size == ncol
        y = np.empty(nrow, F64)
        _chk(lib().spmm_spv8_spmv(self._h, C.c_int32(nrow), C.c_int32(ncol), C.c_int32(nnz), C.byref(o), _p(x), _p(y)))
        return y

    def spv8_spmm(self, T, ncol, X):
        """Y = A X for the columns of X (ncol x nvec, row-major) read from the packed formats in T."""
        o, keep = self._trace_view(T)
        nrow, nnz = keep["seq_bitmap"].size, keep["b_local"].size
        X = np.ascontiguousarray(X, dtype=F64)
        assert X.ndim == 2 and X.shape[0] == ncol
        Y = np.empty((nrow, X.shape[1]), F64)
        _chk(lib().spmm_spv8_spmm(self._h, C.c_int32(nrow), C.c_int32(ncol), C.c_int32(nnz), C.byref(o), _p(X),
                                  C.c_int32(X.shape[1]), _p(Y)))
        return Y

    # ---- the stages one by one (host arrays) ---------------------------------------
    def row_keys(self, A, sect=2048):
        """The bucket of every row (bitmap.h:123-146)."""
        keys = np.empty(A.nrow, I32)
        va = A.view()
        _chk(lib().spmm_row_keys(self._h, C.byref(va), C.c_int32(sect), _p(keys)))
        return keys

    def bitmap_reorder(self, A, sect=2048):
        """a3 (bitmap.h:108-170) -> (seq, Csr of the reordered matrix)."""
        seq = np.empty(A.nrow, I32)
        ip, ix, d = np.empty(A.nrow + 1, I32), np.empty(A.nnz, I32), np.empty(A.nnz, F64)
        va = A.view()
        _chk(lib().spmm_bitmap_reorder(self._h, C.byref(va), C.c_int32(sect), _p(seq), _p(ip), _p(ix), _p(d)))
        return seq, Csr(A.nrow, A.ncol, ip, ix, d)

    def gen_blocks(self, R, threshold=65536, cap=None):
        """a4 (transmat.h:334-376) -> bsize_list (bnum + 1 entries)."""
        cap = cap if cap is not None else max(2, min(R.nrow + 2, 1 << 20))
        bsize, bnum = np.empty(cap, I32), C.c_int32()
        vr = R.view()
        _chk(lib().spmm_gen_blocks(self._h, C.byref(vr), C.c_int32(threshold), _p(bsize), C.c_int32(cap), C.byref(bnum)))
        return bsize[:bnum.value + 1].copy()

    def panels_sort_pack(self, R, bsize_list):
        """a5-a7 (v8sort.h:49-73, 152-232; serial_newblock_clock.cpp:336-399)."""
        bsize_list = _np(bsize_list, I32)
        b, n, z = bsize_list.size - 1, R.nrow, R.nnz
        a = dict(panel_ptr=np.empty(b + 1, I32), panel_list=np.empty(n // 8 + 3 * b + 8, I32),
                 spv8_list=np.empty(n // 8 + 3 * b + 8, I32), out_seq_all=np.empty(n, I32), b_indptr=np.empty(n + b, I32),
                 b_colidx_orig=np.empty(z, I32), b_data=np.empty(z, F64))
        nl, ns = C.c_int32(), C.c_int32()
        vr = R.view()
        _chk(lib().spmm_panels_sort_pack(self._h, C.byref(vr), _p(bsize_list), C.c_int32(b), _p(a["panel_ptr"]),
                                         _p(a["panel_list"]), C.byref(nl), _p(a["spv8_list"]), C.byref(ns),
                                         _p(a["out_seq_all"]), _p(a["b_indptr"]), _p(a["b_colidx_orig"]), _p(a["b_data"])))
        a["panel_list"] = a["panel_list"][:nl.value].copy()
        a["spv8_list"] = a["spv8_list"][:ns.value].copy()
        return a

    def first_touch(self, R, bsize_list, b_colidx_orig):
        """a8 (serial_newblock_clock.cpp:187-204) -> (b_local, keys, seq_offset)."""
        bsize_list, stream = _np(bsize_list, I32), _np(b_colidx_orig, I32)
        b = bsize_list.size - 1
        local, keys, off, nk = np.empty(R.nnz, I32), np.empty(R.nnz, I32), np.empty(b + 1, I32), C.c_int32()
        _chk(lib().spmm_first_touch(self._h, C.c_int32(R.nrow), C.c_int32(R.ncol), C.c_int32(R.nnz), _p(R.indptr),
                                    _p(bsize_list), C.c_int32(b), _p(stream), _p(local), _p(keys), C.byref(nk), _p(off)))
        return local, keys[:nk.value].copy(), off

    def wbseq(self, nrow, seq_bitmap, out_seq_all, bsize_list, keys):
        """a9 (wbsort.h:58-67, 16-34, 81-95) -> (seq_input, rseq)."""
        seq_bitmap, out_seq_all, bsize_list, keys = (_np(x, I32) for x in (seq_bitmap, out_seq_all, bsize_list, keys))
        seq_input, rseq = np.empty(keys.size, I32), np.empty(nrow, I32)
        _chk(lib().spmm_wbseq(self._h, C.c_int32(nrow), _p(seq_bitmap), _p(out_seq_all), _p(bsize_list),
                              C.c_int32(bsize_list.size - 1), _p(keys), C.c_int32(keys.size), _p(seq_input), _p(rseq)))
        return seq_input, rseq

    # ---- product -------------------------------------------------------------
    def spgemm_symbolic(self, A, B=None):
        B = A if B is None else B
        va = A.view()
        vb = va if B is A else B.view()
        c_indptr = np.empty(A.nrow + 1, I64)
        nnz_c, prods = C.c_int64(), C.c_int64()
        _chk(lib().spmm_spgemm_symbolic(self._h, C.byref(va), C.byref(vb), _p(c_indptr), C.byref(nnz_c), C.byref(prods)))
        return c_indptr, int(nnz_c.value), int(prods.value)

    def spgemm_numeric(self, A, B, c_indptr, row_beg=0, row_end=None):
        B = A if B is None else B
        va = A.view()
        vb = va if B is A else B.view()
        row_end = A.nrow if row_end is None else row_end
        c_indptr = _np(c_indptr, I64)
        nz = int(c_indptr[row_end] - c_indptr[row_beg])
        ci, cd = np.empty(nz, I32), np.empty(nz, F64)
        _chk(lib().spmm_spgemm_numeric(self._h, C.byref(va), C.byref(vb), _p(c_indptr), C.c_int32(row_beg),
                                       C.c_int32(row_end), _p(ci), _p(cd)))
        return ci, cd

    def spgemm(self, A, B=None):
        c_indptr, nnz_c, prods = self.spgemm_symbolic(A, B)
        ci, cd = self.spgemm_numeric(A, B, c_indptr)
        return c_indptr, ci, cd

    def spgemm_symbolic_dev(self, va, vb, c_indptr):
        nnz_c, prods = C.c_int64(), C.c_int64()
        _chk(lib().spmm_spgemm_symbolic_dev(self._h, C.byref(va), C.byref(vb), _p(c_indptr), C.byref(nnz_c), C.byref(prods)))
        return int(nnz_c.value), int(prods.value)

    def spgemm_numeric_dev(self, va, vb, c_indptr, row_beg, row_end, c_indices, c_data):
        _chk(lib().spmm_spgemm_numeric_dev(self._h, C.byref(va), C.byref(vb), _p(c_indptr), C.c_int32(row_beg),
                                           C.c_int32(row_end), _p(c_indices), _p(c_data)))

    def row_products_dev(self, va, vb, prod):
        total = C.c_int64()
        _chk(lib().spmm_row_products_dev(self._h, C.byref(va), C.byref(vb), _p(prod), C.byref(total)))
        return int(total.value)

    def digest_dev(self, indices, data, n):
        out = np.zeros(2, np.uint64)
        _chk(lib().spmm_digest_dev(self._h, _p(indices), _p(data), C.c_int64(n), _p(out)))
        return int(out[0]), float(out[1:2].view(np.float64)[0])
