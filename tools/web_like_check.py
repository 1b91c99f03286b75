"""Sanity check on a matrix of the size and rough shape of SNAP/web-Google (not available here): product through the
host-pointer entry points against the oracle."""
import sys, time, numpy as np
sys.path.insert(0, "/root/repo")
from spmm_amd import capi, synth
from oracle import oracle as orc
n, m = 916428, 5105039
rng = np.random.default_rng(3)
# web-like: out-degree geometric-ish, targets with a popular head
deg = rng.geometric(0.18, n); deg = (deg * (m / deg.sum())).astype(np.int64) + 1
row = np.repeat(np.arange(n), deg)[:m].astype(np.int32)
pop = (rng.pareto(1.2, row.size) * 50).astype(np.int64) % n
col = np.where(rng.random(row.size) < 0.3, pop, rng.integers(0, n, row.size)).astype(np.int32)
key = np.unique(col.astype(np.int64) * n + row)        # SuiteSparse order: by column, then row; no duplicates
col, row = (key // n).astype(np.int32), (key % n).astype(np.int32)
ctx = capi.Context(0)
A = ctx.csr_from_coo(n, n, row, col)
for it in range(3):
    t = time.perf_counter()
    c_ip, nnz_c, prods = ctx.spgemm_symbolic(A)
    t1 = time.perf_counter()
    ci, cd = ctx.spgemm_numeric(A, None, c_ip)
    t2 = time.perf_counter()
print("rows", n, "nnz", A.nnz, "products", prods, "nnzC", nnz_c, "symbolic wall ms %.1f numeric wall ms %.1f (host arrays, PCIe included)" % ((t1 - t) * 1e3, (t2 - t1) * 1e3))
ip, ix, d = A.indptr, A.indices, A.data
t = time.perf_counter()
o_ip = orc.spgemm_symbolic(n, n, ip, ix, ip, ix)
oi, od, _ = orc.spgemm_numeric(n, (ip, ix, d), (ip, ix, d), o_ip)
print("oracle ms %.1f" % ((time.perf_counter() - t) * 1e3), "equal", np.array_equal(c_ip, o_ip) and np.array_equal(ci, oi) and np.array_equal(cd, od))
