#!/usr/bin/env python3
"""y = A x and Y = A X (8 right-hand vectors) over the packed per-block formats (spmm_spv8_spmv, spmm_spv8_spmm): kernel
time against the bytes they have to move, with scipy's CSR product on one host core beside it.
usage: bench_spv8.py [scale ...]   (R-MAT with diagonal, edge factor 16, seed 7)"""
import json
import os
import sys
import time

import numpy as np
import scipy.sparse as sp

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spmm_amd import capi, synth  # noqa: E402


def main():
    scales = [int(a) for a in sys.argv[1:]] or [18, 20]
    ctx = capi.Context(0)
    for scale in scales:
        n = 1 << scale
        row, col = synth.rmat(scale, 16, 7, True)
        M = ctx.csr_from_coo(n, n, row, col)
        T = ctx.trace_formats(M)
        x = np.random.default_rng(1).integers(1, 9, n).astype(np.float64)
        ctx.spv8_spmv(T, n, x)
        y = ctx.spv8_spmv(T, n, x)
        ms = ctx.last_kernel_ms()
        S = sp.csr_matrix((M.data, M.indices, M.indptr), shape=(n, n))
        t = time.perf_counter()
        yc = S @ x
        cpu_ms = (time.perf_counter() - t) * 1e3
        nnz, nk = int(M.nnz), int(T["keys"].size)
        # values + local column numbers once, x gathered per block (read x and keys, write and read the
        # gathered copy), row offsets, the two permutations and y
        alg = 12 * nnz + 28 * nk + 4 * (n + T["bnum"]) + 8 * n + 8 * n
        print(json.dumps(dict(case=f"R-MAT scale {scale} edge factor 16 seed 7 with diagonal", rows=n, nnz=nnz, blocks=int(T["bnum"]),
                              keys=nk, rows_in_8_groups=int(np.sum(T["spv8_list"])), gpu_kernel_ms=round(ms, 3),
                              algorithmic_gb_per_s=round(alg / (ms * 1e-3) / 1e9, 1), frac_of_hbm_peak=round(alg / (ms * 1e-3) / 8e12, 4),
                              gflops=round(2.0 * nnz / (ms * 1e-3) / 1e9, 1), cpu_scipy_csr_ms_1_core=round(cpu_ms, 1),
                              gpu_equals_cpu=bool(np.array_equal(y, yc)))), flush=True)
        nv = 8
        X = np.random.default_rng(2).integers(1, 9, (n, nv)).astype(np.float64)
        ctx.spv8_spmm(T, n, X)
        Y = ctx.spv8_spmm(T, n, X)
        ms = ctx.last_kernel_ms()
        t = time.perf_counter()
        Yc = S @ X
        cpu_ms = (time.perf_counter() - t) * 1e3
        # A once; per vector: x and the gathered copy (read x, write and read the copy) and y
        alg = 12 * nnz + 4 * nk + 4 * (n + T["bnum"]) + 8 * n + nv * (24 * nk + 8 * n)
        print(json.dumps(dict(case=f"R-MAT scale {scale} edge factor 16 seed 7 with diagonal, {nv} right-hand vectors", rows=n, nnz=nnz,
                              gpu_kernel_ms=round(ms, 3), algorithmic_gb_per_s=round(alg / (ms * 1e-3) / 1e9, 1),
                              frac_of_hbm_peak=round(alg / (ms * 1e-3) / 8e12, 4), gflops=round(2.0 * nnz * nv / (ms * 1e-3) / 1e9, 1),
                              cpu_scipy_csr_ms_1_core=round(cpu_ms, 1), gpu_equals_cpu=bool(np.array_equal(Y, Yc)))), flush=True)
    ctx.close()


if __name__ == "__main__":
    main()
