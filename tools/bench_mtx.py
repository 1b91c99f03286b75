#!/usr/bin/env python3
"""MatrixMarket text -> entries: GPU converter against the CPU readers on the same file.
usage: bench_mtx.py [scale ...]   (R-MAT, edge factor 16, seed 7, written as "row col" lines)"""
import json
import os
import sys
import tempfile
import time

import numpy as np
import pandas as pd
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import rmat_torch  # noqa: E402
from oracle import oracle as orc  # noqa: E402
from spmm_amd import capi  # noqa: E402


def main():
    scales = [int(a) for a in sys.argv[1:]] or [20]
    dev = torch.device("cuda", 0)
    ctx = capi.Context(0)
    for scale in scales:
        n = 1 << scale
        row, col = rmat_torch(scale, 16, 7, dev)
        r, c = row.cpu().numpy(), col.cpu().numpy()
        body = pd.DataFrame({"r": r + 1, "c": c + 1}).to_csv(sep=" ", header=False, index=False)
        text = (f"%%MatrixMarket matrix coordinate pattern general\n{n} {n} {r.size}\n" + body).encode()
        del body
        with tempfile.NamedTemporaryFile(suffix=".mtx", dir="/tmp", delete=False) as f:
            f.write(text)
            path = f.name
        # CPU: the oracle's C reader on the file (fgets + sscanf per line), one core
        t = time.perf_counter()
        _, _, row_c, col_c = orc.read_mtx(path)
        cpu_ms = (time.perf_counter() - t) * 1e3
        os.unlink(path)
        # GPU: host bytes in, host arrays out (PCIe both ways), and the kernels alone
        ctx.mtx_parse(text)   # warm-up: allocations
        t = time.perf_counter()
        nrow, ncol, row_g, col_g = ctx.mtx_parse(text)
        wall_ms = (time.perf_counter() - t) * 1e3
        kern_ms = ctx.last_kernel_ms()
        ok = bool(np.array_equal(row_g, r) and np.array_equal(col_g, c) and np.array_equal(row_c, r) and np.array_equal(col_c, c))
        alg = len(text) + 8 * r.size          # text read once + row and col written
        print(json.dumps(dict(case=f"R-MAT scale {scale} edge factor 16 seed 7", text_bytes=len(text), entries=int(r.size),
                              gpu_kernel_ms=round(kern_ms, 3), gpu_wall_ms_with_pcie=round(wall_ms, 1), cpu_reader_ms=round(cpu_ms, 1),
                              gpu_equals_cpu_and_source=ok, algorithmic_gb_per_s=round(alg / (kern_ms * 1e-3) / 1e9, 1),
                              frac_of_hbm_peak=round(alg / (kern_ms * 1e-3) / 8e12, 4),
                              hbm_bytes_moved=2 * len(text) + 8 * r.size)), flush=True)
    ctx.close()


if __name__ == "__main__":
    main()
