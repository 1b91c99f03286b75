#!/usr/bin/env python3
"""Time the format-generation pipeline (CSR build + spmm_trace_formats) on the GPU next to the CPU
oracle and, when oracle/_ref/libref.so was built, the compiled reference, on seeded R-MAT inputs.
Prints one JSON line per case.  Run on the GPU box: python tools/bench_preprocess.py [scale ...]"""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import oracle as orc  # noqa: E402
from oracle import ref  # noqa: E402
from spmm_amd import capi, synth  # noqa: E402


def main():
    scales = [int(a) for a in sys.argv[1:]] or [18, 20]
    ctx = capi.Context(0)
    for scale in scales:
        n = 1 << scale
        row, col = synth.rmat(scale, 16, 7, diagonal=True)
        ip, ix, d = orc.csr_from_coo(n, row, col)
        A = capi.Csr(n, n, ip, ix, d)
        ctx.csr_from_coo(n, n, row, col)            # warm-up (allocations)
        ctx.csr_from_coo(n, n, row, col)
        csr_ms = ctx.last_kernel_ms()
        ctx.trace_formats(A)
        ctx.profile(True)
        t0 = time.perf_counter()
        g = ctx.trace_formats(A)
        wall_ms = (time.perf_counter() - t0) * 1e3
        gpu_ms = ctx.last_kernel_ms()
        stages = {name: round(ctx.phase_ms(pid)[0], 3) for name, pid in
                  [("reorder", 7), ("blocks", 8), ("panels", 9), ("pack", 10), ("renumber", 11), ("wbseq", 12)]}
        ctx.profile(False)
        t0 = time.perf_counter()
        o = orc.trace_formats(n, ip, ix, d, sort_mode=1)
        cpu_ms = (time.perf_counter() - t0) * 1e3
        same = all(np.array_equal(g[k], o[k]) for k in ("seq_bitmap", "bsize_list", "seq_offset", "seq_input", "rseq"))
        out = dict(case=f"R-MAT scale {scale} edge factor 16 seed 7 with diagonal", rows=n, nnz=int(ix.size),
                   blocks=int(g["bnum"]), gpu_csr_build_kernel_ms=round(csr_ms, 3), gpu_trace_formats_kernel_ms=round(gpu_ms, 3),
                   gpu_trace_formats_wall_ms_with_pcie=round(wall_ms, 1), cpu_oracle_ms=round(cpu_ms, 1),
                   gpu_equals_oracle=bool(same), gpu_stage_ms=stages)
        if ref.available() and scale > 22:
            # the compiled reference does not survive this size (std::bad_array_new_length or a corrupted heap in its
            # own allocations, which takes the process down): it is not run
            out["cpu_reference_ms"] = None
        elif ref.available():
            ref.set_csr(n, ip, ix, d)
            t0 = time.perf_counter()
            ref.trace_formats(n, int(ix.size))
            out["cpu_reference_ms"] = round((time.perf_counter() - t0) * 1e3, 1)
        out["speedup_vs_cpu_oracle"] = round(cpu_ms / gpu_ms, 1)
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
