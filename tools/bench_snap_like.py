#!/usr/bin/env python3
"""C = A*A on graphs of the size and rough shape of the SNAP inputs the benchmark names (web-Google, sx-askubuntu; the
files are not available offline, spmm_amd.synth draws stand-ins): device-resident symbolic + numeric through the C ABI,
timed per step, the whole of C compared with the oracle bit for bit, and the oracle timed on one host core beside it.
One JSON line per graph.  usage: bench_snap_like.py [steps [graph]]"""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import oracle as orc  # noqa: E402
from spmm_amd import capi, synth  # noqa: E402


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
    dev = torch.device("cuda", 0)
    ctx = capi.Context(0)
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    only = sys.argv[2] if len(sys.argv) > 2 else None      # a graph name: that graph alone (for a kernel trace)
    for name, n, gen in (("web-Stanford-like", 281903, synth.stanford_like), ("web-Google-like", 916428, synth.web_like),
                         ("sx-askubuntu-like", 159316, synth.askubuntu_like)):
        if only and name != only:
            continue
        row, col = gen()
        ip, ix, d = orc.csr_from_coo(n, row, col)
        nnz = int(ix.size)
        t = time.perf_counter()
        o_ip = orc.spgemm_symbolic(n, n, ip, ix, ip, ix)
        oi, od, P = orc.spgemm_numeric(n, (ip, ix, d), (ip, ix, d), o_ip)
        cpu_s = time.perf_counter() - t
        indptr, indices, data = (torch.from_numpy(x).to(dev) for x in (ip, ix, d))
        v = capi.dev_view(n, n, nnz, indptr, indices, data)
        c_ip = torch.empty(n + 1, dtype=torch.int64, device=dev)
        nnz_c = int(o_ip[-1])
        c_ix = torch.empty(nnz_c, dtype=torch.int32, device=dev)
        c_val = torch.empty(nnz_c, dtype=torch.float64, device=dev)

        def step():
            got, _ = ctx.spgemm_symbolic_dev(v, v, c_ip)
            assert got == nnz_c
            ctx.spgemm_numeric_dev(v, v, c_ip, 0, n, c_ix, c_val)

        step()
        torch.cuda.synchronize()
        equal = (np.array_equal(c_ip.cpu().numpy(), o_ip) and np.array_equal(c_ix.cpu().numpy(), oi)
                 and np.array_equal(c_val.cpu().numpy(), od))
        ctx.profile(True)
        t = time.perf_counter()
        for _ in range(steps):
            step()
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t) / steps * 1e3
        ph = {k: round(ctx.phase_ms(p)[0] / steps, 3) for k, p in
              [("bins", capi.PH_BINS), ("window_sym", capi.PH_WINDOW_SYM), ("esc_sym", capi.PH_ESC_SYM),
               ("window_num", capi.PH_WINDOW_NUM), ("esc_num", capi.PH_ESC_NUM), ("dense_num", capi.PH_DENSE_NUM),
               ("light_num", capi.PH_LIGHT_NUM)]}
        ctx.profile(False)
        # algorithmic bytes of a step: symbolic 4 B per product + 12 B per entry of A; numeric 12 B per product,
        # 20 B per entry of A, 12 B per entry of C; row pointers of A and C in both
        algo = 16 * P + 32 * nnz + 12 * nnz_c + 28 * (n + 1)
        print(json.dumps(dict(graph=name, rows=n, nnz=nnz, products=P, nnz_c=nnz_c, ms_per_step=round(ms, 3),
                              gflops=round(2.0 * P / (ms * 1e-3) / 1e9, 2), phases_ms_per_step=ph,
                              algorithmic_bytes_per_step=algo, frac_of_hbm_peak=round(algo / (ms * 1e-3) / 8e12, 4),
                              equals_oracle=bool(equal),
                              cpu_oracle=dict(seconds=round(cpu_s, 2), gflops=round(2.0 * P / cpu_s / 1e9, 3), cores=1))),
              flush=True)
        assert equal, f"{name}: C differs from the oracle"
        del c_ix, c_val
    ctx.close()


if __name__ == "__main__":
    main()
