#!/usr/bin/env python3
"""Cost of C = A*A by row class: for every range of intermediate products per row, the rows of A in that
range form a matrix A' and the phases of A'*A (symbolic + numeric) are timed.
usage: class_cost.py <scale> [edges...]   (default edges 0 512 4096 8192 16384 32768 65536 131072 262144 786432)"""
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import build_csr, rmat_torch  # noqa: E402
from spmm_amd import capi  # noqa: E402


def main():
    scale = int(sys.argv[1])
    edges = [int(x) for x in sys.argv[2:]] or [0, 512, 4096, 8192, 16384, 32768, 65536, 131072, 262144, 786432]
    edges.append(1 << 62)
    dev = torch.device("cuda", 0)
    ctx = capi.Context(0)
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    n = 1 << scale
    row, col = rmat_torch(scale, 16, 7, dev)
    nnz = int(row.numel())
    indptr, indices, data = build_csr(ctx, n, n, row, col)
    del row, col
    vB = capi.dev_view(n, n, nnz, indptr, indices, data)
    prod = torch.empty(n, dtype=torch.int64, device=dev)
    ctx.row_products_dev(vB, vB, prod)
    lens = (indptr[1:] - indptr[:-1]).to(torch.int64)
    cap = int(40 * (1 << 30) / 12)
    c_ix = torch.empty(cap, dtype=torch.int32, device=dev)
    c_val = torch.empty(cap, dtype=torch.float64, device=dev)
    names = [("bins", capi.PH_BINS), ("window_sym", capi.PH_WINDOW_SYM), ("esc_sym", capi.PH_ESC_SYM),
             ("window_num", capi.PH_WINDOW_NUM), ("esc_num", capi.PH_ESC_NUM), ("dense_num", capi.PH_DENSE_NUM),
             ("light_num", capi.PH_LIGHT_NUM)]
    for lo, hi in zip(edges[:-1], edges[1:]):
        sel = torch.nonzero((prod > lo) & (prod <= hi)).flatten()
        if sel.numel() == 0:
            continue
        l = lens[sel]
        ip = torch.zeros(sel.numel() + 1, dtype=torch.int32, device=dev)
        ip[1:] = torch.cumsum(l, 0).to(torch.int32)
        # entries of the selected rows, in order
        starts = indptr[sel].to(torch.int64)
        idx = torch.repeat_interleave(starts - ip[:-1].to(torch.int64), l) + torch.arange(int(ip[-1]), device=dev)
        ix = indices[idx].contiguous()
        d = torch.ones(ix.numel(), dtype=torch.float64, device=dev)
        rows_l = sel.numel()
        vA = capi.dev_view(rows_l, n, int(ix.numel()), ip, ix, d)
        c_ip = torch.empty(rows_l + 1, dtype=torch.int64, device=dev)

        def step():
            ctx.spgemm_symbolic_dev(vA, vB, c_ip)
            h = c_ip.cpu().numpy()
            r0 = 0
            while r0 < rows_l:
                r1 = min(int(np.searchsorted(h, h[r0] + cap, side="right")) - 1, rows_l)
                ctx.spgemm_numeric_dev(vA, vB, c_ip, r0, r1, c_ix, c_val)
                r0 = r1

        step()
        ctx.profile(True)
        torch.cuda.synchronize()
        step()
        torch.cuda.synchronize()
        ph = {k: round(ctx.phase_ms(p)[0], 2) for k, p in names}
        ctx.profile(False)
        P = int(prod[sel].sum())
        tot = sum(ph.values())
        print(json.dumps(dict(products_range=[lo, hi if hi < (1 << 62) else None], rows=rows_l, products=P,
                              entries=int(ix.numel()), phases_ms=ph, total_ms=round(tot, 2),
                              ps_per_product=round(tot * 1e9 / max(P, 1), 1), us_cu_per_row=round(tot * 1e3 * 256 / rows_l, 1))),
              flush=True)


if __name__ == "__main__":
    main()
