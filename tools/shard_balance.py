#!/usr/bin/env python3
"""How evenly does bench.py's row partition spread the work?  One GPU runs the row shard of every rank of
an N-rank job in turn (symbolic + numeric, the same calls as bench.py's step) and reports the time of
each; the N-rank step time is the largest of them (the ranks exchange nothing during a step).
usage: shard_balance.py <scale | askubuntu | stanford> <world> [steps] [seed]
(a scale is an R-MAT graph of edge factor 16; seed 7 is the one the cost model was fitted on, any other is held out;
the two names are the SNAP-sized stand-ins of spmm_amd.synth)"""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import build_csr, rmat_torch  # noqa: E402
from spmm_amd import capi, shard  # noqa: E402


def main():
    graph, world = sys.argv[1], int(sys.argv[2])
    steps = int(sys.argv[3]) if len(sys.argv) > 3 else 2
    seed = int(sys.argv[4]) if len(sys.argv) > 4 else 7
    dev = torch.device("cuda", 0)
    ctx = capi.Context(0)
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    if graph in ("askubuntu", "stanford"):
        from spmm_amd import synth
        n, gen = (159316, synth.askubuntu_like) if graph == "askubuntu" else (281903, synth.stanford_like)
        r_h, c_h = gen()
        row, col = torch.from_numpy(r_h).to(dev), torch.from_numpy(c_h).to(dev)
        scale = graph
    else:
        scale = int(graph)
        n = 1 << scale
        row, col = rmat_torch(scale, 16, seed, dev)
    nnz = int(row.numel())
    indptr, indices, data = build_csr(ctx, n, n, row, col)
    del row, col
    vB = capi.dev_view(n, n, nnz, indptr, indices, data)
    prod = torch.empty(n, dtype=torch.int64, device=dev)
    P_total = ctx.row_products_dev(vB, vB, prod)
    esc_max = ctx.sort_limit(n)
    _, table_windows, dense_min = ctx.plan_info(n)
    bounds = shard.cost_balanced_bounds(prod, world, esc_max, dense_min, max(ctx.light_max(n), esc_max),
                                        indptr[1:] - indptr[:-1], table_windows)
    cap = int(96 * (1 << 30) / 12)
    c_ix = torch.empty(cap, dtype=torch.int32, device=dev)
    c_val = torch.empty(cap, dtype=torch.float64, device=dev)
    out = []
    for r in range(world):
        r_lo, r_hi = bounds[r], bounds[r + 1]
        rows_l = r_hi - r_lo
        vA = capi.dev_view(rows_l, n, nnz, indptr[r_lo:], indices, data)
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
        torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(steps):
            step()
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t) / steps * 1e3
        pl = prod[r_lo:r_hi]
        out.append(dict(rank=r, rows=rows_l, products=int(pl.sum()), ms=round(ms, 2),
                        dense_products=int(pl[pl >= dense_min].sum()), sorted_products=int(pl[pl <= esc_max].sum())))
    mx, mean = max(o["ms"] for o in out), sum(o["ms"] for o in out) / world
    print(json.dumps(dict(scale=scale, seed=seed, world=world, ranks=out, max_ms=mx, mean_ms=round(mean, 2),
                          balance=round(mean / mx, 3), projected_gflops=round(2.0 * P_total / (mx * 1e-3) / 1e9, 1))))


if __name__ == "__main__":
    main()
