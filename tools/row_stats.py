#!/usr/bin/env python3
"""Shape of the rows of one kernel class of the bench matrix: entries of A per row, products, the share of rows with more
than 1024 entries, products that fall into the first table window.  usage: python tools/row_stats.py [scale]"""
import json, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from spmm_amd import capi
scale = int(sys.argv[1]) if len(sys.argv) > 1 else 22
dev = torch.device("cuda", 0)
ctx = capi.Context(0); ctx.set_stream(torch.cuda.current_stream().cuda_stream)
n = 1 << scale
row, col = bench.rmat_torch(scale, 16, 7, dev)
ip, ix, d = bench.build_csr(ctx, n, n, row, col)
nnz = int(ix.numel())
v = capi.dev_view(n, n, nnz, ip, ix, d)
prod = torch.empty(n, dtype=torch.int64, device=dev)
ctx.row_products_dev(v, v, prod)
shift, nbase, dmin = ctx.plan_info(n)
lmax = ctx.light_max(n)
lens = (ip[1:] - ip[:-1]).to(torch.int64)
# products of every row that fall into table window 0: sum over entries k of (number of entries of row k below 2^shift)
first = torch.searchsorted(ix.to(torch.int64) + (torch.repeat_interleave(torch.arange(n, device=dev), lens) << 32),
                           (torch.arange(n, device=dev, dtype=torch.int64) << 32) + (1 << shift)) - ip[:-1].to(torch.int64)
f_per_entry = first[ix.to(torch.int64)]
w0 = torch.zeros(n, dtype=torch.int64, device=dev).index_add_(0, torch.repeat_interleave(torch.arange(n, device=dev), lens), f_per_entry)
for name, m in (("cursor", (prod > lmax) & (prod < dmin)), ("light", (prod > 4096) & (prod <= lmax)), ("dense", prod >= dmin)):
    L, P, W = lens[m].cpu().numpy(), prod[m].cpu().numpy(), w0[m].cpu().numpy()
    big = L > 1024
    out = dict(cls=name, rows=int(m.sum()), products=int(P.sum()), entries=int(L.sum()),
               lenA_quantiles={q: int(np.quantile(L, q)) for q in (0.1, 0.5, 0.9, 0.99)},
               rows_over_1024_entries=int(big.sum()), their_products=int(P[big].sum()),
               products_in_window0=int(W.sum()), rows_with_window0_over_8192=int((W > 8192).sum()),
               products_of_window0_over_8192=int(W[W > 8192].sum()),
               table_loads_for_sums=int((L * (nbase + 1)).sum()))
    print(json.dumps(out))
