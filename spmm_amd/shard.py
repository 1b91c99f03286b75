"""Row sharding of the product across the GPUs of one node (one process per GPU).

C = A*B is independent per row of A once every rank holds B, so the only exchange is the
assembly of B: each rank builds the CSR of an even slice of rows and the slices are all-gathered
(RCCL over xGMI when the tensors live on GPUs; the same code runs on gloo/CPU tensors in the
tests).  Rows of A are then split by intermediate-product count, not by row count: R-MAT rows
differ by seven orders of magnitude in work.
"""
import torch


def even_row_slice(n, rank, world):
    return rank * n // world, (rank + 1) * n // world


def allgather_csr_slices(indptr_local, indices_local, n, rank, world, dist):
    """indptr_local (rows+1, starting at 0) and indices_local of this rank's even row slice ->
    (indptr[n+1], indices[nnz]) of the whole matrix on every rank.  Slices are padded to the
    longest one because all_gather_into_tensor wants equal sizes."""
    dev = indices_local.device
    lo, hi = even_row_slice(n, rank, world)
    cnt = torch.tensor([indices_local.numel()], dtype=torch.int64, device=dev)
    cnts = torch.empty(world, dtype=torch.int64, device=dev)
    dist.all_gather_into_tensor(cnts, cnt)
    cnts_h = [int(x) for x in cnts.cpu().tolist()]
    mx = max(max(cnts_h), 1)
    pad = torch.zeros(mx, dtype=torch.int32, device=dev)
    pad[:indices_local.numel()] = indices_local
    rows_max = max(even_row_slice(n, r, world)[1] - even_row_slice(n, r, world)[0] for r in range(world))
    lens = torch.zeros(max(rows_max, 1), dtype=torch.int32, device=dev)
    lens[:hi - lo] = indptr_local[1:] - indptr_local[:-1]
    allix = torch.empty(world * mx, dtype=torch.int32, device=dev)
    alllen = torch.empty(world * max(rows_max, 1), dtype=torch.int32, device=dev)
    dist.all_gather_into_tensor(allix, pad)
    dist.all_gather_into_tensor(alllen, lens)
    indices = torch.cat([allix[r * mx: r * mx + cnts_h[r]] for r in range(world)])
    parts = []
    for r in range(world):
        a, b = even_row_slice(n, r, world)
        parts.append(alllen[r * max(rows_max, 1): r * max(rows_max, 1) + (b - a)])
    lens_all = torch.cat(parts)
    indptr = torch.zeros(n + 1, dtype=torch.int32, device=dev)
    indptr[1:] = torch.cumsum(lens_all.to(torch.int64), 0).to(torch.int32)
    return indptr, indices


def product_balanced_bounds(prod, world):
    """Row boundaries [b_0=0, ..., b_world=n] such that every rank's rows hold about the same
    number of intermediate products.  prod: int64 tensor of per-row product counts."""
    n = prod.numel()
    cum = torch.cumsum(prod, 0)
    total = int(cum[-1]) if n else 0
    bounds = [0]
    for r in range(1, world):
        target = torch.tensor([total * r // world], dtype=torch.int64, device=prod.device)
        b = int(torch.searchsorted(cum, target)[0])
        bounds.append(max(bounds[-1], min(b, n)))
    bounds.append(n)
    return bounds


# Cost model of a row shard, fitted to the measured times (symbolic + numeric) of the shards of 4- and
# 8-rank partitions of R-MAT scale 22 on one MI355X (tools/shard_balance.py; fit within 5 %): ns per
# intermediate product by the kernel class of its row, plus a cost per row (one workgroup each, whatever
# the class).  The sorted rows are an order of magnitude dearer per product than the window rows.
# Only the ratios matter.
COST_NS = {"dense": 0.0096, "window": 0.0073, "sorted": 0.102}
ROW_NS = 10.2


def cost_balanced_bounds(prod, world, sorted_max, dense_min):
    """Row boundaries like product_balanced_bounds, with every row's products weighted by the cost
    of its kernel class (rows with at most sorted_max products are sorted, rows with at least
    dense_min use dense accumulators) plus a fixed cost per non-empty row, so that the ranks finish
    together and not merely hold the same number of products."""
    n = prod.numel()
    if n == 0:
        return [0] * (world + 1)
    p = prod.to(torch.float64)
    w = torch.full_like(p, COST_NS["window"])
    w[prod <= sorted_max] = COST_NS["sorted"]
    w[prod >= dense_min] = COST_NS["dense"]
    cum = torch.cumsum(p * w + ROW_NS * (prod > 0).to(torch.float64), 0)
    total = float(cum[-1])
    bounds = [0]
    for r in range(1, world):
        target = torch.tensor([total * r / world], dtype=torch.float64, device=prod.device)
        b = int(torch.searchsorted(cum, target)[0])
        bounds.append(max(bounds[-1], min(b, n)))
    bounds.append(n)
    return bounds
