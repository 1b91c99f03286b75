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
    if sum(cnts_h) >= 2 ** 31 - 64:
        raise ValueError(f"the gathered matrix would hold {sum(cnts_h)} entries: row pointers are int32 (csr.h:10-13)")
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


# Cost model of a row shard, in ns per intermediate product and ns per row by the kernel class of the row, fitted
# to the times (symbolic + numeric, one MI355X, R-MAT scale 22) of the rows of every range of products taken as a
# matrix of their own (tools/class_cost.py; the sorted rows, whose keys are sorted in registers, from that and from a
# web-Google-sized graph: 13 ps per product and 0.2 ns per row).  The sorted rows are dear per product, the cursor rows per row.  A
# dense row pays per entry of its row of A as well (the bounds of every table window, gathered into the row's bound
# table and read again per window): 3.0 ns per entry next to 3.6 ps per product when the row lengths are given, a
# tenth more for the rows of more than 8192 entries that several workgroups share; 5.7 ps per product stands for
# both when they are not.  Only the ratios matter.
COST_NS = {"dense": 0.0057, "window": 0.0070, "light": 0.0078, "sorted": 0.0131}
ROW_NS = {"dense": 0.0, "window": 120.0, "light": 47.0, "sorted": 0.2}
DENSE_WITH_LENS_NS = 0.0036
DENSE_ENTRY_NS = 3.0
DENSE_ENTRY_WINDOW_POWER = 0.55   # measured: 8 shards at scale 24 (1024 windows) balance at 2.1 times the cost per entry of 256 windows
GIANT_ENTRIES = 8192      # rows of A above this are shared by several workgroups, each walking the whole table row
GIANT_FACTOR = 1.1
ANY_ROW_NS = 1.0          # row products and classification, paid by empty rows too


def row_costs(prod, sorted_max, dense_min, light_max=None, lens=None, windows=256):
    """Modelled cost (ns) of every row: rows with at most sorted_max products are sorted, rows with at least
    dense_min use dense accumulators, the rows between are cursor rows (the small workgroups up to light_max).
    lens: entries per row of A (optional); windows: table windows of B (a dense row reads the bounds of every window
    for every entry, runs of sparse windows excepted, so its cost per entry grows with them; fitted at 256)."""
    p = prod.to(torch.float64)
    w = torch.full_like(p, COST_NS["window"])
    r = torch.full_like(p, ROW_NS["window"])
    if light_max is not None:
        m = prod <= light_max
        w[m], r[m] = COST_NS["light"], ROW_NS["light"]
    m = prod <= sorted_max
    w[m], r[m] = COST_NS["sorted"], ROW_NS["sorted"]
    m = prod >= dense_min
    w[m], r[m] = COST_NS["dense"], ROW_NS["dense"]
    if lens is not None:
        g = torch.where(lens > GIANT_ENTRIES, GIANT_FACTOR, 1.0).to(torch.float64)
        w[m] = DENSE_WITH_LENS_NS * g[m]
        r[m] = lens.to(torch.float64)[m] * (DENSE_ENTRY_NS * (max(int(windows), 1) / 256.0) ** DENSE_ENTRY_WINDOW_POWER) * g[m]
    return p * w + r * (prod > 0).to(torch.float64) + ANY_ROW_NS


def cost_balanced_bounds(prod, world, sorted_max, dense_min, light_max=None, lens=None, windows=256):
    """Row boundaries like product_balanced_bounds, with every row weighted by the modelled cost of its kernel
    class (row_costs), so that the ranks finish together and not merely hold the same number of products."""
    n = prod.numel()
    if n == 0:
        return [0] * (world + 1)
    cum = torch.cumsum(row_costs(prod, sorted_max, dense_min, light_max, lens, windows), 0)
    total = float(cum[-1])
    bounds = [0]
    for r in range(1, world):
        target = torch.tensor([total * r / world], dtype=torch.float64, device=prod.device)
        b = int(torch.searchsorted(cum, target)[0])
        bounds.append(max(bounds[-1], min(b, n)))
    bounds.append(n)
    return bounds
