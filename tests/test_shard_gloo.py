"""CPU, world sizes 2 and 4 over gloo: the multi-GPU path's sharding logic, called as bench.py calls it.  Each rank
builds the CSR of its even row slice (with the oracle standing in for the GPU CSR build), the slices are
all-gathered into B, rows of A are split by the cost model over the kernel classes (cost_balanced_bounds with the
row lengths), each rank computes its rows of C with the oracle from the offset view of A that bench.py passes
(indptr[r_lo:], whole indices), and the concatenation must equal the single-process product.  One case puts all
the work into one row, so that ranks receive no rows at all."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import oracle as orc
from spmm_amd import shard, synth


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


SORTED_MAX, DENSE_MIN, LIGHT_MAX = 8192, 65536, 16384      # the library's bin limits for a matrix of this width


def _graph(case):
    if case == "rmat":
        n = 1 << 10
        row, col = synth.rmat(10, 8, 4)
        return n, row, col
    # row 0 names 250 long rows, whose own entries name empty rows: all products (75 000, a dense-class row) sit in row 0
    n = 1024
    rng = np.random.default_rng(6)
    rows = [np.zeros(250, np.int64)]
    cols = [np.arange(1, 251)]
    for r in range(1, 251):
        c = np.sort(rng.choice(np.arange(251, n), 300, replace=False))
        rows.append(np.full(c.size, r))
        cols.append(c)
    row, col = np.concatenate(rows), np.concatenate(cols)
    order = np.lexsort((row, col))
    return n, row[order].astype(np.int32), col[order].astype(np.int32)


def _worker(rank, world, port, q, case):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        n, row, col = _graph(case)
        lo, hi = shard.even_row_slice(n, rank, world)
        m = (row >= lo) & (row < hi)
        ip_l, ix_l, _ = orc.csr_from_coo(hi - lo, row[m] - lo, col[m])
        indptr, indices = shard.allgather_csr_slices(torch.from_numpy(ip_l), torch.from_numpy(ix_l), n, rank, world, dist)
        ip, ix = indptr.numpy(), indices.numpy()
        fip, fix, fd = orc.csr_from_coo(n, row, col)
        assert np.array_equal(ip, fip) and np.array_equal(ix, fix)
        prod, total = orc.row_products(ip, ix, ip)
        prod_t = torch.from_numpy(prod)
        bounds = shard.cost_balanced_bounds(prod_t, world, SORTED_MAX, DENSE_MIN, max(LIGHT_MAX, SORTED_MAX), indptr[1:] - indptr[:-1])
        assert bounds[0] == 0 and bounds[-1] == n and all(a <= b for a, b in zip(bounds, bounds[1:]))
        cost = shard.row_costs(prod_t, SORTED_MAX, DENSE_MIN, max(LIGHT_MAX, SORTED_MAX), indptr[1:] - indptr[:-1])
        share = float(cost[bounds[rank]:bounds[rank + 1]].sum())
        assert abs(share - float(cost.sum()) / world) <= 2 * float(cost.max())          # balanced up to a row or two
        r0, r1 = bounds[rank], bounds[rank + 1]
        d = np.ones(ix.size)
        a_view = (ip[r0:], ix, d)                                 # what bench.py hands the library: offset row pointers, all entries
        c_ip = orc.spgemm_symbolic(r1 - r0, n, a_view[0][:r1 - r0 + 1], ix, ip, ix)
        ci, cd, p_local = orc.spgemm_numeric(n, (a_view[0][:r1 - r0 + 1], ix, d), (ip, ix, d), c_ip)
        assert p_local == prod[r0:r1].sum()
        # global row pointers: local + exclusive scan of the per-rank nnz(C) (8-byte all-gather)
        cnt = torch.tensor([int(c_ip[-1])], dtype=torch.int64)
        cnts = torch.empty(world, dtype=torch.int64)
        dist.all_gather_into_tensor(cnts, cnt)
        base = int(cnts[:rank].sum())
        q.put((rank, r0, r1, base, c_ip, ci, cd))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,case", [(2, "rmat"), (4, "rmat"), (4, "one_heavy_row")])
def test_sharded_product_equals_single_process(world, case):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q, case)) for r in range(world)]
    for p in procs:
        p.start()
    got = sorted([q.get(timeout=180) for _ in range(world)])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    n, row, col = _graph(case)
    ip, ix, d = orc.csr_from_coo(n, row, col)
    c_ip, c_ix, c_d = orc.spgemm(n, (ip, ix, d), (ip, ix, d))
    covered = 0
    for rank, r0, r1, base, l_ip, ci, cd in got:
        assert base == c_ip[r0]
        assert np.array_equal(l_ip + base, c_ip[r0:r1 + 1])
        assert np.array_equal(ci, c_ix[c_ip[r0]:c_ip[r1]]) and np.array_equal(cd, c_d[c_ip[r0]:c_ip[r1]])
        covered += r1 - r0
    assert covered == n
    if case == "one_heavy_row":
        assert any(r1 == r0 for _, r0, r1, *_ in got), "this case is meant to leave a rank without rows"


def test_bounds_edge_cases():
    assert shard.product_balanced_bounds(torch.zeros(0, dtype=torch.int64), 4) == [0, 0, 0, 0, 0]
    b = shard.product_balanced_bounds(torch.tensor([0, 0, 100, 0], dtype=torch.int64), 2)
    assert b[0] == 0 and b[-1] == 4 and b == sorted(b)
    assert shard.product_balanced_bounds(torch.tensor([5, 5, 5, 5], dtype=torch.int64), 1) == [0, 4]


def test_cost_balanced_bounds():
    prod = torch.tensor([300000, 300000, 20000, 20000, 20000, 20000, 100, 100, 100, 100] * 50, dtype=torch.int64)
    for world in (1, 2, 4, 8):
        b = shard.cost_balanced_bounds(prod, world, 8192, 262144)
        assert b[0] == 0 and b[-1] == prod.numel() and all(x <= y for x, y in zip(b, b[1:]))
        c = shard.row_costs(prod, 8192, 262144)
        cost = [float(c[b[r]:b[r + 1]].sum()) for r in range(world)]
        assert max(cost) - min(cost) <= 2 * float(c.max())          # balanced up to a row or two
    assert shard.cost_balanced_bounds(torch.zeros(0, dtype=torch.int64), 3, 8192, 262144) == [0, 0, 0, 0]
    # with row lengths, a dense row of many entries weighs more than one of the same products and few entries
    lens = torch.tensor([4000, 40, 8, 8, 8, 8, 2, 2, 2, 2] * 50, dtype=torch.int32)
    c = shard.row_costs(prod, 8192, 262144, 65535, lens)
    assert float(c[0]) > float(c[1]) and float(c[2]) == float(shard.row_costs(prod, 8192, 262144, 65535)[2])
    b = shard.cost_balanced_bounds(prod, 4, 8192, 262144, 65535, lens)
    cost = [float(c[b[r]:b[r + 1]].sum()) for r in range(4)]
    assert max(cost) - min(cost) <= 2 * float(c.max())
