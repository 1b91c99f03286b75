"""GPU, full size: C = A*A on the bench matrices (R-MAT scale 22: 6.5e7 entries, 1.5e11 products; scale 24, the largest
configuration: 2.6e8 entries, 1.0e12 products, 5.2e11 entries of C, one pass) through the C ABI,
checked through properties that do not need C from the CPU: the row counts of the symbolic phase are what the
numeric phase writes; every row of C is strictly increasing in its column ids; with all values 1.0 the values of C
add up to the number of intermediate products, and with all values 2.0 to four times that (both exact in fp64);
two runs give the same column ids; a sample of rows spread over every kernel class matches the oracle bit for bit."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

# scale -> (entries of C per numeric call, second pass with all values 2.0, rows checked against the oracle)
CASES = {22: (1 << 30, True, 48), 24: (3 << 30, False, 16)}


def _blocks(c_ip_h, rows, cap):
    out, r0 = [], 0
    while r0 < rows:
        r1 = min(int(np.searchsorted(c_ip_h, c_ip_h[r0] + cap, side="right")) - 1, rows)
        assert r1 > r0
        out.append((r0, r1))
        r0 = r1
    return out


@pytest.mark.parametrize("SCALE", sorted(CASES))
def test_product_properties_at_full_size(SCALE):
    import torch
    BLOCK_ENTRIES, second_pass, n_picks = CASES[SCALE]
    from bench import build_csr, rmat_torch
    from oracle import oracle as orc
    from spmm_amd import capi

    dev = torch.device("cuda", 0)
    ctx = capi.Context(0)
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    n = 1 << SCALE
    row, col = rmat_torch(SCALE, 16, 7, dev)
    nnz = int(row.numel())
    indptr, indices, data = build_csr(ctx, n, n, row, col)
    del row, col
    v = capi.dev_view(n, n, nnz, indptr, indices, data)
    prod = torch.empty(n, dtype=torch.int64, device=dev)
    P = ctx.row_products_dev(v, v, prod)
    assert P == int(prod.sum()) and P > 10 ** 11

    c_ip = torch.empty(n + 1, dtype=torch.int64, device=dev)
    nnz_c, _ = ctx.spgemm_symbolic_dev(v, v, c_ip)
    c_ip_h = c_ip.cpu().numpy()
    assert c_ip_h[0] == 0 and c_ip_h[-1] == nnz_c and np.all(np.diff(c_ip_h) >= 0)
    prod_h = prod.cpu().numpy()
    assert np.all(np.diff(c_ip_h) <= np.minimum(prod_h, n))                 # a row of C holds no more than its products
    assert np.array_equal(np.diff(c_ip_h) == 0, prod_h == 0)

    c_ix = torch.empty(BLOCK_ENTRIES, dtype=torch.int32, device=dev)
    c_val = torch.empty(BLOCK_ENTRIES, dtype=torch.float64, device=dev)
    blocks = _blocks(c_ip_h, n, BLOCK_ENTRIES)

    def run(check_order):
        total, hashes = 0.0, []
        for b0, b1 in blocks:
            ctx.spgemm_numeric_dev(v, v, c_ip, b0, b1, c_ix, c_val)
            m = int(c_ip_h[b1] - c_ip_h[b0])
            h, s = ctx.digest_dev(c_ix, c_val, m)
            total += s
            hashes.append(h)
            if check_order and m > 1:
                rising = c_ix[1:m] > c_ix[:m - 1]
                starts = (c_ip[b0 + 1:b1] - c_ip[b0])                        # first entry of every later row
                starts = starts[(starts > 0) & (starts < m)]
                rising[starts - 1] = True
                assert bool(rising.all()), f"rows {b0}..{b1}: column ids not strictly increasing inside a row"
                del rising, starts
                assert int(c_ix[:m].min()) >= 0 and int(c_ix[:m].max()) < n
        return total, hashes

    total1, hashes1 = run(True)
    assert total1 == float(P)                                                # all ones: sum of C = products
    if second_pass:
        data.fill_(2.0)
        total2, hashes2 = run(False)
        assert total2 == 4.0 * float(P)                                      # linear in both factors
        assert hashes1 == hashes2                                            # the structure does not depend on values or run
        data.fill_(1.0)

    # a sample of rows of every class against the oracle: the heaviest row, and rows spread evenly over log(products)
    ip_h, ix_h = indptr.cpu().numpy(), indices.cpu().numpy()
    d_h = np.ones(nnz, np.float64)
    order = np.argsort(prod_h, kind="stable")
    order = order[prod_h[order] > 0]
    logp = np.log2(prod_h[order].astype(np.float64))
    picks = {int(order[-1])}
    for t in np.linspace(logp[0], min(logp[-1], 22.5), n_picks):
        picks.add(int(order[min(int(np.searchsorted(logp, t)), order.size - 1)]))
    for r in sorted(picks):
        a = (np.array([0, ip_h[r + 1] - ip_h[r]], np.int32), ix_h[ip_h[r]:ip_h[r + 1]], d_h[ip_h[r]:ip_h[r + 1]])
        o_ip = orc.spgemm_symbolic(1, n, a[0], a[1], ip_h, ix_h)
        oi, od, _ = orc.spgemm_numeric(n, a, (ip_h, ix_h, d_h), o_ip)
        assert int(o_ip[1]) == int(c_ip_h[r + 1] - c_ip_h[r]), f"row {r}: symbolic count"
        ctx.spgemm_numeric_dev(v, v, c_ip, r, r + 1, c_ix, c_val)
        k = int(o_ip[1])
        assert np.array_equal(c_ix[:k].cpu().numpy(), oi), f"row {r} ({prod_h[r]} products): column ids"
        assert np.array_equal(c_val[:k].cpu().numpy(), od), f"row {r} ({prod_h[r]} products): values"
    ctx.close()
