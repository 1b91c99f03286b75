"""GPU: the allocation-free CSR build (spmm_csr_from_coo_dev_noalloc, static device scratch) through the C ABI against
the existing entry point spmm_csr_from_coo_dev AND against the CPU oracle (the restatement of
serial_newblock_clock.cpp:78-112), bit for bit, on calls in a row that shrink and grow again, up to the entry point's
limit of 2^20 entries and 2^20 rows (the static array at its fullest); the context issues no hipMalloc for them; sizes
beyond the static scratch are refused."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _build(ctx, entry, n, ncol, row, col, dev):
    import torch
    nnz = int(row.size)
    t_row = torch.from_numpy(row).to(dev)
    t_col = torch.from_numpy(col).to(dev)
    ip = torch.full((n + 1,), -7, dtype=torch.int32, device=dev)
    ix = torch.full((max(nnz, 1),), -7, dtype=torch.int32, device=dev)
    d = torch.full((max(nnz, 1),), -7.0, dtype=torch.float64, device=dev)
    torch.cuda.synchronize()
    entry(n, ncol, nnz, t_row, t_col, ip, ix, d)
    ctx.sync()
    return ip.cpu().numpy(), ix[:nnz].cpu().numpy(), d[:nnz].cpu().numpy()


def test_csr_build_without_allocation_matches_the_allocating_entry_point():
    import torch
    from oracle import oracle as orc
    from spmm_amd import capi, synth
    dev = torch.device("cuda", 0)
    fresh = capi.Context(0)          # has never allocated a scratch slot
    ref = capi.Context(0)
    try:
        cases = []
        row, col = synth.rmat(16, 12, 9)                   # 2^16 rows, ~7.5e5 entries, hub rows
        cases.append((1 << 16, 1 << 16, row, col))
        rng = np.random.default_rng(2)
        m = 5000                                           # smaller, ragged, unsorted input order, duplicates, empty rows
        r2 = rng.integers(0, 700, m).astype(np.int32)
        c2 = rng.integers(0, 300, m).astype(np.int32)
        r2[:64] = 3
        cases.append((1000, 300, r2, c2))
        cases.append((5, 7, np.zeros(0, np.int32), np.zeros(0, np.int32)))
        # the limit itself: 2^20 entries over 2^20 rows (every scratch request at its largest), after the small calls
        big_n = 1 << 20
        cases.append((big_n, big_n, rng.integers(0, big_n, big_n).astype(np.int32), rng.integers(0, big_n, big_n).astype(np.int32)))
        before = fresh.device_allocs()
        assert before == 0
        for n, ncol, row, col in cases:
            got = _build(fresh, fresh.csr_from_coo_dev_noalloc, n, ncol, row, col, dev)
            want = _build(ref, ref.csr_from_coo_dev, n, ncol, row, col, dev)
            for g, w in zip(got, want):
                assert np.array_equal(g, w)
            assert np.all(got[2] == 1.0) and got[0][0] == 0 and got[0][-1] == row.size
            o_ip, o_ix, o_d = orc.csr_from_coo(n, row, col)       # the checker: the CPU oracle
            assert np.array_equal(got[0], o_ip) and np.array_equal(got[1], o_ix) and np.array_equal(got[2], o_d)
        assert fresh.device_allocs() == before             # nothing was allocated
        assert ref.device_allocs() > 0
        # a context that has allocated before behaves the same
        got = _build(ref, ref.csr_from_coo_dev_noalloc, *cases[1], dev)
        want = _build(ref, ref.csr_from_coo_dev, *cases[1], dev)
        for g, w in zip(got, want):
            assert np.array_equal(g, w)
        # beyond the static scratch: refused, nothing run
        big = (1 << 20) + 1
        z = torch.zeros(big, dtype=torch.int32, device=dev)
        ip = torch.zeros(11, dtype=torch.int32, device=dev)
        ix = torch.zeros(big, dtype=torch.int32, device=dev)
        d = torch.zeros(big, dtype=torch.float64, device=dev)
        with pytest.raises(capi.SpmmError) as e:
            fresh.csr_from_coo_dev_noalloc(10, 10, big, z, z, ip, ix, d)
        assert e.value.code == -3
        with pytest.raises(capi.SpmmError):
            fresh.csr_from_coo_dev_noalloc(10, 5, 4, z[:4] + 60, z[:4], ip, ix, d)    # ids out of range, as the existing entry point
    finally:
        fresh.close()
        ref.close()
