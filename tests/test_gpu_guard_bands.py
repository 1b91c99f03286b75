"""GPU: guard bands around the output arrays.  Every output buffer of the CSR build, of the numeric phase and of the
format pipeline is handed over with sentinel-filled elements in front of and behind the part the call may write; the
sentinels must survive and the part in between must equal the oracle.  (The reference's own overflow of this kind:
v8sort.h:157-160, a scratch array sized by block nnz and indexed by block rows.)"""
import numpy as np
import pytest

from oracle import oracle as orc
from spmm_amd import capi, synth

pytestmark = pytest.mark.gpu
G = 4096          # guard elements on either side
SENT = -77


def _guarded(n, dtype, dev):
    import torch
    t = torch.full((n + 2 * G,), SENT, dtype=dtype, device=dev)
    return t, t[G:G + n]


def _intact(t, n):
    return bool((t[:G] == SENT).all()) and bool((t[G + n:] == SENT).all())


@pytest.fixture(scope="module")
def ctx():
    c = capi.Context(0)
    yield c
    c.close()


def _inputs():
    rng = np.random.default_rng(12)
    yield "rmat14", 1 << 14, synth.rmat(14, 16, 5)
    # ragged: empty rows, one hub row, duplicates
    n, m = 5000, 60000
    row, col = rng.integers(0, n, m).astype(np.int32), rng.integers(0, n, m).astype(np.int32)
    row[:3000] = 17
    row[row % 7 == 3] = 4
    yield "ragged", n, (row, col)


@pytest.mark.parametrize("name,n,rc", list(_inputs()), ids=[c[0] for c in _inputs()])
def test_csr_build_and_product_stay_inside_their_buffers(ctx, name, n, rc):
    import torch
    dev = torch.device("cuda", 0)
    row, col = rc
    nnz = int(row.size)
    t_row, t_col = torch.from_numpy(row).to(dev), torch.from_numpy(col).to(dev)
    for entry in (ctx.csr_from_coo_dev, ctx.csr_from_coo_dev_noalloc):
        ip_t, ip = _guarded(n + 1, torch.int32, dev)
        ix_t, ix = _guarded(nnz, torch.int32, dev)
        d_t, d = _guarded(nnz, torch.float64, dev)
        torch.cuda.synchronize()
        entry(n, n, nnz, t_row, t_col, ip, ix, d)
        ctx.sync()
        assert _intact(ip_t, n + 1) and _intact(ix_t, nnz) and _intact(d_t, nnz)
        o_ip, o_ix, o_d = orc.csr_from_coo(n, row, col)
        assert np.array_equal(ip.cpu().numpy(), o_ip) and np.array_equal(ix.cpu().numpy(), o_ix) and np.array_equal(d.cpu().numpy(), o_d)
    # the product of the matrix with sorted rows (B's rows must ascend): the oracle's CSR of the (col, row)-sorted entries
    order = np.lexsort((row, col))
    o_ip, o_ix, o_d = orc.csr_from_coo(n, row[order], col[order])
    a_ip, a_ix, a_d = (torch.from_numpy(x).to(dev) for x in (o_ip, o_ix, o_d))
    v = capi.dev_view(n, n, nnz, a_ip, a_ix, a_d)
    cip_t, c_ip = _guarded(n + 1, torch.int64, dev)
    torch.cuda.synchronize()
    nnz_c, _ = ctx.spgemm_symbolic_dev(v, v, c_ip)
    ctx.sync()
    assert _intact(cip_t, n + 1)
    w_ip, w_ix, w_d = orc.spgemm(n, (o_ip, o_ix, o_d), (o_ip, o_ix, o_d))
    assert nnz_c == int(w_ip[-1]) and np.array_equal(c_ip.cpu().numpy(), w_ip)
    for r0, r1 in ((0, n), (n // 3, 2 * n // 3), (5, 6)):
        nz = int(w_ip[r1] - w_ip[r0])
        ci_t, ci = _guarded(nz, torch.int32, dev)
        cd_t, cd = _guarded(nz, torch.float64, dev)
        torch.cuda.synchronize()
        ctx.spgemm_numeric_dev(v, v, c_ip, r0, r1, ci, cd)
        ctx.sync()
        assert _intact(ci_t, nz) and _intact(cd_t, nz)
        assert np.array_equal(ci.cpu().numpy(), w_ix[w_ip[r0]:w_ip[r1]]) and np.array_equal(cd.cpu().numpy(), w_d[w_ip[r0]:w_ip[r1]])


@pytest.mark.parametrize("threshold", [65536, 64])
def test_format_pipeline_stays_inside_its_buffers(ctx, threshold):
    """Host arrays of spmm_trace_formats at their worst-case sizes plus a guard: with the reference's threshold (one
    block here) and with a tiny one (hundreds of blocks, blocks with fewer entries than rows)."""
    n = 1 << 12
    row, col = synth.rmat(12, 6, 3, diagonal=True)
    ip, ix, d = orc.csr_from_coo(n, row, col)
    T = ctx.trace_formats(capi.Csr(n, n, ip, ix, d), threshold=threshold, guard=G)
    o = orc.trace_formats(n, ip, ix, d, threshold=threshold, sort_mode=1)
    sizes = {k: v.size - G for k, v in T["_raw"].items()}
    for k, raw in T["_raw"].items():
        assert (raw[sizes[k]:] == SENT).all(), f"{k}: written behind its worst-case size"
    assert T["bnum"] == o["bnum"]
    for k in ("seq_bitmap", "r_indptr", "r_indices", "bsize_list", "seq_offset", "panel_list", "spv8_list", "out_seq_all", "b_indptr",
              "b_colidx_orig", "b_local", "keys", "seq_input", "rseq"):
        assert np.array_equal(T[k], o[k]), k
