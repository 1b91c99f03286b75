"""CPU: the oracle against the fixtures made from the compiled reference
(tools/gen_fixtures.py) and, for the product, from scipy.sparse."""
import numpy as np
import pytest

from oracle import oracle as orc
from oracle import ref
from tests.helpers import case_input, golden_json, golden_npz, hx

PRE = ["tiny", "rmat12d", "rmat17d", "rmat18", "rmat20d"]


@pytest.mark.parametrize("name", PRE)
def test_preprocessing_matches_reference_digests(name):
    J = golden_json("pre_" + name)
    n, row, col = case_input(J)
    ip, ix, d = orc.csr_from_coo(n, row, col)
    assert hx(ip) == J["csr_indptr"] and hx(ix) == J["csr_indices"]
    for mode, pre in ((0, "ref_"), (1, "stable_")):
        o = orc.trace_formats(n, ip, ix, d, sort_mode=mode)
        assert o["bnum"] == J["bnum"]
        assert o["bsize_list"].tolist() == J["bsize_list"]
        assert o["seq_offset"].tolist() == J["seq_offset"]
        assert hx(o["seq_bitmap"]) == J["seq_bitmap"]
        assert hx(o["r_indptr"]) == J["r_indptr"] and hx(o["r_indices"]) == J["r_indices"]
        assert hx(o["panel_list"]) == J["panel_list"] and hx(o["spv8_list"]) == J["spv8_list"]
        assert hx(o["out_seq_all"]) == J[pre + "out_seq"]
        assert hx(o["b_colidx_orig"]) == J[pre + "b_colidx"]
        assert hx(o["b_local"]) == J[pre + "b_local"]
        assert hx(o["keys"]) == J[pre + "keys"]
        assert hx(o["seq_input"]) == J[pre + "seq_input"]
        assert hx(o["rseq"]) == J[pre + "rseq"]
        assert np.array_equal(np.sort(o["rseq"]), np.arange(n))


@pytest.mark.parametrize("name", ["tiny", "rmat12d"])
def test_preprocessing_full_arrays(name):
    J = golden_json("pre_" + name)
    Z = golden_npz("pre_" + name)
    n = J["n"]
    ip, ix, d = orc.csr_from_coo(n, Z["row"], Z["col"])
    assert np.array_equal(ip, Z["csr_indptr"]) and np.array_equal(ix, Z["csr_indices"])
    r_ip, r_ix, r_d, seq = orc.bitmap_reorder(n, ip, ix, d)
    assert np.array_equal(seq, Z["seq_bitmap"]) and np.array_equal(r_ip, Z["r_indptr"]) and np.array_equal(r_ix, Z["r_indices"])
    bs = orc.gen_blocks(n, r_ip, r_ix)
    assert np.array_equal(bs, Z["bsize_list"])
    o = orc.trace_formats(n, ip, ix, d, sort_mode=0)
    for k_g, k_o in [("ref_out_seq", "out_seq_all"), ("ref_b_colidx", "b_colidx_orig"), ("ref_b_local", "b_local"),
                     ("ref_keys", "keys"), ("ref_seq_input", "seq_input"), ("ref_rseq", "rseq"),
                     ("panel_list", "panel_list"), ("spv8_list", "spv8_list")]:
        assert np.array_equal(Z[k_g], o[k_o]), k_g


def test_stable_and_reference_orders_share_invariants():
    """Tie order differs between std::sort and the canonical stable order; everything
    that does not depend on ties must be identical, and 8-groups must hold equal lengths."""
    J = golden_json("pre_rmat17d")
    n, row, col = case_input(J)
    ip, ix, d = orc.csr_from_coo(n, row, col)
    o0 = orc.trace_formats(n, ip, ix, d, sort_mode=0)
    o1 = orc.trace_formats(n, ip, ix, d, sort_mode=1)
    for k in ("bsize_list", "seq_offset", "panel_list", "spv8_list", "b_indptr"):
        assert np.array_equal(o0[k], o1[k]), k
    bs, rl = o1["bsize_list"], np.diff(o1["r_indptr"])
    for b in range(o1["bnum"]):
        r0, r1 = bs[b], bs[b + 1]
        for o in (o0, o1):
            assert np.array_equal(np.sort(o["out_seq_all"][r0:r1]), np.arange(r1 - r0))
        assert np.array_equal(rl[r0:r1][o0["out_seq_all"][r0:r1]], rl[r0:r1][o1["out_seq_all"][r0:r1]])


def test_blocking_small_threshold_and_edge_cases():
    rng = np.random.default_rng(5)
    n = 300
    row, col = rng.integers(0, n, 2000), rng.integers(0, n, 2000)
    ip, ix, d = orc.csr_from_coo(n, row, col)
    bs = orc.gen_blocks(n, ip, ix, threshold=40)
    assert bs[0] == 0 and bs[-1] == n and np.all(np.diff(bs) >= 0)
    # every block but the last reaches the threshold exactly at its last row
    for b in range(len(bs) - 2):
        cols = ix[ip[bs[b]]:ip[bs[b + 1]]]
        assert len(np.unique(cols)) >= 40
        cols_short = ix[ip[bs[b]]:ip[bs[b + 1] - 1]]
        assert len(np.unique(cols_short)) < 40
    # empty matrix: one empty block, nothing else
    ip0 = np.zeros(5, np.int32)
    o = orc.trace_formats(4, ip0, np.zeros(0, np.int32), np.zeros(0), sort_mode=1)
    assert o["bnum"] == 1 and o["bsize_list"].tolist() == [0, 4] and o["seq_offset"].tolist() == [0, 0]
    assert o["spv8_list"].tolist() == [0] and np.array_equal(np.sort(o["rseq"]), np.arange(4))


def test_small_threshold_pipeline_is_consistent():
    """Many blocks (threshold 64): the pieces compose the way serial_newblock_clock.cpp:221-499 does."""
    n = 2048
    from spmm_amd import synth
    row, col = synth.rmat(11, 8, 21, True)
    ip, ix, d = orc.csr_from_coo(n, row, col)
    o = orc.trace_formats(n, ip, ix, d, threshold=64, sort_mode=1)
    assert o["bnum"] > 10
    bs = o["bsize_list"]
    tail = 0
    for b in range(o["bnum"]):
        r0, r1 = int(bs[b]), int(bs[b + 1])
        pl = orc.panel_list(o["r_indptr"], r0, r1)
        seq, spv8 = orc.panel_sort(o["r_indptr"], r0, r1, pl, mode=1)
        assert np.array_equal(seq, o["out_seq_all"][r0:r1])
        b_ip, b_ci, b_d = orc.gather_spv8(o["r_indptr"], o["r_indices"], o["r_data"], r0, r1, seq, pl, spv8)
        nz0, nz1 = o["r_indptr"][r0], o["r_indptr"][r1]
        assert np.array_equal(b_ci, o["b_colidx_orig"][nz0:nz1])
        local, keys = orc.first_touch(n, b_ci)
        assert np.array_equal(local, o["b_local"][nz0:nz1])
        assert np.array_equal(keys, o["keys"][tail:tail + keys.size])
        assert np.array_equal(keys[local], b_ci)
        tail += keys.size
        assert o["seq_offset"][b + 1] == tail


@pytest.mark.parametrize("name", ["tiny", "rmat10", "rmat12d", "rmat14", "rmat16"])
def test_product_matches_scipy_digests(name):
    J = golden_json("spgemm_" + name)
    n, row, col = case_input(J)
    ip, ix, d = orc.csr_from_coo(n, row, col)
    prod, total = orc.row_products(ip, ix, ip)
    assert total == J["products"]
    c_ip, c_ix, c_d = orc.spgemm(n, (ip, ix, d), (ip, ix, d))
    assert c_ip[-1] == J["nnzC"]
    assert hx(c_ip) == J["c_indptr"] and hx(c_ix) == J["c_indices"] and hx(c_d) == J["c_data"]
    assert c_d.sum() == J["products"]  # all-ones inputs: the values count paths


def test_product_rectangular_fp64():
    Z = golden_npz("spgemm_rect")
    a = (Z["a_indptr"], Z["a_indices"], Z["a_data"])
    b = (Z["b_indptr"], Z["b_indices"], Z["b_data"])
    c_ip, c_ix, c_d = orc.spgemm(500, a, b)
    assert np.array_equal(c_ip, Z["c_indptr"]) and np.array_equal(c_ix, Z["c_indices"])
    np.testing.assert_allclose(c_d, Z["c_data"], rtol=1e-12, atol=1e-14)
    # row-range form returns the same rows
    ci, cd, _ = orc.spgemm_numeric(500, a, b, c_ip, 100, 200)
    assert np.array_equal(ci, c_ix[c_ip[100]:c_ip[200]]) and np.array_equal(cd, c_d[c_ip[100]:c_ip[200]])


@pytest.mark.skipif(not ref.available(), reason="compiled reference only exists in the build container")
def test_oracle_equals_compiled_reference_on_fresh_input(tmp_path):
    from spmm_amd import synth
    row, col = synth.rmat(13, 6, 1234, True)
    n = 1 << 13
    p = tmp_path / "m.mtx"
    synth.write_mtx(str(p), n, row, col, pattern=False)   # 3-token lines: value column ignored
    nrow, ncol, ip, ix, d = ref.load(str(p))
    n2, c2, r2, cc2 = orc.read_mtx(str(p))
    oip, oix, od = orc.csr_from_coo(n2, r2, cc2)
    assert (n2, c2) == (nrow, ncol) and np.array_equal(oip, ip) and np.array_equal(oix, ix) and np.all(d == 1.0)
    t = ref.trace_formats(nrow, ix.size)
    o = orc.trace_formats(ncol, oip, oix, od, sort_mode=0)
    assert np.array_equal(o["seq_input"], t["seq_input"]) and np.array_equal(o["rseq"], t["rseq"])
    assert np.array_equal(o["bsize_list"], t["bsize_list"]) and np.array_equal(o["seq_offset"], t["seq_offset"])
