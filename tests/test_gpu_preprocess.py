"""GPU: the format-generation pipeline (spmm_trace_formats) against the oracle in its stable
tie order and against the fixtures made from the compiled reference.  Every stage's output is
compared, bit for bit."""
import numpy as np
import pytest

from oracle import oracle as orc
from spmm_amd import capi, synth
from tests.helpers import case_input, golden_json, golden_npz, hx

pytestmark = pytest.mark.gpu

STAGES = ["seq_bitmap", "r_indptr", "r_indices", "r_data", "bsize_list", "panel_ptr", "panel_list", "spv8_list",
          "out_seq_all", "b_indptr", "b_colidx_orig", "b_data", "b_local", "keys", "seq_offset", "seq_input", "rseq"]


@pytest.fixture(scope="module")
def ctx():
    c = capi.Context(0)
    yield c
    c.close()


def _compare(g, o):
    assert g["bnum"] == o["bnum"]
    for k in STAGES:
        assert g[k].shape == o[k].shape, (k, g[k].shape, o[k].shape)
        if not np.array_equal(g[k], o[k]):
            bad = np.flatnonzero(g[k] != o[k])
            raise AssertionError(f"stage output {k} differs at {bad.size} of {g[k].size} positions, first {bad[:5]}: "
                                 f"gpu {g[k][bad[:5]]} oracle {o[k][bad[:5]]}")


@pytest.mark.parametrize("name", ["tiny", "rmat12d", "rmat17d", "rmat18", "rmat20d"])
def test_pipeline_matches_golden(ctx, name):
    J = golden_json("pre_" + name)
    n, row, col = case_input(J)
    ip, ix, d = orc.csr_from_coo(n, row, col)
    g = ctx.trace_formats(capi.Csr(n, n, ip, ix, d))
    assert g["bnum"] == J["bnum"]
    assert g["bsize_list"].tolist() == J["bsize_list"] and g["seq_offset"].tolist() == J["seq_offset"]
    assert hx(g["seq_bitmap"]) == J["seq_bitmap"]
    assert hx(g["r_indptr"]) == J["r_indptr"] and hx(g["r_indices"]) == J["r_indices"]
    assert hx(g["panel_list"]) == J["panel_list"] and hx(g["spv8_list"]) == J["spv8_list"]
    assert hx(g["out_seq_all"]) == J["stable_out_seq"]
    assert hx(g["b_colidx_orig"]) == J["stable_b_colidx"]
    assert hx(g["b_local"]) == J["stable_b_local"]
    assert hx(g["keys"]) == J["stable_keys"]
    assert hx(g["seq_input"]) == J["stable_seq_input"]
    assert hx(g["rseq"]) == J["stable_rseq"]
    assert np.all(g["b_data"] == 1.0)


@pytest.mark.parametrize("threshold,sect", [(64, 2048), (500, 16), (65536, 2048)])
def test_pipeline_matches_oracle_many_blocks(ctx, threshold, sect):
    n = 1 << 13
    row, col = synth.rmat(13, 8, 77, True)
    ip, ix, d = orc.csr_from_coo(n, row, col)
    rng = np.random.default_rng(1)
    d = rng.integers(1, 9, d.size).astype(np.float64)     # general values: the gathers must carry them
    o = orc.trace_formats(n, ip, ix, d, sect=sect, threshold=threshold, sort_mode=1)
    g = ctx.trace_formats(capi.Csr(n, n, ip, ix, d), sect=sect, threshold=threshold)
    _compare(g, o)


def test_pipeline_edge_cases(ctx):
    # empty rows, rows longer than 32 and groups of short rows, a row of length > 4096, no entries at all
    rng = np.random.default_rng(9)
    n = 3000
    rows = [np.full(5000, 7), rng.integers(0, n, 20000), np.repeat(np.arange(100, 400), 3)]
    cols = [rng.integers(0, n, 5000), rng.integers(0, n, 20000), rng.integers(0, n, 900)]
    row, col = np.concatenate(rows), np.concatenate(cols)
    keep = row % 5 != 0          # every fifth row empty
    row, col = row[keep], col[keep]
    order = np.lexsort((row, col))
    ip, ix, d = orc.csr_from_coo(n, row[order], col[order])
    for thr in (65536, 300):
        o = orc.trace_formats(n, ip, ix, d, threshold=thr, sort_mode=1)
        g = ctx.trace_formats(capi.Csr(n, n, ip, ix, d), threshold=thr)
        _compare(g, o)
    ip0 = np.zeros(9, np.int32)
    o = orc.trace_formats(8, ip0, np.zeros(0, np.int32), np.zeros(0), sort_mode=1)
    g = ctx.trace_formats(capi.Csr(8, 8, ip0, np.zeros(0, np.int32), np.zeros(0)))
    _compare(g, o)


def test_pipeline_block_capacity_is_an_error(ctx):
    n = 1 << 11
    row, col = synth.rmat(11, 8, 21, True)
    ip, ix, d = orc.csr_from_coo(n, row, col)
    with pytest.raises(capi.SpmmError) as e:
        ctx.trace_formats(capi.Csr(n, n, ip, ix, d), threshold=64, cap_blocks=5)
    assert e.value.code == -3
