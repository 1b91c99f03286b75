"""CPU: the oracle under AddressSanitizer + UndefinedBehaviorSanitizer.

oracle/_asan/oracle_asan (make -C oracle asan) runs every stage of the restatement on heap buffers of exactly the
sizes the callers' contract names and prints a digest per output; the digests must equal those of the ordinary
liboracle.so, and the sanitizers must stay silent.  The overflow this kind of run finds in the reference itself:
v8sort.h:157-160 (panel-sort scratch sized by block nnz, indexed by block rows), which inputs with empty rows hit.
"""
import os
import subprocess

import numpy as np
import pytest

from oracle import oracle as orc
from spmm_amd import synth
from tests.helpers import case_input, golden_json, hx

ORACLE_DIR = os.path.dirname(os.path.abspath(orc.__file__))
BIN = os.path.join(ORACLE_DIR, "_asan", "oracle_asan")


@pytest.fixture(scope="module")
def asan_bin():
    subprocess.check_call(["make", "-s", "-C", ORACLE_DIR, "asan"])
    return BIN


def _cases():
    J = golden_json("pre_tiny")
    yield "tiny", case_input(J), 65536
    J = golden_json("pre_rmat12d")
    yield "rmat12d", case_input(J), 65536
    # empty rows, many blocks (threshold 64): blocks with fewer entries than rows, the case the reference overflows on
    row, col = synth.rmat(10, 3, 5)
    yield "rmat10_empty_rows_small_blocks", (1 << 10, row, col), 64
    # no entries at all, and a single entry
    yield "empty", (16, np.zeros(0, np.int32), np.zeros(0, np.int32)), 65536
    yield "one_entry", (9, np.array([8], np.int32), np.array([0], np.int32)), 65536


@pytest.mark.parametrize("name,inp,threshold", list(_cases()), ids=[c[0] for c in _cases()])
def test_oracle_is_clean_under_sanitizers(asan_bin, tmp_path, name, inp, threshold):
    n, row, col = inp
    row, col = np.ascontiguousarray(row, np.int32), np.ascontiguousarray(col, np.int32)
    path = tmp_path / "in.bin"
    with open(path, "wb") as f:
        np.array([n, n, row.size], np.int32).tofile(f)
        row.tofile(f)
        col.tofile(f)
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1", OMP_NUM_THREADS="2")
    r = subprocess.run([asan_bin, str(path), str(threshold)], capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, f"sanitizer run failed:\n{r.stderr[-4000:]}"
    assert "runtime error" not in r.stderr and "AddressSanitizer" not in r.stderr, r.stderr[-4000:]
    got = dict(line.split() for line in r.stdout.strip().splitlines())

    ip, ix, d = orc.csr_from_coo(n, row, col)
    want = {"csr_indptr": hx(ip), "csr_indices": hx(ix)}
    for mode, pre in ((0, "ref_"), (1, "stable_")):
        o = orc.trace_formats(n, ip, ix, d, threshold=threshold, sort_mode=mode)
        want[pre + "bnum"] = str(o["bnum"])
        for k, src in (("seq_bitmap", "seq_bitmap"), ("r_indptr", "r_indptr"), ("r_indices", "r_indices"), ("bsize_list", "bsize_list"),
                       ("seq_offset", "seq_offset"), ("panel_list", "panel_list"), ("spv8_list", "spv8_list"), ("out_seq", "out_seq_all"),
                       ("b_indptr", "b_indptr"), ("b_colidx", "b_colidx_orig"), ("b_local", "b_local"), ("keys", "keys"),
                       ("seq_input", "seq_input"), ("rseq", "rseq")):
            want[pre + k] = hx(o[src])
    M = (ip, ix, d)
    c_ip, c_ix, c_d = orc.spgemm(n, M, M)
    prod, _ = orc.row_products(ip, ix, ip)
    want.update(c_indptr=hx(c_ip), c_indices=hx(c_ix), c_data=hx(c_d), row_products=hx(prod))
    for k, v in want.items():
        assert got.get(k) == v, f"{name}: {k}: sanitizer build {got.get(k)} != liboracle {v}"
