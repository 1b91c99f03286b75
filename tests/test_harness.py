"""The C++ harness keeps the reference's process-level contract (serial_newblock_clock.cpp:501-599):
matrix.txt -> mat/mtx/<name>/<name>.mtx -> result.txt + stdout banner."""
import json
import os
import re
import subprocess

import numpy as np
import pytest

from oracle import oracle as orc
from oracle import ref
from spmm_amd import synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SERIAL = os.path.join(ROOT, "spmm_amd", "harness", "serial")
LINE = re.compile(r"^(\S+) ([0-9.eE+-]+)ms$")


def _workdir(tmp_path, names):
    for name, (n, row, col) in names.items():
        d = tmp_path / "mat" / "mtx" / name
        d.mkdir(parents=True)
        synth.write_mtx(str(d / f"{name}.mtx"), n, row, col, pattern=(name != "vals"))
    (tmp_path / "matrix.txt").write_text("".join(f"{k}.mtx\n" for k in names))


def test_harness_is_built_and_refuses_to_run_without_a_gpu(tmp_path):
    assert os.path.exists(SERIAL), "build first: make -C spmm_amd/harness"
    if os.path.exists("/dev/kfd"):
        pytest.skip("GPU present")
    _workdir(tmp_path, {"tiny": (50,) + synth.uniform(50, 200, 1)})
    r = subprocess.run([SERIAL], cwd=tmp_path, capture_output=True, text=True, timeout=120)
    assert r.returncode == 2 and "no usable MI355X" in r.stderr      # loud failure, no CPU fallback


@pytest.mark.skipif(not ref.available(), reason="compiled reference only exists in the build container")
def test_reference_binary_output_format(tmp_path):
    """Pins the format the harness must reproduce, on the reference's own binary."""
    _workdir(tmp_path, {"tiny": (5000,) + synth.uniform(5000, 40000, 1)})
    r = subprocess.run([os.path.join(ROOT, "oracle", "_ref", "serial")], cwd=tmp_path, capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and r.stdout == "----name:tiny----\n"
    lines = (tmp_path / "result.txt").read_text().splitlines()
    assert len(lines) == 1 and LINE.match(lines[0]).group(1) == "tiny"


@pytest.mark.gpu
def test_harness_end_to_end(tmp_path):
    cases = {"tiny": (5000,) + synth.uniform(5000, 40000, 1), "vals": (1 << 11,) + synth.rmat(11, 8, 21, True)}
    _workdir(tmp_path, cases)
    env = dict(os.environ, SPMM_HARNESS_SPGEMM="1", SPMM_HARNESS_TRACE=str(tmp_path / "trace"))
    r = subprocess.run([SERIAL], cwd=tmp_path, capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stderr
    assert r.stdout == "----name:tiny----\n----name:vals----\n"
    lines = (tmp_path / "result.txt").read_text().splitlines()
    assert [LINE.match(l).group(1) for l in lines] == ["tiny", "vals"]
    assert all(float(LINE.match(l).group(2)) > 0 for l in lines)
    reports = [json.loads(l) for l in (tmp_path / "result_spgemm.txt").read_text().splitlines()]
    for rep, (name, (n, row, col)) in zip(reports, cases.items()):
        ip, ix, d = orc.csr_from_coo(n, row, col)
        c_ip, c_ix, c_d = orc.spgemm(n, (ip, ix, d), (ip, ix, d))
        assert rep["name"] == name and rep["rows"] == n and rep["nnz"] == ix.size
        assert rep["nnz_c"] == int(c_ip[-1]) and rep["products"] == orc.row_products(ip, ix, ip)[1]
        assert rep["c_indices_fnv1a64"] == f"{orc.fnv(c_ix):016x}"
        assert rep["c_value_sum"] == float(c_d.sum())
        T = orc.trace_formats(n, ip, ix, d)
        assert rep["blocks"] == T["bnum"]
        # the on-disk trace: every array of the pipeline, equal to the oracle's (stable tie order)
        for key, want in T.items():
            if key == "bnum":
                continue
            got = np.load(tmp_path / "trace" / name / f"{key}.npy")
            assert got.dtype == want.dtype and np.array_equal(got, want), key


@pytest.mark.gpu
def test_harness_binary_side_cache_and_a_web_sized_graph(tmp_path):
    """SPMM_HARNESS_CACHE=1: the first run parses the text and leaves <name>.mtx.bin, the second reads the entries from
    it (no text parse) and reports the same product digests; a changed text invalidates the cache.  The graph is the
    web-Stanford-sized stand-in (281 903 rows, 2.3 M entries), checked against the oracle as well."""
    n = 281903
    row, col = synth.stanford_like()
    _workdir(tmp_path, {"stanford": (n, row, col)})
    env = dict(os.environ, SPMM_HARNESS_SPGEMM="1", SPMM_HARNESS_CACHE="1")

    def run():
        r = subprocess.run([SERIAL], cwd=tmp_path, capture_output=True, text=True, timeout=600, env=env)
        assert r.returncode == 0, r.stderr
        return json.loads((tmp_path / "result_spgemm.txt").read_text().splitlines()[0])

    first = run()
    bin_path = tmp_path / "mat" / "mtx" / "stanford" / "stanford.mtx.bin"
    assert first["entries_from"] == "text" and bin_path.exists()
    second = run()
    assert second["entries_from"] == "cache"
    for k in ("rows", "nnz", "blocks", "products", "nnz_c", "c_indices_fnv1a64", "c_value_sum"):
        assert first[k] == second[k], k
    ip, ix, d = orc.csr_from_coo(n, row, col)
    c_ip, c_ix, c_d = orc.spgemm(n, (ip, ix, d), (ip, ix, d))
    assert second["nnz"] == ix.size and second["nnz_c"] == int(c_ip[-1]) and second["c_indices_fnv1a64"] == f"{orc.fnv(c_ix):016x}"
    assert second["c_value_sum"] == float(c_d.sum()) and second["blocks"] == orc.trace_formats(n, ip, ix, d)["bnum"]
    # another text under the same name: one entry fewer
    synth.write_mtx(str(tmp_path / "mat" / "mtx" / "stanford" / "stanford.mtx"), n, row[:-1], col[:-1])
    third = run()
    assert third["entries_from"] == "text" and third["nnz"] == ix.size - 1
