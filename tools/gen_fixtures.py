#!/usr/bin/env python3
"""Generate tests/golden/*.  Run in the build container only: it drives the
reference compiled from its own sources (oracle/_ref/libref.so, `make -C oracle`)
and scipy.  Neither travels; the fixtures written here do.

Fixtures are DATA: seeded inputs are regenerated from spmm_amd/synth.py at test
time (the digests of the inputs are stored so a drift is noticed), the expected
outputs are arrays (small cases) or FNV-1a-64 digests (large cases).

  pre_<case>.json   digests / small arrays of every pre-processing stage, from the
                    REFERENCE ("ref_*" keys: libstdc++ std::sort tie order) and the
                    canonical stable-tie order ("stable_*" keys, produced by the
                    oracle after it matched the reference in std::sort mode)
  pre_<case>.npz    full arrays for the small cases
  spgemm_<case>.json / .npz   product C = A*A (or A*B) from scipy.sparse
"""
import json
import os
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import oracle as orc  # noqa: E402
from oracle import ref  # noqa: E402
from spmm_amd import synth  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")

# name -> (generator, args, store full arrays?)
PRE_CASES = {
    "tiny": (synth.uniform, (5000, 40000, 1), True),
    "rmat12d": (synth.rmat, (12, 8, 3, True), True),
    "rmat17d": (synth.rmat, (17, 8, 5, True), False),
    "rmat18": (synth.rmat, (18, 8, 7, False), False),
    "rmat20d": (synth.rmat, (20, 4, 11, True), False),
}
# Digests (FNV-1a-64 over the int32 elements) of the unmodified reference binary's outputs on these inputs, recorded
# before this repository existed; reproducing them shows that the reference compiled here is the same program.
RECORDED = {
    "tiny": dict(seq="09a2a689ebddb6d0", rseq="be43993869c3906d", indptr="69105ed43e567c04", indices="9a4e3afe7e447aec"),
    "rmat18": dict(seq="0237785705ccb788", rseq="8782862c96487e8f", indptr="707b5ef5eba268f5", indices="9f85fc985f8ffa37"),
    "rmat20d": dict(seq="45bca068c6b0eee7", rseq="14ce33b9286f9d29", indptr="4b79329e84cd633f", indices="12aa1be6d27e24d7"),
}


def hx(a):
    a = np.ascontiguousarray(a)
    if a.dtype.itemsize == 8:
        a = a.view(np.uint32)
    return f"{orc.fnv(a):016x}"


def shape_n(gen, args):
    return args[0] if gen is synth.uniform else 1 << args[0]


def pre_case(name, gen, args, full):
    n = shape_n(gen, args)
    row, col = gen(*args)
    with tempfile.TemporaryDirectory() as td:
        path = os.path.join(td, f"{name}.mtx")
        synth.write_mtx(path, n, row, col)
        nrow, ncol, ip, ix, d = ref.load(path)
    nnz = ix.size
    J = dict(case=name, generator=gen.__name__, args=list(args), n=n, nnz=int(nnz),
             in_row=hx(row), in_col=hx(col), csr_indptr=hx(ip), csr_indices=hx(ix))
    t = ref.trace_formats(nrow, nnz)
    J.update(bnum=t["bnum"], bsize_list=t["bsize_list"].tolist(), seq_offset=t["seq_offset"].tolist(),
             ref_seq_input=hx(t["seq_input"]), ref_rseq=hx(t["rseq"]))
    if name in RECORDED:
        s = RECORDED[name]
        assert (J["ref_seq_input"], J["ref_rseq"], J["csr_indptr"], J["csr_indices"]) == \
               (s["seq"], s["rseq"], s["indptr"], s["indices"]), f"{name}: recorded digests not reproduced"
        J["recorded_digests_reproduced"] = True
    # stage level, reference functions
    r_ip, r_ix, r_d, seq_bitmap = ref.bitmap_reorder(nrow, nnz)
    bs = ref.gen_new_panels()
    assert np.array_equal(bs, t["bsize_list"])
    J.update(seq_bitmap=hx(seq_bitmap), r_indptr=hx(r_ip), r_indices=hx(r_ix))
    blocks = [ref.stage_block(r_ip, int(bs[b]), int(bs[b + 1])) for b in range(len(bs) - 1)]
    cat = lambda k: np.concatenate([b[k] for b in blocks]) if blocks else np.zeros(0, np.int32)
    out_seq_all, keys = cat("out_seq"), cat("keys")
    seq_input2, rseq2 = ref.wbseq(seq_bitmap, out_seq_all, bs, keys)
    # the stage-wise replay must reproduce what gen_trace_formats itself returned
    assert np.array_equal(seq_input2, t["seq_input"]) and np.array_equal(rseq2, t["rseq"]), name
    soff = np.concatenate([[0], np.cumsum([b["keys"].size for b in blocks])])
    assert np.array_equal(soff, t["seq_offset"])
    J.update(panel_list=hx(cat("panel_list")), spv8_list=hx(cat("spv8_list")),
             n_panel_entries=int(cat("panel_list").size),
             ref_out_seq=hx(out_seq_all), ref_b_indptr=hx(cat("b_indptr")), ref_b_colidx=hx(cat("b_colidx")),
             ref_b_local=hx(cat("b_local")), ref_keys=hx(keys))
    # oracle in std::sort mode must equal the reference everywhere
    o0 = orc.trace_formats(ncol, ip, ix, d, sort_mode=0)
    for k_ref, k_o in [("ref_out_seq", "out_seq_all"), ("ref_b_indptr", "b_indptr"), ("ref_b_colidx", "b_colidx_orig"),
                       ("ref_b_local", "b_local"), ("ref_keys", "keys"), ("ref_seq_input", "seq_input"),
                       ("ref_rseq", "rseq"), ("panel_list", "panel_list"), ("spv8_list", "spv8_list"),
                       ("seq_bitmap", "seq_bitmap"), ("r_indptr", "r_indptr"), ("r_indices", "r_indices")]:
        assert J[k_ref] == hx(o0[k_o]), (name, k_ref)
    assert o0["bsize_list"].tolist() == J["bsize_list"] and o0["seq_offset"].tolist() == J["seq_offset"]
    # canonical stable tie order (what the GPU path must match bit for bit)
    o1 = orc.trace_formats(ncol, ip, ix, d, sort_mode=1)
    assert o1["bsize_list"].tolist() == J["bsize_list"] and o1["seq_offset"].tolist() == J["seq_offset"]
    assert hx(o1["spv8_list"]) == J["spv8_list"] and hx(o1["panel_list"]) == J["panel_list"]
    J.update(stable_out_seq=hx(o1["out_seq_all"]), stable_b_indptr=hx(o1["b_indptr"]),
             stable_b_colidx=hx(o1["b_colidx_orig"]), stable_b_local=hx(o1["b_local"]), stable_keys=hx(o1["keys"]),
             stable_seq_input=hx(o1["seq_input"]), stable_rseq=hx(o1["rseq"]))
    with open(os.path.join(GOLD, f"pre_{name}.json"), "w") as f:
        json.dump(J, f, indent=1)
    if full:
        np.savez_compressed(
            os.path.join(GOLD, f"pre_{name}.npz"), row=row, col=col, csr_indptr=ip, csr_indices=ix,
            seq_bitmap=seq_bitmap, r_indptr=r_ip, r_indices=r_ix, bsize_list=bs, seq_offset=t["seq_offset"],
            panel_list=cat("panel_list"), spv8_list=cat("spv8_list"),
            ref_out_seq=out_seq_all, ref_b_indptr=cat("b_indptr"), ref_b_colidx=cat("b_colidx"),
            ref_b_local=cat("b_local"), ref_keys=keys, ref_seq_input=t["seq_input"], ref_rseq=t["rseq"],
            stable_out_seq=o1["out_seq_all"], stable_b_colidx=o1["b_colidx_orig"], stable_b_local=o1["b_local"],
            stable_keys=o1["keys"], stable_seq_input=o1["seq_input"], stable_rseq=o1["rseq"])
    print(f"pre_{name}: n={n} nnz={nnz} bnum={t['bnum']} ok")


def spgemm_cases():
    import scipy.sparse as sp
    cases = {
        "tiny": (synth.uniform, (5000, 40000, 1), True),
        "rmat10": (synth.rmat, (10, 16, 2, False), True),
        "rmat12d": (synth.rmat, (12, 8, 3, True), False),
        "rmat14": (synth.rmat, (14, 16, 7, False), False),
        "rmat16": (synth.rmat, (16, 16, 7, False), False),
    }
    for name, (gen, args, full) in cases.items():
        n = shape_n(gen, args)
        row, col = gen(*args)
        ip, ix, d = orc.csr_from_coo(n, row, col)
        A = sp.csr_matrix((d, ix, ip), shape=(n, n))
        Cs = (A @ A).tocsr()
        Cs.sort_indices()
        c_ip = Cs.indptr.astype(np.int64)
        J = dict(case=name, generator=gen.__name__, args=list(args), n=n, nnz=int(ix.size), in_row=hx(row), in_col=hx(col),
                 nnzC=int(c_ip[-1]), products=int(orc.row_products(ip, ix, ip)[1]), c_indptr=hx(c_ip),
                 c_indices=hx(Cs.indices.astype(np.int32)), c_data=hx(Cs.data.astype(np.float64)),
                 c_data_sum=float(Cs.data.sum()))
        # the oracle's Gustavson must equal scipy bit for bit (integer-valued sums are exact)
        o_ip, o_ix, o_d = orc.spgemm(n, (ip, ix, d), (ip, ix, d))
        assert np.array_equal(o_ip, c_ip) and np.array_equal(o_ix, Cs.indices) and np.array_equal(o_d, Cs.data), name
        with open(os.path.join(GOLD, f"spgemm_{name}.json"), "w") as f:
            json.dump(J, f, indent=1)
        if full:
            np.savez_compressed(os.path.join(GOLD, f"spgemm_{name}.npz"), row=row, col=col, c_indptr=c_ip,
                                c_indices=Cs.indices.astype(np.int32), c_data=Cs.data.astype(np.float64))
        print(f"spgemm_{name}: n={n} nnz={ix.size} nnzC={c_ip[-1]} P={J['products']} ok")
    # rectangular A(300x200) * B(200x500), random fp64 values, duplicate entries in A
    rng = np.random.default_rng(99)
    ar, ac = rng.integers(0, 300, 3000), rng.integers(0, 200, 3000)
    br, bc = rng.integers(0, 200, 4000), rng.integers(0, 500, 4000)
    av, bv = rng.standard_normal(3000), rng.standard_normal(4000)
    A = sp.coo_matrix((av, (ar, ac)), shape=(300, 200)).tocsr()   # sums duplicates
    B = sp.coo_matrix((bv, (br, bc)), shape=(200, 500)).tocsr()
    A.sort_indices(); B.sort_indices()
    Cs = (A @ B).tocsr(); Cs.sort_indices()
    np.savez_compressed(os.path.join(GOLD, "spgemm_rect.npz"),
                        a_indptr=A.indptr.astype(np.int32), a_indices=A.indices.astype(np.int32), a_data=A.data,
                        b_indptr=B.indptr.astype(np.int32), b_indices=B.indices.astype(np.int32), b_data=B.data,
                        c_indptr=Cs.indptr.astype(np.int64), c_indices=Cs.indices.astype(np.int32), c_data=Cs.data)
    print("spgemm_rect ok nnzC", Cs.nnz)


if __name__ == "__main__":
    assert ref.available(), "build oracle/_ref first: make -C oracle"
    os.makedirs(GOLD, exist_ok=True)
    only = sys.argv[1:]
    for name, (gen, args, full) in PRE_CASES.items():
        if not only or name in only:
            pre_case(name, gen, args, full)
    if not only or "spgemm" in only:
        spgemm_cases()
