"""CPU, world sizes 2, 3 and 4 over gloo: the format pipeline over row shards (spmm_amd/shard_pre.py) with the oracle's
stage functions standing in for the GPU of every rank.  The pieces the ranks end with, put together, must equal what
one process makes of the whole matrix (oracle.trace_formats, the restatement of serial_newblock_clock.cpp:221-499):
the reordered matrix, the block list, every packed array, the key lists and the write-back sequence.  A small
threshold gives blocks that straddle rank boundaries, ranks that cut no block at all, and a cut exactly at the end."""
import os
import socket

import numpy as np
import pytest
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import oracle as orc
from spmm_amd import shard, shard_pre, synth


class OracleStages:
    """The three per-rank computations, by the CPU oracle (in production: shard_pre.GpuStages)."""

    def row_keys(self, indptr, indices, sect):
        return orc.row_keys(indptr, indices, sect)

    def gen_blocks(self, ncol, indptr, indices, threshold):
        return orc.gen_blocks(ncol, indptr, indices, threshold)

    def block_formats(self, ncol, indptr, indices, data, bsize):
        nb = len(bsize) - 1
        o = dict(panel_ptr=[0], panel_list=[], spv8_list=[], out_seq_all=[], b_indptr=[], b_colidx_orig=[], b_data=[], b_local=[],
                 keys=[], seq_offset=[0])
        for b in range(nb):
            r0, r1 = int(bsize[b]), int(bsize[b + 1])
            pl = orc.panel_list(indptr, r0, r1)
            out_seq, spv8 = orc.panel_sort(indptr, r0, r1, pl, 1)
            b_ip, b_ci, b_d = orc.gather_spv8(indptr, indices, data, r0, r1, out_seq, pl, spv8)
            local, keys = orc.first_touch(ncol, b_ci)
            for k, v in (("panel_list", pl), ("spv8_list", spv8), ("out_seq_all", out_seq), ("b_indptr", b_ip), ("b_colidx_orig", b_ci),
                         ("b_data", b_d), ("b_local", local), ("keys", keys)):
                o[k].append(v)
            o["panel_ptr"].append(o["panel_ptr"][-1] + pl.size)
            o["seq_offset"].append(o["seq_offset"][-1] + keys.size)
        cat = lambda xs, t: np.concatenate(xs).astype(t) if xs else np.zeros(0, t)
        return dict(panel_ptr=np.array(o["panel_ptr"], np.int32), seq_offset=np.array(o["seq_offset"], np.int32),
                    b_data=cat(o["b_data"], np.float64),
                    **{k: cat(o[k], np.int32) for k in ("panel_list", "spv8_list", "out_seq_all", "b_indptr", "b_colidx_orig", "b_local", "keys")})


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _graph(case):
    if case == "rmat":
        n = 1 << 11
        row, col = synth.rmat(11, 8, 4, diagonal=True)
        return n, row, col, 300
    if case == "one_block":                 # the reference's threshold: a single block, ranks 1.. cut nothing
        n = 1 << 10
        row, col = synth.rmat(10, 6, 9, diagonal=True)
        return n, row, col, 65536
    # a cut exactly after the last row: rows of 16 fresh columns each, threshold 64, n a multiple of 4 rows
    n = 64
    row = np.repeat(np.arange(n), 16)
    col = (np.arange(n * 16) % n)
    col = (np.repeat(np.arange(n) * 16, 16) + np.tile(np.arange(16), n)) % n
    order = np.lexsort((row, col))
    return n, row[order].astype(np.int32), col[order].astype(np.int32), 64


def _worker(rank, world, port, q, case):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        n, row, col, threshold = _graph(case)
        lo, hi = shard.even_row_slice(n, rank, world)
        m = (row >= lo) & (row < hi)
        ip_l, ix_l, d_l = orc.csr_from_coo(hi - lo, row[m] - lo, col[m])
        d_l = d_l * (1.0 + (np.arange(d_l.size) % 5))            # values that tell entries apart
        out = shard_pre.trace_formats_sharded(OracleStages(), dist, rank, world, n, n, ip_l, ix_l, d_l, 2048 if case != "rmat" else 256, threshold)
        q.put((rank, out))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,case", [(2, "rmat"), (3, "rmat"), (4, "rmat"), (4, "one_block"), (2, "cut_at_the_end")])
def test_sharded_pipeline_equals_single_process(world, case):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q, case)) for r in range(world)]
    for p in procs:
        p.start()
    got = [o for _, o in sorted([q.get(timeout=240) for _ in range(world)], key=lambda t: t[0])]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    n, row, col, threshold = _graph(case)
    ip, ix, d = orc.csr_from_coo(n, row, col)
    # the same values as the ranks made: by position inside the rank's own slice
    for r in range(world):
        lo, hi = shard.even_row_slice(n, r, world)
        seg = slice(ip[lo], ip[hi])
        d[seg] = 1.0 + (np.arange(ip[hi] - ip[lo]) % 5)
    W = orc.trace_formats(n, ip, ix, d, sect=2048 if case != "rmat" else 256, threshold=threshold, sort_mode=1)
    cat = lambda k: np.concatenate([o[k] for o in got if k in o])
    assert np.array_equal(cat("seq_bitmap"), W["seq_bitmap"])
    r_ip = np.concatenate([[0]] + [o["r_indptr"][1:].astype(np.int64) + sum(int(p["r_indptr"][-1]) for p in got[:i]) for i, o in enumerate(got)])
    assert np.array_equal(r_ip, W["r_indptr"])
    assert np.array_equal(cat("r_indices"), W["r_indices"]) and np.array_equal(cat("r_data"), W["r_data"])
    for o in got:
        assert np.array_equal(o["bsize_list"], W["bsize_list"]) and np.array_equal(o["rseq"], W["rseq"])
    assert W["bnum"] == W["bsize_list"].size - 1
    if case == "rmat":
        assert W["bnum"] > 2 * world, "the case is meant to have many blocks"
        assert sum(1 for o in got if o["blocks"][1] > o["blocks"][0]) == world
    if case == "cut_at_the_end":
        assert W["bsize_list"][-1] == W["bsize_list"][-2] == n, "the case is meant to end with an empty block"
    owned = [o["blocks"] for o in got if o["blocks"][1] > o["blocks"][0]]
    assert owned[0][0] == 0 and owned[-1][1] == W["bnum"] and all(a[1] == b[0] for a, b in zip(owned, owned[1:]))
    for k in ("panel_list", "spv8_list", "out_seq_all", "b_indptr", "b_colidx_orig", "b_data", "b_local", "keys", "seq_input"):
        assert np.array_equal(cat(k), W[k]), k
    holders = [o for o in got if "panel_ptr" in o]
    pp, base = [0], 0
    for o in holders:
        pp += (o["panel_ptr"][1:].astype(np.int64) + base).tolist()
        base += int(o["panel_ptr"][-1])
    assert np.array_equal(np.array(pp), W["panel_ptr"])
    so = [0]
    for o in holders:
        so += o["seq_offset"][1:].tolist()
    assert np.array_equal(np.array(so), W["seq_offset"])
