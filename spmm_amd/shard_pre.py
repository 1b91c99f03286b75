"""The format pipeline (gen_trace_formats, serial_newblock_clock.cpp:221-499) over rows of A sharded across ranks.

Every rank starts with the CSR of an even slice of the rows and ends with the packed formats of the blocks it owns; the
concatenation over the ranks is what one process produces for the whole matrix.  Three steps of the pipeline reach
across rows, and each is one exchange here:

  1. the stable bucket by row key (bitmap.h:144-163): the per-key counts of all ranks (one small all-gather) give every
     row its global position; the rows move to the rank that owns that position;
  2. the greedy block chain (transmat.h:346-362) is serial over the reordered rows: rank r continues the block its
     predecessor left open, for which it needs that block's distinct columns so far (at most threshold - 1 ids), cuts
     its own rows, and hands its open tail on; the cut positions are then all-gathered;
  3. the write-back sequence (wbsort.h:16-34, 58-67, 81-95) inverts a global permutation: (original row, packed
     position) pairs are all-gathered.

Rows of a block that straddles a rank boundary move to the rank where the block starts, so that panels, packing and
first-touch numbering (v8sort.h:49-73, 152-232; serial_newblock_clock.cpp:336-399, 187-204) run on whole blocks.

The per-rank compute comes through `stages`, an object with the methods
    row_keys(indptr, indices, sect) -> int32[nrow]
    gen_blocks(ncol, indptr, indices, threshold) -> bsize list of the local rows (0, cuts.., nrow; the reference's
        rule of always appending nrow included)
    block_formats(ncol, indptr, indices, data, bsize) -> dict with panel_ptr, panel_list, spv8_list, out_seq_all,
        b_indptr, b_colidx_orig, b_data, b_local, keys, seq_offset for the blocks of a local matrix
GpuStages below runs them on the library (one context per rank).  `dist` is torch.distributed or an object with the
same all_gather_into_tensor / isend / irecv / send / recv on CPU tensors: all exchanges are made on host tensors.
"""
import numpy as np
import torch

from . import capi
from .shard import even_row_slice

I32, I64, F64 = np.int32, np.int64, np.float64


class GpuStages:
    """The stages on the GPU of this rank, through the C ABI (host arrays in and out)."""

    def __init__(self, ctx):
        self.ctx = ctx

    def row_keys(self, indptr, indices, sect):
        A = capi.Csr(indptr.size - 1, 0, indptr, indices, np.zeros(indices.size, F64))
        return self.ctx.row_keys(A, sect)

    def gen_blocks(self, ncol, indptr, indices, threshold):
        R = capi.Csr(indptr.size - 1, ncol, indptr, indices, np.zeros(indices.size, F64))
        return self.ctx.gen_blocks(R, threshold)

    def block_formats(self, ncol, indptr, indices, data, bsize):
        R = capi.Csr(indptr.size - 1, ncol, indptr, indices, data)
        out = self.ctx.panels_sort_pack(R, bsize)
        out["b_local"], out["keys"], out["seq_offset"] = self.ctx.first_touch(R, bsize, out["b_colidx_orig"])
        return out


def _t(a):
    return torch.from_numpy(np.ascontiguousarray(a))


def _allgather_var(dist, world, a):
    """Concatenation over the ranks of 1-D arrays of different lengths; also returns the lengths."""
    a = np.ascontiguousarray(a)
    sizes = torch.empty(world, dtype=torch.int64)
    dist.all_gather_into_tensor(sizes, torch.tensor([a.size], dtype=torch.int64))
    m = int(sizes.max())
    pad = np.zeros(max(m, 1), a.dtype)
    pad[:a.size] = a
    out = torch.empty(world * max(m, 1), dtype=_t(pad).dtype)
    dist.all_gather_into_tensor(out, _t(pad))
    out = out.numpy().reshape(world, max(m, 1))
    return np.concatenate([out[r, :int(sizes[r])] for r in range(world)]), sizes.numpy()


def exchange(dist, rank, world, send):
    """send[p]: 1-D array for rank p (one dtype).  Returns the list of arrays received, by source rank."""
    send = [np.ascontiguousarray(s) for s in send]
    sizes = torch.empty(world * world, dtype=torch.int64)
    dist.all_gather_into_tensor(sizes, torch.tensor([s.size for s in send], dtype=torch.int64))
    sizes = sizes.view(world, world)          # [source][destination]
    recv = [np.empty(int(sizes[src, rank]), send[0].dtype) for src in range(world)]
    recv[rank][:] = send[rank]
    reqs, keep = [], []
    for peer in range(world):
        if peer == rank:
            continue
        if recv[peer].size:
            t = torch.from_numpy(recv[peer])
            keep.append(t)
            reqs.append(dist.irecv(t, src=peer))
        if send[peer].size:
            t = torch.from_numpy(send[peer])
            keep.append(t)
            reqs.append(dist.isend(t, dst=peer))
    for r in reqs:
        r.wait()
    return recv


def _move_rows(dist, rank, world, dest, pos, old, ip, ix, d):
    """Rows of a local CSR go to the ranks dest[row]; every rank gets its rows ordered by pos.  Returns
    (pos, old ids, CSR) of what arrived."""
    lens = np.diff(ip).astype(I64)
    order = np.argsort(pos, kind="stable")
    meta, cols, vals = [], [], []
    for p in range(world):
        sel = order[dest[order] == p]
        meta.append(np.stack([pos[sel], old[sel], lens[sel]], axis=1).astype(I64).ravel())
        idx = (np.repeat(ip[sel].astype(I64) - np.concatenate([[0], np.cumsum(lens[sel])[:-1]]), lens[sel]) +
               np.arange(int(lens[sel].sum()), dtype=I64)) if sel.size else np.zeros(0, I64)
        cols.append(ix[idx].astype(I32))
        vals.append(d[idx].astype(F64))
    meta = [m.reshape(-1, 3) for m in exchange(dist, rank, world, meta)]
    cols = exchange(dist, rank, world, cols)
    vals = exchange(dist, rank, world, vals)
    m_all = np.concatenate(meta) if meta else np.zeros((0, 3), I64)
    c_all, v_all = np.concatenate(cols), np.concatenate(vals)
    src_off = np.concatenate([[0], np.cumsum(m_all[:, 2])[:-1]]).astype(I64) if m_all.shape[0] else np.zeros(0, I64)
    o = np.argsort(m_all[:, 0], kind="stable")
    n_pos, n_old, n_len = m_all[o, 0], m_all[o, 1], m_all[o, 2]
    n_ip = np.concatenate([[0], np.cumsum(n_len)]).astype(I64)
    take = (np.repeat(src_off[o] - n_ip[:-1], n_len) + np.arange(int(n_len.sum()), dtype=I64)) if o.size else np.zeros(0, I64)
    return n_pos, n_old.astype(I32), n_ip.astype(I32), c_all[take].astype(I32), v_all[take].astype(F64)


def trace_formats_sharded(stages, dist, rank, world, n, ncol, indptr, indices, data, sect=2048, threshold=65536):
    """indptr/indices/data: the CSR of rows even_row_slice(n, rank, world) of the matrix (local row pointers starting
    at 0).  Returns a dict: this rank's slice of the reordered matrix (pos_lo, seq_bitmap, r_indptr, r_indices,
    r_data), the global bsize_list and rseq, and the packed formats of the blocks it owns (blocks, panel_ptr,
    panel_list, spv8_list, out_seq_all, b_indptr, b_colidx_orig, b_data, b_local, keys, seq_offset, seq_input)."""
    indptr, indices, data = np.asarray(indptr, I32), np.asarray(indices, I32), np.asarray(data, F64)
    lo, hi = even_row_slice(n, rank, world)
    nl = hi - lo
    assert indptr.size == nl + 1
    bounds = np.array([even_row_slice(n, r, world)[0] for r in range(world)] + [n], I64)

    # ---- 1. global bucket order ------------------------------------------------------------------------------------
    keys = np.asarray(stages.row_keys(indptr, indices, sect), I64) if nl else np.zeros(0, I64)
    nk = ncol // sect + 1
    hist = np.bincount(keys, minlength=nk).astype(I64)
    H = torch.empty(world * nk, dtype=torch.int64)
    dist.all_gather_into_tensor(H, _t(hist))
    H = H.numpy().reshape(world, nk)
    key_base = np.concatenate([[0], np.cumsum(H.sum(0))[:-1]])
    rank_base = H[:rank].sum(0)
    order = np.argsort(keys, kind="stable")
    starts = np.concatenate([[0], np.cumsum(hist)[:-1]])
    within = np.empty(nl, I64)
    within[order] = np.arange(nl, dtype=I64) - starts[keys[order]]
    pos = key_base[keys] + rank_base[keys] + within
    dest = np.searchsorted(bounds, pos, side="right") - 1
    r_pos, seqb, r_ip, r_ix, r_d = _move_rows(dist, rank, world, dest, pos, np.arange(lo, hi, dtype=I64), indptr, indices, data)
    plo, phi = int(bounds[rank]), int(bounds[rank + 1])
    assert np.array_equal(r_pos, np.arange(plo, phi)), "every position of this rank's range arrives exactly once"

    # ---- 2. block chain: continue the predecessor's open block, cut, hand the open tail on ---------------------------------
    carried = np.zeros(0, I32)
    if rank > 0:
        sz = torch.empty(1, dtype=torch.int64)
        dist.recv(sz, src=rank - 1)
        if int(sz[0]):
            buf = torch.empty(int(sz[0]), dtype=torch.int32)
            dist.recv(buf, src=rank - 1)
            carried = buf.numpy()
    v = 1 if carried.size else 0            # a virtual first row holds the open block's columns so far
    ip_v = np.concatenate([[0, carried.size], r_ip[1:].astype(I64) + carried.size]).astype(I32) if v else r_ip
    ix_v = np.concatenate([carried, r_ix]).astype(I32) if v else r_ix
    bs = np.asarray(stages.gen_blocks(ncol, ip_v, ix_v, threshold), I64)
    real = bs[1:-1] - v                      # ends of the blocks cut here, in local rows (the last entry is the forced end)
    if real.size:
        tail = np.unique(r_ix[r_ip[int(real[-1])]:])
    else:
        tail = np.unique(np.concatenate([carried, r_ix]))
    if rank < world - 1:
        dist.send(torch.tensor([tail.size], dtype=torch.int64), dst=rank + 1)
        if tail.size:
            dist.send(_t(tail.astype(I32)), dst=rank + 1)
    cuts, _ = _allgather_var(dist, world, real + plo)
    bsize = np.concatenate([[0], cuts, [n]]).astype(I64)       # the reference appends nrow whatever the last cut was
    nb = bsize.size - 1

    # ---- 3. whole blocks to the rank where they start -------------------------------------------------------------------------
    blk_owner = np.searchsorted(bounds, np.minimum(bsize[:-1], max(n - 1, 0)), side="right") - 1
    if nb > 1 and bsize[nb - 1] == n:       # an empty last block goes with the block before it
        blk_owner[nb - 1] = blk_owner[nb - 2]
    my_pos = np.arange(plo, phi, dtype=I64)
    row_blk = np.searchsorted(bsize, my_pos, side="right") - 1
    q_pos, q_old, q_ip, q_ix, q_d = _move_rows(dist, rank, world, blk_owner[row_blk], my_pos, seqb.astype(I64), r_ip, r_ix, r_d)
    mine = np.nonzero(blk_owner == rank)[0]
    out = dict(pos_lo=plo, seq_bitmap=seqb, r_indptr=r_ip, r_indices=r_ix, r_data=r_d, bsize_list=bsize.astype(I32),
               blocks=(int(mine[0]), int(mine[-1]) + 1) if mine.size else (0, 0))
    if mine.size:
        b0, b1 = int(mine[0]), int(mine[-1]) + 1
        qlo = int(bsize[b0])
        assert np.array_equal(q_pos, np.arange(qlo, int(bsize[b1]))), "the rows of this rank's blocks, each once"
        lb = (bsize[b0:b1 + 1] - qlo).astype(I32)
        F = stages.block_formats(ncol, q_ip, q_ix, q_d, lb)
        out_seq = np.asarray(F["out_seq_all"], I64)
        # packed row t of a block sits at reordered position block start + out_seq[t]
        t_blk = np.searchsorted(lb, np.arange(q_pos.size), side="right") - 1
        orig = q_old[lb[t_blk] + out_seq]
        pairs = np.stack([orig.astype(I64), qlo + np.arange(q_pos.size, dtype=I64)], axis=1).ravel()
        nkeys = int(np.asarray(F["keys"]).size)
    else:
        F, pairs, nkeys = None, np.zeros(0, I64), 0

    # ---- 4. write-back sequence: the global inverse permutation, key offsets across ranks ---------------------------------------
    allp, _ = _allgather_var(dist, world, pairs)
    allp = allp.reshape(-1, 2)
    rseq = np.empty(n, I32)
    rseq[allp[:, 0]] = allp[:, 1].astype(I32)
    nks = torch.empty(world, dtype=torch.int64)
    dist.all_gather_into_tensor(nks, torch.tensor([nkeys], dtype=torch.int64))
    key_off = int(nks[:rank].sum())
    out["rseq"] = rseq
    out["n_keys_before"] = key_off
    if F is not None:
        for k in ("panel_ptr", "panel_list", "spv8_list", "out_seq_all", "b_indptr", "b_colidx_orig", "b_data", "b_local", "keys"):
            out[k] = np.asarray(F[k])
        out["seq_offset"] = (np.asarray(F["seq_offset"], I64) + key_off).astype(I32)
        out["seq_input"] = rseq[np.asarray(F["keys"], I64)]
    return out
