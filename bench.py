#!/usr/bin/env python3
"""SpGEMM benchmark: C = A*A on a synthetic R-MAT graph, one process per GPU.

    python bench.py --gpus 1 --steps 3 --warmup 1
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Workload (config.workload): Graph500 R-MAT scale 22, edge factor 16 (4.2M rows, ~65M nnz after
dedup), every stored value 1.0 as the reference loader forces (serial_newblock_clock.cpp:84).
A step = symbolic (nnz per C row) + numeric (sorted column ids + fp64 values) over the rank's
rows of A, with A and B already resident in HBM.  C (about 6e10 entries at scale 22) does not
fit any memory, so numeric runs over row blocks that reuse one output buffer.

N > 1: rows of A are split across ranks by intermediate-product count; every rank builds the
CSR of an even row slice and B is assembled on every GPU by an all-gather over RCCL (set-up,
reported as b_allgather_ms).  The same matrix is used at every N (strong scaling).

Rank 0 prints ONE JSON line.  value = 2 * products / step time, whole job.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
from spmm_amd import capi, shard  # noqa: E402

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E peak


def source_digest():
    """Digest of the library's sources (kernels, shared header, C ABI, build flags): identifies the build a profile
    under profiles/ was taken from; the sources travel with every run, the git history does not."""
    import glob
    import hashlib
    h = hashlib.sha256()
    files = sorted(glob.glob(os.path.join(ROOT, "spmm_amd", "csrc", "*.hip"))) + sorted(glob.glob(os.path.join(ROOT, "spmm_amd", "csrc", "*.h"))) + [
        os.path.join(ROOT, "spmm_amd", "csrc", "Makefile"),
        os.path.join(ROOT, "include", "spmm_hip.h")]
    for f in files:
        h.update(os.path.basename(f).encode())
        with open(f, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def rmat_torch(scale, edge_factor, seed, device, row_lo=None, row_hi=None):
    """Graph500 R-MAT (a,b,c,d = .57,.19,.19,.05), level l decides bit l, no scrambling; dedup and
    (col,row) order like a SuiteSparse file.  Philox stream of torch, seeded.  With row_lo/row_hi only the
    entries of rows [row_lo, row_hi) are kept, BEFORE the dedup sort: every rank draws the same stream and
    sorts only its own slice of it (duplicates of an entry share its row, so they meet in one slice)."""
    a, b, c = 0.57, 0.19, 0.19
    n, m = 1 << scale, edge_factor << scale
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    row = torch.zeros(m, dtype=torch.int64, device=device)
    col = torch.zeros(m, dtype=torch.int64, device=device)
    for level in range(scale):
        u = torch.rand(m, generator=g, device=device, dtype=torch.float64)
        row |= (u >= a + b).to(torch.int64) << level
        col |= (((u >= a) & (u < a + b)) | (u >= a + b + c)).to(torch.int64) << level
        del u
    key = col * n + row
    if row_lo is not None:
        key = key[(row >= row_lo) & (row < row_hi)]
    del row, col
    key = torch.unique(key)
    return (key & (n - 1)).to(torch.int32), (key >> scale).to(torch.int32)


def build_csr(ctx, n_rows, ncol, row, col):
    nnz = row.numel()
    dev = row.device
    indptr = torch.empty(n_rows + 1, dtype=torch.int32, device=dev)
    indices = torch.empty(max(nnz, 1), dtype=torch.int32, device=dev)
    data = torch.empty(max(nnz, 1), dtype=torch.float64, device=dev)
    torch.cuda.synchronize()       # row/col come from torch kernels; the library reads them through its own runtime handle
    ctx.csr_from_coo_dev(n_rows, ncol, nnz, row, col, indptr, indices, data)
    ctx.sync()
    return indptr, indices[:nnz], data[:nnz]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--scale", type=int, default=22)
    ap.add_argument("--edge-factor", type=int, default=16)
    ap.add_argument("--seed", type=int, default=7)
    ap.add_argument("--cbuf-gib", type=float, default=96.0, help="output buffer (indices+values) per GPU")
    ap.add_argument("--cpu-products", type=float, default=2.5e9, help="size of the CPU baseline sample")
    ap.add_argument("--no-cpu", action="store_true")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    local = int(os.environ.get("LOCAL_RANK", 0))
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    # Rehearsal of the multi-rank path on a box with one GPU: SPMM_BENCH_REHEARSE=1 puts every rank on
    # cuda:0 and runs the collectives over gloo with host staging.  Numbers of such a run mean nothing.
    rehearse = os.environ.get("SPMM_BENCH_REHEARSE") == "1"
    if rehearse:
        local = 0
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dist = None
    if world > 1:
        import torch.distributed as tdist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearse:
            tdist.init_process_group("gloo")

            class _HostStaged:
                ReduceOp = tdist.ReduceOp
                barrier = staticmethod(tdist.barrier)
                destroy_process_group = staticmethod(tdist.destroy_process_group)

                @staticmethod
                def all_gather_into_tensor(out, inp):
                    o = torch.empty(out.shape, dtype=out.dtype)
                    tdist.all_gather_into_tensor(o, inp.cpu())
                    out.copy_(o)

                @staticmethod
                def all_reduce(t, op):
                    h = t.cpu()
                    tdist.all_reduce(h, op=op)
                    t.copy_(h)

            dist = _HostStaged
        else:
            tdist.init_process_group("nccl", device_id=dev)
            dist = tdist

    ctx = capi.Context(local)
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)

    # ---- set-up (untimed): graph, CSR, B on every GPU ------------------------
    t0 = time.time()
    n = 1 << args.scale
    b_allgather_ms = 0.0
    if world == 1:
        row, col = rmat_torch(args.scale, args.edge_factor, args.seed, dev)
        nnz = int(row.numel())
        indptr, indices, data = build_csr(ctx, n, n, row, col)
    else:
        lo, hi = shard.even_row_slice(n, rank, world)
        row, col = rmat_torch(args.scale, args.edge_factor, args.seed, dev, lo, hi)    # this rank's rows only
        ip_l, ix_l, _ = build_csr(ctx, hi - lo, n, (row - lo).contiguous(), col)
        ctx.sync()
        torch.cuda.synchronize()
        dist.barrier()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        indptr, indices = shard.allgather_csr_slices(ip_l, ix_l, n, rank, world, dist)
        e1.record()
        torch.cuda.synchronize()
        b_allgather_ms = e0.elapsed_time(e1)
        nnz = int(indices.numel())
        data = torch.ones(nnz, dtype=torch.float64, device=dev)
        assert int(indptr[-1]) == nnz
    del row, col
    torch.cuda.synchronize()

    vB = capi.dev_view(n, n, nnz, indptr, indices, data)
    prod = torch.empty(n, dtype=torch.int64, device=dev)
    P_total = ctx.row_products_dev(vB, vB, prod)
    # rows of A per rank: equal shares of the intermediate products, weighted by the kernel class of the row
    esc_max = ctx.sort_limit(n)
    _, table_windows, dense_min_products = ctx.plan_info(n)
    bounds = shard.cost_balanced_bounds(prod, world, esc_max, dense_min_products, max(ctx.light_max(n), esc_max),
                                        indptr[1:] - indptr[:-1], table_windows)
    r_lo, r_hi = bounds[rank], bounds[rank + 1]
    rows_l = r_hi - r_lo
    P_local = int(prod[r_lo:r_hi].sum())
    vA = capi.dev_view(rows_l, n, nnz, indptr[r_lo:], indices, data)

    c_ip = torch.empty(rows_l + 1, dtype=torch.int64, device=dev)
    cap = int(args.cbuf_gib * (1 << 30) / 12) // (world if rehearse else 1)    # rehearsing ranks share one GPU
    c_ix = torch.empty(cap, dtype=torch.int32, device=dev)
    c_val = torch.empty(cap, dtype=torch.float64, device=dev)
    setup_s = time.time() - t0

    state = {}

    def plan_blocks(c_ip_h):
        blocks, r0 = [], 0
        while r0 < rows_l:
            r1 = int(np.searchsorted(c_ip_h, c_ip_h[r0] + cap, side="right")) - 1
            if r1 <= r0:
                raise RuntimeError(f"row {r_lo + r0} of C has more entries than the output buffer holds")
            r1 = min(r1, rows_l)
            blocks.append((r0, r1))
            r0 = r1
        return blocks

    c_ip_pinned = torch.empty(rows_l + 1, dtype=torch.int64, pin_memory=True)   # the row pointers of C come back through pinned memory

    def step(verify=False):
        nnz_c, _ = ctx.spgemm_symbolic_dev(vA, vB, c_ip)
        c_ip_pinned.copy_(c_ip, non_blocking=True)
        torch.cuda.current_stream().synchronize()
        c_ip_h = c_ip_pinned.numpy()
        blocks = plan_blocks(c_ip_h)
        state.update(nnz_c=nnz_c, blocks=len(blocks), c_ip_h=c_ip_h)
        vsum, hsum = 0.0, 0
        for (b0, b1) in blocks:
            ctx.spgemm_numeric_dev(vA, vB, c_ip, b0, b1, c_ix, c_val)
            if verify:
                h, s = ctx.digest_dev(c_ix, c_val, int(c_ip_h[b1] - c_ip_h[b0]))
                vsum += s
                hsum = (hsum + h) & ((1 << 64) - 1)
        return vsum, hsum

    def sync_all():
        ctx.sync()                 # the library's own runtime handle of the stream
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    # ---- warm-up (first one verifies: all-ones inputs => sum of C's values == products) -------
    for w in range(max(args.warmup, 1)):
        vsum, hsum = step(verify=(w == 0))
        if w == 0 and os.environ.get("SPMM_BENCH_NOCHECK") != "1":    # NOCHECK: timing experiments of the diagnostic build, whose results are wrong
            assert vsum == float(P_local), f"rank {rank}: sum(C) = {vsum} but the rows hold {P_local} products"
    nnz_c_local = state["nnz_c"]

    # ---- timed region --------------------------------------------------------
    ctx.profile(True)
    sync_all()
    t1 = time.perf_counter()
    for _ in range(args.steps):
        step()
    sync_all()
    dt = time.perf_counter() - t1
    ph = {name: ctx.phase_ms(pid) for name, pid in
          [("bins", capi.PH_BINS), ("window_sym", capi.PH_WINDOW_SYM), ("esc_sym", capi.PH_ESC_SYM),
           ("window_num", capi.PH_WINDOW_NUM), ("esc_num", capi.PH_ESC_NUM), ("scan", capi.PH_SCAN),
           ("dense_num", capi.PH_DENSE_NUM), ("light_num", capi.PH_LIGHT_NUM), ("dense_sym", capi.PH_DENSE_SYM),
           ("slab_plan", capi.PH_SLAB_PLAN)]}
    ctx.profile(False)
    # one more step, untimed, with the sum check: the timed steps reuse what the library keeps between steps
    if os.environ.get("SPMM_BENCH_NOCHECK") != "1":
        vsum, _ = step(verify=True)
        assert vsum == float(P_local), f"rank {rank}: after the timed steps sum(C) = {vsum} but the rows hold {P_local} products"
    stamps = None
    if os.environ.get("SPMM_WIN_STAMPS") in ("1", "2", "3", "4"):
        st = ctx.debug_stamps().astype(np.float64)
        names = ["directory", "sweep_own", "sweep_wait", "prefix", "value_own", "value_wait", "output_rest", "output_vals", "windows", "products"]
        stamps = {k: {n: (round(float(v)) if n in ("windows", "products") else round(float(v) / max(st[o:o + 8].sum(), 1), 3))
                      for n, v in zip(names, st[o:o + 10])} for k, o in (("symbolic", 0), ("numeric", 16))}
        for nm, sl in (("sweep", 26), ("value", 27), ("dir", 28), ("postvalue", 29), ("stage", 30)):
            stamps["numeric"][f"skew_{nm}_barrier_cycles_per_window"] = round(float(st[sl] / max(st[24], 1)))
        stamps["numeric"]["cycles_per_window"] = round(float(st[16:24].sum() / max(st[24], 1)))
        # 1024-thread form only: its register-resident windows (the three skew slots above are then not skews)
        stamps["numeric"]["fast1024"] = dict(windows=int(st[28]), cycles=int(st[29]), products=int(st[30]))
        if os.environ.get("SPMM_WIN_STAMPS") == "2":     # register-resident windows of the 1024-thread form by products held
            stamps["numeric"]["fast1024_by_products"] = {nm: dict(windows=int(st[2 * i]), cycles=int(st[2 * i + 1]))
                                                         for i, nm in enumerate(("below_2048", "below_4096", "below_6144", "from_6144"))}
        if os.environ.get("SPMM_WIN_STAMPS") == "3":     # 256-thread workgroups only: set-up and whole-row cycles of thread 0
            stamps["numeric"]["small_workgroups"] = dict(rows=int(st[2]), setup_cycles_per_row=round(float(st[0] / max(st[2], 1))),
                                                         cycles_per_row=round(float(st[1] / max(st[2], 1))))
        stamps["symbolic"]["cycles_per_window"] = round(float(st[0:8].sum() / max(st[8], 1)))
        # the dense kernel uses slots 0..5: sweep, wait for the sweep barrier, flags and sums, wait for the
        # scan barrier, stores, windows
        dn = ["sweep", "sweep_wait", "flags_sums", "scan_wait", "stores"]
        stamps["dense"] = {n: round(float(st[i] / max(st[5], 1))) for i, n in enumerate(dn)}
        stamps["dense"]["windows"] = int(st[5])
        for nm, sl in (("below_1024", 6), ("below_2048", 8), ("below_4096", 10), ("from_4096", 12)):   # by entries of C
            stamps["dense"]["windows_" + nm] = dict(count=int(st[sl]), cycles_per_window=round(float(st[sl + 1] / max(st[sl], 1))))
    tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
    tot = torch.tensor([float(nnz_c_local)], dtype=torch.float64, device=dev)
    if dist is not None:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(tot, op=dist.ReduceOp.SUM)
    dt = float(tmax[0])
    nnz_c_total = int(tot[0])
    ms_per_step = dt / args.steps * 1e3
    gflops = 2.0 * P_total / (dt / args.steps) / 1e9

    if rank != 0:
        if dist is not None:
            dist.barrier()
            dist.destroy_process_group()
        return

    # ---- roofline of the dominant kernel (rank 0's share) ------------------------------
    # Numeric runs three kernels over the rows above the expand-sort-compress limit: k_dense over the
    # rows with at least dense_min products, k_win over the rest (as 256-thread workgroups for the
    # lightest).  The one with more time per step is reported, with the algorithmic bytes of its own rows:
    #   12 B per product (column id + value of B) + 20 B per entry of A (column id, value, two
    #   table bounds) + 12 B per row (row pointers of A and C) + 12 B per entry of C written.
    esc_l = ctx.sort_limit(n)
    shift, nbase, dense_min = ctx.plan_info(n)
    pl = prod[r_lo:r_hi]
    lens_l = (indptr[r_lo + 1:r_hi + 1] - indptr[r_lo:r_hi]).to(torch.int64)
    rowc = c_ip[1:] - c_ip[:-1]
    light_max = max(ctx.light_max(n), esc_l)
    slab_windows, slab_launches, slab_pieces, slab_widest = ctx.slab_info()
    dense_name = "k_dense"
    if slab_windows > 0:       # no window table: every window row went through the launches over B's slab order
        dense_name, dense_min = "k_slab_acc<numeric>", esc_l + 1
    masks = {dense_name: pl >= dense_min, "k_win<numeric>": (pl > light_max) & (pl < dense_min),
             "k_win<numeric, 256 threads>": (pl > esc_l) & (pl <= light_max) & (pl < dense_min)}
    times = {dense_name: ph["dense_num"], "k_win<numeric>": ph["window_num"], "k_win<numeric, 256 threads>": ph["light_num"]}
    kname = max(times, key=lambda k: times[k][0])
    wmask = masks[kname]
    P_w, nnzA_w, rows_w, nnzC_w = int(pl[wmask].sum()), int(lens_l[wmask].sum()), int(wmask.sum()), int(rowc[wmask].sum())
    bytes_w = 12 * P_w + 20 * nnzA_w + 12 * (rows_w + 1) + 12 * nnzC_w
    t_w_ms = times[kname][0] / args.steps
    # launches per step: the slab kernel is launched once per column window of B (that holds entries) in every numeric
    # call; the cursor kernels once per call
    calls = times[kname][1] // max(args.steps, 1)
    launches = calls * slab_launches if kname.startswith("k_slab") else calls
    achieved = bytes_w / (t_w_ms * 1e-3) / 1e9 if t_w_ms > 0 else 0.0
    # HBM bytes per launch from PMC counters cannot be collected inside this process: they come from
    # separate rocprofv3 --pmc passes over this same command, stored under profiles/ per workload.
    workload = f"R-MAT scale {args.scale} edge factor {args.edge_factor} seed {args.seed}, C=A*A, all values 1.0"
    # A traffic figure is only taken from a profile of THIS build (same digest of the library's sources); a profile of
    # another build is named, not used.
    traffic, traffic_profile = None, None
    import glob
    build = source_digest()
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "*pmc_traffic*.json"))):
        try:
            T = json.load(open(f))
            if T.get("workload") == workload and kname in T.get("kernels", {}) and world == 1:
                same_build = T.get("source_digest") == build
                traffic_profile = dict(file=os.path.basename(f), source_digest=T.get("source_digest"), same_build=same_build)
                traffic = T["kernels"][kname]["hbm_bytes_per_launch"] if same_build else None
        except (OSError, ValueError, KeyError):
            pass
    roofline = dict(bound="hbm", kernel=kname, achieved=round(achieved, 1), peak=HBM_PEAK_GBS, unit="GB/s",
                    frac=round(achieved / HBM_PEAK_GBS, 4), traffic=traffic, traffic_profile=traffic_profile, source_digest=build,
                    algorithmic_bytes_per_launch=bytes_w // max(launches, 1),
                    algorithmic_bytes_per_step=bytes_w, kernel_ms_per_step=round(t_w_ms, 3),
                    launches_per_step=launches, avg_launch_ms=round(t_w_ms / max(launches, 1), 5),
                    rows=rows_w, products=P_w, nnz_c=nnzC_w,
                    numeric_gflops=round(2.0 * P_w / (t_w_ms * 1e-3) / 1e9, 1) if t_w_ms > 0 else None,
                    table=dict(shift=shift, windows=nbase, dense_min_products=dense_min),
                    slab=dict(windows=slab_windows, launched_windows=slab_launches, pieces=slab_pieces, widest_window=slab_widest))

    # ---- the two phases as wholes (rank 0's rows), against the same 8 TB/s --------------------------
    # symbolic: 4 B per product (column id of B) + 12 B per entry of A + row pointers + 8 B per row count;
    # numeric: 12 B per product + 20 B per entry of A + row pointers of A and C + 12 B per entry of C.
    nnzA_l, nnzC_l = int(lens_l.sum()), int(rowc.sum())
    t_num = sum(ph[k][0] for k in ("dense_num", "window_num", "light_num", "esc_num")) / args.steps
    t_sym = sum(ph[k][0] for k in ("window_sym", "esc_sym", "scan", "dense_sym", "slab_plan")) / args.steps
    b_num = 12 * P_local + 20 * nnzA_l + 12 * (rows_l + 1) + 12 * nnzC_l
    b_sym = 4 * P_local + 12 * nnzA_l + 4 * (rows_l + 1) + 8 * rows_l
    phase_totals = {
        "numeric": dict(ms_per_step=round(t_num, 3), gflops=round(2.0 * P_local / (t_num * 1e-3) / 1e9, 1) if t_num > 0 else None,
                        algorithmic_bytes=b_num, frac_of_hbm_peak=round(b_num / (t_num * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if t_num > 0 else None),
        "symbolic": dict(ms_per_step=round(t_sym, 3), algorithmic_bytes=b_sym,
                         frac_of_hbm_peak=round(b_sym / (t_sym * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if t_sym > 0 else None),
        "bins_ms_per_step": round(ph["bins"][0] / args.steps, 3)}

    # ---- CPU baseline: the oracle's Gustavson on a sample of rows of the same matrix, spread over the product
    # distribution (every 1/S-th point of the cumulative products, so heavy and light rows weigh as in the whole job),
    # on one core and row-parallel on the host cores of this box's share; the pre-processing pipeline beside it ----
    cpu, pre = None, None
    if not args.no_cpu and world == 1:
        from oracle import oracle as orc
        threads = max(1, min(int(os.environ.get("OMP_NUM_THREADS", "16")), os.cpu_count() or 1, 16))
        prod_h = prod.cpu().numpy()
        order = np.argsort(prod_h, kind="stable")
        cum_sorted = np.cumsum(prod_h[order])
        mean_hit = float((prod_h.astype(np.float64) ** 2).sum() / max(cum_sorted[-1], 1))   # products of the row under a random product
        S = max(8, int(args.cpu_products / max(mean_hit, 1.0)))
        picks = np.unique(np.searchsorted(cum_sorted, (np.arange(S) + 0.5) * (cum_sorted[-1] / S)))
        rows_s = np.sort(order[picks]).astype(np.int32)
        ip_h, ix_h = indptr.cpu().numpy(), indices.cpu().numpy()
        d_h = np.ones(nnz, np.float64)
        M = (ip_h, ix_h, d_h)
        tc = time.perf_counter()
        cnt1, vs1, p_s = orc.spgemm_rows(n, M, M, rows_s, 1)
        t1c = time.perf_counter() - tc
        tc = time.perf_counter()
        cntN, vsN, p_n = orc.spgemm_rows(n, M, M, rows_s, threads)
        tNc = time.perf_counter() - tc
        assert p_s == p_n == int(prod_h[rows_s].sum()) and vs1 == float(p_s) and np.array_equal(cnt1, cntN)
        cpu = dict(value=round(2.0 * p_s / t1c / 1e9, 4), unit="GFLOP/s", cores=1, kind="port",
                   sample=f"{rows_s.size} rows of the same matrix at equal steps of the cumulative product count "
                          f"({p_s} products, {int(cnt1.sum())} entries of C, one pass that counts and sums, {t1c:.1f} s)",
                   all_cores=dict(value=round(2.0 * p_s / tNc / 1e9, 4), cores=threads, seconds=round(tNc, 2)),
                   host_cpus=os.cpu_count())
        # the same rows from the GPU: row counts bit for bit; the heaviest sampled row entry for entry
        if r_lo == 0 and rows_l == n:
            assert np.array_equal(np.diff(state["c_ip_h"])[rows_s], cnt1), "GPU row counts differ from the oracle"
            rh = int(rows_s[np.argmax(prod_h[rows_s])])
            o_ip1 = np.array([0, int(cnt1[np.argmax(prod_h[rows_s])])], np.int64)
            if int(o_ip1[1]) <= cap:
                sub = (np.array([0, ip_h[rh + 1] - ip_h[rh]], np.int32), ix_h[ip_h[rh]:ip_h[rh + 1]], d_h[:ip_h[rh + 1] - ip_h[rh]])
                oi, od, _ = orc.spgemm_numeric(n, sub, M, o_ip1)
                ctx.spgemm_numeric_dev(vA, vB, c_ip, rh, rh + 1, c_ix, c_val)
                sync_all()
                k = int(o_ip1[1])
                assert np.array_equal(c_ix[:k].cpu().numpy(), oi) and np.array_equal(c_val[:k].cpu().numpy(), od), \
                    "the heaviest sampled row differs from the oracle"
            cpu["gpu_rows_match_oracle"] = True
        # pre-processing (the reference's own metric: the stages of gen_trace_formats, serial_newblock_clock.cpp:565)
        A_h = capi.Csr(n, n, ip_h, ix_h, d_h)
        ctx.trace_formats(A_h)                                    # warm-up: scratch allocations
        tw = time.perf_counter()
        g = ctx.trace_formats(A_h)
        wall_ms = (time.perf_counter() - tw) * 1e3
        gpu_ms = ctx.last_kernel_ms()
        tc = time.perf_counter()
        o = orc.trace_formats(n, ip_h, ix_h, d_h, sort_mode=1)
        cpu_ms = (time.perf_counter() - tc) * 1e3
        same = all(np.array_equal(g[k], o[k]) for k in ("seq_bitmap", "bsize_list", "seq_offset", "seq_input", "rseq"))
        pre = dict(gpu_ms=round(gpu_ms, 2), gpu_wall_ms_host_arrays_in_and_out=round(wall_ms, 1), cpu_oracle_ms=round(cpu_ms, 1), cores=1,
                   blocks=int(g["bnum"]), gpu_equals_oracle=bool(same),
                   what="spmm_trace_formats on the same matrix: bitmap reorder + blocks + panels + pack + renumber + write-back sequence")
        del g, o, A_h
        # the compiled reference itself (oracle/_ref/libref.so, built from the reference's own sources where the
        # reference tree was present) beside the GPU, on the same matrix WITH its diagonal: the reference's panel sort
        # overflows a scratch array on matrices with empty rows (v8sort.h:157-160), and this graph has 2 million
        from oracle import ref
        if ref.available() and args.scale <= 22:
            dg = torch.arange(n, dtype=torch.int64, device=dev)
            rows_t = torch.repeat_interleave(dg, (indptr[1:] - indptr[:-1]).to(torch.int64))
            keyd = torch.unique(torch.cat([indices.to(torch.int64) * n + rows_t, dg * n + dg]))   # (col, row) order, as the files
            r_d, c_d = (keyd % n).to(torch.int32).cpu().numpy(), (keyd // n).to(torch.int32).cpu().numpy()
            del keyd, rows_t, dg
            M_d = ctx.csr_from_coo(n, n, r_d, c_d)
            ctx.trace_formats(M_d)
            tw = time.perf_counter()
            gd = ctx.trace_formats(M_d)
            wall_d = (time.perf_counter() - tw) * 1e3
            gpu_d = ctx.last_kernel_ms()
            ref.set_csr(n, M_d.indptr, M_d.indices, M_d.data)
            tc = time.perf_counter()
            rd = ref.trace_formats(n, int(M_d.indices.size))
            ref_ms = (time.perf_counter() - tc) * 1e3
            pre["reference"] = dict(kind="reference", cores=1, value_ms=round(ref_ms, 1),
                                    matrix="the same matrix with its diagonal added (no empty rows)", nnz=int(M_d.indices.size),
                                    gpu_ms=round(gpu_d, 2), gpu_wall_ms_host_arrays_in_and_out=round(wall_d, 1),
                                    same_blocks=bool(np.array_equal(gd["bsize_list"], rd["bsize_list"]) and
                                                     np.array_equal(gd["seq_offset"], rd["seq_offset"])),
                                    what="gen_trace_formats of the compiled reference, its own stage timers' scope "
                                         "(serial_newblock_clock.cpp:565), one thread")
            del gd, rd, M_d

    out = dict(metric="spgemm_gflops", value=round(gflops, 2), unit="GFLOP/s", n_gpus=world, steps=args.steps,
               warmup=max(args.warmup, 1), ms_per_step=round(ms_per_step, 3), higher_is_better=True, scaling="strong",
               vs_baseline=None, dtype="f64", data="synthetic",
               config=dict(workload=workload,
                           rows=n, nnz=nnz, products=P_total, nnz_c=nnz_c_total, partition="rows of A by product count weighted by kernel class",
                           step="symbolic+numeric", c_row_blocks_per_step=state["blocks"], rank0_rows=rows_l,
                           b_allgather_ms=round(b_allgather_ms, 3), setup_s=round(setup_s, 1)),
               phases_ms_per_step={k: round(v[0] / args.steps, 3) for k, v in ph.items()},
               phase_totals_rank0=phase_totals, roofline=roofline, cpu_baseline=cpu, preprocess=pre)
    if stamps:
        out["stamps"] = stamps
    print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
