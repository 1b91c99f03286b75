#!/usr/bin/env python3
"""HBM bytes per launch of the two numeric window kernels from the FETCH_SIZE / WRITE_SIZE passes of
tools/prof_pmc.sh.  FETCH_SIZE and WRITE_SIZE count kilobytes; on gfx950 FETCH_SIZE counts a 128-byte
request as 64 bytes, so reads are doubled (k_digest streams 12 bytes per entry of C exactly once and
serves as the calibration).
usage: pmc_traffic.py <pmc output dir> "<workload string of bench.py>" "<command>" > profiles/..json"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

out, workload, command = sys.argv[1:4]
tot = defaultdict(lambda: defaultdict(float))
calls = defaultdict(lambda: defaultdict(int))
for f in glob.glob(os.path.join(out, "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"]
        short = ("k_dense" if "k_dense<" in name else "k_win<numeric>" if "k_win<1024, 8192, 8192" in name else
                 "k_win<numeric, 256 threads>" if "k_win<256, " in name else
                 "k_digest" if "k_digest" in name else None)
        if short:
            tot[short][r["Counter_Name"]] += float(r["Counter_Value"])
            calls[short][r["Counter_Name"]] += 1
kernels = {}
for k in ("k_dense", "k_win<numeric>", "k_win<numeric, 256 threads>", "k_digest"):
    if k not in tot:
        continue
    n = calls[k]["FETCH_SIZE"]
    rd = tot[k]["FETCH_SIZE"] * 1024 * 2 / n
    wr = tot[k]["WRITE_SIZE"] * 1024 / calls[k]["WRITE_SIZE"]
    kernels[k] = dict(fetch_kb_total=int(tot[k]["FETCH_SIZE"]), write_kb_total=int(tot[k]["WRITE_SIZE"]), launches=n,
                      read_bytes_per_launch_fetch_size_doubled=int(rd), write_bytes_per_launch=int(wr))
    if "TCC_EA0_RDREQ_sum" in tot[k]:
        # read requests of the L2 to the fabric by size: the byte count needs no correction
        m = calls[k]["TCC_EA0_RDREQ_sum"]
        n_all, n32, n64, n128 = (tot[k][c] / m for c in ("TCC_EA0_RDREQ_sum", "TCC_EA0_RDREQ_32B_sum", "TCC_EA0_RDREQ_64B_sum",
                                                         "TCC_EA0_RDREQ_128B_sum"))
        rd = 32 * n32 + 64 * n64 + 128 * n128
        kernels[k].update(read_requests_per_launch=int(n_all), read_requests_32B=int(n32), read_requests_64B=int(n64),
                          read_requests_128B=int(n128), read_bytes_per_launch=int(rd))
    if "TCC_EA0_RDREQ_DRAM_sum" in tot[k]:
        m = calls[k]["TCC_EA0_RDREQ_DRAM_sum"]
        kernels[k].update(read_requests_dram_per_launch=int(tot[k]["TCC_EA0_RDREQ_DRAM_sum"] / m),
                          write_requests_per_launch=int(tot[k]["TCC_EA0_WRREQ_sum"] / m),
                          write_requests_64B_per_launch=int(tot[k]["TCC_EA0_WRREQ_64B_sum"] / m))
    kernels[k]["hbm_bytes_per_launch"] = int(rd + wr)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402  (for the digest of the library's sources: bench.py takes a traffic figure only from a profile of its own build)
print(json.dumps(dict(workload=workload, command=command, source_digest=bench.source_digest(),
                      note="rocprofv3 --pmc, separate passes; read bytes from the request-size split of TCC_EA0_RDREQ when that pass "
                           "was taken (32/64/128-byte requests), else FETCH_SIZE doubled (gfx950 counts 128-byte requests as 64 "
                           "bytes); calibration: k_digest streams 12 B per entry of C once",
                      kernels=kernels), indent=1))
