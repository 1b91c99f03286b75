#!/usr/bin/env python3
"""HBM bytes per launch of the sorting kernels from FETCH_SIZE / WRITE_SIZE passes over tools/bench_snap_like.py
(rocprofv3 --pmc, one pass per counter set).  FETCH_SIZE and WRITE_SIZE count kilobytes; on gfx950 FETCH_SIZE counts
a 128-byte request as 64 bytes, so reads are doubled.  usage: pmc_esc.py <pmc output dir>"""
import csv
import glob
import json
import os
import re
import sys
from collections import defaultdict

tot = defaultdict(lambda: defaultdict(float))
calls = defaultdict(lambda: defaultdict(int))
for f in glob.glob(os.path.join(sys.argv[1], "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        m = re.search(r"(k_esc4<[^>]*>|k_escr<[^>]*>|k_esc<[^>]*>)", r["Kernel_Name"])
        if m:
            tot[m.group(1)][r["Counter_Name"]] += float(r["Counter_Value"])
            calls[m.group(1)][r["Counter_Name"]] += 1
out = {}
for k in sorted(tot):
    n = max(calls[k]["FETCH_SIZE"], 1)
    rd = tot[k]["FETCH_SIZE"] * 1024 * 2 / n
    wr = tot[k]["WRITE_SIZE"] * 1024 / max(calls[k]["WRITE_SIZE"], 1)
    out[k] = dict(launches=n, read_bytes_per_launch=int(rd), write_bytes_per_launch=int(wr))
print(json.dumps(dict(note="launches of both graphs of bench_snap_like.py together; the last template argument is the phase "
                           "(true = numeric)", kernels=out), indent=1))
