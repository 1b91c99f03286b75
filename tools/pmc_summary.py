#!/usr/bin/env python3
"""Sum rocprofv3 --pmc counter_collection CSVs per kernel (short name) and counter."""
import csv
import glob
import os
import re
import sys
from collections import defaultdict

out = sys.argv[1]
tot = defaultdict(lambda: defaultdict(float))
calls = defaultdict(lambda: defaultdict(int))
for f in glob.glob(os.path.join(out, "**", "*counter_collection.csv"), recursive=True):
    with open(f) as fh:
        for r in csv.DictReader(fh):
            name = r.get("Kernel_Name", "")
            m = re.search(r"(k_[a-z_0-9]+(<[^>]*>)?)", name)
            short = m.group(1) if m else name[:40]
            tot[short][r["Counter_Name"]] += float(r["Counter_Value"])
            calls[short][r["Counter_Name"]] += 1
for k in sorted(tot, key=lambda k: -tot[k].get("SQ_WAVE_CYCLES", tot[k].get("FETCH_SIZE", 0))):
    if not k.startswith("k_"):
        continue
    print(k)
    for c, v in sorted(tot[k].items()):
        print(f"    {c:28s} {v:18.0f}   dispatches {calls[k][c]}")
