#!/usr/bin/env python3
"""Resident waves per CU and issue/wait shares per kernel from the SQ passes of tools/prof_pmc.sh (sq1, sq2, sq3).
SQ_WAVE_CYCLES and SQ_BUSY_CYCLES count in units of four clocks; GRBM_GUI_ACTIVE is summed over the 8 XCDs.
waves per CU = 4 * SQ_WAVE_CYCLES / (GRBM_GUI_ACTIVE / 8 * 256 CUs); the limit is what LDS and VGPR allocation allow
(given per kernel below from the compiler's resource remarks and the launch's LDS size).
usage: pmc_occupancy.py <pmc output dir> "<command>" > profiles/..json"""
import csv
import glob
import json
import os
import re
import sys
from collections import defaultdict

out, command = sys.argv[1:3]
tot = defaultdict(lambda: defaultdict(float))
calls = defaultdict(lambda: defaultdict(int))
for f in glob.glob(os.path.join(out, "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        m = re.search(r"(k_[a-z_0-9]+(<[^>]*>)?)", r["Kernel_Name"])
        if m:
            tot[m.group(1)][r["Counter_Name"]] += float(r["Counter_Value"])
            calls[m.group(1)][r["Counter_Name"]] += 1
# allocation limits: (workgroups per CU, waves per workgroup, VGPRs, LDS bytes per workgroup)
LIMITS = {"k_dense<1024, false, true>": (1, 16, 121, 151808), "k_win<1024, 8192, 8192, 8, 4, true, false, false>": (1, 16, 127, 155000),
          "k_win<256, 2048, 2048, 8, 4, true, false, true>": (3, 4, 145, 51300), "k_sym2": (6, 4, 56, 25712),
          "k_symd<1024>": (1, 16, 105, 135184), "k_escr<512, 4096, true>": (2, 8, 43, 76000),
          "k_escr<512, 4096, false>": (4, 8, 38, 35000)}
res = {}
for k, c in tot.items():
    if "SQ_WAVE_CYCLES" not in c or "GRBM_GUI_ACTIVE" not in c:
        continue
    per = lambda n: c[n] / max(calls[k][n], 1)
    gui = per("GRBM_GUI_ACTIVE") / 8.0
    if gui < 1e6:
        continue
    waves_per_cu = 4.0 * per("SQ_WAVE_CYCLES") / (gui * 256.0)
    d = dict(dispatches=calls[k]["GRBM_GUI_ACTIVE"], clocks_per_dispatch=int(gui), waves_launched_per_dispatch=int(per("SQ_WAVES")),
             resident_waves_per_cu=round(waves_per_cu, 2), resident_waves_per_simd=round(waves_per_cu / 4, 2))
    if k in LIMITS:
        wg, wv, vg, lds = LIMITS[k]
        d.update(limit_waves_per_cu=wg * wv, limit_from=f"{wg} workgroup(s) of {wv} waves per CU (LDS {lds} B per workgroup, VGPRs {vg})")
    if "SQ_WAIT_ANY" in c:
        d.update(wave_cycles_waiting=round(per("SQ_WAIT_ANY") / per("SQ_WAVE_CYCLES"), 3),
                 wave_cycles_waiting_on_lds=round(per("SQ_WAIT_INST_LDS") / per("SQ_WAVE_CYCLES"), 3))
    if "SQ_ACTIVE_INST_VALU" in c:
        d.update(simd_cycles_issuing_valu=round(4.0 * per("SQ_ACTIVE_INST_VALU") / (gui * 1024.0), 3))
    if "SQ_INSTS_VALU" in c:
        d.update(valu_wave_instructions=int(per("SQ_INSTS_VALU")), salu_wave_instructions=int(per("SQ_INSTS_SALU")),
                 lds_wave_instructions=int(per("SQ_INSTS_LDS")), vmem_read_wave_instructions=int(per("SQ_INSTS_VMEM_RD")))
    if "SQ_LDS_BANK_CONFLICT" in c:
        d.update(lds_active_cycles_share=round(per("SQ_LDS_IDX_ACTIVE") / (gui * 256.0), 3),
                 lds_bank_conflict_share_of_active=round(per("SQ_LDS_BANK_CONFLICT") / max(per("SQ_LDS_IDX_ACTIVE"), 1), 3))
    res[k] = d
print(json.dumps(dict(command=command, note=__doc__.split("usage")[0].strip(), kernels=res), indent=1))
