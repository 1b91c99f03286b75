"""List the durations of every dispatch of one kernel from a rocprofv3 --kernel-trace CSV, in launch order.
usage: trace_durations.py <dir> <kernel-name-prefix> [grid-size column shown too]"""
import csv, glob, sys

rows = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if r["Kernel_Name"].startswith(sys.argv[2]):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Grid_Size_X", r.get("Grid_Size", "?"))))
rows.sort()
for s, e, g in rows:
    print("%10.3f ms  grid %s" % ((e - s) / 1e6, g))
print("total %.3f ms over %d dispatches" % (sum(e - s for s, e, _ in rows) / 1e6, len(rows)))
