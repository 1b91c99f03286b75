"""Per-window durations of the slab kernels from a rocprofv3 --kernel-trace CSV: the launches of one numeric or
symbolic pass come in window order (discover, accumulate, discover, ...), so pair t of a pass is window t.
usage: slab_window_times.py <trace dir> <numeric|symbolic> <windows per pass>"""
import csv, glob, sys

want = "true" if sys.argv[2] == "numeric" else "false"
nwin = int(sys.argv[3])
ev = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "k_slab_" in k:
            ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "disc" if "k_slab_disc" in k else ("acc_" + ("true" if "true>" in k or "Lb1" in k else "false")), "compact" in k,
                       int(r.get("Grid_Size_X", 0))))
ev.sort()
pairs = []
last_disc = None
for s, e, kind, cmpk, g in ev:
    if kind == "disc":
        last_disc = (s, e)
    elif kind == "acc_" + want and last_disc:
        pairs.append((last_disc[1] - last_disc[0], e - s, g, s - last_disc[1], cmpk))
        last_disc = None
disc = [0.0] * nwin
acc = [0.0] * nwin
gap = 0.0
for i, (d, a, g, gp, ck) in enumerate(pairs):
    disc[i % nwin] += d / 1e3
    acc[i % nwin] += a / 1e3
    gap += gp / 1e3
for p0 in range(0, len(pairs), nwin):
    pp = pairs[p0:p0 + nwin]
    print("pass", p0 // nwin, "disc ms", round(sum(x[0] for x in pp) / 1e6, 1), "acc ms", round(sum(x[1] for x in pp) / 1e6, 1),
          "acc max us", round(max(x[1] for x in pp) / 1e3), "acc min us", round(min(x[1] for x in pp) / 1e3))
print("pairs", len(pairs), "passes", len(pairs) / nwin, "disc us", round(sum(disc)), "acc us", round(sum(acc)), "gaps disc->acc us", round(gap))
for w in range(nwin):
    print(w, round(disc[w], 1), round(acc[w], 1), pairs[w][2] if w < len(pairs) else 0, int(pairs[w][4]) if w < len(pairs) else 0)
