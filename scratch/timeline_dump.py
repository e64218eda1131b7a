"""rocprofv3 kernel-trace CSV of `bench.py` (XE) -> every kernel of ONE step in start order: queue, start (us), duration, gap to the
previous kernel of its queue, grid size, name.  argv: csv [step index] [adam launches per step]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Queue_Id"], r["Kernel_Name"], r.get("Grid_Size", "")) for r in rows)
ends = [i for i, e in enumerate(ev) if "adam_clip" in e[3]]
k = int(sys.argv[2]) if len(sys.argv) > 2 else 4
APS = int(sys.argv[3]) if len(sys.argv) > 3 else 2
a, b = ends[APS * k - 1] + 1, ends[APS * (k + 1) - 1] + 1
g = ev[a:b]
t0 = g[0][0]
last = {}
qs = sorted({e[2] for e in g})
for s, e, q, n, grid in g:
    gap = (s - last[q]) / 1e3 if q in last else 0.0
    last[q] = e
    short = n.replace("(anonymous namespace)::", "").replace("_ZN12_GLOBAL__N_1", "")[:70]
    print(f"q{qs.index(q)} {(s - t0) / 1e3:9.1f} +{(e - s) / 1e3:7.1f} us  gap {gap:6.1f}  grid {grid:>8}  {short}")
