"""rocprofv3 kernel-trace CSV -> per-queue timeline of the LAST decode in the file: span, busy time, biggest gaps."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Queue_Id"], r["Kernel_Name"][:50]) for r in rows)
# split into decodes by big idle gaps (> 2 ms) over all queues
groups, cur, last_end = [], [], None
for e in ev:
    if last_end is not None and e[0] - last_end > 2_000_000:
        groups.append(cur); cur = []
    cur.append(e); last_end = max(last_end or 0, e[1])
groups.append(cur)
g = groups[-1]
t0 = g[0][0]
print("decodes in trace:", len(groups), "| last one: kernels", len(g), "span ms", (max(e[1] for e in g) - t0) / 1e6)
for q in sorted({e[2] for e in g}):
    k = [e for e in g if e[2] == q]
    busy = sum(e[1] - e[0] for e in k) / 1e6
    print("queue", q, "kernels", len(k), "first at ms", (k[0][0] - t0) / 1e6, "last end ms", (k[-1][1] - t0) / 1e6, "busy ms", round(busy, 2))
    gaps = sorted(((b[0] - a[1]) / 1e3, a[3], b[3]) for a, b in zip(k, k[1:]))[-4:]
    for gp in gaps:
        print("    gap us", round(gp[0], 1), "after", gp[1], "before", gp[2])
