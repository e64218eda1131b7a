"""rocprofv3 kernel-trace CSV of `bench.py` (XE) -> timeline of the LAST step: span, per-queue busy time, time with 0 / 1 / 2+
kernels running, the largest idle gaps, and per-kernel-name time on each queue."""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Queue_Id"], r["Kernel_Name"]) for r in rows)
# steps end with adam_clip_kernel
ends = [i for i, e in enumerate(ev) if "adam_clip" in e[3]]
k = int(sys.argv[2]) if len(sys.argv) > 2 else 4        # step index (0-based); the last steps of a bench run are the serialised roofline steps
APS = int(sys.argv[3]) if len(sys.argv) > 3 else 2      # Adam launches per step (2: the decoder half goes early on its own stream)
a, b = ends[APS * k - 1] + 1, ends[APS * (k + 1) - 1] + 1
g = ev[a:b]
t0, t1 = g[0][0], max(e[1] for e in g)
print("kernels in the step:", len(g), "span ms %.3f" % ((t1 - t0) / 1e6))
pts = []
for s, e, q, n in g:
    pts.append((s, 1)); pts.append((e, -1))
pts.sort()
lvl, last, hist = 0, t0, collections.Counter()
for t, d in pts:
    hist[min(lvl, 3)] += t - last
    lvl += d; last = t
for k in sorted(hist): print("  time with %d%s kernels running: %.3f ms" % (k, "+" if k == 3 else "", hist[k] / 1e6))
for q in sorted({e[2] for e in g}):
    k = [e for e in g if e[2] == q]
    print("queue", q, "kernels", len(k), "busy ms %.3f" % (sum(e[1] - e[0] for e in k) / 1e6), "first %.3f last end %.3f" % ((k[0][0] - t0) / 1e6, (k[-1][1] - t0) / 1e6))
    by = collections.defaultdict(lambda: [0, 0])
    for s, e, _, n in k:
        by[n[:60]][0] += 1; by[n[:60]][1] += e - s
    for n, (c, t) in sorted(by.items(), key=lambda x: -x[1][1])[:12]:
        print("     %-60s %4d  %.3f ms  avg %.1f us" % (n, c, t / 1e6, t / c / 1e3))
# idle gaps (no kernel at all)
pts2 = sorted((s, e) for s, e, _, _ in g)
cur_end = pts2[0][1]; gaps = []
for s, e in pts2[1:]:
    if s > cur_end: gaps.append((s - cur_end, cur_end - t0))
    cur_end = max(cur_end, e)
gaps.sort(reverse=True)
print("idle gaps: total %.3f ms in %d gaps; largest (us @ ms):" % (sum(x[0] for x in gaps) / 1e6, len(gaps)), [(round(x[0] / 1e3, 1), round(x[1] / 1e6, 2)) for x in gaps[:8]])
