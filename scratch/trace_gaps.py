"""From a rocprofv3 kernel-trace CSV: idle gaps longer than argv[2] ms (default 3) with the kernels on either side, after the last
gap longer than 200 ms (a sleep the traced script puts in front of the steps of interest)."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
thr = float(sys.argv[2]) if len(sys.argv) > 2 else 3.0
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:64]) for r in rows))
start = 0
end = ev[0][1]
for i, (s, e, n) in enumerate(ev[1:], 1):
    if s - end > 200e6: start = i
    end = max(end, e)
ev = ev[start:]
t0 = ev[0][0]
print("kernels", len(ev), "span ms", round((max(e for _, e, _ in ev) - t0) / 1e6, 2))
end = ev[0][1]
for i, (s, e, n) in enumerate(ev[1:], 1):
    if s - end > thr * 1e6:
        print(f"GAP {(s - end) / 1e6:7.2f} ms at {(end - t0) / 1e6:8.2f}")
        for j in range(max(0, i - 3), min(len(ev), i + 3)):
            print(f"      {'>' if j == i else ' '} {(ev[j][0] - t0) / 1e6:8.2f} +{(ev[j][1] - ev[j][0]) / 1e3:7.1f} us  {ev[j][2]}")
    end = max(end, e)
