"""From a rocprofv3 kernel-trace CSV: do launches of one kernel overlap in time (two decode chains on two streams)?"""
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if sys.argv[2] in r["Kernel_Name"]]
iv = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Queue_Id", r.get("Stream_Id", "?"))) for r in rows)
print(len(iv), "launches; queues:", sorted({q for _, _, q in iv}))
ov = 0
for (s0, e0, q0), (s1, e1, q1) in zip(iv, iv[1:]):
    if s1 < e0:
        ov += 1
print("consecutive launches overlapping in time:", ov)
for s, e, q in iv[40:52]:
    print(q, (s - iv[40][0]) / 1e3, (e - s) / 1e3)
