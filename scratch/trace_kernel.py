"""Print the per-dispatch durations (us) of kernels whose name contains argv[2] from a rocprofv3 kernel_trace.csv."""
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if sys.argv[2] in r["Kernel_Name"]]
d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows]
print(len(d), "dispatches; us:", " ".join(f"{x:.0f}" for x in d[:int(sys.argv[3]) if len(sys.argv) > 3 else 60]))
