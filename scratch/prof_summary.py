"""Condense a rocprofv3 kernel-stats CSV into per-kernel lines (name shortened)."""
import csv, sys, re
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
steps = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
print(f"total kernel time {tot/1e6:.2f} ms over {steps:g} steps -> {tot/1e6/steps:.3f} ms/step")
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:40]:
    n = re.sub(r"\(anonymous namespace\)::", "", r["Name"])
    n = re.sub(r"void ", "", n)[:110]
    print(f'{float(r["TotalDurationNs"])/1e6/steps:9.3f} ms/step {int(r["Calls"])/steps:8.1f} calls/step {float(r["AverageNs"])/1e3:9.1f} us avg {float(r["Percentage"]):6.2f}%  {n}')
