"""Print a compact per-kernel summary of a rocprofv3 *_kernel_stats.csv (ms per step)."""
import csv, sys
path, steps = sys.argv[1], float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
rows = list(csv.DictReader(open(path)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(f"total kernel time: {tot/1e6/steps:.1f} ms/step over {steps:g} steps")
for r in rows[:int(sys.argv[3]) if len(sys.argv) > 3 else 30]:
    print(f"{float(r['TotalDurationNs'])/1e6/steps:9.2f} ms/step {int(r['Calls'])/steps:7.1f} calls/step  avg {float(r['AverageNs'])/1e3:9.1f} us  {float(r['Percentage']):6.2f}%  {r['Name'][:120]}")
