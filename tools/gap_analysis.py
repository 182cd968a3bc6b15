"""Idle time between consecutive kernels of a rocprofv3 kernel trace (one stream): where does the step lose time that is not
kernel time?  usage: python tools/gap_analysis.py <kernel_trace.csv> [min_gap_us]"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
thr = float(sys.argv[2]) if len(sys.argv) > 2 else 8.0
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# keep the second half of the run (steady-state steps)
rows = rows[len(rows) // 2:]
tot_k = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rows) / 1e3
span = (int(rows[-1]["End_Timestamp"]) - int(rows[0]["Start_Timestamp"])) / 1e3
gaps = collections.Counter(); cnt = collections.Counter(); small = 0.0
for a, b in zip(rows, rows[1:]):
    g = (int(b["Start_Timestamp"]) - int(a["End_Timestamp"])) / 1e3
    if g > thr:
        key = (a["Kernel_Name"][:60], b["Kernel_Name"][:60])
        gaps[key] += g; cnt[key] += 1
    elif g > 0:
        small += g
print(f"span {span / 1e3:.1f} ms, kernel time {tot_k / 1e3:.1f} ms, idle {(span - tot_k) / 1e3:.1f} ms; gaps <= {thr} us sum to {small / 1e3:.2f} ms over {len(rows)} kernels")
for k, v in gaps.most_common(25):
    print(f"{v / 1e3:7.2f} ms {cnt[k]:4d}x  after {k[0]}\n                    before {k[1]}")
