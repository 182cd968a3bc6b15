"""How much of a step runs with more than one kernel in flight?  Input: a rocprofv3 --kernel-trace CSV (kernel_trace.csv with
Start_Timestamp / End_Timestamp per dispatch).  Prints the sum of kernel durations, the length of their union (device busy
time) and the time with >= 2 kernels in flight, per step, plus the kernels that overlap most.
usage: python tools/overlap_report.py <kernel_trace.csv> <steps>"""
import csv
import collections
import re
import sys

path, steps = sys.argv[1], float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
rows = list(csv.DictReader(open(path)))
ev = []
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"])[:60]
    ev.append((s, e, name, r.get("Queue_Id", "?")))
ev.sort()
tot = sum(e - s for s, e, _, _ in ev)
pts = []
for s, e, n, q in ev:
    pts.append((s, 1)); pts.append((e, -1))
pts.sort()
busy = multi = 0
depth, last = 0, None
for t, d in pts:
    if last is not None and depth > 0:
        busy += t - last
        if depth > 1:
            multi += t - last
    depth += d
    last = t
span = ev[-1][1] - ev[0][0]
print(f"kernels {len(ev) / steps:.0f}/step; sum of durations {tot / 1e6 / steps:.2f} ms/step; device busy (union) {busy / 1e6 / steps:.2f} ms/step; "
      f">= 2 kernels in flight {multi / 1e6 / steps:.2f} ms/step; queues: {sorted(set(q for *_, q in ev))}")
# per kernel name: time during which it overlapped with a kernel of ANOTHER queue
ov = collections.Counter()
act = []
j = 0
for i, (s, e, n, q) in enumerate(ev):
    act = [(s2, e2, n2, q2) for (s2, e2, n2, q2) in act if e2 > s]
    for (s2, e2, n2, q2) in act:
        if q2 != q:
            o = min(e, e2) - s
            if o > 0:
                ov[n] += o; ov[n2] += o
    act.append((s, e, n, q))
print("most-overlapped kernels (ms/step overlapped with another queue):")
for n, v in ov.most_common(14):
    print(f"  {v / 1e6 / steps:7.2f}  {n}")

# ---- occupancy table of the kernels with the largest summed duration: waves per SIMD the launch configuration allows
# (LDS: 160 KiB per CU; VGPRs: 512 per SIMD lane incl. accumulation registers; 4 SIMDs per CU)
agg = {}
for r in rows:
    n = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"])[:70]
    a = agg.setdefault(n, [0, 0, r])
    a[0] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"]); a[1] += 1
print("\nlaunch configuration of the 30 largest kernels (ms/step, LDS bytes, VGPRs incl. AGPRs, workgroup, workgroups; waves per SIMD allowed by LDS / by registers):")
for n, (t, c, r) in sorted(agg.items(), key=lambda kv: -kv[1][0])[:30]:
    lds = int(r["LDS_Block_Size"]); vg = int(r["VGPR_Count"]) + int(r.get("Accum_VGPR_Count", 0) or 0)
    wg = int(r["Workgroup_Size_X"]) * int(r["Workgroup_Size_Y"]) * int(r["Workgroup_Size_Z"])
    nwg = (int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) * int(r["Grid_Size_Z"])) // max(wg, 1)
    waves = wg // 64
    by_lds = (160 * 1024 // lds) * waves / 4.0 if lds else 99.0
    by_reg = 512 // max(vg, 1)
    print(f"  {t / 1e6 / steps:7.2f} ms  lds {lds:7d}  vgpr {vg:4d}  wg {wg:5d}  n {nwg:7d}   {min(by_lds, 8):4.1f} / {min(by_reg, 8):2d}   {n}")
