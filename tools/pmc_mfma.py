"""Matrix-pipe utilisation per kernel from a rocprofv3 --pmc pass (SQ_VALU_MFMA_BUSY_CYCLES, GRBM_GUI_ACTIVE):
MfmaUtil = 100 * SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE * CUs * 4)   (rocprofiler-sdk derived_counters.xml; the gfx94x
formula, which is what ROCm 7.2 falls back to on gfx950 -- MI355X_MICROARCH.md, "rocprofv3 PMC slots").
On this part GRBM_GUI_ACTIVE comes back SUMMED over the 8 XCDs while CU_NUM already counts all 256 CUs, so the stock formula
reads 8x low: conv_taph_kernel shows 8.0 % where its executed FLOPs / measured time / 157.3 TF/s give 63 %.  The second column
applies that factor (it calibrates on conv_taph_kernel and then agrees with the FLOP-derived figures of the other kernels).
usage: python tools/pmc_mfma.py <counter_collection.csv> [CUs=256]"""
import collections, csv, sys
cus = int(sys.argv[2]) if len(sys.argv) > 2 else 256
per = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    per[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
rows = []
for k, c in per.items():
    if "SQ_VALU_MFMA_BUSY_CYCLES" not in c or "GRBM_GUI_ACTIVE" not in c:
        continue
    busy, act = sum(c["SQ_VALU_MFMA_BUSY_CYCLES"]), sum(c["GRBM_GUI_ACTIVE"])
    if busy <= 0 or act <= 0:
        continue
    rows.append((act, 100.0 * busy / (act * cus * 4), len(c["GRBM_GUI_ACTIVE"]), k))
rows.sort(reverse=True)
print(f"{'GUI-active cycles':>18s} {'MfmaUtil % (stock)':>19s} {'x8 (per-XCD sum)':>17s} {'launches':>9s}  kernel")
for act, util, n, k in rows[:30]:
    print(f"{act:18.0f} {util:19.1f} {8 * util:17.1f} {n:9d}  {k[:110]}")
