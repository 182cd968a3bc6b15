mkdir -p gpurun_out/r3j; cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES --kernel-trace --output-format csv -d /tmp/pmc_j -o p -- python bench.py --steps 2 --warmup 1 --cpu-sample none --skip-forward-extra --skip-serial-replay > /tmp/pmc.log 2>&1
tail -3 /tmp/pmc.log
f=$(find /tmp/pmc_j -name "*counter_collection.csv" | head -1); ls -la $f
python - "$f" <<'PY' | tee gpurun_out/r3j/sq_counters.txt
import csv, sys, collections, re
per = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    k = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"])[:58]
    per[k][r["Counter_Name"]] += float(r["Counter_Value"])
    if r["Counter_Name"] == "SQ_WAVE_CYCLES": n[k] += 1
rows = sorted(per.items(), key=lambda kv: -kv[1].get("SQ_WAVE_CYCLES", 0))
print(f"{'kernel':58s} {'launch':>6s} {'wave-cyc(G)':>11s} {'active%':>8s} {'wait_any%':>9s} {'wait_inst%':>10s} {'(lds-issue%)':>12s} {'mfma-busy/wave-cyc':>18s} {'lds_conflict%':>13s}")
for k, c in rows[:26]:
    w = c["SQ_WAVE_CYCLES"]
    if w <= 0: continue
    print(f"{k:58s} {n[k]:6d} {w/1e9:11.2f} {100*c['SQ_ACTIVE_INST_ANY']/w:8.1f} {100*c['SQ_WAIT_ANY']/w:9.1f} {100*c['SQ_WAIT_INST_ANY']/w:10.1f} {100*c['SQ_WAIT_INST_LDS']/w:12.1f} {c['SQ_VALU_MFMA_BUSY_CYCLES']/w:18.3f} {100*c['SQ_LDS_BANK_CONFLICT']/max(c['SQ_LDS_IDX_ACTIVE'],1):13.1f}")
PY
