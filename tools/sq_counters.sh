#!/bin/bash
# SQ wait / issue / LDS-conflict counters per kernel over the serial fp32 step (separate --pmc passes, no tracing):
# usage (inside gpurun) bash tools/sq_counters.sh <tag>  ->  gpurun_out/<tag>/sq_counters.txt
set -u
tag=${1:-r5z}
out=gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
i=0
for set in "SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY" "SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU"; do
  i=$((i+1))
  SSBEV_WGRAD_STREAM=0 SSBEV_VT_STREAMS=0 timeout 900 rocprofv3 --pmc $set --output-format csv -d /tmp/sq_${tag}_$i -o p -- python bench.py --steps 2 --warmup 1 --cpu-sample none --skip-forward-extra --skip-serial-replay > /dev/null 2>&1 || echo "pass $i failed: $set"
done
python - $tag > $out/sq_counters.txt <<'PY'
import csv, glob, collections, sys
tag = sys.argv[1]
res = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob(f"/tmp/sq_{tag}_*/**/*counter_collection.csv", recursive=True)):
    for r in csv.DictReader(open(f)):
        res[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
rows = []
for k, d in res.items():
    g = lambda c: sum(d.get(c, [0.0]))
    wc = g("SQ_WAVE_CYCLES")
    if wc <= 0:
        continue
    n = len(d.get("SQ_WAVE_CYCLES", []))
    lds = g("SQ_LDS_IDX_ACTIVE")
    rows.append((wc, k, n, 100 * g("SQ_ACTIVE_INST_ANY") / wc, 100 * g("SQ_WAIT_ANY") / wc, 100 * g("SQ_WAIT_INST_ANY") / wc,
                 100 * g("SQ_ACTIVE_INST_VALU") / wc, 100 * g("SQ_ACTIVE_INST_LDS") / wc, g("SQ_VALU_MFMA_BUSY_CYCLES") / wc,
                 100 * g("SQ_LDS_BANK_CONFLICT") / lds if lds else 0.0))
rows.sort(reverse=True)
print("per kernel over 3 serial steps of the fp32 bench (quad-cycle units, summed over waves): share of wave cycles issuing any instruction /\n"
      "parked at s_waitcnt or a barrier / waiting to issue (pipe busy, dependency) / issuing VALU / issuing LDS; MFMA-busy cycles per wave\n"
      "cycle; LDS bank-conflict cycles per LDS-active cycle")
print(f"{'kernel':60s} {'launch':>6s} {'wave-cyc(G)':>11s} {'active%':>8s} {'wait_any%':>9s} {'wait_inst%':>10s} {'valu%':>6s} {'lds%':>5s} {'mfma/wc':>8s} {'lds_conf%':>9s}")
for wc, k, n, a, w, wi, va, ld, mf, cf in rows[:40]:
    print(f"{k[:60]:60s} {n:6d} {wc / 1e9:11.2f} {a:8.1f} {w:9.1f} {wi:10.1f} {va:6.1f} {ld:5.1f} {mf:8.3f} {cf:9.1f}")
PY
head -30 $out/sq_counters.txt | cut -c1-170
