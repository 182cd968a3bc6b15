# PMC passes over one conv layer (fwd + dgrad + wgrad); args: Cin Cout D H W
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/pmc3; mkdir -p $out
i=0
for set in "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS" "SQ_WAIT_INST_ANY SQ_WAVE_CYCLES" "SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL" "SQ_WAIT_ANY SQ_BUSY_CYCLES" "SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA" "SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_ANY"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $out/p$i -- python tools/one_layer.py "$@" > $out/p$i.log 2>&1 || echo "set $i failed: $set"
done
find $out -name "*kernel_trace.csv" -delete
python - <<'PY'
import csv, glob, collections
res = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob("gpurun_out/pmc3/p*/*/*_counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        short = "wgrad_lds" if "wgrad_lds" in k else "wgrad_cf" if "wgrad_cf" in k else "gather" if "conv_gather" in k else None
        if short: res[short][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in res.items():
    print(k)
    for c, v in d.items(): print(f"   {c:32s} n={len(v):3d} mean={sum(v)/len(v):.4g}")
PY
