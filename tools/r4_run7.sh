set -u
out=gpurun_out/${1:-r4h}
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_model.py -q -x -k "bf16" 2>&1 | tail -6
echo "== side-stream CU mask, fp32 B=1, ms/step (two runs each)" > $out/side_stream_cu_mask.txt
for m in "" 32 64 96 128 32s 64s 96s 128s 192s; do
  for rep in 1 2; do
  r=$(SSBEV_SIDE_CU_MASK=$m timeout 300 python bench.py --steps 10 --warmup 4 --cpu-sample none --skip-forward-extra --skip-serial-replay 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],2), round(d['roofline']['avg_launch_us'],1))")
  echo "mask='$m' : $r (ms/step, conv_taph avg us in step)" >> $out/side_stream_cu_mask.txt
  done
done
cat $out/side_stream_cu_mask.txt
timeout 2400 python -m pytest tests/test_gpu_fullsize.py -q -x -k "bf16" 2>&1 | tail -6
