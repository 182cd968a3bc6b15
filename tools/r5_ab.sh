# A/B on ONE box: alternate bench runs of variants; each variant = "name|dir|ENV=.. ENV=..|extra bench args".  usage: bash tools/r5_ab.sh <tag> <rounds> <variant>...
set -u
tag=$1; rounds=$2; shift 2
out=$GRAFT_REPO_ROOT/gpurun_out/$tag
mkdir -p $out
for r in $(seq 1 $rounds); do
  for v in "$@"; do
    IFS='|' read -r name dir envs extra <<< "$v"
    ( cd $GRAFT_REPO_ROOT/$dir && env $envs timeout 400 python bench.py --steps 12 --warmup 4 --cpu-sample none --skip-forward-extra --skip-serial-replay $extra 2>$out/$name.err | grep '^{"metric"' | tail -1 > $out/${name}_$r.json )
    python - $out/${name}_$r.json $name $r <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read())
    print(f"{sys.argv[2]:28s} round {sys.argv[3]}: {d['ms_per_step']:.2f} ms")
except Exception as e:
    print(sys.argv[2], "FAILED", e)
PY
  done
done
