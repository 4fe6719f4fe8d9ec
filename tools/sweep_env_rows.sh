#!/bin/bash
# Run ON THE GPU BOX:  bash tools/sweep_env_rows.sh <tag> VAR "v1 v2 ..." <row-name substring> [bench args]
# the bench with an environment variable swept; prints the value / ms per step and the per-layer rows whose name contains the substring
TAG=$1; VAR=$2; VALS=$3; SUB=$4; shift 4
ROOT=${GRAFT_REPO_ROOT:-$PWD}; OUT=$ROOT/gpurun_out/$TAG; mkdir -p $OUT; cd $ROOT
for v in $VALS; do
  env $VAR=$v timeout -k 10 300 python3 bench.py --cpu-clips 0 --steps 30 --layers-json $OUT/layers_${VAR}_$v.json "$@" > $OUT/bench_${VAR}_$v.log 2>&1
  tail -1 $OUT/bench_${VAR}_$v.log | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$VAR=$v', d['value'], d['ms_per_step'], d.get('device_ms_per_step'))"
  python3 - $OUT/layers_${VAR}_$v.json "$SUB" <<'PY'
import json, sys
for r in json.load(open(sys.argv[1])):
    if sys.argv[2] in r["name"]:
        print("   %-50s %.4f ms" % (r["name"][-50:], r["ms"]))
PY
done
