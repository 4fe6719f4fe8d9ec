#!/bin/bash
# Run ON THE GPU BOX:  bash tools/ab_env.sh <tag> VAR "v1 v2 ..." [bench args]  - the same bench with an environment variable swept, same box
TAG=$1; VAR=$2; VALS=$3; shift 3
ROOT=${GRAFT_REPO_ROOT:-$PWD}; OUT=$ROOT/gpurun_out/$TAG; mkdir -p $OUT; cd $ROOT
for rep in 1 2; do for v in $VALS; do
  env $VAR=$v timeout -k 10 300 python3 bench.py --cpu-clips 0 --steps 50 "$@" > $OUT/${VAR}_${v}_$rep.log 2>&1
  tail -1 $OUT/${VAR}_${v}_$rep.log | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$VAR=$v rep $rep', d['value'], d['ms_per_step'], d.get('device_ms_per_step'))"
done; done
