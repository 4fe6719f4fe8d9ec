#!/bin/bash
# Run ON THE GPU BOX:  bash tools/ab_layers.sh <tag> VAR "v1 v2 ..." [bench args]  - per-layer tables of the same bench with an
# environment variable swept on the same box (layers_<VAR>_<v>.json), then a side-by-side of the layers that moved
TAG=$1; VAR=$2; VALS=$3; shift 3
ROOT=${GRAFT_REPO_ROOT:-$PWD}; OUT=$ROOT/gpurun_out/$TAG; mkdir -p $OUT; cd $ROOT
for v in $VALS; do
  env $VAR=$v timeout -k 10 300 python3 bench.py --cpu-clips 0 --steps 30 --layers-json $OUT/layers_${VAR}_$v.json "$@" > $OUT/bench_${VAR}_$v.log 2>&1
  tail -1 $OUT/bench_${VAR}_$v.log | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$VAR=$v', d['value'], d['ms_per_step'], d.get('device_ms_per_step'))"
done
python3 - $OUT $VAR $VALS <<'PY'
import json, sys
out, var, vals = sys.argv[1], sys.argv[2], sys.argv[3:]
tabs = [json.load(open("%s/layers_%s_%s.json" % (out, var, v))) for v in vals]
print("%-46s" % "layer", *["%10s" % v for v in vals])
for rows in zip(*tabs):
    ms = [r["ms"] for r in rows]
    if max(ms) - min(ms) > 0.004:
        print("%-46s" % rows[0]["name"][-46:], *["%10.4f" % m for m in ms], " | ".join(r.get("kernel", "")[:26] for r in rows))
print("%-46s" % "total", *["%10.4f" % sum(r["ms"] for r in t) for t in tabs])
PY
