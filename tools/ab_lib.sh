#!/bin/bash
# Run ON THE GPU BOX:  bash tools/ab_lib.sh <tag> [bench args]  - the bench (per-layer tables) on the shipped libafhip.so ("new") and on
# libafhip_prev.so ("prev": a build of the previous kernels linked beside it), interleaved twice on the same box
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$PWD}; OUT=$ROOT/gpurun_out/$TAG; mkdir -p $OUT; cd $ROOT
PKG=$(ls -d spatiotemporal*_amd)
for rep in 1 2; do for v in prev new; do
  LIB=$ROOT/$PKG/libafhip.so; [ $v = prev ] && LIB=$ROOT/$PKG/libafhip_prev.so
  AF_HIP_LIB=$LIB timeout -k 10 300 python3 bench.py --cpu-clips 0 --steps 30 --layers-json $OUT/layers_${v}_$rep.json "$@" > $OUT/bench_${v}_$rep.log 2>&1
  tail -1 $OUT/bench_${v}_$rep.log | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v rep $rep', d['value'], d['ms_per_step'], d.get('device_ms_per_step'), d.get('max_abs_logit_err_vs_cpu_fp32'))"
done; done
python3 - $OUT <<'PY'
import json, sys
out = sys.argv[1]
tabs = {v: [json.load(open("%s/layers_%s_%d.json" % (out, v, r))) for r in (1, 2)] for v in ("prev", "new")}
print("%-46s %9s %9s" % ("layer", "prev", "new"))
for i in range(len(tabs["prev"][0])):
    p = min(t[i]["ms"] for t in tabs["prev"]); n = min(t[i]["ms"] for t in tabs["new"])
    if abs(p - n) > 0.004:
        print("%-46s %9.4f %9.4f  %s" % (tabs["new"][0][i]["name"][-46:], p, n, tabs["new"][0][i].get("kernel", "")[:30]))
print("%-46s %9.4f %9.4f" % ("total (min of reps per layer)", sum(min(t[i]["ms"] for t in tabs["prev"]) for i in range(len(tabs["prev"][0]))),
                              sum(min(t[i]["ms"] for t in tabs["new"]) for i in range(len(tabs["new"][0])))))
PY
