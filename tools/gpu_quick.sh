#!/bin/bash
# Run ON THE GPU BOX (through gpurun):  bash tools/gpu_quick.sh <tag> "<pytest -k expression>" [bench args...]
# A subset of the GPU parity tests, then (only if green) one bench line with the per-layer table.
set -u
TAG=$1; KEXPR=$2; shift 2
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $ROOT
timeout -k 10 600 python3 -m pytest tests -m gpu -x -q -s -k "$KEXPR" > $OUT/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -4 $OUT/pytest.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 400 python3 bench.py --layers-json $OUT/layers.json "$@" > $OUT/bench.log 2>$OUT/bench.err; rc=$?; echo "bench rc=$rc"; tail -1 $OUT/bench.log > $OUT/bench.json
python3 - <<PY
import json
d=json.load(open("$OUT/bench.json"))
print({k:d[k] for k in ("value","ms_per_step","max_abs_logit_err_vs_cpu_fp32") if k in d})
print(d.get("classes")); print(d.get("kernels"))
PY
exit $rc
