#!/bin/bash
# Run ON THE GPU BOX: the whole GPU parity suite, then the default bench line (what the driver runs at round end)
ROOT=${GRAFT_REPO_ROOT:-$PWD}; TAG=${1:-r04_full}; OUT=$ROOT/gpurun_out/$TAG; mkdir -p $OUT; cd $ROOT
timeout -k 10 1100 python3 -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; rc=$?; tail -4 $OUT/pytest.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 400 python3 bench.py --layers-json $OUT/layers.json > $OUT/bench.log 2>$OUT/bench.err; rc=$?; echo "bench rc=$rc"; tail -1 $OUT/bench.log > $OUT/bench.json
python3 - <<PY
import json
d=json.load(open("$OUT/bench.json"))
print({k:d[k] for k in ("value","ms_per_step","max_abs_logit_err_vs_cpu_fp32","device_ms_per_step") if k in d})
print(d.get("classes")); print(d.get("kernels")); print(d.get("conv3x3x3")); print(d.get("latency_ms_by_batch")); print(d.get("parity_mode"))
PY
exit $rc
