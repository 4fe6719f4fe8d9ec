#!/bin/bash
# Run ON THE GPU BOX (through gpurun) from the repo root:  bash tools/gpu_check.sh <tag>
# GPU parity suite + the default bench line + the synthetic 3x3x3 number + a 2-rank rehearsal of the self-launching bench.
set -u
TAG=${1:-check}
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $ROOT
timeout -k 10 600 python3 -m pytest tests -m gpu -x -q -s > $OUT/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -3 $OUT/pytest.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 400 python3 bench.py --layers-json $OUT/layers.json > $OUT/bench.log 2>$OUT/bench.err; rc=$?; echo "bench rc=$rc"; tail -1 $OUT/bench.log > $OUT/bench.json
[ $rc -eq 0 ] || exit $rc
timeout -k 10 200 python3 bench.py --model conv3x3x3 > $OUT/conv3x3x3.json 2>$OUT/conv3x3x3.err; rc=$?; echo "conv3x3x3 rc=$rc"
[ $rc -eq 0 ] || exit $rc
AF_BENCH_REHEARSAL=1 timeout -k 10 300 python3 bench.py --gpus 2 --batch 8 --cpu-clips 0 > $OUT/bench_2rank.json 2>$OUT/bench_2rank.err; rc=$?; echo "2-rank rehearsal rc=$rc"; cat $OUT/bench_2rank.json | cut -c1-300
exit $rc
