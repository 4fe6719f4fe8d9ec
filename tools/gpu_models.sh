#!/bin/bash
# Run ON THE GPU BOX:  bash tools/gpu_models.sh <tag> "<pytest -k expr>"   - subset tests, then the side benches
set -u
TAG=$1; KEXPR=$2
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $ROOT
timeout -k 10 600 python3 -m pytest tests -m gpu -x -q -s -k "$KEXPR" > $OUT/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -4 $OUT/pytest.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python3 bench.py --model dualrun_rgb > $OUT/dualrun_rgb.json 2>$OUT/dualrun_rgb.err; rc=$?; echo "dualrun_rgb rc=$rc"; cut -c1-1500 $OUT/dualrun_rgb.json
[ $rc -eq 0 ] || { tail -5 $OUT/dualrun_rgb.err; exit $rc; }
timeout -k 10 200 python3 bench.py --model conv3x3x3 > $OUT/conv3x3x3.json 2>$OUT/conv3x3x3.err; rc=$?; echo "conv3x3x3 rc=$rc"; cut -c1-1800 $OUT/conv3x3x3.json
[ $rc -eq 0 ] || { tail -5 $OUT/conv3x3x3.err; exit $rc; }
AF_BENCH_REHEARSAL=1 timeout -k 10 300 python3 bench.py --gpus 2 --batch 8 --cpu-clips 0 > $OUT/bench_2rank.json 2>$OUT/bench_2rank.err; rc=$?; echo "2-rank rehearsal rc=$rc"; cut -c1-400 $OUT/bench_2rank.json
[ $rc -eq 0 ] || tail -5 $OUT/bench_2rank.err
exit $rc
