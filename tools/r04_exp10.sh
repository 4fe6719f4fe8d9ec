#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$PWD}; OUT=$ROOT/gpurun_out/r04k; mkdir -p $OUT; cd $ROOT
timeout -k 10 600 python3 -m pytest tests/test_hip_layers.py -m gpu -x -q -k "c64 or 1x3x3" > $OUT/pytest.log 2>&1; rc=$?; tail -3 $OUT/pytest.log
[ $rc -eq 0 ] || exit $rc
bash tools/ab_lib.sh r04k_ab
