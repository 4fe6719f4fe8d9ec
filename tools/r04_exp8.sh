#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$PWD}; OUT=$ROOT/gpurun_out/r04i; mkdir -p $OUT; cd $ROOT
timeout -k 10 600 python3 -m pytest tests/test_hip_layers.py -m gpu -x -q -k "t311 or halo or fused_b_c" > $OUT/pytest.log 2>&1; rc=$?; tail -3 $OUT/pytest.log
[ $rc -eq 0 ] || exit $rc
bash tools/ab_layers.sh r04i_ab AF_G_PERSIST "0 1"
