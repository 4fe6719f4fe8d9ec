#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$PWD}; OUT=$ROOT/gpurun_out/r04g; mkdir -p $OUT; cd $ROOT
PKG=$(ls -d spatiotemporal*_amd)
timeout -k 10 600 python3 -m pytest tests/test_hip_layers.py -m gpu -x -q -k "t311 or halo" > $OUT/pytest.log 2>&1; rc=$?; tail -3 $OUT/pytest.log
[ $rc -eq 0 ] || exit $rc
bash tools/ab_layers.sh r04g_ab AF_T311G "0 1 0 1"
