#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$PWD}; OUT=$ROOT/gpurun_out/r04f; mkdir -p $OUT; cd $ROOT
PKG=$(ls -d spatiotemporal*_amd)
timeout -k 10 900 python3 -m pytest tests/test_hip_layers.py -m gpu -x -q > $OUT/pytest.log 2>&1; rc=$?; tail -3 $OUT/pytest.log
[ $rc -eq 0 ] || exit $rc
AF_HIP_LIB=$ROOT/$PKG/libafhip_stamps.so timeout -k 10 400 python3 tools/exp_stamps_igemm.py > $OUT/stamps_igemm.log 2>&1; cat $OUT/stamps_igemm.log
bash tools/ab_lib.sh r04f_ab
