#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$PWD}; OUT=$ROOT/gpurun_out/r04j; mkdir -p $OUT; cd $ROOT
PKG=$(ls -d spatiotemporal*_amd)
timeout -k 10 600 python3 -m pytest tests/test_hip_layers.py -m gpu -x -q -k "stream111" > $OUT/pytest.log 2>&1; rc=$?; tail -3 $OUT/pytest.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python3 tools/exp_conv111.py > $OUT/new.log 2>&1; cat $OUT/new.log
AF_HIP_LIB=$ROOT/$PKG/libafhip_prev.so timeout -k 10 300 python3 tools/exp_conv111.py > $OUT/prev.log 2>&1; cat $OUT/prev.log
