#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$PWD}; OUT=$ROOT/gpurun_out/r04e; mkdir -p $OUT; cd $ROOT
PKG=$(ls -d spatiotemporal*_amd)
AF_HIP_LIB=$ROOT/$PKG/libafhip_stamps.so timeout -k 10 400 python3 tools/exp_stamps_igemm.py > $OUT/stamps_igemm.log 2>&1; cat $OUT/stamps_igemm.log
