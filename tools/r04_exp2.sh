#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$PWD}; OUT=$ROOT/gpurun_out/r04b; mkdir -p $OUT; cd $ROOT
PKG=$(ls -d spatiotemporal*_amd)
AF_HIP_LIB=$ROOT/$PKG/libafhip_stamps.so timeout -k 10 300 python3 tools/exp_stamps133g.py > $OUT/stamps.log 2>&1; cat $OUT/stamps.log
