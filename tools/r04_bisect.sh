#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$PWD}; OUT=$ROOT/gpurun_out/r04_bisect; mkdir -p $OUT; cd $ROOT
PKG=$(ls -d spatiotemporal*_amd)
for v in libafhip libafhip; do
  AF_HIP_LIB=$ROOT/$PKG/$v.so timeout -k 10 300 python3 -m pytest tests/test_hip_forward.py -m gpu -q -s -k "test_reduced_precision_logits" > $OUT/$v.log 2>&1
  echo "== $v"; grep -h "f16 \|bf16 \|passed\|failed" $OUT/$v.log | cut -c1-120
done
bash tools/r04_full.sh r04_full2
