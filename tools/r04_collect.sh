#!/bin/bash
# Run ON THE GPU BOX:  bash tools/r04_collect.sh <tag> [part]
# The round's measured artefacts.  part "i3d" (default): the bf16 profile set of the headline (bench line + per-layer table +
# rocprofv3 kernel stats + PMC passes), the f16 / f32 lines, the sustained 400-step run, conv3x3x3, dualrun_rgb, stream, aligner,
# the B = 16 CPU baseline (--cpu-clips 16).  part "models": per-layer + rocprofv3 + PMC sets of SlowFast and FTCN-TT.
# part "kernels": in-kernel stamp tables of the patch-resident and the generic kernels (stamp build), the single-layer A/B.
set -u
TAG=${1:-r04}; PART=${2:-i3d}
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $ROOT
PKG=$(ls -d spatiotemporal*_amd)
if [ "$PART" = "models" ]; then
  bash tools/profile_gpu.sh ${TAG}_slowfast bf16 --model slowfast
  cd $ROOT; bash tools/profile_gpu.sh ${TAG}_ftcn_tt bf16 --model ftcn_tt
  exit 0
fi
if [ "$PART" = "kernels" ]; then
  AF_G_PERSIST=0 AF_HIP_LIB=$ROOT/$PKG/libafhip_stamps.so timeout -k 10 300 python3 tools/exp_stamps133g.py > $OUT/stamps_133g.log 2>&1; echo "stamps133g rc=$?"   # (one unit per workgroup: the stamp slots are per unit)
  AF_HIP_LIB=$ROOT/$PKG/libafhip_stamps.so timeout -k 10 400 python3 tools/exp_stamps_igemm.py > $OUT/stamps_igemm.log 2>&1; echo "stamps igemm rc=$?"
  AF_HIP_LIB=$ROOT/$PKG/libafhip_stamps.so timeout -k 10 300 python3 tools/exp_stamps_c64.py > $OUT/stamps_c64.log 2>&1; echo "stamps c64 rc=$?"
  AF_HIP_LIB=$ROOT/$PKG/libafhip_stamps.so timeout -k 10 300 python3 tools/exp_ca_dbg.py > $OUT/ca_ablations.log 2>&1; echo "ca ablations rc=$?"
  timeout -k 10 300 python3 tools/exp_ca_ab.py > $OUT/ca_cwl_ab.log 2>&1; echo "ca cwl a/b rc=$?"
  timeout -k 10 300 python3 tools/exp_b133g.py > $OUT/b133g_new.log 2>&1
  # (the A/B against the round-3 library - tools/ab_lib.sh with libafhip_prev.so - was run in the first half of the round; that library
  #  speaks ABI 3 and no longer loads beside the ABI-4 host code)
  exit 0
fi
bash tools/profile_gpu.sh $TAG bf16
cd $ROOT
for dt in f16 f32; do
  timeout -k 10 300 python3 bench.py --dtype $dt --layers-json $OUT/layers_$dt.json > $OUT/bench_$dt.log 2>&1; echo "bench $dt rc=$?"; tail -1 $OUT/bench_$dt.log > $OUT/bench_$dt.json
done
timeout -k 10 300 python3 bench.py --steps 400 --warmup 10 --cpu-clips 0 --no-roofline > $OUT/bench_sustained.log 2>&1; echo "sustained rc=$?"; tail -1 $OUT/bench_sustained.log > $OUT/bench_sustained.json
timeout -k 10 500 python3 bench.py --cpu-clips 16 --no-roofline --no-parity-mode > $OUT/bench_cpu16.log 2>&1; echo "cpu16 rc=$?"; tail -1 $OUT/bench_cpu16.log > $OUT/bench_cpu16.json
for m in conv3x3x3 dualrun_rgb dualrun aligner; do
  timeout -k 10 300 python3 bench.py --model $m > $OUT/$m.log 2>&1; echo "$m rc=$?"; tail -1 $OUT/$m.log > $OUT/$m.json
done
timeout -k 10 300 python3 bench.py --model stream --steps 20 > $OUT/stream.log 2>&1; echo "stream rc=$?"; tail -1 $OUT/stream.log > $OUT/stream.json
for b in 64; do
  timeout -k 10 200 python3 bench.py --batch $b --cpu-clips 0 --no-roofline --steps 50 > $OUT/lat_$b.log 2>&1; tail -1 $OUT/lat_$b.log > $OUT/lat_$b.json
done
echo done
