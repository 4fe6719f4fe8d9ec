#!/bin/bash
# Run ON THE GPU BOX:  bash tools/r03_collect.sh <tag> [part]
# The round's measured artefacts.  part "i3d" (default): the bf16 profile set of the headline (bench line + per-layer table +
# rocprofv3 kernel stats + PMC passes), the f16 / f32 lines, the sustained 400-step run, latencies by batch, conv3x3x3, dualrun_rgb,
# stream, aligner.  part "models": per-layer + rocprofv3 + PMC sets of SlowFast and FTCN-TT.
set -u
TAG=${1:-r03}; PART=${2:-i3d}
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $ROOT
if [ "$PART" = "models" ]; then
  bash tools/profile_gpu.sh ${TAG}_slowfast bf16 --model slowfast
  cd $ROOT; bash tools/profile_gpu.sh ${TAG}_ftcn_tt bf16 --model ftcn_tt
  exit 0
fi
bash tools/profile_gpu.sh $TAG bf16
cd $ROOT
for dt in f16 f32; do
  timeout -k 10 300 python3 bench.py --dtype $dt --layers-json $OUT/layers_$dt.json > $OUT/bench_$dt.log 2>&1; echo "bench $dt rc=$?"; tail -1 $OUT/bench_$dt.log > $OUT/bench_$dt.json
done
timeout -k 10 300 python3 bench.py --steps 400 --warmup 10 --cpu-clips 0 --no-roofline > $OUT/bench_sustained.log 2>&1; echo "sustained rc=$?"; tail -1 $OUT/bench_sustained.log > $OUT/bench_sustained.json
for m in conv3x3x3 dualrun_rgb dualrun aligner; do
  timeout -k 10 300 python3 bench.py --model $m > $OUT/$m.log 2>&1; echo "$m rc=$?"; tail -1 $OUT/$m.log > $OUT/$m.json
done
timeout -k 10 300 python3 bench.py --model stream --steps 20 > $OUT/stream.log 2>&1; echo "stream rc=$?"; tail -1 $OUT/stream.log > $OUT/stream.json
for b in 1 2 4 8 64; do
  timeout -k 10 200 python3 bench.py --batch $b --cpu-clips 0 --no-roofline --steps 50 > $OUT/lat_$b.log 2>&1; tail -1 $OUT/lat_$b.log > $OUT/lat_$b.json
done
echo done
