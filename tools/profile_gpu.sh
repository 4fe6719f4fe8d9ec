#!/bin/bash
# Run ON THE GPU BOX (through gpurun) from the repo root:  bash tools/profile_gpu.sh <tag> [dtype] [extra bench args, e.g. --model slowfast]
# Produces, under gpurun_out/<tag>/: the bench JSON line, per-layer table, rocprofv3 kernel-trace stats of the
# very same bench command, and separate PMC passes (FETCH_SIZE, WRITE_SIZE, MFMA busy) as MI355X_MICROARCH.md prescribes
# (TCC slots cannot hold both; never combined with other trace domains).
set -u
TAG=${1:-prof}; DT=${2:-bf16}; shift; shift
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $ROOT
timeout -k 10 500 python3 bench.py --dtype $DT --layers-json $OUT/layers.json "$@" > $OUT/bench.log 2>&1; echo "bench rc=$?"
tail -1 $OUT/bench.log > $OUT/bench.json
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $ROOT/bench.py --dtype $DT --cpu-clips 0 --no-roofline "$@" > $OUT/trace.log 2>&1; echo "trace rc=$?"
timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 $ROOT/bench.py --dtype $DT --cpu-clips 0 --steps 2 --warmup 1 --no-roofline "$@" > $OUT/pmc_fetch.log 2>&1; echo "fetch rc=$?"
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_mfma -- python3 $ROOT/bench.py --dtype $DT --cpu-clips 0 --steps 2 --warmup 1 --no-roofline "$@" > $OUT/pmc_mfma.log 2>&1; echo "mfma rc=$?"
timeout -k 10 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 $ROOT/bench.py --dtype $DT --cpu-clips 0 --steps 2 --warmup 1 --no-roofline "$@" > $OUT/pmc_write.log 2>&1; echo "write rc=$?"
# keep what the summariser needs small enough to travel back (the raw traces of a 20-step run are tens of MB)
find $OUT -name "*kernel_trace.csv" -path "*trace/*" -size +8M -delete 2>/dev/null
