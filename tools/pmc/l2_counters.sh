# Run ON THE GPU BOX (through gpurun) from the repo root:  bash tools/pmc/l2_counters.sh
# L2 hit / miss / request counters per kernel (two separate --pmc passes); results under gpurun_out/l2/.
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/l2
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --output-format csv -d $OUT/a -- python3 $ROOT/bench.py --cpu-clips 0 --steps 2 --warmup 1 --no-roofline > $OUT/a.log 2>&1; echo "a rc=$?"
timeout -k 10 300 rocprofv3 --kernel-trace --pmc TCP_TCC_READ_REQ_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum --output-format csv -d $OUT/b -- python3 $ROOT/bench.py --cpu-clips 0 --steps 2 --warmup 1 --no-roofline > $OUT/b.log 2>&1; echo "b rc=$?"
ls -R $OUT | head -30
