# Run ON THE GPU BOX (through gpurun) from the repo root:  bash tools/pmc/sq_counters.sh
# Separate rocprofv3 --pmc passes (SQ wait / active / instruction-mix / LDS counters) over a short bench run;
# results under gpurun_out/sq/.  (TCP_* / TA_* passes did not finish within 5 minutes on this pool and are left out.)
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/sq
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
run() { tag=$1; shift; timeout -k 10 300 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/$tag -- python3 $ROOT/bench.py --cpu-clips 0 --steps 2 --warmup 1 --no-roofline > $OUT/$tag.log 2>&1; echo "$tag rc=$?"; }
run p1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS
run p2 SQ_INSTS_VMEM_RD SQ_INST_LEVEL_VMEM SQ_INSTS_LDS SQ_INST_LEVEL_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL
run p3 SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_DATA_FIFO_FULL
