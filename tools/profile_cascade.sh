#!/bin/bash
# rocprofv3 kernel stats + PMC passes (separate runs) of the train_dpd cascade kernels at bench size: bench.py's side figures
# DGRU13 -> frozen DGRU13, config 3 (TRes-DeltaGRU15 -> frozen DGRU23), config 5 (QAT QGRU10 W8A8 -> frozen DGRU23), 65 536 x 200.
# usage (GPU box): tools/profile_cascade.sh <outdir-under-gpurun_out>
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1; shift
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp ODPD_BENCH_NO_SUSTAINED=1
CMD="python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-strong --ref-batch 0 $*"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- $CMD > $OUT/stats.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/pmc_sq -- $CMD > $OUT/pmc_sq.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -- $CMD > $OUT/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -- $CMD > $OUT/pmc_write.log 2>&1
cp $(find $OUT/stats -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats.csv
