#!/bin/bash
# rocprofv3 kernel stats + one PMC pass (separate runs) of the train_dpd step at the reference's batch sizes (tools/cascade_spans.py:
# config 3 at 64 x 200 on the chained one-sequence-per-wave launches; GRU15 -> GRU23 and DGRU13 -> DGRU13 at 256 x 200 on the one-launch
# cascade kernel, csrc/gru_cascade.hip).  usage (GPU box): tools/profile_cascade_ref.sh <outdir-under-gpurun_out>
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1; shift
mkdir -p $OUT
export PYTHONPATH=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
CMD="python3 $GRAFT_REPO_ROOT/tools/cascade_spans.py"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- $CMD > $OUT/stats.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_LDS SQ_VALU_MFMA_BUSY_CYCLES --kernel-trace --output-format csv -d $OUT/pmc_sq -- $CMD > $OUT/pmc_sq.log 2>&1
cp $(find $OUT/stats -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats.csv
head -14 $OUT/kernel_stats.csv
