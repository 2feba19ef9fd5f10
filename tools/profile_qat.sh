#!/bin/bash
# rocprofv3 kernel stats + PMC passes (separate runs, as the MI355X guide prescribes) for the quantisation-aware kernels at bench size
# (run on the GPU box).  usage: tools/profile_qat.sh <outdir-under-gpurun_out> [qat_bench args...]
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1; shift
mkdir -p $OUT
export PYTHONPATH=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
CMD="python3 $GRAFT_REPO_ROOT/tools/qat_bench.py --mappings s16 --B 32768 $*"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- $CMD > $OUT/stats.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVES --kernel-trace --output-format csv -d $OUT/pmc_sq1 -- $CMD > $OUT/pmc_sq1.log 2>&1
rocprofv3 --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES --kernel-trace --output-format csv -d $OUT/pmc_sq2 -- $CMD > $OUT/pmc_sq2.log 2>&1
rocprofv3 --pmc FETCH_SIZE GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/pmc_fetch -- $CMD > $OUT/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -- $CMD > $OUT/pmc_write.log 2>&1
cp $(find $OUT/stats -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats.csv
head -12 $OUT/kernel_stats.csv
