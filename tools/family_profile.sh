#!/bin/bash
# rocprofv3 per-kernel stats over tools/family_table.py (run on the GPU box): writes gpurun_out/family/{table.md,kernel_stats.csv}
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/family
mkdir -p $OUT
export PYTHONPATH=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
python3 $GRAFT_REPO_ROOT/tools/family_table.py --out $OUT/table.md > $OUT/table.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -- python3 $GRAFT_REPO_ROOT/tools/family_table.py > $OUT/prof.log 2>&1
cp $(find $OUT/prof -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats.csv
tail -3 $OUT/table.log
