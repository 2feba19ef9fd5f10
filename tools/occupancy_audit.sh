#!/bin/bash
# tools/occupancy_audit.py over the family table (every HIP backbone at 256 and 32 768 x 200) and bench.py (headline, the train_dpd cascades at 65 536 x 200, cfg 4, train_pa H23) (run on the GPU box)
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/occupancy
mkdir -p $OUT
export PYTHONPATH=$GRAFT_REPO_ROOT ODPD_AUDIT_LDS=1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT/fam -- python3 $GRAFT_REPO_ROOT/tools/family_table.py > $OUT/fam.log 2>&1
python3 $GRAFT_REPO_ROOT/tools/occupancy_audit.py $(find $OUT/fam -name "*kernel_trace.csv" | head -1) $OUT/fam.log 2.0 > $OUT/family.md
rocprofv3 --kernel-trace --output-format csv -d $OUT/casc -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 > $OUT/casc.log 2>&1
python3 $GRAFT_REPO_ROOT/tools/occupancy_audit.py $(find $OUT/casc -name "*kernel_trace.csv" | head -1) $OUT/casc.log 2.0 > $OUT/cascade.md
grep -c "LDS" $OUT/family.md $OUT/cascade.md
