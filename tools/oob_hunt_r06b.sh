#!/bin/bash
# r06, second half: the kernels whose staging changed (delta S16 forward / backward without halo, the TCN skip parked from global taps, janet16 forward
# on 16-step chunks, the quantised forward kernels on 16-step chunks with three / two unit slots, lstm16 checkpoint loads, compact GRU checkpoints)
# under the out-of-bounds hunt: x, target and dy at the very end (and x at the very start) of their own allocations, caching allocator off.
# usage: bash tools/oob_hunt_r06b.sh [cases]      -> gpurun_out/oob_hunt_r06b.txt
N=${1:-60}
OUT=gpurun_out/oob_hunt_r06b.txt
mkdir -p gpurun_out; : > $OUT
export PYTORCH_NO_CUDA_MEMORY_CACHING=1 PYTHONPATH=.
for bb in deltagru_tcnskip deltagru deltajanet pgjanet lstm vdlstm gru qgru; do
    timeout 900 python tools/oob_hunt.py $bb 7 $N > /tmp/oob7_$bb.log 2>&1; rc=$?
    echo "$bb: rc=$rc last: $(tail -1 /tmp/oob7_$bb.log)" >> $OUT
done
cat $OUT
