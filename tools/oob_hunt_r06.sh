#!/bin/bash
# r06 kernels under the out-of-bounds hunt (GPU box): the bf16x3 TRAIN step of the GRU family (hidden 17 .. 24 drawn among 1 .. 32, the S16 kernels
# forced in half of the cases: gru16x_train_kernel, its one-step tail block, the batched staging, the float4 + float2 checkpoints of both gru16x
# kernels) and the cascades that end in the frozen kernel.      usage: bash tools/oob_hunt_r06.sh [cases]      -> gpurun_out/oob_hunt_r06.txt
N=${1:-60}
OUT=gpurun_out/oob_hunt_r06.txt
mkdir -p gpurun_out; : > $OUT
export PYTORCH_NO_CUDA_MEMORY_CACHING=1 PYTHONPATH=.
for bb in gru dgru qgru qgru_amp1; do
    timeout 900 python tools/oob_hunt.py $bb 6 $N > /tmp/oob6_$bb.log 2>&1; rc=$?
    echo "$bb: rc=$rc last: $(tail -1 /tmp/oob6_$bb.log)" >> $OUT
done
cat $OUT
