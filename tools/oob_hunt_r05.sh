#!/bin/bash
# r05 kernels under the out-of-bounds hunt (GPU box): the bf16x3 frozen-PA step (gru family as PA of a cascade, hidden 17 .. 24 drawn among 1 .. 32, the S16
# kernels forced in half of the cases) and the eight-wave delta backward + the TCN weight-gradient rows (deltagru, deltagru_tcnskip, deltajanet); the quantised pgjanet (pgjanet_q.hip); lstm / vdlstm (the K-packed fused train kernel at hidden <= 13).
# usage: bash tools/oob_hunt_r05.sh [cases]      -> gpurun_out/oob_hunt_r05.txt
N=${1:-60}
OUT=gpurun_out/oob_hunt_r05.txt
mkdir -p gpurun_out; : > $OUT
export PYTORCH_NO_CUDA_MEMORY_CACHING=1 PYTHONPATH=.
for bb in gru dgru qgru qgru_amp1 deltagru deltagru_tcnskip deltajanet lstm vdlstm; do
    timeout 900 python tools/oob_hunt.py $bb 5 $N > /tmp/oob5_$bb.log 2>&1; rc=$?
    echo "$bb: rc=$rc last: $(tail -1 /tmp/oob5_$bb.log)" >> $OUT
done
# the quantised pgjanet's rows-in-registers kernel pair (hidden 1 .. 32 drawn at random: both padded sizes)
timeout 900 python tools/oob_hunt.py pgjanet 13 $N > /tmp/oob5_pgq.log 2>&1; rc=$?
echo "quant pgjanet: rc=$rc last: $(tail -1 /tmp/oob5_pgq.log)" >> $OUT
cat $OUT
