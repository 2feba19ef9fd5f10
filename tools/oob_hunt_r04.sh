#!/bin/bash
# r04 kernels under the out-of-bounds hunt (GPU box): lane-per-unit / two-layer kernels (`wide`) and every `--quant` model with kernels.
# usage: bash tools/oob_hunt_r04.sh [cases]      -> gpurun_out/oob_hunt_r04.txt
N=${1:-20}
OUT=gpurun_out/oob_hunt_r04.txt
mkdir -p gpurun_out; : > $OUT
export PYTORCH_NO_CUDA_MEMORY_CACHING=1 PYTHONPATH=.
for bb in gru dgru qgru qgru_amp1 lstm vdlstm deltagru deltagru_tcnskip deltajanet pgjanet; do
    timeout 600 python tools/oob_hunt.py $bb 7 $N wide > /tmp/oob_$bb.log 2>&1; rc=$?
    echo "wide  $bb: rc=$rc last: $(tail -1 /tmp/oob_$bb.log)" >> $OUT
done
for bb in gru dgru qgru deltagru_tcnskip lstm vdlstm deltajanet neuraltx rvtdcnn pgjanet; do
    timeout 600 python tools/oob_hunt.py $bb 11 $N > /tmp/oobq_$bb.log 2>&1; rc=$?
    echo "quant $bb: rc=$rc last: $(tail -1 /tmp/oobq_$bb.log)" >> $OUT
done
cat $OUT
