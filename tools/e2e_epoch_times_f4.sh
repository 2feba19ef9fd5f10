#!/bin/bash
# As e2e_epoch_times.sh for the SURVEY §8-f4 recurrent backbones: steady-state train_pa epoch on DPA_200MHz (batch 64, frame 50: 360 steps + validation + test
# evaluation) with the default dispatch (one-frame-per-workgroup train and evaluation kernels) and with them off (ODPD_GP_MAX_BATCH=0: the 16-sequences-per-wave chain).
export PYTHONPATH=${GRAFT_REPO_ROOT:-.}
for bb in "deltajanet 15" "bojanet 12" "dvrjanet 12" "mcldnn 8"; do
    set -- $bb
    a=$(E2E_BACKBONE=$1 E2E_HIDDEN=$2 python tools/e2e_profile.py 10 2>&1 | grep "epochs:" | sed 's/.*= \([0-9.]*\) ms per epoch.*/\1/')
    b=$(ODPD_GP_MAX_BATCH=0 E2E_BACKBONE=$1 E2E_HIDDEN=$2 python tools/e2e_profile.py 10 2>&1 | grep "epochs:" | sed 's/.*= \([0-9.]*\) ms per epoch.*/\1/')
    printf "%-18s H%-3s  %7s ms per epoch   (one-sequence-per-wave kernels off: %7s ms)\n" $1 $2 "$a" "$b"
done
