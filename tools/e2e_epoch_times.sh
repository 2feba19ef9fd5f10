#!/bin/bash
# Steady-state epoch time of train_pa on DPA_200MHz (batch 64, frame 50: 360 steps + validation + test evaluation per epoch) per backbone,
# with the default dispatch and with the one-sequence-per-wave kernels off (ODPD_GP_MAX_BATCH=0).  usage (GPU box): bash tools/e2e_epoch_times.sh
export PYTHONPATH=${GRAFT_REPO_ROOT:-.}
for bb in "gru 11" "dgru 13" "dgru 23" "lstm 14" "vdlstm 13" "pgjanet 11" "deltagru 15" "deltagru_tcnskip 15" "tcnn 35"; do
    set -- $bb
    a=$(E2E_BACKBONE=$1 E2E_HIDDEN=$2 python tools/e2e_profile.py 10 2>&1 | grep "epochs:" | sed 's/.*= \([0-9.]*\) ms per epoch.*/\1/')
    b=$(ODPD_GP_MAX_BATCH=0 E2E_BACKBONE=$1 E2E_HIDDEN=$2 python tools/e2e_profile.py 10 2>&1 | grep "epochs:" | sed 's/.*= \([0-9.]*\) ms per epoch.*/\1/')
    printf "%-18s H%-3s  %7s ms per epoch   (one-sequence-per-wave kernels off: %7s ms)\n" $1 $2 "$a" "$b"
done
