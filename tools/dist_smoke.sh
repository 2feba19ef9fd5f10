#!/bin/bash
# Functional check of bench.py's N > 1 path on a ONE-GPU box: two ranks share device 0 and talk over gloo (the real
# runs use one GPU per rank and RCCL).  Exercises: rendezvous, sharded frames, the P+4-float all-reduce between the
# reduction and the optimiser step, barrier + max-over-ranks timing, rank-0 JSON.
export ODPD_BENCH_BACKEND=gloo ODPD_BENCH_SINGLE_DEVICE=1
python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 \
    bench.py --gpus 2 --steps 5 --warmup 2 --batch 16384 "$@"
