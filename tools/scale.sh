#!/bin/bash
# The driver's scaling run, line for line: bench.py at N = 1, 2, 4, 8 GPUs of one node, one process per GPU over RCCL (N > 1 through
# torch.distributed.run), each run's JSON line appended to $OUT (default gpurun_out/scale.jsonl).  Weak scaling: 65 536 frames per GPU.
#   tools/scale.sh [N ...]            e.g. tools/scale.sh 1 2        STEPS / WARMUP / PORT override the defaults
# Every line also carries, for N > 1: which communicator carried the gradient and which candidates failed their self-test and why
# (config.collective.candidates), the bare collective in microseconds on both transports (config.collective.bare_us_per_allreduce) and
# the strong-scaling epochs at the reference's global batches 256 / 64 over both transports (strong_scaling).
set -u
cd "$(dirname "$0")/.."
OUT=${OUT:-gpurun_out/scale.jsonl}
STEPS=${STEPS:-20}; WARMUP=${WARMUP:-3}; PORT=${PORT:-29500}
mkdir -p "$(dirname "$OUT")"
# (bench.py sets HSA_ENABLE_IPC_MODE_LEGACY=0 itself before any GPU call: dmabuf IPC, needed by RCCL and the hipIpc exchange on this image)
NS=("$@"); [ ${#NS[@]} -eq 0 ] && NS=(1 2 4 8)
HAVE=$(python3 -c 'import torch; print(torch.cuda.device_count())')
for N in "${NS[@]}"; do
  if [ "$N" -gt "$HAVE" ]; then echo "[scale] skipping N=$N: $HAVE GPU(s) visible" >&2; continue; fi
  if [ "$N" -eq 1 ]; then
    python3 bench.py --gpus 1 --steps "$STEPS" --warmup "$WARMUP" | tail -1 >> "$OUT"
  else
    python3 -m torch.distributed.run --nnodes=1 --nproc-per-node "$N" --master-addr 127.0.0.1 --master-port "$PORT" \
        bench.py --gpus "$N" --steps "$STEPS" --warmup "$WARMUP" | grep '^{' | tail -1 >> "$OUT"
  fi
  python3 - "$OUT" "$N" <<'PY'
import json, sys
line = [l for l in open(sys.argv[1]) if l.startswith("{")][-1]
d = json.loads(line)
c = d["config"].get("collective") or {}
print(f"[scale] N={sys.argv[2]}: {d['value'] / 1e9:.2f} G IQ samples/s ({d['ms_per_step']:.3f} ms per step), collective {c.get('kind')}, "
      f"bare {c.get('bare_us_per_allreduce')}, candidates {c.get('candidates')}", file=sys.stderr)
PY
done
