#!/bin/bash
# 1 / 2 / 4 / 8-GPU lines of the bench on ONE node (weak scaling: 256 clips per GPU, clip-sharded, weights replicated,
# the only collective is the all_gather of per-rank counters after the timed region).  At 8 GPUs bench.py defaults to
# BASELINE configs[4] (hil_music, 2048 clips); pass MODEL=hil_speech / hil_music to pin one model for the whole curve.
# Usage (from the repo root, on a multi-GPU box):  bash tools/scale_run.sh [tag]      -> gpurun_out/<tag>/scale_N.json
TAG=${1:-scale}
O=gpurun_out/$TAG
mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
NGPU=$(python -c "import torch; print(torch.cuda.device_count())")
for N in 1 2 4 8; do
  [ "$N" -gt "$NGPU" ] && break
  ARGS="--gpus $N --steps 10 --warmup 3 --no-cpu-baseline"
  [ -n "${MODEL:-}" ] && ARGS="$ARGS --model $MODEL"
  if [ "$N" -eq 1 ]; then
    python bench.py $ARGS > $O/scale_$N.json 2> $O/scale_$N.err
  else
    python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port $((29500 + N)) \
      bench.py $ARGS > $O/scale_$N.json 2> $O/scale_$N.err
  fi
  python - "$O/scale_$N.json" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
r = d["ranks"]
print(f"N={d['n_gpus']}: {d['value']:.0f} xRT  {d['ms_per_step']:.2f} ms/step  skew {r['wall_skew_s'] * 1e3:.2f} ms  rccl_ranks {r['rccl_ranks']}  {d['config']['workload']}")
PY
done
