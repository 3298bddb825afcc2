#!/bin/bash
# 1 / 2 / 4 / 8-GPU lines of the bench on ONE node, BOTH curves from one command (weak scaling, nothing on the data path
# crosses GPUs; the only collective is the all_gather of per-rank counters after the timed region):
#   offline    256 clips per GPU, clip-sharded, weights replicated.  At 8 GPUs bench.py defaults to BASELINE configs[4]
#              (hil_music, 2048 clips); pass MODEL=hil_speech / hil_music to pin one model for the whole curve.
#   streaming  BASELINE configs[3] per GPU: 1024 concurrent streams pinned to their rank for life (the 22 + 30 caches of a
#              stream never leave that GPU's HBM), one HIP-graph replay per hop (--mode streaming --graph); rank r owns streams
#              [1024 r, 1024 (r + 1)) of the global stream set.  PIPELINE=1 adds --pipeline.
# Usage (from the repo root, on a multi-GPU box):  bash tools/scale_run.sh [tag]      -> gpurun_out/<tag>/{offline,streaming}_N.json
#        LEGS="offline" or LEGS="streaming" runs one curve only.
TAG=${1:-scale}
O=gpurun_out/$TAG
mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
NGPU=$(python -c "import torch; print(torch.cuda.device_count())")
for LEG in ${LEGS:-offline streaming}; do
  for N in 1 2 4 8; do
    [ "$N" -gt "$NGPU" ] && break
    if [ "$LEG" = offline ]; then
      ARGS="--gpus $N --steps 10 --warmup 3 --no-cpu-baseline --no-other-configs"
      [ -n "${MODEL:-}" ] && ARGS="$ARGS --model $MODEL"
    else
      ARGS="--gpus $N --mode streaming --graph --steps 150 --warmup 10 --no-cpu-baseline"
      [ -n "${PIPELINE:-}" ] && ARGS="$ARGS --pipeline"
    fi
    if [ "$N" -eq 1 ]; then
      python bench.py $ARGS > $O/${LEG}_$N.json 2> $O/${LEG}_$N.err
    else
      python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port $((29500 + N)) \
        bench.py $ARGS > $O/${LEG}_$N.json 2> $O/${LEG}_$N.err
    fi
    python - "$O/${LEG}_$N.json" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
r = d["ranks"]
sus = d["roofline"].get("sustained_per_rank") or ([d["roofline"]["sustained"]] if d["roofline"].get("sustained") else [])
clk = " ".join(f"{int(x.get('sclk_mhz') or 0)}MHz/{int(x.get('power_w') or 0)}W" for x in sus)      # every rank's device while all ranks keep stepping
print(f"N={d['n_gpus']}: {d['value']:.0f} xRT  {d['ms_per_step']:.3f} ms/step  skew {r['wall_skew_s'] * 1e3:.2f} ms  rccl_ranks {r['rccl_ranks']}  [{clk}]  {d['config']['workload']}")
PY
  done
done
