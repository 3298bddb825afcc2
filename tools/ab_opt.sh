#!/bin/bash
# same-box, alternating A/B of one boolean field of engine.ExecOptions on the offline headline:  bash tools/ab_opt.sh <field> [rounds]
F=$1; N=${2:-3}
for i in $(seq $N); do
 for V in 0 1; do
  python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-other-configs --no-clock-probe --exec-opt $F=$V 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('$F=$V', round(d['ms_per_step'],3), d['index_checksum'])"
 done
done
