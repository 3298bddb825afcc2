#!/bin/bash
# same-box, alternating A/B of one boolean field of engine.ExecOptions on the streaming hop (graph and pipelined):  bash tools/ab_opt_stream.sh <field> [rounds]
F=$1; N=${2:-2}
for i in $(seq $N); do
 for M in "--graph" "--graph --pipeline"; do
  for V in 0 1; do
   python bench.py --mode streaming $M --no-cpu-baseline --exec-opt $F=$V 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('$M $F=$V', round(d['ms_per_step'],4), d['index_checksum'])"
  done
 done
done
