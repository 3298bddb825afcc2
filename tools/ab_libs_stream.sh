#!/bin/bash
# same-box, alternating A/B of two builds of the library on the streaming hop (plain graph and pipelined):  bash tools/ab_libs_stream.sh <libA.so> <libB.so>
for i in 1 2; do
 for M in "--graph" "--graph --pipeline"; do
  for L in "$@"; do
   HILC_LIB=$PWD/hilcodec_amd/lib/$L python bench.py --mode streaming $M --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('$M $L', round(d['ms_per_step'],4), d['index_checksum'])"
  done
 done
done
