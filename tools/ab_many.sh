#!/bin/bash
# same-box A/B of several builds of the library: bash tools/ab_many.sh <lib> <lib> ...   (alternating, two rounds)
for i in 1 2; do
 for L in "$@"; do
  HILC_LIB=$PWD/hilcodec_amd/lib/$L python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-other-configs --no-clock-probe 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('$L', round(d['ms_per_step'],3), d['index_checksum'])"
 done
done
