#!/bin/bash
# tuning aid: run shares of the dispatch classes of the offline fused block (-DHILC_RES_SHARE_ENV build), same box
L=$PWD/hilcodec_amd/lib/libhilcodec_amd_share.so
for S in "0.5 0.3333 0.3333" "0.55 0.37 0.33" "0.58 0.39 0.33" "0.60 0.41 0.33" "0.62 0.43 0.32" "0.5 0.3333 0.3333"; do
  set -- $S
  HILC_LIB=$L HILC_SHARE2_0=$1 HILC_SHARE3_0=$2 HILC_SHARE3_1=$3 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-other-configs --no-clock-probe 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('shares $S', round(d['ms_per_step'],3), d['index_checksum'])"
done
