#!/bin/bash
# tuning aid: streaming hop time against the two constants of the launch heuristic (pick_mb), -DHILC_PICK_ENV build
L=$PWD/hilcodec_amd/lib/libhilcodec_amd_pick.so
run() { # args: label, env..., -- bench args
  HILC_LIB=$L env "$1" "$2" python bench.py --mode streaming --graph --no-cpu-baseline --no-clock-probe --no-other-configs "${@:3}" 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('$1 $2 ${*:3}', round(d['ms_per_step'],3), d['index_checksum'])"
}
for G in ${GS:-1 2 4}; do
  for CUS in ${CUSS:-256 128 64}; do
    for F in ${FS:-0.5 1.0 2.0}; do
      run HILC_PICK_CUS=$CUS HILC_PICK_FIXED=$F --groups $G
    done
  done
done
