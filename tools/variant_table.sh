#!/bin/bash
# per-launch times of the fused-block launches of an offline step for several builds of the library (tools/build_variants.py), one box:
#   bash tools/variant_table.sh libhilcodec_amd.so libv_kp8d2.so ...
for L in "$@"; do
  HILC_LIB=$PWD/hilcodec_amd/lib/$L python tools/layer_profile.py --reps 3 2>/dev/null | python -c "
import sys
out = []
for l in sys.stdin:
    if 'resblock' in l or l.startswith('total'):
        f = l.split()
        out.append((f[2] + ' ' + [x for x in f if x.replace('.', '').isdigit() and '.' in x][0]) if 'resblock' in l else l.strip())
print('$L:', ' | '.join(out))"
done
