#!/bin/bash
# offline chain / stage kernels against the run share of the first dispatch class (two workgroups per CU: C = 96 chain, C = 64 stage)
# needs the -DHILC_RES_SHARE_ENV build:  python -c "import __graft_entry__ as g, os; g.compile_library(os.path.join(g.LIBDIR,'libhilcodec_amd_share.so'), defines=('HILC_RES_SHARE_ENV',), only=('resblock.hip','resblock_chain.hip'))"
L=$PWD/hilcodec_amd/lib/libhilcodec_amd_share.so
for s in ${SHARES:-0.50 0.56 0.60 0.64 0.68 0.72}; do
  HILC_LIB=$L HILC_SHARE2_0=$s python tools/layer_profile.py --reps 3 2>/dev/null | python -c "
import sys
out = []
for l in sys.stdin:
    if 'resblock' in l or l.startswith('total'):
        f = l.split()
        out.append((f[2] + ' ' + [x for x in f if x.replace('.', '').isdigit() and '.' in x][0]) if 'resblock' in l else l.strip())
print('share2 $s:', ' | '.join(out))"
done
