#!/bin/bash
# stand-alone fused-block times against the run shares of the dispatch classes (-DHILC_RES_SHARE_ENV build)
L=$PWD/hilcodec_amd/lib/libhilcodec_amd_share.so
for S in "0.5 0.3333 0.3333" "0.62 0.44 0.31" "0.65 0.46 0.30" "0.68 0.48 0.30" "0.71 0.50 0.29" "0.74 0.52 0.28" "0.65 0.42 0.32" "0.5 0.3333 0.3333"; do
  set -- $S
  echo "shares $S: $(HILC_LIB=$L HILC_SHARE2_0=$1 HILC_SHARE3_0=$2 HILC_SHARE3_1=$3 python tools/res_bench.py --reps 9 2>/dev/null | awk '{printf "C%s %s | ", $2, $6}')"
done
