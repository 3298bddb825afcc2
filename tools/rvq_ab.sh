#!/bin/bash
# RVQ encode of the offline step: two builds of the library on one box; the last line (the VALU form of large batches) needs a
# -DHILC_RVQ_ENV build:  python tools/build_variants.py rvqenv=HILC_RVQ_ENV  and  HILC_LIB=.../libv_rvqenv.so
python -m pytest tests/test_gpu_rvq.py -q 2>&1 | tail -3
for i in 1 2; do
 for L in "$@"; do
  for M in hil_speech hil_music; do
   HILC_LIB=$PWD/hilcodec_amd/lib/$L python tools/layer_profile.py --model $M 2>&1 | grep -E "rvq" | sed "s/^/$L $M /"
  done
 done
done
HILC_RVQ_VALU=1 python tools/layer_profile.py 2>&1 | grep -E "rvq" | sed "s/^/VALU form /"
