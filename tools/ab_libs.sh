A=$1; B=$2; TAG=$3
mkdir -p gpurun_out/$TAG
for i in 1 2; do
 for L in $A $B; do
  HILC_LIB=$PWD/hilcodec_amd/lib/$L python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-other-configs --no-clock-probe 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('$L', round(d['ms_per_step'],3), d['index_checksum'])"
 done
done
for L in $A $B; do HILC_LIB=$PWD/hilcodec_amd/lib/$L python tools/layer_profile.py > gpurun_out/$TAG/layers_$L.txt 2>&1; done
paste <(cut -c1-60 gpurun_out/$TAG/layers_$A.txt) <(cut -c38-60 gpurun_out/$TAG/layers_$B.txt) | tail -52
