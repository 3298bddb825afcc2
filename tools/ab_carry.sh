bash tools/ab_libs.sh libhilcodec_amd_prev.so libhilcodec_amd.so ab_carry 2>&1 | head -60
for L in libhilcodec_amd_prev.so libhilcodec_amd.so; do for A in "" "--groups 2" "--pipeline"; do HILC_LIB=$PWD/hilcodec_amd/lib/$L python bench.py --mode streaming --graph --no-cpu-baseline --no-clock-probe --no-other-configs $A 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('$L $A', round(d['ms_per_step'],3), d['index_checksum'])"; done; done
