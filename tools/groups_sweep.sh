for A in "--groups 2" "--groups 3" "--groups 4" "--pipeline" "--pipeline --groups 2" "--pipeline --groups 3"; do python bench.py --mode streaming --graph --no-cpu-baseline --no-clock-probe --no-other-configs $A 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('$A', round(d['ms_per_step'],3), d['index_checksum'])"; done
