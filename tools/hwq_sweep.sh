for Q in 4 8 16 2; do
 for A in "--groups 2" "--groups 3" "--groups 4" "--pipeline --groups 2" "--pipeline --groups 1"; do
  GPU_MAX_HW_QUEUES=$Q python bench.py --mode streaming --graph --no-cpu-baseline --no-clock-probe --no-other-configs $A 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('Q=$Q $A', round(d['ms_per_step'],3), d['index_checksum'])"
 done
done
