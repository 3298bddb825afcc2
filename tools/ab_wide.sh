for i in 1 2 3; do
 for V in 0 1; do
  python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-other-configs --no-clock-probe --exec-opt offline_wide_blocks=$V 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('wide=$V', round(d['ms_per_step'],3), d['index_checksum'])"
 done
done
