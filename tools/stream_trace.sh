#!/bin/bash
# kernel trace of the streaming hop graph (one chain): per-kernel durations inside the replayed graph and the gaps between them
TAG=${1:-stream_trace}; R=$(pwd); O=$R/gpurun_out/$TAG; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $O/t -o st -- python $R/bench.py --mode streaming --graph $EXTRA --steps 20 --warmup 4 --no-cpu-baseline --no-clock-probe --no-launch-timing --no-other-configs > $O/bench.json 2> $O/err.txt
cd $R
f=$(find $O/t -name "*kernel_trace.csv" | head -1)
python - "$f" <<'PY' | tee $O/summary.txt
import csv, sys, collections
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(sys.argv[1]))]
rows.sort()
# the last 20 hops: find the hop period from the repeating first kernel of a hop (spec_block N=64)
starts = [i for i, r in enumerate(rows) if "spec_block_kernel<64" in r[2]]
starts = starts[-20:]
per = collections.defaultdict(lambda: [0, 0.0])
tot_k = tot_gap = 0.0; n = 0
for a, b in zip(starts[:-1], starts[1:]):
    hop = rows[a:b]
    n += 1
    end = hop[0][0]
    for s, e, k in hop:
        short = k.replace("hilc::", "").replace("(anonymous namespace)::", "")[:60]
        per[short][0] += 1; per[short][1] += (e - s) / 1e3
        tot_k += (e - s) / 1e3
        if s > end: tot_gap += (s - end) / 1e3
        end = max(end, e)
print(f"{n} hops: kernel time {tot_k / n:.1f} us per hop, idle gaps between kernels {tot_gap / n:.1f} us per hop, "
      f"period {(rows[starts[-1]][0] - rows[starts[0]][0]) / n / 1e3:.1f} us, kernels per hop {sum(v[0] for v in per.values()) / n:.1f}")
for k, v in sorted(per.items(), key=lambda kv: -kv[1][1]):
    print(f"{v[1] / n:9.1f} us/hop  {v[0] / n:5.1f} launches  {k}")
# timeline of the last complete hop: start offset, duration, gap to the previous kernel's end (negative = overlap with another branch)
a, b = starts[-2], starts[-1]
t0 = rows[a][0]
end = t0
print("\ntimeline of one hop (us):   start   dur   gap  kernel")
for s_, e_, k in rows[a:b]:
    short = k.replace("hilc::", "").replace("(anonymous namespace)::", "")[:90]
    print(f"  {(s_ - t0) / 1e3:8.1f} {(e_ - s_) / 1e3:7.1f} {(s_ - end) / 1e3:6.1f}  {short}")
    end = max(end, e_)
PY
rm -rf $O/t
