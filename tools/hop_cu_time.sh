#!/bin/bash
# CU-time budget of one streaming hop (configs[3]: 1024 streams, hop 320) — per kernel of the replayed graph:
#   wall duration inside the graph (kernel trace of `bench.py --mode streaming --graph`), workgroups, mean workgroup lifetime,
#   workgroups x lifetime / 256 CUs (= the CU-time the kernel occupies), MFMA-busy share, tiles per workgroup (stage kernels).
# The SQ counters come from separate --pmc passes over the EAGER hop loop (same launches as the captured graph; counter
# collection serialises kernels, so the side branch's kernels are counted alone — their in-graph wall time is the trace's).
#   bash tools/hop_cu_time.sh <tag> [extra bench.py flags]
TAG=${1:-hop_cu_time}; shift
R=$(pwd); O=$R/gpurun_out/$TAG; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
COMMON="--mode streaming --no-cpu-baseline --no-clock-probe --no-launch-timing --no-other-configs"
rocprofv3 --kernel-trace --output-format csv -d $O/t -o st -- python $R/bench.py $COMMON --graph --steps 20 --warmup 4 "$@" > $O/bench_graph.json 2> $O/err_t.txt
cp "$(find $O/t -name '*kernel_trace.csv' | head -1)" $O/trace.csv 2>/dev/null
rm -rf $O/t
: > $O/raw.csv
i=0
for CNT in "SQ_WAVE_CYCLES SQ_WAVES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA GRBM_GUI_ACTIVE" \
           "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAIT_INST_LDS"; do
  i=$((i+1))
  rocprofv3 --pmc $CNT --output-format csv -d $O/p$i -o p$i -- python $R/bench.py $COMMON --steps 4 --warmup 2 "$@" > $O/p$i.log 2>&1
  f=$(find $O/p$i -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && cat "$f" >> $O/raw.csv
  rm -rf $O/p$i
done
cd $R
python tools/hop_cu_time.py $O/trace.csv $O/raw.csv $O/bench_graph.json | tee $O/hop_cu_time.txt
