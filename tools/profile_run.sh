#!/bin/bash
# Round-end measurement on the GPU box: bench lines + rocprofv3 kernel trace + the two PMC passes
# (FETCH_SIZE / WRITE_SIZE collected separately, never together with a trace domain), condensed by
# tools/summarize_profile.py.  Usage (from the repo root, on the GPU box):  bash tools/profile_run.sh <tag>
# Everything lands in gpurun_out/<tag>/ ; copy the summaries you want to keep into profiles/.  .git does not travel to the GPU
# box: pass the commit as HILC_GIT_SHA=$(git rev-parse --short=12 HEAD) in the gpurun command (the summary is stamped with it
# and with a hash of the kernel sources; bench.py marks roofline.traffic stale when the sources have changed since).
TAG=${1:-profile}
R=$(pwd)
O=$R/gpurun_out/$TAG
mkdir -p $O
python bench.py > $O/bench.json 2> $O/bench.err
python bench.py --model hil_music --no-cpu-baseline > $O/bench_hil_music.json 2>> $O/bench.err
python bench.py --mode streaming --no-cpu-baseline > $O/bench_streaming.json 2>> $O/bench.err
python bench.py --mode streaming --graph --no-cpu-baseline > $O/bench_streaming_graph.json 2>> $O/bench.err
python bench.py --mode streaming --graph --groups 2 --no-cpu-baseline > $O/bench_streaming_graph_groups2.json 2>> $O/bench.err
python bench.py --mode streaming --graph --pipeline --no-cpu-baseline > $O/bench_streaming_pipelined.json 2>> $O/bench.err
python bench.py --mode streaming --graph --pipeline --groups 2 --no-cpu-baseline > $O/bench_streaming_pipelined_groups2.json 2>> $O/bench.err
python tools/layer_profile.py > $O/layer_table.txt 2>> $O/bench.err
python tools/layer_profile.py --mode streaming --batch 1024 > $O/layer_table_streaming.txt 2>> $O/bench.err
cd /tmp && export TMPDIR=/tmp
STEPS=4; WARM=2
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o stats -- python $R/bench.py --steps $STEPS --warmup $WARM --no-cpu-baseline --no-clock-probe --no-launch-timing --no-other-configs > $O/trace_bench.json 2> $O/trace.err
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -o fetch -- python $R/bench.py --steps $STEPS --warmup $WARM --no-cpu-baseline --no-clock-probe --no-launch-timing --no-other-configs > /dev/null 2> $O/pmc_fetch.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -o write -- python $R/bench.py --steps $STEPS --warmup $WARM --no-cpu-baseline --no-clock-probe --no-launch-timing --no-other-configs > /dev/null 2> $O/pmc_write.err
cd $R
mkdir -p $O/flat
for f in $(find $O/trace $O/pmc_fetch $O/pmc_write -name "*.csv"); do cp $f $O/flat/$(basename $f); done
ls -la $O/flat
python tools/summarize_profile.py $O/flat $O/summary $((STEPS + WARM)) > $O/summary_mfma_family.json
cat $O/summary_mfma_family.json
# keep the transfer small: the raw per-dispatch traces are not needed once condensed
rm -rf $O/trace $O/pmc_fetch $O/pmc_write
find $O/flat -size +2M -delete
