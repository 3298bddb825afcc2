#!/bin/bash
# copy the summaries of one tools/profile_run.sh directory into profiles/ under a round tag, refresh latest_mfma_family.json and the documents
#   bash tools/save_profile.sh gpurun_out/r06_a r06_a
D=$1; T=$2
for f in bench bench_hil_music bench_streaming bench_streaming_graph bench_streaming_graph_groups2 bench_streaming_pipelined bench_streaming_pipelined_groups2; do
  grep '^{' $D/$f.json | tail -1 > profiles/${T}_$f.json
done
cp $D/summary_kernel_stats.csv profiles/${T}_kernel_stats.csv
cp $D/layer_table.txt profiles/${T}_layer_table.txt
cp $D/layer_table_streaming.txt profiles/${T}_layer_table_streaming.txt
cp $D/summary_mfma_family.json profiles/${T}_mfma_family.json
cp $D/summary_mfma_family.json profiles/latest_mfma_family.json
python tools/render_docs.py $D
