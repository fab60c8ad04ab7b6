#!/bin/bash
# Per-op tables of one round (run on the GPU box from the repo root): forward layer table (JN_LAYER_PROFILE) and backward
# per-op table (JN_BWD_PROFILE, single stream) of the headline iteration.  usage: tools/profile_tables.sh <tag>
TAG=${1:-r03}
OUT=$PWD/gpurun_out
JN_LAYER_PROFILE=1 timeout -k 10 300 python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline 2> $OUT/${TAG}_layer_raw.txt > /dev/null
# the last complete table of the run (one per glimpse step is printed)
awk '/^# layer profile/{buf=""} {buf=buf $0 "\n"} /^# total/{last=buf} END{printf "%s", last}' $OUT/${TAG}_layer_raw.txt > $OUT/${TAG}_layer_table_train_f32.txt
rm -f $OUT/${TAG}_layer_raw.txt
JN_BWD_PROFILE=1 JN_NO_AUX_STREAM=1 timeout -k 10 300 python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline 2> $OUT/${TAG}_bwd_raw.txt > /dev/null
awk '/^# backward profile/{buf=""} {buf=buf $0 "\n"} /^# total/{last=buf} END{printf "%s", last}' $OUT/${TAG}_bwd_raw.txt > $OUT/${TAG}_backward_table_f32.txt
rm -f $OUT/${TAG}_bwd_raw.txt
tail -2 $OUT/${TAG}_layer_table_train_f32.txt $OUT/${TAG}_backward_table_f32.txt
