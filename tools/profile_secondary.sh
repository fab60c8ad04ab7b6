#!/bin/bash
# Secondary bench lines of a round (GPU box): detector training step, supervised iteration, rollout with detection
# (fp32 / bf16), bf16 rollout.   usage: tools/profile_secondary.sh <tag>  -> gpurun_out/<tag>_bench_*.json
TAG=${1:-r03}
OUT=$PWD/gpurun_out
run() { local name=$1; shift; timeout -k 10 300 python3 bench.py "$@" --no-cpu-baseline > $OUT/${TAG}_bench_$name.json 2>/dev/null; echo "$name $(grep -o '"ms_per_step": [0-9.]*' $OUT/${TAG}_bench_$name.json) $(grep -o '"value": [0-9.]*' $OUT/${TAG}_bench_$name.json | head -1)"; }
run detector_training --mode detector
run supervised --mode supervised
run rollout_detect --mode rollout --detect
run rollout_detect_bf16 --mode rollout --detect --dtype bf16
run rollout_bf16 --mode rollout --dtype bf16
