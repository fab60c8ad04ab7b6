#!/bin/bash
# Serialised (JN_NO_AUX_STREAM=1) kernel-trace profile of the configs[4] training iteration -> gpurun_out/
export TMPDIR=/tmp JN_NO_AUX_STREAM=1
rm -rf gpurun_out/prof_c5
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_c5 -- python3 bench.py --config c5 --mode train --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/c5prof.log 2>&1
find gpurun_out/prof_c5 -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} gpurun_out/r03_w_c5_train_kernel_stats_serial.csv
rm -rf gpurun_out/prof_c5
