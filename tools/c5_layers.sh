#!/bin/bash
# per-layer forward table of configs[4] with the dense 3x3 layers on the fp32 pipe and on the x3 kernel
for v in "JN_NO_CONV3_X3=1" "JN_C3X_VAR=1"; do
  env $v JN_LAYER_PROFILE=1 python3 bench.py --config c5 --mode rollout --steps 1 --warmup 1 --no-cpu-baseline 2> gpurun_out/c5_layer_raw.txt > /dev/null
  awk '/^# layer profile/{buf=""} {buf=buf $0 "\n"} /^# total/{last=buf} END{printf "%s", last}' gpurun_out/c5_layer_raw.txt > gpurun_out/c5_layers_$(echo $v | tr -d '=').txt
done
rm -f gpurun_out/c5_layer_raw.txt
