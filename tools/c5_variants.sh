#!/bin/bash
# A/B of the dense-3x3 kernels on configs[4] (run on the GPU box): rollout (forward only) and training iteration with the
# dense 3x3 layers on the fp32 matrix pipe, with the stride-1 layers on the three-way split kernel, and with stride 2 too
for v in "JN_NO_CONV3_X3=1" "JN_NO_CONV3_X3S2=1" "JN_DUMMY=0"; do
  for mode in rollout train; do
    r=$(env $v python3 bench.py --config c5 --mode $mode --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | grep -o '"ms_per_step": [0-9.]*')
    echo "$v $mode $r"
  done
done
