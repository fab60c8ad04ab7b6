#!/bin/bash
# forward conv-section time per pass (ms) and iteration time for a list of switch settings: tools/ab_fwd.sh "A=1" "B=1" ...
for v in "JN_DUMMY=0" "$@"; do
  for i in 1 2; do
    env $v python3 bench.py --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', 'fwd ms/pass', d['roofline']['ms_per_launch'], 'bwd ms/step', d['roofline_backward']['ms_per_launch'], 'iter', d['ms_per_step'])"
  done
done
