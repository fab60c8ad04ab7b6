#!/bin/bash
# A/B of two LIBRARIES on one box (the right way to measure a kernel change: the previous binary against the new one):
#   tools/ab_lib.sh <other libjnroll.so> [bench args]   -> forward / backward section times and iteration time, two runs each
other=$1; shift
for v in "JN_DUMMY=0" "JNROLL_LIB=$other"; do
  for i in 1 2; do
    env $v python3 bench.py "$@" --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d.get('roofline',{}); b=d.get('roofline_backward',{})
print('$v'.split('/')[-2] if '/' in '$v' else 'this tree', 'fwd ms/pass', r.get('ms_per_launch'), 'bwd ms/step', b.get('ms_per_launch'), 'iter', d['ms_per_step'])"
  done
done
