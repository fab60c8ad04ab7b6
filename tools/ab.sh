#!/bin/bash
# A/B of one switch on the GPU box: tools/ab.sh SWITCH=1 [bench args]  -> ms per step with the switch off (production) and on
sw=$1; shift
for v in "JN_DUMMY=0" "$sw"; do
  r=$(env $v python3 bench.py "$@" --no-cpu-baseline 2>/dev/null | grep -o '"ms_per_step": [0-9.]*')
  echo "$v $* $r"
done
