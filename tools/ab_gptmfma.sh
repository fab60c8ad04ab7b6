#!/bin/bash
# configs[4] (gpt-mini + yolox-s encoder) with the decision step on the VALU kernel, on the MFMA kernel with 4 and with 16 agents per workgroup
for v in JN_DUMMY=0 "JN_GPT_MFMA=1 JN_GPT_MFMA_AGENTS=4" "JN_GPT_MFMA=1 JN_GPT_MFMA_AGENTS=16"; do
  for m in "--mode rollout" ""; do
    r=$(env $v python3 bench.py --config c5 $m --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | grep -o '"ms_per_step": [0-9.]*')
    echo "$v c5 $m $r"
  done
done
