#!/bin/bash
# Per-op backward tables (single stream) with debug builds of the library that leave out one part of pw_bwd_fused_kernel each
# (scratch/dbg/<variant>/libjnroll.so, built with -DJN_DBG_<variant>): where the fused 1x1 backward spends its time.
# Build them HERE first (no GPU needed), one per variant NO_P3 NO_RED NO_P2 NO_STAGE_MATH and ALL (= all four flags):
#   touch jolineedle_amd/csrc/kernels_bwd.hip; JN_EXTRA_FLAGS=-DJN_DBG_NO_P3 JN_LIB_OUT=$PWD/scratch/dbg/NO_P3 bash jolineedle_amd/csrc/build.sh
# and rebuild the product library afterwards (touch + build.sh without flags).  Result of round 4: profiles/r04_bwd_attrib.txt.
OUT=$PWD/gpurun_out
for v in BASE NO_P3 NO_RED NO_P2 NO_STAGE_MATH ALL; do
  lib=""; [ $v != BASE ] && lib="JNROLL_LIB=$PWD/scratch/dbg/$v/libjnroll.so"
  env $lib JN_BWD_PROFILE=1 JN_NO_AUX_STREAM=1 timeout -k 10 200 python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline 2> $OUT/attr_raw.txt > /dev/null
  awk '/^# backward profile/{buf=""} {buf=buf $0 "\n"} /^# total/{last=buf} END{printf "%s", last}' $OUT/attr_raw.txt > $OUT/attr_$v.txt
done
rm -f $OUT/attr_raw.txt
python3 - <<'PY'
import re,os
out=os.environ.get('PWD','.')+'/gpurun_out'
vs=['BASE','NO_P3','NO_RED','NO_P2','NO_STAGE_MATH','ALL']
tabs={}
for v in vs:
    d={}
    for l in open(f'{out}/attr_{v}.txt'):
        m=re.match(r'(\w+)\s+(\S+)\s+in\s+(\S+)\s+out\s+(\S+)\s+s\d acc \d\s+([\d.]+) us\s+([\d.]+) MB',l)
        if m: d[m.group(2)]=(m.group(1),m.group(3),m.group(4),float(m.group(5)),float(m.group(6)))
    tabs[v]=d
print(f"{'layer':44s} {'in':>12s} {'MB':>8s} " + ' '.join(f'{v:>13s}' for v in vs))
for k,(kind,i,o,t,mb) in tabs['BASE'].items():
    if kind!='pw': continue
    print(f"{k:44s} {i:>12s} {mb:8.0f} " + ' '.join(f"{tabs[v].get(k,(0,0,0,0,0))[3]:13.1f}" for v in vs))
PY
