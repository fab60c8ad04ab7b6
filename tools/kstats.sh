#!/bin/bash
# kernel statistics (rocprofv3 --kernel-trace --stats) of a short headline bench run, filtered: tools/kstats.sh <out file> <grep pattern> [bench args]
OUTF=$1; PAT=$2; shift 2
export TMPDIR=/tmp
D=$PWD/gpurun_out/kstats_tmp
rm -rf $D
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $D -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline "$@" > /dev/null 2>&1
f=$(find $D -name "*kernel_stats.csv" | head -1)
python3 - "$f" "$PAT" > $OUTF <<'PY'
import csv,sys,re
rows=list(csv.DictReader(open(sys.argv[1])))
pat=re.compile(sys.argv[2])
for r in rows:
    n=r['Name'].split('(')[0].replace('void jnr::','')
    if pat.search(n): print(f"{n[:72]:72s} {int(r['Calls']):6d} calls {float(r['TotalDurationNs'])/1e6:9.3f} ms {float(r['AverageNs'])/1e3:9.1f} us avg")
PY
rm -rf $D
cat $OUTF
