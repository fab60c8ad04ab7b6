#!/bin/bash
# Secondary workload of BASELINE configs[4] (gpt-mini + yolox-s dense-3x3 encoder, 640 px, T = 32, B = 16) on one GPU:
# bench lines, kernel-trace statistics and the LDS / MFMA PMC table for both modes.   usage: tools/profile_c5.sh <tag>
set -u
TAG=${1:-r02}
OUT=$PWD/gpurun_out
export TMPDIR=/tmp
for mode in train rollout; do
  timeout -k 10 200 python3 bench.py --config c5 --mode $mode --steps 3 --warmup 1 --no-cpu-baseline > $OUT/${TAG}_bench_c5_$mode.json 2>/dev/null
  rm -rf $OUT/prof_c5_$mode
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_c5_$mode -- python3 bench.py --config c5 --mode $mode --steps 2 --warmup 1 --no-cpu-baseline > $OUT/${TAG}_c5_$mode.log 2>&1
  find $OUT/prof_c5_$mode -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/${TAG}_c5_${mode}_kernel_stats.csv
  rm -rf $OUT/prof_c5_$mode
  rm -rf $OUT/pmc_c5_$mode
  timeout -k 10 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES --output-format csv -d $OUT/pmc_c5_$mode -- python3 bench.py --config c5 --mode $mode --steps 1 --warmup 1 --no-cpu-baseline > $OUT/${TAG}_c5_pmc_$mode.log 2>&1
  python3 tools/pmc_summary.py $OUT/pmc_c5_$mode $OUT/${TAG}_c5_pmc_lds_mfma_$mode.txt
  rm -rf $OUT/pmc_c5_$mode
done
tail -c 200 $OUT/${TAG}_bench_c5_train.json
