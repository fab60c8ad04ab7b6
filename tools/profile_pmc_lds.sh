#!/bin/bash
# LDS bank-conflict share and matrix-pipe busy share per kernel of the headline iteration (counters only, one pass).
# usage: tools/profile_pmc_lds.sh <tag>    -> gpurun_out/<tag>_pmc_lds_mfma_train_iteration.txt
TAG=${1:-r03}
OUT=$PWD/gpurun_out
export TMPDIR=/tmp
rm -rf $OUT/pmc_lds
timeout -k 10 400 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES --output-format csv -d $OUT/pmc_lds -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline > $OUT/${TAG}_pmc_lds.log 2>&1
python3 tools/pmc_summary.py $OUT/pmc_lds $OUT/${TAG}_pmc_lds_mfma_train_iteration.txt
rm -rf $OUT/pmc_lds
head -30 $OUT/${TAG}_pmc_lds_mfma_train_iteration.txt
