#!/bin/bash
# The decision step of gpt-mini outside the engine: time per step of the VALU and of the MFMA kernel with the phase stamps
# (tools/gptstepbench.hip), and the matrix-pipe busy share of each by counters.  -> gpurun_out/<tag>_gptstepbench*.txt
TAG=${1:-r04}
OUT=$PWD/gpurun_out
export TMPDIR=/tmp
timeout -k 10 60 tools/gptstepbench 192 6 6 > $OUT/${TAG}_gptstepbench.txt 2>&1 || exit 1
timeout -k 10 60 tools/gptstepbench 192 1 6 > $OUT/${TAG}_gptstepbench_one_layer_l2_warm.txt 2>&1 || exit 1
rm -rf $OUT/pmc_gptstep
timeout -k 10 200 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES --output-format csv -d $OUT/pmc_gptstep -- tools/gptstepbench 192 6 6 > $OUT/${TAG}_gptstep_pmc.log 2>&1 || exit 1
python3 tools/pmc_summary.py $OUT/pmc_gptstep $OUT/${TAG}_gptstepbench_pmc_lds_mfma.txt
rm -rf $OUT/pmc_gptstep
cat $OUT/${TAG}_gptstepbench.txt $OUT/${TAG}_gptstepbench_pmc_lds_mfma.txt
