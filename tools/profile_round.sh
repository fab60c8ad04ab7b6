#!/bin/bash
# Collects the judged profiles of one round on the GPU box (run through gpurun from the repo root):
#   kernel-trace statistics of the headline bench command, the two PMC passes for the HBM traffic of the forward conv
#   stack (FETCH_SIZE, WRITE_SIZE — separate runs, counters only), and the secondary bench lines.
# usage: tools/profile_round.sh <tag>      -> gpurun_out/<tag>_*
set -u
TAG=${1:-r02}
OUT=$PWD/gpurun_out
cd /tmp && export TMPDIR=/tmp
REPO=${GRAFT_REPO_ROOT:-/root/repo}
run_prof() {   # name, extra rocprofv3 args..., --, bench args
  local name=$1; shift
  rm -rf $OUT/prof_$name
  timeout -k 10 400 rocprofv3 "$@" > $OUT/${TAG}_${name}.log 2>&1
}
cd $REPO
run_prof train --kernel-trace --stats --output-format csv -d $OUT/prof_train -- python3 bench.py --steps 4 --warmup 2 --no-cpu-baseline
find $OUT/prof_train -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/${TAG}_train_iteration_kernel_stats.csv
run_prof fetch --pmc FETCH_SIZE --output-format csv -d $OUT/prof_fetch -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline
find $OUT/prof_fetch -name "*counter_collection.csv" | head -1 | xargs -I{} cp {} $OUT/${TAG}_pmc_fetch.csv
run_prof write --pmc WRITE_SIZE --output-format csv -d $OUT/prof_write -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline
find $OUT/prof_write -name "*counter_collection.csv" | head -1 | xargs -I{} cp {} $OUT/${TAG}_pmc_write.csv
python3 tools/pmc_traffic.py $OUT/${TAG}_pmc_fetch.csv $OUT/${TAG}_pmc_write.csv 40 64 train $OUT/${TAG}_pmc_traffic.json profiles/${TAG}_pmc_conv_stack_traffic_train.txt > $OUT/${TAG}_pmc_conv_stack_traffic_train.txt 2>&1
python3 tools/pmc_traffic.py $OUT/${TAG}_pmc_fetch.csv $OUT/${TAG}_pmc_write.csv 40 64 backward $OUT/${TAG}_pmc_traffic.json profiles/${TAG}_pmc_conv_stack_traffic_backward.txt > $OUT/${TAG}_pmc_conv_stack_traffic_backward.txt 2>&1
rm -rf $OUT/prof_fetch $OUT/prof_write
for m in "rollout" "train --sample" "rollout --sample"; do
  n=$(echo $m | tr -d ' -')
  timeout -k 10 200 python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --mode $m > $OUT/${TAG}_bench_$n.json 2>/dev/null
done
timeout -k 10 200 python3 bench.py --steps 5 --warmup 2 > $OUT/${TAG}_bench_train.json 2>/dev/null
tail -c 300 $OUT/${TAG}_bench_train.json
