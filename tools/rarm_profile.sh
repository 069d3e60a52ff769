#!/bin/bash
# RARM decode (config #5) at several batch sizes: bench line + rocprofv3 kernel stats per batch  ->  gpurun_out/rarm_<tag>/
TAG=${1:-r05}; shift
BATCHES=${@:-"64 256 512"}
REPO=$(pwd); OUT=$REPO/gpurun_out/rarm_$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for b in $BATCHES; do
  python3 $REPO/bench.py --config 5 --batch $b --db-rows 2000000 --steps 2 --warmup 1 --no-cpu-baseline > $OUT/bench_b$b.json 2> $OUT/bench_b$b.err
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/st_b$b -- python3 $REPO/bench.py --config 5 --batch $b --db-rows 2000000 --steps 1 --warmup 1 --no-cpu-baseline --no-extras > $OUT/st_b$b.log 2>&1
  python3 $REPO/tools/pmc_sum.py stats $OUT/kernel_stats_b$b.csv $OUT/st_b$b
  rm -rf $OUT/st_b$b
  python3 -c "import json,sys; d=json.load(open('$OUT/bench_b$b.json')); print('batch $b:', round(d['value'],1), 'img/s', round(d['ms_per_step'],1), 'ms/step')"
  head -14 $OUT/kernel_stats_b$b.csv
done
