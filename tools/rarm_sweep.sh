#!/bin/bash
# RARM decode (config #5) img/s at several batch sizes under dev-switch settings given as "ENV=.. ENV=.." strings (A/B on one box)
OUT=gpurun_out/rarm_sweep.log; mkdir -p gpurun_out; : > $OUT
BATCHES=${BATCHES:-"64 128 256 512"}
for v in "$@"; do
  for b in $BATCHES; do
    line=$(env $v python3 bench.py --config 5 --batch $b --db-rows 2000000 --steps 2 --warmup 1 --no-cpu-baseline --no-extras 2>/dev/null | tail -1)
    echo "[${v:-default}] batch $b: $(echo "$line" | python3 -c 'import sys,json; d=json.loads(sys.stdin.read()); print("%.1f img/s  %.1f ms/step" % (d["value"], d["ms_per_step"]))')" | tee -a $OUT
  done
done
