#!/bin/bash
# Same-box comparison of several builds / switch settings of the headline bench: tools/ab_multi.sh TAG "ENV.." "ENV.." ...  (two rounds, interleaved)
TAG=$1; shift
OUT=gpurun_out/abm_$TAG.log; mkdir -p gpurun_out; : > $OUT
for rep in 1 2; do
  for v in "$@"; do
    line=$(env $v python3 bench.py --no-cpu-baseline --no-extras --steps 3 --warmup 1 2>>gpurun_out/abm_$TAG.err | tail -1)
    echo "[${v:-default}] $(echo "$line" | python3 -c 'import sys,json; d=json.loads(sys.stdin.read()); print("%.3f img/s  %.1f ms/step" % (d["value"], d["ms_per_step"]))')" | tee -a $OUT
  done
done
