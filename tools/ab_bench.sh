#!/bin/bash
# Same-box A/B of the headline bench: tools/ab_bench.sh "ENV_A=1" "ENV_B=1" [extra bench args]  -> alternating runs A B A B, img/s each
# (an empty string = the default tree).  Output: gpurun_out/ab_<tag>.log
A="$1"; B="$2"; shift 2
TAG=${AB_TAG:-ab}
OUT=gpurun_out/ab_$TAG.log
mkdir -p gpurun_out; : > $OUT
for rep in 1 2; do
  for v in "$A" "$B"; do
    line=$(env $v python3 bench.py --no-cpu-baseline --no-extras --steps 3 --warmup 1 "$@" 2>>gpurun_out/ab_$TAG.err | tail -1)
    echo "[${v:-default}] $(echo "$line" | python3 -c 'import sys,json; d=json.loads(sys.stdin.read()); print("%.3f img/s  %.1f ms/step  conv frac %.3f" % (d["value"], d["ms_per_step"], d["roofline"]["frac"]))')" | tee -a $OUT
  done
done
