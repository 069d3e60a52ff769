#!/bin/bash
set -u
REPO=$(cd "$(dirname "$0")/../.." && pwd)
OUT="$REPO/gpurun_out/r04_12"; mkdir -p "$OUT"
cd "$REPO"
timeout 900 python -m pytest tests/test_gpu_backward.py -x -q -s -k "wgrad" 2>&1 | grep -v "^$" | tail -25
timeout 600 python tools/wgrad_bench.py > "$OUT/wgrad_conv9.log" 2>&1; head -12 "$OUT/wgrad_conv9.log"
RDM_NO_WGRAD_CONV9=1 timeout 600 python tools/wgrad_bench.py > "$OUT/wgrad_pertap.log" 2>&1; head -12 "$OUT/wgrad_pertap.log"
