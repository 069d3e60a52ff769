#!/bin/bash
# round 4, GPU call 5: flash-attention inner-loop variants A/B, quantize_x0 surface test
set -u
REPO=$(cd "$(dirname "$0")/../.." && pwd)
OUT="$REPO/gpurun_out/r04_5"; mkdir -p "$OUT"
cd "$REPO"
timeout 300 python -m pytest tests/test_gpu_surface.py -x -q -s -k "quantize" > "$OUT/t_surface.log" 2>&1; echo "surface rc=$?" >> "$OUT/summary.txt"
for v in 0 1 2 3 0 3; do echo "== RDM_FLASH_VAR=$v" >> "$OUT/attn_ab.log"; RDM_FLASH_VAR=$v timeout 200 python tools/attn_bench.py >> "$OUT/attn_ab.log" 2>&1; done
RDM_FLASH_VAR=3 timeout 300 python -m pytest tests/test_gpu_ops.py -x -q -k "flash or attention" > "$OUT/t_attn_v3.log" 2>&1; echo "attn-v3 rc=$?" >> "$OUT/summary.txt"
tail -n 5 "$OUT/t_surface.log" "$OUT/t_attn_v3.log"; cat "$OUT/summary.txt"; grep -v amdgpu.ids "$OUT/attn_ab.log"
