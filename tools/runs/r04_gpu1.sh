#!/bin/bash
# round 4, GPU call 1: LayerNorm-fold parity + row-63 goldens + same-box A/B of the fold on the headline bench
set -u
REPO=$(cd "$(dirname "$0")/../.." && pwd)
OUT="$REPO/gpurun_out/r04_1"; mkdir -p "$OUT"
cd "$REPO"
timeout 600 python -m pytest tests/test_gpu_ops.py -x -q -k "linear" > "$OUT/t_ops.log" 2>&1; echo "ops rc=$?" >> "$OUT/summary.txt"
timeout 900 python -m pytest tests/test_gpu_full.py -x -q -s -k "ddim_50_steps_shipped_batch64 or ddpm_250_steps_shipped_k16_batch64 or vq_decode_shipped or test_ddim_50_steps_shipped" > "$OUT/t_full.log" 2>&1; echo "full rc=$?" >> "$OUT/summary.txt"
timeout 600 python -m pytest tests/test_gpu_models.py -x -q > "$OUT/t_models.log" 2>&1; echo "models rc=$?" >> "$OUT/summary.txt"
RDM_NO_LNFOLD=1 timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline > "$OUT/bench_nofold.json" 2> "$OUT/bench_nofold.err"
timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline > "$OUT/bench_fold.json" 2> "$OUT/bench_fold.err"
RDM_LIN4_PROF=1 timeout 300 python tools/lin_ln_bench.py > "$OUT/lin_ln_bench.log" 2>&1
for f in "$OUT"/t_*.log; do tail -n 4 "$f"; done; cat "$OUT/summary.txt"
python - <<PY
import json
for n in ("nofold","fold"):
    try:
        d=json.load(open("$OUT/bench_%s.json"%n)); r=d["roofline"]
        print(n, round(d["value"],2), "img/s; conv frac", round(r["frac"],3), "lin", round(r["linear_gemm"]["frac"],3), r["linear_gemm"]["time_ms_per_step"], "ln", r.get("layernorm",{}).get("time_ms_per_step"))
    except Exception as e: print(n, "failed", e)
PY
cat "$OUT/lin_ln_bench.log" | tail -30
