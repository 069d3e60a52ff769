#!/bin/bash
# Upsample convs by output phase: op tests, kernel A/B, model goldens, headline A/B
set -u
REPO=$(cd "$(dirname "$0")/../.." && pwd)
OUT="$REPO/gpurun_out/r04_14"; mkdir -p "$OUT"
cd "$REPO"
timeout 600 python -m pytest tests/test_gpu_ops.py -x -q -k "conv3x3" 2>&1 | tail -4
echo "== phase" > "$OUT/ups_ab.log"; timeout 300 python tools/ups_bench.py >> "$OUT/ups_ab.log" 2>&1
echo "== nine taps (RDM_NO_UPS_PHASE=1)" >> "$OUT/ups_ab.log"; RDM_NO_UPS_PHASE=1 timeout 300 python tools/ups_bench.py >> "$OUT/ups_ab.log" 2>&1
grep -v amdgpu "$OUT/ups_ab.log"
timeout 900 python -m pytest tests/test_gpu_full.py tests/test_gpu_models.py -x -q 2>&1 | tail -4
RDM_NO_UPS_PHASE=1 timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras > "$OUT/bench_old.json" 2> "$OUT/bench_old.err"
timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras > "$OUT/bench_new.json" 2> "$OUT/bench_new.err"
RDM_NO_UPS_PHASE=1 timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras > "$OUT/bench_old2.json" 2> "$OUT/bench_old2.err"
timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras > "$OUT/bench_new2.json" 2> "$OUT/bench_new2.err"
python - <<PY
import json
for n in ("old","new","old2","new2"):
    try: d=json.load(open("$OUT/bench_%s.json"%n)); print(n, round(d["value"],2), "img/s", round(d["ms_per_step"],1), "ms/step", "conv frac", round(d["roofline"]["frac"],3))
    except Exception as e: print(n, "failed", e)
PY
