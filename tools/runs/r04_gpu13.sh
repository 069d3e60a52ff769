#!/bin/bash
# flash d32 inner loop with packed fp32 ops: kernel A/B and headline A/B on one box (old build = librdm_hip_old.so)
set -u
REPO=$(cd "$(dirname "$0")/../.." && pwd)
OUT="$REPO/gpurun_out/r04_13"; mkdir -p "$OUT"
cd "$REPO"
L="$REPO/retrieval-augmented-diffusion-models_amd/librdm_hip_old.so"
echo "== old" > "$OUT/attn_ab.log"; RDM_HIP_LIB=$L timeout 300 python tools/attn_bench.py >> "$OUT/attn_ab.log" 2>&1
echo "== new" >> "$OUT/attn_ab.log"; timeout 300 python tools/attn_bench.py >> "$OUT/attn_ab.log" 2>&1
echo "== old" >> "$OUT/attn_ab.log"; RDM_HIP_LIB=$L timeout 300 python tools/attn_bench.py >> "$OUT/attn_ab.log" 2>&1
echo "== new" >> "$OUT/attn_ab.log"; timeout 300 python tools/attn_bench.py >> "$OUT/attn_ab.log" 2>&1
grep -v amdgpu.ids "$OUT/attn_ab.log"
timeout 600 python -m pytest tests/test_gpu_ops.py -x -q -k "attention" 2>&1 | tail -3
RDM_HIP_LIB=$L timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras > "$OUT/bench_old.json" 2> "$OUT/bench_old.err"
timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras > "$OUT/bench_new.json" 2> "$OUT/bench_new.err"
RDM_HIP_LIB=$L timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras > "$OUT/bench_old2.json" 2> "$OUT/bench_old2.err"
timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras > "$OUT/bench_new2.json" 2> "$OUT/bench_new2.err"
python - <<PY
import json
for n in ("old","new","old2","new2"):
    try: d=json.load(open("$OUT/bench_%s.json"%n)); print(n, round(d["value"],2), "img/s", round(d["ms_per_step"],1), "ms/step")
    except Exception as e: print(n, "failed", e)
PY
