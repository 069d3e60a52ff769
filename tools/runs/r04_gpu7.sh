#!/bin/bash
# round 4, GPU call 7: gn_stats LDS layout, sampler scans, surface tolerances (measured values), headline + config 5
set -u
REPO=$(cd "$(dirname "$0")/../.." && pwd)
OUT="$REPO/gpurun_out/r04_7"; mkdir -p "$OUT"
cd "$REPO"
timeout 600 python -m pytest tests/test_gpu_ops.py tests/test_gpu_models.py -x -q -k "groupnorm or unet or vq or head" > "$OUT/t_gn.log" 2>&1; echo "gn rc=$?" >> "$OUT/summary.txt"
timeout 600 python -m pytest tests/test_gpu_rarm.py -x -q > "$OUT/t_rarm.log" 2>&1; echo "rarm rc=$?" >> "$OUT/summary.txt"
timeout 900 python -m pytest tests/test_gpu_surface.py -x -q -s > "$OUT/t_surface.log" 2>&1; echo "surface rc=$?" >> "$OUT/summary.txt"
timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline > "$OUT/bench.json" 2> "$OUT/bench.err"
timeout 400 python bench.py --config 5 --steps 2 --warmup 1 --no-cpu-baseline --no-extras > "$OUT/bench_c5.json" 2> "$OUT/bench_c5.err"
for f in "$OUT"/t_gn.log "$OUT"/t_rarm.log; do echo "== $f"; tail -n 4 "$f"; done; tail -n 3 "$OUT/t_surface.log"; grep "\[surface\]" "$OUT/t_surface.log"; cat "$OUT/summary.txt"
python - <<PY
import json
for n in ("bench","bench_c5"):
    try:
        d=json.load(open("$OUT/%s.json"%n)); r=d["roofline"]
        print(n, round(d["value"],2), "img/s", round(d["ms_per_step"],1), "ms/step", {k: round(v.get("time_ms_per_step",0),1) for k,v in r.items() if isinstance(v,dict) and "time_ms_per_step" in v})
    except Exception as e: print(n, "failed", e)
PY
