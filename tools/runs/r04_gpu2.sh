#!/bin/bash
# round 4, GPU call 2: training surface (encoder, glue ops, training_step, shipped-topology gradients), backward suite, RARM batch-64 golden,
# whole-step timing, power probe
set -u
REPO=$(cd "$(dirname "$0")/../.." && pwd)
OUT="$REPO/gpurun_out/r04_2"; mkdir -p "$OUT"
cd "$REPO"
timeout 900 python -m pytest tests/test_gpu_training.py -x -q -s > "$OUT/t_training.log" 2>&1; echo "training rc=$?" >> "$OUT/summary.txt"
timeout 900 python -m pytest tests/test_gpu_backward.py -q > "$OUT/t_backward.log" 2>&1; echo "backward rc=$?" >> "$OUT/summary.txt"
timeout 600 python -m pytest tests/test_gpu_rarm.py -q -s -k "deep" > "$OUT/t_rarm.log" 2>&1; echo "rarm rc=$?" >> "$OUT/summary.txt"
timeout 600 python tools/train_step_bench.py 64 4 > "$OUT/train_step_b64.log" 2>&1
timeout 300 python tools/train_step_bench.py 8 3 > "$OUT/train_step_b8.log" 2>&1
timeout 300 python tools/power_probe.py > "$OUT/power_probe.log" 2>&1
for b in 64 128 256; do timeout 400 python bench.py --config 5 --batch $b --steps 2 --warmup 1 --no-cpu-baseline --no-extras > "$OUT/bench_c5_b$b.json" 2> "$OUT/bench_c5_b$b.err"; done
for f in "$OUT"/t_*.log; do echo "== $f"; tail -n 25 "$f"; done
python - <<PY
import json
for b in (64, 128, 256):
    try: d=json.load(open("$OUT/bench_c5_b%d.json" % b)); print("config5 batch", b, round(d["value"],1), "img/s", round(d["ms_per_step"],1), "ms/step")
    except Exception as e: print("config5 batch", b, "failed", e)
PY
cat "$OUT/summary.txt"; tail -n 5 "$OUT"/train_step_b*.log; cat "$OUT/power_probe.log"
