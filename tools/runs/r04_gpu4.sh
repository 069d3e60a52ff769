#!/bin/bash
# round 4, GPU call 4: wide-image strips of conv_halo4 (op tests, VQ decode goldens, A/B on the headline), quantize_x0 / all-reduce surface tests
set -u
REPO=$(cd "$(dirname "$0")/../.." && pwd)
OUT="$REPO/gpurun_out/r04_4"; mkdir -p "$OUT"
cd "$REPO"
timeout 600 python -m pytest tests/test_gpu_ops.py -x -q -k "conv3x3" > "$OUT/t_conv.log" 2>&1; echo "conv rc=$?" >> "$OUT/summary.txt"
timeout 600 python -m pytest tests/test_gpu_full.py -x -q -s -k "vq_decode_shipped" > "$OUT/t_vq.log" 2>&1; echo "vq rc=$?" >> "$OUT/summary.txt"
timeout 600 python -m pytest tests/test_gpu_models.py tests/test_gpu_rarm.py -x -q -k "vq or decode or vqgan" > "$OUT/t_vq2.log" 2>&1; echo "vq2 rc=$?" >> "$OUT/summary.txt"
timeout 600 python -m pytest tests/test_gpu_surface.py -x -q -s -k "quantize or rccl or ddim_sampler_surface" > "$OUT/t_surface.log" 2>&1; echo "surface rc=$?" >> "$OUT/summary.txt"
timeout 600 python -m pytest tests/test_gpu_training.py -x -q -k "vq_encode or training_step_from_images" > "$OUT/t_training.log" 2>&1; echo "training rc=$?" >> "$OUT/summary.txt"
RDM_NO_HALO4_STRIP=1 timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras > "$OUT/bench_nostrip.json" 2> "$OUT/bench_nostrip.err"
timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras > "$OUT/bench_strip.json" 2> "$OUT/bench_strip.err"
timeout 300 python tools/vq_decode_bench.py > "$OUT/vq_decode_bench.log" 2>&1
for f in "$OUT"/t_*.log; do echo "== $f"; tail -n 6 "$f"; done
cat "$OUT/summary.txt"
python - <<PY
import json
for n in ("nostrip","strip"):
    try: d=json.load(open("$OUT/bench_%s.json"%n)); print(n, round(d["value"],2), "img/s", round(d["ms_per_step"],1), "ms/step")
    except Exception as e: print(n, "failed", e)
PY
cat "$OUT/vq_decode_bench.log"
