#!/bin/bash
# round 4, GPU call 6: RARM decode kernels (16 x 16 skinny tiles, 4-wave cache attention, split cross-attention): parity + A/B
set -u
REPO=$(cd "$(dirname "$0")/../.." && pwd)
OUT="$REPO/gpurun_out/r04_6"; mkdir -p "$OUT"
cd "$REPO"
timeout 900 python -m pytest tests/test_gpu_rarm.py -x -q > "$OUT/t_rarm.log" 2>&1; echo "rarm rc=$?" >> "$OUT/summary.txt"
timeout 600 python -m pytest tests/test_gpu_ops.py tests/test_gpu_models.py -x -q -k "skinny or linear or unet or clip" > "$OUT/t_lin.log" 2>&1; echo "lin rc=$?" >> "$OUT/summary.txt"
timeout 600 python -m pytest tests/test_gpu_surface.py -x -q -k "deterministic" > "$OUT/t_det.log" 2>&1; echo "det rc=$?" >> "$OUT/summary.txt"
for b in 64 128 256 512; do timeout 400 python bench.py --config 5 --batch $b --steps 2 --warmup 1 --no-cpu-baseline --no-extras > "$OUT/bench_c5_b$b.json" 2> "$OUT/bench_c5_b$b.err"; done
RDM_NO_RARM_XSPLIT=1 timeout 400 python bench.py --config 5 --batch 64 --steps 2 --warmup 1 --no-cpu-baseline --no-extras > "$OUT/bench_c5_b64_nosplit.json" 2> "$OUT/bench_c5_b64_nosplit.err"
RDM_SGEMM_MA=2 RDM_SGEMM_NB=2 RDM_NO_RARM_XSPLIT=1 timeout 400 python bench.py --config 5 --batch 64 --steps 2 --warmup 1 --no-cpu-baseline --no-extras > "$OUT/bench_c5_b64_oldtiles.json" 2> "$OUT/bench_c5_b64_oldtiles.err"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/rstats" -- python3 "$REPO/bench.py" --config 5 --steps 1 --warmup 1 --no-cpu-baseline --no-extras > "$OUT/rstats.log" 2>&1
python3 "$REPO/tools/pmc_sum.py" stats "$OUT/rarm_stats.csv" "$OUT/rstats"; rm -rf "$OUT/rstats"
cd "$REPO"
for f in "$OUT"/t_*.log; do echo "== $f"; tail -n 5 "$f"; done; cat "$OUT/summary.txt"
python - <<PY
import json
for n in ("b64","b128","b256","b512","b64_nosplit","b64_oldtiles"):
    try: d=json.load(open("$OUT/bench_c5_%s.json" % n)); print("config5", n, round(d["value"],1), "img/s", round(d["ms_per_step"],1), "ms/step")
    except Exception as e: print("config5", n, "failed", e)
PY
head -12 "$OUT/rarm_stats.csv"
