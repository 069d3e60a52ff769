#!/bin/bash
# round 4, final-tree evidence: profile_round (kernel stats, SQ / FETCH / WRITE passes, RARM stats + SQ), bench lines of configs 2 / 4 / 5,
# the headline with the FULL cpu baseline, the whole-step timing
set -u
REPO=$(cd "$(dirname "$0")/../.." && pwd)
TAG=${1:-r04}
cd "$REPO"
bash tools/profile_round.sh "$TAG" > "gpurun_out/profile_$TAG.log" 2>&1
OUT="$REPO/gpurun_out/prof_$TAG"
for c in 2 4; do timeout 900 python bench.py --config $c --no-cpu-baseline > "$OUT/bench_config$c.json" 2> "$OUT/bench_config$c.err"; done
for b in 64 256; do timeout 600 python bench.py --config 5 --batch $b --steps 2 --warmup 1 --no-cpu-baseline > "$OUT/bench_config5_batch$b.json" 2> "$OUT/bench_config5_batch$b.err"; done
timeout 1500 python bench.py --steps 3 --warmup 1 --full-cpu-baseline > "$OUT/bench_full_cpu_baseline.json" 2> "$OUT/bench_full_cpu_baseline.err"
timeout 600 python tools/train_step_bench.py 64 4 > "$OUT/train_step_b64.log" 2>&1
timeout 600 python tools/wgrad_bench.py > "$OUT/wgrad_shapes.log" 2>&1
(cd /tmp && export TMPDIR=/tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/train_trace" -- python3 "$REPO/tools/train_step_bench.py" 64 3 > "$OUT/train_trace.log" 2>&1)
python3 "$REPO/tools/pmc_sum.py" stats "$OUT/train_step_kernel_stats.csv" "$OUT/train_trace"; rm -rf "$OUT/train_trace"
python - <<PY
import json, glob
for f in sorted(glob.glob("$OUT/bench*.json")):
    try:
        d = json.load(open(f)); r = d.get("roofline", {})
        print(f.split("/")[-1], round(d["value"], 2), d["unit"], "ms/step", round(d["ms_per_step"], 1), "frac", round(r.get("frac", 0), 3), "cpu", d.get("cpu_baseline", {}).get("value"))
    except Exception as e: print(f, "failed", e)
PY
tail -n 3 "$OUT/train_step_b64.log"
