#!/bin/bash
# train step with the TN wgrad kernel: wall time, kernel stats, gradient/training parity tests
set -u
REPO=$(cd "$(dirname "$0")/../.." && pwd)
OUT="$REPO/gpurun_out/r04_9"; mkdir -p "$OUT"
cd "$REPO"
timeout 900 python -m pytest tests/test_gpu_training.py -x -q > "$OUT/t_training.log" 2>&1; echo "training rc=$?" > "$OUT/summary.txt"
timeout 600 python tools/train_step_bench.py 64 4 > "$OUT/train_step_b64.log" 2>&1
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/train_trace" -- python3 "$REPO/tools/train_step_bench.py" 64 3 > "$OUT/train_trace.log" 2>&1
python3 "$REPO/tools/pmc_sum.py" stats "$OUT/train_step_kernel_stats.csv" "$OUT/train_trace"
rm -rf "$OUT/train_trace"
tail -n 5 "$OUT/t_training.log"; cat "$OUT/summary.txt"; tail -n 4 "$OUT/train_step_b64.log"; head -45 "$OUT/train_step_kernel_stats.csv"
