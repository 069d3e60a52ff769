#!/bin/bash
# round 4, GPU call 3: training surface test again + kernel trace of the whole optimisation step at the shipped topology
set -u
REPO=$(cd "$(dirname "$0")/../.." && pwd)
OUT="$REPO/gpurun_out/r04_3"; mkdir -p "$OUT"
cd "$REPO"
timeout 900 python -m pytest tests/test_gpu_training.py -x -q -s > "$OUT/t_training.log" 2>&1; echo "training rc=$?" >> "$OUT/summary.txt"
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/train_trace" -- python3 "$REPO/tools/train_step_bench.py" 64 3 > "$OUT/train_trace.log" 2>&1
python3 "$REPO/tools/pmc_sum.py" stats "$OUT/train_step_kernel_stats.csv" "$OUT/train_trace"
rm -rf "$OUT/train_trace"
cd "$REPO"
tail -n 30 "$OUT/t_training.log"; cat "$OUT/summary.txt"; head -40 "$OUT/train_step_kernel_stats.csv"
