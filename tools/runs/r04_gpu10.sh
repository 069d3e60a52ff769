#!/bin/bash
# vectorised GN / LN backward + colsum: parity tests, train step wall time and kernel stats
set -u
REPO=$(cd "$(dirname "$0")/../.." && pwd)
OUT="$REPO/gpurun_out/r04_10"; mkdir -p "$OUT"
cd "$REPO"
timeout 900 python -m pytest tests/test_gpu_backward.py tests/test_gpu_training.py -x -q -s > "$OUT/t_bwd.log" 2>&1; echo "bwd rc=$?" > "$OUT/summary.txt"
timeout 600 python tools/train_step_bench.py 64 4 > "$OUT/train_step_b64.log" 2>&1
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/train_trace" -- python3 "$REPO/tools/train_step_bench.py" 64 3 > "$OUT/train_trace.log" 2>&1
python3 "$REPO/tools/pmc_sum.py" stats "$OUT/train_step_kernel_stats.csv" "$OUT/train_trace"
rm -rf "$OUT/train_trace"
grep -i "groupnorm bwd\|layernorm bwd\|passed\|failed\|error" "$OUT/t_bwd.log" | tail -12; cat "$OUT/summary.txt"; tail -n 4 "$OUT/train_step_b64.log"; head -40 "$OUT/train_step_kernel_stats.csv"
