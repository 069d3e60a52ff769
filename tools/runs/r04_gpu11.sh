#!/bin/bash
# PMC passes over the TN wgrad kernel at the training shapes
set -u
REPO=$(cd "$(dirname "$0")/../.." && pwd)
OUT="$REPO/gpurun_out/r04_11"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --output-format csv --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM GRBM_GUI_ACTIVE -d "$OUT/sq1" -- python3 "$REPO/tools/wgrad_bench.py" --reps 2 > "$OUT/sq1.log" 2>&1
timeout 600 rocprofv3 --kernel-trace --output-format csv --pmc SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_WAVES -d "$OUT/sq2" -- python3 "$REPO/tools/wgrad_bench.py" --reps 2 > "$OUT/sq2.log" 2>&1
python3 "$REPO/tools/pmc_sum.py" counters "$OUT/sq.csv" "$OUT/sq1" "$OUT/sq2"
rm -rf "$OUT/sq1" "$OUT/sq2"
grep "wgrad_tn" "$OUT/sq.csv"
