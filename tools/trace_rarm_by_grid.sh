#!/bin/bash
# Per-(kernel, grid) launch statistics of one RARM decode step of config #5 (64 sequences x 256 tokens):  -> gpurun_out/grid_5.csv
set -euo pipefail
exec "$(cd "$(dirname "$0")" && pwd)/trace_by_grid.sh" 5 --config 5 --steps 1 --warmup 0 "$@"
