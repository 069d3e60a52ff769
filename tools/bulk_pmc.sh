#!/bin/bash
# PMC passes over the bulk neighbour scan with its ablation bits (GPU box only).  The ablation switch RDM_KNN_BULK_DBG exists
# only in a library built with  make -C retrieval-augmented-diffusion-models_amd/csrc clean all CXXFLAGS+=-DRDM_DEBUG_ABLATION
set -euo pipefail
R=$(cd "$(dirname "$0")/.." && pwd)
O=$R/gpurun_out/bulkpmc; mkdir -p "$O"
cd /tmp; export TMPDIR=/tmp
for d in 8 1 2; do
  export RDM_KNN_BULK_DBG=$d
  rocprofv3 --kernel-trace --output-format csv --pmc ${PMC:-SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL GRBM_GUI_ACTIVE} -d "$O/d$d" -- python3 "$R/tools/knn_bulk_bench.py" 20927907 1024 20 > "$O/d$d.log" 2>&1
  python3 "$R/tools/pmc_sum.py" counters "$O/d$d.csv" "$O/d$d"
  grep -i bulk "$O/d$d.csv" | head -3
done
