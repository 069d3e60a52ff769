#!/bin/bash
# round 5: two-pass GroupNorm walked in sample ranges (memory-side cache reuse): parity + A/B over the range size
mkdir -p gpurun_out; cd $GRAFT_REPO_ROOT
timeout 600 python3 -m pytest tests/test_gpu_ops.py -x -q -k "groupnorm or group_norm or gn" 2>&1 </dev/null | tail -3 > gpurun_out/e3_tests.log
timeout 1200 tools/ab_multi.sh e3 "RDM_GN_RANGE_MB=0" "RDM_GN_RANGE_MB=24" "RDM_GN_RANGE_MB=48" "RDM_GN_RANGE_MB=96" "" </dev/null
