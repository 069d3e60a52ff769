#!/bin/bash
# round 5: sgemm at 512 rows: 64 x 96 tiles for q | k | v (one round of CUs), six k-steps per load batch for K = 3072: parity + A/B
mkdir -p gpurun_out; cd $GRAFT_REPO_ROOT
timeout 900 python3 -m pytest tests/test_gpu_rarm.py -x -q -k "big_batches or repeats" 2>&1 </dev/null | tail -4 > gpurun_out/e7_tests.log
timeout 300 python3 -m pytest tests/test_gpu_ops.py -x -q -k "sgemm or skinny" 2>&1 </dev/null | tail -3 >> gpurun_out/e7_tests.log
BATCHES="512" timeout 1200 tools/rarm_sweep.sh "RDM_SGEMM_N96=0 RDM_SGEMM_U6=0" "RDM_SGEMM_N96=0" "RDM_SGEMM_U6=0" "X=0" "RDM_SGEMM_N96=0 RDM_SGEMM_U6=0" "X=0" </dev/null
cp gpurun_out/rarm_sweep.log gpurun_out/e7_rarm_sweep.log
