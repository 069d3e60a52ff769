#!/bin/bash
# round 5: LayerNorm inside the eight-wave 64-row skinny GEMM from 384 rows on: parity at 256 / 512 sequences + A/B
mkdir -p gpurun_out; cd $GRAFT_REPO_ROOT
timeout 900 python3 -m pytest tests/test_gpu_rarm.py -x -q -k "big_batches or batch_64 or repeats" 2>&1 </dev/null | tail -4 > gpurun_out/e6_tests.log
BATCHES="512" timeout 900 tools/rarm_sweep.sh "RDM_SGEMM_LN8_FROM=0" "X=0" "RDM_SGEMM_LN8_FROM=0" "X=0" </dev/null
cp gpurun_out/rarm_sweep.log gpurun_out/e6_rarm_sweep.log
BATCHES="256" timeout 900 tools/rarm_sweep.sh "X=0" "RDM_SGEMM_LN8_FROM=256" </dev/null
cat gpurun_out/rarm_sweep.log >> gpurun_out/e6_rarm_sweep.log
