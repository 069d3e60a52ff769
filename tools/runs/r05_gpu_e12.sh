#!/bin/bash
# round 5: mid-size GEMM (mgemm.hip) for the RARM decode step at 1024+ sequences: parity + A/B
mkdir -p gpurun_out; cd $GRAFT_REPO_ROOT
timeout 900 python3 -m pytest tests/test_gpu_rarm.py -x -q -s -k "mid_size or big_batches" 2>&1 </dev/null | tail -12 > gpurun_out/e12_tests.log
BATCHES="1024 2048 4096" timeout 1200 tools/rarm_sweep.sh "RDM_MGEMM_FROM=0" "X=0" "RDM_MGEMM_BM=64" "RDM_MGEMM_BM=128" </dev/null
cp gpurun_out/rarm_sweep.log gpurun_out/e12_rarm_sweep.log
