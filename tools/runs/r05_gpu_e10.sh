#!/bin/bash
# round 5: eight-wave K/V-cache attention: parity (RARM suite) + A/B at 256 / 512 / 1024 sequences
mkdir -p gpurun_out; cd $GRAFT_REPO_ROOT
timeout 900 python3 -m pytest tests/test_gpu_rarm.py -x -q 2>&1 </dev/null | tail -4 > gpurun_out/e10_tests.log
BATCHES="256 512 1024" timeout 1200 tools/rarm_sweep.sh "RDM_RARM_ATTN_NW8_FROM=0" "X=0" "RDM_RARM_ATTN_NW8_FROM=0" "X=0" </dev/null
cp gpurun_out/rarm_sweep.log gpurun_out/e10_rarm_sweep.log
