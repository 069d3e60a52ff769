#!/bin/bash
# round 5: head-major K/V cache: parity (RARM suite) + A/B at 256 / 512 / 1024 sequences
mkdir -p gpurun_out; cd $GRAFT_REPO_ROOT
timeout 900 python3 -m pytest tests/test_gpu_rarm.py -x -q 2>&1 </dev/null | tail -4 > gpurun_out/e9_tests.log
BATCHES="64 256 512 1024" timeout 1200 tools/rarm_sweep.sh "RDM_RARM_CACHE_ROWMAJOR=1" "X=0" </dev/null
cp gpurun_out/rarm_sweep.log gpurun_out/e9_rarm_sweep.log
