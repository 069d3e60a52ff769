#!/bin/bash
# round 5: first-stage decode walked in sample ranges (halo convs instead of the generic implicit GEMM on > 2^31-element activations)
mkdir -p gpurun_out; cd $GRAFT_REPO_ROOT
timeout 600 python3 -m pytest tests/test_gpu_rarm.py -x -q -k "vq_decode_walked or vqgan_decode" 2>&1 </dev/null | tail -4 > gpurun_out/e5_tests.log
BATCHES="256 512" timeout 900 tools/rarm_sweep.sh "RDM_VQ_RANGE=100000" "X=0" "RDM_VQ_RANGE=64" </dev/null
cp gpurun_out/rarm_sweep.log gpurun_out/e5_rarm_sweep.log
