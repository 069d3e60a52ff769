#!/bin/bash
# round 6, call 22: norm3 of the unconditional rows on the side stream beside the cross-attention kernel: parity (stage-level emulator tests, goldens), A/B
mkdir -p gpurun_out; cd $GRAFT_REPO_ROOT
O=gpurun_out/r06_22; mkdir -p $O
timeout 1200 python3 -m pytest tests/test_gpu_emul.py tests/test_gpu_full.py -x -q 2>&1 </dev/null | grep -v amdgpu.ids | tail -3 > $O/tests.log
AB_TAG=ln3_overlap timeout 900 tools/ab_bench.sh "RDM_LN3_OVERLAP=0" "" </dev/null
cp gpurun_out/ab_ln3_overlap.log $O/
echo done
