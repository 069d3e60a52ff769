#!/bin/bash
# round 6, call 16: fused feed-forward kernel: parity test and timing against the two kernels it would replace
mkdir -p gpurun_out; cd $GRAFT_REPO_ROOT
O=gpurun_out/r06_16; mkdir -p $O
timeout 600 python3 -m pytest tests/test_gpu_ops.py -x -q -s -k "ffn_fused" 2>&1 </dev/null | grep -v amdgpu.ids | tail -15 > $O/test.log
timeout 600 python3 tools/ffn_bench.py 2>&1 </dev/null | grep -v amdgpu.ids > $O/ffn_bench.log
echo done
