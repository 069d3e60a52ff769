#!/bin/bash
# round 6, call 4: one-pass GroupNorm with 52 vectors per thread at the 64 x 64 level (8-group slices of 192 / 384-channel tensors): op timing, parity, headline A/B
mkdir -p gpurun_out; cd $GRAFT_REPO_ROOT
O=gpurun_out/r06_4; mkdir -p $O
RDM_GN1PASS_MAXHW=4096 RDM_GN1PASS_NV512=52 timeout 600 python3 -m pytest tests/test_gpu_ops.py -x -q -k "groupnorm" 2>&1 </dev/null | tail -5 > $O/tests_gn52.log
RDM_GN1PASS_MAXHW=4096 RDM_GN1PASS_NV512=52 timeout 300 python3 tools/gn_bench.py 2>&1 </dev/null | grep -v amdgpu.ids > $O/gn_bench_52.log
timeout 300 python3 tools/gn_bench.py 2>&1 </dev/null | grep -v amdgpu.ids > $O/gn_bench_default.log
AB_TAG=gn52 timeout 900 tools/ab_bench.sh "" "RDM_GN1PASS_MAXHW=4096 RDM_GN1PASS_NV512=52" </dev/null
cp gpurun_out/ab_gn52.log $O/
echo done
