#!/bin/bash
# round 6, call 3: skip_connection GEMM on a side stream (A/B), one-pass GroupNorm with 48 vectors per thread at the 64 x 64 level (A/B), parity of both
mkdir -p gpurun_out; cd $GRAFT_REPO_ROOT
O=gpurun_out/r06_3; mkdir -p $O
timeout 900 python3 -m pytest tests/test_gpu_emul.py tests/test_gpu_models.py -x -q 2>&1 </dev/null | tail -5 > $O/tests_default.log
RDM_GN1PASS_MAXHW=4096 RDM_GN1PASS_NV512=48 timeout 600 python3 -m pytest tests/test_gpu_ops.py -x -q -k "groupnorm" 2>&1 </dev/null | tail -5 > $O/tests_gn48.log
RDM_GN1PASS_MAXHW=4096 RDM_GN1PASS_NV512=48 timeout 300 python3 tools/gn_bench.py 2>&1 </dev/null | grep -v amdgpu.ids > $O/gn_bench_48.log
timeout 300 python3 tools/gn_bench.py 2>&1 </dev/null | grep -v amdgpu.ids > $O/gn_bench_default.log
AB_TAG=skip_overlap timeout 900 tools/ab_bench.sh "RDM_SKIP_OVERLAP=0" "" </dev/null
AB_TAG=gn48 timeout 900 tools/ab_bench.sh "" "RDM_GN1PASS_MAXHW=4096 RDM_GN1PASS_NV512=48" </dev/null
cp gpurun_out/ab_skip_overlap.log gpurun_out/ab_gn48.log $O/
echo done
