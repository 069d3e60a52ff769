#!/bin/bash
# round 6, call 23: second same-box A/B of the one kept change (skip_connection GEMM on the side stream), then the GPU suite once more on HEAD
mkdir -p gpurun_out; cd $GRAFT_REPO_ROOT
O=gpurun_out/r06_23; mkdir -p $O
AB_TAG=skip_overlap2 timeout 900 tools/ab_bench.sh "RDM_SKIP_OVERLAP=0" "" </dev/null
cp gpurun_out/ab_skip_overlap2.log $O/
timeout 1500 python3 -m pytest tests -x -q -m gpu > $O/tests_run1.log 2>&1 </dev/null
echo "run 1 exit code $?" >> $O/tests_run1.log
grep -v amdgpu.ids $O/tests_run1.log | grep "passed\|failed\|exit code\|Fatal\|Error" | tail -6 > $O/tests_run1_tail.log
echo done
