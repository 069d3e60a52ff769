#!/bin/bash
# round 6, call 8: conv remainder split as two sub-range launches (RANGE variant of conv_halo4): parity, A/B, per-op table
mkdir -p gpurun_out; cd $GRAFT_REPO_ROOT
O=gpurun_out/r06_8; mkdir -p $O
timeout 1500 python3 -m pytest tests/test_gpu_full.py tests/test_gpu_emul.py -x -q -s -k "remainder_split or batch64 or ddim_50 or emul" 2>&1 </dev/null | grep -v amdgpu.ids | grep -i "remainder\|passed\|failed\|error" | tail -8 > $O/tests.log
AB_TAG=conv_rem timeout 900 tools/ab_bench.sh "RDM_CONV_REM=0" "" </dev/null
cp gpurun_out/ab_conv_rem.log $O/
timeout 300 python3 tools/op_trace.py --batch 64 --k 4 --steps 4 --out $O/op_trace_rem.csv > $O/op_trace_rem.log 2>&1 </dev/null
echo done
