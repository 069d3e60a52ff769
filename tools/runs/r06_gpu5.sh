#!/bin/bash
# round 6, call 5: conv tail split (16 x 16 level: head rows + K-split tail on the side stream): parity, A/B
mkdir -p gpurun_out; cd $GRAFT_REPO_ROOT
O=gpurun_out/r06_5; mkdir -p $O
timeout 1200 python3 -m pytest tests/test_gpu_full.py -x -q -s -k "tail_split or batch64 or ddim_50" 2>&1 </dev/null | grep -v amdgpu.ids | tail -25 > $O/tests.log
AB_TAG=conv_tail timeout 900 tools/ab_bench.sh "RDM_CONV_TAIL=0" "" </dev/null
cp gpurun_out/ab_conv_tail.log $O/
timeout 300 python3 tools/op_trace.py --batch 64 --k 4 --steps 4 --out $O/op_trace_tail.csv > $O/op_trace_tail.log 2>&1 </dev/null
echo done
