#!/bin/bash
# round 6, call 7: in-kernel tail split (conv_halo4 TAIL variant): parity, A/B, per-op table
mkdir -p gpurun_out; cd $GRAFT_REPO_ROOT
O=gpurun_out/r06_7; mkdir -p $O
timeout 1500 python3 -m pytest tests/test_gpu_full.py tests/test_gpu_emul.py -x -q -s -k "tail_split or batch64 or ddim_50 or emul" 2>&1 </dev/null | grep -v amdgpu.ids | tail -25 > $O/tests.log
AB_TAG=tailk timeout 900 tools/ab_bench.sh "RDM_CONV_TAILK=0" "" </dev/null
cp gpurun_out/ab_tailk.log $O/
timeout 300 python3 tools/op_trace.py --batch 64 --k 4 --steps 4 --out $O/op_trace_tailk.csv > $O/op_trace_tailk.log 2>&1 </dev/null
echo done
