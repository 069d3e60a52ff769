#!/bin/bash
# round 6, call 6: side stream at normal priority: conv tail split and skip overlap A/B
mkdir -p gpurun_out; cd $GRAFT_REPO_ROOT
O=gpurun_out/r06_6; mkdir -p $O
AB_TAG=tail_normprio timeout 900 tools/ab_bench.sh "RDM_SIDE_LOWPRIO=0" "RDM_SIDE_LOWPRIO=0 RDM_CONV_TAIL=1" </dev/null
AB_TAG=skip_prio timeout 900 tools/ab_bench.sh "" "RDM_SIDE_LOWPRIO=0" </dev/null
cp gpurun_out/ab_tail_normprio.log gpurun_out/ab_skip_prio.log $O/
RDM_SIDE_LOWPRIO=0 RDM_CONV_TAIL=1 timeout 300 python3 tools/op_trace.py --batch 64 --k 4 --steps 4 --out $O/op_trace_tail.csv > $O/op_trace_tail.log 2>&1 </dev/null
echo done
