#!/bin/bash
# round 5: time-embedding table + staged conv_in stores: kernel timing per variant, same-box A/B of the headline
mkdir -p gpurun_out; cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
: > gpurun_out/e1_convin.log
for oct in 0 4 8; do
  RDM_CONVIN_OCT=$oct timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/e1_prof_oct$oct -o p -- python3 tools/op_trace.py > gpurun_out/e1_optrace_oct$oct.log 2>&1 </dev/null
  f=$(find gpurun_out/e1_prof_oct$oct -name '*kernel_stats.csv' | head -1)
  [ -n "$f" ] && echo "oct=$oct $(grep conv_in_kernel "$f" </dev/null)" >> gpurun_out/e1_convin.log
  rm -rf gpurun_out/e1_prof_oct$oct
done
timeout 900 tools/ab_multi.sh e1 "RDM_NO_EMB_TABLE=1 RDM_CONVIN_OCT=0" "RDM_CONVIN_OCT=0" "RDM_CONVIN_OCT=4" "" </dev/null
