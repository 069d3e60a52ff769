#!/bin/bash
# round 5: conv_in tap loop re-cut (constexpr taps, op_sel-broadcast packed FMAs): parity subset + kernel timing per store variant
mkdir -p gpurun_out; cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 600 python3 -m pytest tests/test_gpu_models.py tests/test_gpu_emul.py -x -q 2>&1 </dev/null | tail -5 > gpurun_out/e2_tests.log
: > gpurun_out/e2_convin.log
for oct in 0 4; do
  RDM_CONVIN_OCT=$oct timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/e2_prof_oct$oct -o p -- python3 tools/op_trace.py > gpurun_out/e2_optrace_oct$oct.log 2>&1 </dev/null
  f=$(find gpurun_out/e2_prof_oct$oct -name '*kernel_stats.csv' | head -1)
  [ -n "$f" ] && echo "oct=$oct $(grep conv_in_kernel "$f" </dev/null)" >> gpurun_out/e2_convin.log
  rm -rf gpurun_out/e2_prof_oct$oct
done
timeout 600 tools/ab_multi.sh e2 "RDM_CONVIN_OCT=0" "" </dev/null
