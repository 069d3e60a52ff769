#!/bin/bash
# round 6, call 15: the GPU suite twice on the final tree (whole logs kept) + smoke
mkdir -p gpurun_out; cd $GRAFT_REPO_ROOT
O=gpurun_out/r06_15; mkdir -p $O
for i in 1 2; do
  timeout 1500 python3 -m pytest tests -x -q -m gpu > $O/tests_run$i.log 2>&1 </dev/null
  echo "run $i exit code $?" >> $O/tests_run$i.log
  grep -v amdgpu.ids $O/tests_run$i.log | grep "passed\|failed\|exit code\|Fatal\|Error" | tail -6 > $O/tests_run${i}_tail.log
done
timeout 300 python3 -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1 </dev/null
echo done
