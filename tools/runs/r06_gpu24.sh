#!/bin/bash
# round 6, call 24: the GPU suite, smoke and the default bench line on the FINAL default path (no side stream, one block per sequence in the RARM cross-attention)
mkdir -p gpurun_out; cd $GRAFT_REPO_ROOT
O=gpurun_out/r06_24; mkdir -p $O
timeout 1500 python3 -m pytest tests -x -q -m gpu > $O/tests_run1.log 2>&1 </dev/null
echo "run 1 exit code $?" >> $O/tests_run1.log
grep -v amdgpu.ids $O/tests_run1.log | grep "passed\|failed\|exit code\|Fatal\|Error" | tail -6 > $O/tests_run1_tail.log
timeout 300 python3 -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1 </dev/null
timeout 600 python3 bench.py > $O/bench.json 2> $O/bench.err </dev/null
echo done
