#!/bin/bash
# round 6, call 13: the GPU suite twice more exactly as the driver runs it (call 11 dumped core once, call 12 passed), whole logs kept; RARM split on / off at 64 and 128 sequences
mkdir -p gpurun_out; cd $GRAFT_REPO_ROOT
O=gpurun_out/r06_13; mkdir -p $O
for i in 1 2; do
  timeout 1500 python3 -m pytest tests -x -q -m gpu > $O/tests_run$i.log 2>&1 </dev/null
  echo "run $i exit code $?" >> $O/tests_run$i.log
  tail -5 $O/tests_run$i.log | grep -v amdgpu.ids > $O/tests_run${i}_tail.log
done
for b in 64 128; do
  for v in 0 1 0 1; do
    RDM_RARM_XSPLIT=$v timeout 300 python3 bench.py --config 5 --batch $b --db-rows 2000000 --steps 3 --warmup 1 --no-cpu-baseline --no-extras --no-calibration 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('split=$v batch=$b', round(d['value'],2), 'img/s')" >> $O/rarm_split_ab.log
  done
done
echo done
