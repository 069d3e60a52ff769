#!/bin/bash
# same-box comparison: the round-4 tree (git worktree of a241888 under tmp_ab/r04, its own library and bench.py) against the final tree
mkdir -p gpurun_out; cd $GRAFT_REPO_ROOT
OUT=gpurun_out/ab_r04_vs_r05.log; : > $OUT
for rep in 1 2; do
  for tree in tmp_ab/r04 .; do
    line=$(cd $tree && python3 bench.py --no-cpu-baseline --no-extras --steps 3 --warmup 1 2>/dev/null </dev/null | tail -1)
    echo "[$( [ "$tree" = "." ] && echo "final tree (round 5)" || echo "round-4 tree (a241888)" )] $(echo "$line" | python3 -c 'import sys,json; d=json.loads(sys.stdin.read()); print("%.3f img/s  %.1f ms/step" % (d["value"], d["ms_per_step"]))')" | tee -a $OUT
  done
done
