#!/bin/bash
# round 6, call 10: the round's evidence tree on the final kernels (tools/profile_round.sh r06: bench line, kernel stats, SQ / FETCH / WRITE passes, RARM at 64 and 2048 sequences, configs 2 and 4, per-op table)
mkdir -p gpurun_out; cd $GRAFT_REPO_ROOT
RARM_BATCHES="64 2048" timeout 3000 bash tools/profile_round.sh r06 > gpurun_out/profile_round_r06.log 2>&1 </dev/null
echo done
