#!/bin/bash
# round 6, call 12: the GPU suite again with the whole log kept (call 11 died with a core dump and only its tail was saved)
mkdir -p gpurun_out; cd $GRAFT_REPO_ROOT
O=gpurun_out/r06_12; mkdir -p $O
timeout 1800 python3 -X faulthandler -m pytest tests -m gpu -x -v 2>&1 </dev/null | grep -v amdgpu.ids > $O/tests_full.log
tail -60 $O/tests_full.log > $O/tests_tail.log
echo done
