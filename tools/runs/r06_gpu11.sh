#!/bin/bash
# round 6, call 11: the whole GPU suite, smoke() and the default bench line on the final tree
mkdir -p gpurun_out; cd $GRAFT_REPO_ROOT
O=gpurun_out/r06_11; mkdir -p $O
timeout 1800 python3 -m pytest tests -m gpu -x -q 2>&1 </dev/null | grep -v amdgpu.ids | tail -8 > $O/tests.log
timeout 300 python3 -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1 </dev/null
timeout 600 python3 bench.py > $O/bench.json 2> $O/bench.err </dev/null
echo done
