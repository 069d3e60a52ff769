#!/bin/bash
# round 6, call 21: fused feed-forward with hand-pipelined LDS fragment reads (asm reads, counted waits tied to the fragments): parity + timing
mkdir -p gpurun_out; cd $GRAFT_REPO_ROOT
O=gpurun_out/r06_21; mkdir -p $O
timeout 600 python3 -m pytest tests/test_gpu_ops.py -x -q -s -k "ffn_fused" 2>&1 </dev/null | grep -v amdgpu.ids | tail -4 > $O/test.log
for s in 4 6; do
  echo "RDM_FFN_SKS=$s" >> $O/ffn_bench.log
  RDM_FFN_SKS=$s timeout 600 python3 tools/ffn_bench.py 2>&1 </dev/null | grep -v amdgpu.ids >> $O/ffn_bench.log
done
echo done
