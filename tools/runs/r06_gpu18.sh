#!/bin/bash
# round 6, call 18: fused feed-forward kernel: where the time goes (ablation bits, wrong results)
mkdir -p gpurun_out; cd $GRAFT_REPO_ROOT
O=gpurun_out/r06_18; mkdir -p $O
timeout 600 python3 -m pytest tests/test_gpu_ops.py -x -q -s -k "ffn_fused" 2>&1 </dev/null | grep -v amdgpu.ids | tail -5 > $O/test.log
for dbgv in 0 1 2 4 6 8 9 15; do
  echo "RDM_FFN_DBG=$dbgv" >> $O/ffn_ablate.log
  RDM_FFN_SKS=6 RDM_FFN_DBG=$dbgv timeout 300 python3 tools/ffn_bench.py 2>&1 </dev/null | grep "^M = " | sed 's/ | GEGLU.*//' >> $O/ffn_ablate.log
done
echo done
