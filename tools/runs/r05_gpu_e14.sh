#!/bin/bash
# mgemm per-shape kernel time (rocprofv3 kernel trace of tools/lin_bench.py, one shape per process)
mkdir -p gpurun_out; R=$(pwd); cd /tmp; export TMPDIR=/tmp
: > $R/gpurun_out/e14_mgemm_shapes.log
for cfg in "64 3" "128 3" "64 2"; do
set -- $cfg
for shp in 2048,768,768,1 2048,2304,768 2048,768,3072,1 4096,768,768,1 4096,2304,768 4096,768,3072,1; do
  RDM_MGEMM_ANY=1024 RDM_MGEMM_BM=$1 RDM_MGEMM_NS=$2 timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/e14_p -o p -- python3 $R/tools/lin_bench.py $shp > /dev/null 2>&1 </dev/null
  f=$(find $R/gpurun_out/e14_p -name '*kernel_stats.csv' | head -1)
  [ -n "$f" ] && echo "BM=$1 NS=$2 $shp: $(grep mgemm_kernel "$f" </dev/null | cut -d, -f1-4 | tail -1)" >> $R/gpurun_out/e14_mgemm_shapes.log
  rm -rf $R/gpurun_out/e14_p
done
done
cat $R/gpurun_out/e14_mgemm_shapes.log
