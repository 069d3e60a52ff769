#!/bin/bash
# round 6, call 2: Winograd K-loop probe (operands in LDS: ingest free), kernel trace of config #4 (k = 16, no CFG, UNet batch 64),
# the RARM stress continued to > 20 000 repeats per form, the default bench line on another box (calibration check)
mkdir -p gpurun_out; cd $GRAFT_REPO_ROOT
REPO=$(pwd); O=$REPO/gpurun_out/r06_2; mkdir -p $O
timeout 300 tools/ubench/wino_loop 20000 > $O/wino_loop.log 2>&1 </dev/null
timeout 600 python3 bench.py > $O/bench.json 2> $O/bench.err </dev/null
tail -3 $O/bench.err > $O/bench.err.tail
( cd /tmp && export TMPDIR=/tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats4 -- python3 $REPO/bench.py --config 4 --ddim-steps 20 --steps 1 --warmup 1 --no-cpu-baseline --no-extras --no-calibration > $O/stats4.log 2>&1 </dev/null )
python3 tools/pmc_sum.py stats $O/config4_kernel_stats.csv $O/stats4 > /dev/null 2>&1
rm -rf $O/stats4
RDM_RARM_XSPLIT=1 timeout 500 python3 tools/rarm_stress.py 11500 24 0 0 400 > $O/stress_split.log 2>&1 </dev/null
RDM_RARM_XSPLIT=0 timeout 500 python3 tools/rarm_stress.py 11500 24 0 0 400 > $O/stress_oneblock.log 2>&1 </dev/null
echo done
