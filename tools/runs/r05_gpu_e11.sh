#!/bin/bash
# round 5: sgemm to 4096 rows, default 2048 sequences: RARM suite, bench lines at 1024 / 2048 / 4096, kernel stats + SQ pass at 2048
mkdir -p gpurun_out; cd $GRAFT_REPO_ROOT
timeout 900 python3 -m pytest tests/test_gpu_rarm.py -x -q 2>&1 </dev/null | tail -4 > gpurun_out/e11_tests.log
for b in 64 256 512 1024 2048 4096; do timeout 300 python3 bench.py --config 5 --batch $b --db-rows 2000000 --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null </dev/null | tail -1 > gpurun_out/c5_b$b.json; done
timeout 600 bash tools/rarm_profile.sh e11 2048 </dev/null > gpurun_out/e11_profile.log 2>&1
R=$(pwd); cd /tmp; export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --output-format csv --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM GRBM_GUI_ACTIVE -d $R/gpurun_out/e11_rsq -- python3 $R/bench.py --config 5 --db-rows 2000000 --ddim-steps 32 --steps 1 --warmup 0 --no-cpu-baseline --no-extras > $R/gpurun_out/e11_rsq.log 2>&1 </dev/null
python3 $R/tools/pmc_sum.py counters $R/gpurun_out/e11_rarm_b2048_sq.csv $R/gpurun_out/e11_rsq
rm -rf $R/gpurun_out/e11_rsq
