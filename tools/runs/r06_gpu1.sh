#!/bin/bash
# round 6, call 1: changed-path tests, the default bench line with the new calibration object, RARM stress (both hand-over forms),
# per-op table of config #4's geometry (UNet batch 64, k = 16, no CFG)
mkdir -p gpurun_out; cd $GRAFT_REPO_ROOT
O=gpurun_out/r06_1
mkdir -p $O
timeout 1500 python3 -m pytest tests/test_gpu_bench.py tests/test_gpu_rarm.py "tests/test_gpu_surface.py::test_deterministic_mode_rows_do_not_depend_on_the_batch" "tests/test_gpu_surface.py::test_deterministic_mode_rows_across_the_eight_wave_threshold_shipped_unet" -x -q 2>&1 </dev/null | tail -15 > $O/tests.log
timeout 600 python3 bench.py > $O/bench.json 2> $O/bench.err </dev/null
tail -3 $O/bench.err > $O/bench.err.tail
RDM_RARM_XSPLIT=1 timeout 400 python3 tools/rarm_stress.py 100000 24 0 0 240 > $O/stress_split.log 2>&1 </dev/null
RDM_RARM_XSPLIT=0 timeout 400 python3 tools/rarm_stress.py 100000 24 0 0 240 > $O/stress_oneblock.log 2>&1 </dev/null
RDM_RARM_XSPLIT=1 timeout 300 python3 tools/rarm_stress.py 100000 24 64 1 120 > $O/stress_split_heavy.log 2>&1 </dev/null
timeout 300 python3 tools/op_trace.py --batch 64 --k 16 --scale 1.0 --steps 4 --out $O/op_trace_config4.csv > $O/op_trace_config4.log 2>&1 </dev/null
timeout 300 python3 tools/op_trace.py --batch 64 --k 4 --steps 4 --out $O/op_trace_config3.csv > $O/op_trace_config3.log 2>&1 </dev/null
rocm-smi --showpower --showclocks > $O/smi.log 2>&1
ls /sys/class/drm/card*/device/hwmon/hwmon*/ > $O/hwmon.log 2>&1
echo done
