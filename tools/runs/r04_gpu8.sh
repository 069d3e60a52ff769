#!/bin/bash
# TN wgrad kernel: parity tests, then the shape bench new vs old
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_backward.py -x -q -s 2>&1 | grep -v "^$" | tail -40
timeout 600 python tools/wgrad_bench.py > gpurun_out/wgrad_new.log 2>&1; tail -22 gpurun_out/wgrad_new.log
RDM_NO_WGRAD_TN=1 timeout 600 python tools/wgrad_bench.py > gpurun_out/wgrad_old.log 2>&1; tail -22 gpurun_out/wgrad_old.log
