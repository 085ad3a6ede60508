#!/bin/bash
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
( time timeout 1000 python -m pytest tests/test_gpu_reference_driven.py -x -q -s -k "tuned_plans" ) > gpurun_out/r05e_test_tuned_refdrive.log 2>&1; tail -30 gpurun_out/r05e_test_tuned_refdrive.log | cut -c1-400
