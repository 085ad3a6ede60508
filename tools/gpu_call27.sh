#!/bin/bash
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
( time timeout 150 python -m pytest tests/test_gpu_comm.py -m gpu -q -x ) > gpurun_out/c27_pytest_comm.log 2>&1
tail -25 gpurun_out/c27_pytest_comm.log | cut -c1-220
