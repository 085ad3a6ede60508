#!/bin/bash
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
( time timeout 170 python -m pytest tests/test_gpu_model.py -m gpu -q -x -k tp_code_path ) > gpurun_out/c28_pytest_tp.log 2>&1
tail -30 gpurun_out/c28_pytest_tp.log | cut -c1-220
