#!/bin/bash
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
( time timeout 1500 python -m pytest tests -m gpu -q -x ) > gpurun_out/r05f_pytest.log 2>&1
tail -25 gpurun_out/r05f_pytest.log | cut -c1-300
