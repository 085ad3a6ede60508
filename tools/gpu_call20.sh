#!/bin/bash
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
( time timeout 900 python -m pytest tests/test_gpu_gemm.py -m gpu -q -x -k wstream ) > gpurun_out/c20_pytest_gemm.log 2>&1
grep -E "passed|failed|Error|error|assert" gpurun_out/c20_pytest_gemm.log | head -12 | cut -c1-220
( timeout 900 python tools/gemm_sweep.py --batch 40 64 128 192 256 --mode heuristic --out gpurun_out/c20_gemm_sweep.json ) > gpurun_out/c20_gemm_sweep.log 2>&1
grep -E "^bs=|wstream:" gpurun_out/c20_gemm_sweep.log | cut -c1-200
