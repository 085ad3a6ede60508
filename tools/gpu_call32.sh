#!/bin/bash
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
( timeout 100 python -m pytest tests/test_gpu_gemm.py -m gpu -q -x -k wstream ) > gpurun_out/c32_pytest_gemm.log 2>&1
grep -E "passed|failed|Error|assert" gpurun_out/c32_pytest_gemm.log | head -5 | cut -c1-200
( timeout 200 python tools/gemm_sweep.py --batch 64 128 256 --mode heuristic --out gpurun_out/c32_gemm_sweep.json ) > gpurun_out/c32_gemm_sweep.log 2>&1
grep -E "GEMM time|wstream:" gpurun_out/c32_gemm_sweep.log | cut -c1-120
