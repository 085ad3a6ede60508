#!/bin/bash
R=$(pwd)
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
( time timeout 900 python -m pytest tests/test_gpu_gemm.py -m gpu -q -x -s -k skinny ) > gpurun_out/c14_pytest_gemm.log 2>&1
grep -E "passed|failed|down-proj|Error|error" gpurun_out/c14_pytest_gemm.log | head -12 | cut -c1-220
( timeout 600 python tools/gemm_sweep.py --batch 1 4 8 16 32 64 --mode heuristic --out gpurun_out/c14_gemm_sweep_small.json ) > gpurun_out/c14_gemm_sweep_small.log 2>&1
grep -E "^bs=|skinny:" gpurun_out/c14_gemm_sweep_small.log | cut -c1-170
tail -3 gpurun_out/c14_gemm_sweep_small.log | cut -c1-200
( time timeout 600 python -m pytest tests/test_gpu_model.py -m gpu -q -x -k config0 ) > gpurun_out/c14_pytest_cfg0.log 2>&1
tail -5 gpurun_out/c14_pytest_cfg0.log | cut -c1-220
for M in qwen3-14b; do
  ( time timeout 600 python tools/offline_bench.py --model $M --out gpurun_out/c14_offline_$M.json ) > gpurun_out/c14_offline_$M.log 2>&1
  grep '^{' gpurun_out/c14_offline_$M.log | cut -c1-700; tail -4 gpurun_out/c14_offline_$M.log | grep -v "^{" | cut -c1-200
done
