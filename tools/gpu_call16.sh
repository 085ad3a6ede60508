#!/bin/bash
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
( timeout 600 python tools/gemm_sweep.py --batch 64 128 192 --mode heuristic --out gpurun_out/c16_gemm_sweep_mid.json ) > gpurun_out/c16_gemm_sweep_mid.log 2>&1
grep -E "^bs=|skinny:" gpurun_out/c16_gemm_sweep_mid.log | cut -c1-175
( timeout 600 python tools/offline_bench.py --model qwen3-14b --out gpurun_out/c16_offline_qwen3-14b.json ) > gpurun_out/c16_offline_14b.log 2>&1
grep '^{' gpurun_out/c16_offline_14b.log | cut -c1-600
