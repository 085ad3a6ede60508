#!/bin/bash
R=$(pwd)
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
( time timeout 900 python -m pytest tests/test_gpu_sampling.py tests/test_gpu_model.py tests/test_gpu_attn_decode.py -m gpu -q -x ) > gpurun_out/c12_pytest.log 2>&1
tail -15 gpurun_out/c12_pytest.log | cut -c1-220
( timeout 600 python tools/gemm_sweep.py --batch 1 8 32 64 128 --mode heuristic --out gpurun_out/c12_gemm_sweep_small.json ) > gpurun_out/c12_gemm_sweep_small.log 2>&1
grep "^bs=" gpurun_out/c12_gemm_sweep_small.log | cut -c1-200
( time timeout 700 python bench.py ) > gpurun_out/c12_bench.log 2>&1
grep '^{"metric' gpurun_out/c12_bench.log > gpurun_out/c12_bench.json
python - <<'P'
import json
try:
    d=json.load(open("gpurun_out/c12_bench.json"))
    for k in ("value","ms_per_step","ttft_p50_ms","roofline","step_roofline"): print(k, d.get(k))
except Exception as e: print("no bench json", e)
P
tail -3 gpurun_out/c12_bench.log | cut -c1-300
( timeout 600 python tools/offline_bench.py --model qwen3-0.6b --out gpurun_out/c12_offline_qwen3-0.6b.json ) > gpurun_out/c12_offline_0.6b.log 2>&1
grep '^{' gpurun_out/c12_offline_0.6b.log | cut -c1-500
