#!/bin/bash
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
( time timeout 900 python tools/offline_bench.py --model qwen3-14b --out gpurun_out/c21_offline_qwen3-14b.json ) > gpurun_out/c21_offline_14b.log 2>&1
grep '^{' gpurun_out/c21_offline_14b.log | cut -c1-600; grep real gpurun_out/c21_offline_14b.log
( time timeout 1500 python -m pytest tests -m gpu -q -x ) > gpurun_out/c21_pytest.log 2>&1
tail -5 gpurun_out/c21_pytest.log | cut -c1-220
