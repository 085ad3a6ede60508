#!/bin/bash
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
( timeout 100 python tools/offline_bench.py --model qwen3-14b --out gpurun_out/c33_offline_qwen3-14b.json ) > gpurun_out/c33_offline_14b.log 2>&1
grep '^{' gpurun_out/c33_offline_14b.log | cut -c1-500
