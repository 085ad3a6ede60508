#!/bin/bash
R=$(pwd)
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
( timeout 300 python tools/offline_bench.py --model qwen3-0.6b --profile 200 ) > gpurun_out/c10_hostprofile_0.6b.log 2>&1
grep -v "^$" gpurun_out/c10_hostprofile_0.6b.log | head -75 | cut -c1-180
