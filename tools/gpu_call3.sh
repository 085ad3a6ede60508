#!/bin/bash
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
( time timeout 900 python -m pytest tests/test_gpu_attn_decode.py tests/test_gpu_model.py -x -q ) > gpurun_out/c3_pytest.log 2>&1
tail -15 gpurun_out/c3_pytest.log
( time timeout 300 python tools/microbench.py --only decode --out gpurun_out/c3_micro_decode.json ) > gpurun_out/c3_micro.log 2>&1
grep -v amdgpu.ids gpurun_out/c3_micro.log | cut -c1-330
( time timeout 900 python bench.py --steps 20 --warmup 3 --no-cpu-baseline ) > gpurun_out/c3_bench.log 2>&1
tail -3 gpurun_out/c3_bench.log | cut -c1-1500
