#!/bin/bash
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
( time timeout 300 python -m pytest tests/test_gpu_gemm.py -x -q ) > gpurun_out/c2_pytest.log 2>&1
tail -3 gpurun_out/c2_pytest.log
( time timeout 300 python tools/microbench.py --only decode --out gpurun_out/c2_micro_decode.json ) > gpurun_out/c2_micro.log 2>&1
grep -v amdgpu.ids gpurun_out/c2_micro.log | cut -c1-420
( time timeout 300 python tools/microbench.py --only decode --out gpurun_out/c2_micro_decode_b.json ) > gpurun_out/c2_micro_b.log 2>&1
grep "p1\|tp8" gpurun_out/c2_micro_b.log | cut -c1-420
( time timeout 600 python tools/gemm_sweep.py --batch 256 --mode full --out gpurun_out/c2_gemm_sweep.json ) > gpurun_out/c2_gemm.log 2>&1
grep -v amdgpu.ids gpurun_out/c2_gemm.log
