#!/bin/bash
# round-1 session-2 GPU call 1: parity after the decode-attention restructure + GEMM wrapper, then measurements
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
( time timeout 900 python -m pytest tests -m gpu -x -q ) > gpurun_out/c1_pytest.log 2>&1
echo "pytest rc=$?" >> gpurun_out/c1_pytest.log
tail -5 gpurun_out/c1_pytest.log
( time timeout 300 python tools/microbench.py --only decode --out gpurun_out/c1_micro_decode.json ) > gpurun_out/c1_micro.log 2>&1
tail -12 gpurun_out/c1_micro.log
( time timeout 600 python tools/gemm_sweep.py --batch 256 --mode full --out gpurun_out/c1_gemm_sweep.json ) > gpurun_out/c1_gemm.log 2>&1
tail -14 gpurun_out/c1_gemm.log
( time LD_PRELOAD=/opt/rocm/lib/libhipblaslt.so.1 timeout 600 python tools/gemm_sweep.py --batch 256 --mode full --out gpurun_out/c1_gemm_sweep_rocm72.json ) > gpurun_out/c1_gemm72.log 2>&1
tail -14 gpurun_out/c1_gemm72.log
( time timeout 900 python bench.py --steps 20 --warmup 3 --no-cpu-baseline ) > gpurun_out/c1_bench.log 2>&1
tail -3 gpurun_out/c1_bench.log
