#!/bin/bash
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 120 tools/build/cu_pipe_probe > gpurun_out/c3_cu_pipe_probe.txt 2>&1; tail -14 gpurun_out/c3_cu_pipe_probe.txt | cut -c1-200
( time timeout 600 python tools/step_ab.py --rounds 2 --knobs decode71 plan:o:256/0/3/1 plan:o:256/0/4/1 plan:o:256/0/8/1 plan:qkv:256/0/3/1 plan:qkv:256/0/2/1 plan:qkv:256/0/8/1 plan:down:256/0/3/1 plan:down:256/0/4/1 plan:down:256/0/8/1 plan:down:256/0/12/1 lib_o lib_qkv --out gpurun_out/c3_step_ab.json ) > gpurun_out/c3_step_ab.log 2>&1
tail -5 gpurun_out/c3_step_ab.log | cut -c1-900
( time timeout 900 python tools/trace_replay.py --model qwen3-32b --requests 200 --rate 6.0 --out gpurun_out/c3_trace_replay.json ) > gpurun_out/c3_trace.log 2>&1
tail -12 gpurun_out/c3_trace.log | cut -c1-600
( time timeout 1200 python -m pytest tests -m gpu -q -p no:cacheprovider ) > gpurun_out/c3_pytest.log 2>&1
tail -12 gpurun_out/c3_pytest.log | cut -c1-250
