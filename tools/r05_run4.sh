#!/bin/bash
mkdir -p gpurun_out
( time timeout 300 python tools/ro_ablate.py --out gpurun_out/r05d_ro_ablate.json ) 2>&1 | grep -v "^$" | tail -14 | cut -c1-260
( time timeout 300 python tools/ro_bench.py --shapes lm_head --batches 256 128 64 16 --out gpurun_out/r05d_ro_bench_lm_head.json ) 2>&1 | grep -v "^parity" | tail -8 | cut -c1-400
( time timeout 300 python tools/ro_bench.py --shapes qkv o gate_up down --batches 8 4 --no-library --out gpurun_out/r05d_ro_bench_small.json ) 2>&1 | grep -v "^parity" | tail -12 | cut -c1-300
