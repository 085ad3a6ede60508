#!/bin/bash
# round 5, first GPU call: the new row-owner kernel (parity + bench), world-8 / root-cause p2p tests, mixed-plan fold test
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
( time timeout 500 python -m pytest tests/test_gpu_ro.py -x -q ) > gpurun_out/r05a_test_ro.log 2>&1; tail -15 gpurun_out/r05a_test_ro.log | cut -c1-300
( time timeout 400 python tools/ro_bench.py --batches 256 128 64 16 --out gpurun_out/r05a_ro_bench.json ) > gpurun_out/r05a_ro_bench.log 2>&1; grep -v "^parity" gpurun_out/r05a_ro_bench.log | tail -22 | cut -c1-330
( time timeout 700 python -m pytest tests/test_gpu_tp.py -q -k "known_answers or root_cause or timeout" ) > gpurun_out/r05a_test_tp.log 2>&1; tail -12 gpurun_out/r05a_test_tp.log | cut -c1-300
( time timeout 300 python -m pytest tests/test_gpu_small_batch.py -q -k mixed ) > gpurun_out/r05a_test_small.log 2>&1; tail -6 gpurun_out/r05a_test_small.log | cut -c1-300
