#!/bin/bash
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
( time timeout 700 python bench.py --no-cpu-baseline ) > gpurun_out/c17_bench.log 2>&1
grep '^{"metric' gpurun_out/c17_bench.log > gpurun_out/c17_bench.json
python - <<'P'
import json
try:
    d=json.load(open("gpurun_out/c17_bench.json"))
    for k in ("value","ms_per_step","ttft_p50_ms","small_batch_ms_per_step","prefill_gemm_tune"): print(k, d.get(k))
except Exception as e: print("no bench json", e)
P
tail -4 gpurun_out/c17_bench.log | cut -c1-300
( MSGL_PREFILL_TUNE=0 timeout 700 python bench.py --no-cpu-baseline --no-prefill-roofline --small-batches ) > gpurun_out/c17_bench_notune.log 2>&1
grep '^{"metric' gpurun_out/c17_bench_notune.log | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('no prefill tune: ttft', d['ttft_p50_ms'], 'value', d['value'])"
