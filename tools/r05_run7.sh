#!/bin/bash
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
( time timeout 600 python bench.py --steps 20 --warmup 5 --no-e2e --no-cpu-baseline --no-prefill-roofline --small-batches 1 2 4 8 16 32 64 96 128 192 ) > gpurun_out/r05g_bench_small.json 2> gpurun_out/r05g_bench_small.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r05g_bench_small.json").read().strip().splitlines()[-1])
print(json.dumps(d["summary"]))
for r in d["gemm_tune"]["shapes"]:
    if r["M"] in (8, 4, 2, 16):
        print(r["M"], r["name"], r["tuned_us"], r.get("hand_written", "library")[:60], r.get("library_best_us"))
PY
bash tools/trace_small_batch.sh r05g 128 2>&1 | tail -24
bash tools/trace_small_batch.sh r05g 8 2>&1 | tail -24
