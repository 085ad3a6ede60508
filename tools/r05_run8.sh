#!/bin/bash
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
( time MSGL_BENCH_SHARE_GPU=1 MSGL_GEMM_TUNE=off timeout 700 python bench.py --gpus 2 --steps 3 --warmup 1 ) > gpurun_out/r05_bench_tp2_ranks_on_one_gpu_code_path_check.json 2> gpurun_out/r05_bench_tp2_share.err
python - <<'PY'
import json
f = "gpurun_out/r05_bench_tp2_ranks_on_one_gpu_code_path_check.json"
try:
    d = json.loads(open(f).read().strip().splitlines()[-1])
    print("self-launched --gpus 2 on one GPU:", d["launch"], "| ms/step", round(d["ms_per_step"], 1), "|", d["collectives"]["paths"], "| rccl ranks", d["collectives"].get("rccl_ranks_seen"))
except Exception as e:
    print("self-launch check unreadable:", e)
PY
tail -3 gpurun_out/r05_bench_tp2_share.err | cut -c1-200
( time timeout 300 python -m pytest tests/test_gpu_gemm.py -q -k "travel" ) 2>&1 | tail -4
