#!/bin/bash
# Round-1c GPU call: gen-2 prefill validation first, full GPU parity suite, bench, kernel-trace of the bench.
R=$(pwd)
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
( /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 tools/tr_probe.hip -o /tmp/tr_probe && timeout 60 /tmp/tr_probe ) > gpurun_out/c7_tr_probe.log 2>&1
head -20 gpurun_out/c7_tr_probe.log | cut -c1-100
( time timeout 400 python -m pytest tests/test_gpu_attn_prefill.py -m gpu -q ) > gpurun_out/c7_pytest_prefill.log 2>&1
tail -25 gpurun_out/c7_pytest_prefill.log | cut -c1-200
( time timeout 900 python -m pytest tests -m gpu -q --deselect tests/test_gpu_attn_prefill.py ) > gpurun_out/c7_pytest.log 2>&1
tail -6 gpurun_out/c7_pytest.log | cut -c1-200
( timeout 300 python tools/microbench.py --only prefill --out gpurun_out/c7_microbench_prefill.json ) > gpurun_out/c7_microbench_prefill.log 2>&1
grep "^prefill" gpurun_out/c7_microbench_prefill.log | cut -c1-400
( time timeout 700 python bench.py ) > gpurun_out/c7_bench.log 2>&1
grep '^{"metric' gpurun_out/c7_bench.log > gpurun_out/c7_bench.json
python - <<'P'
import json
try:
    d=json.load(open("gpurun_out/c7_bench.json"))
    for k in ("value","ms_per_step","ttft_p50_ms","roofline","step_roofline","prefill_roofline","cpu_baseline"): print(k, d.get(k))
except Exception as e: print("no bench json", e)
P
tail -5 gpurun_out/c7_bench.log | cut -c1-300
cd /tmp && export TMPDIR=/tmp
MSGL_GEMM_TUNE=heuristic timeout 500 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/c7_kt -- python $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline > $R/gpurun_out/c7_kt.log 2>&1
DB=$(find $R/gpurun_out/c7_kt -name "*results.db" | head -1)
python $R/tools/rocpd_summary.py $DB --top 40 > $R/gpurun_out/c7_kt.txt 2>&1; cut -c1-170 $R/gpurun_out/c7_kt.txt | head -45
find $R/gpurun_out/c7_kt -name "*.db" -delete
