#!/bin/bash
# Round-1c GPU call: tr-read probe, GPU parity suite, micro-benchmarks (prefill A/B), bench, PMC + kernel trace.
R=$(pwd)
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
( /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 tools/tr_probe.hip -o /tmp/tr_probe && /tmp/tr_probe ) > gpurun_out/c6_tr_probe.log 2>&1
head -40 gpurun_out/c6_tr_probe.log | cut -c1-100
( time timeout 1500 python -m pytest tests -m gpu -q -x --deselect tests/test_gpu_attn_prefill.py ) > gpurun_out/c6_pytest.log 2>&1
tail -4 gpurun_out/c6_pytest.log
( time timeout 600 python -m pytest tests/test_gpu_attn_prefill.py -m gpu -q ) > gpurun_out/c6_pytest_prefill.log 2>&1
tail -25 gpurun_out/c6_pytest_prefill.log | cut -c1-200
( MSGL_TR_VARIANT=1 timeout 600 python -m pytest tests/test_gpu_attn_prefill.py -m gpu -q -k "tr_read or generations" ) > gpurun_out/c6_pytest_prefill_var1.log 2>&1
tail -3 gpurun_out/c6_pytest_prefill_var1.log | cut -c1-200
( timeout 600 python tools/microbench.py --only prefill --out gpurun_out/c6_microbench_prefill.json ) > gpurun_out/c6_microbench_prefill.log 2>&1
grep "^prefill" gpurun_out/c6_microbench_prefill.log | cut -c1-600
( time timeout 1200 python bench.py ) > gpurun_out/c6_bench.log 2>&1
grep '^{"metric' gpurun_out/c6_bench.log > gpurun_out/c6_bench.json
python - <<'P'
import json
d=json.load(open("gpurun_out/c6_bench.json"))
for k in ("value","ms_per_step","ttft_p50_ms","roofline","step_roofline","prefill_roofline","cpu_baseline"): print(k, d.get(k))
P
tail -5 gpurun_out/c6_bench.log | cut -c1-300
