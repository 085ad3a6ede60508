#!/bin/bash
# full GPU parity suite, bench (default command), rocprofv3 --kernel-trace --stats of the same command
R=$(pwd)
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
( time timeout 1500 python -m pytest tests -m gpu -q -x ) > gpurun_out/c15_pytest.log 2>&1
tail -6 gpurun_out/c15_pytest.log | cut -c1-220
( time timeout 700 python bench.py ) > gpurun_out/c15_bench.log 2>&1
grep '^{"metric' gpurun_out/c15_bench.log > gpurun_out/c15_bench.json
python - <<'P'
import json
try:
    d=json.load(open("gpurun_out/c15_bench.json"))
    for k in ("value","ms_per_step","ttft_p50_ms","small_batch_ms_per_step","roofline","step_roofline","cpu_baseline"): print(k, d.get(k))
    for r in d["gemm_tune"]["shapes"]: print(r)
except Exception as e: print("no bench json", e)
P
tail -4 gpurun_out/c15_bench.log | cut -c1-300
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/c15_kt -- python $R/bench.py --no-cpu-baseline > $R/gpurun_out/c15_kt.log 2>&1
DB=$(find $R/gpurun_out/c15_kt -name "*results.db" | head -1)
python $R/tools/rocpd_summary.py $DB --top 40 > $R/gpurun_out/c15_kt_stats.txt 2>&1
python $R/tools/rocpd_summary.py $DB --top 30 --steps-by sample_logits_kernel --last-steps 10 > $R/gpurun_out/c15_kt_tail_steps.txt 2>&1
head -14 $R/gpurun_out/c15_kt_stats.txt | cut -c1-170
find $R/gpurun_out/c15_kt -name "*.db" -delete
