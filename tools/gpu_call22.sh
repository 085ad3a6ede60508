#!/bin/bash
# final round-1 artefacts: bench (default command), rocprofv3 --kernel-trace --stats of the same command, smoke
R=$(pwd)
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
( time timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" ) > gpurun_out/c22_smoke.log 2>&1
tail -3 gpurun_out/c22_smoke.log | cut -c1-200
( time timeout 700 python bench.py ) > gpurun_out/c22_bench.log 2>&1
grep '^{"metric' gpurun_out/c22_bench.log > gpurun_out/c22_bench.json
python - <<'P'
import json
try:
    d=json.load(open("gpurun_out/c22_bench.json"))
    for k in ("value","ms_per_step","ttft_p50_ms","small_batch_ms_per_step","roofline","step_roofline","prefill_roofline","cpu_baseline"): print(k, d.get(k))
    for r in d["gemm_tune"]["shapes"]:
        if r["M"] == 256: print(r)
except Exception as e: print("no bench json", e)
P
tail -4 gpurun_out/c22_bench.log | cut -c1-300
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/c22_kt -- python $R/bench.py --no-cpu-baseline > $R/gpurun_out/c22_kt.log 2>&1
DB=$(find $R/gpurun_out/c22_kt -name "*results.db" | head -1)
python $R/tools/rocpd_summary.py $DB --top 40 > $R/gpurun_out/c22_kt_stats.txt 2>&1
head -8 $R/gpurun_out/c22_kt_stats.txt | cut -c1-170
find $R/gpurun_out/c22_kt -name "*.db" -delete
