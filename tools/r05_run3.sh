#!/bin/bash
# round 5, third GPU call: rank-shard TP bounds with the row-owner kernel in the search; kernel trace of the timed steps
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
R=$(pwd)
MS=17.66
for cfg in "qwen3-14b 4" "qwen3-32b 4" "qwen3-14b 2" "qwen3-14b 8" "llama-3.1-70b 8"; do set -- $cfg; timeout 240 python bench.py --model $1 --rank-shard $2 --tp1-ms $MS --steps 20 --warmup 5 2>gpurun_out/r05c_rank_shard_$1_tp$2.err | tail -1 > gpurun_out/r05c_rank_shard_$1_tp$2.json; python - <<PY
import json
try:
    d = json.loads(open("gpurun_out/r05c_rank_shard_$1_tp$2.json").read())
    print("$1 tp$2", round(d["ms_per_step"], 3), "ms", {k: (v["us"], v["kernel"][:44]) for k, v in d.get("gemm_plans_at_full_batch", {}).items()})
except Exception as e:
    print("$1 tp$2 unreadable", e)
PY
done
cd /tmp && export TMPDIR=/tmp
timeout 500 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r05c_kt -- python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-e2e --no-prefill-roofline --small-batches > $R/gpurun_out/r05c_kt.log 2>&1
DB=$(find $R/gpurun_out/r05c_kt -name "*results.db" | head -1)
timeout 120 python $R/tools/rocpd_summary.py $DB --top 30 --steps-by sample --last-steps 20 > $R/gpurun_out/r05c_bench_timed_steps_kernel_breakdown.txt 2>&1
head -24 $R/gpurun_out/r05c_bench_timed_steps_kernel_breakdown.txt | cut -c1-170
find $R/gpurun_out/r05c_kt -name "*.db" -delete
