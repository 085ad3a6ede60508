#!/bin/bash
# final numbers of round 4 on the final code: changed tests, bench, kernel trace of the same command, PMC passes of the dominant
# kernel, prefill variants, the self-launched --gpus 4 code-path check (Qwen3-14B + the Qwen3-32B entry, 4 ranks on one GPU)
TAG=r04
R=$(pwd)
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
( time timeout 600 python -m pytest -q -p no:cacheprovider tests/test_gpu_attn_decode.py tests/test_gpu_gemm.py ) > gpurun_out/${TAG}_pytest_changed.log 2>&1; tail -3 gpurun_out/${TAG}_pytest_changed.log | head -1
( time timeout 600 python bench.py --steps 20 --warmup 5 ) > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err
python - <<PY
import json
try:
    d = json.loads(open("gpurun_out/${TAG}_bench.json").read().strip().splitlines()[-1])
    print("bench:", round(d["value"]), "tok/s", round(d["ms_per_step"], 2), "ms/step; attn", round(d["roofline"]["us_per_launch"], 1), "us frac",
          round(d["roofline"]["frac"], 3), "; step frac", round(d["step_roofline"]["frac"], 3), "; cpu", d.get("cpu_baseline", {}).get("value"), d.get("cpu_baseline", {}).get("cores"),
          "; e2e", {k: round(v.get("throughput_tok_s", 0)) for k, v in d.get("e2e_offline", {}).items()})
except Exception as e:
    print("bench line unreadable:", e)
PY
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/${TAG}_kt -- python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-e2e --no-prefill-roofline --small-batches > $R/gpurun_out/${TAG}_kt.log 2>&1
DB=$(find $R/gpurun_out/${TAG}_kt -name "*results.db" | head -1)
timeout 120 python $R/tools/rocpd_summary.py $DB --top 30 > $R/gpurun_out/${TAG}_bench_kernel_stats.txt 2>&1
timeout 120 python $R/tools/rocpd_summary.py $DB --top 30 --steps-by sample --last-steps 20 > $R/gpurun_out/${TAG}_bench_timed_steps_kernel_breakdown.txt 2>&1
head -12 $R/gpurun_out/${TAG}_bench_timed_steps_kernel_breakdown.txt | cut -c1-170
find $R/gpurun_out/${TAG}_kt -name "*.db" -delete
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --pmc $C -d $R/gpurun_out/${TAG}_pmc_$C -- python $R/tools/profile_attn.py --advance 26 > $R/gpurun_out/${TAG}_pmc_$C.log 2>&1
  DB=$(find $R/gpurun_out/${TAG}_pmc_$C -name "*results.db" | head -1)
  timeout 120 python $R/tools/rocpd_summary.py $DB --top 12 > $R/gpurun_out/${TAG}_pmc_$C.txt 2>&1
  find $R/gpurun_out/${TAG}_pmc_$C -name "*.db" -delete
done
timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/${TAG}_attn_kt -- python $R/tools/profile_attn.py --advance 26 > $R/gpurun_out/${TAG}_attn_kt.log 2>&1
DB=$(find $R/gpurun_out/${TAG}_attn_kt -name "*results.db" | head -1)
timeout 120 python $R/tools/rocpd_summary.py $DB --top 8 > $R/gpurun_out/${TAG}_attn_decode_kernel_trace.txt 2>&1; cut -c1-160 $R/gpurun_out/${TAG}_attn_decode_kernel_trace.txt | head -6
find $R/gpurun_out/${TAG}_attn_kt -name "*.db" -delete
ALGO=$(grep algorithmic_bytes_per_launch $R/gpurun_out/${TAG}_attn_kt.log | awk '{print $2}')
python $R/tools/pmc_json.py $R/gpurun_out/${TAG}_pmc_FETCH_SIZE.txt $R/gpurun_out/${TAG}_pmc_WRITE_SIZE.txt \
  $R/gpurun_out/${TAG}_attn_decode_kernel_trace.txt $ALGO $R/gpurun_out/${TAG}_pmc_attn_decode.json "tools/c5.sh (final code of round 4)" | cut -c1-300
cd $R
timeout 200 python tools/prefill_ablate.py --rounds 3 --only "tr: full" "dma: full" --out gpurun_out/${TAG}_prefill_variants.json > gpurun_out/${TAG}_prefill_variants.log 2>&1; tail -3 gpurun_out/${TAG}_prefill_variants.log | cut -c1-300
timeout 200 python tools/decode_ab.py --shape 14b,14b_tp4,70b_tp8 --impls 0,72 --out gpurun_out/${TAG}_decode_ab_final.json 2>&1 | grep impl > gpurun_out/${TAG}_decode_ab_final.txt; cat gpurun_out/${TAG}_decode_ab_final.txt | cut -c1-120
( time MSGL_BENCH_SHARE_GPU=1 MSGL_GEMM_TUNE=off timeout 900 python bench.py --gpus 4 --steps 2 --warmup 1 ) > gpurun_out/${TAG}_bench_tp4_ranks_on_one_gpu_code_path_check.json 2> gpurun_out/${TAG}_bench_tp4_share.err
python - <<PY
import json
try:
    d = json.loads(open("gpurun_out/${TAG}_bench_tp4_ranks_on_one_gpu_code_path_check.json").read().strip().splitlines()[-1])
    print("self-launched --gpus 4 on one GPU:", d["launch"], "| ms/step", round(d["ms_per_step"], 1), "|", d["collectives"]["paths"],
          "| 32B entry:", {k: (round(v, 1) if isinstance(v, float) else v) for k, v in d.get("qwen3_32b_tp4", {}).items() if k in ("ms_per_step", "error")})
except Exception as e:
    print("self-launch --gpus 4: unreadable:", e)
PY
tail -3 gpurun_out/${TAG}_bench_tp4_share.err | cut -c1-300
