#!/bin/bash
# GPU call 1 of round 4: new / changed tests, baseline bench, g3 experiments, self-launch code-path check
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
( time timeout 1200 python -m pytest -q --tb=short -s -p no:cacheprovider \
    tests/test_gpu_tp_shard.py tests/test_gpu_model_14b.py \
    "tests/test_gpu_model.py::test_teacher_forced_parity_vs_cpu_oracle" "tests/test_gpu_model.py::test_baseline_config0_qwen3_0p6b_single_prompt_greedy" \
    tests/test_gpu_attn_prefill.py "tests/test_gpu_tp.py::test_p2p_barrier_timeout_poisons_and_raises" \
    "tests/test_gpu_reference_driven.py::test_reference_driven_radix_chunked_prefill_and_repeats_tiny" \
    tests/test_gpu_gemm.py ) > gpurun_out/c1_tests.log 2>&1
tail -15 gpurun_out/c1_tests.log | cut -c1-250
( time timeout 400 python bench.py --steps 20 --warmup 5 --no-e2e --no-cpu-baseline ) > gpurun_out/c1_bench.json 2> gpurun_out/c1_bench.err
python - <<'PY'
import json
try:
    d = json.loads(open("gpurun_out/c1_bench.json").read().strip().splitlines()[-1])
    print("bench:", round(d["value"]), "tok/s", round(d["ms_per_step"], 3), "ms/step; attn", round(d["roofline"]["us_per_launch"], 1), "us; gemm b2b", round(d["step_roofline"]["gemm_ms_per_step_back_to_back"], 2))
    for r in d["gemm_tune"]["refined_in_graph"]:
        print("  ", r["name"], r["chosen"], r["changed"], r["step_ms"])
except Exception as e:
    print("bench line unreadable:", e)
PY
tail -3 gpurun_out/c1_bench.err | cut -c1-300
( time timeout 300 python tools/g3_exp.py --out gpurun_out/c1_g3_exp.json ) > gpurun_out/c1_g3_exp.log 2>&1
cut -c1-700 gpurun_out/c1_g3_exp.log | tail -12
( time MSGL_BENCH_SHARE_GPU=1 MSGL_GEMM_TUNE=off timeout 400 python bench.py --gpus 2 --steps 3 --warmup 1 ) > gpurun_out/c1_bench_tp2_share.json 2> gpurun_out/c1_bench_tp2_share.err
tail -c 1500 gpurun_out/c1_bench_tp2_share.json | cut -c1-1500; tail -5 gpurun_out/c1_bench_tp2_share.err | cut -c1-300
