#!/bin/bash
# GPU call 2 of round 4: TP tests, share-GPU self-launch diagnostics, in-step A/Bs
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
( time timeout 900 python -m pytest -q --tb=short -s -p no:cacheprovider tests/test_gpu_tp.py \
    "tests/test_gpu_tp_shard.py::test_rank_shard_full_decode_batch_with_tuned_plans_vs_oracle" \
    "tests/test_gpu_reference_driven.py::test_reference_driven_tp2_through_the_plugin_two_ranks_on_one_gpu" ) > gpurun_out/c2_tests.log 2>&1
tail -8 gpurun_out/c2_tests.log | cut -c1-250
export MSGL_BENCH_SHARE_GPU=1 MSGL_GEMM_TUNE=off MSGL_P2P_SPIN_LIMIT=3000000
for v in "a:" "b:MSGL_BENCH_COMM_SPLIT=0" "c:NOPREFILL"; do
  tag=${v%%:*}; env_=${v#*:}; extra=""
  if [ "$env_" = "NOPREFILL" ]; then env_=""; extra="--no-prefill"; fi
  ( time env $env_ timeout 300 python bench.py --gpus 2 --model qwen3-0.6b --steps 3 --warmup 1 $extra ) > gpurun_out/c2_share_$tag.json 2> gpurun_out/c2_share_$tag.err
  echo "share $tag ($env_ $extra): $(tail -c 400 gpurun_out/c2_share_$tag.json | cut -c1-400)"; grep -m2 "MsglError\|Error:" gpurun_out/c2_share_$tag.err | cut -c1-400
done
unset MSGL_BENCH_SHARE_GPU MSGL_GEMM_TUNE MSGL_P2P_SPIN_LIMIT
( time timeout 500 python tools/step_ab.py --rounds 3 --knobs decode72 lib_o lib_gate_up no_slab_norm --out gpurun_out/c2_step_ab.json ) > gpurun_out/c2_step_ab.log 2>&1
tail -6 gpurun_out/c2_step_ab.log | cut -c1-500
( time MSGL_BENCH_SHARE_GPU=1 MSGL_GEMM_TUNE=off MSGL_P2P_SPIN_LIMIT=6000000 timeout 400 python bench.py --gpus 2 --steps 3 --warmup 1 ) > gpurun_out/c2_share_14b.json 2> gpurun_out/c2_share_14b.err
echo "share 14b: $(tail -c 600 gpurun_out/c2_share_14b.json | cut -c1-600)"; grep -m2 "MsglError\|Error:" gpurun_out/c2_share_14b.err | cut -c1-400
