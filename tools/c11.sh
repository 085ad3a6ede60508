mkdir -p gpurun_out
timeout 700 python -m pytest tests/test_gpu_small_batch.py tests/test_gpu_model.py tests/test_gpu_model_14b.py tests/test_gpu_gemm.py tests/test_gpu_rowops.py tests/test_gpu_reference_driven.py tests/test_gpu_tp_shard.py -q -m gpu -x --durations=6 2>&1 | tail -22 > gpurun_out/c11_tests.txt; tail -16 gpurun_out/c11_tests.txt | cut -c1-200
