mkdir -p gpurun_out
timeout 400 python -m pytest tests/test_gpu_rowstream.py -q -m gpu -x 2>&1 | tail -25 > gpurun_out/c8_rowstream_tests.txt; tail -12 gpurun_out/c8_rowstream_tests.txt
timeout 200 python tools/rowstream_bench.py --out gpurun_out/c8_rowstream_bench.json > gpurun_out/c8_rowstream_bench.txt 2>&1; cat gpurun_out/c8_rowstream_bench.txt | cut -c1-420
timeout 300 python -m pytest tests/test_gpu_small_batch.py -q -m gpu -x -s 2>&1 | tail -30 > gpurun_out/c8_small_batch_test.txt; tail -25 gpurun_out/c8_small_batch_test.txt | cut -c1-300
