mkdir -p gpurun_out
timeout 500 python -m pytest tests/test_gpu_rowstream.py tests/test_gpu_small_batch.py -q -m gpu -x 2>&1 | tail -25 > gpurun_out/c10_tests.txt; tail -12 gpurun_out/c10_tests.txt | cut -c1-300
timeout 300 python tools/rowstream_bench.py --iters 8 --out gpurun_out/c10_rowstream_bench.json 2>&1 | grep -v amdgpu.ids | python -c "
import sys, json
for line in sys.stdin:
    try: r = json.loads(line)
    except Exception: print(line.rstrip()[:300]); continue
    if 'name' in r:
        print(r['name'], r['M'], 'skinny', r['skinny_us'], '| vec', r.get('vec8_us'), r.get('vec16_us'), '| mfma', r.get('mfma8_us'), r.get('mfma16_us'), '| pair/fold', r.get('norm_then_rowstream_us', r.get('act_then_rowstream_us')), r.get('rowstream_with_norm_us', r.get('rowstream_with_act_us')))
    else: print(r)
"
timeout 400 python tools/small_batch_ab.py --out gpurun_out/c10_small_batch_ab.json 2>&1 | grep -v "^\[gemm_tune\]\|amdgpu.ids" | cut -c1-260 | tail -32
